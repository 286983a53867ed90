#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE (separate
# passes) of the decode attention at batch 256 x 4097 tokens (tools/bench_attention.py).
# Summaries land in gpurun_out/profiles_<tag>/ ; copy to profiles/.
set -u
TAG="${1:-r1}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/bench_attention.py --only 256,4097"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/astats" --output-format csv -- $CMD > "$OUT/attn_under_rocprof.txt" 2> "$OUT/astats.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/afetch" --output-format csv -- $CMD > /dev/null 2> "$OUT/afetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/awrite" --output-format csv -- $CMD > /dev/null 2> "$OUT/awrite.log"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{out}/astats/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(stats)) if "kvc" in r["Name"]]
with open(f"{out}/{tag}_attention_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"]])
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: v for k, v in agg.items()}
fetch, write = pmc("afetch", "FETCH_SIZE"), pmc("awrite", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "kvc" not in k:
        continue
    # the bench runs record=True first, then record=False: report both halves
    fl, wl = fetch.get(k, []), write.get(k, [])
    def halves(v):
        h = len(v) // 2
        return (sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1)) if v else (0.0, 0.0)
    (f1, f2), (w1, w2) = halves(fl), halves(wl)
    res[k[:100]] = {"launches": len(fl),
                    "FETCH_SIZE_KB_first_half": f1, "FETCH_SIZE_KB_second_half": f2,
                    "WRITE_SIZE_KB_first_half": w1, "WRITE_SIZE_KB_second_half": w2,
                    # gfx950: FETCH_SIZE counts 128 B requests as 64 B for wide coalesced streams
                    "hbm_bytes_first_half": (2.0 * f1 + w1) * 1024.0,
                    "hbm_bytes_second_half": (2.0 * f2 + w2) * 1024.0}
bench = [json.loads(l) for l in open(f"{out}/attn_under_rocprof.txt") if l.startswith("{")]
json.dump({"tag": tag, "workload": "decode attention, 256 seqs x 8 KV heads x 4097 tokens, qpk 4, hd 128, fp16",
           "bench_lines": bench, "kernels": res}, open(f"{out}/{tag}_attention_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
