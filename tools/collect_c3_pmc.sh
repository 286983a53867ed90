#!/bin/bash
# HBM traffic of `bench.py --config c3` per kernel from the TCC counters (GPU box, through gpurun):
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: TCC has 4 counter slots,
# never mixed with trace domains).   tools/collect_c3_pmc.sh <tag> [bench flags]
#   -> gpurun_out/<tag>_c3_pmc.json
set -u
TAG="${1:-r3}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --config c3 --steps 6 --warmup 1 --no-cpu-baseline --no-adjacent --no-s0 --no-probe --no-other-configs --no-native-layout --no-live-traffic --no-parity-gate $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $BENCH > /dev/null 2> "$OUT/write.log"
python - "$OUT" "$TAG" "$REPO" "$*" <<'PY'
import csv, glob, json, sys, collections
out, tag, repo, flags = sys.argv[1:5]
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # (the first launch of a kernel is left out when there are several: the first step of a run sets up the
    # tracked move table / output buffer with their one-time full fills -- the steady state is what is counted)
    agg = {k: (v[1:] if len(v) >= 3 else v) for k, v in agg.items()}
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
line = json.loads(open(f"{out}/bench_fetch.json").read().strip().splitlines()[-1])
slots = line["config"]["candidate_slots"]
S1 = ("stream_", "seq_select_topk", "seq_sums_topk", "seq_prepare", "emit_topk", "scan_pick", "hist_round",
      "build_keys", "clear_chunk_table", "fix_unclaimed", "select_emit", "scan_round", "pick_round", "seq_totals",
      "fallback_general")
res, s1_fetch, s1_write = {}, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    fkb, nf = fetch.get(k, (0.0, 0)); wkb, nw = write.get(k, (0.0, 0))
    if "kvc::" not in k and "fillBuffer" not in k:
        continue
    res[k[:90]] = {"FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb, "launches_seen": max(nf, nw)}
    if any(t in k for t in S1):
        s1_fetch += fkb * 1024; s1_write += wkb * 1024
# the null padding of the output list is a runtime fill kernel: the launches of N x 4 B
fill = [(k, v) for k, v in write.items() if "fillBuffer" in k]
summary = {"tag": tag, "command": f"bench.py --config c3 {flags}", "candidate_slots": slots,
           "S1_kernels_FETCH_SIZE_bytes": s1_fetch, "S1_kernels_WRITE_SIZE_bytes": s1_write,
           "S1_bytes_per_slot_fetch_as_counted": s1_fetch / slots, "S1_bytes_per_slot_write": s1_write / slots,
           "note": "FETCH_SIZE tallies a 64-byte request in full and a 128-byte one at half (profiles/r2o_row_gather.json): "
                   "the coalesced stream of stream_collect is to be doubled, its gathers are not; the output list's null "
                   "padding (4 B/slot) is the runtime's fillBuffer kernel, averaged with the small fills of the step",
           "kernels": res}
json.dump(summary, open(f"{repo}/gpurun_out/{tag}_c3_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
for k, v in res.items():
    print(k[:70].ljust(70), round(v["FETCH_SIZE_KB_per_launch"] / 1024, 1), "MB fetch", round(v["WRITE_SIZE_KB_per_launch"] / 1024, 1), "MB write")
PY
