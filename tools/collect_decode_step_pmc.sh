#!/bin/bash
# HBM traffic of config 3's decode step per kernel from the TCC counters (GPU box, through gpurun): FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes over tools/decode_step.py (MI355X_MICROARCH.md: never mixed with trace domains).
#   tools/collect_decode_step_pmc.sh <tag>   -> gpurun_out/<tag>_decode_step_pmc.json
set -u
TAG="${1:-r4}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_ds_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/decode_step.py --no-parity-gate $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $CMD > "$OUT/ds_fetch.json" 2> "$OUT/fetch.log"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $CMD > /dev/null 2> "$OUT/write.log"
python - "$OUT" "$TAG" "$REPO" <<'PY'
import csv, glob, json, sys, collections
out, tag, repo = sys.argv[1:4]
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    agg = {k: (v[1:] if len(v) >= 3 else v) for k, v in agg.items()}      # (steady state: a kernel's first launch left out)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
line = json.loads(open(f"{out}/ds_fetch.json").read().strip().splitlines()[-1])
slots = line["candidate_slots"]
res = {}
for k in sorted(set(fetch) | set(write)):
    if "kvc::" not in k:
        continue
    fkb, nf = fetch.get(k, (0.0, 0)); wkb, nw = write.get(k, (0.0, 0))
    res[k[:90]] = {"FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb, "launches_seen": max(nf, nw),
                   # wide coalesced streams: FETCH_SIZE counts a 128-byte request at half (MI355X_MICROARCH.md) -> doubled
                   "hbm_bytes_per_launch_stream_doubled": (2.0 * fkb + wkb) * 1024.0}
summary = {"tag": tag, "command": "tools/decode_step.py (config 3, 256 sequences; S0 + S1 + S2 + S3, both variants)",
           "candidate_slots": slots, "kernels": res}
agg = [k for k in res if "aggregate_" in k]
for k in agg:
    summary.setdefault("aggregation", {})[k[:60]] = {
        "hbm_bytes_per_launch": res[k]["hbm_bytes_per_launch_stream_doubled"],
        "algorithmic_bytes": None}
json.dump(summary, open(f"{repo}/gpurun_out/{tag}_decode_step_pmc.json", "w"), indent=1)
for k, v in res.items():
    print(k[:70].ljust(70), round(v["FETCH_SIZE_KB_per_launch"] / 1024, 1), "MB fetch (as counted)", round(v["WRITE_SIZE_KB_per_launch"] / 1024, 1), "MB write")
PY
