#!/bin/bash
# SQ counters per kernel of `bench.py --config c3` (GPU box, through gpurun): tools/pmc_c3.sh <tag> "<counters>"
set -u
TAG="${1:-r3}"; CTRS="${2:-SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-adjacent --no-s0 --no-probe --no-other-configs --no-native-layout --no-live-traffic"
timeout 600 rocprofv3 --pmc $CTRS -d "$OUT/pmc" --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc.log"
python - "$OUT" "$TAG" "$REPO" <<'PY'
import csv, glob, sys, collections, json
out, tag, repo = sys.argv[1:4]
f = glob.glob(f"{out}/pmc/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "kvc::" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(f"{repo}/gpurun_out/{tag}_c3_pmc.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
