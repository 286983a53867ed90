#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2p"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python $REPO/bench.py --spare-blocks 60 --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-probe > "$OUT/b.json" 2> "$OUT/stats.log"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/stats/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "kvc::" in r["Name"] or "rocclr" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
