#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2h"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_schedule_paths.py -m gpu -x -q > "$OUT/pytest_paths.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_paths.log"; tail -3 "$OUT/pytest_paths.log"
export TMPDIR=/tmp
cd /tmp
for cfg in "b64:--batch 64 --steady-cap 4096" "c3:--config c3"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  for lib in default nosort; do
    if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_${name}_$lib" --output-format csv -- python $REPO/bench.py $flags --steps 5 --warmup 1 --no-cpu-baseline --no-adjacent --no-s0 --no-probe > "$OUT/${name}_$lib.json" 2> "$OUT/stats_${name}_$lib.log"
    python - "$OUT" ${name}_$lib <<'PY'
import csv, glob, sys, json
out, lib = sys.argv[1], sys.argv[2]
try:
    r = json.loads([l for l in open(f"{out}/{lib}.json") if l.startswith("{")][-1])
    print(lib, r["S1_schedule"], {k: round(v, 3) for k, v in r["stages_ms"].items()})
except Exception as e:
    print(lib, "no json", e)
for f in glob.glob(f"{out}/stats_{lib}/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "topk" in r["Name"] or "chunk_table" in r["Name"] or "build_keys" in r["Name"]]
    for r in rows:
        print("   %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
