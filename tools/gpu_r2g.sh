#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2g"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_schedule_paths.py -m gpu -x -q > "$OUT/pytest_paths.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_paths.log"; tail -5 "$OUT/pytest_paths.log"
export TMPDIR=/tmp
cd /tmp
for lib in default nopad nosort nogather nothing; do
  if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$lib" --output-format csv -- python $REPO/bench.py --batch 64 --steady-cap 4096 --steps 5 --warmup 1 --no-cpu-baseline --no-adjacent --no-s0 --no-probe > "$OUT/b64_$lib.json" 2> "$OUT/stats_$lib.log"
  python - "$OUT" $lib <<'PY'
import csv, glob, sys, json
out, lib = sys.argv[1], sys.argv[2]
try:
    r = json.loads([l for l in open(f"{out}/b64_{lib}.json") if l.startswith("{")][-1])
    print(lib, r["S1_schedule"], {k: round(v, 3) for k, v in r["stages_ms"].items()})
except Exception as e:
    print(lib, "no json", e)
for f in glob.glob(f"{out}/stats_{lib}/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "topk" in r["Name"] or "chunk_table" in r["Name"] or "build_keys" in r["Name"]]
    for r in rows:
        print("   %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
