#!/bin/bash
# round-2 GPU batch F: small-eviction schedule v2 (gather pass), full suite
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2f"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_schedule_paths.py -m gpu -x -q > "$OUT/pytest_paths.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_paths.log"; tail -15 "$OUT/pytest_paths.log"
KVC_SCHEDULE_PATH=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_scale.py -m gpu -x -q > "$OUT/pytest_path2.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_path2.log"; tail -4 "$OUT/pytest_path2.log"
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_schedule_paths.py > "$OUT/pytest_all.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_all.log"; tail -4 "$OUT/pytest_all.log"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-probe"
for cfg in "--batch 16 --steady-cap 4096" "--config c3" "--config c3 --lean"; do
  for lib in new r1path; do
    echo "== $cfg $lib" >> "$OUT/steady.log"
    if [ "$lib" = r1path ]; then export KVC_SCHEDULE_PATH=1; else unset KVC_SCHEDULE_PATH; fi
    timeout 600 $B $cfg >> "$OUT/steady.log" 2>> "$OUT/steady.err"
  done
done
unset KVC_SCHEDULE_PATH
python - "$OUT/steady.log" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        tag = line.strip()
    elif line.startswith("{"):
        r = json.loads(line)
        print(tag, r["S1_schedule"], "step %.3f ms" % r["ms_per_step"], {k: round(v, 3) for k, v in r["stages_ms"].items()}, "cand", r["config"]["candidate_slots"])
PY
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_c3" --output-format csv -- python $REPO/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --no-adjacent --no-s0 --no-probe > "$OUT/c3_under_rocprof.json" 2> "$OUT/stats_c3.log"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/stats_c3/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "kvc::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:20]:
        print("%-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
