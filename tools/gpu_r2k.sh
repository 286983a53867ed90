#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2k"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_schedule_paths.py -m gpu -x -q > "$OUT/pytest_paths.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_paths.log"; tail -4 "$OUT/pytest_paths.log"
( time timeout 1500 python -m pytest tests/test_gpu_scale.py -k config4 -m gpu -x -q ) > "$OUT/pytest_c4.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_c4.log"; tail -12 "$OUT/pytest_c4.log"
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0"
run() { echo "== $1" >> "$OUT/bench.log"; shift; ( time timeout 900 "$@" ) >> "$OUT/bench.log" 2>> "$OUT/bench.err"; }
run "c4" $B --config c4
run "c3" $B --config c3
run "c3 block_tables" $B --config c3 --pass-block-tables
run "c5" $B --config c5
python - "$OUT/bench.log" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        tag = line.strip()
    elif line.startswith("{"):
        r = json.loads(line)
        rf = r["roofline"]
        print(tag, r["S1_schedule"], "value %.3g step %.3f ms" % (r["value"], r["ms_per_step"]), {k: round(v, 3) for k, v in r["stages_ms"].items()},
              "S3 kernel %.3f ms frac %.3f floor_frac %.3f ceil %s" % (rf["avg_launch_ms"], rf["frac"], rf["frac_of_floor"], rf["pattern_ceiling_GBps"] and round(rf["pattern_ceiling_GBps"]["rmw_2R1W"])))
        print("   ", r["config"]["workload"])
PY
tail -5 "$OUT/bench.err"
