#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2q"
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_all.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_all.log"; tail -4 "$OUT/pytest_all.log"
KVC_SCHEDULE_PATH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > "$OUT/pytest_path1.log" 2>&1; tail -2 "$OUT/pytest_path1.log"
timeout 600 python tools/soak_medium.py 200 > "$OUT/soak_medium.log" 2>&1; tail -1 "$OUT/soak_medium.log"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0"
for sp in 0.02 30 60; do
  echo "== spare $sp" >> "$OUT/spare.log"
  timeout 600 $B --spare-blocks $sp >> "$OUT/spare.log" 2>> "$OUT/spare.err"
done
echo "== c3 general" >> "$OUT/spare.log"; KVC_SCHEDULE_PATH=1 timeout 600 $B --config c3 >> "$OUT/spare.log" 2>> "$OUT/spare.err"
echo "== c4" >> "$OUT/spare.log"; timeout 600 $B --config c4 >> "$OUT/spare.log" 2>> "$OUT/spare.err"
python - "$OUT/spare.log" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        tag = line.strip()
    elif line.startswith("{"):
        q = json.loads(line); f = q["roofline"]; c = f["pattern_ceiling_GBps"]
        print(tag, "value %.3g step %.3f" % (q["value"], q["ms_per_step"]), {k: round(v, 3) for k, v in q["stages_ms"].items()}, "kernel %.3f ms frac %.3f" % (f["avg_launch_ms"], f["frac"]), "ceil", c and (round(c["rmw_2R1W"]), round(c["copy_1R1W"])))
PY
