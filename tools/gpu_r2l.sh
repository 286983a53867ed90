#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2l"
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0"
for sp in 0.02 0.5 3 7 15 30 60; do
  echo "== spare $sp" >> "$OUT/spare.log"
  timeout 600 $B --spare-blocks $sp >> "$OUT/spare.log" 2>> "$OUT/spare.err"
done
echo "== spare 30 oldest" >> "$OUT/spare.log"; timeout 600 $B --spare-blocks 30 --metric-shape oldest >> "$OUT/spare.log" 2>> "$OUT/spare.err"
echo "== spare 30 contiguous" >> "$OUT/spare.log"; timeout 600 $B --spare-blocks 30 --contiguous-blocks >> "$OUT/spare.log" 2>> "$OUT/spare.err"
python - "$OUT/spare.log" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        tag = line.strip()
    elif line.startswith("{"):
        q = json.loads(line); f = q["roofline"]; c = f["pattern_ceiling_GBps"]
        print(tag, "kernel %.3f ms frac %.3f floor %.0f GB/s" % (f["avg_launch_ms"], f["frac"], f["floor_GBps"]), "ceil", c and (round(c["rmw_2R1W"]), round(c["copy_1R1W"])), "step %.3f" % q["ms_per_step"])
PY
