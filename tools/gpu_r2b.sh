#!/bin/bash
# round-2 GPU batch A: parity suite, access-pattern ceiling, compaction sweep (run via gpurun)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2b"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
# (blockmix_bw: measured in batch A)

B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent"
for lib in default wpb2 wpb4; do
  if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
  for shape in perm decay oldest; do
    for keep in 0.5 0.125; do
      echo "== $lib $shape $keep" >> "$OUT/sweep.log"
      timeout 300 $B --metric-shape $shape --keep $keep >> "$OUT/sweep.log" 2>> "$OUT/sweep.err"
    done
  done
done
unset KVC_MI355X_LIB
python - "$OUT/sweep.log" <<'PY'
import json, sys
tag = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        tag = line.strip()
    elif line.startswith("{"):
        r = json.loads(line)
        rf = r["roofline"]
        print(tag, "S3 kernel %.3f ms  alg %.0f GB/s  frac %.3f  step %.3f ms  S1 %.3f S2 %.3f" % (
            rf["avg_launch_ms"], rf["achieved"], rf["frac"], r["ms_per_step"],
            r["stages_ms"]["S1_schedule_evictions"], r["stages_ms"]["S2_schedule_moves"]))
PY
