#!/bin/bash
# round-2 GPU batch E: small-eviction schedule (S1), new bench.py
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2e"
mkdir -p "$OUT"
cd "$REPO"
(rocm-smi --showuniqueid --showclocks --showpower --showmemvendor 2>&1 | head -60) > "$OUT/box.txt"
timeout 900 python -m pytest tests/test_gpu_schedule_paths.py -m gpu -x -q > "$OUT/pytest_paths.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_paths.log"; tail -15 "$OUT/pytest_paths.log"
for path in 0 2 1; do
  KVC_SCHEDULE_PATH=$path timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_schedule_paths.py > "$OUT/pytest_path$path.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_path$path.log"; tail -4 "$OUT/pytest_path$path.log"
done
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 1500 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("value %.3g  step %.3f ms" % (r["value"], r["ms_per_step"]), r["stages_ms"], r["S1_schedule"])
    print({k: r["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "floor_GBps", "frac_of_floor", "pattern_ceiling_GBps", "floor_frac_of_pattern_ceiling")})
    print({k: (v["ms"], v.get("GBps")) for k, v in r["stages_ms_S0"].items()})
    print(r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["stage_seconds"], r["cpu_baseline"]["single_core_port"]["value"])
except Exception as e:
    print("bench parse failed", e)
PY
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-probe"
for cfg in "--batch 16 --steady-cap 4096" "--batch 64 --steady-cap 4096" "--config c3"; do
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
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:24]:
        print("%-70s calls %4s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
