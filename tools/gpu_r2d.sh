#!/bin/bash
# round-2 GPU batch D: workgroup shape / K write-only variants, next to the access-pattern ceiling of the same box
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2d"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"
KVC_MI355X_LIB="$REPO/tools/bin/libkvc_kwo.so" timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_configs.py -m gpu -x -q > "$OUT/pytest_kwo.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_kwo.log"; tail -3 "$OUT/pytest_kwo.log"
timeout 200 tools/bin/blockmix_bw 19 > "$OUT/blockmix_19.json" 2>> "$OUT/blockmix.err"
grep -E "rmw 2R:1W src->LDS nt|copy 1R:1W src->LDS nt" "$OUT/blockmix_19.json"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent"
run() { echo "== $1" >> "$OUT/sweep.log"; shift; timeout 300 "$@" >> "$OUT/sweep.log" 2>> "$OUT/sweep.err"; }
for lib in default wpb2 wpb5 wpb8 kwo; do
  if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
  run "$lib perm 0.5" $B
  run "$lib perm 0.125" $B --keep 0.125
  run "$lib oldest 0.5" $B --metric-shape oldest
  run "$lib decay 0.5" $B --metric-shape decay
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
        print(tag, "S3 kernel %.3f ms  alg %.0f GB/s  frac %.3f" % (rf["avg_launch_ms"], rf["achieved"], rf["frac"]))
PY
export TMPDIR=/tmp
cd /tmp
for lib in kwo; do
  export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"
  BB="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-adjacent"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_${lib}_fetch" --output-format csv -- $BB > /dev/null 2> "$OUT/pmc_${lib}_fetch.log"
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_${lib}_write" --output-format csv -- $BB > /dev/null 2> "$OUT/pmc_${lib}_write.log"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(f"{out}/pmc_*")):
    if not d.endswith(("fetch", "write", "sq")): continue
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "compact_runs" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        res[d.split("/")[-1]] = {k: sum(v) / len(v) for k, v in agg.items()}
print(json.dumps(res))
PY
