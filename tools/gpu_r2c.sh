#!/bin/bash
# round-2 GPU batch C: why is the compaction kernel below its access pattern's ceiling?
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2c"
mkdir -p "$OUT"
cd "$REPO"
for lg in 17 18 19 20; do
  timeout 200 tools/bin/blockmix_bw $lg > "$OUT/blockmix_$lg.json" 2>> "$OUT/blockmix.err"
done
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent"
run() { echo "== $1" >> "$OUT/sweep.log"; shift; timeout 300 "$@" >> "$OUT/sweep.log" 2>> "$OUT/sweep.err"; }
for lib in default chunk0 chunk0_wpb4; do
  if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
  run "$lib perm 0.5" $B
  run "$lib perm 0.125" $B --keep 0.125
  run "$lib oldest 0.5" $B --metric-shape oldest
  run "$lib decay 0.5" $B --metric-shape decay
  run "$lib perm 0.5 contiguous" $B --contiguous-blocks
  run "$lib perm 0.5 T8192" $B --seq-len 8192
  run "$lib perm 0.5 T16384x4" $B --seq-len 16384 --batch 4
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
        print(tag, "S3 kernel %.3f ms  alg %.0f GB/s  frac %.3f  moved %d" % (
            rf["avg_launch_ms"], rf["achieved"], rf["frac"], r["config"]["moved_slots"]))
PY
# PMC traffic + SQ counters of the default and chunk0 builds (separate passes, counters only)
export TMPDIR=/tmp
cd /tmp
for lib in default chunk0; do
  if [ "$lib" = default ]; then unset KVC_MI355X_LIB; else export KVC_MI355X_LIB="$REPO/tools/bin/libkvc_$lib.so"; fi
  BB="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-adjacent"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_${lib}_fetch" --output-format csv -- $BB > /dev/null 2> "$OUT/pmc_${lib}_fetch.log"
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_${lib}_write" --output-format csv -- $BB > /dev/null 2> "$OUT/pmc_${lib}_write.log"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$OUT/pmc_${lib}_sq" --output-format csv -- $BB > /dev/null 2> "$OUT/pmc_${lib}_sq.log"
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
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
