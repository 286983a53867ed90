#!/bin/bash
# usage (GPU box): tools/prof_prefill.sh <tag> <K> [lib]  -> per-kernel totals of one fused collection
TAG="$1"; K="$2"; LIB="${3:-}"
export TMPDIR=/tmp
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_pf_$TAG"
mkdir -p "$OUT"
[ -n "$LIB" ] && export KVC_MI355X_LIB="$LIB"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o pf -- python -c "
import sys; sys.path.insert(0,'$REPO'); sys.path.insert(0,'$REPO/tools')
import bench_prefill_fused as b
b.run($K, iters=1)
" > /dev/null 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, sys
out, tag = sys.argv[1], sys.argv[2]
for r in list(csv.DictReader(open(f"{out}/pf_kernel_stats.csv")))[:3]:
    print(tag, r["Name"][:52], r["Calls"], "total_ms/run=%.3f" % (float(r["TotalDurationNs"]) / 2e6))
PY
