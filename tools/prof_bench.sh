#!/bin/bash
# rocprofv3 kernel stats of `bench.py <flags>` (run on the GPU box through gpurun)
#   tools/prof_bench.sh <tag> [bench flags]   -> gpurun_out/<tag>_kernel_stats.csv, <tag>.json
set -u
TAG="${1:-r3}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-probe --no-other-configs --no-native-layout --no-live-traffic --no-engine-cache $*"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $BENCH > "$REPO/gpurun_out/${TAG}.json" 2> "$OUT/stats.log"
python - "$OUT" "$TAG" "$REPO" <<'PY'
import csv, glob, sys, json
out, tag, repo = sys.argv[1:4]
stats = glob.glob(f"{out}/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{repo}/gpurun_out/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "kvc::" in r["Name"] or "rocclr" in r["Name"] or "fill" in r["Name"].lower():
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
            print(r["Name"][:64].ljust(64), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e3, 1))
d = json.load(open(f"{repo}/gpurun_out/{tag}.json"))
print(d["stages_ms"], d.get("S1_schedule"), d["ms_per_step"])
PY
