#!/bin/bash
# rocprofv3 kernel stats of tools/decode_step.py (config 3 as a whole decode step; run on the GPU box through gpurun)
#   tools/prof_decode_step.sh <tag> [flags]   -> gpurun_out/<tag>_kernel_stats.csv, <tag>.json
set -u
TAG="${1:-ds}"; shift || true
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python $REPO/tools/decode_step.py $* > "$REPO/gpurun_out/${TAG}.json" 2> "$OUT/stats.log"
python - "$OUT" "$TAG" "$REPO" <<'PY'
import csv, glob, sys, json
out, tag, repo = sys.argv[1:4]
stats = glob.glob(f"{out}/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{repo}/gpurun_out/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "kvc::" in r["Name"]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
            print(r["Name"][:72].ljust(72), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e3, 1), round(float(r["MinNs"]) / 1e3, 1))
d = json.load(open(f"{repo}/gpurun_out/{tag}.json"))
print(json.dumps(d["decode_step"]["two_sweeps"]["stages_ms"]), json.dumps(d["decode_step"]["harvest_ahead"]["stages_ms"]))
PY
