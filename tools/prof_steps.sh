#!/bin/bash
# usage (on the GPU box): tools/prof_steps.sh <tag> <bench args...>
# rocprofv3 kernel trace of one bench command; prints the per-launch durations of the last step.
TAG="$1"; shift
export TMPDIR=/tmp
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python "$REPO/bench.py" --no-cpu-baseline --no-adjacent --steps 10 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/bench.json").read())
print(tag, "value=%.4g" % d["value"], {k: round(v, 4) for k, v in d["stages_ms"].items()})
rows = [r for r in csv.DictReader(open(f"{out}/{tag}_kernel_trace.csv"))
        if "kvc::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
# one step = from build_keys to the compaction kernel
idx = [i for i, r in enumerate(rows) if "build_keys" in r["Kernel_Name"]]
b = idx[-1] - 2
for r in rows[b:]:
    print("  %-52s %8.2f" % (r["Kernel_Name"][:52], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))
PY
