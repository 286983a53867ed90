#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r2j"
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_dispatch_bindings.py tests/test_gpu_cabi.py "tests/test_gpu_configs.py" -m gpu -x -q > "$OUT/pytest_new.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_new.log"; tail -25 "$OUT/pytest_new.log"
timeout 600 python tools/bench_dispatch_overhead.py > "$OUT/dispatch_overhead.json" 2> "$OUT/dispatch_overhead.err"; cat "$OUT/dispatch_overhead.json" | head -40
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_all.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_all.log"; tail -4 "$OUT/pytest_all.log"
