#!/bin/bash
# Build libkvc_mi355x.so (gfx950 only) in-tree: vllm_kvcompress_amd/libkvc_mi355x.so
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
# KVC_OUT / KVC_EXTRA_FLAGS: experiment builds of the same library (tools/, loaded through KVC_MI355X_LIB)
OUT="${KVC_OUT:-$HERE/../libkvc_mi355x.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value ${KVC_EXTRA_FLAGS:-} \
  "$HERE/kvc_api.hip" "$HERE/kvc_moves.hip" "$HERE/kvc_compact.hip" \
  "$HERE/kvc_schedule.hip" "$HERE/kvc_aggregate.hip" "$HERE/kvc_blockstate.hip" \
  "$HERE/kvc_attention.hip" "$HERE/kvc_prefill_attn.hip" -o "$OUT"
echo "built $OUT"
# the compiled dispatcher binding (host code only; links the library above and torch)
if [ -z "${KVC_OUT:-}" ]; then
  TI="$(python3 -c 'import os, torch; print(os.path.dirname(torch.__file__))')"
  ABI="$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')"
  "$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -x hip "$HERE/kvc_torch_binding.cpp" \
    -I"$TI/include" -I"$TI/include/torch/csrc/api/include" -D_GLIBCXX_USE_CXX11_ABI="$ABI" -DUSE_ROCM \
    -Wno-unused-value -L"$TI/lib" -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -L"$HERE/.." -lkvc_mi355x \
    -Wl,-rpath,'$ORIGIN' -o "$HERE/../libkvc_torch.so"
  echo "built $HERE/../libkvc_torch.so"
fi
# measurement aid for bench.py / tools (never loaded by the package): the bare access pattern of
# the compaction kernel
if [ -z "${KVC_OUT:-}" ]; then
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$HERE/../../tools/kvc_probe.hip" -o "$HERE/../../tools/libkvc_probe.so"
  echo "built tools/libkvc_probe.so"
fi
