#!/bin/bash
# Build libkvc_mi355x.so (gfx950 only) in-tree: vllm_kvcompress_amd/libkvc_mi355x.so
# The translation units are compiled side by side (one hipcc each) and linked once.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
# KVC_OUT / KVC_EXTRA_FLAGS: experiment builds of the same library (tools/, loaded through KVC_MI355X_LIB)
OUT="${KVC_OUT:-$HERE/../libkvc_mi355x.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJ="$(mktemp -d "${TMPDIR:-/tmp}/kvc_build.XXXXXX")"
trap 'rm -rf "$OBJ"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value ${KVC_EXTRA_FLAGS:-}"
pids=()
for src in kvc_api kvc_moves kvc_compact kvc_schedule kvc_aggregate kvc_blockstate kvc_attention kvc_attention_inst_f16 kvc_attention_inst_bf16 kvc_attention_inst_fp8_f16 kvc_attention_inst_fp8_bf16 kvc_prefill_attn; do
  "$HIPCC" $FLAGS -c "$HERE/$src.hip" -o "$OBJ/$src.o" &
  pids+=($!)
done
# measurement aid for bench.py / tools (never loaded by the package): the bare access pattern of
# the compaction kernel
if [ -z "${KVC_OUT:-}" ]; then
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$HERE/../../tools/kvc_probe.hip" -o "$HERE/../../tools/libkvc_probe.so" &
  pids+=($!)
fi
for pid in "${pids[@]}"; do wait "$pid"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$OUT"
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
  echo "built tools/libkvc_probe.so"
fi
