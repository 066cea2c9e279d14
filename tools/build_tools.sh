#!/bin/bash
# Compile the stand-alone probes / micro-benchmarks under tools/ for gfx950 into tools/bin/ (git-ignored; the binaries
# travel to the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
#   bash tools/build_tools.sh [name ...]        # default: all
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/bin
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -I fish_speech_amd/csrc"
ALL="engine_probe sampler_bench gemv_bench gemv_ksplit_probe l2_prefetch_probe mall_probe mall_resident_probe gemm_bench overlap_bench runahead_bench persist_probe gemv_lds_probe gemv_q8_bench gemv_rows_bench"
for t in ${*:-$ALL}; do
  src=tools/$t.hip
  [ -f "$src" ] || { echo "no such tool: $t"; continue; }
  echo "hipcc $src"
  # (tools that include a kernel translation unit of the library need its host helpers: common.cpp)
  if [ "$t" = gemm_bench ]; then   # includes dualar_gemm.hip; the weight packers live in dualar_kernels.hip
    $HIPCC $FLAGS "$src" fish_speech_amd/csrc/dualar_kernels.hip fish_speech_amd/csrc/common.cpp -o tools/bin/$t 2>&1 | grep -E "error" || true
  elif grep -q 'include "../fish_speech_amd/csrc' "$src"; then
    $HIPCC $FLAGS "$src" fish_speech_amd/csrc/common.cpp -o tools/bin/$t 2>&1 | grep -E "error" || true
  else
    $HIPCC $FLAGS "$src" -o tools/bin/$t 2>&1 | grep -E "error" || true
  fi
done
ls -la tools/bin
