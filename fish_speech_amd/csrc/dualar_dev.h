// dualar_dev.h -- device helpers shared by the Dual-AR translation units (dualar_kernels / _gemm / _attn / _sample .hip).
#pragma once
#include "common.h"

namespace fmi {

// first k of lane group kg in k-tile j of the packed weight layout (see dualar_kernels.hip: weight packing)
__host__ __device__ inline int packed_k0(int j, int kg, int KT) {  // first k of lane group kg in k-tile j
  if (j < (KT & ~1)) return (j >> 1) * 64 + (kg >> 1) * 32 + (((kg & 1) << 1) + (j & 1)) * 8;
  return j * 32 + kg * 8;
}

__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

}  // namespace fmi
