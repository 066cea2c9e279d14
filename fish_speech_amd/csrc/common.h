// common.h -- shared helpers for libfishmi (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/fishmi.h"

namespace fmi {

extern thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...);

#define FMI_CHECK_HIP(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fmi::set_error(FMI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                            __FILE__, __LINE__);                                              \
  } while (0)

#define FMI_CHECK(expr)        \
  do {                         \
    int _r = (expr);           \
    if (_r != FMI_OK) return _r; \
  } while (0)

#define FMI_REQUIRE(cond, ...)                              \
  do {                                                      \
    if (!(cond)) return fmi::set_error(FMI_EINVAL, __VA_ARGS__); \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

__host__ __device__ inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
// round-to-nearest-even, NaN preserved (same as torch's c10::BFloat16 conversion)
__host__ __device__ inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
  uint32_t r = c.u + 0x7fffu + ((c.u >> 16) & 1u);
  return (bf16_t)(r >> 16);
}
__host__ __device__ inline float rbf(float f) { return bf2f(f2bf(f)); }  // round through bf16

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16-byte load unit
typedef __attribute__((ext_vector_type(4))) float f32x4;   // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16; // 32x32 accumulator

// ---- lane exchanges inside the VALU (round 6): lane i <- lane i ^ MASK.  Inside a row of 16 lanes these are DPP moves
// (xor 1 / 2 = quad_perm, xor 8 = row_ror:8, xor 4 = row_shl:4 / row_shr:4 selected by lane bit 2); across rows they stay
// ds_bpermute round trips.  Pure data movement: the same partner lanes as __shfl_xor, hence the same bits in any
// reduction built on them.
template <int MASK>
__device__ inline int xor_lane(int v, int lane) {
  if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm(1,0,3,2)
  else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm(2,3,0,1)
  else if constexpr (MASK == 4) {
    const int up = __builtin_amdgcn_mov_dpp(v, 0x104, 0xF, 0xF, true);   // row_shl:4 (lane i <- i + 4)
    const int dn = __builtin_amdgcn_mov_dpp(v, 0x114, 0xF, 0xF, true);   // row_shr:4 (lane i <- i - 4)
    return (lane & 4) ? dn : up;
  } else if constexpr (MASK == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);   // row_ror:8
  else return __shfl_xor(v, MASK, 64);
}
template <int MASK>
__device__ inline float xor_lane_f(float v, int lane) { return __int_as_float(xor_lane<MASK>(__float_as_int(v), lane)); }

// xor trees over the 64 lanes in the order 32, 16, 8, 4, 2, 1 (the order every fixture and parity bound was made with)
__device__ inline float wave_sum(float v) {
  const int lane = (int)(threadIdx.x & 63);
  v += xor_lane_f<32>(v, lane);
  v += xor_lane_f<16>(v, lane);
  v += xor_lane_f<8>(v, lane);
  v += xor_lane_f<4>(v, lane);
  v += xor_lane_f<2>(v, lane);
  v += xor_lane_f<1>(v, lane);
  return v;
}
__device__ inline float wave_max(float v) {
  const int lane = (int)(threadIdx.x & 63);
  v = fmaxf(v, xor_lane_f<32>(v, lane));
  v = fmaxf(v, xor_lane_f<16>(v, lane));
  v = fmaxf(v, xor_lane_f<8>(v, lane));
  v = fmaxf(v, xor_lane_f<4>(v, lane));
  v = fmaxf(v, xor_lane_f<2>(v, lane));
  v = fmaxf(v, xor_lane_f<1>(v, lane));
  return v;
}

// ---- DPP reductions: 16-lane rows reduce inside the VALU (no ds_bpermute round trips) ----
template <int CTRL>
__device__ inline float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ inline int dpp_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
// all-reduce (sum) over aligned groups of N lanes, N in {4, 8, 16}; every lane gets the group total
template <int N>
__device__ inline float group_allsum(float v) {
  v += dpp_f<0xB1>(v);                 // quad_perm(1,0,3,2)
  v += dpp_f<0x4E>(v);                 // quad_perm(2,3,0,1)
  if (N >= 8) v += dpp_f<0x141>(v);    // row_half_mirror
  if (N >= 16) v += dpp_f<0x140>(v);   // row_mirror
  return v;
}
// full-wave reductions: rows by DPP, the four row totals by scalar broadcasts (result is wave-uniform)
__device__ inline float wave_sum_dpp(float v) {
  v = group_allsum<16>(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return ((r0 + r1) + r2) + r3;
}
__device__ inline int wave_sum_dpp_i(int v) {
  v += dpp_i<0xB1>(v);
  v += dpp_i<0x4E>(v);
  v += dpp_i<0x141>(v);
  v += dpp_i<0x140>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}
__device__ inline uint32_t wave_max_dpp_u(uint32_t v) {
  v = max(v, (uint32_t)dpp_i<0xB1>((int)v));
  v = max(v, (uint32_t)dpp_i<0x4E>((int)v));
  v = max(v, (uint32_t)dpp_i<0x141>((int)v));
  v = max(v, (uint32_t)dpp_i<0x140>((int)v));
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace fmi
