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

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace fmi
