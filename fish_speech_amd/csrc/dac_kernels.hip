// dac_kernels.hip -- hand-written gfx950 kernels of the modded-DAC codec (fp32, channel-major).
//
// Reference semantics (fish_speech/models/dac/): CausalConvNet / CausalTransConvNet
// modded_dac.py:521-588, ResidualUnit 599-620, Encoder/Decoder 670-801, WindowLimitedTransformer
// 349-439 (+ Transformer/Attention/FeedForward/RMSNorm/LayerScale 97-346), ConvNeXtBlock rvq.py:129-191,
// DownsampleResidualVectorQuantize rvq.py:293-366, third-party Snake1d / VectorQuantize.
//
// The codec is compute-bound (727 GMAC per 10 s utterance, 94 % in the decoder's dilated convs,
// arithmetic intensity ~200 flop/byte).  Its arithmetic is fp32 in the reference's codec CLI, and the
// parity bar is waveform RMS <= 1e-4, so the convolutions run as implicit GEMMs on the fp32-input
// matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, 157 TFLOP/s peak = the fp32
// vector peak, reached with far fewer issue slots than v_fma).  One kernel serves dilated, strided
// and transposed convolutions and all linear layers (k = 1): M = output channels, N = time,
// reduction = input channels x taps; input tile (with fused Snake) and weight tile are staged
// through LDS, 8 input channels at a time.
#include <stdlib.h>

#include "dac_kernels.h"

namespace fmi {

// =====================================================================================
// weight packing: w[phase][tap][ci_pad/8][co_pad][8]
// =====================================================================================
// One (tap, 8-channel group) of a 32*MT-row tile is a contiguous block of MT KiB, so the LDS-DMA copies it
// in linear 1 KiB pieces, and the 8 channels of one output row are adjacent: a lane fetches the four
// reduction steps of its MFMA A operand with ONE ds_read_b128.  The two 16-byte halves of a row (channels
// 0-3 / 4-7) are swapped on rows with bit 3 set, which makes those reads bank-conflict free (ds_read_b128 is
// serviced in 16-lane groups that must cover 64 distinct banks; MI355X_MICROARCH.md LDS table).
__host__ __device__ inline int64_t conv_w_index(int tapg, int ci, int co, int cin_pad, int cout_pad) {
  return ((((int64_t)tapg * (cin_pad >> 3) + (ci >> 3)) * cout_pad + co) << 3) +
         ((((ci >> 2) ^ (co >> 3)) & 1) << 2) + (ci & 3);
}

__global__ void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int k,
                                 int cin_pad, int cout_pad) {
  const int64_t total = (int64_t)k * cin_pad * cout_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout_pad);
    const int ci = (int)((i / cout_pad) % cin_pad);
    const int tap = (int)(i / ((int64_t)cout_pad * cin_pad));
    float v = 0.f;
    if (co < cout && ci < cin) v = src[((int64_t)co * cin + ci) * k + tap];
    dst[conv_w_index(tap, ci, co, cin_pad, cout_pad)] = v;
  }
}

int launch_pack_conv(const float* w_src, float* dst, int cout, int cin, int k, int cin_pad, int cout_pad,
                     hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(1024), dim3(256), 0, s, w_src, dst, cout, cin, k, cin_pad, cout_pad);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// rows [co_off, co_off + cout) of a stacked k=1 weight (src [cout][cin]); nothing else is touched
__global__ void pack_conv_part_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin,
                                      int cin_pad, int cout_pad, int co_off) {
  const int64_t total = (int64_t)cout * cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout), ci = (int)(i / cout);
    dst[conv_w_index(0, ci, co_off + co, cin_pad, cout_pad)] = src[(int64_t)co * cin + ci];
  }
}

int launch_pack_conv_part(const float* w_src, float* dst, int cout, int cin, int cin_pad, int cout_pad, int co_off,
                          hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_part_kernel, dim3(1024), dim3(256), 0, s, w_src, dst, cout, cin, cin_pad, cout_pad,
                     co_off);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// ConvTranspose1d weight [cin][cout][k], stride s, k = taps*s:  w[phase j][tap m][ci][co] = src[ci][co][j + m*s]
__global__ void pack_convtr_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int k,
                                   int stride, int cin_pad, int cout_pad) {
  const int taps = k / stride;
  const int64_t total = (int64_t)stride * taps * cin_pad * cout_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout_pad);
    const int ci = (int)((i / cout_pad) % cin_pad);
    const int m = (int)((i / ((int64_t)cout_pad * cin_pad)) % taps);
    const int j = (int)(i / ((int64_t)cout_pad * cin_pad * taps));
    float v = 0.f;
    if (co < cout && ci < cin) v = src[((int64_t)ci * cout + co) * k + j + m * stride];
    dst[conv_w_index(j * taps + m, ci, co, cin_pad, cout_pad)] = v;
  }
}

int launch_pack_convtr(const float* w_src, float* dst, int cin, int cout, int k, int stride, int cin_pad,
                       int cout_pad, hipStream_t s) {
  FMI_REQUIRE(k % stride == 0, "convtr: kernel %d must be a multiple of stride %d", k, stride);
  hipLaunchKernelGGL(pack_convtr_kernel, dim3(1024), dim3(256), 0, s, w_src, dst, cin, cout, k, stride, cin_pad,
                     cout_pad);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// the implicit-GEMM convolution on fp32 matrix cores
// =====================================================================================

__device__ inline float snake_f(float v, float alpha) {
  // dac.nn.layers.snake: x + (alpha + 1e-9)^-1 * sin(alpha*x)^2.  Hardware sine (v_sin_f32 after the 1/(2 pi)
  // range reduction): absolute error ~1e-6, far inside the 1e-4 waveform RMS bar (5e-7 measured end to end).
  const float sn = __sinf(alpha * v);
  return v + (1.0f / (alpha + 1e-9f)) * (sn * sn);
}

__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Work-group = 4 waves; output tile = (MT*32 channels) x (4 waves * NT*32 columns); every wave multiplies the
// whole weight tile with its own columns.  Per step G groups of 8 input channels are staged:
//   Ws [taps][G][CO_T][8]  by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR/VALU)
//   Xs [G][wx][8]          input columns x 8 channels, Snake applied on the way in, 16-byte writes
// both with the half-swap described at conv_w_index.  The MFMA loop then needs (MT + NT) ds_read_b128 per
// 4*MT*NT v_mfma_f32_32x32x2_f32: the first version of this kernel (b32 operand reads, rolled loop, 11 other
// instructions per MFMA) was bound by the SIMD's instruction issue, not by the matrix pipe or the LDS
// (profiles/r01_pmc_conv.txt).  Reduction step s of a group pairs channels (s, 4+s): lane half 0 / 1.
template <int MT, int NT, int G>
__global__ __launch_bounds__(256, (MT == 4 && NT == 1) ? 4 : (MT * NT <= 4 ? 3 : 2)) void conv_mfma_kernel(ConvArgs a, int ncols, int wx, int tap_off0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CO_T = MT * 32, TT = 4 * NT * 32;
  const int taps = a.w.taps;
  float* Ws = smem;                           // first: 1 KiB-aligned pieces for the DMA
  const int nw = taps * G * CO_T * 8;
  float* Xs = smem + nw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile bookkeeping stays on the SALU
  const int li = lane & 31, lk = lane >> 5;
  const int q0 = blockIdx.x * TT;
  const int co0 = blockIdx.y * CO_T;
  const int b = blockIdx.z / a.w.phases, phase = blockIdx.z % a.w.phases;
  const int c0 = q0 * a.x_stride + a.tap_base - tap_off0;  // input column of LDS column 0
  const float* xb = a.x + (int64_t)b * a.w.cin * a.lin;
  const int cgs = a.w.cin_pad >> 3;
  const float* wph = a.w.w + (((int64_t)phase * taps * cgs * a.w.cout_pad + co0) << 3);
  const bool do_snake = a.snake_alpha != nullptr;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int qw = wave * NT * 32;       // this wave's first column inside the tile
  const int n_rows = taps * G;         // (tap, group) rows of the weight tile, MT KiB each
  const int nblk = (wx + 63) >> 6;     // 64-column blocks of the input tile
  // A operand: row li of 32-row block i, half lk (swapped on rows with bit 3 set)
  const char* a_lane = reinterpret_cast<const char*>(Ws) + li * 32 + (((lk ^ (li >> 3)) & 1) << 4);

  for (int cg0 = 0; cg0 < cgs; cg0 += G) {
    // ---- weight tile: row = tap*G + g, MT pieces of 1 KiB per row, pieces dealt round-robin to the waves
    for (int p = wave; p < n_rows * MT; p += 4) {
      const int row = p / MT, pc = p - row * MT;
      const int tp = row / G, g = row - tp * G;
      const char* gsrc = reinterpret_cast<const char*>(wph + (((int64_t)tp * cgs + cg0 + g) * a.w.cout_pad << 3)) +
                         pc * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)(reinterpret_cast<char*>(Ws) + p * 1024), 16, 0, 0);
    }
    // ---- input tile: a wave takes (group, half, 64-column block) items; lanes run along time (coalesced)
    for (int it = wave; it < 2 * G * nblk; it += 4) {
      const int pair = it / nblk, cb = it - pair * nblk;
      const int g = pair >> 1, h = pair & 1;
      const int c = cb * 64 + lane;
      const int ci = (cg0 + g) * 8 + h * 4;   // wave-uniform
      const int col = c0 + c;
      const int nval = a.w.cin - ci;          // real channels among the 4 (padding channels read as zero)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < wx) {
        if (col >= 0 && col < a.lin) {
          const float* xp = xb + (int64_t)ci * a.lin + col;
          if (nval >= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xp[(int64_t)e * a.lin];
            if (do_snake) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = snake_f(v[e], a.snake_alpha[ci + e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < nval) {
                const float t = xp[(int64_t)e * a.lin];
                v[e] = do_snake ? snake_f(t, a.snake_alpha[ci + e]) : t;
              }
          }
        }
        *reinterpret_cast<f32x4*>(Xs + ((g * wx + c) << 3) + (((h ^ (c >> 3)) & 1) << 2)) = v;
      }
    }
    __syncthreads();  // also drains the LDS-DMA (vmcnt(0) is part of the barrier's fence)
    // ---- multiply
    for (int tp = 0; tp < taps; ++tp) {
      const int xoff = tap_off0 + tp * a.tap_step;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(a_lane + ((tp * G + g) * CO_T + i * 32) * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = (qw + j * 32 + li) * a.x_stride + xoff;
          bf[j] = *reinterpret_cast<const f32x4*>(Xs + ((g * wx + c) << 3) + (((lk ^ (c >> 3)) & 1) << 2));
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][st], bf[j][st], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: out = res + gamma * act(acc + bias); lanes run along time (coalesced).  A branchy version
  // (per-element tests of bias / gamma / res / row bound) cost 2-3 serialized memory round trips per output
  // element and dominated the k = 1 layers: bias and gamma of the tile's rows go through LDS (the operand
  // tiles are dead after the last barrier), residual loads use clamped always-valid 32-bit offsets and are
  // issued 16 at a time.
  float* sb = smem;          // [CO_T] bias (0 when absent), then [CO_T] gamma (1 when absent)
  float* sg = smem + CO_T;
  const int co_last = a.w.cout - 1;
  if (tid < CO_T) {
    const int co = min(co0 + tid, co_last);
    sb[tid] = a.w.bias ? a.w.bias[co] : 0.f;
    sg[tid] = a.gamma ? a.gamma[co] : 1.f;
  }
  __syncthreads();
  float* ob = a.out + (int64_t)b * a.w.cout * a.lout;
  const float* rb = a.res ? a.res + (int64_t)b * a.w.cout * a.lout : ob;  // same layout as out
  const bool has_res = a.res != nullptr;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = q0 + qw + j * 32 + li;
      const bool live = q < ncols;
      const int col = (live ? q : 0) * a.out_stride + phase;
      const int row0 = i * 32 + 4 * lk;   // tile row of r = 0; r -> row0 + 8*(r>>2) + (r&3)
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = min(co0 + row0 + 8 * (r >> 2) + (r & 3), co_last);
        rv[r] = has_res ? rb[co * a.lout + col] : 0.f;
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb + row0 + 8 * r4);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sg + row0 + 8 * r4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = r4 * 4 + e;
          float v = acc[i][j][r] + b4[e];
          if (a.act == ACT_GELU) v = gelu_f(v);
          v = v * g4[e] + rv[r];
          const int co = co0 + row0 + 8 * r4 + e;
          if (live && co <= co_last) ob[co * a.lout + col] = v;
        }
      }
    }
}

// =====================================================================================
// the same implicit GEMM on the bf16 matrix cores, fp32-class through operand splitting
// =====================================================================================
//
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md).  An fp32 value is the exact sum of three
// bf16 pieces (8 + 8 + 8 significant bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), so
//   x * w = (xh + xm + xl)(wh + wm + wl) = xh wh + xh wm + xm wh + xm wm + xh wl + xl wh   (+ terms < 2^-24 |x w|)
// is six v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation) per 16 reduction steps: 6 x 32 cycles
// against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same 16 steps -- 2.7x the matrix throughput at the accuracy
// of fp32 arithmetic (the dropped terms are below the rounding of an fp32 product).  PLANES = 2 keeps three products
// (~2^-16), PLANES = 1 one (operands rounded to bf16 = what torch.autocast(bfloat16) computes for conv / linear).
//
// Same tiling, LDS-DMA weight staging and epilogue as conv_mfma_kernel; a reduction group is 16 input channels per
// plane, and the byte layout of one (tap, group, plane) weight tile / input tile equals the fp32 kernel's 8-channel
// tile (32 bytes per row, 16-byte half per lane, same half swap), so the operand addressing carries over.
__device__ inline uint32_t cvt_pk_bf16_f32(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// two fp32 values -> packed bf16 pairs of the three planes
__device__ inline void split3_pk(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf16_f32(x0, x1);
  float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16_f32(r0, r1);
  r0 -= __uint_as_float(m << 16);
  r1 -= __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16_f32(r0, r1);
}

// fp16 two-term split with a scaled low part: x = hi + lo, hi = f16(x), lo16 = f16((x - hi) * 2048).  The products
// hi*hi accumulate in one fp32 accumulator, hi*lo16 + lo16*hi in a second one that is folded in with weight 2^-11:
// three v_mfma_f32_32x32x16_f16 per 16 reduction steps, and what is dropped (lo*lo, the f16 rounding of lo) is below
// 2^-22 of a product -- fp32-class, at half the matrix work of the six-product bf16 scheme (f16 carries 11 significant
// bits against bf16's 8).  Range: |x| < 65504; the scaling keeps lo16 out of the f16 subnormals.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_LO_SCALE = 2048.0f;
constexpr float F16_MAX = 65504.0f;
// Sticky: *ovf is set when a value outside the fp16 range reached the split (fmi_dac_fp16_overflow reads and clears
// it).  The split SATURATES there (both terms clamped to the largest finite fp16, one v_med3_f32 each) instead of
// producing inf - inf = NaN that would spread through the whole waveform; the caller can then redo the call with
// precision 0.  The word belongs to the codec HANDLE whose entry point launched the kernel (round 5; a process-wide flag
// let two handles, or two request threads on two handles, consume each other's overflow): the entry point names it with
// set_f16_overflow_target() under the handle's mutex, the launch wrappers below pass it on.  Launches outside any
// entry point (tools/, benches) raise g_f16_overflow_unowned.
__device__ int g_f16_overflow_unowned = 0;
static thread_local int* t_f16_ovf = nullptr;
void set_f16_overflow_target(int* dev_word) { t_f16_ovf = dev_word; }
static int* f16_overflow_target() {
  if (t_f16_ovf) return t_f16_ovf;
  static int* unowned = []() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_f16_overflow_unowned)) != hipSuccess) p = nullptr;
    return (int*)p;
  }();
  return unowned;
}
__device__ inline void split2_f16_pk(float x0, float x1, uint32_t& h, uint32_t& l, int* __restrict__ ovf) {
  if (fmaxf(fabsf(x0), fabsf(x1)) > F16_MAX) *ovf = 1;
  x0 = __builtin_amdgcn_fmed3f(x0, -F16_MAX, F16_MAX);
  x1 = __builtin_amdgcn_fmed3f(x1, -F16_MAX, F16_MAX);
  const f16x2 hv = {(_Float16)x0, (_Float16)x1};
  const float r0 = __builtin_amdgcn_fmed3f((x0 - (float)hv[0]) * F16_LO_SCALE, -F16_MAX, F16_MAX);
  const float r1 = __builtin_amdgcn_fmed3f((x1 - (float)hv[1]) * F16_LO_SCALE, -F16_MAX, F16_MAX);
  const f16x2 lv = {(_Float16)r0, (_Float16)r1};
  h = *reinterpret_cast<const uint32_t*>(&hv);
  l = *reinterpret_cast<const uint32_t*>(&lv);
}

__host__ __device__ inline int64_t conv_wb_index(int tapg, int ci, int co, int plane, int cin_pad16, int cout_pad) {
  return (((((int64_t)tapg * (cin_pad16 >> 4) + (ci >> 4)) * 3 + plane) * cout_pad + co) << 4) +
         ((((ci >> 3) ^ (co >> 3)) & 1) << 3) + (ci & 7);
}

__global__ void split_conv_planes_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb, int tapgroups,
                                         int cin_pad, int cin_pad16, int cout_pad, int* __restrict__ ovf) {
  const int64_t total = (int64_t)tapgroups * (cin_pad16 >> 1) * cout_pad;   // one thread per channel PAIR
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout_pad);
    const int cp = (int)((i / cout_pad) % (cin_pad16 >> 1));
    const int tg = (int)(i / ((int64_t)cout_pad * (cin_pad16 >> 1)));
    const int ci = cp * 2;
    const float x0 = ci < cin_pad ? w[conv_w_index(tg, ci, co, cin_pad, cout_pad)] : 0.f;
    const float x1 = ci + 1 < cin_pad ? w[conv_w_index(tg, ci + 1, co, cin_pad, cout_pad)] : 0.f;
    // slot 0: bf16(w) (the autocast mode's operand); slots 1, 2: the fp16 split (hi, scaled lo)
    uint32_t h, l;
    split2_f16_pk(x0, x1, h, l, ovf);
    *reinterpret_cast<uint32_t*>(wb + conv_wb_index(tg, ci, co, 0, cin_pad16, cout_pad)) = cvt_pk_bf16_f32(x0, x1);
    *reinterpret_cast<uint32_t*>(wb + conv_wb_index(tg, ci, co, 1, cin_pad16, cout_pad)) = h;
    *reinterpret_cast<uint32_t*>(wb + conv_wb_index(tg, ci, co, 2, cin_pad16, cout_pad)) = l;
  }
}

int launch_split_conv_planes(const float* w_packed, bf16_t* wb, int tapgroups, int cin_pad, int cin_pad16, int cout_pad,
                             hipStream_t s) {
  FMI_REQUIRE(cin_pad16 % 16 == 0 && cin_pad16 >= cin_pad && cout_pad % 32 == 0, "split planes: bad padding");
  hipLaunchKernelGGL(split_conv_planes_kernel, dim3(2048), dim3(256), 0, s, w_packed, wb, tapgroups, cin_pad, cin_pad16,
                     cout_pad, f16_overflow_target());
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// TAPS = 7: the tap loop is unrolled (operand reads of the next tap overlap the MFMAs of the current one).
// TC > 0: the weight tile is brought in TC taps at a time (k = 7 as 4 + 3): a third of the LDS per work-group, so two
// or three work-groups share a CU and one's staging phase hides behind another's matrix phase.
// PH (transposed convs): the MT row tiles of a work-group are MT consecutive PHASES of the same 32 output channels
// instead of 32 * MT channels of one phase.  A lane then holds, for its input column q, the output columns
// q * stride + phase0 .. + MT - 1 of a channel: it stores them as one vector, and the operand planes of MT consecutive
// output columns -- with one phase per work-group every work-group wrote every stride-th float / 32-byte plane entry
// of lines that stride work-groups shared.  Same products in the same order per output element.
template <int MT, int NT, int G, int NP, int TAPS, int TC, bool PH = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_bf16_kernel(ConvArgs a, int ncols, int wx, int tap_off0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CO_T = MT * 32, TT = 4 * NT * 32;
  constexpr int CO_W = PH ? 32 : CO_T;        // output channels per work-group
  const int taps = TAPS > 0 ? TAPS : a.w.taps;
  const int wt = TC > 0 ? TC : taps;                        // taps resident in Ws at a time
  char* Ws = reinterpret_cast<char*>(smem);                 // [wt][G][NP][CO_T] rows of 32 bytes
  const int nw_bytes = wt * G * NP * CO_T * 32;
  char* Xs = Ws + nw_bytes;                                 // [G][NP][wx] columns of 32 bytes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int q0 = blockIdx.x * TT;
  const int co0 = blockIdx.y * CO_W;
  const int zph = PH ? a.w.phases / MT : a.w.phases;          // phase groups per utterance
  const int b = blockIdx.z / zph, phase = (blockIdx.z % zph) * (PH ? MT : 1);
  const int c0 = q0 * a.x_stride + a.tap_base - tap_off0;
  const int xld = a.x_ld ? a.x_ld : a.lin;                    // (streaming: row strides / left context, see ConvArgs)
  const int old_ = a.out_ld ? a.out_ld : a.lout, opld = a.outp_ld ? a.outp_ld : a.lout;
  const int xlo = -a.x_left;
  const float* xb = a.x + (int64_t)b * a.w.cin * xld;
  const int cgs = a.w.cin_pad16 >> 4;
  // plane 0 of (phase, tap 0, group 0), rows from co0
  const bf16_t* wph = a.w.wb + ((((int64_t)phase * taps * cgs * 3) * a.w.cout_pad + co0) << 4);
  const int64_t ph_bytes = ((int64_t)taps * cgs * 3 * a.w.cout_pad) << 5;   // one phase of the packed weight, in bytes
  const bool do_snake = a.snake_alpha != nullptr;

  // NP = 1: bf16 operands (autocast), one product.  NP = 2: fp16 split, acc = hi*hi, acx = hi*lo16 + lo16*hi.
  constexpr int SLOT0 = NP == 2 ? 1 : 0;       // first weight slot of this mode (see split_conv_planes_kernel)
  f32x16 acc[MT][NT], acx[NP == 2 ? MT : 1][NP == 2 ? NT : 1];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if (NP == 2) acx[i][j][r] = 0.f;
      }

  const int qw = wave * NT * 32;
  const int nblk = (wx + 63) >> 6;
  constexpr int SU = G >= 8 ? 8 : (G >= 4 ? 4 : 2);
  const char* a_lane = Ws + li * 32 + (((lk ^ (li >> 3)) & 1) << 4);
  auto fetch_w = [&](int cg0, int t0, int nt) {   // (tap, group, plane) rows of taps [t0, t0 + nt), MT KiB each
    for (int p = wave; p < nt * G * NP * MT; p += 4) {
      const int row = p / MT, pc = p - row * MT;
      const int pl = row % NP, tg = row / NP;
      const int tp = t0 + tg / G, g = tg % G;
      const char* gsrc = reinterpret_cast<const char*>(wph + (((((int64_t)tp * cgs + cg0 + g) * 3 + SLOT0 + pl) * a.w.cout_pad) << 4)) +
                         (PH ? pc * ph_bytes : (int64_t)pc * 1024) + lane * 16;   // row tile pc: next 32 rows, or next phase
      __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)(Ws + p * 1024), 16, 0, 0);
    }
  };

  // (A register prefetch of the next step's input tile before the matrix phase was measured and is NOT used: 111.9 vs
  // 106.2 ms per batch-8 decode -- the waves wait on the barrier-separated phases, not on that load latency.)
  for (int cg0 = 0; cg0 < cgs; cg0 += G) {
    fetch_w(cg0, 0, wt < taps ? wt : taps);
    if (a.xp) {   // operand planes written by the producing conv: a plain copy (two lanes per 32-byte column)
      const char* xpb = reinterpret_cast<const char*>(a.xp) + (int64_t)b * cgs * NP * xld * 32;
      const int nb32 = (wx + 31) >> 5;
      for (int it = wave; it < G * NP * nb32; it += 4) {
        const int gp = it / nb32, cb = it - gp * nb32;
        const int g = gp / NP, pl = gp - g * NP;
        const int c = cb * 32 + (lane >> 1), h = lane & 1;
        const int col = c0 + c;
        if (c < wx) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (col >= xlo && col < a.lin)
            v = *reinterpret_cast<const u32x4*>(xpb + (((int64_t)(cg0 + g) * NP + pl) * xld + col) * 32 + h * 16);
          *reinterpret_cast<u32x4*>(Xs + ((int64_t)(g * NP + pl) * wx + c) * 32 + (((h ^ (c >> 3)) & 1) << 4)) = v;
        }
      }
    } else
    // input tile: item = (group, 8-channel half, 64-column block); a lane owns one column, splits its 8 channels.
    // SU items per wave are in flight together (all their loads issue before the first conversion): a k = 1 layer
    // of the codec transformer is a chain of load round trips, one per item, when they go one at a time.
    for (int it0 = wave; it0 < 2 * G * nblk; it0 += 4 * SU) {
      float v[SU][8];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int it = it0 + 4 * u;
        const int pair = it / nblk, cb = it - pair * nblk;
        const int c = cb * 64 + lane;
        const int ci = (cg0 + (pair >> 1)) * 16 + (pair & 1) * 8;   // wave-uniform
        const int col = c0 + c;
        const int nval = a.w.cin - ci;           // real channels among the 8 (padding reads as zero)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
        if (it < 2 * G * nblk && c < wx && col >= xlo && col < a.lin && nval > 0) {
          const float* xp = xb + (int64_t)ci * xld + col;
          if (nval >= 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = xp[(int64_t)e * xld];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e < nval) v[u][e] = xp[(int64_t)e * xld];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int it = it0 + 4 * u;
        const int pair = it / nblk, cb = it - pair * nblk;
        const int g = pair >> 1, h = pair & 1;
        const int c = cb * 64 + lane;
        const int ci = (cg0 + g) * 16 + h * 8;
        const int col = c0 + c;
        const int nval = a.w.cin - ci;
        if (it < 2 * G * nblk && c < wx) {
          if (do_snake && col >= xlo && col < a.lin) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e < nval) v[u][e] = snake_f(v[u][e], a.snake_alpha[ci + e]);
          }
          u32x4 ph, pl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (NP == 2) {
              uint32_t hh, ll;
              split2_f16_pk(v[u][2 * e], v[u][2 * e + 1], hh, ll, a.ovf);
              ph[e] = hh; pl[e] = ll;
            } else {
              ph[e] = cvt_pk_bf16_f32(v[u][2 * e], v[u][2 * e + 1]);
            }
          }
          char* dst = Xs + ((int64_t)(g * NP) * wx + c) * 32 + (((h ^ (c >> 3)) & 1) << 4);
          *reinterpret_cast<u32x4*>(dst) = ph;
          if (NP == 2) *reinterpret_cast<u32x4*>(dst + (int64_t)wx * 32) = pl;
        }
      }
    }
    __syncthreads();   // drains the LDS-DMA too (vmcnt(0) is part of the barrier's fence)
    // products of (resident tap slot tl, tap offset xoff, channel group g of the step)
    auto mm = [&](int tl, int xoff, int g) {
      bf16x8 af[NP][MT], bf[NP][NT];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[pl][i] = *reinterpret_cast<const bf16x8*>(a_lane + (((tl * G + g) * NP + pl) * CO_T + i * 32) * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = (qw + j * 32 + li) * a.x_stride + xoff;
          bf[pl][j] = *reinterpret_cast<const bf16x8*>(Xs + ((int64_t)(g * NP + pl) * wx + c) * 32 + (((lk ^ (c >> 3)) & 1) << 4));
        }
      }
      if constexpr (NP == 2) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(&af[0][i]), a1 = *reinterpret_cast<const f16x8*>(&af[1][i]);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(&bf[0][j]), b1 = *reinterpret_cast<const f16x8*>(&bf[1][j]);
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acx[i][j], 0, 0, 0);
            acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acx[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
      }
    };
    if constexpr (TAPS == 0 && TC == 0 && G > 1) {
      // several channel groups per step with more than one tap (the transposed convs): group-major, so that an
      // output element accumulates (group, tap) in the order the one-group-per-step kernel does -- same bits
#pragma unroll
      for (int g = 0; g < G; ++g)
        for (int tp = 0; tp < taps; ++tp) mm(tp, tap_off0 + tp * a.tap_step, g);
    } else {
#pragma unroll(TAPS > 0 ? TAPS : 1)
      for (int tp = 0; tp < taps; ++tp) {
        const int tl = TC > 0 ? tp % TC : tp;      // tap's place in the resident part of the weight tile
        if (TC > 0 && tp > 0 && tl == 0) {         // next part: everybody is done with the resident one
          __syncthreads();
          fetch_w(cg0, tp, taps - tp < TC ? taps - tp : TC);
          __syncthreads();
        }
        const int xoff = tap_off0 + tp * a.tap_step;
#pragma unroll
        for (int g = 0; g < G; ++g) mm(tl, xoff, g);
      }
    }
    __syncthreads();
  }

  // ---- epilogue (as conv_mfma_kernel): out = res + gamma * act(acc + bias)
  float* sb = smem;
  float* sg = smem + CO_T;
  const int co_last = a.w.cout - 1;
  if (tid < CO_W) {
    const int co = min(co0 + tid, co_last);
    sb[tid] = a.w.bias ? (NP == 1 ? rbf(a.w.bias[co]) : a.w.bias[co]) : 0.f;
    sg[tid] = a.gamma ? a.gamma[co] : 1.f;
  }
  __syncthreads();
  float* ob = a.out ? a.out + (int64_t)b * a.w.cout * old_ : nullptr;
  const float* rb = a.res ? a.res + (int64_t)b * a.w.cout * old_ : ob;
  const bool has_res = a.res != nullptr;
  auto finish = [&](float accv, float acxv, float bias, float gam, float resv) {
    float v = (NP == 2 ? accv + acxv * (1.0f / F16_LO_SCALE) : accv) + bias;
    if (NP == 1) v = rbf(v);   // autocast(bf16): the conv / linear returns bf16 (bias already bf16-rounded)
    if (a.act == ACT_GELU) {
      v = gelu_f(v);
      if (NP == 1) v = rbf(v);  // GELU of a bf16 tensor is a bf16 tensor
    }
    return v * gam + resv;      // LayerScale / ConvNeXt gamma (fp32 parameter) and the residual promote to fp32
  };
  // operand planes for the consumer: Snake of ITS alpha, split, 4 channels = 8 bytes
  auto put_planes = [&](const float* o4, int co, int col) {
    float t[4] = {o4[0], o4[1], o4[2], o4[3]};
    if (a.next_alpha) {
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = snake_f(t[e], a.next_alpha[co + e]);
    }
    uint32_t h0, l0 = 0, h1, l1 = 0;
    if (NP == 2) {
      split2_f16_pk(t[0], t[1], h0, l0, a.ovf);
      split2_f16_pk(t[2], t[3], h1, l1, a.ovf);
    } else {
      h0 = cvt_pk_bf16_f32(t[0], t[1]);
      h1 = cvt_pk_bf16_f32(t[2], t[3]);
    }
    char* dst = reinterpret_cast<char*>(a.outp) +
                ((((int64_t)b * (a.w.cout >> 4) + (co >> 4)) * NP) * opld + col) * 32 + (co & 15) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    if (NP == 2) *reinterpret_cast<uint2*>(dst + (int64_t)opld * 32) = make_uint2(l0, l1);
  };
  if constexpr (PH) {
    // tile i = phase + i: the lane's MT values of (channel, input column q) are MT consecutive output columns
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = q0 + qw + j * 32 + li;
      const bool live = q < ncols;
      const int col0 = (live ? q : 0) * a.out_stride + phase;
      const int row0 = 4 * lk;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb + row0 + 8 * r4);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sg + row0 + 8 * r4);
        const int co = co0 + row0 + 8 * r4;
        float o[MT][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = r4 * 4 + e;
          const int coe = min(co + e, co_last);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const float rv = has_res ? rb[(int64_t)coe * old_ + col0 + i] : 0.f;
            o[i][e] = finish(acc[i][j][r], NP == 2 ? acx[i][j][r] : 0.f, b4[e], g4[e], rv);
          }
          if (a.out && live && co + e <= co_last) {
            float* dst = ob + (int64_t)(co + e) * old_ + col0;
            if constexpr (MT == 4) *reinterpret_cast<f32x4*>(dst) = (f32x4){o[0][e], o[1][e], o[2][e], o[3][e]};
            else if constexpr (MT == 2) *reinterpret_cast<float2*>(dst) = make_float2(o[0][e], o[1][e]);
            else {
#pragma unroll
              for (int i = 0; i < MT; ++i) dst[i] = o[i][e];
            }
          }
        }
        if (a.outp && live && co + 3 <= co_last) {
#pragma unroll
          for (int i = 0; i < MT; ++i) put_planes(o[i], co, col0 + i);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int q = q0 + qw + j * 32 + li;
        const bool live = q < ncols;
        const int col = (live ? q : 0) * a.out_stride + phase;
        const int row0 = i * 32 + 4 * lk;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = min(co0 + row0 + 8 * (r >> 2) + (r & 3), co_last);
          rv[r] = has_res ? rb[(int64_t)co * old_ + col] : 0.f;
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float o4[4];
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb + row0 + 8 * r4);
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(sg + row0 + 8 * r4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = r4 * 4 + e;
            const float v = finish(acc[i][j][r], NP == 2 ? acx[i][j][r] : 0.f, b4[e], g4[e], rv[r]);
            const int co = co0 + row0 + 8 * r4 + e;
            if (a.out && live && co <= co_last) ob[(int64_t)co * old_ + col] = v;
            o4[e] = v;
          }
          if (a.outp) {
            const int co = co0 + row0 + 8 * r4;
            if (live && co + 3 <= co_last) put_planes(o4, co, col);
          }
        }
      }
  }
}

// "Background" occupancy of the decode-side conv kernels (fmi_dac_set_background): a floor under the dynamic LDS a launch
// asks for.  Above 80 KiB only ONE work-group of these kernels fits a CU (4 waves x <= 256 registers = half of the
// register file, the other half and >= 75 KiB of LDS stay free), so that the work-groups of ANOTHER queue -- the Dual-AR
// frame loop's 8-wave GEMVs at 72-80 registers -- can be co-resident instead of waiting for a conv work-group to retire.
static thread_local int t_conv_lds_floor = 0;
void set_conv_lds_floor(int bytes) { t_conv_lds_floor = bytes; }

// transposed conv, phase-tiled (conv_mfma_bf16_kernel<..., PH = true>): MT phases x 32 channels per work-group
template <int MT, int G, int NP>
static int launch_conv_bf16_ph(const ConvArgs& a, int ncols, int tap_off0, int span, hipStream_t s) {
  const ConvW& w = a.w;
  constexpr int TT = 128, CO_T = MT * 32;
  const int wx = (TT - 1) * a.x_stride + span;
  size_t smem = (size_t)(G * NP * wx + w.taps * G * NP * CO_T) * 32;
  FMI_REQUIRE(smem <= 160 * 1024, "conv(bf16, phases): LDS tile of %zu bytes exceeds 160 KiB", smem);
  if (smem < (size_t)t_conv_lds_floor) smem = (size_t)t_conv_lds_floor;
  FMI_REQUIRE(w.phases % MT == 0 && w.cout_pad % 32 == 0 && (w.cin_pad16 >> 4) % G == 0, "conv(bf16, phases): bad tiling");
  dim3 grid(cdiv(ncols, TT), w.cout_pad / 32, a.B * (w.phases / MT)), block(256);
  if (smem > 64 * 1024)
    FMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_mfma_bf16_kernel<MT, 1, G, NP, 0, 0, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((conv_mfma_bf16_kernel<MT, 1, G, NP, 0, 0, true>), grid, block, smem, s, a, ncols, wx, tap_off0);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

template <int MT, int NT, int G, int NP, int TAPS, int TC>
static int launch_conv_bf16_tt(const ConvArgs& a, int ncols, int tap_off0, int span, hipStream_t s) {
  const ConvW& w = a.w;
  constexpr int TT = 4 * NT * 32, CO_T = MT * 32;
  const int wx = (TT - 1) * a.x_stride + span;
  const int wt = TC > 0 ? TC : w.taps;
  size_t smem = (size_t)(G * NP * wx + wt * G * NP * CO_T) * 32;
  FMI_REQUIRE(smem <= 160 * 1024, "conv(bf16): LDS tile of %zu bytes exceeds 160 KiB", smem);
  if (smem < (size_t)t_conv_lds_floor) smem = (size_t)t_conv_lds_floor;
  FMI_REQUIRE(w.cout_pad % CO_T == 0 && (w.cin_pad16 >> 4) % G == 0, "conv(bf16): tile does not divide the packed weight");
  dim3 grid(cdiv(ncols, TT), w.cout_pad / CO_T, a.B * w.phases), block(256);
  if (smem > 64 * 1024)
    FMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_mfma_bf16_kernel<MT, NT, G, NP, TAPS, TC>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((conv_mfma_bf16_kernel<MT, NT, G, NP, TAPS, TC>), grid, block, smem, s, a, ncols, wx, tap_off0);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

template <int MT, int NT, int G, int NP>
static int launch_conv_bf16_t(const ConvArgs& a, int ncols, int tap_off0, int span, hipStream_t s) {
  static const int tc_env = []() { const char* e = getenv("FMI_CONV_TC"); return e ? atoi(e) : -1; }();
  if (G == 1 && a.w.taps == 7) {
    const int tc = tc_env >= 0 ? tc_env : (NP >= 2 ? 4 : 0);   // one plane: the whole k = 7 tile is small enough
    if (tc == 3) return launch_conv_bf16_tt<MT, NT, G, NP, 7, 3>(a, ncols, tap_off0, span, s);
    if (tc > 0) return launch_conv_bf16_tt<MT, NT, G, NP, 7, 4>(a, ncols, tap_off0, span, s);
    return launch_conv_bf16_tt<MT, NT, G, NP, 7, 0>(a, ncols, tap_off0, span, s);
  }
  return launch_conv_bf16_tt<MT, NT, G, NP, 0, 0>(a, ncols, tap_off0, span, s);
}

template <int NP>
static int launch_conv_bf16(const ConvArgs& a, int ncols, int tap_off0, int span, hipStream_t s) {
  const ConvW& w = a.w;
  const int ct = w.cout_pad / 32;
  static const int env_mt = []() { const char* e = getenv("FMI_CONV_MT"); return e ? atoi(e) : 0; }();
  int MT = 1;
  for (int m : {3, 4, 2})   // 96-row tiles first: measured 95.6 vs 99.1 ms per batch-8 decode against 128-row-first
    if (ct % m == 0) { MT = m; break; }
  // the weight tile is NP planes of taps x 16 channels: 128 rows x 7 taps x 3 planes is 84 KiB -- one work-group per
  // CU; 64 rows keep two resident, which hides the staging phase of one behind the matrix phase of the other
  const size_t wbytes = (size_t)(w.taps == 7 && NP >= 2 ? 4 : w.taps) * NP * MT * 32 * 32;   // resident part (TC = 4)
  if (MT == 4 && wbytes > 50 * 1024) MT = 2;
  if (env_mt >= 1 && env_mt <= 4 && ct % env_mt == 0) MT = env_mt;
  const bool k1 = (w.taps == 1 && a.x_stride == 1 && (w.cin_pad16 >> 4) % 2 == 0);
  // k = 1 layers stage G x 16 channels per step.  A small grid (the codec transformer and the chunks of a streaming
  // decode: one or two column tiles per utterance) is a serial chain of steps per work-group with most CUs idle: it
  // takes 32-row tiles (4x the work-groups) and 64-channel steps.  Neither changes the accumulation order of an
  // output element, so results do not depend on the choice (incremental decodes stay bit-identical to offline).
  static const int env_nt = []() { const char* e = getenv("FMI_CONV_NT"); return e ? atoi(e) : 0; }();
  static const int env_g = []() { const char* e = getenv("FMI_CONV_K1G"); return e ? atoi(e) : 0; }();
  static const int env_small = []() { const char* e = getenv("FMI_CONV_K1SMALL"); return e ? atoi(e) : 1; }();
  int kg = 2;
  bool small = false;
  if (k1) {
    const int cgs = w.cin_pad16 >> 4;
    const int64_t col_tiles = (int64_t)cdiv(ncols, 128) * a.B;
    small = env_small && col_tiles * (ct / MT) < 256 && env_mt == 0;
    if (small)
      for (int m : {4, 3, 2, 1})
        if (ct % m == 0 && (col_tiles * (ct / m) >= 256 || m == 1)) { MT = m; break; }
    static const int env_gs = []() { const char* e = getenv("FMI_CONV_K1G_SMALL"); return e ? atoi(e) : 4; }();
    // few column tiles (the codec transformer, a streaming chunk): every work-group walks the whole reduction as a chain
    // of load -> split -> barrier -> MFMA steps with little else to overlap it; 64-channel steps halve the chain
    static const int env_few = []() { const char* e = getenv("FMI_CONV_K1_FEWCOLS"); return e ? atoi(e) : 16; }();
    const int want = env_g ? env_g : (small ? env_gs : (col_tiles <= env_few ? 4 : 2));
    const int nt = small || MT == 4 ? 1 : MT == 3 ? ((env_nt == 1 || (NP == 2 && env_nt != 2)) ? 1 : 2) : MT == 2 ? 2 : 4;
    for (int g : {8, 4, 2})   // both tiles of a step within 128 KiB of LDS
      if (g <= want && cgs % g == 0 && (size_t)g * NP * (4 * nt * 32 + MT * 32) * 32 <= 128 * 1024) { kg = g; break; }
  }
  // transposed convs (two taps per phase): one 16-channel group per step is a chain of copy -> barrier -> 18 MFMAs ->
  // barrier steps, C_in / 16 of them; 2 or 4 groups per step shorten the chain (bit-identical: group-major products)
  static const int env_upg = []() { const char* e = getenv("FMI_CONV_UPG"); return e ? atoi(e) : 4; }();
  // phase-tiled transposed convs (FMI_CONV_PH=0: one phase per work-group, A/B): 4 (stride 4, 8) or 2 phases per
  // work-group; the output-column vector store needs lout rows 16-byte aligned per 4 columns
  static const int env_ph = []() { const char* e = getenv("FMI_CONV_PH"); return e ? atoi(e) : 1; }();
  if (env_ph && w.phases >= 2 && w.taps != 7 && a.out_stride == w.phases && w.cout_pad % 32 == 0 &&
      (a.out_ld ? a.out_ld : a.lout) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
    const int cgs = w.cin_pad16 >> 4;
    const int g = (env_ph >= 2 || cgs % 2) ? 1 : 2;
    if (w.phases % 4 == 0) {
      if (g == 2) return launch_conv_bf16_ph<4, 2, NP>(a, ncols, tap_off0, span, s);
      return launch_conv_bf16_ph<4, 1, NP>(a, ncols, tap_off0, span, s);
    }
    if (w.phases % 2 == 0) {
      if (g == 2) return launch_conv_bf16_ph<2, 2, NP>(a, ncols, tap_off0, span, s);
      return launch_conv_bf16_ph<2, 1, NP>(a, ncols, tap_off0, span, s);
    }
  }
  int ug = 1;
  if (!k1 && w.taps > 1 && w.taps != 7 && NP == 2 && MT == 3) {
    const int cgs = w.cin_pad16 >> 4;
    const int wxu = 127 * a.x_stride + span;
    for (int g : {4, 2})
      if (g <= env_upg && cgs % g == 0 && (size_t)g * NP * (wxu + w.taps * MT * 32) * 32 <= 80 * 1024) { ug = g; break; }
  }
  // k = 7 convs on a SMALL grid (the first chunks of a streaming decode: a few hundred columns per utterance): every
  // work-group walks C_in / 16 copy -> barrier -> products -> barrier steps with nobody else on its CU to hide them
  // (block 0 of the decoder: 48 steps, ~130 us whatever the width).  Two channel groups per step, the whole k = 7 tile
  // resident (109 KiB of LDS: one work-group per CU, which a small grid has anyway), halve the chain; group-major
  // products, so the bits do not change.  FMI_CONV_K7_SMALL = work-group count below which this form is used (0: never).
  static const int env_k7 = []() { const char* e = getenv("FMI_CONV_K7_SMALL"); return e ? atoi(e) : 300; }();
  if (!k1 && w.taps == 7 && NP == 2 && MT == 3 && a.x_stride == 1 && env_k7 > 0) {
    const int cgs = w.cin_pad16 >> 4;
    const int64_t wgs = (int64_t)cdiv(ncols, 128) * (w.cout_pad / 96) * a.B * w.phases;
    const int wxu = 127 * a.x_stride + span;
    if (wgs < env_k7 && cgs % 2 == 0 && (size_t)2 * NP * (wxu + w.taps * MT * 32) * 32 <= 150 * 1024) ug = 2;
  }
#define FMI_CONVB(MT_, NT_)                                                              \
  do {                                                                                   \
    if (!k1 && ug == 4 && MT_ == 3 && NT_ == 1) return launch_conv_bf16_t<3, 1, 4, NP>(a, ncols, tap_off0, span, s); \
    if (!k1 && ug == 2 && MT_ == 3 && NT_ == 1) return launch_conv_bf16_t<3, 1, 2, NP>(a, ncols, tap_off0, span, s); \
    if (!k1) return launch_conv_bf16_t<MT_, NT_, 1, NP>(a, ncols, tap_off0, span, s);    \
    if (kg == 8) return launch_conv_bf16_t<MT_, NT_, 8, NP>(a, ncols, tap_off0, span, s); \
    if (kg == 4) return launch_conv_bf16_t<MT_, NT_, 4, NP>(a, ncols, tap_off0, span, s); \
    return launch_conv_bf16_t<MT_, NT_, 2, NP>(a, ncols, tap_off0, span, s);             \
  } while (0)
  if (small && MT == 2) FMI_CONVB(2, 1);
  if (small && MT == 1) FMI_CONVB(1, 1);
  if (MT == 4) FMI_CONVB(4, 1);
  // two accumulator sets (fp16 split): 96 x 256 tiles need 192 accumulator registers and spill; 96 x 128 fit
  // (74.5 -> 67.8 ms per batch-8 decode)
  if (MT == 3 && (env_nt == 1 || (NP == 2 && env_nt != 2))) FMI_CONVB(3, 1);
  if (MT == 3) FMI_CONVB(3, 2);
  if (MT == 2) FMI_CONVB(2, 2);
  FMI_CONVB(1, 4);
#undef FMI_CONVB
}

// =====================================================================================
// k = 1 layer on ready operand planes, every wave on its own (the codec transformer's linears)
// =====================================================================================
// The conv kernel above stages the input tile of EVERY 32..128-row work-group through LDS (load, fp16 split, barrier):
// for a linear over a few hundred columns that staging is the whole cost (a 1024 x 1024 x 256 GEMM took 55 us, a
// 1024 x 3072 one 150 us).  Here the producer (column norm, SiLU-mul, attention) has already written the activations as
// fp16 hi/lo operand planes [B][cin/16][2][L][16] -- the layout of ConvArgs::xp -- and the weights are operand planes
// too (ConvW::wb), so a wave fetches both MFMA operands straight from L2 with one 16-byte load per lane and plane, no
// LDS, no barrier; four independent waves per work-group, each a (32 MT rows x 32 columns) tile over the whole
// reduction.  Same products in the same order per output element as conv_mfma_bf16_kernel<.., NP = 2>.
template <int MT>
__global__ __launch_bounds__(256) void linear_planes_kernel(ConvW w, const bf16_t* __restrict__ xp, float* __restrict__ out,
                                                           const float* __restrict__ res, const float* __restrict__ gamma,
                                                           int act, int L) {
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rt = blockIdx.y * 4 + wave;
  const int co0 = rt * (32 * MT);
  if (co0 >= w.cout_pad) return;
  const int q = blockIdx.x * 32 + li, b = blockIdx.z;
  const bool live = q < L;
  const int cgs = w.cin_pad16 >> 4;
  // A: slot 1 / 2 of group g, row co: 16 halfs at ((g * 3 + slot) * cout_pad + co) * 16, halves swapped on rows with bit 3
  const char* wa = reinterpret_cast<const char*>(w.wb) + ((int64_t)(cgs > 0 ? 1 : 0) * w.cout_pad + co0 + li) * 32 +
                   (((lk ^ (li >> 3)) & 1) << 4);
  const int64_t wa_group = (int64_t)3 * w.cout_pad * 32, wa_plane = (int64_t)w.cout_pad * 32;
  // B: plane pl of group g, column q: 16 halfs at (((b * cgs + g) * 2 + pl) * L + q) * 16
  const char* xb = reinterpret_cast<const char*>(xp) + (((int64_t)b * cgs * 2) * L + (live ? q : 0)) * 32 + lk * 16;
  const int64_t xb_plane = (int64_t)L * 32, xb_group = 2 * xb_plane;

  f32x16 acc[MT], acx[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = acx[i][r] = 0.f;

  // a ring of R channel groups in registers: the operands of group g + R are requested right after group g's products
  // are issued, so R - 1 groups of matrix work (and their load latency) overlap every L2 round trip
  constexpr int R = MT == 1 ? 8 : 4;
  f16x8 a0[R][MT], a1[R][MT], b0[R], b1[R];
  auto fetch = [&](int g, int r) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      a0[r][i] = *reinterpret_cast<const f16x8*>(wa + g * wa_group + i * 1024);
      a1[r][i] = *reinterpret_cast<const f16x8*>(wa + g * wa_group + wa_plane + i * 1024);
    }
    b0[r] = *reinterpret_cast<const f16x8*>(xb + g * xb_group);
    b1[r] = *reinterpret_cast<const f16x8*>(xb + g * xb_group + xb_plane);
  };
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < cgs) fetch(r, r);
  for (int g0 = 0; g0 < cgs; g0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int g = g0 + r;
      if (g < cgs) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[r][i], b1[r], acx[i], 0, 0, 0);
          acx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[r][i], b0[r], acx[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[r][i], b0[r], acc[i], 0, 0, 0);
        }
        if (g + R < cgs) fetch(g + R, r);
      }
    }
  }

  // epilogue as conv_mfma_bf16_kernel: out = res + gamma * act(acc + bias)
  if (!live) return;
  const int co_last = w.cout - 1;
  float* ob = out + (int64_t)b * w.cout * L;
  const float* rb = res ? res + (int64_t)b * w.cout * L : nullptr;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row0 = co0 + i * 32 + 4 * lk;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = row0 + 8 * (r >> 2) + (r & 3);
      if (co > co_last) continue;
      float v = (acc[i][r] + acx[i][r] * (1.0f / F16_LO_SCALE)) + (w.bias ? w.bias[co] : 0.f);
      if (act == ACT_GELU) v = gelu_f(v);
      v = v * (gamma ? gamma[co] : 1.f) + (rb ? rb[(int64_t)co * L + q] : 0.f);
      ob[(int64_t)co * L + q] = v;
    }
  }
}

int launch_linear_planes(const ConvW& w, const bf16_t* xp, float* out, const float* res, const float* gamma, int act,
                         int B, int L, hipStream_t s) {
  FMI_REQUIRE(w.wb && w.taps == 1 && w.phases == 1 && w.cin % 16 == 0 && w.cin_pad16 == w.cin && w.cout_pad % 32 == 0,
              "linear_planes: layer is not a plain k = 1 layer with operand planes");
  const int ct = w.cout_pad / 32;
  // 64-row wave tiles only when that still leaves every SIMD several waves
  static const int env_mt = []() { const char* e = getenv("FMI_LINP_MT"); return e ? atoi(e) : 0; }();
  const bool mt2 = env_mt ? env_mt == 2 : (ct % 2 == 0 && (int64_t)cdiv(L, 32) * (ct / 2) * B >= 4096);
  if (mt2 && ct % 2 == 0) {
    dim3 grid(cdiv(L, 32), cdiv(ct / 2, 4), B);
    hipLaunchKernelGGL(linear_planes_kernel<2>, grid, dim3(256), 0, s, w, xp, out, res, gamma, act, L);
  } else {
    dim3 grid(cdiv(L, 32), cdiv(ct, 4), B);
    hipLaunchKernelGGL(linear_planes_kernel<1>, grid, dim3(256), 0, s, w, xp, out, res, gamma, act, L);
  }
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// one value as its two fp16 operand-plane halves
__device__ inline void split1_f16(float x, uint16_t& h, uint16_t& l, int* __restrict__ ovf) {
  uint32_t hh, ll;
  split2_f16_pk(x, 0.f, hh, ll, ovf);
  h = (uint16_t)hh;
  l = (uint16_t)ll;
}

template <int MT, int NT, int G>
static int launch_conv_t(const ConvArgs& a, int ncols, int tap_off0, int span, hipStream_t s) {
  const ConvW& w = a.w;
  constexpr int TT = 4 * NT * 32, CO_T = MT * 32;
  const int wx = (TT - 1) * a.x_stride + span;
  const size_t smem = (size_t)(G * wx * 8 + w.taps * G * CO_T * 8) * sizeof(float);
  FMI_REQUIRE(smem <= 160 * 1024, "conv: LDS tile of %zu bytes exceeds 160 KiB", smem);
  FMI_REQUIRE(w.cout_pad % CO_T == 0 && (w.cin_pad >> 3) % G == 0, "conv: tile does not divide the packed weight");
  dim3 grid(cdiv(ncols, TT), w.cout_pad / CO_T, a.B * w.phases), block(256);
  if (smem > 64 * 1024)
    FMI_CHECK_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<MT, NT, G>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, G>), grid, block, smem, s, a, ncols, wx, tap_off0);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int launch_conv(const ConvArgs& a0, hipStream_t s) {
  ConvArgs a = a0;
  if (!a.ovf) a.ovf = f16_overflow_target();
  const ConvW& w = a.w;
  FMI_REQUIRE(w.w && w.cin_pad % 8 == 0 && w.cout_pad % 32 == 0, "conv: weights not packed");
  const int ncols = (a.out_stride == 1) ? a.lout : a.lout / a.out_stride;
  const int tap_off0 = (a.tap_step < 0) ? -(w.taps - 1) * a.tap_step : 0;
  const int span = (w.taps - 1) * (a.tap_step < 0 ? -a.tap_step : a.tap_step) + 1;
  FMI_REQUIRE(a.planes > 0 || (!a.x_ld && !a.x_left && !a.out_ld && !a.outp_ld), "conv: streaming strides need the bf16-plane kernel");
  if (a.planes > 0) {
    FMI_REQUIRE(w.wb && w.cin_pad16 % 16 == 0, "conv: this layer has no bf16 planes (planes=%d requested)", a.planes);
    if (a.planes == 1) return launch_conv_bf16<1>(a, ncols, tap_off0, span, s);
    return launch_conv_bf16<2>(a, ncols, tap_off0, span, s);
  }
  const int ct = w.cout_pad / 32;
  // tile height: the largest of 4/3/2/1 (x32 rows) that divides the channel tiles, so that no
  // work-group multiplies padding (C = 192 -> 2 x 96, not 128 + 64) and every DMA piece is inside the weight
  int MT = 1;
  for (int m : {4, 3, 2})
    if (ct % m == 0) { MT = m; break; }
  // k = 1 layers hold almost no math per 8 channels: stage 32 at a time
  const bool k1 = (w.taps == 1 && a.x_stride == 1 && w.cin_pad % 32 == 0);
#define FMI_CONV(MT_, NT_)                                                            \
  return k1 ? launch_conv_t<MT_, NT_, 4>(a, ncols, tap_off0, span, s)                 \
            : launch_conv_t<MT_, NT_, 1>(a, ncols, tap_off0, span, s)
  if (MT == 4) FMI_CONV(4, 1);
  if (MT == 3) FMI_CONV(3, 2);
  if (MT == 2) FMI_CONV(2, 2);
  FMI_CONV(1, 4);
#undef FMI_CONV
}

// =====================================================================================
// column-wise norms ([B][C][L]: statistics over C for each (b, t))
// =====================================================================================

// block (32 columns, NG channel groups): thread (column, group) walks the channels group, group + NG, ...; the group
// partials meet in LDS and are summed in group order.  32 groups (1024 threads): a streaming chunk of 32 frames is one
// work-group per utterance, so the length of the per-thread channel walk IS the kernel's duration (8 groups: 40-48 us
// for a megabyte of data, 17 launches per decode).
constexpr int NORM_NG = 32;

__global__ __launch_bounds__(32 * NORM_NG) void rmsnorm_cols_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    float eps, float* __restrict__ out, int C, int L) {
  __shared__ float part[NORM_NG][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + tx, b = blockIdx.y;
  const float* xb = x + (int64_t)b * C * L;
  float ss = 0.f;
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG) {
      const float v = xb[(int64_t)c * L + t];
      ss += v * v;
    }
  }
  part[ty][tx] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_NG; ++i) tot += part[i][tx];
  const float rstd = rsqrtf(tot / (float)C + eps);
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG) out[(int64_t)b * C * L + (int64_t)c * L + t] = xb[(int64_t)c * L + t] * rstd * w[c];
  }
}

int launch_rmsnorm_cols(const float* x, const float* w, float eps, float* out, int B, int C, int L, hipStream_t s) {
  hipLaunchKernelGGL(rmsnorm_cols_kernel, dim3(cdiv(L, 32), B), dim3(32 * NORM_NG), 0, s, x, w, eps, out, C, L);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// the same statistics; the normalised tensor leaves as fp16 hi/lo operand planes [B][C/16][2][L][16] (C % 16 == 0):
// a thread writes four channels (8 bytes) of a column per plane
__global__ __launch_bounds__(32 * NORM_NG) void rmsnorm_cols_planes_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                           float eps, bf16_t* __restrict__ outp, int C, int L, int* __restrict__ ovf) {
  __shared__ float part[NORM_NG][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + tx, b = blockIdx.y;
  const float* xb = x + (int64_t)b * C * L;
  float ss = 0.f;
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG) {
      const float v = xb[(int64_t)c * L + t];
      ss += v * v;
    }
  }
  part[ty][tx] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_NG; ++i) tot += part[i][tx];
  const float rstd = rsqrtf(tot / (float)C + eps);
  if (t >= L) return;
  char* ob = reinterpret_cast<char*>(outp) + ((int64_t)b * (C >> 4) * 2) * L * 32;
  for (int c4 = ty; c4 < (C >> 2); c4 += NORM_NG) {
    const int c = c4 * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = xb[(int64_t)(c + e) * L + t] * rstd * w[c + e];
    uint32_t h0, l0, h1, l1;
    split2_f16_pk(v[0], v[1], h0, l0, ovf);
    split2_f16_pk(v[2], v[3], h1, l1, ovf);
    char* dst = ob + (((int64_t)(c >> 4) * 2) * L + t) * 32 + (c & 15) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + (int64_t)L * 32) = make_uint2(l0, l1);
  }
}

int launch_rmsnorm_cols_planes(const float* x, const float* w, float eps, bf16_t* outp, int B, int C, int L, hipStream_t s) {
  FMI_REQUIRE(C % 16 == 0, "rmsnorm planes: C %% 16");
  hipLaunchKernelGGL(rmsnorm_cols_planes_kernel, dim3(cdiv(L, 32), B), dim3(32 * NORM_NG), 0, s, x, w, eps, outp, C, L, f16_overflow_target());
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ __launch_bounds__(32 * NORM_NG) void layernorm_cols_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, float eps,
                                                                      float* __restrict__ out, int C, int L) {
  __shared__ float part[NORM_NG][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + tx, b = blockIdx.y;
  const float* xb = x + (int64_t)b * C * L;
  float sm = 0.f;
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG) sm += xb[(int64_t)c * L + t];
  }
  part[ty][tx] = sm;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_NG; ++i) tot += part[i][tx];
  const float mean = tot / (float)C;
  __syncthreads();
  float sv = 0.f;
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG) {
      const float d = xb[(int64_t)c * L + t] - mean;
      sv += d * d;
    }
  }
  part[ty][tx] = sv;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_NG; ++i) tot += part[i][tx];
  const float rstd = rsqrtf(tot / (float)C + eps);
  if (t < L) {
#pragma unroll 4
    for (int c = ty; c < C; c += NORM_NG)
      out[(int64_t)b * C * L + (int64_t)c * L + t] = (xb[(int64_t)c * L + t] - mean) * rstd * w[c] + bias[c];
  }
}

int launch_layernorm_cols(const float* x, const float* w, const float* b, float eps, float* out, int B, int C, int L,
                          hipStream_t s) {
  hipLaunchKernelGGL(layernorm_cols_kernel, dim3(cdiv(L, 32), B), dim3(32 * NORM_NG), 0, s, x, w, b, eps, out, C, L);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// depthwise causal conv k=7 (ConvNeXt dwconv, rvq.py:154-160)
__global__ void dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                               float* __restrict__ out, int C, int L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* xr = x + ((int64_t)b * C + c) * L;
  float acc = bias[c];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int col = t - 6 + k;
    if (col >= 0) acc += w[c * 7 + k] * xr[col];
  }
  out[((int64_t)b * C + c) * L + t] = acc;
}

int launch_dwconv7(const float* x, const float* w, const float* b, float* out, int B, int C, int L, hipStream_t s) {
  hipLaunchKernelGGL(dwconv7_kernel, dim3(cdiv(L, 256), C, B), dim3(256), 0, s, x, w, b, out, C, L);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void silu_mul_kernel(const float* __restrict__ ab, float* __restrict__ out, int F, int L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float g = ab[((int64_t)b * 2 * F + f) * L + t];
  const float u = ab[((int64_t)b * 2 * F + F + f) * L + t];
  out[((int64_t)b * F + f) * L + t] = (g / (1.0f + expf(-g))) * u;
}

// SiLU(gate) * up as operand planes [B][F/16][2][L][16]: thread = (column, channel quad)
__global__ __launch_bounds__(256) void silu_mul_planes_kernel(const float* __restrict__ ab, bf16_t* __restrict__ outp, int F, int L, int* __restrict__ ovf) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63), f = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4, b = blockIdx.z;
  if (t >= L || f >= F) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float g = ab[((int64_t)b * 2 * F + f + e) * L + t];
    const float u = ab[((int64_t)b * 2 * F + F + f + e) * L + t];
    v[e] = (g / (1.0f + expf(-g))) * u;
  }
  uint32_t h0, l0, h1, l1;
  split2_f16_pk(v[0], v[1], h0, l0, ovf);
  split2_f16_pk(v[2], v[3], h1, l1, ovf);
  char* dst = reinterpret_cast<char*>(outp) + ((((int64_t)b * (F >> 4) + (f >> 4)) * 2) * L + t) * 32 + (f & 15) * 2;
  *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(dst + (int64_t)L * 32) = make_uint2(l0, l1);
}

int launch_silu_mul_planes(const float* ab, bf16_t* outp, int B, int F, int L, hipStream_t s) {
  FMI_REQUIRE(F % 16 == 0, "silu_mul planes: F %% 16");
  hipLaunchKernelGGL(silu_mul_planes_kernel, dim3(cdiv(L, 64), F / 16, B), dim3(256), 0, s, ab, outp, F, L, f16_overflow_target());
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int launch_silu_mul(const float* ab, float* out, int B, int F, int L, hipStream_t s) {
  hipLaunchKernelGGL(silu_mul_kernel, dim3(cdiv(L, 256), F, B), dim3(256), 0, s, ab, out, F, L);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// RoPE on the q and k blocks of qkv [B][3C][L]; pair (2p, 2p+1) inside each head, bf16 table
// (modded_dac.py:442-473)
__global__ void rope_cols_kernel(float* __restrict__ qkv, const bf16_t* __restrict__ table, int C, int L, int hd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int pr = blockIdx.y;  // pair index over q and k: [0, C) (C/2 pairs each)
  const int b = blockIdx.z;
  if (t >= L) return;
  const int which = pr / (C / 2), p = pr % (C / 2);
  const int row = which * C + 2 * p;
  const int pin = p % (hd / 2);
  float* r0 = qkv + ((int64_t)b * 3 * C + row) * L + t;
  float* r1 = r0 + L;
  const float c = bf2f(table[((int64_t)t * (hd / 2) + pin) * 2]), sn = bf2f(table[((int64_t)t * (hd / 2) + pin) * 2 + 1]);
  const float x0 = *r0, x1 = *r1;
  *r0 = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, sn));
  *r1 = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(x0, sn));
}

int launch_rope_cols(float* qkv, const bf16_t* table, int B, int C, int L, int hd, hipStream_t s) {
  hipLaunchKernelGGL(rope_cols_kernel, dim3(cdiv(L, 256), C, B), dim3(256), 0, s, qkv, table, C, L, hd);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// causal window-limited attention (modded_dac.py:380-398): query t sees keys max(0,t-w+1)..t.
// one wave per (b, head, t); lanes run over keys (coalesced along time), head_dim <= 64.
// qkv rows have stride ld (>= L: a persistent buffer of an incremental decode); only queries q_lo..L-1 are computed and
// written to the compact out [B][C][L - q_lo].
__global__ __launch_bounds__(256) void window_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C,
                                                          int L, int hd, int window, int ld, int q_lo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = q_lo + blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* qb = qkv + ((int64_t)b * 3 * C + h * hd) * ld;
  const float* kb = qb + (int64_t)C * ld;
  const float* vb = kb + (int64_t)C * ld;
  const int lo = max(0, t - window + 1);
  const float scale = 1.0f / sqrtf((float)hd);
  float q[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) q[d] = d < hd ? qb[(int64_t)d * ld + t] : 0.f;
  // pass 1: scores, running max
  float mx = -INFINITY;
  for (int j0 = lo; j0 <= t; j0 += 64) {
    const int j = j0 + lane;
    if (j <= t) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d)
        if (d < hd) sc += q[d] * kb[(int64_t)d * ld + j];
      mx = fmaxf(mx, sc * scale);
    }
  }
  mx = wave_max(mx);
  // pass 2: probabilities and the weighted sum of values
  float o[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) o[d] = 0.f;
  float den = 0.f;
  for (int j0 = lo; j0 <= t; j0 += 64) {
    const int j = j0 + lane;
    if (j <= t) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d)
        if (d < hd) sc += q[d] * kb[(int64_t)d * ld + j];
      const float p = expf(sc * scale - mx);
      den += p;
#pragma unroll
      for (int d = 0; d < 64; ++d)
        if (d < hd) o[d] += p * vb[(int64_t)d * ld + j];
    }
  }
  den = wave_sum(den);
  const int lo_ = L - q_lo;
  float* ob = out + ((int64_t)b * C + h * hd) * lo_ + (t - q_lo);
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    if (d < hd) {
      const float v = wave_sum(o[d]);
      if (lane == 0) ob[(int64_t)d * lo_] = v / den;
    }
  }
}

// The same attention with the K / V window of QT consecutive queries staged once in LDS (head_dim 64, window <= 128:
// the quantizer's pre/post transformers).  Work-group = (16 queries, head, batch item), a wave takes 4 of the queries:
//   scores   lane <-> key (two keys per lane), q broadcast from LDS, K columns read conflict-free (odd row stride);
//   softmax  as in window_attn_kernel (same lane <-> key assignment, same reduction trees: identical p and den);
//   output   lane <-> head dimension: o[d] = sum_j p[j] * V[d][j] in key order -- no cross-lane reduction at all
//            (the old kernel paid 64 wave-wide reductions per query: 0.91 ms per layer at 8 x 215 frames).
// A query's result depends only on its own position, so it is invariant under batch, total length and tiling.
template <int QT>
__global__ __launch_bounds__(256) void window_attn_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              int C, int L, int window, int ld, int q_lo,
                                                              bf16_t* __restrict__ outp, int* __restrict__ ovf) {
  constexpr int HD = 64;
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  const int KT = QT + window - 1, RS = KT | 1;
  float* Ks = wsm;
  float* Vs = Ks + HD * RS;
  float* Qs = Vs + HD * RS;          // [QT][HD]
  float* Ps = Qs + QT * HD;          // [4][128]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int t0 = q_lo + blockIdx.x * QT, h = blockIdx.y, b = blockIdx.z;
  const float* qb = qkv + ((int64_t)b * 3 * C + h * HD) * ld;
  const float* kb = qb + (int64_t)C * ld;
  const float* vb = kb + (int64_t)C * ld;
  const int k0 = max(0, t0 - window + 1);
  const int nk = min(L, t0 + QT) - k0;
  for (int d = wave; d < HD; d += 4)
    for (int c = lane; c < nk; c += 64) {
      Ks[d * RS + c] = kb[(int64_t)d * ld + k0 + c];
      Vs[d * RS + c] = vb[(int64_t)d * ld + k0 + c];
    }
  for (int i = tid; i < QT * HD; i += 256) {
    const int d = i / QT, q = i % QT;
    Qs[q * HD + d] = t0 + q < L ? qb[(int64_t)d * ld + t0 + q] : 0.f;
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)HD);
  float* pw = Ps + wave * 128;
  for (int qi = wave; qi < QT; qi += 4) {
    const int t = t0 + qi;
    if (t >= L) break;
    const int lo = max(0, t - window + 1);
    const float* qrow = Qs + qi * HD;
    float sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lo + u * 64 + lane;
      float a = 0.f;
      if (j <= t) {
        const float* kc = Ks + (j - k0);
#pragma unroll 8
        for (int d = 0; d < HD; ++d) a += qrow[d] * kc[d * RS];
      }
      sc[u] = j <= t ? a * scale : -INFINITY;
    }
    const float mx = wave_max(fmaxf(sc[0], sc[1]));
    const float p0 = sc[0] > -INFINITY ? expf(sc[0] - mx) : 0.f, p1 = sc[1] > -INFINITY ? expf(sc[1] - mx) : 0.f;
    const float den = wave_sum(p0 + p1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();               // the previous query's reads of pw are done
    pw[lane] = p0;
    pw[64 + lane] = p1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int n = t - lo + 1;
    const float* vr = Vs + lane * RS + (lo - k0);  // lane = head dimension
    float o = 0.f;
    for (int jj = 0; jj < n; ++jj) o += pw[jj] * vr[jj];
    if (outp) {   // fp16 hi/lo operand planes [B][C/16][2][L - q_lo][16] for linear_planes_kernel
      const int c = h * HD + lane, n = L - q_lo;
      uint16_t hh, ll;
      split1_f16(o / den, hh, ll, ovf);
      char* dst = reinterpret_cast<char*>(outp) + ((((int64_t)b * (C >> 4) + (c >> 4)) * 2) * n + (t - q_lo)) * 32 + (c & 15) * 2;
      *reinterpret_cast<uint16_t*>(dst) = hh;
      *reinterpret_cast<uint16_t*>(dst + (int64_t)n * 32) = ll;
    } else {
      out[((int64_t)b * C + h * HD + lane) * (L - q_lo) + (t - q_lo)] = o / den;
    }
  }
}

int launch_window_attn(const float* qkv, float* out, int B, int C, int L, int hd, int window, hipStream_t s, int ld,
                       int q_lo, bf16_t* outp) {
  FMI_REQUIRE(hd <= 64 && C % hd == 0, "window_attn: head_dim %d unsupported", hd);
  if (ld <= 0) ld = L;
  FMI_REQUIRE(ld >= L && q_lo >= 0 && q_lo < L, "window_attn: bad stride / query range");
  const int nq = L - q_lo;
  constexpr int QT = 16;
  const size_t smem = (size_t)(2 * 64 * ((QT + window - 1) | 1) + QT * 64 + 4 * 128) * sizeof(float);
  static const bool lds_off = []() { const char* e = getenv("FMI_WATTN_OLD"); return e && atoi(e) != 0; }();
  if (hd == 64 && window <= 128 && smem <= 80 * 1024 && !lds_off) {
    if (smem > 64 * 1024)
      FMI_CHECK_HIP(hipFuncSetAttribute((const void*)window_attn_lds_kernel<QT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)smem));
    hipLaunchKernelGGL((window_attn_lds_kernel<QT>), dim3(cdiv(nq, QT), C / hd, B), dim3(256), smem, s, qkv, out, C, L,
                       window, ld, q_lo, outp, f16_overflow_target());
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  FMI_REQUIRE(!outp, "window_attn: operand-plane output needs head_dim 64 and window <= 128");
  hipLaunchKernelGGL(window_attn_kernel, dim3(cdiv(nq, 4), C / hd, B), dim3(256), 0, s, qkv, out, C, L, hd, window, ld, q_lo);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// quantizer: decode tables, lookup, encode step
// =====================================================================================

__global__ void clamp_indices_kernel(int64_t* idx, int n1, int T, int sem_size, int cb_size, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int book = (int)((i / T) % n1);
  const int64_t hi = (book == 0 ? sem_size : cb_size) - 1;
  if (idx[i] > hi) idx[i] = hi;  // upper clamp only, in place (rvq.py:354-359)
}

int launch_clamp_indices(int64_t* idx, int B, int n1, int T, int sem_size, int cb_size, hipStream_t s) {
  const int64_t total = (int64_t)B * n1 * T;
  hipLaunchKernelGGL(clamp_indices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, idx, n1, T, sem_size,
                     cb_size, total);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// table[code][c] = bias[c] + sum_d w[c][d] * codebook[code][d]   (out_proj of each code, from_codes)
__global__ void build_lut_kernel(const float* __restrict__ codebook, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ table, int n, int d, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, code = blockIdx.y;
  if (c >= C) return;
  float acc = 0.f;
  for (int j = 0; j < d; ++j) acc += w[c * d + j] * codebook[code * d + j];
  table[(int64_t)code * C + c] = acc + bias[c];
}

int launch_build_lut(const float* codebook, const float* w, const float* bias, float* table, int n, int d, int C,
                     hipStream_t s) {
  hipLaunchKernelGGL(build_lut_kernel, dim3(cdiv(C, 256), n), dim3(256), 0, s, codebook, w, bias, table, n, d, C);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// z[b][c][t] = table_sem[idx0][c] + (((0 + r_0) + r_1) + ...)   (ResidualVectorQuantize.from_codes order)
__global__ void lut_decode_kernel(const int64_t* __restrict__ idx, const float* __restrict__ tables,
                                  const int* __restrict__ rows_off, int n_books, float* __restrict__ out, int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const int64_t* ib = idx + (int64_t)b * (n_books + 1) * T;
  float r = 0.f;
  for (int i = 0; i < n_books; ++i) r += tables[((int64_t)rows_off[i + 1] + ib[(int64_t)(i + 1) * T + t]) * C + c];
  const float sem = tables[((int64_t)rows_off[0] + ib[t]) * C + c];
  out[((int64_t)b * C + c) * T + t] = sem + r;
}

int launch_lut_decode(const int64_t* idx, const float* tables, const int* rows_off, int n_books, int sem_size,
                      int cb_size, float* out, int B, int C, int T, hipStream_t s) {
  (void)sem_size;
  (void)cb_size;
  hipLaunchKernelGGL(lut_decode_kernel, dim3(cdiv(T, 64), C, B), dim3(64), 0, s, idx, tables, rows_off, n_books, out,
                     C, T);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// One VectorQuantize.forward at inference (dac.nn.quantize): in_proj, L2-normalised nearest code
// (argmax of -dist, first index on ties), straight-through sum, out_proj, residual update.
// one work-group per (b, t).
__global__ __launch_bounds__(256) void vq_step_kernel(VqArgs a) {
  __shared__ float s_part[256];
  __shared__ float s_ze[16], s_en[16], s_st[16];
  __shared__ float s_best[256];
  __shared__ int s_besti[256];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, b = blockIdx.y;
  float* res = a.residual + (int64_t)b * a.C * a.T + t;
  const int d = a.d;
  // z_e[j] = in_b[j] + sum_c in_w[j][c] * residual[c]
  for (int j = 0; j < d; ++j) {
    float p = 0.f;
    for (int c = tid; c < a.C; c += 256) p += a.in_w[(int64_t)j * a.C + c] * res[(int64_t)c * a.T];
    s_part[tid] = p;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) s_part[tid] += s_part[tid + o];
      __syncthreads();
    }
    if (tid == 0) s_ze[j] = s_part[0] + a.in_b[j];
    __syncthreads();
  }
  if (tid == 0) {  // F.normalize: x / max(||x||, 1e-12)
    float n2 = 0.f;
    for (int j = 0; j < d; ++j) n2 += s_ze[j] * s_ze[j];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    for (int j = 0; j < d; ++j) s_en[j] = s_ze[j] * inv;
  }
  __syncthreads();
  float e2 = 0.f;
  for (int j = 0; j < d; ++j) e2 += s_en[j] * s_en[j];
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int code = tid; code < a.n; code += 256) {
    const float* cb = a.codebook + (int64_t)code * d;
    float n2 = 0.f;
    for (int j = 0; j < d; ++j) n2 += cb[j] * cb[j];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    float dot = 0.f, c2 = 0.f;
    for (int j = 0; j < d; ++j) {
      const float cn = cb[j] * inv;
      dot += s_en[j] * cn;
      c2 += cn * cn;
    }
    const float negdist = -((e2 - 2.0f * dot) + c2);
    if (negdist > best) {  // strictly greater: the lowest index wins ties within a thread
      best = negdist;
      besti = code;
    }
  }
  s_best[tid] = best;
  s_besti[tid] = besti;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const float ob = s_best[tid + o];
      const int oi = s_besti[tid + o];
      if (ob > s_best[tid] || (ob == s_best[tid] && oi < s_besti[tid])) {
        s_best[tid] = ob;
        s_besti[tid] = oi;
      }
    }
    __syncthreads();
  }
  int code = s_besti[0];
  if (code < 0 || code >= a.n) code = 0;  // NaN input: no candidate compared greater; never index out of bounds
  if (tid == 0) {
    a.codes[((int64_t)b * a.books + a.book) * a.T + t] = code;
    for (int j = 0; j < d; ++j) {
      const float zq = a.codebook[(int64_t)code * d + j];
      s_st[j] = s_ze[j] + (zq - s_ze[j]);  // straight-through estimator kept at inference
    }
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += 256) {
    float v = 0.f;
    for (int j = 0; j < d; ++j) v += a.out_w[(int64_t)c * d + j] * s_st[j];
    res[(int64_t)c * a.T] -= (v + a.out_b[c]);
  }
}

int launch_vq_step(const VqArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.d <= 16, "vq: codebook_dim %d > 16", a.d);
  hipLaunchKernelGGL(vq_step_kernel, dim3(a.T, a.B), dim3(256), 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// first / last convolution of the codec (1 input resp. 1 output channel: not GEMM shaped)
// =====================================================================================

// encoder.block.0: causal conv k=7, 1 -> C
__global__ void first_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, int C, int L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* xr = x + (int64_t)b * L;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int col = t - 6 + k;
    if (col >= 0) acc += w[c * 7 + k] * xr[col];
  }
  out[((int64_t)b * C + c) * L + t] = acc + bias[c];
}

int launch_first_conv(const float* x, const float* w, const float* bias, float* out, int B, int C, int L, hipStream_t s) {
  hipLaunchKernelGGL(first_conv_kernel, dim3(cdiv(L, 256), C, B), dim3(256), 0, s, x, w, bias, out, C, L);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// decoder tail: Snake -> causal conv k=7, C -> 1 -> tanh (modded_dac.py:792-796).  The tile of
// Snake(x) is staged once in LDS (each element is used by 7 outputs).
__global__ __launch_bounds__(256) void final_conv_tanh_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int C, int L, int col0) {
  // columns [col0, L) are produced, packed to rows of L - col0 samples (col0 > 0: the incremental decode drops
  // the left-context samples; the sum order of a column does not depend on col0, so results are bit-identical)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TT = 256, CC = 32;  // 32 channels x (256 + 6) columns per stage
  float* Xs = smem;                 // [CC][TT + 6]
  float* Wl = smem + CC * (TT + 6); // [CC][7]
  const int tid = threadIdx.x;
  const int t0 = col0 + blockIdx.x * TT, b = blockIdx.y;
  const float* xb = x + (int64_t)b * C * L;
  float acc = 0.f;
  for (int c0 = 0; c0 < C; c0 += CC) {
    for (int e = tid; e < CC * (TT + 6); e += 256) {
      const int r = e / (TT + 6), cidx = e - r * (TT + 6);
      const int c = c0 + r, col = t0 - 6 + cidx;
      float v = 0.f;
      if (c < C && col >= 0 && col < L) v = snake_f(xb[(int64_t)c * L + col], alpha[c]);
      Xs[e] = v;
    }
    for (int e = tid; e < CC * 7; e += 256) Wl[e] = (c0 + e / 7 < C) ? w[(c0 + e / 7) * 7 + e % 7] : 0.f;
    __syncthreads();
    for (int r = 0; r < CC; ++r)
#pragma unroll
      for (int k = 0; k < 7; ++k) acc += Wl[r * 7 + k] * Xs[r * (TT + 6) + tid + k];
    __syncthreads();
  }
  const int t = t0 + tid;
  if (t < L) out[(int64_t)b * (L - col0) + (t - col0)] = tanhf(acc + bias[0]);
}

int launch_final_conv_tanh(const float* x, const float* alpha, const float* w, const float* bias, float* out, int B,
                           int C, int L, int col0, hipStream_t s) {
  const size_t smem = (size_t)(32 * (256 + 6) + 32 * 7) * sizeof(float);
  hipLaunchKernelGGL(final_conv_tanh_kernel, dim3(cdiv(L - col0, 256), B), dim3(256), smem, s, x, alpha, w, bias, out,
                     C, L, col0);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// one thread per (row, halo column, 4-byte word of the element)
__global__ void halo_swap_kernel(uint32_t* __restrict__ buf, const uint32_t* __restrict__ old_state, uint32_t* __restrict__ new_state,
                                 int64_t total, int h, int n, int ew) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int wd = (int)(idx % ew);
  const int j = (int)((idx / ew) % h);
  const int64_t row = idx / ((int64_t)ew * h);
  const int64_t ld = (int64_t)(h + n) * ew;
  uint32_t* r = buf + row * ld;
  const uint32_t* o = old_state + row * (int64_t)h * ew;
  // S = [old (h) | new (n)]; the next state is S[n .. n + h): taken from old where it still reaches into the halo
  const uint32_t keep = (n + j < h) ? o[(int64_t)(n + j) * ew + wd] : r[(int64_t)(n + j) * ew + wd];
  const uint32_t front = o[(int64_t)j * ew + wd];
  new_state[(row * h + j) * ew + wd] = keep;
  r[(int64_t)j * ew + wd] = front;    // (position j < h: read above only as S[n + j] with n + j >= h, i.e. never: no hazard)
}

int launch_halo_swap(void* buf, const void* old_state, void* new_state, int64_t rows, int h, int n, int elem_bytes,
                     hipStream_t s) {
  FMI_REQUIRE(h >= 1 && n >= 1 && elem_bytes % 4 == 0 && old_state != new_state, "halo_swap: bad arguments");
  const int ew = elem_bytes / 4;
  const int64_t total = rows * h * ew;
  hipLaunchKernelGGL(halo_swap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (uint32_t*)buf,
                     (const uint32_t*)old_state, (uint32_t*)new_state, total, h, n, ew);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int read_clear_f16_overflow(int* dev_word, int* flag, hipStream_t s) {
  int v = 0;
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  FMI_CHECK_HIP(hipMemcpy(&v, dev_word, sizeof(int), hipMemcpyDeviceToHost));
  if (v) {   // cleared ON the handle's stream and waited for: h->stream is non-blocking, a null-stream memset would
             // not be ordered before the next launch on it and could wipe (or lose) a freshly raised flag
    FMI_CHECK_HIP(hipMemsetAsync(dev_word, 0, sizeof(int), s));
    FMI_CHECK_HIP(hipStreamSynchronize(s));
  }
  *flag = v;
  return FMI_OK;
}

}  // namespace fmi
