// dualar_gemm.hip -- prefill GEMM of the Dual-AR path (llama.py:895,946,979-987 over whole prompts); split out of
// dualar_kernels.hip so that the translation units compile in parallel.
#include "dualar_kernels.h"
#include "dualar_dev.h"

namespace fmi {
// =====================================================================================
// tiled linear (any M; prefill): 128x128 block, 4 waves (2x2), each 64x64 = 4x4 MFMA tiles.
// Operands go straight from global/L2 into fragments (weights are already fragment-ordered).
// =====================================================================================

// bf16 output of a linear, times the per-row scale of a weight-only-int8 checkpoint (packed row order) if present
__device__ inline float lin_out(float acc, const bf16_t* scale, int packed_row) {
  float o = rbf(acc);
  if (scale) o = rbf(o * bf2f(scale[packed_row]));
  return o;
}

template <int EPI>
__global__ __launch_bounds__(256) void linear_tiled_kernel(LinearArgs a) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wn = wave & 1, wm = wave >> 1;
  const int KT = a.K >> 5;
  const int n_tile0 = blockIdx.x * 8 + wn * 4;  // 16-row weight tiles
  const int m0 = blockIdx.y * 128 + wm * 64;
  const int NT = a.N >> 4;
  const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(a.wp);
  const int mi = lane & 15, g = lane >> 4;

  const bf16_t* xrow[4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    int m = m0 + tm * 16 + mi;
    if (m >= a.M) m = a.M - 1;
    xrow[tm] = a.x + (int64_t)m * a.ldx;
  }
  int ntile[4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) ntile[tn] = min(n_tile0 + tn, NT - 1);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < KT; ++kt) {
    uint4 wv[4], xv[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) wv[tn] = wp[((int64_t)ntile[tn] * KT + kt) * 64 + lane];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) xv[tm] = *reinterpret_cast<const uint4*>(xrow[tm] + packed_k0(kt, g, KT));
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv[tn]),
                                                              *reinterpret_cast<bf16x8*>(&xv[tm]), acc[tn][tm], 0, 0, 0);
  }

  // lane holds D[n = tile*16 + g*4 + j][m = tile*16 + mi]
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + tm * 16 + mi;
    if (m >= a.M) continue;
    if (EPI == EPI_SILU) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int nt_gate = n_tile0 + tp * 2;
        if (nt_gate >= NT) continue;
        const int n = (nt_gate >> 1) * 16 + g * 4;
        bf16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gate = rbf(silu_f(lin_out(acc[tp * 2][tm][j], a.scale, nt_gate * 16 + g * 4 + j)));
          float up = lin_out(acc[tp * 2 + 1][tm][j], a.scale, (nt_gate + 1) * 16 + g * 4 + j);
          o[j] = f2bf(gate * up);
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        if (n_tile0 + tn >= NT) continue;
        const int n = (n_tile0 + tn) * 16 + g * 4;
        bf16_t o[4];
        if (EPI == EPI_RESIDUAL) {
          uint2 rv = *reinterpret_cast<const uint2*>(a.res + (int64_t)m * a.ldr + n);
          const bf16_t* re = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(bf2f(re[j]) + lin_out(acc[tn][tm][j], a.scale, n + j));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(lin_out(acc[tn][tm][j], a.scale, n + j));
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    }
  }
}

// LDS-staged variant of the prefill GEMM (default).  The direct variant above feeds every MFMA from L2
// (8 wave-loads per 16 MFMAs per wave: 17 % of the bf16 peak at M = 1600); here a 128 x 128 output tile shares its
// operands through LDS: per k-step (2 k-tiles = 64 k) 16 KiB of weights arrive by linear LDS-DMA (the packed
// layout already is fragment order) and 16 KiB of activations by per-lane DMA (lane (row, kg) fetches the 16
// bytes the B operand lane needs, packed_k0 map), double-buffered, one barrier per k-step; the operands are read
// back with ds_read_b128 at lane*16 (conflict-free).  LDS reads are inline asm so that the compiler does not drain
// the in-flight DMA of the next step before every read (cf. tools/gemv_lds_probe.hip).  Same MFMA order per
// output element as the direct variant: identical results.
__device__ inline u32x4 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

constexpr char FMI_GEMM_DEFAULT = 'a';   // 'a' = by shape: 'x' (128 x 256 tile) or 'w' (128 x 128), see launch_linear_tiled;   // prefill GEMM variant when FMI_GEMM is unset: wave-specialised (8 x 200 rows: 30.3 -> 25.4 ms, 8 x 2048: 225.9 -> 215.1 ms on MI355X; bit-identical)

template <int EPI>
__global__ __launch_bounds__(256) void linear_tiled_lds_kernel(LinearArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2 stages][A 16 pieces | B 16 pieces] x 1 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int KT = a.K >> 5, KS = (KT + 1) >> 1;  // k-steps of two k-tiles (the last may hold one)
  const int NT = a.N >> 4;
  const int n_blk0 = blockIdx.x * 8, m_blk0 = blockIdx.y * 128;
  const int mi = lane & 15, g = lane >> 4;
  const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(a.wp);

  auto stage = [&](int ks, int buf) {
    char* base = smem + buf * 32768;
    for (int p = wave; p < 32; p += 4) {       // pieces 0-15: weights, 16-31: activations; piece = tile*2 + kk
      const int kk = p & 1, j = 2 * ks + kk;
      if (j >= KT) continue;                   // unpaired last k-tile: second half of the step is empty
      if (p < 16) {
        const int nt = min(n_blk0 + (p >> 1), NT - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + ((int64_t)nt * KT + j) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
      } else {
        const int mt = (p - 16) >> 1;
        const int m = min(m_blk0 + mt * 16 + mi, a.M - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.x + (int64_t)m * a.ldx + packed_k0(j, g, KT)),
                                         (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 16u;

  stage(0, 0);
  for (int ks = 0; ks < KS; ++ks) {
    __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): this wave's pieces of step ks have landed
    __syncthreads();                           // ... everyone's have, and everyone finished reading buffer (ks+1)&1
    if (ks + 1 < KS) stage(ks + 1, (ks + 1) & 1);
    const unsigned b0 = lds0 + (unsigned)((ks & 1) * 32768);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (2 * ks + kk >= KT) break;
      u32x4 wv[4], xv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wv[t] = lds_read_b128(b0 + (unsigned)(((wn * 4 + t) * 2 + kk) * 1024));
        xv[t] = lds_read_b128(b0 + (unsigned)((16 + (wm * 4 + t) * 2 + kk) * 1024));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv[tn]),
                                                                *reinterpret_cast<bf16x8*>(&xv[tm]), acc[tn][tm], 0, 0, 0);
    }
  }

  // epilogue identical to the direct variant: lane holds D[n = tile*16 + g*4 + j][m = tile*16 + mi]
  const int n_tile0 = n_blk0 + wn * 4, m0 = m_blk0 + wm * 64;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + tm * 16 + mi;
    if (m >= a.M) continue;
    if (EPI == EPI_SILU) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int nt_gate = n_tile0 + tp * 2;
        if (nt_gate >= NT) continue;
        const int n = (nt_gate >> 1) * 16 + g * 4;
        bf16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gate = rbf(silu_f(lin_out(acc[tp * 2][tm][j], a.scale, nt_gate * 16 + g * 4 + j)));
          float up = lin_out(acc[tp * 2 + 1][tm][j], a.scale, (nt_gate + 1) * 16 + g * 4 + j);
          o[j] = f2bf(gate * up);
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        if (n_tile0 + tn >= NT) continue;
        const int n = (n_tile0 + tn) * 16 + g * 4;
        bf16_t o[4];
        if (EPI == EPI_RESIDUAL) {
          uint2 rv = *reinterpret_cast<const uint2*>(a.res + (int64_t)m * a.ldr + n);
          const bf16_t* re = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(bf2f(re[j]) + lin_out(acc[tn][tm][j], a.scale, n + j));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(lin_out(acc[tn][tm][j], a.scale, n + j));
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    }
  }
}

// Wave-specialised variant of the LDS-staged prefill GEMM (round 3).  Same tile, same LDS images, same MFMA order per
// output element (identical results) -- but the work-group has EIGHT waves: waves 0-3 only read operands from LDS and
// issue MFMAs, waves 4-7 only issue the LDS-DMA of the next k-step.  Why: a 128 x 128 x 64 step moves 32 KiB through
// the CU's texture-addresser path (64 B/clk: 512 cycles) for 128 MFMAs (16 cycles each on 4 SIMDs: 512 cycles) --
// the tile sits exactly on the ridge, so the two must OVERLAP to get anywhere, and in the 4-wave kernel the wave that
// issues a DMA piece (60-185 cycles each, MI355X_MICROARCH.md) is the wave whose MFMAs then starve.  The cyclic
// wave -> SIMD placement puts one compute and one loader wave of a work-group on every SIMD.  Operand reads of the
// second k-tile are issued before the first k-tile's MFMAs.
// WNT = 16-row weight tiles per compute wave: 4 (128 x 128 output tile, shipped) or 2 (128 rows x 64 columns: twice
// the work-groups for the GEMMs whose 128-wide tiling leaves the chip half empty at M = 1600 -- wo / w2 260
// work-groups, wqkv 624; 24 KiB per stage, three work-groups per CU; measured slower, see launch_linear_tiled).
// CW = compute waves (CW/2 along N x 2 along M, each 16*WNT columns x 64 rows), NS = LDS stages.  CW = 8, NS = 3 is the
// 128-row x 256-column tile: 48 KiB per k-step for twice the products of the 128 x 128 tile's 32 KiB -- the operand
// path of a CU delivers ~20-23 B/clk whatever the L2 hit rate and whether the bytes go by LDS-DMA or through
// registers (tools/gemm_bench.hip ablations, profiles/r03_gemm_ablation.txt), so bytes per product are what counts.
// One 12-wave work-group per CU (two compute waves + one loader wave per SIMD); the loaders run two k-steps ahead.
template <int EPI, int WNT, int CW = 4, int NS = 2>
__global__ __launch_bounds__((CW + 4) * 64, CW == 8 ? 3 : WNT == 4 ? 4 : 6) void linear_tiled_ws_kernel(LinearArgs a) {
  constexpr int WN = CW / 2;
  constexpr int NTW = WN * WNT, AP = NTW * 2, NP = AP + 16, STAGE = NP * 1024;   // n-tiles, weight pieces, pieces, bytes
  constexpr int PW = NP / 4;                                                     // pieces per loader wave and k-step
  static_assert(NP % 4 == 0 && (NS == 2 || NS == 3), "linear_tiled_ws_kernel: bad configuration");
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [NS stages][A AP pieces | B 16 pieces] x 1 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= CW;
  const int wave = loader ? wave8 - CW : wave8;
  const int wn = wave % WN, wm = wave / WN;
  const int KT = a.K >> 5, KS = (KT + 1) >> 1;  // k-steps of two k-tiles (the last may hold one)
  const int NT = a.N >> 4;
  const int n_blk0 = blockIdx.x * NTW, m_blk0 = blockIdx.y * 128;
  const int mi = lane & 15, g = lane >> 4;
  const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(a.wp);
  if (loader) {
    // One loader wave owns pieces p = wave + 4 i (i < PW) of every stage: k-tile kk = wave & 1 of the step, weight
    // tiles 2 i + (wave >> 1) for i < AP / 4, then activation row tiles.  All per-lane source addresses are formed
    // ONCE; a k-step adds a constant (two packed weight tiles = 2 KiB; 64 activation columns = 128 B, packed_k0 is
    // linear in the step for paired k-tiles).  The loop this replaced recomputed tile / row / packed_k0 / min() per
    // piece behind a non-unrolled branchy loop: ~30 instructions per 1 KiB piece, and the loaders -- not the L2, not
    // the LDS-DMA path -- bounded the kernel (profiles/r03_gemm_ablation.txt: 1 KiB per ~300 cycles per loader wave
    // whatever the hit rate or the number of pieces in flight).
    const int kk = wave & 1, half = wave >> 1;
    const char* src[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      if (i < AP / 4) {
        const int nt = min(n_blk0 + 2 * i + half, NT - 1);
        src[i] = reinterpret_cast<const char*>(wp + ((int64_t)nt * KT + kk) * 64 + lane);
      } else {
        const int m = min(m_blk0 + (2 * (i - AP / 4) + half) * 16 + mi, a.M - 1);
        src[i] = reinterpret_cast<const char*>(a.x + (int64_t)m * a.ldx + (g >> 1) * 32 + (((g & 1) << 1) + kk) * 8);
      }
    }
    auto stage = [&](int ks, int buf) {
      char* base = smem + buf * STAGE + wave * 1024;
#pragma unroll
      for (int i = 0; i < PW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)ks * (i < AP / 4 ? 2048 : 128)),
                                         (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
    };
    for (int i = 0; i < NS - 1 && i < KS; ++i) stage(i, i);
    int nb = NS - 1;                           // buffer of the next step to issue
    for (int ks = 0; ks < KS; ++ks) {
      // this wave's pieces of step ks have landed (with three stages the next step's PW pieces may stay in flight)
      if (NS == 3 && ks + 1 < KS) {
        if constexpr (PW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        static_assert(NS == 2 || PW == 12 || PW == 8, "vmcnt literal");
      } else {
        __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0)
      }
      __syncthreads();                         // ... everyone's have; the compute waves are done with the buffer of step ks-1
      if (ks + NS - 1 < KS) stage(ks + NS - 1, nb);
      nb = nb + 1 == NS ? 0 : nb + 1;
    }
    return;
  }

  f32x4 acc[WNT][4];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 16u;

  int cb = 0;
  for (int ks = 0; ks < KS; ++ks) {            // K / 32 is even here (launch_linear_tiled): every step holds two k-tiles
    __syncthreads();
    const unsigned b0 = lds0 + (unsigned)(cb * STAGE);
    cb = cb + 1 == NS ? 0 : cb + 1;
    u32x4 wv0[WNT], xv[4], wv1[WNT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < WNT) wv0[t] = lds_read_b128(b0 + (unsigned)(((wn * WNT + t) * 2) * 1024));
      xv[t] = lds_read_b128(b0 + (unsigned)((AP + (wm * 4 + t) * 2) * 1024));
    }
#pragma unroll
    for (int t = 0; t < WNT; ++t) wv1[t] = lds_read_b128(b0 + (unsigned)(((wn * WNT + t) * 2 + 1) * 1024));
    // (the waits name the registers they make valid: MFMA builtins are not memory operations, so nothing else keeps
    // the compiler from scheduling a product above the wait for its operand)
    if constexpr (WNT == 4)
      asm volatile("s_waitcnt lgkmcnt(4)"   // the first k-tile's reads are back
                   : "+v"(wv0[0]), "+v"(wv0[1]), "+v"(wv0[2]), "+v"(wv0[3]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3])
                   :: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(2)"
                   : "+v"(wv0[0]), "+v"(wv0[1]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3])
                   :: "memory");
    // activation-tile-major order: once the products of activation tile tm are issued its register is free
    // for the SECOND k-tile's fragment, which then arrives under the remaining products (128-register budget:
    // two work-groups = four waves per SIMD)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
      for (int tn = 0; tn < WNT; ++tn)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv0[tn]),
                                                              *reinterpret_cast<bf16x8*>(&xv[tm]), acc[tn][tm], 0, 0, 0);
      xv[tm] = lds_read_b128(b0 + (unsigned)((AP + (wm * 4 + tm) * 2 + 1) * 1024));
      __builtin_amdgcn_sched_barrier(0);   // keep each reload right behind the products that freed its register
    }
    if constexpr (WNT == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(wv1[0]), "+v"(wv1[1]), "+v"(wv1[2]), "+v"(wv1[3]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3])
                   :: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(wv1[0]), "+v"(wv1[1]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3])
                   :: "memory");
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < WNT; ++tn)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv1[tn]),
                                                              *reinterpret_cast<bf16x8*>(&xv[tm]), acc[tn][tm], 0, 0, 0);
  }

  // epilogue identical to the other variants: lane holds D[n = tile*16 + g*4 + j][m = tile*16 + mi]
  const int n_tile0 = n_blk0 + wn * WNT, m0 = m_blk0 + wm * 64;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + tm * 16 + mi;
    if (m >= a.M) continue;
    if (EPI == EPI_SILU) {
#pragma unroll
      for (int tp = 0; tp < WNT / 2; ++tp) {
        const int nt_gate = n_tile0 + tp * 2;
        if (nt_gate >= NT) continue;
        const int n = (nt_gate >> 1) * 16 + g * 4;
        bf16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gate = rbf(silu_f(lin_out(acc[tp * 2][tm][j], a.scale, nt_gate * 16 + g * 4 + j)));
          float up = lin_out(acc[tp * 2 + 1][tm][j], a.scale, (nt_gate + 1) * 16 + g * 4 + j);
          o[j] = f2bf(gate * up);
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < WNT; ++tn) {
        if (n_tile0 + tn >= NT) continue;
        const int n = (n_tile0 + tn) * 16 + g * 4;
        bf16_t o[4];
        if (EPI == EPI_RESIDUAL) {
          uint2 rv = *reinterpret_cast<const uint2*>(a.res + (int64_t)m * a.ldr + n);
          const bf16_t* re = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(bf2f(re[j]) + lin_out(acc[tn][tm][j], a.scale, n + j));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(lin_out(acc[tn][tm][j], a.scale, n + j));
        }
        *reinterpret_cast<uint2*>(a.out + (int64_t)m * a.ldo + n) = *reinterpret_cast<uint2*>(o);
      }
    }
  }
}

int launch_linear_tiled(const LinearArgs& a, hipStream_t s, bool force_direct, int variant) {
  FMI_REQUIRE(a.norm_w == nullptr, "linear_tiled: fused norm not supported (use rmsnorm_rows)");
  FMI_REQUIRE(a.bias == nullptr, "linear_tiled: no bias epilogue (skinny kernel only)");
  FMI_REQUIRE(a.K % 32 == 0 && a.N % 16 == 0 && a.ldx % 8 == 0 && a.ldo % 4 == 0, "linear_tiled: bad shape");
  if (a.epi == EPI_SILU) FMI_REQUIRE(a.N % 32 == 0, "linear_tiled: SwiGLU needs N %% 32");
  dim3 grid(cdiv(a.N / 16, 8), cdiv(a.M, 128)), block(256);
  // A/B switch: FMI_GEMM = d (operands straight from L2), l (LDS-staged, 4 waves), w (LDS-staged, wave-specialised)
  static const char env_mode = []() { const char* e = getenv("FMI_GEMM"); return e ? e[0] : '\0'; }();
  char mode = force_direct ? 'd' : variant == 1 ? 'l' : variant == 2 ? 'w' : variant == 3 ? 'x' : env_mode ? env_mode : FMI_GEMM_DEFAULT;
  if (mode == 'a') {
    // 128 x 256 tiles for the long prefills only.  In isolation the wide tile wins from ~160 work-groups on
    // (profiles/r03_gemm_sweep.txt: 11 row counts x 4 shapes, e.g. 8 x 200 rows wqkv 96 vs 99 us, w1|w3 237 vs 259), but
    // inside the layer sequence of a prefill that does not carry over below ~4 k rows: 8 x 200 tokens 26.0 vs 25.5 ms,
    // 8 x 300 33.2 vs 32.0 with the rule "from 160 work-groups"; 8 x 1024 100.1 vs 101.3, 8 x 2048 204.8 vs 211.4.
    // All variants give identical bits, so the choice may depend on the row count without touching batch invariance.
    mode = (a.M >= 4096 && !(a.epi == EPI_SILU && a.M > 12288)) ? 'x' : 'w';
  }
  if ((mode == 'w' || mode == 'x') && ((a.K >> 5) & 1)) mode = 'l';   // the wave-specialised loop takes k-tiles in pairs
  constexpr int smem = 2 * 32768;
  if (mode == 'x') {   // 128 x 256 tile, 8 compute + 4 loader waves, three 48 KiB stages
    constexpr int smem_x = 3 * 48 * 1024;
    static const hipError_t x0 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_STORE, 4, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_x);
    static const hipError_t x1 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_RESIDUAL, 4, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_x);
    static const hipError_t x2 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_SILU, 4, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_x);
    FMI_CHECK_HIP(x0); FMI_CHECK_HIP(x1); FMI_CHECK_HIP(x2);
    dim3 grid_x(cdiv(a.N / 16, 16), grid.y);
    if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_STORE, 4, 8, 3>), grid_x, dim3(768), smem_x, s, a);
    else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_RESIDUAL, 4, 8, 3>), grid_x, dim3(768), smem_x, s, a);
    else hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_SILU, 4, 8, 3>), grid_x, dim3(768), smem_x, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  if (mode == 'w') {
    // 64-column tiles (FMI_GEMM_NT=4, A/B only): twice the work-groups for wo / w2 / wqkv at 8 x 200 rows, but a third
    // less reuse per staged byte on a tile that already sits on the address-path ridge -- measured 28.5 against 26.0 ms
    // for the prefill of 8 x 200 tokens (101.3 / 101.4 at 8 x 1024, 208.6 / 207.9 at 8 x 2048): not used
    static const int env_nt = []() { const char* e = getenv("FMI_GEMM_NT"); return e ? atoi(e) : 0; }();
    const bool narrow = env_nt == 4 && a.N % 64 == 0;
    if (narrow) {
      constexpr int smem4 = 2 * 24 * 1024;
      static const hipError_t b0 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_STORE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
      static const hipError_t b1 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_RESIDUAL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
      static const hipError_t b2 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_SILU, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
      FMI_CHECK_HIP(b0); FMI_CHECK_HIP(b1); FMI_CHECK_HIP(b2);
      dim3 grid4(cdiv(a.N / 16, 4), grid.y);
      if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_STORE, 2>), grid4, dim3(512), smem4, s, a);
      else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_RESIDUAL, 2>), grid4, dim3(512), smem4, s, a);
      else hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_SILU, 2>), grid4, dim3(512), smem4, s, a);
      FMI_CHECK_HIP(hipGetLastError());
      return FMI_OK;
    }
    static const hipError_t at0 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_STORE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    static const hipError_t at1 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_RESIDUAL, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    static const hipError_t at2 = hipFuncSetAttribute((const void*)linear_tiled_ws_kernel<EPI_SILU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    FMI_CHECK_HIP(at0); FMI_CHECK_HIP(at1); FMI_CHECK_HIP(at2);
    if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_STORE, 4>), grid, dim3(512), smem, s, a);
    else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_RESIDUAL, 4>), grid, dim3(512), smem, s, a);
    else hipLaunchKernelGGL((linear_tiled_ws_kernel<EPI_SILU, 4>), grid, dim3(512), smem, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  if (mode != 'd') {
    static const hipError_t at0 = hipFuncSetAttribute((const void*)linear_tiled_lds_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    static const hipError_t at1 = hipFuncSetAttribute((const void*)linear_tiled_lds_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    static const hipError_t at2 = hipFuncSetAttribute((const void*)linear_tiled_lds_kernel<EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    FMI_CHECK_HIP(at0); FMI_CHECK_HIP(at1); FMI_CHECK_HIP(at2);
    if (a.epi == EPI_STORE) hipLaunchKernelGGL(linear_tiled_lds_kernel<EPI_STORE>, grid, block, smem, s, a);
    else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL(linear_tiled_lds_kernel<EPI_RESIDUAL>, grid, block, smem, s, a);
    else hipLaunchKernelGGL(linear_tiled_lds_kernel<EPI_SILU>, grid, block, smem, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  if (a.epi == EPI_STORE) hipLaunchKernelGGL(linear_tiled_kernel<EPI_STORE>, grid, block, 0, s, a);
  else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL(linear_tiled_kernel<EPI_RESIDUAL>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(linear_tiled_kernel<EPI_SILU>, grid, block, 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

}  // namespace fmi
