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

constexpr char FMI_GEMM_DEFAULT = 'a';   // 'a' = by shape: 'p' / 's' (256 / 128 rows x 256 columns, linear_tiled_256p_kernel), see launch_linear_tiled

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

// two fp32 -> packed bf16 pair, round-to-nearest-even (one VALU instruction on gfx950), and the two halves back
__device__ inline uint32_t pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ inline float pk_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ inline float pk_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
// lin_out (above) of two neighbouring output columns, packed: bf16(acc), times the int8 row scale if present
__device__ inline uint32_t lin2(float a0, float a1, const bf16_t* scale, int packed_row) {
  uint32_t p = pk_bf16(a0, a1);
  if (scale) p = pk_bf16(pk_lo(p) * bf2f(scale[packed_row]), pk_hi(p) * bf2f(scale[packed_row + 1]));
  return p;
}
#if defined(FMI_Y_TIMING)   // tools/gemm_bench.hip: cycle stamps of work-group (0, 0): entry, first barrier passed, loop done, stores done
__device__ long long g_ytime[8];
__device__ long long g_ylog[8192][4];   // per work-group: entry stamp, end stamp, HW_ID, XCC_ID
#define FMI_YSTAMP(i)                                                                                      \
  do {                                                                                                     \
    if (threadIdx.x == 0) {                                                                                \
      const long long t_ = (long long)__builtin_amdgcn_s_memrealtime();   /* 100 MHz */                                                   \
      if (blockIdx.x == 0 && blockIdx.y == 0) g_ytime[i] = t_;                                             \
      const int id_ = blockIdx.y * gridDim.x + blockIdx.x;                                                 \
      if (id_ < 8192 && ((i) == 0 || (i) == 3)) {                                                          \
        g_ylog[id_][(i) == 0 ? 0 : 1] = t_;                                                                \
        if ((i) == 0) {                                                                                    \
          g_ylog[id_][2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                    \
          g_ylog[id_][3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));                   \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
#else
#define FMI_YSTAMP(i) do { } while (0)
#endif
#if defined(FMI_Y_ABLATE) && FMI_Y_ABLATE == 3   // ablation: no operand reads (products on whatever the registers hold)
__device__ inline u32x4 yread(unsigned addr) { return (u32x4){addr, addr, addr, addr}; }
#else
__device__ inline u32x4 yread(unsigned addr) { return lds_read_b128(addr); }
#endif
// 256 x 256 output tile, one 8-wave work-group per CU (round 4).  What bounds the 128 x 128 kernels above is the rate at
// which a CU takes operand bytes in (20-23 B/clk through the LDS-DMA path whatever the hit rate, profiles/
// r03_gemm_ablation.txt): a 128 x 128 x 64 step needs 32 KiB per 128 MFMAs = 64 B/clk at the matrix peak, the 256 x 256
// step 64 KiB per 512 MFMAs = 32 B/clk.  Waves 2 (rows) x 4 (weight columns), each 128 rows x 64 columns = 8 x 4
// accumulator tiles; every wave both computes and issues an eighth of the next step's DMA (4 weight + 4 activation
// pieces of 1 KiB) right after the barrier, so the transfer has the whole 64-MFMA compute phase to land.
// Activations arrive as FULL 128-byte lines (8 rows x 128 B per DMA instruction: half the address-path work of the
// fragment-shaped 16 rows x 64 B pieces of the kernels above).  The LDS image of a piece is row-major [8][128 B] with
// the 16-byte chunks of row mi (of 16) XOR-ed by h(mi) = bit1(mi) | bit3(mi) << 2 ON THE SOURCE SIDE (the DMA destination
// is lane-linear), which makes the fragment reads -- lane (mi, g) wants chunk 2 g + t of row mi, packed_k0 -- free of
// bank conflicts for ds_read_b128's lane groups.  Same products in the same order per output element as every other
// variant: identical bits.
template <int EPI, int MF = 8>   // MF = 16-row tiles per wave (8: eight waves of 128 rows x 64 columns, 4: sixteen of 64 x 64)
__device__ __forceinline__ void epilogue_256(const LinearArgs& a, f32x4 (&acc)[4][MF], unsigned lds0, int wave, int wn, int wm,
                                             int n_blk0, int m_blk0, int lane) {
  const int mi = lane & 15, g = lane >> 4;
  const int NT = a.N >> 4;
  // ---- epilogue through LDS (the operand stages are dead): a lane holds D[n = tile*16 + g*4 + j][m = tile*16 + mi], i.e.
  // 4 consecutive output columns of 16 different rows -- stored straight from there a wave instruction writes sixteen
  // 32-byte segments (measured: 25-50 k cycles per tile, 10-15 % of the kernel).  Each wave instead parks its bf16
  // results (the linear's own rounding, int8 row scale and SwiGLU applied) in its private 16 KiB of LDS as rows of OW
  // output columns, chunks of 16 bytes XOR-swizzled by the row so that both the 8-byte writes and the 16-byte read-back
  // are free of bank conflicts, and writes whole 128-byte (64-byte: SwiGLU) row segments; the residual is added on the
  // way out (fp32 add of two bf16 values, one rounding: llama.py:842).
  __syncthreads();                               // every wave is done with the operand stages
  constexpr int OW = EPI == EPI_SILU ? 32 : 64;  // output columns of a wave
  constexpr int CPR = OW / 8;                    // 16-byte chunks per row
  constexpr int RSH = EPI == EPI_SILU ? 2 : 1;   // rows per 256-byte bank row = 1 << RSH; swizzle = (row >> RSH) & (CPR - 1)
  const unsigned ebase = lds0 + (unsigned)(wave * (MF * 2048));
  const int n_tile0 = n_blk0 + wn * 4, m0 = m_blk0 + wm * (MF * 16);
#pragma unroll
  for (int f = 0; f < MF; ++f) {
    const int ml = f * 16 + mi;
    const int sw = (ml >> RSH) & (CPR - 1);
#pragma unroll
    for (int t = 0; t < (EPI == EPI_SILU ? 2 : 4); ++t) {
      // bf16 roundings by v_cvt_pk_bf16_f32 (round-to-nearest-even like f2bf; two values per instruction): with f2bf's
      // seven integer operations per value this loop was 20-25 k cycles per tile with nothing to overlap it
      uint32_t o2[2];
      if (EPI == EPI_SILU) {
        const int nt_gate = min(n_tile0 + t * 2, NT - 2);   // (tiles past the edge: computed on clamped rows, never stored)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int rg = nt_gate * 16 + g * 4 + 2 * h, ru = rg + 16;
          const uint32_t pg = lin2(acc[t * 2][f][2 * h], acc[t * 2][f][2 * h + 1], a.scale, rg);
          const uint32_t pu = lin2(acc[t * 2 + 1][f][2 * h], acc[t * 2 + 1][f][2 * h + 1], a.scale, ru);
          const uint32_t ps = pk_bf16(silu_f(pk_lo(pg)), silu_f(pk_hi(pg)));
          o2[h] = pk_bf16(pk_lo(ps) * pk_lo(pu), pk_hi(ps) * pk_hi(pu));
        }
      } else {
        const int n = min(n_tile0 + t, NT - 1) * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) o2[h] = lin2(acc[t][f][2 * h], acc[t][f][2 * h + 1], a.scale, n + 2 * h);
      }
      const int chunk = t * 2 + (g >> 1);
      const unsigned addr = ebase + (unsigned)(ml * (OW * 2) + ((chunk ^ sw) * 16) + (g & 1) * 8);
      const uint64_t ov = (uint64_t)o2[0] | ((uint64_t)o2[1] << 32);
      asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(ov) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (own writes only: the region is private to the wave)
  {
    constexpr int RPI = 64 / CPR;                // rows per read-back instruction
    const int rl = lane / CPR, c = lane % CPR;
    const int n_out0 = EPI == EPI_SILU ? (n_tile0 >> 1) * 16 : n_tile0 * 16;
    const int n_lim = EPI == EPI_SILU ? a.N / 2 : a.N;
    const int n = n_out0 + c * 8;
    // the residual rows of the whole sub-tile are requested before the first is used (the accumulators are dead by now):
    // loaded one per read-back iteration they were sixteen dependent round trips (32 k cycles per tile)
    constexpr int NRV = EPI == EPI_RESIDUAL ? MF * 16 / RPI : 1;
    uint4 rvs[NRV];
    if (EPI == EPI_RESIDUAL) {
#pragma unroll
      for (int it = 0; it < NRV; ++it) {
        const int m = m0 + it * RPI + rl;
        rvs[it] = make_uint4(0, 0, 0, 0);
        if (m < a.M && n < n_lim) rvs[it] = *reinterpret_cast<const uint4*>(a.res + (int64_t)m * a.ldr + n);
      }
    }
#pragma unroll
    for (int it = 0; it < MF * 16 / RPI; ++it) {
      const int ml = it * RPI + rl;
      const int m = m0 + ml;
      const int sw = (ml >> RSH) & (CPR - 1);
      u32x4 v = lds_read_b128(ebase + (unsigned)(ml * (OW * 2) + ((c ^ sw) * 16)));
      const bool ok = m < a.M && n < n_lim;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
      if (EPI == EPI_RESIDUAL) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(&rvs[EPI == EPI_RESIDUAL ? it : 0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = pk_bf16(pk_lo(rp[q]) + pk_lo(v[q]), pk_hi(rp[q]) + pk_hi(v[q]));
      }
      if (ok) *reinterpret_cast<u32x4*>(a.out + (int64_t)m * a.ldo + n) = v;
    }
  }
}

// (Measured and removed, profiles/r04_gemm_bench.txt: the DMA of step ks + 2 issued in the middle of step ks behind a second,
// raw barrier -- 3581 vs 3488 us per layer at 8 x 2048 rows --, and the pieces dealt one per four products of the first
// k-tile -- 3585 vs 3499.)
template <int EPI>
__global__ __launch_bounds__(512, 2) void linear_tiled_256_kernel(LinearArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][W 32 pieces | X 32 pieces] x 1 KiB
  FMI_YSTAMP(0);
  constexpr int STAGE = 65536, XOFF = 32768;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int KT = a.K >> 5, KS = KT >> 1;     // paired k-tiles only (launch_linear_tiled)
  const int NT = a.N >> 4;
  const int n_blk0 = blockIdx.x * 16, m_blk0 = blockIdx.y * 256;
  const int mi = lane & 15, g = lane >> 4;

  // ---- DMA sources of this wave: weight tiles 2 wave, 2 wave + 1 (two k-tiles each = 2 KiB contiguous per step) and
  // activation pieces 4 wave .. 4 wave + 3 (rows 32 wave .. 32 wave + 31 of the tile)
  const char* wsrc[2];
  const char* xsrc[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nt = min(n_blk0 + 2 * wave + i, NT - 1);
    wsrc[i] = reinterpret_cast<const char*>(a.wp) + ((int64_t)nt * KT) * 1024 + lane * 16;
  }
  {
    const int r = lane >> 3, c = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 4 * wave + i;
      const int row = min(m_blk0 + 8 * q + r, a.M - 1);
      const int h = ((r >> 1) & 1) | ((q & 1) << 2);
      xsrc[i] = reinterpret_cast<const char*>(a.x) + ((int64_t)row * a.ldx) * 2 + ((c ^ h) * 16);
    }
  }
  auto piece = [&](int ks, int buf, int i) {   // piece i of this wave's eight: 0-3 weights (tile i >> 1, k-tile i & 1), 4-7 activations
#if defined(FMI_Y_ABLATE)   // tools/gemm_bench.hip resource ablations (garbage results): 1, 2, 3 = no DMA in the steady state
    if (ks > 1) return;
#endif
    char* base = smem + buf * STAGE;
    if (i < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i >> 1] + (int64_t)ks * 2048 + (i & 1) * 1024),
                                       (__attribute__((address_space(3))) void*)(base + ((2 * wave + (i >> 1)) * 2 + (i & 1)) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i - 4] + (int64_t)ks * 128),
                                       (__attribute__((address_space(3))) void*)(base + XOFF + (4 * wave + i - 4) * 1024), 16, 0, 0);
  };
  auto stage = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(ks, buf, i);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wbase = lds0 + (unsigned)(wn * 8 * 1024 + lane * 16);
  const int hm = ((mi >> 1) & 1) | (((mi >> 3) & 1) << 2);
  unsigned xbase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    xbase[t] = lds0 + (unsigned)(XOFF + wm * 16 * 1024 + (mi >> 3) * 1024 + (mi & 7) * 128 + (((2 * g + t) ^ hm) * 16));
#if defined(FMI_Y_ABLATE) && FMI_Y_ABLATE == 2   // ablation: lane-linear (conflict-free by construction) activation reads, garbage results
  xbase[0] = xbase[1] = lds0 + (unsigned)(XOFF + wm * 16 * 1024 + lane * 16);
#endif

  stage(0, 0);
  for (int ks = 0; ks < KS; ++ks) {
    const unsigned so = (unsigned)((ks & 1) * STAGE);
    __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): this wave's pieces of step ks have landed
    __syncthreads();                           // ... everyone's have, and everyone finished reading the other buffer
    if (ks == 0) FMI_YSTAMP(1);
    u32x4 wv0[4], wv1[4], xv[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) wv0[t] = yread(wbase + so + (unsigned)((t * 2) * 1024));
#pragma unroll
    for (int f = 0; f < 8; ++f) xv[f] = yread(xbase[0] + so + (unsigned)(2 * f * 1024));
    const bool more = ks + 1 < KS;
    if (more) stage(ks + 1, (ks + 1) & 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) wv1[t] = yread(wbase + so + (unsigned)((t * 2 + 1) * 1024));
    asm volatile("s_waitcnt lgkmcnt(4)"        // the first k-tile's reads are back
                 : "+v"(wv0[0]), "+v"(wv0[1]), "+v"(wv0[2]), "+v"(wv0[3]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]),
                   "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                 :: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv0[tn]),
                                                             *reinterpret_cast<bf16x8*>(&xv[f]), acc[tn][f], 0, 0, 0);
      xv[f] = yread(xbase[1] + so + (unsigned)(2 * f * 1024));
      __builtin_amdgcn_sched_barrier(0);       // keep each reload right behind the products that freed its register
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wv1[0]), "+v"(wv1[1]), "+v"(wv1[2]), "+v"(wv1[3]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]),
                   "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7])
                 :: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv1[tn]),
                                                             *reinterpret_cast<bf16x8*>(&xv[f]), acc[tn][f], 0, 0, 0);
  }

  FMI_YSTAMP(2);
  epilogue_256<EPI>(a, acc, lds0, wave, wn, wm, n_blk0, m_blk0, lane);
#if defined(FMI_Y_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FMI_YSTAMP(3);
#endif
}

// linear_tiled_256p_kernel: products DEFERRED across the barrier, weights two steps ahead.
// In linear_tiled_256_kernel both waves of a SIMD sit behind the step's first twelve operand reads right after every barrier
// (~0.9 k of 3.4 k cycles per step; one 8-wave work-group per CU has nobody else to fill the matrix pipe).  Here the
// fragments of the two k-tiles of a step live in two register sets: after the barrier of step ks a wave requests set A
// (k-tile 0 of step ks) and, while those reads fly, multiplies set B of step ks - 1 -- with its eight DMA pieces dealt
// between the products --, then requests set B of step ks and multiplies set A: every read burst is covered by 32
// products of the same wave.  W3 (three weight stages, 160 KiB of LDS in all): the weight half of a step's operands is
// requested TWO steps ahead, the activation half one step ahead -- an LDS-DMA batch takes about a step to land
// (64 KiB through the CU's 64 B/clk address path + L2 latency), so with one step of look-ahead the step time was
// pinned to that latency.  Per accumulator the k-tiles still arrive in order: identical bits.
// MF = 16-row tiles per wave: 8 = the 256-row tile, 4 = a 128-row x 256-column tile (twice the work-groups for the GEMMs whose
// 256 x 256 tiling leaves most CUs idle: wo / w2 at 8 x 200 rows are 70 tiles).
template <int EPI, bool W3, int MF = 8>
__global__ __launch_bounds__(512, 2) void linear_tiled_256p_kernel(LinearArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // X stages 2 x 32 KiB | W stages (2 or 3) x 32 KiB
  static_assert(MF == 2 || MF == 4 || MF == 6 || MF == 8, "linear_tiled_256p_kernel: 64 / 128 / 192 / 256-row tiles");
  constexpr int BM = 32 * MF, XPW = MF / 2;       // rows of the tile; activation pieces (8 rows x 128 B) per wave and step
  constexpr int XS = BM * 128, WOFF = 2 * XS, WSN = W3 ? 3 : 2;
  FMI_YSTAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int KT = a.K >> 5;
  // gridDim.z > 1 (linear_tiled_ksplit): this work-group takes k-steps [ks0, ks0 + KS) of the KT / 2 and leaves its
  // accumulators as an fp32 partial tile; the sources below start at ks0, the loop itself is unchanged
  const int ks0 = (int)(((int64_t)(KT >> 1) * blockIdx.z) / gridDim.z);
  const int KS = (int)(((int64_t)(KT >> 1) * (blockIdx.z + 1)) / gridDim.z) - ks0;
  const int NT = a.N >> 4;
  const int n_blk0 = blockIdx.x * 16, m_blk0 = blockIdx.y * BM;
  const int mi = lane & 15, g = lane >> 4;

  // DMA sources: wave-uniform 64-bit bases + 32-bit per-lane offsets
  const char* wb0 = reinterpret_cast<const char*>(a.wp) + ((int64_t)min(n_blk0 + 2 * wave, NT - 1) * KT) * 1024 + (int64_t)ks0 * 2048;
  const char* wb1 = reinterpret_cast<const char*>(a.wp) + ((int64_t)min(n_blk0 + 2 * wave + 1, NT - 1) * KT) * 1024 + (int64_t)ks0 * 2048;
  const uint32_t woff = (uint32_t)lane * 16u;
  const char* xb = reinterpret_cast<const char*>(a.x) + (int64_t)ks0 * 128;
  uint32_t xoff[XPW];
  {
    const int r = lane >> 3, c = lane & 7;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int q = XPW * wave + i;
      const int row = min(m_blk0 + 8 * q + r, a.M - 1);
      const int h = ((r >> 1) & 1) | ((q & 1) << 2);
      xoff[i] = (uint32_t)row * (uint32_t)a.ldx * 2u + (uint32_t)((c ^ h) * 16);
    }
  }
  auto wpiece = [&](int ks, int i) {   // weight piece i (tile i >> 1, k-tile i & 1) of step ks -> weight stage ks % WSN
#if defined(FMI_Y_ABLATE)
    if (ks > 2) return;
#endif
    char* base = smem + WOFF + (ks % WSN) * 32768;
    const char* src = ((i >> 1) ? wb1 : wb0) + ((int64_t)ks * 2048 + (i & 1) * 1024);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + woff),
                                     (__attribute__((address_space(3))) void*)(base + ((2 * wave + (i >> 1)) * 2 + (i & 1)) * 1024), 16, 0, 0);
  };
  auto xpiece = [&](int ks, int i) {   // activation piece i (rows 32 wave + 8 i ..) of step ks -> activation stage ks & 1
#if defined(FMI_Y_ABLATE)
    if (ks > 2) return;
#endif
    char* base = smem + (ks & 1) * XS;
    const char* src = xb + (int64_t)ks * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + xoff[i]),
                                     (__attribute__((address_space(3))) void*)(base + (XPW * wave + i) * 1024), 16, 0, 0);
  };
  constexpr int NP = XPW + 4;                      // DMA pieces of a wave per step: activations first, then weights
  auto dpiece = [&](int ksx, int ksw, bool dmax, bool dmaw, int i) {
    if (i < XPW) { if (dmax) xpiece(ksx, i); }
    else if (dmaw) wpiece(ksw, i - XPW);
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wbase = lds0 + (unsigned)(WOFF + wn * 8 * 1024 + lane * 16);
  const int hm = ((mi >> 1) & 1) | (((mi >> 3) & 1) << 2);
  const unsigned xbase0 = lds0 + (unsigned)(wm * (2 * MF) * 1024 + (mi >> 3) * 1024 + (mi & 7) * 128 + (((2 * g) ^ hm) * 16));
  const unsigned xbase1 = xbase0 ^ 16u;          // chunk 2 g + 1: bit 0 of the chunk index is not touched by the swizzle

  u32x4 wA[4], xA[MF], wB[4], xB[MF];
  auto read_set = [&](u32x4 (&w)[4], u32x4 (&x)[MF], unsigned wso, unsigned xso, int kk) {
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = yread(wbase + wso + (unsigned)((t * 2 + kk) * 1024));
#pragma unroll
    for (int f = 0; f < MF; ++f) x[f] = yread((kk ? xbase1 : xbase0) + xso + (unsigned)(2 * f * 1024));
  };
#define FMI_WAIT_SET(w, x)                                                                                              \
  do {                                                                                                                  \
    if constexpr (MF == 8)                                                                                              \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                   : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]),    \
                     "+v"(x[MF - 4]), "+v"(x[MF - 3]), "+v"(x[MF - 2]), "+v"(x[MF - 1])                                 \
                   :: "memory");                                                                                        \
    else if constexpr (MF == 6)                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                   : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]),    \
                     "+v"(x[MF - 2]), "+v"(x[MF - 1])                                                                   \
                   :: "memory");                                                                                        \
    else if constexpr (MF == 4)                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                   : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])     \
                   :: "memory");                                                                                        \
    else                                                                                                                \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                   : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(x[0]), "+v"(x[1])                             \
                   :: "memory");                                                                                        \
  } while (0)
  // dma: this block also issues the wave's NP DMA pieces -- activations of step ksx first, then weights of step ksw --
  // dealt evenly between the groups of four products
  auto mma_set = [&](u32x4 (&w)[4], u32x4 (&x)[MF], bool dmax, int ksx, bool dmaw, int ksw) {
#pragma unroll
    for (int f = 0; f < MF; ++f) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w[tn]), *reinterpret_cast<bf16x8*>(&x[f]),
                                                             acc[tn][f], 0, 0, 0);
      if (dmax || dmaw) {
#pragma unroll
        for (int i = f * NP / MF; i < (f + 1) * NP / MF; ++i) dpiece(ksx, ksw, dmax, dmaw, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: operands of step 0 (and, W3, the weights of step 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) wpiece(0, i);
#pragma unroll
  for (int i = 0; i < XPW; ++i) xpiece(0, i);
  const bool pre1 = W3 && KS > 1;
  if (pre1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wpiece(1, i);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  FMI_YSTAMP(1);
  read_set(wA, xA, 0u, 0u, 0);
  // (no deferred products yet: the DMA of step 1 -- activations, and the weights of step 2 resp. 1 -- goes out in a row)
  if (KS > 1) {
#pragma unroll
    for (int i = 0; i < XPW; ++i) xpiece(1, i);
  }
  if (KS > (W3 ? 2 : 1)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wpiece(W3 ? 2 : 1, i);
  }
  FMI_WAIT_SET(wA, xA);
  read_set(wB, xB, 0u, 0u, 1);
  mma_set(wA, xA, false, 0, false, 0);
  FMI_WAIT_SET(wB, xB);
  for (int ks = 1; ks < KS; ++ks) {
    const unsigned xso = (unsigned)((ks & 1) * XS), wso = (unsigned)((ks % WSN) * 32768);
    // this wave's pieces of step ks have landed (W3: the four weight pieces of step ks + 1, issued last, may still fly)
    if (W3 && ks + 1 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // ... everyone's have; everyone's reads of the stages to refill are back
    read_set(wA, xA, wso, xso, 0);
    mma_set(wB, xB, ks + 1 < KS, ks + 1, ks + (W3 ? 2 : 1) < KS, ks + (W3 ? 2 : 1));   // k-tile 1 of step ks - 1 + the DMA
    FMI_WAIT_SET(wA, xA);
    read_set(wB, xB, wso, xso, 1);
    mma_set(wA, xA, false, 0, false, 0);
    FMI_WAIT_SET(wB, xB);
  }
  mma_set(wB, xB, false, 0, false, 0);
#undef FMI_WAIT_SET
  FMI_YSTAMP(2);
  if (gridDim.z > 1) {   // partial tile in fragment order: 1 KiB per store instruction (linear_tiled_reduce_kernel reads it back the same way)
    f32x4* dst = reinterpret_cast<f32x4*>(a.part) +
                 ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * (4 * MF * 64) + lane;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int f = 0; f < MF; ++f) dst[(tn * MF + f) * 64] = acc[tn][f];
    return;
  }
  epilogue_256<EPI, MF>(a, acc, lds0, wave, wn, wm, n_blk0, m_blk0, lane);
#if defined(FMI_Y_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FMI_YSTAMP(3);
#endif
}

// Second launch of a split contraction (linear_tiled_ksplit): same grid (x, y) and wave / lane roles as the
// linear_tiled_256p_kernel<EPI, false, MF> launch whose gridDim.z = S work-groups per tile left their accumulators in
// a.part; the S partial tiles are added in range order (fixed: the bits do not depend on who finished first) and the
// ordinary epilogue runs on the sums.
template <int EPI, int MF>
__global__ __launch_bounds__(512, 2) void linear_tiled_reduce_kernel(LinearArgs a, int S) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int n_blk0 = blockIdx.x * 16, m_blk0 = blockIdx.y * (32 * MF);
  const int64_t zstride = (int64_t)gridDim.y * gridDim.x * 8 * (4 * MF * 64);
  const f32x4* src = reinterpret_cast<const f32x4*>(a.part) + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * (4 * MF * 64) + lane;
  f32x4 acc[4][MF];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[tn][f] = __builtin_nontemporal_load(src + (tn * MF + f) * 64);
  for (int z = 1; z < S; ++z) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const f32x4 v = __builtin_nontemporal_load(src + z * zstride + (tn * MF + f) * 64);
        acc[tn][f] = acc[tn][f] + v;
      }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  epilogue_256<EPI, MF>(a, acc, lds0, wave, wn, wm, n_blk0, m_blk0, lane);
}

// linear_tiled_256w16_kernel: the 256 x 256 tile on SIXTEEN waves (4 x 4, each 64 rows x 64 columns, <= 128 registers:
// four waves per SIMD).  A wave issues one 16x16x32 product per ~33 cycles at best (tools/gemm_bench mfma peak: 0.48 of
// the matrix peak with one wave per SIMD, 0.96 with two), so with the two waves per SIMD of the eight-wave kernels above
// EVERY cycle a wave spends on anything else -- operand reads, DMA issue (60-185 cycles a piece), address arithmetic, the
// barrier -- is lost to the matrix pipe: their product loop with no memory operation at all reaches 0.62.  With four
// waves per SIMD a wave needs the pipe only half of the time.  Same LDS image, same DMA pieces (four per wave and
// step), same products in the same order per output element.
template <int EPI>
__global__ __launch_bounds__(1024, 4) void linear_tiled_256w16_kernel(LinearArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][W 32 pieces | X 32 pieces] x 1 KiB
  constexpr int STAGE = 65536, XOFF = 32768;
  FMI_YSTAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int KT = a.K >> 5, KS = KT >> 1;
  const int NT = a.N >> 4;
  const int n_blk0 = blockIdx.x * 16, m_blk0 = blockIdx.y * 256;
  const int mi = lane & 15, g = lane >> 4;

  // DMA of this wave: weight tile `wave` (two k-tiles = 2 KiB contiguous per step), activation rows 16 wave .. 16 wave + 15
  const char* wb = reinterpret_cast<const char*>(a.wp) + ((int64_t)min(n_blk0 + wave, NT - 1) * KT) * 1024;
  const uint32_t woff = (uint32_t)lane * 16u;
  const char* xb = reinterpret_cast<const char*>(a.x);
  uint32_t xoff[2];
  {
    const int r = lane >> 3, c = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = 2 * wave + i;
      const int row = min(m_blk0 + 8 * q + r, a.M - 1);
      const int h = ((r >> 1) & 1) | ((q & 1) << 2);
      xoff[i] = (uint32_t)row * (uint32_t)a.ldx * 2u + (uint32_t)((c ^ h) * 16);
    }
  }
  auto piece = [&](int ks, int buf, int i) {   // 0, 1: the weight tile's two k-tiles; 2, 3: the two activation pieces
    char* base = smem + buf * STAGE;
    if (i < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + ((int64_t)ks * 2048 + i * 1024) + woff),
                                       (__attribute__((address_space(3))) void*)(base + (wave * 2 + i) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + (int64_t)ks * 128 + xoff[i - 2]),
                                       (__attribute__((address_space(3))) void*)(base + XOFF + (2 * wave + i - 2) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wbase = lds0 + (unsigned)(wn * 8 * 1024 + lane * 16);
  const int hm = ((mi >> 1) & 1) | (((mi >> 3) & 1) << 2);
  const unsigned xbase0 = lds0 + (unsigned)(XOFF + wm * 8 * 1024 + (mi >> 3) * 1024 + (mi & 7) * 128 + (((2 * g) ^ hm) * 16));
  const unsigned xbase1 = xbase0 ^ 16u;

#pragma unroll
  for (int i = 0; i < 4; ++i) piece(0, 0, i);
  for (int ks = 0; ks < KS; ++ks) {
    const unsigned so = (unsigned)((ks & 1) * STAGE);
    const bool more = ks + 1 < KS;
    __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): this wave's pieces of step ks have landed
    __syncthreads();                           // ... everyone's have, and everyone finished reading the other buffer
    if (ks == 0) FMI_YSTAMP(1);
    u32x4 w0[4], w1[4], x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      w0[t] = lds_read_b128(wbase + so + (unsigned)((t * 2) * 1024));
      x[t] = lds_read_b128(xbase0 + so + (unsigned)(2 * t * 1024));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) w1[t] = lds_read_b128(wbase + so + (unsigned)((t * 2 + 1) * 1024));
    asm volatile("s_waitcnt lgkmcnt(4)"
                 : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
                 :: "memory");
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w0[tn]), *reinterpret_cast<bf16x8*>(&x[f]),
                                                             acc[tn][f], 0, 0, 0);
      x[f] = lds_read_b128(xbase1 + so + (unsigned)(2 * f * 1024));
      if (more) piece(ks + 1, (ks + 1) & 1, f);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
                 :: "memory");
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tn][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w1[tn]), *reinterpret_cast<bf16x8*>(&x[f]),
                                                             acc[tn][f], 0, 0, 0);
  }
  FMI_YSTAMP(2);
  epilogue_256<EPI, 4>(a, acc, lds0, wave, wn, wm, n_blk0, m_blk0, lane);
#if defined(FMI_Y_TIMING)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FMI_YSTAMP(3);
#endif
}

int linear_tiled_ksplit(int N, int K) {
  static const bool off = []() { const char* e = getenv("FMI_GEMM_NOSPLIT"); return e && atoi(e) != 0; }();   // (A/B runs)
  // wo / w2 of the S2-Pro shape (2560 x 4096 / 9728).  At 8 x 200 rows: 250 work-groups of 64 rows x 256 columns, each
  // bound by the ~50 GB/s a CU takes operands in at; 210 work-groups of 256 x 256 over a third of K each move 0.52 of the
  // bytes per CU.  w2: 126 -> 102 us in tools/gemm_bench, prefill 21.5 -> 19.9 ms.  wo: the micro-benchmark says no
  // (54 us either way -- its three weight copies stay in the 256 MB Infinity Cache), the STEP says yes (cold weights:
  // 74 us unsplit; prefill 19.9 -> 19.3 ms with the split, A/B on one box via FMI_GEMM_SPLIT_KMIN).
  static const int kmin = []() { const char* e = getenv("FMI_GEMM_SPLIT_KMIN"); return e ? atoi(e) : 4096; }();   // (A/B: 8192 = w2 only)
  if (off || N % 16 != 0 || N < 1024 || N > 2560 || K < kmin || ((K >> 5) & 1)) return 1;
  return 3;
}

static int ksplit_mf(int M, int N, int S) {   // the smallest tile (least operand bytes per work-group) that still fits one round
  const int ct = cdiv(N, 256);
  for (int mf = 2; mf <= 8; mf += 2)
    if (ct * cdiv(M, 32 * mf) * S <= 256) return mf;
  return 8;
}

int64_t linear_tiled_part_floats(int M, int N, int K) {
  const int S = linear_tiled_ksplit(N, K);
  if (S <= 1) return 0;
  const int mf = ksplit_mf(M, N, S);
  return (int64_t)S * cdiv(M, 32 * mf) * cdiv(N, 256) * 8 * (4 * mf * 64) * 4;
}

int launch_linear_tiled(const LinearArgs& a, hipStream_t s, bool force_direct, int variant) {
  FMI_REQUIRE(a.norm_w == nullptr, "linear_tiled: fused norm not supported (use rmsnorm_rows)");
  if (const int S = linear_tiled_ksplit(a.N, a.K); S > 1 && !force_direct && variant == 0) {
    FMI_REQUIRE(a.part != nullptr, "linear_tiled: this shape runs with a split contraction and needs LinearArgs::part");
    FMI_REQUIRE(a.bias == nullptr && a.ldx % 8 == 0 && a.ldo % 8 == 0 && (a.epi != EPI_RESIDUAL || a.ldr % 8 == 0) && a.N % 8 == 0,
                "linear_tiled: bad shape for the split contraction");
    const int mf = ksplit_mf(a.M, a.N, S);
#define FMI_LAUNCH_SPLIT(EPI_, MF_, SMEM_)                                                                                     \
    do {                                                                                                                      \
      static const hipError_t e0 = hipFuncSetAttribute((const void*)linear_tiled_256p_kernel<EPI_, false, MF_>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_); \
      static const hipError_t e1 = hipFuncSetAttribute((const void*)linear_tiled_reduce_kernel<EPI_, MF_>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_ * 16384); \
      FMI_CHECK_HIP(e0); FMI_CHECK_HIP(e1);                                                                                   \
      const dim3 g3(cdiv(a.N / 16, 16), cdiv(a.M, 32 * MF_), S), g2(g3.x, g3.y);                                              \
      hipLaunchKernelGGL((linear_tiled_256p_kernel<EPI_, false, MF_>), g3, dim3(512), SMEM_, s, a);                            \
      hipLaunchKernelGGL((linear_tiled_reduce_kernel<EPI_, MF_>), g2, dim3(512), MF_ * 16384, s, a, S);                        \
    } while (0)
#define FMI_LAUNCH_SPLIT_MF(EPI_)                                   \
    do {                                                           \
      if (mf == 2) FMI_LAUNCH_SPLIT(EPI_, 2, 81920);               \
      else if (mf == 4) FMI_LAUNCH_SPLIT(EPI_, 4, 98304);          \
      else if (mf == 6) FMI_LAUNCH_SPLIT(EPI_, 6, 114688);         \
      else FMI_LAUNCH_SPLIT(EPI_, 8, 131072);                      \
    } while (0)
    if (a.epi == EPI_STORE) FMI_LAUNCH_SPLIT_MF(EPI_STORE);
    else if (a.epi == EPI_RESIDUAL) FMI_LAUNCH_SPLIT_MF(EPI_RESIDUAL);
    else FMI_LAUNCH_SPLIT_MF(EPI_SILU);
#undef FMI_LAUNCH_SPLIT_MF
#undef FMI_LAUNCH_SPLIT
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  FMI_REQUIRE(a.bias == nullptr, "linear_tiled: no bias epilogue (skinny kernel only)");
  FMI_REQUIRE(a.K % 32 == 0 && a.N % 16 == 0 && a.ldx % 8 == 0 && a.ldo % 4 == 0, "linear_tiled: bad shape");
  if (a.epi == EPI_SILU) FMI_REQUIRE(a.N % 32 == 0, "linear_tiled: SwiGLU needs N %% 32");
  dim3 grid(cdiv(a.N / 16, 8), cdiv(a.M, 128)), block(256);
  // A/B switch: FMI_GEMM = d (operands straight from L2), l (LDS-staged, 4 waves), w (LDS-staged, wave-specialised)
  static const char env_mode = []() { const char* e = getenv("FMI_GEMM"); return e ? e[0] : '\0'; }();
  char mode = force_direct ? 'd' : variant == 1 ? 'l' : variant == 2 ? 'w' : variant == 3 ? 'x' : variant == 4 ? 'y' : variant == 7 ? 'p' : variant == 8 ? 'q' : variant == 9 ? 'r' : variant == 10 ? 's' : variant == 11 ? 't' : variant == 12 ? 'u' : env_mode ? env_mode : FMI_GEMM_DEFAULT;
  if (mode == 'a') {
    // Round 4: the 256-column tiles of linear_tiled_256p_kernel, 256 / 192 / 128 / 64 rows by which finishes first on
    // 256 CUs.  Cost model fitted to the tools/gemm_bench sweep over 200 .. 16384 rows (profiles/r04_gemm_bench.txt
    // section 12): rounds of one work-group per CU (two for the 64-row form), a work-group of 256 / 192 / 128 / 64
    // rows costing 1.0 / 0.76 / 0.60 / 0.69 (per round of 512) when the chip is full and 1.0 / 0.83 / 0.69 / 0.59
    // in a single, under-filled round (a work-group streams its 256 weight columns whatever its row count).  At 8 x 200
    // rows: wqkv 98 -> 53 us, wo 74 -> 55, w1|w3 259 -> 185, w2 162 -> 132; at 8 x 2048 rows the layer 4.70 -> 3.28 ms.
    // All variants give identical bits, so the choice may depend on the row count without touching batch invariance.
    const int ct = cdiv(a.N, 256);
    auto cost = [&](int mf) {
      static const int w_full[5] = {0, 69, 60, 76, 100}, w_lone[5] = {0, 59, 69, 83, 100};
      const int wgs = ct * cdiv(a.M, 32 * mf), rounds = cdiv(wgs, mf == 2 ? 512 : 256);
      if (rounds == 1) return (mf == 2 && wgs > 256) ? 69 : w_lone[mf / 2];
      return rounds * w_full[mf / 2];
    };
    int best = 8;
    for (int mf = 6; mf >= 2; mf -= 2)
      if (cost(mf) < cost(best)) best = mf;
    mode = best == 8 ? 'p' : best == 6 ? 'u' : best == 4 ? 's' : 't';
  }
  if ((mode == 'w' || mode == 'x' || mode == 'y' || mode == 'p' || mode == 'q' || mode == 'r' || mode == 's' || mode == 't' || mode == 'u') && ((a.K >> 5) & 1)) mode = 'l';   // these loops take k-tiles in pairs
  if ((mode == 'y' || mode == 'p' || mode == 'q' || mode == 'r' || mode == 's' || mode == 't' || mode == 'u') && (a.ldo % 8 != 0 || (a.epi == EPI_RESIDUAL && a.ldr % 8 != 0) || a.N % 8 != 0))
    mode = 'w';         // the 256 x 256 kernels write 16-byte row chunks
  if (mode == 'r') {   // 256 x 256 tile on sixteen waves
    constexpr int smem_r = 2 * 65536;
    static const hipError_t r0 = hipFuncSetAttribute((const void*)linear_tiled_256w16_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_r);
    static const hipError_t r1 = hipFuncSetAttribute((const void*)linear_tiled_256w16_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_r);
    static const hipError_t r2 = hipFuncSetAttribute((const void*)linear_tiled_256w16_kernel<EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_r);
    FMI_CHECK_HIP(r0); FMI_CHECK_HIP(r1); FMI_CHECK_HIP(r2);
    dim3 grid_r(cdiv(a.N / 16, 16), cdiv(a.M, 256));
    if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_256w16_kernel<EPI_STORE>), grid_r, dim3(1024), smem_r, s, a);
    else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_256w16_kernel<EPI_RESIDUAL>), grid_r, dim3(1024), smem_r, s, a);
    else hipLaunchKernelGGL((linear_tiled_256w16_kernel<EPI_SILU>), grid_r, dim3(1024), smem_r, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  if (mode == 'q') mode = 'p';   // (W3 = true: weights two steps ahead through a third stage -- measured no faster, profiles/r04_gemm_bench.txt; not instantiated)
  if (mode == 'p' || mode == 's' || mode == 't' || mode == 'u') {   // 256 (u: 192, s: 128, t: 64) x 256 tile, products deferred across the barrier
#define FMI_LAUNCH_P(W3_, MF_, SMEM_)                                                                                              \
    do {                                                                                                                            \
      static const hipError_t y0 = hipFuncSetAttribute((const void*)linear_tiled_256p_kernel<EPI_STORE, W3_, MF_>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_);    \
      static const hipError_t y1 = hipFuncSetAttribute((const void*)linear_tiled_256p_kernel<EPI_RESIDUAL, W3_, MF_>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_); \
      static const hipError_t y2 = hipFuncSetAttribute((const void*)linear_tiled_256p_kernel<EPI_SILU, W3_, MF_>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_);     \
      FMI_CHECK_HIP(y0); FMI_CHECK_HIP(y1); FMI_CHECK_HIP(y2);                                                                      \
      const dim3 grid_y(cdiv(a.N / 16, 16), cdiv(a.M, 32 * MF_));                                                                   \
      if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_256p_kernel<EPI_STORE, W3_, MF_>), grid_y, dim3(512), SMEM_, s, a);  \
      else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_256p_kernel<EPI_RESIDUAL, W3_, MF_>), grid_y, dim3(512), SMEM_, s, a); \
      else hipLaunchKernelGGL((linear_tiled_256p_kernel<EPI_SILU, W3_, MF_>), grid_y, dim3(512), SMEM_, s, a);                      \
    } while (0)
    // LDS = two stages of (32 MF rows x 128 B activations + 32 KiB weights); the epilogue's 8 x MF x 2 KiB fit inside
    if (mode == 's') FMI_LAUNCH_P(false, 4, 98304);
    else if (mode == 't') FMI_LAUNCH_P(false, 2, 81920);    // (80 KiB and 98 registers: two work-groups per CU)
    else if (mode == 'u') FMI_LAUNCH_P(false, 6, 114688);
    else FMI_LAUNCH_P(false, 8, 131072);
#undef FMI_LAUNCH_P
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  if (mode == 'y') {   // 256 x 256 tile, 8 waves, two 64 KiB stages
    constexpr int smem_y = 2 * 65536;
    static const hipError_t y0 = hipFuncSetAttribute((const void*)linear_tiled_256_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_y);
    static const hipError_t y1 = hipFuncSetAttribute((const void*)linear_tiled_256_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_y);
    static const hipError_t y2 = hipFuncSetAttribute((const void*)linear_tiled_256_kernel<EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_y);
    FMI_CHECK_HIP(y0); FMI_CHECK_HIP(y1); FMI_CHECK_HIP(y2);
    dim3 grid_y(cdiv(a.N / 16, 16), cdiv(a.M, 256));
    if (a.epi == EPI_STORE) hipLaunchKernelGGL((linear_tiled_256_kernel<EPI_STORE>), grid_y, dim3(512), smem_y, s, a);
    else if (a.epi == EPI_RESIDUAL) hipLaunchKernelGGL((linear_tiled_256_kernel<EPI_RESIDUAL>), grid_y, dim3(512), smem_y, s, a);
    else hipLaunchKernelGGL((linear_tiled_256_kernel<EPI_SILU>), grid_y, dim3(512), smem_y, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
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
