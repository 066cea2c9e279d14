// dualar_kernels.hip -- hand-written gfx950 kernels of the Dual-AR decode step.
//
// Reference semantics (fish_speech/models/text2semantic/):
//   embed            llama.py:400-420        RMSNorm          llama.py:990-1001
//   linear layers    llama.py:895,946,979-987 (bf16 in, fp32 accumulate, bf16 out)
//   q/k head norm    llama.py:862-864,901-903 RoPE            llama.py:1004-1038
//   KV cache         llama.py:196-214        slow attention   llama.py:928-934 (MATH backend)
//   fast attention   llama.py:948-976        sampler          inference.py:43-93,118-144
//
// Design notes (MI355X): the decode step is a weight-streaming problem (15.5 GB of bf16 weights
// per frame shared by all utterances, arithmetic intensity ~= batch).  Weights are pre-tiled at
// load time into MFMA-fragment order so that ONE wave instruction (global_load_dwordx4, 64 lanes)
// fetches ONE contiguous 1 KiB 16x32 tile that feeds v_mfma_f32_16x16x32_bf16 directly -- no LDS
// round trip for operands that are used once (guide: "GEMV / M<=16: load straight to VGPRs").
// The batch sits in the MFMA N dimension (16 columns), so results are batch-invariant:
// an utterance's numbers do not depend on which other utterances share the step.
#include "dualar_kernels.h"
#include "dualar_dev.h"

namespace fmi {

// =====================================================================================
// weight packing
// =====================================================================================

// Packed layout [N/16][K/32][64 lanes][8]: one 1 KiB tile = one v_mfma_f32_16x16x32_bf16 A operand, lane = kg*16 + n%16
// holds 8 consecutive k.  Inside every PAIR of k-tiles (64 k values = 2 tiles x 4 lane groups x 8) the 8-element
// chunks are dealt so that tile t (0/1) of pair p, lane group kg holds
//     k = 64 p + 32 (kg >> 1) + 8 (2 (kg & 1) + t) ... + 7.
// Reason (linear_skinny_kernel): with at most 8 utterances a wave fetches the activations of a whole pair with ONE
// all-lanes load of eight full cache lines (lane = kt*32 + g*8 + row reads x[row][64p + 32kt + 8g ..]); used
// directly as the MFMA B operand that register is column (g&1)*8 + row, lane group kt*2 + (g>>1), i.e. columns 0-7
// carry exactly the chunks of tile 0 and columns 8-15 those of tile 1.  An unpaired last k-tile (K/32 odd) keeps
// the plain order k = 32 j + 8 kg.  A reduction may take its k values in any order as long as both operands agree.
__device__ inline int64_t packed_index(int n, int k, int KT) {
  int j, kg;
  if ((k >> 5) < (KT & ~1)) {
    const int kt = (k >> 5) & 1, g = (k >> 3) & 3;
    j = ((k >> 6) << 1) + (g & 1);
    kg = kt * 2 + (g >> 1);
  } else {
    j = k >> 5;
    kg = (k >> 3) & 3;
  }
  return ((int64_t)(n >> 4) * KT + j) * 512 + (kg * 16 + (n & 15)) * 8 + (k & 7);
}

__global__ void pack_weight_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                   int interleave) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 elements
  int64_t total = (int64_t)N * K / 8;
  if (idx >= total) return;
  int kc = (int)(idx % (K / 8));
  int n = (int)(idx / (K / 8));
  int nd = n;
  if (interleave == 1) nd = (n >> 4) * 32 + (n & 15);
  if (interleave == 2) nd = (n >> 4) * 32 + 16 + (n & 15);
  uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)n * K + kc * 8);
  *reinterpret_cast<uint4*>(dst + packed_index(nd, kc * 8, K / 32)) = v;
}

int launch_pack_weight(const bf16_t* src, bf16_t* dst, int N, int K, int interleave, hipStream_t s) {
  FMI_REQUIRE(N % 16 == 0 && K % 32 == 0, "pack_weight: N=%d must be a multiple of 16 and K=%d of 32", N, K);
  int64_t total = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, N, K,
                     interleave);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// ---- row-balanced decode copy (linear_skinny_kernel<..., ROWS < 16>) ------------------------------------------------
// [tile][K/32][4 lane groups][ROWS rows][8]: tile t, k-tile j, lane group kg, row b holds the 8 weights the 16-row
// layout keeps at (row, k-tile j, lane group kg) -- same k permutation, so the activation operand logic is shared.
RowPlan skinny_row_plan(int N, int K, int epi) {
  RowPlan p{false, 16, 1, 0, 0};
  constexpr int CUS = 256;
  if ((K % 64) != 0 || K < 1024) return p;   // paired k-tiles only; tiny test models keep the 16-row tiles
  // w1|w3 (SwiGLU, N = 2 x 9728: 608 work-groups = 2.375 per CU) stays on the 16-row tiles: 19 gate + 19 up rows per
  // work-group x 512 work-groups (two zero-padded 10-row tile pairs, +5 % bytes) measured 21.65 us against 20.9-22.0
  // (profiles/r03_gemv_rows_bench.txt) -- a launch that already streams at 4.7 TB/s is not occupancy-bound
  if (epi == EPI_SILU || N % CUS != 0) return p;
  const int per = N / CUS;                   // rows per CU
  if (per < 1 || per > 32 || per % 16 == 0) return p;
  const int tiles = (per + 15) / 16;
  if (per % tiles != 0) return p;
  p = {true, per / tiles, tiles, CUS, 0};
  p.elems = (int64_t)p.wgs * p.tiles * p.rows * K;
  return p;
}

__global__ void repack_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K, int rows,
                                   int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 16-byte chunk of dst
  if (idx >= total) return;
  const int KT = K >> 5;
  const int b = (int)(idx % rows);
  const int kg = (int)((idx / rows) & 3);
  const int j = (int)((idx / (4 * rows)) % KT);
  const int tile = (int)(idx / ((int64_t)4 * rows * KT));
  const int nd = tile * rows + b;                                       // row of the 16-row packed source
  uint4 v = make_uint4(0, 0, 0, 0);
  if (nd < N) v = *reinterpret_cast<const uint4*>(src + ((int64_t)(nd >> 4) * KT + j) * 512 + (kg * 16 + (nd & 15)) * 8);
  *reinterpret_cast<uint4*>(dst + idx * 8) = v;
}

int launch_repack_rows(const bf16_t* packed16, bf16_t* dst, int N, int K, int epi, const RowPlan& plan, hipStream_t s) {
  FMI_REQUIRE(plan.ok, "repack_rows: no row-balanced plan for N=%d K=%d", N, K);
  const int64_t total = plan.elems / 8;
  hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, packed16, dst, N, K,
                     plan.rows, total);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void pack_weight_int8_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst, int N, int K,
                                        int interleave) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 elements
  int64_t total = (int64_t)N * K / 8;
  if (idx >= total) return;
  const int kc = (int)(idx % (K / 8)), n = (int)(idx / (K / 8));
  int nd = n;
  if (interleave == 1) nd = (n >> 4) * 32 + (n & 15);
  if (interleave == 2) nd = (n >> 4) * 32 + 16 + (n & 15);
  const int k = kc * 8, P = K >> 6;
  const int p = k >> 6, kt = (k >> 5) & 1, g = (k >> 3) & 3;
  const int t = g & 1, kg = kt * 2 + (g >> 1);                   // same chunk map as packed_index
  const int64_t o = ((((int64_t)(nd >> 4) * P + p) * 64 + kg * 16 + (nd & 15)) * 2 + t) * 8;
  *reinterpret_cast<uint2*>(dst + o) = *reinterpret_cast<const uint2*>(src + (int64_t)n * K + k);
}

int launch_pack_weight_int8(const int8_t* src, int8_t* dst, int N, int K, int interleave, hipStream_t s) {
  FMI_REQUIRE(N % 16 == 0 && K % 64 == 0, "pack_weight_int8: N=%d must be a multiple of 16 and K=%d of 64", N, K);
  int64_t total = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_weight_int8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, N, K,
                     interleave);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void dequant_int8_kernel(const int8_t* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f2bf((float)src[i]);   // |v| <= 128: exact in bf16
}

int launch_dequant_int8(const int8_t* src, bf16_t* dst, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(dequant_int8_kernel, dim3(2048), dim3(256), 0, s, src, dst, n);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void pack_scale_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int interleave) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int nd = n;
  if (interleave == 1) nd = (n >> 4) * 32 + (n & 15);
  if (interleave == 2) nd = (n >> 4) * 32 + 16 + (n & 15);
  dst[nd] = src[n];
}

int launch_pack_scale(const bf16_t* src, bf16_t* dst, int N, int interleave, hipStream_t s) {
  hipLaunchKernelGGL(pack_scale_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, src, dst, N, interleave);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void pack_rows_gather_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ ids,
                                        bf16_t* __restrict__ dst, int n, int n_pad, int K) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)n_pad * K / 8;
  if (idx >= total) return;
  int kc = (int)(idx % (K / 8));
  int r = (int)(idx / (K / 8));
  uint4 v = make_uint4(0, 0, 0, 0);
  if (r < n) v = *reinterpret_cast<const uint4*>(src + (int64_t)ids[r] * K + kc * 8);
  *reinterpret_cast<uint4*>(dst + packed_index(r, kc * 8, K / 32)) = v;
}

int launch_pack_rows_gather(const bf16_t* src, const int32_t* ids_dev, bf16_t* dst, int n, int n_pad, int K,
                            hipStream_t s) {
  FMI_REQUIRE(n_pad % 16 == 0 && K % 32 == 0, "pack_rows_gather: bad shape");
  int64_t total = (int64_t)n_pad * K / 8;
  hipLaunchKernelGGL(pack_rows_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ids_dev,
                     dst, n, n_pad, K);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// The RoPE table (llama.py:1004-1023) is built by the host mirror with torch, exactly as the reference does, and
// loaded through fmi_dualar_load_tensor ("rope", "fast_rope"); there is no device-side table kernel.

// =====================================================================================
// embedding (llama.py:400-420)
// =====================================================================================

__global__ void embed_kernel(EmbedArgs a) {
  const int r = blockIdx.x;
  const int32_t* tok = a.row_slot ? a.tokens + (int64_t)a.row_slot[r] * (a.ncb + 1)
                                  : a.tokens + (int64_t)r * (a.ncb + 1);
  const int t0 = tok[0];
  const bool sem = (t0 >= a.sem_begin) && (t0 <= a.sem_end);
  const float inv = sqrtf((float)(a.ncb + 1));  // x / math.sqrt(ncb+1), divisor rounded to fp32
  for (int d = threadIdx.x * 8; d < a.dim; d += blockDim.x * 8) {
    // torch.stack(embeds).sum(dim=1) on bf16: fp32 accumulation in codebook order, one rounding
    float vq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vq[j] = 0.f;
    if (sem) {  // all codebook rows are requested before the first is summed (one memory round trip, not ncb)
      constexpr int MAXCB = 16;
      uint4 cv[MAXCB];
#pragma unroll
      for (int i = 0; i < MAXCB; ++i)
        if (i < a.ncb) cv[i] = *reinterpret_cast<const uint4*>(a.cb_emb + (int64_t)(tok[i + 1] + i * a.cbs) * a.dim + d);
#pragma unroll
      for (int i = 0; i < MAXCB; ++i)
        if (i < a.ncb) {
          const bf16_t* e = reinterpret_cast<const bf16_t*>(&cv[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) vq[j] += bf2f(e[j]);
        }
      for (int i = MAXCB; i < a.ncb; ++i) {
        uint4 v = *reinterpret_cast<const uint4*>(a.cb_emb + (int64_t)(tok[i + 1] + i * a.cbs) * a.dim + d);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) vq[j] += bf2f(e[j]);
      }
    }
    uint4 ev = *reinterpret_cast<const uint4*>(a.emb + (int64_t)t0 * a.dim + d);
    const bf16_t* ee = reinterpret_cast<const bf16_t*>(&ev);
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = rbf(bf2f(ee[j]) + (sem ? rbf(vq[j]) : 0.f));
      if (a.scale && sem) x = rbf(x / inv);
      o[j] = f2bf(x);
    }
    *reinterpret_cast<uint4*>(a.out + (int64_t)r * a.dim + d) = *reinterpret_cast<uint4*>(o);
  }
}

int launch_embed(const EmbedArgs& a, hipStream_t s) {
  // one thread per 8-element chunk of the row when that fits a work-group: every row request of the gather is in
  // flight at once (dim 2560 = 320 chunks took two dependent trips with 256 threads)
  const int chunks = a.dim / 8;
  static const int env_t = []() { const char* e = getenv("FMI_EMBED_T"); return e ? atoi(e) : 0; }();
  const int threads = env_t > 0 ? env_t : (chunks <= 1024 ? ((chunks + 63) / 64) * 64 : 256);
  hipLaunchKernelGGL(embed_kernel, dim3(a.rows), dim3(threads), 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, int ld_src, const int32_t* __restrict__ row_idx,
                                   bf16_t* __restrict__ dst, int ld_dst, int cols) {
  int r = blockIdx.x;
  const bf16_t* s = src + (int64_t)row_idx[r] * ld_src;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8)
    *reinterpret_cast<uint4*>(dst + (int64_t)r * ld_dst + c) = *reinterpret_cast<const uint4*>(s + c);
}

int launch_gather_rows(const bf16_t* src, int ld_src, const int32_t* row_idx, bf16_t* dst, int ld_dst, int rows,
                       int cols, hipStream_t s) {
  FMI_REQUIRE(cols % 8 == 0, "gather_rows: cols %% 8");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, src, ld_src, row_idx, dst, ld_dst, cols);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// RMSNorm helpers (llama.py:990-1001): fp32 normalise -> cast -> * weight (bf16)
// =====================================================================================

// sum of squares of one row, one wave, fixed order (lane-strided 16-byte chunks, then xor tree).
__device__ inline float row_rstd(const bf16_t* __restrict__ x, int K, float eps, int lane) {
  float ss = 0.f;
  for (int c = lane * 8; c < K; c += 64 * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(x + c);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = bf2f(e[j]);
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  return rsqrtf(ss / (float)K + eps);
}

// two fp32 -> packed bf16 pair, round-to-nearest-even (one VALU instruction on gfx950; no builtin)
__device__ inline uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// RMSNorm of one 8-element fragment: bf16(bf16(x * rstd) * w), both roundings RNE like the reference's
// `.type_as(x) * weight` (llama.py:999-1001).  Works on packed pairs: ~6 VALU ops per element.
__device__ inline bf16x8 norm_frag(uint4 xv, uint4 wv, float rstd) {
  const uint32_t* xp = reinterpret_cast<const uint32_t*>(&xv);
  const uint32_t* wp = reinterpret_cast<const uint32_t*>(&wv);
  uint4 o;
  uint32_t* op = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = __uint_as_float(xp[j] << 16), x1 = __uint_as_float(xp[j] & 0xffff0000u);
    const uint32_t n = cvt_pk_bf16(x0 * rstd, x1 * rstd);
    const float n0 = __uint_as_float(n << 16), n1 = __uint_as_float(n & 0xffff0000u);
    const float w0 = __uint_as_float(wp[j] << 16), w1 = __uint_as_float(wp[j] & 0xffff0000u);
    op[j] = cvt_pk_bf16(n0 * w0, n1 * w1);
  }
  return *reinterpret_cast<bf16x8*>(&o);
}

__global__ void rmsnorm_rows_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ w, float eps,
                                    bf16_t* __restrict__ out, int ldo, int M, int K, bf16_t* __restrict__ out2) {
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int r = blockIdx.x * 4 + wave;
  if (r >= M) return;
  const bf16_t* xr = x + (int64_t)r * ldx;
  float rstd = row_rstd(xr, K, eps, lane);
  for (int c = lane * 8; c < K; c += 64 * 8) {
    uint4 xv = *reinterpret_cast<const uint4*>(xr + c);
    uint4 wv = *reinterpret_cast<const uint4*>(w + c);
    bf16x8 o = norm_frag(xv, wv, rstd);
    *reinterpret_cast<bf16x8*>(out + (int64_t)r * ldo + c) = o;
    if (out2) *reinterpret_cast<bf16x8*>(out2 + (int64_t)r * ldo + c) = o;
  }
}

int launch_rmsnorm_rows(const bf16_t* x, int ldx, const bf16_t* w, float eps, bf16_t* out, int ldo, int M, int K,
                        hipStream_t s, bf16_t* out2) {
  FMI_REQUIRE(K % 8 == 0, "rmsnorm: K %% 8");
  hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, w, eps, out, ldo, M, K, out2);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// skinny linear (M <= 16): weight-streaming MFMA GEMV
// =====================================================================================
//
// Work-group = WAVES waves, owns TILES 16-row weight tiles over the whole K; wave w owns the k-tile PAIRS
// [w*P/WAVES, (w+1)*P/WAVES) (P = K/64; the last wave also takes an unpaired last tile).  Per pair a lane issues
// two 16-byte weight loads per tile (the wave: two contiguous 1 KiB tiles) and the activation fragments of the pair with
// ONE all-lanes load of eight full cache lines per set of 8 rows (see packed_k0), used as the B operand of both tiles:
// tile 0 accumulates in one accumulator (valid in columns 0-7), tile 1 in a second one (valid in columns 8-15) that is
// shifted down 8 columns (DPP row_shl) and added at the end.  Rows 8-15 (M > 8, round 4) arrive by a second load and
// have their own accumulator pair; every row sees the same products in the same order whichever set it arrives in, so a
// row's result does not depend on the batch it is in.
// What bounds this kernel is requests in flight per CU, not bytes (tools/gemv_lds_probe.hip): halving the activation
// requests took w1|w3 21.8 -> 20.6, wqkv 10.4 -> 9.8, wo 7.1 -> 6.5, w2 14.7 -> 13.3 us (round 1).
// Split-K partials meet in LDS and are summed in wave order (deterministic).


// float(int8 in byte BYTE of w): byte select and sign extension are SDWA operand modifiers of the convert itself
template <int BYTE>
__device__ inline float cvt_i8_f32(uint32_t w) {
  float r;
  if constexpr (BYTE == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
  else if constexpr (BYTE == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
  else if constexpr (BYTE == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
  else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
  return r;
}

template <bool NT>
__device__ inline u32x4 wload(const u32x4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

__device__ inline float dpp_row_shl8(float v) {  // lane n of every 16-lane row receives lane n + 8
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x108, 0xf, 0xf, false));
}

__device__ inline float dpp_row_shr8(float v) {  // lane n of every 16-lane row receives lane n - 8
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x118, 0xf, 0xf, false));
}

constexpr int SKINNY_XH = 5;   // k-tile pairs of activations a wave can hold (K = 2560 over 8 waves: exactly 5)

// UNR = k-tile pairs in flight per wave; TILES = 16-row weight tiles per work-group (SwiGLU: gate/up tiles
// alternate, so TILES is even).
// XR = activation rows the launch can take, in sets of 8: 8 (M <= 8: one all-lanes load of eight full cache lines per
// pair), 16 (M <= 16, round 4: a second load brings rows 8-15; tile 0 / tile 1 of the pair then accumulate rows 8-15 in a
// second accumulator pair, columns 0-7 / 8-15 again) or 32 (round 6: four sets -- the merged fast positions 0/1 of a
// batch of 9-16).  A row's products and their order are the same in every form and the same whichever load the row
// arrives in, so its result does not depend on the batch it is in.
// (Round 6 also built the "native" 16-row form -- 16 activation rows in the 16 MFMA columns, half the products and
// accumulators of two 8-row sets, bit-identical -- and measured it SLOWER: w1|w3 at M = 16 30.5 us against 25.7, because
// its two loads per pair touch 16 half lines each where a set's load touches 8 full lines; the GEMV is bound by cache-line
// requests in flight per CU, not by products.  profiles/r06_gemv_native16.txt.)
// Q8: the weights are the int8 tiles of a weight-only-int8 checkpoint (a.wq, launch_pack_weight_int8): one 16-byte
// load per lane brings both k-tiles of a pair, converted to bf16 in registers (exact: |v| <= 128) right before the
// same MFMAs -- the products, their order and hence the result bits equal the bf16 kernel on the dequantised
// weights, at half the streamed bytes.  a.scale (any variant) applies the per-row scale of the int8 linear.
//
// ROWS < 16 (row-balanced decode copy, launch_repack_rows): a tile carries only ROWS weight rows, so that N / ROWS
// tiles divide evenly over the 256 CUs (N = 2560: 256 work-groups of 10 rows instead of 160 of 16).  The MFMA still
// multiplies a 16-row A operand: lanes of rows ROWS..15 re-read row ROWS-1 (same cache lines, no extra request) and
// their products land in accumulator rows nobody reads.  Rows 0..ROWS-1 see the same products in the same order as
// in the 16-row layout, so the result bits do not depend on ROWS.
//
// XH > 0 (round 4, NORM only, K % 64 == 0, at most XH pairs per wave): the wave requests the activation fragments of
// its WHOLE k-slice first thing (K = 2560: five 16-byte loads per lane and row set) and the RMSNorm statistics come from
// those registers -- per lane a sequential sum over its chunks, an xor tree over the eight chunk lanes of a row, the
// waves' partials through LDS summed in wave order (the same tree in every variant, so still batch-invariant).  The
// round-1..3 prologue read every row twice (a statistics pass over the whole row, then the fragments) with a barrier and
// a second exposed round trip in between: 40 of the ~160 wave-level loads of a wqkv work-group and ~2 us of every
// norm-fused launch.  The weight stream runs as a ring of UNR pairs refilled load by load.  Measured in the frame
// (profiles/r04_gemv_ab.txt): w1|w3 21.1 -> 20.3, wqkv 10.35 -> 9.05, heads 9.0 -> 7.3 us.
//
// Epilogue operands that do not depend on the products (residual, int8 row scale, bias) are requested at the top of the
// kernel: loaded after the last barrier they were one more dependent L2 miss at the tail of every wo / w2 launch.
// KSL > 1 (tools/gemv_ksplit_probe.hip only, never launched by the library): gridDim.y = KSL work-groups share a row block,
// each over 1 / KSL of the k-tile pairs; partial sums meet in a.part, the last arriver (a.cnt) adds them in slice order and
// runs the epilogue.
template <int WAVES, int EPI, bool NORM, int UNR, int TILES, int XR, bool NT = true, bool Q8 = false, int ROWS = 16, int XH = 0, int KSL = 1>
__global__ __launch_bounds__(WAVES * 64, (EPI == EPI_SILU && WAVES == 8 && UNR == 1) ? (((XR == 8 || XH == 0) && XR != 32) ? 6 : 4) : 1) void linear_skinny_kernel(LinearArgs a) {
  static_assert(KSL == 1 || (!NORM && !Q8 && XR == 8), "k-slices: plain bf16 linears of up to 8 rows");
  static_assert(EPI != EPI_SILU || TILES % 2 == 0, "SwiGLU needs gate/up tile pairs");
  static_assert(XR == 8 || XR == 16 || XR == 32, "activation row sets of 8");
  static_assert(!(Q8 && XR != 8), "int8 tiles: one row set");
  static_assert(ROWS >= 1 && ROWS <= 16 && (ROWS == 16 || (!Q8 && EPI != EPI_SILU)), "row-balanced tiles: bf16, no SwiGLU");
  static_assert(XH == 0 || NORM, "held activation fragments belong to the norm-fused variants");
  constexpr int XS = XR / 8;          // all-lanes activation loads per k-tile pair
  constexpr int CS = XS > 1 ? XS / 2 : 1;   // 16-column groups of results: row sets 2 cs, 2 cs + 1
  constexpr int XHN = XH > 0 ? XH : 1;
  constexpr int TSTRIDE = ROWS * 4;   // u32x4 per (tile, k-tile): ROWS rows x 4 lane groups
  __shared__ float red[WAVES][TILES][CS * 256];
  __shared__ float s_rstd[32];
  __shared__ float s_part[WAVES][32];
  __shared__ uint4 s_nw[(XH > 0 ? XH : 0) * WAVES * 8 + 1];   // XH > 0: the norm weight vector (K / 8 chunks of 8)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: k-slice bounds and their branches go scalar
  const int b = lane & 15, g = lane >> 4;
  const int KT = a.K >> 5, P = KT >> 1;
  const int tile0 = blockIdx.x * TILES;

  const int Psl = KSL > 1 ? P / KSL : P;                 // pairs of this work-group's k-slice
  const int pz0 = KSL > 1 ? (int)blockIdx.y * Psl : 0;
  const int pbeg = pz0 + (int)((int64_t)wave * Psl / WAVES), pend = pz0 + (int)((int64_t)(wave + 1) * Psl / WAVES);
  // weight addressing: a wave-uniform 64-bit base per (tile, k-tile) + one 32-bit per-lane byte offset (SGPR base +
  // VGPR offset loads: per-lane 64-bit pointers cost the norm-fused w1|w3 variant its third work-group per CU).
  // bf16: [tile][KT][TSTRIDE] u32x4; int8: [tile][P][64] u32x4 (a pair per entry)
  const char* __restrict__ wbase = Q8 ? reinterpret_cast<const char*>(a.wq) : reinterpret_cast<const char*>(a.wp);
  const uint32_t wlane = (uint32_t)(Q8 || ROWS == 16 ? lane : g * ROWS + min(b, ROWS - 1)) * 16u;
  auto wptr = [&](int t, int j) -> const u32x4* {   // bf16: k-tile j of tile t; int8: pair j of tile t
    const int64_t unit = Q8 ? ((int64_t)(tile0 + t) * P + j) * 64 : ((int64_t)(tile0 + t) * KT + j) * TSTRIDE;
    return reinterpret_cast<const u32x4*>(wbase + unit * 16 + wlane);
  };

  // activation fragment addressing: fetch lane = kt*32 + g'*8 + row -> x[8 s + row][64 p + 32 kt + 8 g' ..] for row
  // set s; as MFMA B operand that register is column (g'&1)*8 + row, lane group kt*2 + (g'>>1) (see packed_k0)
  const int f_off = (lane >> 5) * 32 + ((lane >> 3) & 3) * 8;
  const char* __restrict__ xbase = reinterpret_cast<const char*>(a.x);
  uint32_t xoff[XS];
#pragma unroll
  for (int s = 0; s < XS; ++s) xoff[s] = (uint32_t)(min(s * 8 + (lane & 7), a.M - 1) * a.ldx + f_off) * 2u;
  auto xptr = [&](int s, int p) -> const uint4* { return reinterpret_cast<const uint4*>(xbase + (int64_t)p * 128 + xoff[s]); };
  const char* __restrict__ nbase = reinterpret_cast<const char*>(a.norm_w);

  // XH > 0: the norm weight vector (K / 8 = XH * WAVES * 8 <= WAVES * 64 chunks: one per thread) is requested FIRST, so
  // that its wait -- loads return in order -- covers nothing issued after it.  Loaded inside the statistics block it sat
  // behind the first weight tiles: an s_waitcnt vmcnt(0) there made every norm-fused launch wait for its first HBM
  // round trip BEFORE the row statistics instead of beside them.
  uint4 nw_stage = make_uint4(0, 0, 0, 0);
  const bool nw_has = XH > 0 && NORM && tid < (a.K >> 3);
  if (nw_has && a.late_epi != 2) nw_stage = reinterpret_cast<const uint4*>(a.norm_w)[tid];   // (FMI_GEMV_LATE_EPI=2: A/B)
  uint4 xh[XS][XHN];   // XH > 0: the wave's activation fragments, raw, then normalised in place
  if (XH > 0) {
#pragma unroll
    for (int u = 0; u < XHN; ++u)
#pragma unroll
      for (int s = 0; s < XS; ++s)
        xh[s][u] = *xptr(s, pbeg + u);   // XH > 0: every wave owns exactly XH pairs (launcher)
  }

  // first weights of the wave: issued BEFORE the RMSNorm prologue so that HBM latency overlaps the row statistics
  u32x4 wa[TILES][UNR][2];
  auto load_pair = [&](int slot, int p) {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      if (Q8) {
        wa[t][slot][0] = wload<NT>(wptr(t, p));
      } else {
        wa[t][slot][0] = wload<NT>(wptr(t, 2 * p));
        wa[t][slot][1] = wload<NT>(wptr(t, 2 * p + 1));
      }
    }
  };
  const int nfull = (pend - pbeg) / UNR;
  if (XH > 0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (u < XHN) load_pair(u, pbeg + u);
    // (pinned here: without a fence the scheduler sinks these requests below the row statistics -- they have no user
    // before the products -- and the first HBM round trip of the launch starts a microsecond late)
    __builtin_amdgcn_sched_barrier(0);
  } else if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) load_pair(u, pbeg + u);
  }

  // epilogue operands (see above); a.late_epi (A/B measurements only) loads them where rounds 1-3 did
  // (thread (e_bb, e_r) of the first 256 finishes output row e_r of every tile for activation rows e_bb, e_bb + 16)
  const int e_bb = tid >> 4, e_r = tid & 15;
  const bool e_thr = tid < 256 && e_r < ROWS;
  const bool e_on = e_thr && e_bb < a.M;
  constexpr int NRES = EPI == EPI_RESIDUAL ? TILES : 1, NSC = Q8 ? TILES : 1, NBIAS = EPI == EPI_STORE ? TILES : 1;
  bf16_t e_res[CS][NRES], e_scale[NSC], e_bias[NBIAS];   // (the scale of a bf16-dequantised int8 linear, !Q8, stays a late load)
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int es = 0; es < CS; ++es)
      if (t < NRES) e_res[es][t] = 0;
    if (t < NSC) e_scale[t] = 0;
    if (t < NBIAS) e_bias[t] = 0;
    if (e_on && a.late_epi != 1) {
      if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int es = 0; es < CS; ++es)
          if (es * 16 + e_bb < a.M) e_res[es][t < NRES ? t : 0] = a.res[(int64_t)(es * 16 + e_bb) * a.ldr + (tile0 + t) * ROWS + e_r];
      }
      if (Q8 && a.scale) e_scale[t < NSC ? t : 0] = a.scale[(tile0 + t) * 16 + e_r];
      if (EPI == EPI_STORE && a.bias) e_bias[t < NBIAS ? t : 0] = a.bias[(tile0 + t) * ROWS + e_r];
    }
  }

  // int8 pair register (tile 0's 8 values | tile 1's 8 values) -> the two bf16 A operands.  One SDWA convert per
  // weight (byte select + sign extension inside v_cvt_f32_i32) and one pack per two: the compiler's own sequence for
  // (float)(int8_t)(w >> 8n) is shift + v_bfe_i32 + convert, 3.5 VALU instructions per weight against 1.5.
  auto unpack_q8 = [](const u32x4& q, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const uint32_t w0 = q[d], w1 = q[2 + d];
#ifdef FMI_Q8_PLAIN_CVT
      lo[2 * d] = cvt_pk_bf16((float)(int8_t)(w0), (float)(int8_t)(w0 >> 8));
      lo[2 * d + 1] = cvt_pk_bf16((float)(int8_t)(w0 >> 16), (float)(int8_t)(w0 >> 24));
      hi[2 * d] = cvt_pk_bf16((float)(int8_t)(w1), (float)(int8_t)(w1 >> 8));
      hi[2 * d + 1] = cvt_pk_bf16((float)(int8_t)(w1 >> 16), (float)(int8_t)(w1 >> 24));
#else
      lo[2 * d] = cvt_pk_bf16(cvt_i8_f32<0>(w0), cvt_i8_f32<1>(w0));
      lo[2 * d + 1] = cvt_pk_bf16(cvt_i8_f32<2>(w0), cvt_i8_f32<3>(w0));
      hi[2 * d] = cvt_pk_bf16(cvt_i8_f32<0>(w1), cvt_i8_f32<1>(w1));
      hi[2 * d + 1] = cvt_pk_bf16(cvt_i8_f32<2>(w1), cvt_i8_f32<3>(w1));
#endif
    }
  };

  float rstd[XS];
#pragma unroll
  for (int s = 0; s < XS; ++s) rstd[s] = 0.f;
  if (NORM && XH > 0) {
    // the norm weights travel through LDS (one 16-byte load per thread -- requested at the top of the kernel --, read
    // back as broadcasts after the statistics barrier): held in registers next to the fragments they cost the w1|w3
    // launch its third work-group per CU
    // sum of squares: lane (chunk c = lane >> 3, row) sequentially over its pairs and elements, then the eight chunk
    // lanes of the row by an xor tree (c bit 0, 1, 2), then the waves in order
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      float ss = 0.f;
#pragma unroll
      for (int u = 0; u < XHN; ++u) {
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&xh[s][u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f(e[j]);
          ss = fmaf(f, f, ss);
        }
      }
      ss += __shfl_xor(ss, 8, 64);
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 8) s_part[wave][s * 8 + lane] = ss;
    }
    if (nw_has && a.late_epi == 2) nw_stage = reinterpret_cast<const uint4*>(a.norm_w)[tid];   // where round 4 began: behind the weight requests
    if (nw_has) s_nw[tid] = nw_stage;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += s_part[w][s * 8 + (lane & 7)];
      rstd[s] = rsqrtf(tot / (float)a.K + a.eps);
#pragma unroll
      for (int u = 0; u < XHN; ++u) {
        const bf16x8 nf = norm_frag(xh[s][u], s_nw[(pbeg + u) * 8 + (lane >> 3)], rstd[s]);
        xh[s][u] = *reinterpret_cast<const uint4*>(&nf);
      }
    }
  } else if (NORM && (a.K & 63) == 0 && (a.K >> 6) == SKINNY_XH * WAVES) {
    // A shape the held-fragment variants (XH > 0) also serve, e.g. the 9-16-row SwiGLU form next to the <= 8-row one:
    // the statistics take THEIR partition and reduction tree -- wave w over its own k-slice, lane (chunk, row)
    // sequentially over its pairs and elements, xor tree over the eight chunk lanes, waves summed in order -- so that a
    // row's rstd, hence every bit of its result, does not depend on which variant the row count of the call selects
    // (row_rstd's lane-strided sum + wave_sum is another fp32 order).  The fragments are not held: they are
    // re-read (L2 hits) by the product loop below.
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      uint4 xt[SKINNY_XH];
#pragma unroll
      for (int u = 0; u < SKINNY_XH; ++u) xt[u] = *xptr(s, pbeg + u);
      float ss = 0.f;
#pragma unroll
      for (int u = 0; u < SKINNY_XH; ++u) {
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&xt[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = bf2f(e[j]);
          ss = fmaf(f, f, ss);
        }
      }
      ss += __shfl_xor(ss, 8, 64);
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 8) s_part[wave][s * 8 + lane] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += s_part[w][s * 8 + (lane & 7)];
      rstd[s] = rsqrtf(tot / (float)a.K + a.eps);
    }
  } else if (NORM) {
    for (int r = wave; r < a.M; r += WAVES) {
      float v = row_rstd(a.x + (int64_t)r * a.ldx, a.K, a.eps, lane);
      if (lane == 0) s_rstd[r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < XS; ++s) rstd[s] = s_rstd[min(s * 8 + (lane & 7), a.M - 1)];
  }

  f32x4 acc[2 * XS][TILES];   // [2 s + tile-of-pair]: row set s, valid in columns 0-7 (tile 0) / 8-15 (tile 1)
#pragma unroll
  for (int i = 0; i < 2 * XS; ++i)
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto frag = [&](int s, int p) -> bf16x8 {  // activation (optionally normalised) fragment of k-tile pair p
    uint4 xv = *xptr(s, p);
    if (NORM) {
      uint4 nv = *reinterpret_cast<const uint4*>(nbase + (int64_t)p * 128 + (uint32_t)f_off * 2u);
      return norm_frag(xv, nv, rstd[s]);
    }
    return *reinterpret_cast<bf16x8*>(&xv);
  };
  auto mma_pair = [&](int slot, const bf16x8* xs) {   // one k-tile pair: weights of ring slot `slot`, fragments xs[XS]
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      u32x4 w0 = wa[t][slot][0], w1 = wa[t][slot][1];
      if (Q8) unpack_q8(wa[t][slot][0], w0, w1);
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        acc[2 * s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w0), xs[s], acc[2 * s][t], 0, 0, 0);
        acc[2 * s + 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w1), xs[s], acc[2 * s + 1][t], 0, 0, 0);
      }
    }
  };

  if (XH > 0) {
    // ring of UNR pairs, refilled load by load: a k-tile's register is requested again (for pair u + UNR) right after
    // the products that consumed it, so the wave never has fewer than (2 TILES UNR - 1) tile loads in flight
#pragma unroll
    for (int u = 0; u < XHN; ++u) {
      const int slot = u % UNR;
      const bool more = u + UNR < XHN;
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        if (Q8) {
          u32x4 w0, w1;
          unpack_q8(wa[t][slot][0], w0, w1);
          if (more) wa[t][slot][0] = wload<NT>(wptr(t, pbeg + u + UNR));
#pragma unroll
          for (int s = 0; s < XS; ++s) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(&xh[s][u]);
            acc[2 * s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w0), xv, acc[2 * s][t], 0, 0, 0);
            acc[2 * s + 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&w1), xv, acc[2 * s + 1][t], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int s = 0; s < XS; ++s)
              acc[2 * s + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wa[t][slot][h]),
                                                                           *reinterpret_cast<const bf16x8*>(&xh[s][u]), acc[2 * s + h][t], 0, 0, 0);
            if (more) wa[t][slot][h] = wload<NT>(wptr(t, 2 * (pbeg + u + UNR) + h));
          }
        }
      }
    }
  } else {
    int p = pbeg;
    for (int c = 0; c < nfull; ++c) {
      bf16x8 xs[UNR][XS];
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int s = 0; s < XS; ++s) xs[u][s] = frag(s, p + u);
      // every fragment of the chunk is requested before its first product (without this fence the scheduler sinks the
      // second pair's load below the first pair's products: one more exposed L2 round trip per chunk, wo / w2 +2 us)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) mma_pair(u, xs[u]);
      p += UNR;
      if (c + 1 < nfull) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) load_pair(u, p + u);
      } else {   // the wave's leftover pairs (fewer than UNR) are requested like one more chunk
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (p + u < pend) load_pair(u, p + u);
      }
    }
    if (p < pend) {  // leftover pairs of this wave: ONE round trip (rounds 1-3 took them one at a time, a trip each)
      if (nfull == 0) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (p + u < pend) load_pair(u, p + u);
      }
      bf16x8 xs[UNR][XS];
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (p + u < pend) {
#pragma unroll
          for (int s = 0; s < XS; ++s) xs[u][s] = frag(s, p + u);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (p + u < pend) mma_pair(u, xs[u]);
    }
  }
  // fold: tile 1's sums sit in columns 8-15 of the same rows; of every two row sets the second (rows 8-15 of the group of 16)
  // moves to columns 8-15: acc[4 cs] then holds activation rows 16 cs .. 16 cs + 15 in its 16 columns
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        float v = acc[4 * cs][t][j] + dpp_row_shl8(acc[4 * cs + 1][t][j]);
        if (XS >= 2) {
          const float v1 = acc[XS >= 2 ? 4 * cs + 2 : 0][t][j] + dpp_row_shl8(acc[XS >= 2 ? 4 * cs + 3 : 0][t][j]);
          const float v1s = dpp_row_shr8(v1);   // every lane executes the DPP move (a disabled source lane reads as invalid)
          v = (b < 8) ? v : v1s;
        }
        acc[4 * cs][t][j] = v;
      }
    }
  if (!Q8 && (KT & 1) && wave == WAVES - 1) {  // unpaired last k-tile (plain k order), after the fold: same in every variant
    const int kt = KT - 1;
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
      const int row = cs * 16 + b < a.M ? cs * 16 + b : 0;
      uint4 xv = *reinterpret_cast<const uint4*>(a.x + (int64_t)row * a.ldx + kt * 32 + g * 8);
      bf16x8 xb;
      if (NORM) {
        uint4 nv = *reinterpret_cast<const uint4*>(a.norm_w + kt * 32 + g * 8);
        xb = norm_frag(xv, nv, s_rstd[row]);
      } else {
        xb = *reinterpret_cast<bf16x8*>(&xv);
      }
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        u32x4 wv = wload<NT>(wptr(t, kt));
        acc[4 * cs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv), xb, acc[4 * cs][t], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) *reinterpret_cast<f32x4*>(&red[wave][t][cs * 256 + lane * 4]) = acc[4 * cs][t];
  __syncthreads();

  if constexpr (KSL > 1) {
    // this slice's sums (wave order, as below) -> a.part; with a.cnt the last of the KSL arrivals of the row block adds the
    // slices in slice order -- the same bits whichever work-group happens to arrive last -- and goes on to the epilogue
    __shared__ int s_last;
    float* mine = a.part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (TILES * 256);
    if (tid < 256) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sacc += red[w][t][tid];
        if (a.part_proto == 1) __hip_atomic_store(reinterpret_cast<unsigned*>(mine) + t * 256 + tid, __float_as_uint(sacc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else mine[t * 256 + tid] = sacc;
      }
    }
    if (a.cnt == nullptr) return;
    if (a.part_proto == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores are acknowledged
    else __threadfence();
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = a.part_proto == 1 ? __hip_atomic_fetch_add(&a.cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : __hip_atomic_fetch_add(&a.cnt[blockIdx.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      s_last = prev == (unsigned)(KSL - 1);
      if (s_last) a.cnt[blockIdx.x] = 0u;   // (ready for the next launch: stream order separates launches)
    }
    __syncthreads();
    if (!s_last) return;
    if (a.part_proto != 1) __threadfence();
    if (tid < 256) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        float pz[KSL];
#pragma unroll
        for (int z = 0; z < KSL; ++z) {
          const float* src = a.part + ((int64_t)z * gridDim.x + blockIdx.x) * (TILES * 256) + t * 256 + tid;
          pz[z] = a.part_proto == 1 ? __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                    : __builtin_nontemporal_load(src);
        }
        float sacc = 0.f;
#pragma unroll
        for (int z = 0; z < KSL; ++z) sacc += pz[z];
        red[0][t][tid] = sacc;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) red[w][t][tid] = 0.f;
      }
    }
    __syncthreads();
  }

  if (e_thr) {
#pragma unroll
   for (int es = 0; es < CS; ++es) {
    const int bb = es * 16 + e_bb, r = e_r;  // consecutive threads -> consecutive output columns
    if (bb >= a.M) break;
    const int ridx = es * 256 + (((r >> 2) * 16) + e_bb) * 4 + (r & 3);
    float v[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      bf16_t sc = 0;
      if (a.late_epi == 1 || !Q8) {
        if (EPI == EPI_RESIDUAL && a.late_epi == 1) e_res[es][t < NRES ? t : 0] = a.res[(int64_t)bb * a.ldr + (tile0 + t) * ROWS + r];
        if (a.scale) sc = a.scale[(tile0 + t) * 16 + r];
        if (EPI == EPI_STORE && a.bias && a.late_epi == 1) e_bias[t < NBIAS ? t : 0] = a.bias[(tile0 + t) * ROWS + r];
      } else {
        sc = e_scale[t < NSC ? t : 0];
      }
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sacc += red[w][t][ridx];
      if (EPI == EPI_STORE && a.bias) sacc += bf2f(e_bias[t < NBIAS ? t : 0]);   // Linear bias: added before the ONE rounding
      v[t] = rbf(sacc);   // the linear's bf16 output ...
      if (a.scale) v[t] = rbf(v[t] * bf2f(sc));  // ... times the int8 row scale
    }
    if (EPI == EPI_STORE) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) a.out[(int64_t)bb * a.ldo + (tile0 + t) * ROWS + r] = f2bf(v[t]);
    } else if (EPI == EPI_RESIDUAL) {
#pragma unroll
      for (int t = 0; t < TILES; ++t)
        a.out[(int64_t)bb * a.ldo + (tile0 + t) * ROWS + r] = f2bf(bf2f(e_res[es][t < NRES ? t : 0]) + v[t]);
    } else {  // SwiGLU: even tiles = gate rows, odd tiles = up rows (llama.py:987)
#pragma unroll
      for (int t = 0; t < TILES; t += 2) {
        const int n = ((tile0 + t) >> 1) * 16 + r;
        float gate = rbf(silu_f(v[t]));
        float up = v[t + 1 < TILES ? t + 1 : t];
        a.out[(int64_t)bb * a.ldo + n] = f2bf(gate * up);
      }
    }
   }
  }
}

// FMI_GEMV_NOHOLD=1: the norm-fused variants keep the round-3 prologue (statistics pass, then fragments); A/B runs
static bool skinny_hold_enabled() {
  static const bool off = []() { const char* e = getenv("FMI_GEMV_NOHOLD"); return e && atoi(e) != 0; }();
  return !off;
}
static int skinny_late_epi() {   // FMI_GEMV_LATE_EPI=1: residual / scale / bias loaded in the epilogue as in rounds 1-3; 2: only the norm weights late
  static const int on = []() { const char* e = getenv("FMI_GEMV_LATE_EPI"); return e ? atoi(e) : 0; }();
  return on;
}
// Rows 9-16 (XR = 16): held fragments are TWO registers per pair (40 registers at K = 2560), which pushes the SwiGLU
// variant past 80 registers = two work-groups per CU, so its 608 work-groups need a second, nearly empty round (measured
// with round 4's two-row-set form: 31.7 us against 20.4 at M = 8); with the round-3 prologue it fits three work-groups
// per CU = one round.  FMI_GEMV_WIDE_HOLD: 0 = no wide variant holds, 1 (default) = all but SwiGLU hold, 2 = all hold
// (A/B runs).  Rows 17-32 (XR = 32, the merged fast pass of a batch of 9-16) never hold: 80 registers of fragments.
static int skinny_wide_hold() {
  static const int v = []() { const char* e = getenv("FMI_GEMV_WIDE_HOLD"); return e ? atoi(e) : 1; }();
  return v;
}
static bool skinny_can_hold(const LinearArgs& a, int waves) {
  if (a.M > 16) return false;
  if (a.M > 8 && (skinny_wide_hold() == 0 || (skinny_wide_hold() == 1 && a.epi == EPI_SILU))) return false;
  return a.norm_w != nullptr && (a.K % 64) == 0 && a.K / 64 == SKINNY_XH * waves && skinny_hold_enabled();
}

template <int WAVES, int UNR, int TILES>
static int launch_skinny_t(const LinearArgs& a0, hipStream_t s) {
  LinearArgs a = a0;
  a.late_epi = skinny_late_epi();
  const bool norm = a.norm_w != nullptr;
  const bool wide = a.M > 8, wide2 = a.M > 16;
  const bool hold = skinny_can_hold(a, WAVES);
  dim3 grid(a.N / (16 * TILES)), block(WAVES * 64);
#define FMI_LAUNCH_X(EPI_, NORM_, XH_)                                                                                       \
  do {                                                                                                                      \
    if (wide2) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 32, true, false, 16, 0>), grid, block, 0, s, a); \
    else if (wide) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 16, true, false, 16, XH_>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 8, true, false, 16, XH_>), grid, block, 0, s, a);       \
  } while (0)
#define FMI_LAUNCH(EPI_)                                          \
  do {                                                            \
    if (!norm) FMI_LAUNCH_X(EPI_, false, 0);                      \
    else if (hold) FMI_LAUNCH_X(EPI_, true, SKINNY_XH);           \
    else FMI_LAUNCH_X(EPI_, true, 0);                             \
  } while (0)
#define FMI_LAUNCH_Q8(EPI_)                                                                                                  \
  do {                                                                                                                      \
    if (!norm) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, false, UNR, TILES, 8, true, true, 16, 0>), grid, block, 0, s, a); \
    else if (hold) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, true, UNR, TILES, 8, true, true, 16, SKINNY_XH>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, true, UNR, TILES, 8, true, true, 16, 0>), grid, block, 0, s, a); \
  } while (0)
  if (a.wq && !wide && (a.K % 64) == 0) {   // weight-only int8 checkpoint: stream the int8 tiles
    if (a.epi == EPI_STORE) FMI_LAUNCH_Q8(EPI_STORE);
    else if (a.epi == EPI_RESIDUAL) FMI_LAUNCH_Q8(EPI_RESIDUAL);
    else if constexpr (TILES % 2 == 0) FMI_LAUNCH_Q8(EPI_SILU);
  } else if (a.epi == EPI_STORE) FMI_LAUNCH(EPI_STORE);
  else if (a.epi == EPI_RESIDUAL) FMI_LAUNCH(EPI_RESIDUAL);
  else if constexpr (TILES % 2 == 0) FMI_LAUNCH(EPI_SILU);
#undef FMI_LAUNCH
#undef FMI_LAUNCH_X
#undef FMI_LAUNCH_Q8
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// Row-balanced variants (bf16): the instantiations that exist, i.e. the S2-Pro decode shapes whose 16-row
// tilings leave CUs idle -- wo / w2 (N = 2560: 10 rows x 256 work-groups) and wqkv (N = 6144: 2 x 12 rows x 256).
// Measured on MI355X (tools/gemv_rows_bench.hip, profiles/r03_gemv_rows_bench.txt), bit-identical outputs:
// wo 6.5-6.6 -> 6.3 us, w2 12.4-14.7 -> 12.2, wqkv 10.3-10.5 -> 10.05; decode frame 4.73 -> 4.60 ms.
// Round 4: batches of 9-16 rows stream the same copies (XR = 16); round 6: 17-32 rows too (XR = 32).
template <int WAVES, int UNR, int TILES, int ROWS, int EPI_, bool NORM_>
static int launch_skinny_rows(const LinearArgs& a, const RowPlan& p, hipStream_t s) {
  LinearArgs b = a;
  b.wp = a.wr;
  b.late_epi = skinny_late_epi();
  constexpr int XH_ = NORM_ ? SKINNY_XH : 0;
  const bool hold = NORM_ && skinny_can_hold(a, WAVES);
  const dim3 grid(p.wgs), block(WAVES * 64);
  if (a.M > 16) {
    hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 32, true, false, ROWS, 0>), grid, block, 0, s, b);
  } else if (a.M > 8) {
    if (hold) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 16, true, false, ROWS, XH_>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 16, true, false, ROWS, 0>), grid, block, 0, s, b);
  } else {
    if (hold) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 8, true, false, ROWS, XH_>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 8, true, false, ROWS, 0>), grid, block, 0, s, b);
  }
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

bool skinny_rows_supported(int N, int K, int epi, bool norm) {
  const RowPlan p = skinny_row_plan(N, K, epi);
  if (!p.ok) return false;
  if (epi == EPI_RESIDUAL) return !norm && p.rows == 10 && p.tiles == 1;
  return norm && p.rows == 12 && p.tiles == 2;
}

// Variant choice from tools/gemv_bench.hip on MI355X (profiles/gemv_bench_r01.txt), M = 8; UNR counts k-tile pairs:
//   w13  (19456x2560, norm, SwiGLU)  8 waves, 1 pair,  2 tiles
//   wqkv (6144x2560, norm)           8 waves, 1 pair,  2 tiles
//   heads (4096..4128x2560, norm)    8 waves, 2 pairs, 1 tile
//   wo   (2560x4096, residual)       8 waves, 2 pairs, 1 tile
//   w2   (2560x9728, residual)       8 waves, 2 pairs, 1 tile
// A bare streaming read of the same bytes per launch reaches 3.6 / 4.1 / 4.4 / 5.1 TB/s at
// 21 / 32 / 50 / 100 MiB (about 3 us of every launch is ramp), 6.4-6.6 TB/s at 1 GiB.
// (The norm-fused variants were VALU-bound on software bf16 rounding until norm_frag moved to
// v_cvt_pk_bf16_f32: 26.4 -> 21.8 us and 13.5 -> 10.0 us.)
int launch_linear_skinny(const LinearArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.M >= 1 && a.M <= 32, "linear_skinny: M=%d not in [1,32]", a.M);
  FMI_REQUIRE(a.K % 32 == 0 && a.N % 16 == 0 && a.ldx % 8 == 0, "linear_skinny: bad shape N=%d K=%d", a.N, a.K);
  if (a.epi == EPI_SILU) FMI_REQUIRE(a.N % 32 == 0, "linear_skinny: SwiGLU needs N %% 32");
  const int KT = a.K / 32;
  const int ntile = a.N / 16;
  if (a.wr && !a.wq && !a.scale && !a.bias && skinny_rows_supported(a.N, a.K, a.epi, a.norm_w != nullptr)) {
    const RowPlan p = skinny_row_plan(a.N, a.K, a.epi);
    // (3 / 4 / 8 pairs in flight per wave instead of 2, measured in round 4: wo 6.56 / 6.46 / 7.02 against 6.57 us, w2
    // 11.44 / 11.82 / 12.61 against 11.69 -- the CU's ~256 outstanding cache lines are what is full, not the registers)
    if (a.epi == EPI_RESIDUAL) return launch_skinny_rows<8, 2, 1, 10, EPI_RESIDUAL, false>(a, p, s);
    return launch_skinny_rows<8, 1, 2, 12, EPI_STORE, true>(a, p, s);
  }
  if (a.wq && a.M <= 8 && a.K % 64 == 0 && KT >= 32) {
    // int8 stream (tools/gemv_q8_bench.hip, profiles/r02_gemv_q8.txt).  The wave count stays 8 -- the split-K
    // boundaries, hence the result bits, equal the bf16 kernel on the dequantised weights; pairs in flight and tiles
    // per work-group do not change the order of accumulation.  us, int8 vs bf16: w1|w3 15.4 vs 21.1, wqkv 8.9 vs 10.3,
    // wo 5.9 vs 7.4, w2 11.0 vs 14.7, heads 7.2 vs 8.4 -- once the int8 -> bf16 conversion is one SDWA convert per
    // weight; with the compiler's shift + bfe + convert sequence the kernel was VALU-bound and no faster than bf16.
    if (a.epi == EPI_SILU) return launch_skinny_t<8, 1, 2>(a, s);
    if (a.norm_w) {
      if (ntile % 2 == 0 && ntile > 320) return launch_skinny_t<8, 4, 2>(a, s);
      return launch_skinny_t<8, 2, 1>(a, s);
    }
    if (a.K <= 4096) return launch_skinny_t<8, 4, 1>(a, s);
    return launch_skinny_t<8, 2, 1>(a, s);
  }
  if (KT < 32) {  // tiny test models
    if (a.epi == EPI_SILU || ntile % 2 == 0) return launch_skinny_t<4, 1, 2>(a, s);
    return launch_skinny_t<4, 1, 1>(a, s);
  }
  if (a.epi == EPI_SILU) return launch_skinny_t<8, 1, 2>(a, s);
  if (a.norm_w) {  // norm-fused projections: wqkv (384 tiles) shares activations over 2 tiles; the heads (256-258
    // tiles: fast_output, live LM-head rows) would fill only half of the CUs that way (9.3 vs 8.2 us)
    if (ntile % 2 == 0 && ntile > 320) return launch_skinny_t<8, 1, 2>(a, s);
    return launch_skinny_t<8, 2, 1>(a, s);
  }
  return launch_skinny_t<8, 2, 1>(a, s);  // wo, w2, LM head
}


}  // namespace fmi
