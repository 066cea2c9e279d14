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
__host__ __device__ inline int packed_k0(int j, int kg, int KT) {  // first k of lane group kg in k-tile j
  if (j < (KT & ~1)) return (j >> 1) * 64 + (kg >> 1) * 32 + (((kg & 1) << 1) + (j & 1)) * 8;
  return j * 32 + kg * 8;
}

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

__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

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
// XR = activation rows per set of all-lanes loads: 8 (M <= 8: one load of eight full cache lines per pair) or 16
// (M <= 16, round 4: a second load brings rows 8-15; tile 0 / tile 1 of the pair then accumulate rows 8-15 in a second
// accumulator pair, columns 0-7 / 8-15 again).  A row's products and their order are the same in both forms and the same
// whichever of the two loads the row arrives in, so its result does not depend on the batch it is in.
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
template <int WAVES, int EPI, bool NORM, int UNR, int TILES, int XR, bool NT = true, bool Q8 = false, int ROWS = 16, int XH = 0>
__global__ __launch_bounds__(WAVES * 64, (EPI == EPI_SILU && WAVES == 8 && UNR == 1) ? (XR == 8 ? 6 : 4) : 1) void linear_skinny_kernel(LinearArgs a) {
  static_assert(EPI != EPI_SILU || TILES % 2 == 0, "SwiGLU needs gate/up tile pairs");
  static_assert(XR == 8 || XR == 16, "activation row sets of 8");
  static_assert(ROWS >= 1 && ROWS <= 16 && (ROWS == 16 || (!Q8 && EPI != EPI_SILU)), "row-balanced tiles: bf16, no SwiGLU");
  static_assert(XH == 0 || NORM, "held activation fragments belong to the norm-fused variants");
  constexpr int XS = XR / 8;          // all-lanes activation loads per k-tile pair
  constexpr int XHN = XH > 0 ? XH : 1;
  constexpr int TSTRIDE = ROWS * 4;   // u32x4 per (tile, k-tile): ROWS rows x 4 lane groups
  __shared__ float red[WAVES][TILES][256];
  __shared__ float s_rstd[16];
  __shared__ float s_part[WAVES][16];
  __shared__ uint4 s_nw[(XH > 0 ? XH : 0) * WAVES * 8 + 1];   // XH > 0: the norm weight vector (K / 8 chunks of 8)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: k-slice bounds and their branches go scalar
  const int b = lane & 15, g = lane >> 4;
  const int KT = a.K >> 5, P = KT >> 1;
  const int tile0 = blockIdx.x * TILES;

  const int pbeg = (int)((int64_t)wave * P / WAVES), pend = (int)((int64_t)(wave + 1) * P / WAVES);
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
  } else if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) load_pair(u, pbeg + u);
  }

  // epilogue operands (see above); a.late_epi (A/B measurements only) loads them where rounds 1-3 did
  const int e_bb = tid >> 4, e_r = tid & 15;
  const bool e_on = tid < 256 && e_bb < a.M && e_r < ROWS;
  constexpr int NRES = EPI == EPI_RESIDUAL ? TILES : 1, NSC = Q8 ? TILES : 1, NBIAS = EPI == EPI_STORE ? TILES : 1;
  bf16_t e_res[NRES], e_scale[NSC], e_bias[NBIAS];   // (the scale of a bf16-dequantised int8 linear, !Q8, stays a late load)
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    if (t < NRES) e_res[t] = 0;
    if (t < NSC) e_scale[t] = 0;
    if (t < NBIAS) e_bias[t] = 0;
    if (e_on && !a.late_epi) {
      if (EPI == EPI_RESIDUAL) e_res[t < NRES ? t : 0] = a.res[(int64_t)e_bb * a.ldr + (tile0 + t) * ROWS + e_r];
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
    // the norm weights travel through LDS (one 16-byte load per thread, read back as broadcasts after the statistics
    // barrier): held in registers next to the fragments they cost the w1|w3 launch its third work-group per CU
    for (int i = tid; i < (a.K >> 3); i += WAVES * 64) s_nw[i] = reinterpret_cast<const uint4*>(a.norm_w)[i];
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
      }
    }
    for (; p < pend; ++p) {  // leftover pairs of this wave, one at a time
      bf16x8 xs[XS];
#pragma unroll
      for (int s = 0; s < XS; ++s) xs[s] = frag(s, p);
      load_pair(0, p);
      mma_pair(0, xs);
    }
  }
  // fold: tile 1's sums sit in columns 8-15 of the same rows; row set 1 (rows 8-15) moves to columns 8-15
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[0][t][j] + dpp_row_shl8(acc[1][t][j]);
      if (XS == 2) {
        const float v1 = acc[2 * (XS - 1)][t][j] + dpp_row_shl8(acc[2 * (XS - 1) + 1][t][j]);
        const float v1s = dpp_row_shr8(v1);   // every lane executes the DPP move (a disabled source lane reads as invalid)
        v = (b < 8) ? v : v1s;
      }
      acc[0][t][j] = v;
    }
  if (!Q8 && (KT & 1) && wave == WAVES - 1) {  // unpaired last k-tile (plain k order), after the fold: same in every variant
    const int kt = KT - 1;
    const int row = b < a.M ? b : 0;
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
      acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wv), xb, acc[0][t], 0, 0, 0);
    }
  }

#pragma unroll
  for (int t = 0; t < TILES; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane * 4]) = acc[0][t];
  __syncthreads();

  if (e_on) {
    const int bb = e_bb, r = e_r;  // consecutive threads -> consecutive output columns
    const int ridx = (((r >> 2) * 16) + bb) * 4 + (r & 3);
    float v[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      bf16_t sc = 0;
      if (a.late_epi || !Q8) {
        if (EPI == EPI_RESIDUAL && a.late_epi) e_res[t < NRES ? t : 0] = a.res[(int64_t)bb * a.ldr + (tile0 + t) * ROWS + r];
        if (a.scale) sc = a.scale[(tile0 + t) * 16 + r];
        if (EPI == EPI_STORE && a.bias && a.late_epi) e_bias[t < NBIAS ? t : 0] = a.bias[(tile0 + t) * ROWS + r];
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
        a.out[(int64_t)bb * a.ldo + (tile0 + t) * ROWS + r] = f2bf(bf2f(e_res[t < NRES ? t : 0]) + v[t]);
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

// FMI_GEMV_NOHOLD=1: the norm-fused variants keep the round-3 prologue (statistics pass, then fragments); A/B runs
static bool skinny_hold_enabled() {
  static const bool off = []() { const char* e = getenv("FMI_GEMV_NOHOLD"); return e && atoi(e) != 0; }();
  return !off;
}
static bool skinny_late_epi() {   // FMI_GEMV_LATE_EPI=1: residual / scale / bias loaded in the epilogue as in rounds 1-3
  static const bool on = []() { const char* e = getenv("FMI_GEMV_LATE_EPI"); return e && atoi(e) != 0; }();
  return on;
}
static bool skinny_can_hold(const LinearArgs& a, int waves) {
  return a.norm_w != nullptr && (a.K % 64) == 0 && a.K / 64 == SKINNY_XH * waves && skinny_hold_enabled();
}

template <int WAVES, int UNR, int TILES>
static int launch_skinny_t(const LinearArgs& a0, hipStream_t s) {
  LinearArgs a = a0;
  a.late_epi = skinny_late_epi() ? 1 : 0;
  const bool norm = a.norm_w != nullptr;
  const bool wide = a.M > 8;
  const bool hold = skinny_can_hold(a, WAVES);
  dim3 grid(a.N / (16 * TILES)), block(WAVES * 64);
#define FMI_LAUNCH_X(EPI_, NORM_, XH_)                                                                                       \
  do {                                                                                                                      \
    if (wide) hipLaunchKernelGGL((linear_skinny_kernel<WAVES, EPI_, NORM_, UNR, TILES, 16, true, false, 16, XH_>), grid, block, 0, s, a); \
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
// Round 4: batches of 9-16 rows stream the same copies (XR = 16).
template <int WAVES, int UNR, int TILES, int ROWS, int EPI_, bool NORM_>
static int launch_skinny_rows(const LinearArgs& a, const RowPlan& p, hipStream_t s) {
  LinearArgs b = a;
  b.wp = a.wr;
  b.late_epi = skinny_late_epi() ? 1 : 0;
  constexpr int XH_ = NORM_ ? SKINNY_XH : 0;
  const bool hold = NORM_ && skinny_can_hold(a, WAVES);
  const dim3 grid(p.wgs), block(WAVES * 64);
  if (a.M > 8) {
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
  FMI_REQUIRE(a.M >= 1 && a.M <= 16, "linear_skinny: M=%d not in [1,16]", a.M);
  FMI_REQUIRE(a.K % 32 == 0 && a.N % 16 == 0 && a.ldx % 8 == 0, "linear_skinny: bad shape N=%d K=%d", a.N, a.K);
  if (a.epi == EPI_SILU) FMI_REQUIRE(a.N % 32 == 0, "linear_skinny: SwiGLU needs N %% 32");
  const int KT = a.K / 32;
  const int ntile = a.N / 16;
  if (a.wr && !a.wq && !a.scale && !a.bias && skinny_rows_supported(a.N, a.K, a.epi, a.norm_w != nullptr)) {
    const RowPlan p = skinny_row_plan(a.N, a.K, a.epi);
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

// =====================================================================================
// attention: prep (q/k head norm + RoPE + paged KV write), then attention over the cache
// =====================================================================================

// one wave per (row, head); lane p owns the RoPE pair (2p, 2p+1).  grid (rows, ceil(heads/4)).
__global__ __launch_bounds__(256) void attn_prep_kernel(AttnArgs a) {
  const int r = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int D = a.D, H = a.H, KVH = a.KVH;
  const int total = H + 2 * KVH;
  const int h = blockIdx.y * 4 + wave;
  if (h >= total) return;
  const int slot = a.row_slot[r];
  const int pos = a.row_pos ? a.row_pos[r] : a.slot_pos[slot];
  const bf16_t* src = a.qkv + (int64_t)r * total * D;
  const bool act = lane < D / 2;
  const int p = act ? lane : 0;
  uint32_t raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
  float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
  bf16_t o0 = (bf16_t)(raw & 0xffff), o1 = (bf16_t)(raw >> 16);
  if (h < H + KVH) {
    const bf16_t* nw = (h < H) ? a.qnw : a.knw;
    float y0 = x0, y1 = x1;
    if (nw) {  // torch.nn.RMSNorm: fp32 normalise * weight, one cast (llama.py:862-864)
      float ss = wave_sum(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(nw[2 * p])));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(nw[2 * p + 1])));
    }
    uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
    float c = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));
    // llama.py:1026-1038, separate fp32 mul / sub / add (no fused multiply-add)
    o0 = f2bf(__fsub_rn(__fmul_rn(y0, c), __fmul_rn(y1, sn)));
    o1 = f2bf(__fadd_rn(__fmul_rn(y1, c), __fmul_rn(y0, sn)));
  }
  if (!act) return;
  const uint32_t packed = (uint32_t)o0 | ((uint32_t)o1 << 16);
  if (h < H) {
    *reinterpret_cast<uint32_t*>(a.q + ((int64_t)r * H + h) * D + 2 * p) = packed;
  } else {
    const int page = a.block_table[(int64_t)slot * a.max_pages + pos / KV_PAGE];
    const int kh = (h < H + KVH) ? h - H : h - H - KVH;
    bf16_t* pool = (h < H + KVH) ? a.kpool : a.vpool;
    *reinterpret_cast<uint32_t*>(pool + (((int64_t)page * KVH + kh) * KV_PAGE + pos % KV_PAGE) * D + 2 * p) = packed;
  }
}

int launch_attn_prep(const AttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.D % 2 == 0 && a.D <= 128 && a.D >= 16, "attn_prep: head_dim=%d unsupported", a.D);
  hipLaunchKernelGGL(attn_prep_kernel, dim3(a.rows, cdiv(a.H + 2 * a.KVH, 4)), dim3(256), 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// grid (rows, KVH): one work-group handles the G query heads sharing one KV head.  LPT = D/8 lanes
// share a token (16 bytes each, coalesced 2*D-byte rows); a wave covers 64/LPT tokens per step.
// Online softmax per lane group; partial (m, l, acc) states are merged through LDS.
template <int D, int G>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  constexpr int LPT = D / 8, TPW = 64 / LPT, NP = 4 * TPW;
  __shared__ float s_m[NP][G], s_l[NP][G];
  __shared__ float s_acc[NP][G][D];
  const int r = blockIdx.x, kvh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPT, dl = lane % LPT;
  const int slot = a.row_slot[r];
  const int pos = a.row_pos ? a.row_pos[r] : a.slot_pos[slot];
  const int32_t* bt = a.block_table + (int64_t)slot * a.max_pages;
  const float scale = 1.0f / sqrtf((float)D);

  float q[G][8];
#pragma unroll
  for (int gq = 0; gq < G; ++gq) {
    uint4 v = *reinterpret_cast<const uint4*>(a.q + ((int64_t)r * a.H + kvh * G + gq) * D + dl * 8);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[gq][j] = bf2f(e[j]);
  }
  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int gq = 0; gq < G; ++gq) {
    m[gq] = -1e30f;
    l[gq] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[gq][j] = 0.f;
  }

  const int n_tok = pos + 1;
  for (int t0 = wave * TPW; t0 < n_tok; t0 += NP) {
    const int t = t0 + sub;
    const bool valid = t < n_tok;
    const int tc = valid ? t : pos;
    const int page = bt[tc / KV_PAGE];
    const int64_t base = (((int64_t)page * a.KVH + kvh) * KV_PAGE + (tc % KV_PAGE)) * D + dl * 8;
    uint4 kv = *reinterpret_cast<const uint4*>(a.kpool + base);
    uint4 vv = *reinterpret_cast<const uint4*>(a.vpool + base);
    const bf16_t* ke = reinterpret_cast<const bf16_t*>(&kv);
    const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vv);
    float kf[8], vf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      kf[j] = bf2f(ke[j]);
      vf[j] = bf2f(ve[j]);
    }
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += q[gq][j] * kf[j];
      d = group_allsum<LPT>(d);
      if (valid) {
        const float sc = d * scale;
        const float mn = fmaxf(m[gq], sc);
        const float corr = __expf(m[gq] - mn);
        const float p = __expf(sc - mn);
        l[gq] = l[gq] * corr + p;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[gq][j] = acc[gq][j] * corr + p * vf[j];
        m[gq] = mn;
      }
    }
  }

  const int pidx = wave * TPW + sub;
#pragma unroll
  for (int gq = 0; gq < G; ++gq) {
    if (dl == 0) {
      s_m[pidx][gq] = m[gq];
      s_l[pidx][gq] = l[gq];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_acc[pidx][gq][dl * 8 + j] = acc[gq][j];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < G * D; o += 256) {
    const int gq = o / D, d = o % D;
    float M = -1e30f;
    for (int p = 0; p < NP; ++p) M = fmaxf(M, s_m[p][gq]);
    float L = 0.f, O = 0.f;
    for (int p = 0; p < NP; ++p) {
      const float w = __expf(s_m[p][gq] - M);
      L += s_l[p][gq] * w;
      O += s_acc[p][gq][d] * w;
    }
    a.out[((int64_t)r * a.H + kvh * G + gq) * D + d] = f2bf(O / L);
  }
}

template <int D>
static int launch_attn_d(const AttnArgs& a, hipStream_t s) {
  const int G = a.H / a.KVH;
  dim3 grid(a.rows, a.KVH), block(256);
  switch (G) {
    case 1: hipLaunchKernelGGL((attn_kernel<D, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((attn_kernel<D, 2>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((attn_kernel<D, 4>), grid, block, 0, s, a); break;
    default: return set_error(FMI_EINVAL, "attn: GQA ratio %d unsupported", G);
  }
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int launch_attn(const AttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.H % a.KVH == 0, "attn: n_head %% n_local_heads");
  switch (a.D) {
    case 32: return launch_attn_d<32>(a, s);
    case 64: return launch_attn_d<64>(a, s);
    case 128: return launch_attn_d<128>(a, s);
    default: return set_error(FMI_EINVAL, "attn: head_dim %d unsupported (32/64/128)", a.D);
  }
}

// =====================================================================================
// prefill attention on MFMA with LDS-staged K/V tiles (llama.py:910-934, MATH-backend numerics: fp32 scores,
// fp32 softmax, fp32 accumulation, one bf16 rounding of the output)
// =====================================================================================
//
// grid (query tiles, KVH), G waves (G = n_head / n_local_heads <= 4): the work-group owns 16 consecutive query rows
// of one utterance and one kv head; wave w serves query head kvh*G + w, so all waves share the same K/V tiles.
// Keys come in blocks of 32 (a block never straddles a 64-token KV page).  Per block:
//   stage   K block -> s_k [32 keys][D] (row-major, 16-byte pad), V block -> s_vt TRANSPOSED [d][32 keys]
//           (two keys per thread packed into one dword; rows permuted j*(D/8+1)+dchunk so the eight transposing
//           stores of a thread's 8 d-values are bank-conflict free); the next block's global loads are issued
//           before the maths of the current one (register prefetch);
//   scores  S^T = K Q^T with v_mfma_f32_16x16x32_bf16: A = K rows (16 keys x 32 d, ds_read_b128), B = Q^T held in
//           registers for the whole kernel; a lane then holds, for ITS query column, 8 of the 32 keys -- which is
//           exactly a B-operand fragment of the next MFMA if the key order inside the block is permuted the same
//           way for V (so the probabilities never leave registers: no LDS round trip, no shuffles);
//   softmax online, fp32, per query column (reductions over the 4 lane groups by two xor-shuffles);
//   output  O^T += V^T P^T: A = V^T (16 d x 32 keys from s_vt, two ds_read_b64), B = P split into bf16 hi + lo
//           halves (two MFMAs: the weights keep ~16 significant bits, the reference multiplies fp32 weights).
// Causal tiles are skipped (key blocks beyond the tile's last position are never visited), the diagonal block is
// masked by select.  Tile descriptors (row0, rows, slot, first position) come from the host, heaviest first.
template <int D, int G>
__global__ __launch_bounds__(64 * G) void attn_prefill_mfma_kernel(AttnArgs a) {
  constexpr int KB = 32, NT = 64 * G, DC = D / 8;
  constexpr int KROW = D + 8;                 // bf16 elements per s_k row (+16 bytes)
  constexpr int VROWB = KB * 2 + 8;           // bytes per s_vt row (32 keys + 8 bytes pad)
  constexpr int VROWS = 8 * (DC + 1);         // permuted row index j*(DC+1) + dchunk
  constexpr int KCH = KB * DC / NT > 0 ? KB * DC / NT : 1;        // 16-byte K chunks per thread
  constexpr int VCH = (KB / 2) * DC / NT > 0 ? (KB / 2) * DC / NT : 1;  // key-pair chunks per thread
  static_assert((KB * DC) % NT == 0 || KB * DC < NT, "K staging");
  __shared__ __attribute__((aligned(16))) bf16_t s_k[KB * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char s_vt[VROWS * VROWB];

  const int4 td = a.qtiles[blockIdx.x];  // x row0, y rows (1..16), z slot, w first position
  const int kvh = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int H = a.H, KVH = a.KVH;
  const int head = kvh * G + wave;
  const int32_t* bt = a.block_table + (int64_t)td.z * a.max_pages;
  const int last_pos = td.w + td.y - 1;
  const int n_blocks = last_pos / KB + 1;
  const int qpos = td.w + c;

  // Q^T fragments (B operand: column = query c, k rows = d chunk g of k-step kk)
  bf16x8 qf[D / 32];
  {
    const int qrow = td.x + (c < td.y ? c : td.y - 1);
    const bf16_t* qp = a.q + ((int64_t)qrow * H + head) * D + g * 8;
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32);
  }

  f32x4 o[D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f;
  const float scale = 1.0f / sqrtf((float)D);

  // ---- staging helpers: K chunk i -> (key i / DC, dchunk i % DC); V chunk i -> (key pair i / DC, dchunk i % DC)
  uint4 kreg[KCH], vreg[VCH][2];
  auto fetch = [&](int kb) {
    const int page = bt[(kb * KB) / KV_PAGE];
    const int64_t pbase = ((int64_t)page * KVH + kvh) * KV_PAGE + (kb * KB) % KV_PAGE;
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
      const int i = tid + it * NT;
      if (i < KB * DC) {
        const int key = i / DC, dc = i % DC;
        kreg[it] = *reinterpret_cast<const uint4*>(a.kpool + (pbase + key) * D + dc * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = tid + it * NT;
      if (i < (KB / 2) * DC) {
        const int kp = i / DC, dc = i % DC;
        const bool v0 = kb * KB + 2 * kp <= last_pos, v1 = kb * KB + 2 * kp + 1 <= last_pos;
        // rows beyond the tile's last position have not been written (stale pool contents): they must read as 0,
        // a masked probability of 0 times a stale NaN/Inf would poison the accumulator
        vreg[it][0] = v0 ? *reinterpret_cast<const uint4*>(a.vpool + (pbase + 2 * kp) * D + dc * 8) : make_uint4(0, 0, 0, 0);
        vreg[it][1] = v1 ? *reinterpret_cast<const uint4*>(a.vpool + (pbase + 2 * kp + 1) * D + dc * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
      const int i = tid + it * NT;
      if (i < KB * DC) *reinterpret_cast<uint4*>(&s_k[(i / DC) * KROW + (i % DC) * 8]) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = tid + it * NT;
      if (i < (KB / 2) * DC) {
        const int kp = i / DC, dc = i % DC;
        const bf16_t* e0 = reinterpret_cast<const bf16_t*>(&vreg[it][0]);
        const bf16_t* e1 = reinterpret_cast<const bf16_t*>(&vreg[it][1]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<uint32_t*>(&s_vt[(j * (DC + 1) + dc) * VROWB + kp * 4]) = (uint32_t)e0[j] | ((uint32_t)e1[j] << 16);
      }
    }
  };

  fetch(0);
  for (int kb = 0; kb < n_blocks; ++kb) {
    __syncthreads();          // every wave is done reading the previous block's tiles
    stage();
    __syncthreads();
    if (kb + 1 < n_blocks) fetch(kb + 1);

    // ---- scores: two 16-key tiles, lane (c, g) ends up with keys kt*16 + g*4 + j of query column c
    f32x4 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk) {
        const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(&s_k[(kt * 16 + c) * KROW + kk * 32 + g * 8]);
        sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[kk], sacc[kt], 0, 0, 0);
      }
    }
    float sc[8];
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kb * KB + kt * 16 + g * 4 + j;
        const float v = key <= qpos ? sacc[kt][j] * scale : -1e30f;
        sc[kt * 4 + j] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float corr = __expf(m - mn);
    float ps = 0.f, pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pr[j] = sc[j] > -1e29f ? __expf(sc[j] - mn) : 0.f;
      ps += pr[j];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l = l * corr + ps;
    m = mn;
    // probabilities as the B operand (k slots g*8 + jj = keys {g*4 + jj | jj < 4} and {16 + g*4 + jj - 4}), hi + lo
    bf16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16_t hi = f2bf(pr[j]);
      ph[j] = (short)hi;
      pl[j] = (short)f2bf(pr[j] - bf2f(hi));
    }
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      const int d = dt * 16 + c;
      const unsigned char* row = &s_vt[((d & 7) * (DC + 1) + (d >> 3)) * VROWB];
      const uint2 lo = *reinterpret_cast<const uint2*>(row + g * 8);        // keys g*4 .. g*4+3
      const uint2 hi = *reinterpret_cast<const uint2*>(row + 32 + g * 8);   // keys 16 + g*4 ..
      u32x4 av = {lo.x, lo.y, hi.x, hi.y};
      const bf16x8 vfrag = *reinterpret_cast<bf16x8*>(&av);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] *= corr;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, ph, o[dt], 0, 0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pl, o[dt], 0, 0, 0);
    }
  }

  if (c < td.y) {   // lane holds query column c, output dims dt*16 + g*4 + j
    const float inv = 1.0f / l;
    bf16_t* op = a.out + ((int64_t)(td.x + c) * H + head) * D + g * 4;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      const uint32_t w0 = (uint32_t)f2bf(o[dt][0] * inv) | ((uint32_t)f2bf(o[dt][1] * inv) << 16);
      const uint32_t w1 = (uint32_t)f2bf(o[dt][2] * inv) | ((uint32_t)f2bf(o[dt][3] * inv) << 16);
      *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(w0, w1);
    }
  }
}

template <int D>
static int launch_attn_prefill_d(const AttnArgs& a, hipStream_t s) {
  const int G = a.H / a.KVH;
  dim3 grid(a.n_qtiles, a.KVH);
  switch (G) {
    case 1: hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, 1>), grid, dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, 2>), grid, dim3(128), 0, s, a); break;
    case 4: hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, 4>), grid, dim3(256), 0, s, a); break;
    default: return set_error(FMI_EINVAL, "attn: GQA ratio %d unsupported", G);
  }
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int launch_attn_prefill_mfma(const AttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.H % a.KVH == 0 && a.qtiles && a.n_qtiles > 0, "attn_prefill_mfma: bad arguments");
  switch (a.D) {
    case 32: return launch_attn_prefill_d<32>(a, s);
    case 64: return launch_attn_prefill_d<64>(a, s);
    case 128: return launch_attn_prefill_d<128>(a, s);
    default: return set_error(FMI_EINVAL, "attn: head_dim %d unsupported (32/64/128)", a.D);
  }
}

// Decode-time fusion of attn_prep + attn (one row per utterance, so a work-group only ever needs the
// K/V of its own (slot, kv-head) -- no cross-work-group dependency).  grid (B, KVH), 8 waves.
//   phase 1: k head (norm+RoPE -> cache + LDS), v head (-> cache + LDS), G q heads (norm+RoPE -> LDS)
//   phase 2: tokens [0, pos) stream from the paged cache, 4 tokens per wave-load, 4 loads in flight per
//            lane; token `pos` comes from LDS.  Online softmax per lane group, merged in-wave by
//            shuffles, across waves through LDS.
template <int D, int G>
__global__ __launch_bounds__(512) void attn_decode_fused_kernel(AttnArgs a) {
  constexpr int NW = 8, LPT = D / 8, TPW = 64 / LPT, UN = 4;
  __shared__ float s_q[G][D];
  __shared__ float s_k[D], s_v[D];
  __shared__ float s_m[NW][G], s_l[NW][G];
  __shared__ float s_acc[NW][G][D];
  const int r = blockIdx.x, kvh = blockIdx.y, gz = blockIdx.z;  // gz: which G of this kv head's Gt query heads
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = a.row_slot[r];
  const int pos = a.row_pos ? a.row_pos[r] : a.slot_pos[slot];
  if (a.long_thr > 0 && pos >= a.long_thr) return;   // this row is served by attn_decode_mfma_kernel + merge
  const int H = a.H, KVH = a.KVH, Gt = H / KVH;
  const int32_t* bt = a.block_table + (int64_t)slot * a.max_pages;
  const bf16_t* src = a.qkv + (int64_t)r * (H + 2 * KVH) * D;

  // ---- prefetch: the first trip of cached K/V rows does not depend on q, so it is issued before the
  // norm/RoPE phase and its HBM latency overlaps that phase
  const int sub = lane / LPT, dl = lane % LPT;
  const int n_groups = (pos + TPW - 1) / TPW;
  uint4 kv[UN], vv[UN];
  bool valid[UN];
  auto load_trip = [&](int g0) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = (g0 + u * NW) * TPW + sub;
      valid[u] = t < pos;
      const int tc = valid[u] ? t : 0;
      const int page = bt[tc / KV_PAGE];
      const int64_t base = (((int64_t)page * KVH + kvh) * KV_PAGE + (tc % KV_PAGE)) * D + dl * 8;
      kv[u] = *reinterpret_cast<const uint4*>(a.kpool + base);
      vv[u] = *reinterpret_cast<const uint4*>(a.vpool + base);
    }
  };
  if (wave < n_groups) load_trip(wave);

  // ---- phase 1
  for (int item = wave; item < G + 2; item += NW) {
    const int h = item == 0 ? H + kvh : (item == 1 ? H + KVH + kvh : kvh * Gt + gz * G + (item - 2));
    const bool act = lane < D / 2;
    const int p = act ? lane : 0;
    uint32_t raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    bf16_t o0 = (bf16_t)(raw & 0xffff), o1 = (bf16_t)(raw >> 16);
    if (item != 1) {
      const bf16_t* nw = (item == 0) ? a.knw : a.qnw;
      float y0 = x0, y1 = x1;
      if (nw) {
        float ss = wave_sum(x0 * x0 + x1 * x1);
        float rstd = rsqrtf(ss / (float)D + a.eps);
        y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(nw[2 * p])));
        y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(nw[2 * p + 1])));
      }
      uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
      float c = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));
      o0 = f2bf(__fsub_rn(__fmul_rn(y0, c), __fmul_rn(y1, sn)));
      o1 = f2bf(__fadd_rn(__fmul_rn(y1, c), __fmul_rn(y0, sn)));
    }
    if (act) {
      if (item >= 2) {
        s_q[item - 2][2 * p] = bf2f(o0);
        s_q[item - 2][2 * p + 1] = bf2f(o1);
      } else {
        // every split recomputes the new k/v row for its LDS copy; one of them appends it -- unless the slot has
        // finished (its position no longer advances and may sit one past the pages it reserved)
        if (gz == 0 && !(a.slot_done && a.slot_done[slot])) {
          const int page = bt[pos / KV_PAGE];
          bf16_t* pool = (item == 0) ? a.kpool : a.vpool;
          *reinterpret_cast<uint32_t*>(pool + (((int64_t)page * KVH + kvh) * KV_PAGE + pos % KV_PAGE) * D + 2 * p) =
              (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        float* dst = (item == 0) ? s_k : s_v;
        dst[2 * p] = bf2f(o0);
        dst[2 * p + 1] = bf2f(o1);
      }
    }
  }
  __syncthreads();

  // ---- phase 2
  const float scale = 1.0f / sqrtf((float)D);
  float q[G][8];
#pragma unroll
  for (int gq = 0; gq < G; ++gq)
#pragma unroll
    for (int j = 0; j < 8; ++j) q[gq][j] = s_q[gq][dl * 8 + j];
  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int gq = 0; gq < G; ++gq) {
    m[gq] = -1e30f;
    l[gq] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[gq][j] = 0.f;
  }

  auto consume = [&](const float (&kf)[8], const float (&vf)[8], bool valid) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += q[gq][j] * kf[j];
      d = group_allsum<LPT>(d);
      if (valid) {
        const float sc = d * scale;
        const float mn = fmaxf(m[gq], sc);
        const float corr = __expf(m[gq] - mn);
        const float pr = __expf(sc - mn);
        l[gq] = l[gq] * corr + pr;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[gq][j] = acc[gq][j] * corr + pr * vf[j];
        m[gq] = mn;
      }
    }
  };

  // cached tokens [0, pos): wave w takes token groups w, w+NW, ... of TPW tokens; UN groups per trip
  for (int g0 = wave; g0 < n_groups; g0 += NW * UN) {
    float kf[UN][8], vf[UN][8];
    bool vld[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bf16_t* ke = reinterpret_cast<const bf16_t*>(&kv[u]);
      const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vv[u]);
      vld[u] = valid[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kf[u][j] = bf2f(ke[j]);
        vf[u][j] = bf2f(ve[j]);
      }
    }
    if (g0 + NW * UN < n_groups) load_trip(g0 + NW * UN);  // next trip in flight during the maths
#pragma unroll
    for (int u = 0; u < UN; ++u) consume(kf[u], vf[u], vld[u]);
  }
  if (wave == 0) {  // the current token, straight from LDS (lane group 0 only)
    float kf[8], vf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      kf[j] = s_k[dl * 8 + j];
      vf[j] = s_v[dl * 8 + j];
    }
    consume(kf, vf, sub == 0);
  }

  // in-wave merge of the TPW lane-group states
#pragma unroll
  for (int off = LPT; off < 64; off <<= 1) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      const float mo = __shfl_xor(m[gq], off, 64), lo = __shfl_xor(l[gq], off, 64);
      const float mn = fmaxf(m[gq], mo);
      const float ws = __expf(m[gq] - mn), wo = __expf(mo - mn);
      l[gq] = l[gq] * ws + lo * wo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ao = __shfl_xor(acc[gq][j], off, 64);
        acc[gq][j] = acc[gq][j] * ws + ao * wo;
      }
      m[gq] = mn;
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
      if (dl == 0) {
        s_m[wave][gq] = m[gq];
        s_l[wave][gq] = l[gq];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s_acc[wave][gq][dl * 8 + j] = acc[gq][j];
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < G * D; o += 512) {
    const int gq = o / D, d = o % D;
    float M = -1e30f;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][gq]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wgt = __expf(s_m[w][gq] - M);
      L += s_l[w][gq] * wgt;
      O += s_acc[w][gq][d] * wgt;
    }
    a.out[((int64_t)r * H + kvh * Gt + gz * G + gq) * D + d] = f2bf(O / L);
  }
}

// =====================================================================================
// decode attention at LONG contexts on MFMA (round 3): rows whose position is >= AttnArgs::long_thr
// =====================================================================================
//
// Measured (profiles/r03_attn_decode*.txt): the VALU kernel above is latency-bound at the benchmark's context (~300 keys,
// 8.9 us) but bound by per-key VALU work at voice-clone lengths (32 us at 2 k keys = 2.1 TB/s; fewer, fatter
// work-groups and deeper prefetch both lose).  Here the per-key arithmetic is on the matrix cores:
//
//   attn_decode_mfma_kernel   grid (rows, KVH, ATTN_Z), 4 waves.  The work-group owns one utterance, one kv head --
//     all G query heads at once, as the first G of the 16 MFMA columns, so K/V are read once per kv head -- and the
//     z-th of ATTN_Z (= 8) equal ranges of 32-key blocks of the cached keys [0, pos); wave w takes blocks w, w + 4, ... of
//     that range.  Per block: S^T = K Q^T with the K rows straight from the paged cache as the A operand (a 16-byte
//     piece per lane IS a fragment: no LDS), fp32 online softmax per column, P stays in registers as the B operand of
//     O^T += V^T P (bf16 hi + lo, like the prefill kernel), V^T through a per-wave LDS transpose (the cache keeps V
//     key-major).  The four waves' (m, l, O) states are merged through LDS in wave order, the result goes to a
//     partials buffer [row][kvh][z][G][2 + D] fp32.
//   attn_decode_merge_kernel  grid (rows, KVH), 4 waves: q / k head norm + RoPE of the NEW token (as in the VALU
//     kernel's phase 1), K/V append, the new key's score per head, merge of the ATTN_Z partials and that key, output.
//
// Which kernel serves a row depends only on the row's own position (and ATTN_Z and the block ranges only on it too), so
// an utterance's numbers still do not depend on its batch-mates.  MATH-backend numerics like the prefill kernel: fp32
// scores, fp32 softmax, fp32 accumulation, one bf16 rounding of the output.
constexpr int ATTN_Z = 8;

template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(AttnArgs a) {
  constexpr int KB = 32, DC = D / 8, NW = 4;
  constexpr int VROWB = KB * 2 + 8;           // bytes per s_vt row (32 keys + 8 bytes pad)
  constexpr int VROWS = 8 * (DC + 1);         // permuted row index j*(DC+1) + dchunk
  constexpr int VCH = (KB / 2) * DC / 64;     // key-pair chunks per lane (D = 128: 4)
  static_assert(((KB / 2) * DC) % 64 == 0 && G <= NW, "V staging / one wave per query head in phase 1");
  __shared__ __attribute__((aligned(16))) unsigned char s_vt[NW][VROWS * VROWB];
  __shared__ __attribute__((aligned(16))) bf16_t s_q[G][D];
  __shared__ float s_m[NW][16], s_l[NW][16];
  __shared__ float s_o[NW][G][D];

  const int r = blockIdx.x, kvh = blockIdx.y, z = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int slot = a.row_slot[r];
  const int pos = a.row_pos ? a.row_pos[r] : a.slot_pos[slot];
  if (pos < a.long_thr) return;
  const int H = a.H, KVH = a.KVH, Gt = H / KVH;
  const int32_t* bt = a.block_table + (int64_t)slot * a.max_pages;
  const bf16_t* src = a.qkv + (int64_t)r * (H + 2 * KVH) * D;

  // ---- this work-group's block range of the cached keys [0, pos)
  const int nb = (pos + KB - 1) / KB;
  const int b_lo = (int)((int64_t)z * nb / ATTN_Z), b_hi = (int)((int64_t)(z + 1) * nb / ATTN_Z);
  f32x4 o[D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f;
  const float scale = 1.0f / sqrtf((float)D);
  unsigned char* vt = s_vt[wave];

  u32x4 kfr[2][D / 32];       // K fragments of a block: [key tile][k-step], straight from the cache
  uint4 vreg[VCH][2];
  auto fetch = [&](int kb) {
    const int page = bt[(kb * KB) / KV_PAGE];
    const int64_t pbase = ((int64_t)page * KVH + kvh) * KV_PAGE + (kb * KB) % KV_PAGE;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk)
        kfr[kt][kk] = *reinterpret_cast<const u32x4*>(a.kpool + (pbase + kt * 16 + c) * D + kk * 32 + g * 8);
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = lane + it * 64;
      const int kp = i / DC, dc = i % DC;
      // rows at or beyond `pos` have not been written by this utterance (stale pool contents): they must read as 0,
      // a masked probability of 0 times a stale NaN / Inf would poison the accumulator
      const bool v0 = kb * KB + 2 * kp < pos, v1 = kb * KB + 2 * kp + 1 < pos;
      vreg[it][0] = v0 ? *reinterpret_cast<const uint4*>(a.vpool + (pbase + 2 * kp) * D + dc * 8) : make_uint4(0, 0, 0, 0);
      vreg[it][1] = v1 ? *reinterpret_cast<const uint4*>(a.vpool + (pbase + 2 * kp + 1) * D + dc * 8) : make_uint4(0, 0, 0, 0);
    }
  };

  int kb = b_lo + wave;
  if (kb < b_hi) fetch(kb);
  // (the first block's rows are in flight during the query phase: they depend on nothing computed here)
  // ---- the query heads: norm + RoPE exactly as attn_decode_fused_kernel's phase 1, wave w -> head w
  if (wave < G) {
    const int h = kvh * Gt + wave;
    const bool act = lane < D / 2;
    const int p = act ? lane : 0;
    const uint32_t raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    const float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.qnw) {
      const float ss = wave_sum(x0 * x0 + x1 * x1);
      const float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(a.qnw[2 * p])));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(a.qnw[2 * p + 1])));
    }
    const uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
    const float cc = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));
    if (act) {
      s_q[wave][2 * p] = f2bf(__fsub_rn(__fmul_rn(y0, cc), __fmul_rn(y1, sn)));
      s_q[wave][2 * p + 1] = f2bf(__fadd_rn(__fmul_rn(y1, cc), __fmul_rn(y0, sn)));
    }
  }
  __syncthreads();
  // Q^T fragments (B operand): column c = query head c (columns >= G are zero), k rows = d chunk g of k-step kk
  bf16x8 qf[D / 32];
#pragma unroll
  for (int kk = 0; kk < D / 32; ++kk) {
    qf[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (c < G) qf[kk] = *reinterpret_cast<const bf16x8*>(&s_q[c][kk * 32 + g * 8]);
  }

  for (; kb < b_hi; kb += NW) {
    // ---- V block -> this wave's LDS image, transposed [d][32 keys] (row permutation as in the prefill kernel)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();            // the previous block's reads of `vt` are done (same wave)
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = lane + it * 64;
      const int kp = i / DC, dc = i % DC;
      const bf16_t* e0 = reinterpret_cast<const bf16_t*>(&vreg[it][0]);
      const bf16_t* e1 = reinterpret_cast<const bf16_t*>(&vreg[it][1]);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<uint32_t*>(&vt[(j * (DC + 1) + dc) * VROWB + kp * 4]) = (uint32_t)e0[j] | ((uint32_t)e1[j] << 16);
    }
    // ---- scores: two 16-key tiles, lane (c, g) ends up with keys kt*16 + g*4 + j of query column (head) c
    f32x4 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sacc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk)
        sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&kfr[kt][kk]), qf[kk], sacc[kt], 0, 0, 0);
    }
    const int kb_cur = kb;
    if (kb + NW < b_hi) fetch(kb + NW);         // next block's rows in flight during the softmax / PV maths
    float sc[8];
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kb_cur * KB + kt * 16 + g * 4 + j;
        const float v = key < pos ? sacc[kt][j] * scale : -1e30f;
        sc[kt * 4 + j] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float corr = __expf(m - mn);
    float ps = 0.f, pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pr[j] = sc[j] > -1e29f ? __expf(sc[j] - mn) : 0.f;
      ps += pr[j];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l = l * corr + ps;
    m = mn;
    bf16x8 ph, pl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16_t hi = f2bf(pr[j]);
      ph[j] = (short)hi;
      pl[j] = (short)f2bf(pr[j] - bf2f(hi));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();            // the transposed V image is complete (written by this wave's lanes)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      const int d = dt * 16 + c;
      const unsigned char* row = &vt[((d & 7) * (DC + 1) + (d >> 3)) * VROWB];
      const uint2 lo = *reinterpret_cast<const uint2*>(row + g * 8);        // keys g*4 .. g*4+3
      const uint2 hi = *reinterpret_cast<const uint2*>(row + 32 + g * 8);   // keys 16 + g*4 ..
      u32x4 av = {lo.x, lo.y, hi.x, hi.y};
      const bf16x8 vfrag = *reinterpret_cast<bf16x8*>(&av);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] *= corr;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, ph, o[dt], 0, 0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pl, o[dt], 0, 0, 0);
    }
  }

  // ---- merge the four waves (wave order), write the partial state of this key range
  if (g == 0) {
    s_m[wave][c] = m;
    s_l[wave][c] = l;
  }
  if (c < G) {
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) s_o[wave][c][dt * 16 + g * 4 + j] = o[dt][j];
  }
  __syncthreads();
  float* part = a.part + ((((int64_t)r * KVH + kvh) * ATTN_Z + z) * G) * (D + 2);
  for (int i = tid; i < G * D; i += 256) {
    const int gq = i / D, d = i % D;
    float M = -1e30f;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][gq]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wgt = __expf(s_m[w][gq] - M);
      L += s_l[w][gq] * wgt;
      O += s_o[w][gq][d] * wgt;
    }
    part[gq * (D + 2) + 2 + d] = O;
    if (d == 0) {
      part[gq * (D + 2)] = M;
      part[gq * (D + 2) + 1] = L;
    }
  }
}

template <int D, int G>
__global__ __launch_bounds__(256) void attn_decode_merge_kernel(AttnArgs a) {
  __shared__ float s_q[G][D];
  __shared__ float s_k[D], s_v[D];
  __shared__ float s_s[G];
  const int r = blockIdx.x, kvh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = a.row_slot[r];
  const int pos = a.row_pos ? a.row_pos[r] : a.slot_pos[slot];
  if (pos < a.long_thr) return;
  const int H = a.H, KVH = a.KVH, Gt = H / KVH;
  const int32_t* bt = a.block_table + (int64_t)slot * a.max_pages;
  const bf16_t* src = a.qkv + (int64_t)r * (H + 2 * KVH) * D;
  // ---- the new token: k head (norm + RoPE -> cache + LDS), v head (-> cache + LDS), G q heads (norm + RoPE -> LDS);
  // the same arithmetic as attn_decode_fused_kernel's phase 1
  for (int item = wave; item < G + 2; item += 4) {
    const int h = item == 0 ? H + kvh : (item == 1 ? H + KVH + kvh : kvh * Gt + (item - 2));
    const bool act = lane < D / 2;
    const int p = act ? lane : 0;
    const uint32_t raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    const float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    bf16_t o0 = (bf16_t)(raw & 0xffff), o1 = (bf16_t)(raw >> 16);
    if (item != 1) {
      const bf16_t* nw = (item == 0) ? a.knw : a.qnw;
      float y0 = x0, y1 = x1;
      if (nw) {
        const float ss = wave_sum(x0 * x0 + x1 * x1);
        const float rstd = rsqrtf(ss / (float)D + a.eps);
        y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(nw[2 * p])));
        y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(nw[2 * p + 1])));
      }
      const uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
      const float cc = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));
      o0 = f2bf(__fsub_rn(__fmul_rn(y0, cc), __fmul_rn(y1, sn)));
      o1 = f2bf(__fadd_rn(__fmul_rn(y1, cc), __fmul_rn(y0, sn)));
    }
    if (act) {
      if (item >= 2) {
        s_q[item - 2][2 * p] = bf2f(o0);
        s_q[item - 2][2 * p + 1] = bf2f(o1);
      } else {
        if (!(a.slot_done && a.slot_done[slot])) {   // a finished slot no longer appends (its position may sit past its pages)
          const int page = bt[pos / KV_PAGE];
          bf16_t* pool = (item == 0) ? a.kpool : a.vpool;
          *reinterpret_cast<uint32_t*>(pool + (((int64_t)page * KVH + kvh) * KV_PAGE + pos % KV_PAGE) * D + 2 * p) =
              (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        float* dst = (item == 0) ? s_k : s_v;
        dst[2 * p] = bf2f(o0);
        dst[2 * p + 1] = bf2f(o1);
      }
    }
  }
  __syncthreads();
  // ---- the new key's score per head
  const float scale = 1.0f / sqrtf((float)D);
  if (wave < G) {
    float d = 0.f;
    for (int i = lane; i < D; i += 64) d += s_q[wave][i] * s_k[i];
    d = wave_sum(d);
    if (lane == 0) s_s[wave] = d * scale;
  }
  __syncthreads();
  // ---- merge: ATTN_Z partial states over the cached keys (in z order) and the new key
  const float* part = a.part + (((int64_t)r * KVH + kvh) * ATTN_Z) * G * (D + 2);
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int gq = i / D, d = i % D;
    float M = s_s[gq];
#pragma unroll
    for (int zz = 0; zz < ATTN_Z; ++zz) M = fmaxf(M, part[(zz * G + gq) * (D + 2)]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int zz = 0; zz < ATTN_Z; ++zz) {
      const float* pz = part + (zz * G + gq) * (D + 2);
      const float wgt = __expf(pz[0] - M);
      L += pz[1] * wgt;
      O += pz[2 + d] * wgt;
    }
    const float wn = __expf(s_s[gq] - M);
    L += wn;
    O += wn * s_v[d];
    a.out[((int64_t)r * H + kvh * Gt + gq) * D + d] = f2bf(O / L);
  }
}

// The query heads of a kv head are split over `split` work-groups (each re-reads the K/V rows, which are L2
// hits): rows x KVH work-groups alone (64 at batch 8) leave three quarters of the CUs idle, and the per-head
// score/softmax/PV arithmetic is the serial part of this latency-bound kernel.  FMI_ATTN_SPLIT overrides.
template <int D>
static int launch_attn_decode_d(const AttnArgs& a, hipStream_t s) {
  const int Gt = a.H / a.KVH;
  static const int env_split = []() { const char* e = getenv("FMI_ATTN_SPLIT"); return e ? atoi(e) : 0; }();
  int split = env_split > 0 ? env_split : Gt;   // measured at batch 8, S2 shape: frame 5.31 / 5.18 / 5.07 ms for 1 / 2 / 4
  if (Gt % split != 0) split = 1;
  const int G = Gt / split;
  dim3 grid(a.rows, a.KVH, split), block(512);
  switch (G) {
    case 1: hipLaunchKernelGGL((attn_decode_fused_kernel<D, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((attn_decode_fused_kernel<D, 2>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((attn_decode_fused_kernel<D, 4>), grid, block, 0, s, a); break;
    default: return set_error(FMI_EINVAL, "attn: GQA ratio %d unsupported", Gt);
  }
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

int launch_attn_decode_fused(const AttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.H % a.KVH == 0, "attn: n_head %% n_local_heads");
  switch (a.D) {
    case 32: return launch_attn_decode_d<32>(a, s);
    case 64: return launch_attn_decode_d<64>(a, s);
    case 128: return launch_attn_decode_d<128>(a, s);
    default: return set_error(FMI_EINVAL, "attn: head_dim %d unsupported (32/64/128)", a.D);
  }
}

bool attn_decode_long_supported(int H, int KVH, int D) {
  const int G = KVH > 0 && H % KVH == 0 ? H / KVH : 0;
  return D == 128 && (G == 1 || G == 2 || G == 4);
}

int64_t attn_decode_long_part_floats(int rows, int H, int D) { return (int64_t)rows * H * ATTN_Z * (D + 2); }

int launch_attn_decode_long(const AttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(attn_decode_long_supported(a.H, a.KVH, a.D) && a.part && a.long_thr > 0, "attn_decode_long: unsupported shape");
  const int G = a.H / a.KVH;
  dim3 g1(a.rows, a.KVH, ATTN_Z), g2(a.rows, a.KVH);
#define FMI_LONG(G_)                                                                                  \
  do {                                                                                                \
    hipLaunchKernelGGL((attn_decode_mfma_kernel<128, G_>), g1, dim3(256), 0, s, a);                   \
    hipLaunchKernelGGL((attn_decode_merge_kernel<128, G_>), g2, dim3(256), 0, s, a);                  \
  } while (0)
  if (G == 1) FMI_LONG(1);
  else if (G == 2) FMI_LONG(2);
  else FMI_LONG(4);
#undef FMI_LONG
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// fast-AR attention (llama.py:948-976), S <= num_codebooks <= 16, everything rounded through bf16 like
// the reference's explicit matmul/softmax chain.  grid (B, KVH), 4 waves; a wave serves query heads
// g = wave, wave+4, ...  All keys are scored in parallel: lane = (key t, 32-dim chunk c), partial dots
// meet by a 4-lane DPP sum; the weighted sum of values runs with lanes along the head dimension.
__global__ __launch_bounds__(256) void fast_attn_kernel(FastAttnArgs a) {
  __shared__ float s_k[128], s_v[128];
  __shared__ float s_q[4][128];
  __shared__ float s_p[4][16];
  const int b = blockIdx.x, kvh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int D = a.D, H = a.H, KVH = a.KVH, G = H / KVH;
  const int slot = a.row_slot ? a.row_slot[b] : b;
  const int pos = a.pos;
  const bf16_t* src = a.qkv + (int64_t)b * (H + 2 * KVH) * D;
  bf16_t* kc = a.kc + (((int64_t)slot * KVH + kvh) * a.ncb) * D;
  bf16_t* vc = a.vc + (((int64_t)slot * KVH + kvh) * a.ncb) * D;
  const bool act = lane < D / 2;
  const int p = act ? lane : 0;
  uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
  const float c = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));

  // Requested up front (none of it depends on this step's projections): the cached key rows this lane scores,
  // the cached value elements it accumulates, and the first query head of this wave -- the kernel is a chain of
  // memory round trips otherwise.
  const int CH = D / 4;                      // dims per chunk lane (32 for D = 128)
  const int kt = lane >> 2, kcn = lane & 3;  // lane = (key, chunk)
  uint4 kpre[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4)
    if (kt < pos && j4 * 8 < CH) kpre[j4] = *reinterpret_cast<const uint4*>(kc + (int64_t)kt * D + kcn * CH + j4 * 8);
  uint32_t vpre[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < pos) vpre[t] = *reinterpret_cast<const uint32_t*>(vc + (int64_t)t * D + 2 * p);
  uint32_t qraw0 = (wave < G) ? *reinterpret_cast<const uint32_t*>(src + (kvh * G + wave) * D + 2 * p) : 0u;

  if (wave == 0) {  // key head: norm + rope -> cache + LDS
    uint32_t raw = *reinterpret_cast<const uint32_t*>(src + (H + kvh) * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.knw) {
      float ss = wave_sum_dpp(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(a.knw[2 * p])));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(a.knw[2 * p + 1])));
    }
    bf16_t o0 = f2bf(__fsub_rn(__fmul_rn(y0, c), __fmul_rn(y1, sn)));
    bf16_t o1 = f2bf(__fadd_rn(__fmul_rn(y1, c), __fmul_rn(y0, sn)));
    if (act) {
      *reinterpret_cast<uint32_t*>(kc + (int64_t)pos * D + 2 * p) = (uint32_t)o0 | ((uint32_t)o1 << 16);
      s_k[2 * p] = bf2f(o0);
      s_k[2 * p + 1] = bf2f(o1);
    }
  } else if (wave == 1) {  // value head
    if (act) {
      uint32_t raw = *reinterpret_cast<const uint32_t*>(src + (H + KVH + kvh) * D + 2 * p);
      *reinterpret_cast<uint32_t*>(vc + (int64_t)pos * D + 2 * p) = raw;
      s_v[2 * p] = bf2f((bf16_t)(raw & 0xffff));
      s_v[2 * p + 1] = bf2f((bf16_t)(raw >> 16));
    }
  }
  __syncthreads();

  const float scale = (float)(1.0 / sqrt((double)D));
  for (int gq = wave; gq < G; gq += 4) {
    const int h = kvh * G + gq;
    uint32_t raw = (gq == wave) ? qraw0 : *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.qnw) {
      float ss = wave_sum_dpp(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f(a.qnw[2 * p])));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f(a.qnw[2 * p + 1])));
    }
    if (act) {
      s_q[wave][2 * p] = rbf(__fsub_rn(__fmul_rn(y0, c), __fmul_rn(y1, sn)));
      s_q[wave][2 * p + 1] = rbf(__fadd_rn(__fmul_rn(y1, c), __fmul_rn(y0, sn)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // scores: lane (kt, kcn) takes dims [kcn*CH, +CH) of key kt; keys > pos are masked out
    float d = 0.f;
    if (kt <= pos) {
      if (kt == pos) {
        for (int j = 0; j < CH; ++j) d += s_q[wave][kcn * CH + j] * s_k[kcn * CH + j];
      } else {
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
          if (j4 * 8 < CH) {
            const bf16_t* ke = reinterpret_cast<const bf16_t*>(&kpre[j4]);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += s_q[wave][kcn * CH + j4 * 8 + e] * bf2f(ke[e]);
          }
      }
    }
    d = group_allsum<4>(d);
    // query @ key^T -> bf16, * scale -> bf16 (llama.py:971); masked keys -> -inf
    const float sc = (kt <= pos) ? rbf(rbf(d) * scale) : -INFINITY;
    if (kcn == 0) s_p[wave][kt] = sc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float mx = -INFINITY, e[16], sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, s_p[wave][t]);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      e[t] = (t <= pos) ? expf(s_p[wave][t] - mx) : 0.f;
      sum += e[t];
    }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t <= pos) {
        const float pr = rbf(e[t] / sum);  // softmax output rounded to bf16
        float v0, v1;
        if (t == pos) {
          v0 = s_v[2 * p];
          v1 = s_v[2 * p + 1];
        } else {
          const uint32_t vr = vpre[t];
          v0 = bf2f((bf16_t)(vr & 0xffff));
          v1 = bf2f((bf16_t)(vr >> 16));
        }
        o0 += pr * v0;
        o1 += pr * v1;
      }
    if (act)
      *reinterpret_cast<uint32_t*>(a.out + ((int64_t)b * H + h) * D + 2 * p) =
          (uint32_t)f2bf(o0) | ((uint32_t)f2bf(o1) << 16);
  }
}

int launch_fast_attn(const FastAttnArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.D <= 128 && a.D % 32 == 0 && a.ncb <= 16 && a.pos < a.ncb, "fast_attn: unsupported shape");
  hipLaunchKernelGGL(fast_attn_kernel, dim3(a.B, a.KVH), dim3(256), 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

// =====================================================================================
// sampler (inference.py:43-93, 118-144)
// =====================================================================================

__device__ inline uint32_t fmi_rand_u8(uint32_t seed, uint32_t stream, uint32_t frame, uint32_t draw, uint32_t i) {
  uint32_t x = seed * 0x9E3779B1u + stream * 0x85EBCA77u + frame * 0xC2B2AE3Du + draw * 0x27D4EB2Fu + i * 0x165667B1u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x >> 24;
}

__device__ inline uint32_t order_key(bf16_t v) {  // larger key <=> larger value
  return (v & 0x8000) ? (uint32_t)(~v & 0xffff) : (uint32_t)(v | 0x8000);
}

struct SamplerShared {
  uint32_t hist[256];
  uint32_t scan[256];
  int sel[4];             // b1, cnt_above, b2, ...
  float redf[8];
  int redi[8];
  int cand_idx[SAMPLER_MAXK];
  uint32_t cand_key[SAMPLER_MAXK];
  float s_val[SAMPLER_MAXK];   // sorted logits (fp32 of bf16)
  int s_idx[SAMPLER_MAXK];     // sorted row indices
  float s_p[SAMPLER_MAXK];     // softmax probs (bf16 values)
  float s_cum[SAMPLER_MAXK];   // cumulative (bf16 values)
  float s_e[SAMPLER_MAXK];
};

// suffix counts: scan[b] = sum_{j>=b} hist[j]  (256 threads)
__device__ inline void suffix_scan(SamplerShared& sh, int tid) {
  sh.scan[tid] = sh.hist[tid];
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    uint32_t v = (tid + o < 256) ? sh.scan[tid + o] : 0;
    __syncthreads();
    sh.scan[tid] += v;
    __syncthreads();
  }
}

__device__ inline float block_sum(SamplerShared& sh, float v, int tid) {
  v = wave_sum(v);
  if ((tid & 63) == 0) sh.redf[tid >> 6] = v;
  __syncthreads();
  float r = sh.redf[0] + sh.redf[1] + sh.redf[2] + sh.redf[3];
  __syncthreads();
  return r;
}

// One constrained draw from the prepared candidate list.  Returns the ROW index (or -1 when every
// value is 0 -- the reference's argmax then lands on vocabulary index 0).
__device__ int sampler_draw(SamplerShared& sh, int k, float temperature, float top_p, uint32_t seed, uint32_t stream,
                            uint32_t frame, uint32_t draw, const int32_t* ids, int tid) {
  const float tc = rbf(fmaxf(temperature, rbf(1e-5f)));
  // kept_r = r==0 || !(cum_r > top_p); tempered logits; exp against the rank-0 value
  float esum = 0.f;
  const float l0 = rbf(sh.s_val[0] / tc);
  for (int r = tid; r < k; r += 256) {
    const bool keep = (r == 0) || !(sh.s_cum[r] > top_p);
    float e = 0.f;
    if (keep) e = expf(rbf(sh.s_val[r] / tc) - l0);
    sh.s_e[r] = e;
    esum += e;
  }
  esum = block_sum(sh, esum, tid);
  float best = -1.f;
  int best_id = 0x7fffffff, best_row = -1;
  for (int r = tid; r < k; r += 256) {
    const float e = sh.s_e[r];
    if (e > 0.f) {
      const float pr = rbf(e / esum);
      const int row = sh.s_idx[r];
      const int vid = ids ? ids[row] : row;
      const uint32_t u8 = fmi_rand_u8(seed, stream, frame, draw, (uint32_t)vid);
      const float qv = -rbf(logf((float)u8 * (1.0f / 256.0f)));  // -log(u) in bf16; u=0 -> +inf
      const float val = rbf(pr / qv);
      if (val > best || (val == best && vid < best_id)) {
        best = val;
        best_id = vid;
        best_row = row;
      }
    }
  }
  // block arg-max with lowest-vocab-id tie break
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(best_id, o, 64);
    int orow = __shfl_xor(best_row, o, 64);
    if (ov > best || (ov == best && oi < best_id)) {
      best = ov;
      best_id = oi;
      best_row = orow;
    }
  }
  __shared__ float wb[4];
  __shared__ int wi[4], wr[4];
  if ((tid & 63) == 0) {
    wb[tid >> 6] = best;
    wi[tid >> 6] = best_id;
    wr[tid >> 6] = best_row;
  }
  __syncthreads();
  best = wb[0];
  best_id = wi[0];
  best_row = wr[0];
  for (int w = 1; w < 4; ++w)
    if (wb[w] > best || (wb[w] == best && wi[w] < best_id)) {
      best = wb[w];
      best_id = wi[w];
      best_row = wr[w];
    }
  __syncthreads();
  if (!(best > 0.f)) return -1;
  return best_row;
}

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SamplerShared& sh = *reinterpret_cast<SamplerShared*>(smem_raw);
  uint16_t* skey = reinterpret_cast<uint16_t*>(smem_raw + sizeof(SamplerShared));

  const int b = blockIdx.x, tid = threadIdx.x;
  const int slot = a.row_slot ? a.row_slot[b] : b;
  const bf16_t* lg = a.logits + (int64_t)b * a.ld;
  const int n = a.n;

  float temperature, top_p;
  int top_k;
  uint32_t seed;
  int frame, draw0, use_ras;
  if (a.mode == 2) {
    temperature = a.temperature; top_p = a.top_p; top_k = a.top_k; seed = a.seed;
    frame = a.frame; draw0 = a.draw; use_ras = a.prev != nullptr;
  } else {
    temperature = a.st.temperature[slot]; top_p = a.st.top_p[slot]; top_k = a.st.top_k[slot];
    seed = a.st.seed[slot]; frame = a.st.frame[slot];
    draw0 = (a.mode == 0) ? 0 : 1 + a.cb;
    use_ras = a.st.use_ras[slot] && frame > 0;
  }
  int k = top_k < n ? top_k : n;
  if (k > SAMPLER_MAXK) k = SAMPLER_MAXK;
  if (k < 1) k = 1;

  // --- pass 1: keys, max, high-byte histogram
  sh.hist[tid] = 0;
  __syncthreads();
  const int ept = (n + 255) / 256;
  const int i0 = tid * ept, i1 = min(n, i0 + ept);
  uint32_t kmax = 0;
  for (int i = i0; i < i1; ++i) {
    uint32_t key = order_key(lg[i]);
    skey[i] = (uint16_t)key;
    kmax = max(kmax, key);
    atomicAdd(&sh.hist[key >> 8], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
  if ((tid & 63) == 0) sh.redi[tid >> 6] = (int)kmax;
  __syncthreads();
  kmax = (uint32_t)max(max(sh.redi[0], sh.redi[1]), max(sh.redi[2], sh.redi[3]));
  const bf16_t maxbits = (kmax & 0x8000) ? (bf16_t)(kmax & 0x7fff) : (bf16_t)(~kmax & 0xffff);
  const float vmax = bf2f(maxbits);

  // --- softmax denominator over ALL entries (softmax of the un-tempered sorted logits)
  float se = 0.f;
  for (int i = tid; i < n; i += 256) se += expf(bf2f(lg[i]) - vmax);
  const float sumexp = block_sum(sh, se, tid);

  // --- radix select of the k-th largest key (two 8-bit levels)
  suffix_scan(sh, tid);
  {
    const uint32_t here = sh.scan[tid], above = (tid < 255) ? sh.scan[tid + 1] : 0;
    if (here >= (uint32_t)k && above < (uint32_t)k) {
      sh.sel[0] = tid;
      sh.sel[1] = (int)above;
    }
  }
  __syncthreads();
  const int b1 = sh.sel[0];
  const int above1 = sh.sel[1];
  sh.hist[tid] = 0;
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    uint32_t key = skey[i];
    if ((int)(key >> 8) == b1) atomicAdd(&sh.hist[key & 255], 1u);
  }
  __syncthreads();
  suffix_scan(sh, tid);
  {
    const uint32_t k2 = (uint32_t)(k - above1);
    const uint32_t here = sh.scan[tid], above = (tid < 255) ? sh.scan[tid + 1] : 0;
    if (here >= k2 && above < k2) {
      sh.sel[2] = tid;
      sh.sel[3] = (int)above;
    }
  }
  __syncthreads();
  const uint32_t thr = ((uint32_t)b1 << 8) | (uint32_t)sh.sel[2];
  const int c_gt = above1 + sh.sel[3];
  const int need_eq = k - c_gt;

  // --- collect candidates in index order: keys > thr, then the first need_eq keys == thr
  int my_gt = 0, my_eq = 0;
  for (int i = i0; i < i1; ++i) {
    uint32_t key = skey[i];
    my_gt += key > thr;
    my_eq += key == thr;
  }
  sh.hist[tid] = (uint32_t)my_gt;
  sh.scan[tid] = (uint32_t)my_eq;
  __syncthreads();
  // exclusive prefix sums over threads (thread chunks are contiguous index ranges)
  int off_gt = 0, off_eq = 0;
  for (int t = 0; t < tid; ++t) {
    off_gt += (int)sh.hist[t];
    off_eq += (int)sh.scan[t];
  }
  for (int i = i0; i < i1; ++i) {
    uint32_t key = skey[i];
    if (key > thr) {
      sh.cand_idx[off_gt] = i;
      sh.cand_key[off_gt] = key;
      ++off_gt;
    } else if (key == thr) {
      if (off_eq < need_eq) {
        sh.cand_idx[c_gt + off_eq] = i;
        sh.cand_key[c_gt + off_eq] = key;
      }
      ++off_eq;
    }
  }
  __syncthreads();
  // --- rank sort: (key desc, index asc); ties among equal logits -> ascending index
  for (int c = tid; c < k; c += 256) {
    const uint32_t kc = sh.cand_key[c];
    const int ic = sh.cand_idx[c];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const uint32_t kj = sh.cand_key[j];
      rank += (kj > kc) || (kj == kc && sh.cand_idx[j] < ic);
    }
    const bf16_t bits = (kc & 0x8000) ? (bf16_t)(kc & 0x7fff) : (bf16_t)(~kc & 0xffff);
    const float v = bf2f(bits);
    sh.s_val[rank] = v;
    sh.s_idx[rank] = ic;
    sh.s_p[rank] = rbf(expf(v - vmax) / sumexp);
  }
  __syncthreads();
  if (tid == 0) {  // torch.cumsum on bf16: fp32 running sum, each output rounded to bf16
    float c = 0.f;
    for (int r = 0; r < k; ++r) {
      c += sh.s_p[r];
      sh.s_cum[r] = rbf(c);
    }
  }
  __syncthreads();

  const int32_t* ids = a.ids;
  int row = sampler_draw(sh, k, temperature, top_p, seed, 0u /* stream: an utterance's draws depend on its seed only, not on the slot it occupies */, (uint32_t)frame, (uint32_t)draw0, ids, tid);
  int tok = (row < 0) ? 0 : (ids ? ids[row] : row);

  if (a.mode == 1) {  // fast codebook draw
    if (tid == 0) a.st.cur[(int64_t)slot * a.st.ncb1 + 1 + a.cb] = tok;
  } else {
    // second draw at RAS_HIGH_TEMP / RAS_HIGH_TOP_P (inference.py:126-131); always consumed
    const bool second = (a.mode == 0) || (a.prev != nullptr);
    if (second) {
      int row_h = sampler_draw(sh, k, 1.0f, rbf(0.9f), seed, 0u /* stream: an utterance's draws depend on its seed only, not on the slot it occupies */, (uint32_t)frame, (uint32_t)draw0 + 1, ids, tid);
      int tok_h = (row_h < 0) ? 0 : (ids ? ids[row_h] : row_h);
      if (use_ras) {
        const int32_t* win = (a.mode == 2) ? a.prev + (int64_t)b * RAS_WIN
                                           : a.st.window + (int64_t)slot * a.st.ncb1 * RAS_WIN;
        bool inwin = false;
        for (int j = 0; j < RAS_WIN; ++j) inwin |= (win[j] == tok);
        const bool sem = tok >= a.sem_begin && tok <= a.sem_end;
        if (inwin && sem) tok = tok_h;
      }
    }
    if (a.mode == 2) {
      if (tid == 0) a.out_tok[b] = tok;
      return;
    }
    int cb0 = tok - a.sem_begin;
    cb0 = cb0 < 0 ? 0 : (cb0 > a.cbs - 1 ? a.cbs - 1 : cb0);
    if (tid == 0) {
      a.st.cur[(int64_t)slot * a.st.ncb1 + 0] = tok;
      a.st.cur[(int64_t)slot * a.st.ncb1 + 1] = cb0;
    }
    tok = cb0;
  }
  // gather fast_embeddings[code] as the next fast step's input (inference.py:157,172)
  if (a.xf) {
    const bf16_t* src = a.fast_emb + (int64_t)tok * a.fdim;
    for (int c = tid * 8; c < a.fdim; c += 256 * 8)
      *reinterpret_cast<uint4*>(a.xf + (int64_t)b * a.fdim + c) = *reinterpret_cast<const uint4*>(src + c);
    if (a.qkv0_tab) {  // first fast layer's q|k|v of the drawn code (see SampleArgs)
      const bf16_t* q = a.qkv0_tab + (int64_t)tok * a.qkv0_dim;
      for (int c = tid * 8; c < a.qkv0_dim; c += 256 * 8)
        *reinterpret_cast<uint4*>(a.qkv0_out + (int64_t)b * a.qkv0_dim + c) = *reinterpret_cast<const uint4*>(q + c);
    }
  }
  // frame bookkeeping after the last codebook (decode_n_tokens, inference.py:224-233)
  if (a.mode == 1 && a.cb == a.st.ncb1 - 2) {
    __syncthreads();
    if (tid == 0) {
      const int ncb1 = a.st.ncb1;
      int32_t* cur = a.st.cur + (int64_t)slot * ncb1;
      cur[ncb1 - 1] = tok;
      if (!a.st.done[slot]) {
        const int f = a.st.frame[slot];
        if (f < a.st.max_frames) {
          int32_t* o = a.st.out + ((int64_t)slot * a.st.max_frames + f) * ncb1;
          for (int j = 0; j < ncb1; ++j) o[j] = cur[j];
        }
        if (f > 0) {  // the prefill frame is not inserted into the RAS window
          int32_t* win = a.st.window + (int64_t)slot * ncb1 * RAS_WIN;
          for (int j = 0; j < ncb1; ++j) {
            for (int w = 0; w < RAS_WIN - 1; ++w) win[j * RAS_WIN + w] = win[j * RAS_WIN + w + 1];
            win[j * RAS_WIN + RAS_WIN - 1] = cur[j];
          }
        }
        a.st.frame[slot] = f + 1;
        // the prefill step leaves pos at T (set by the host); decode steps advance by one
        if (f > 0) a.st.pos[slot] += 1;
        if (cur[0] == a.im_end) a.st.done[slot] = 1;
        else if (a.st.pos[slot] >= a.st.limit[slot] || f + 1 >= a.st.max_frames) a.st.done[slot] = 2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fast path, top_k <= 64 (the reference default is 30): same arithmetic as sample_kernel, but after
// the radix select a single wave finishes the job with shuffles (rank sort, sequential fp32 cumsum,
// both draws, bookkeeping) -- about ten barriers instead of fifty.
// ------------------------------------------------------------------------------------------------

struct SmallShared {
  uint32_t hist[256];
  float wsum[4];
  uint32_t wmax[4];
  int wcnt_gt[4], wcnt_eq[4];
  int sel[4];
  int cand_idx[64];
  uint32_t cand_key[64];
  __attribute__((aligned(16))) float s_val[64];
  int s_idx[64];
  uint32_t wc[4][64];   // per wave: its k candidates as key << 16 | ~index, then sorted descending
};

// suffix[b] = sum_{j >= b} hist[j] evaluated by wave 0; returns via sel[o], sel[o+1] the bin where the
// k-th largest key lives and the number of keys in bins above it
__device__ inline void wave_find_bin(SmallShared& sh, int lane, uint32_t k, int o) {
  uint32_t h[4], loc = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = sh.hist[lane * 4 + i];
    loc += h[i];
  }
  // inclusive suffix over lanes (lane l gets sum over lanes >= l)
  uint32_t suf = loc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t v = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += v;
  }
  uint32_t above = suf - loc;  // keys in bins of higher lanes
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    const uint32_t here = above + h[i];
    if (here >= k && above < k) {
      sh.sel[o] = lane * 4 + i;
      sh.sel[o + 1] = (int)above;
    }
    above = here;
  }
}

__device__ inline int small_draw(float v, float cum, int vid, int lane, int k, float temperature, float top_p,
                                 uint32_t seed, uint32_t stream, uint32_t frame, uint32_t draw) {
  const float tc = rbf(fmaxf(temperature, rbf(1e-5f)));
  const bool in = lane < k;
  const bool keep = in && ((lane == 0) || !(cum > top_p));
  const float lt = rbf(v / tc);
  const float l0 = __shfl(lt, 0, 64);
  const float e = keep ? expf(lt - l0) : 0.f;
  const float esum = wave_sum_dpp(e);
  float best = -1.f;
  int best_id = 0x7fffffff;
  if (e > 0.f) {
    const float pr = rbf(e / esum);
    const uint32_t u8 = fmi_rand_u8(seed, stream, frame, draw, (uint32_t)vid);
    const float qv = -rbf(logf((float)u8 * (1.0f / 256.0f)));
    best = rbf(pr / qv);
    best_id = vid;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_id, o, 64);
    if (ov > best || (ov == best && oi < best_id)) {
      best = ov;
      best_id = oi;
    }
  }
  return (best > 0.f) ? best_id : 0;  // all-zero race -> the reference's argmax lands on index 0
}

constexpr int SMALL_EPT = 17;  // keys per thread held in registers: n <= 256 * 17 = 4352

__device__ inline float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__global__ __launch_bounds__(256) void sample_small_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SmallShared& sh = *reinterpret_cast<SmallShared*>(smem_raw);
  uint16_t* skey = reinterpret_cast<uint16_t*>(smem_raw + sizeof(SmallShared));
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = a.row_slot ? a.row_slot[b] : b;
  const bf16_t* lg = a.logits + (int64_t)b * a.ld;
  const int n = a.n;

  float temperature, top_p;
  int top_k, frame, draw0, use_ras;
  uint32_t seed;
  if (a.mode == 2) {
    temperature = a.temperature; top_p = a.top_p; top_k = a.top_k; seed = a.seed;
    frame = a.frame; draw0 = a.draw; use_ras = a.prev != nullptr;
  } else {
    temperature = a.st.temperature[slot]; top_p = a.st.top_p[slot]; top_k = a.st.top_k[slot];
    seed = a.st.seed[slot]; frame = a.st.frame[slot];
    draw0 = (a.mode == 0) ? 0 : 1 + a.cb;
    use_ras = a.st.use_ras[slot] && frame > 0;
  }
  int k = top_k < n ? top_k : n;
  if (k > 64) k = 64;
  if (k < 1) k = 1;
  if (a.dbg_stop == 1) return;

  // ---- pass 1: coalesced loads (element tid + 256 j), keys to LDS, block max
  float xv[SMALL_EPT];
  uint32_t kmax = 0;
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    const int i = tid + 256 * j;
    const bf16_t raw = i < n ? lg[i] : (bf16_t)0xff80;  // -inf padding
    xv[j] = bf2f(raw);
    const uint32_t key = order_key(raw);
    if (i < n) {
      skey[i] = (uint16_t)key;
      kmax = max(kmax, key);
    }
  }
  kmax = wave_max_dpp_u(kmax);
  if (lane == 0) sh.wmax[wave] = kmax;
  __syncthreads();
  kmax = max(max(sh.wmax[0], sh.wmax[1]), max(sh.wmax[2], sh.wmax[3]));
  const bf16_t maxbits = (kmax & 0x8000) ? (bf16_t)(kmax & 0x7fff) : (bf16_t)(~kmax & 0xffff);
  const float vmax = bf2f(maxbits);
  if (a.dbg_stop == 2) return;
  // softmax denominator over ALL entries, same summation order as sample_kernel (thread-strided
  // partials j = 0.., xor tree per wave, waves summed 0..3)
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j)
    if (tid + 256 * j < n) se += expf(xv[j] - vmax);
  se = wave_sum_dpp(se);
  if (lane == 0) sh.wsum[wave] = se;
  // this thread's CONTIGUOUS chunk of keys into registers (index order matters for ties)
  const int ept = (n + 255) / 256;
  const int i0 = tid * ept;
  uint32_t kr[SMALL_EPT];
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) kr[j] = (j < ept && i0 + j < n) ? (uint32_t)skey[i0 + j] : 0u;
  // (key 0 never occurs for a real entry: order_key(x) >= 0x007f for -inf and above)
  __syncthreads();
  const float sumexp = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
  if (a.dbg_stop == 3) return;

  // ---- top-k without block-wide rounds.  Every wave picks the k largest of ITS keys by a radix-4 descent whose
  // counts meet inside the wave (DPP reductions, no LDS, no barrier), compacts them in index order (ties on the
  // k-th key: lowest indices first), sorts them with a 64-lane bitonic network on the 32-bit word
  // key << 16 | ~index (unique, and "larger word" == "larger logit, then lower index": the reference's stable
  // descending sort); wave 0 then merges the four sorted lists pairwise (max of one list against the reverse of the
  // other is bitonic and holds the 64 largest of both: six more stages sort it).  One barrier in total; lane r of
  // wave 0 ends up with the rank-r candidate.
  uint32_t thr = 0;
#pragma unroll 1
  for (int step = 0; step < 8; ++step) {
    const int sh_bits = 14 - 2 * step;
    const uint32_t c1 = thr | (1u << sh_bits), c2 = thr | (2u << sh_bits), c3 = thr | (3u << sh_bits);
    int n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
    for (int j = 0; j < SMALL_EPT; ++j) {
      n1 += kr[j] >= c1;
      n2 += kr[j] >= c2;
      n3 += kr[j] >= c3;
    }
    const int tp = wave_sum_dpp_i(n1 | (n2 << 16));   // each count <= 17 * 64 = 1088
    const int t3 = wave_sum_dpp_i(n3);
    const int t1 = tp & 0xffff, t2 = (int)((uint32_t)tp >> 16);
    if (t3 >= k) thr = c3;
    else if (t2 >= k) thr = c2;
    else if (t1 >= k) thr = c1;
  }
  int my_gt = 0, my_eq = 0;
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    my_gt += kr[j] > thr;
    my_eq += (kr[j] == thr) && (thr != 0);
  }
  if (a.dbg_stop == 4) return;
  int inc_gt = my_gt, inc_eq = my_eq;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int g = __shfl_up(inc_gt, off, 64), e = __shfl_up(inc_eq, off, 64);
    if (lane >= off) {
      inc_gt += g;
      inc_eq += e;
    }
  }
  const int c_gt = __builtin_amdgcn_readlane(inc_gt, 63);     // this wave's keys above its threshold (< k)
  const int need_eq = k - c_gt;
  int off_gt = inc_gt - my_gt, off_eq = inc_eq - my_eq;
  uint32_t* wc = sh.wc[wave];
  wc[lane] = 0;                                                 // word 0 = empty place (real words have key >= 0x7f)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    const uint32_t key = kr[j];
    const uint32_t word = (key << 16) | (uint32_t)(0xffff - (i0 + j));
    if (key > thr) {
      wc[off_gt++] = word;
    } else if (key == thr && thr != 0) {
      if (off_eq < need_eq) wc[c_gt + off_eq] = word;
      ++off_eq;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  uint32_t w = wc[lane];
  // bitonic sort, descending over the 64 lanes
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)w, stride, 64);
      const bool take_max = ((lane & stride) == 0) == ((lane & size) == 0);
      w = take_max ? max(w, o) : min(w, o);
    }
  wc[lane] = w;
  __syncthreads();
  // the rows gathered for the next fast step (embedding, tabulated layer-0 q|k|v) are fetched by ALL waves once wave 0
  // knows the code: 17 dependent load -> store trips of one wave (5 + 12 KB at the S2 shape) were ~10 us of this kernel
  const bool gather_all = a.xf != nullptr && a.dbg_stop == 0 && a.mode != 2;
  int tok = 0;
  if (wave == 0 && a.dbg_stop != 5) {
  auto merge_desc = [&](uint32_t x, uint32_t y_rev) -> uint32_t {   // x sorted desc, y_rev = other list reversed
    uint32_t m = max(x, y_rev);
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)m, stride, 64);
      m = ((lane & stride) == 0) ? max(m, o) : min(m, o);
    }
    return m;
  };
  const uint32_t m01 = merge_desc(w, sh.wc[1][63 - lane]);
  const uint32_t m23 = merge_desc(sh.wc[2][lane], sh.wc[3][63 - lane]);
  const uint32_t m23_rev = (uint32_t)__shfl((int)m23, 63 - lane, 64);
  const uint32_t top = merge_desc(m01, m23_rev);

  // ---- wave 0: lane r holds the rank-r candidate; sequential cumsum evaluated by every lane
  const bool in = lane < k;
  const uint32_t ks = top >> 16;
  const bf16_t vbits = (ks & 0x8000) ? (bf16_t)(ks & 0x7fff) : (bf16_t)(~ks & 0xffff);
  const float v = in ? bf2f(vbits) : -INFINITY;
  const int row = in ? (int)(0xffff - (top & 0xffff)) : 0;
  const int vid = a.ids ? a.ids[row] : row;
  const float p = in ? rbf(expf(v - vmax) / sumexp) : 0.f;
  sh.s_val[lane] = p;                    // lanes >= k hold 0
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float run = 0.f, cum = 0.f;  // torch.cumsum on bf16: fp32 running sum in rank order, outputs rounded
#pragma unroll
  for (int i4 = 0; i4 < 16; ++i4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(&sh.s_val[i4 * 4]);   // same address in every lane: a broadcast
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      run += q[e];
      cum = (lane == i4 * 4 + e) ? rbf(run) : cum;
    }
  }

  if (a.dbg_stop == 6) return;
  tok = small_draw(v, cum, vid, lane, k, temperature, top_p, seed, 0u /* stream: an utterance's draws depend on its seed only, not on the slot it occupies */, (uint32_t)frame, (uint32_t)draw0);
  if (a.dbg_stop == 7) return;
  if (a.mode == 1) {
    if (lane == 0) a.st.cur[(int64_t)slot * a.st.ncb1 + 1 + a.cb] = tok;
  } else {
    const bool second = (a.mode == 0) || (a.prev != nullptr);
    if (second) {
      const int tok_h = small_draw(v, cum, vid, lane, k, 1.0f, rbf(0.9f), seed, 0u /* stream: an utterance's draws depend on its seed only, not on the slot it occupies */, (uint32_t)frame,
                                   (uint32_t)draw0 + 1);
      if (use_ras) {
        const int32_t* win = (a.mode == 2) ? a.prev + (int64_t)b * RAS_WIN
                                           : a.st.window + (int64_t)slot * a.st.ncb1 * RAS_WIN;
        bool inwin = false;
        for (int j = 0; j < RAS_WIN; ++j) inwin |= (win[j] == tok);
        if (inwin && tok >= a.sem_begin && tok <= a.sem_end) tok = tok_h;
      }
    }
    if (a.mode == 2) {
      if (lane == 0) a.out_tok[b] = tok;
      return;
    }
    int cb0 = tok - a.sem_begin;
    cb0 = cb0 < 0 ? 0 : (cb0 > a.cbs - 1 ? a.cbs - 1 : cb0);
    if (lane == 0) {
      a.st.cur[(int64_t)slot * a.st.ncb1 + 0] = tok;
      a.st.cur[(int64_t)slot * a.st.ncb1 + 1] = cb0;
    }
    tok = cb0;
  }
  if (lane == 0) sh.sel[0] = tok;
  } else if (!gather_all) {
    return;
  }
  if (a.xf) {  // fast_embeddings[code] -> next fast step's input (inference.py:157,172)
    if (gather_all) {
      __syncthreads();
      tok = sh.sel[0];
    } else if (wave != 0) {
      return;
    }
    // ... and the first fast layer's q|k|v of that code (precomputed with the very same GEMV); every load of a
    // thread is requested before its first store
    const int n1 = a.fdim >> 3, n2 = a.qkv0_tab ? (a.qkv0_dim >> 3) : 0;
    const uint4* src1 = reinterpret_cast<const uint4*>(a.fast_emb + (int64_t)tok * a.fdim);
    const uint4* src2 = a.qkv0_tab ? reinterpret_cast<const uint4*>(a.qkv0_tab + (int64_t)tok * a.qkv0_dim) : nullptr;
    uint4* dst1 = reinterpret_cast<uint4*>(a.xf + (int64_t)b * a.fdim);
    uint4* dst2 = a.qkv0_tab ? reinterpret_cast<uint4*>(a.qkv0_out + (int64_t)b * a.qkv0_dim) : nullptr;
    const int nthr = gather_all ? 256 : 64, t0 = gather_all ? tid : lane;
    constexpr int GB = 6;
    for (int base = 0; base < n1 + n2; base += GB * nthr) {
      uint4 gv[GB];
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        const int i = base + t0 + j * nthr;
        if (i < n1) gv[j] = src1[i];
        else if (i < n1 + n2) gv[j] = src2[i - n1];
      }
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        const int i = base + t0 + j * nthr;
        if (i < n1) dst1[i] = gv[j];
        else if (i < n1 + n2) dst2[i - n1] = gv[j];
      }
    }
  }
  if (wave != 0) return;
  if (a.mode == 1 && a.cb == a.st.ncb1 - 2 && lane == 0) {  // frame bookkeeping, as in sample_kernel
    const int ncb1 = a.st.ncb1;
    int32_t* cur = a.st.cur + (int64_t)slot * ncb1;
    cur[ncb1 - 1] = tok;
    if (!a.st.done[slot]) {
      const int f = a.st.frame[slot];
      if (f < a.st.max_frames) {
        int32_t* o = a.st.out + ((int64_t)slot * a.st.max_frames + f) * ncb1;
        for (int j = 0; j < ncb1; ++j) o[j] = cur[j];
      }
      if (f > 0) {
        int32_t* win = a.st.window + (int64_t)slot * ncb1 * RAS_WIN;
        for (int j = 0; j < ncb1; ++j) {
          for (int w = 0; w < RAS_WIN - 1; ++w) win[j * RAS_WIN + w] = win[j * RAS_WIN + w + 1];
          win[j * RAS_WIN + RAS_WIN - 1] = cur[j];
        }
      }
      a.st.frame[slot] = f + 1;
      if (f > 0) a.st.pos[slot] += 1;
      if (cur[0] == a.im_end) a.st.done[slot] = 1;
      else if (a.st.pos[slot] >= a.st.limit[slot] || f + 1 >= a.st.max_frames) a.st.done[slot] = 2;
    }
  }
}

int launch_sample(const SampleArgs& a, hipStream_t s) {
  FMI_REQUIRE(a.n >= 1 && a.n <= 65536, "sample: n=%d out of range", a.n);
  if (a.small_k && a.n <= 256 * SMALL_EPT) {  // every slot draws with top_k <= 64, keys fit in registers
    size_t smem = sizeof(SmallShared) + (size_t)a.n * 2 + 16;
    hipLaunchKernelGGL(sample_small_kernel, dim3(a.B), dim3(256), smem, s, a);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  size_t smem = sizeof(SamplerShared) + (size_t)a.n * 2 + 16;
  hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(256), smem, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

}  // namespace fmi
