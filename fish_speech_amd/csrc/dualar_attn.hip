// dualar_attn.hip -- attention kernels of the Dual-AR path (slow prefill / decode, fast); split out of
// dualar_kernels.hip so that the translation units compile in parallel.
#include "dualar_kernels.h"
#include "dualar_dev.h"

namespace fmi {
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
  // the three operands of a head -- its slice of the q|k|v row, the norm weight pair, the RoPE pair -- are requested
  // together (none of the addresses depends on another's data): they had been three dependent round trips per wave
  const bf16_t* nw = (h < H) ? a.qnw : (h < H + KVH ? a.knw : nullptr);
  uint32_t raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
  uint32_t nwp = 0, cs = 0;
  if (h < H + KVH) {
    if (nw) nwp = *reinterpret_cast<const uint32_t*>(nw + 2 * p);
    cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
  }
  float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
  bf16_t o0 = (bf16_t)(raw & 0xffff), o1 = (bf16_t)(raw >> 16);
  if (h < H + KVH) {
    float y0 = x0, y1 = x1;
    if (nw) {  // torch.nn.RMSNorm: fp32 normalise * weight, one cast (llama.py:862-864)
      float ss = wave_sum(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f((bf16_t)(nwp & 0xffff))));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f((bf16_t)(nwp >> 16))));
    }
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
// MQ (round 4) = 16-row query tiles per work-group (tile descriptors then cover up to 16 MQ rows): the staged K/V block and
// its fragment reads are shared by MQ column groups -- with one group a 32-key block cost two barriers and 16 KiB of
// staging for 24 products per wave (1.18 ms per layer at 8 x 2048 tokens: 0.09 of the matrix peak).  A query row's
// arithmetic does not depend on the group it sits in (key blocks past a group's own last position are skipped, exactly
// as its own 16-row work-group would), so the output bits do not depend on MQ.  The bf16 hi + lo split of the
// probabilities and the output rounding use v_cvt_pk_bf16_f32 (round-to-nearest-even like f2bf, 2 values per
// instruction instead of 7 integer operations per value).
__device__ inline uint32_t attn_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <int D, int G, int MQ>
__global__ __launch_bounds__(64 * G, MQ > 1 ? 2 : 1) void attn_prefill_mfma_kernel(AttnArgs a) {
  constexpr int KB = 32, NT = 64 * G, DC = D / 8;
  constexpr int KROW = D + 8;                 // bf16 elements per s_k row (+16 bytes)
  constexpr int VROWB = KB * 2 + 8;           // bytes per s_vt row (32 keys + 8 bytes pad)
  constexpr int VROWS = 8 * (DC + 1);         // permuted row index j*(DC+1) + dchunk
  constexpr int KCH = KB * DC / NT > 0 ? KB * DC / NT : 1;        // 16-byte K chunks per thread
  constexpr int VCH = (KB / 2) * DC / NT > 0 ? (KB / 2) * DC / NT : 1;  // key-pair chunks per thread
  static_assert((KB * DC) % NT == 0 || KB * DC < NT, "K staging");
  constexpr bool KFULL = (KB * DC) % NT == 0, VFULL = ((KB / 2) * DC) % NT == 0;   // every thread has work in every staging pass
  __shared__ __attribute__((aligned(16))) bf16_t s_k[KB * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char s_vt[VROWS * VROWB];

  const int4 td = a.qtiles[blockIdx.x];  // x row0, y rows (1..16 MQ), z slot, w first position
  const int kvh = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int H = a.H, KVH = a.KVH;
  const int head = kvh * G + wave;
  const int32_t* bt = a.block_table + (int64_t)td.z * a.max_pages;
  const int last_pos = td.w + td.y - 1;
  const int n_blocks = last_pos / KB + 1;

  // Q^T fragments (B operand: column = query 16 q + c, k rows = d chunk g of k-step kk)
  bf16x8 qf[MQ][D / 32];
#pragma unroll
  for (int q = 0; q < MQ; ++q) {
    const int qrow = td.x + min(16 * q + c, td.y - 1);
    const bf16_t* qp = a.q + ((int64_t)qrow * H + head) * D + g * 8;
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk) qf[q][kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32);
  }

  f32x4 o[MQ][D / 16];
  float m[MQ], l[MQ];
#pragma unroll
  for (int q = 0; q < MQ; ++q) {
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) o[q][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[q] = -1e30f;
    l[q] = 0.f;
  }
  const float scale = 1.0f / sqrtf((float)D);

  // ---- staging helpers: K chunk i -> (key i / DC, dchunk i % DC); V chunk i -> (key pair i / DC, dchunk i % DC)
  // (vector types, not HIP's uint4 struct: copied as a struct the K registers were kept in SCRATCH -- stored right behind
  // their global load and re-loaded for the LDS write, which also made the 'prefetch' wait for the load at once)
  u32x4 kreg[KCH], vreg[VCH][2];
  auto fetch = [&](int kb) {
    const int page = bt[(kb * KB) / KV_PAGE];
    const int64_t pbase = ((int64_t)page * KVH + kvh) * KV_PAGE + (kb * KB) % KV_PAGE;
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
      const int i = tid + it * NT;
      if (KFULL || i < KB * DC) {
        const int key = i / DC, dc = i % DC;
        kreg[it] = *reinterpret_cast<const u32x4*>(a.kpool + (pbase + key) * D + dc * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = tid + it * NT;
      if (VFULL || i < (KB / 2) * DC) {
        const int kp = i / DC, dc = i % DC;
        const bool v0 = kb * KB + 2 * kp <= last_pos, v1 = kb * KB + 2 * kp + 1 <= last_pos;
        // rows beyond the tile's last position have not been written (stale pool contents): they must read as 0,
        // a masked probability of 0 times a stale NaN/Inf would poison the accumulator
        vreg[it][0] = v0 ? *reinterpret_cast<const u32x4*>(a.vpool + (pbase + 2 * kp) * D + dc * 8) : (u32x4){0, 0, 0, 0};
        vreg[it][1] = v1 ? *reinterpret_cast<const u32x4*>(a.vpool + (pbase + 2 * kp + 1) * D + dc * 8) : (u32x4){0, 0, 0, 0};
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < KCH; ++it) {
      const int i = tid + it * NT;
      if (KFULL || i < KB * DC) *reinterpret_cast<u32x4*>(&s_k[(i / DC) * KROW + (i % DC) * 8]) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VCH; ++it) {
      const int i = tid + it * NT;
      if (VFULL || i < (KB / 2) * DC) {
        const int kp = i / DC, dc = i % DC;
        // element j of the two keys side by side: one v_perm_b32 per dword (no element-wise register addressing)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<uint32_t*>(&s_vt[(j * (DC + 1) + dc) * VROWB + kp * 4]) =
              __builtin_amdgcn_perm(vreg[it][1][j >> 1], vreg[it][0][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
      }
    }
  };

  fetch(0);
  for (int kb = 0; kb < n_blocks; ++kb) {
    __syncthreads();          // every wave is done reading the previous block's tiles
    stage();
    __syncthreads();
    if (kb + 1 < n_blocks) fetch(kb + 1);

    // ---- scores: two 16-key tiles per column group, lane (c, g) ends up with keys kt*16 + g*4 + j of query column c.
    // (A group whose rows all lie before this key block is done: its own 16-row work-group would have stopped.)
    bool act[MQ];
    f32x4 sacc[MQ][2];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      act[q] = 16 * q < td.y && kb * KB <= td.w + min(16 * q + 15, td.y - 1);
      sacc[q][0] = sacc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk) {
        const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(&s_k[(kt * 16 + c) * KROW + kk * 32 + g * 8]);
#pragma unroll
        for (int q = 0; q < MQ; ++q)
          if (act[q]) sacc[q][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[q][kk], sacc[q][kt], 0, 0, 0);
      }
    u32x4 ph[MQ], pl[MQ];      // probabilities as the B operand (k slots g*8 + jj = keys {g*4 + jj | jj < 4} and {16 + g*4 + jj - 4}), hi + lo
    float corr[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      corr[q] = 1.f;
      if (!act[q]) continue;
      const int qpos = td.w + 16 * q + c;
      float sc[8];
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = kb * KB + kt * 16 + g * 4 + j;
          const float v = key <= qpos ? sacc[q][kt][j] * scale : -1e30f;
          sc[kt * 4 + j] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[q], mx);
      corr[q] = __expf(m[q] - mn);
      float ps = 0.f, pr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pr[j] = sc[j] > -1e29f ? __expf(sc[j] - mn) : 0.f;
        ps += pr[j];
      }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l[q] = l[q] * corr[q] + ps;
      m[q] = mn;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t hi = attn_pk_bf16(pr[2 * j], pr[2 * j + 1]);
        ph[q][j] = hi;
        pl[q][j] = attn_pk_bf16(pr[2 * j] - __uint_as_float(hi << 16), pr[2 * j + 1] - __uint_as_float(hi & 0xffff0000u));
      }
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
      for (int q = 0; q < MQ; ++q) {
        if (!act[q]) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[q][dt][j] *= corr[q];
        o[q][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, *reinterpret_cast<bf16x8*>(&ph[q]), o[q][dt], 0, 0, 0);
        o[q][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, *reinterpret_cast<bf16x8*>(&pl[q]), o[q][dt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int q = 0; q < MQ; ++q) {
    if (16 * q + c < td.y) {   // lane holds query column 16 q + c, output dims dt*16 + g*4 + j
      const float inv = 1.0f / l[q];
      bf16_t* op = a.out + ((int64_t)(td.x + 16 * q + c) * H + head) * D + g * 4;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt)
        *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(attn_pk_bf16(o[q][dt][0] * inv, o[q][dt][1] * inv),
                                                            attn_pk_bf16(o[q][dt][2] * inv, o[q][dt][3] * inv));
    }
  }
}

template <int D>
static int launch_attn_prefill_d(const AttnArgs& a, hipStream_t s) {
  const int G = a.H / a.KVH;
  dim3 grid(a.n_qtiles, a.KVH);
  FMI_REQUIRE(a.qtile_rows == 16 || a.qtile_rows == 32 || a.qtile_rows == 48, "attn: query tiles of 16 / 32 / 48 rows");
#define FMI_PREFILL(G_)                                                                                                     \
  do {                                                                                                                      \
    if (a.qtile_rows == 16) hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, G_, 1>), grid, dim3(64 * G_), 0, s, a);         \
    else if (a.qtile_rows == 32) hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, G_, 2>), grid, dim3(64 * G_), 0, s, a);    \
    else hipLaunchKernelGGL((attn_prefill_mfma_kernel<D, G_, 3>), grid, dim3(64 * G_), 0, s, a);                            \
  } while (0)
  switch (G) {
    case 1: FMI_PREFILL(1); break;
    case 2: FMI_PREFILL(2); break;
    case 4: FMI_PREFILL(4); break;
    default: return set_error(FMI_EINVAL, "attn: GQA ratio %d unsupported", G);
  }
#undef FMI_PREFILL
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
template <int D, int G, bool BTR>
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
  // the slot's block-table row (<= 64 pages = 4096 positions) in one register per wave: a token's page then comes from a
  // lane of it (ds_bpermute) instead of a global load in front of every K/V trip -- one dependent L2 round trip less
  // per trip.  Longer tables fall back to the loads.
  // (BTR, chosen by the launcher: a run-time select between the two kept both, with a wait behind every load)
  int btv = 0;
  if constexpr (BTR) btv = bt[min(lane, a.max_pages - 1)];
  auto page_of = [&](int pg) -> int {
    if constexpr (BTR) return __shfl(btv, pg, 64);
    else return bt[pg];
  };

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
      const int page = page_of(tc / KV_PAGE);
      const int64_t base = (((int64_t)page * KVH + kvh) * KV_PAGE + (tc % KV_PAGE)) * D + dl * 8;
      kv[u] = *reinterpret_cast<const uint4*>(a.kpool + base);
      vv[u] = *reinterpret_cast<const uint4*>(a.vpool + base);
    }
  };
  // The operands of phase 1 (this wave's head of the q|k|v row, its norm weights, the RoPE pair: L2 hits) are requested
  // BEFORE the cached K/V rows (HBM) -- loads return in order, so requested after them, as rounds 1-3 did, the
  // norm / RoPE phase could not start before the first K/V trip had landed, and its three small loads were three
  // dependent round trips of their own.  Now the K/V trip flies under phase 1, as it was meant to.
  uint32_t p1_raw = 0, p1_nw = 0, p1_cs = 0;
  {
    const int item = wave;
    if (item < G + 2) {
      const int h = item == 0 ? H + kvh : (item == 1 ? H + KVH + kvh : kvh * Gt + gz * G + (item - 2));
      const int p = lane < D / 2 ? lane : 0;
      p1_raw = *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
      if (item != 1) {
        const bf16_t* nw = (item == 0) ? a.knw : a.qnw;
        if (nw) p1_nw = *reinterpret_cast<const uint32_t*>(nw + 2 * p);
        p1_cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
      }
    }
  }
  // (unconditional: a wave without a first group re-reads token 0's rows and ignores them -- behind a branch the
  // compiler cannot count the outstanding loads at the join and falls back to s_waitcnt vmcnt(0) before phase 1)
  load_trip(wave);

  // ---- phase 1
  for (int item = wave; item < G + 2; item += NW) {
    const int h = item == 0 ? H + kvh : (item == 1 ? H + KVH + kvh : kvh * Gt + gz * G + (item - 2));
    const bool act = lane < D / 2;
    const int p = act ? lane : 0;
    const bool first = item == wave;   // (the first -- for G <= 6 the only -- item of a wave was requested above)
    uint32_t raw = first ? p1_raw : *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    bf16_t o0 = (bf16_t)(raw & 0xffff), o1 = (bf16_t)(raw >> 16);
    if (item != 1) {
      const bf16_t* nw = (item == 0) ? a.knw : a.qnw;
      float y0 = x0, y1 = x1;
      if (nw) {
        float ss = wave_sum(x0 * x0 + x1 * x1);
        float rstd = rsqrtf(ss / (float)D + a.eps);
        const uint32_t nwp = first ? p1_nw : *reinterpret_cast<const uint32_t*>(nw + 2 * p);
        y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), bf2f((bf16_t)(nwp & 0xffff))));
        y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), bf2f((bf16_t)(nwp >> 16))));
      }
      uint32_t cs = first ? p1_cs : *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
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
          const int page = page_of(pos / KV_PAGE);
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
#define FMI_DECODE(G_)                                                                                     \
  do {                                                                                                      \
    if (a.max_pages <= 64) hipLaunchKernelGGL((attn_decode_fused_kernel<D, G_, true>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((attn_decode_fused_kernel<D, G_, false>), grid, block, 0, s, a);                \
  } while (0)
  switch (G) {
    case 1: FMI_DECODE(1); break;
    case 2: FMI_DECODE(2); break;
    case 4: FMI_DECODE(4); break;
    default: return set_error(FMI_EINVAL, "attn: GQA ratio %d unsupported", Gt);
  }
#undef FMI_DECODE
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
  __shared__ __attribute__((aligned(16))) bf16_t s_k0b[128], s_v0b[128];
  // a.merge (positions 0 and 1 of a frame in ONE pass, dualar.hip: tail): row b < B is utterance b at position 0, row
  // B + u utterance u at position 1.  The position-1 work-group cannot read key / value 0 from the cache (the
  // position-0 work-group of this very launch writes them): waves 2 / 3 rebuild them from row u's k / v heads -- the same
  // arithmetic, so the same bits -- and hand them over through LDS in the cache's own bf16 form.
  const int b = blockIdx.x, kvh = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int D = a.D, H = a.H, KVH = a.KVH, G = H / KVH;
  const int utt = a.merge ? b % a.B : b;
  const int slot = a.row_slot ? a.row_slot[utt] : utt;
  const int pos = a.merge ? b / a.B : a.pos;
  const bool rebuild0 = a.merge && pos == 1;
  const bf16_t* src = a.qkv + (int64_t)b * (H + 2 * KVH) * D;
  const bf16_t* src0 = a.qkv + (int64_t)utt * (H + 2 * KVH) * D;   // rebuild0: the utterance's position-0 row
  bf16_t* kc = a.kc + (((int64_t)slot * KVH + kvh) * a.ncb) * D;
  bf16_t* vc = a.vc + (((int64_t)slot * KVH + kvh) * a.ncb) * D;
  const bool act = lane < D / 2;
  const int p = act ? lane : 0;
  uint32_t cs = *reinterpret_cast<const uint32_t*>(a.rope + ((int64_t)pos * (D / 2) + p) * 2);
  const float c = bf2f((bf16_t)(cs & 0xffff)), sn = bf2f((bf16_t)(cs >> 16));
  // the q / k head-norm weight pairs of this lane, requested here with everything else that does not depend on the
  // step's projections (loaded where they are used they were one more dependent L2 round trip per head)
  const uint32_t knw_p = a.knw ? *reinterpret_cast<const uint32_t*>(a.knw + 2 * p) : 0u;
  const uint32_t qnw_p = a.qnw ? *reinterpret_cast<const uint32_t*>(a.qnw + 2 * p) : 0u;
  const float knw0 = bf2f((bf16_t)(knw_p & 0xffff)), knw1 = bf2f((bf16_t)(knw_p >> 16));
  const float qnw0 = bf2f((bf16_t)(qnw_p & 0xffff)), qnw1 = bf2f((bf16_t)(qnw_p >> 16));

  // Requested up front (none of it depends on this step's projections): the cached key rows this lane scores,
  // the cached value elements it accumulates, and the first query head of this wave -- the kernel is a chain of
  // memory round trips otherwise.
  const int CH = D / 4;                      // dims per chunk lane (32 for D = 128)
  const int kt = lane >> 2, kcn = lane & 3;  // lane = (key, chunk)
  uint4 kpre[4];
#pragma unroll
  for (int j4 = 0; j4 < 4; ++j4)
    if (kt < pos && j4 * 8 < CH && !rebuild0) kpre[j4] = *reinterpret_cast<const uint4*>(kc + (int64_t)kt * D + kcn * CH + j4 * 8);
  uint32_t vpre[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (t < pos && !rebuild0) vpre[t] = *reinterpret_cast<const uint32_t*>(vc + (int64_t)t * D + 2 * p);
  uint32_t qraw0 = (wave < G) ? *reinterpret_cast<const uint32_t*>(src + (kvh * G + wave) * D + 2 * p) : 0u;

  if (wave == 0) {  // key head: norm + rope -> cache + LDS
    uint32_t raw = *reinterpret_cast<const uint32_t*>(src + (H + kvh) * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.knw) {
      float ss = wave_sum_dpp(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), knw0));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), knw1));
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
  } else if (wave == 2 && rebuild0) {  // key 0 of this utterance: wave 0's arithmetic on the position-0 row
    uint32_t cs0 = *reinterpret_cast<const uint32_t*>(a.rope + (int64_t)p * 2);
    const float c0 = bf2f((bf16_t)(cs0 & 0xffff)), sn0 = bf2f((bf16_t)(cs0 >> 16));
    uint32_t raw = *reinterpret_cast<const uint32_t*>(src0 + (H + kvh) * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.knw) {
      float ss = wave_sum_dpp(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), knw0));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), knw1));
    }
    bf16_t o0 = f2bf(__fsub_rn(__fmul_rn(y0, c0), __fmul_rn(y1, sn0)));
    bf16_t o1 = f2bf(__fadd_rn(__fmul_rn(y1, c0), __fmul_rn(y0, sn0)));
    if (act) *reinterpret_cast<uint32_t*>(&s_k0b[2 * p]) = (uint32_t)o0 | ((uint32_t)o1 << 16);
  } else if (wave == 3 && rebuild0) {  // value 0
    if (act) *reinterpret_cast<uint32_t*>(&s_v0b[2 * p]) = *reinterpret_cast<const uint32_t*>(src0 + (H + KVH + kvh) * D + 2 * p);
  }
  __syncthreads();
  if (rebuild0) {   // what the cache loads above would have brought for key / value 0
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
      if (kt < pos && j4 * 8 < CH) kpre[j4] = *reinterpret_cast<const uint4*>(&s_k0b[kcn * CH + j4 * 8]);
    vpre[0] = *reinterpret_cast<const uint32_t*>(&s_v0b[2 * p]);
  }

  const float scale = (float)(1.0 / sqrt((double)D));
  for (int gq = wave; gq < G; gq += 4) {
    const int h = kvh * G + gq;
    uint32_t raw = (gq == wave) ? qraw0 : *reinterpret_cast<const uint32_t*>(src + h * D + 2 * p);
    float x0 = act ? bf2f((bf16_t)(raw & 0xffff)) : 0.f, x1 = act ? bf2f((bf16_t)(raw >> 16)) : 0.f;
    float y0 = x0, y1 = x1;
    if (a.qnw) {
      float ss = wave_sum_dpp(x0 * x0 + x1 * x1);
      float rstd = rsqrtf(ss / (float)D + a.eps);
      y0 = rbf(__fmul_rn(__fmul_rn(x0, rstd), qnw0));
      y1 = rbf(__fmul_rn(__fmul_rn(x1, rstd), qnw1));
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
  FMI_REQUIRE(!a.merge || a.ncb >= 2, "fast_attn: merged positions need two codebooks");
  hipLaunchKernelGGL(fast_attn_kernel, dim3(a.merge ? 2 * a.B : a.B, a.KVH), dim3(256), 0, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

}  // namespace fmi
