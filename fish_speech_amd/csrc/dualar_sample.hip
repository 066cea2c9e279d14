// dualar_sample.hip -- the sampler of the Dual-AR path (inference.py:43-93,118-144); split out of dualar_kernels.hip
// so that the translation units compile in parallel.
#include "dualar_kernels.h"
#include "dualar_dev.h"

namespace fmi {
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
    if (a.forced) tok = a.forced[(int64_t)slot * a.st.ncb1 + 1 + a.cb];
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
    if (a.forced) tok = a.forced[(int64_t)slot * a.st.ncb1];
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
  int bkw[4][32];       // per wave: how many of its keys lie 16 j .. 16 j + 15 key units below the block maximum
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

// ---- lane exchanges inside the VALU (round 6).  sample_small_kernel's tail is ONE wave walking a chain of dependent lane
// exchanges -- a 64-lane bitonic sort (21 stages), two prefix scans, the arg-max of the race -- and __shfl_* is a
// ds_bpermute round trip through the LDS crossbar per step.  Exchanges inside a row of 16 lanes are DPP moves:
// xor_lane<MASK> (common.h).  Pure data movement: the results are the same lanes' values as before, bit for bit.
__device__ inline int xor_lane_rt(int v, int stride, int lane) {   // stride known at compile time after unrolling
  switch (stride) {
    case 1: return xor_lane<1>(v, lane);
    case 2: return xor_lane<2>(v, lane);
    case 4: return xor_lane<4>(v, lane);
    case 8: return xor_lane<8>(v, lane);
    case 16: return xor_lane<16>(v, lane);
    default: return xor_lane<32>(v, lane);
  }
}
// inclusive prefix sum over the 64 lanes (WIDTH = 64) or over each half-wave (WIDTH = 32): Hillis-Steele inside the rows of
// 16 (row_shr:1, 2, 4, 8 with zeros shifted in), then the row totals carried across rows (row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3)
template <int WIDTH>
__device__ inline int incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);            // row_bcast15 -> rows 1, 3
  if constexpr (WIDTH == 64) v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2, 3
  return v;
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
    const float ov = __int_as_float(xor_lane_rt(__float_as_int(best), o, lane));
    const int oi = xor_lane_rt(best_id, o, lane);
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
  const bf16_t* lg = a.logits + (int64_t)b * a.ld;
  const int n = a.n;
  if (a.dbg_stop == 1) return;

  // ---- pass 1: coalesced loads (element tid + 256 j), keys to LDS, block max
  // (all 17 loads requested before the first is used -- clamped index, then a select: behind the `i < n` branch each
  // load had its own s_waitcnt vmcnt(0), seventeen dependent L2 round trips at the top of every sampler launch.
  // Round 5: they are also requested BEFORE the slot's parameters, which they do not depend on -- slot index, then six
  // state words, then the wait for top_k used to stand in front of them: two more round trips, to memory another XCD's
  // sampler wrote, before the first logit was even asked for)
  bf16_t lraw[SMALL_EPT];
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) lraw[j] = lg[min(tid + 256 * j, n - 1)];
  __builtin_amdgcn_sched_barrier(0);
  const int slot = a.row_slot ? a.row_slot[b] : b;

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

  float xv[SMALL_EPT];
  uint32_t kmax = 0;
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    const int i = tid + 256 * j;
    const bf16_t raw = i < n ? lraw[j] : (bf16_t)0xff80;  // -inf padding
    xv[j] = bf2f(raw);
    const uint32_t key = order_key(raw);
    if (i < n) {
      skey[i] = (uint16_t)key;
      kmax = max(kmax, key);
    }
  }
  kmax = wave_max_dpp_u(kmax);
  if (lane == 0) sh.wmax[wave] = kmax;
  if (tid < 128) (&sh.bkw[0][0])[tid] = 0;
  if (wave == 0) sh.wc[0][lane] = 0;     // (the one shared candidate list of the short path below)
  __syncthreads();
  kmax = max(max(sh.wmax[0], sh.wmax[1]), max(sh.wmax[2], sh.wmax[3]));
  const bf16_t maxbits = (kmax & 0x8000) ? (bf16_t)(kmax & 0x7fff) : (bf16_t)(~kmax & 0xffff);
  const float vmax = bf2f(maxbits);
  if (a.dbg_stop == 2) return;
  int k = top_k < n ? top_k : n;
  if (k > 64) k = 64;
  if (k < 1) k = 1;
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
  // Round 5: where do the k largest keys END?  Every wave counts its keys by their distance below the block maximum, in
  // 32 buckets of 16 key units (one octave of a bf16 is 128 units: four octaves) -- an LDS add per key that is in range,
  // and at most a few hundred of the 4097 are.  After the barrier every wave knows the smallest distance that already
  // holds k keys of the WHOLE row: nothing farther from the maximum can be among the k largest, so the waves only hand
  // on their keys inside that distance instead of each finding its own k largest by an eight-step radix descent.
  // (a thread's keys are CONTIGUOUS entries: runs of keys in one bucket -- a flat or masked row puts all 17 into the same
  // one -- are counted in a register and added once, so the worst row costs 256 LDS adds, not 4097: ADVICE r05)
  {
    int cb = -1, cc = 0;
#pragma unroll
    for (int j = 0; j < SMALL_EPT; ++j) {
      const uint32_t d = kmax - kr[j];
      const int bk = (kr[j] != 0u && d < 512u) ? (int)(d >> 4) : -1;
      if (bk != cb) {
        if (cc) atomicAdd(&sh.bkw[wave][cb], cc);
        cb = bk;
        cc = 0;
      }
      cc += bk >= 0;
    }
    if (cc) atomicAdd(&sh.bkw[wave][cb], cc);
  }
  __syncthreads();
  const float sumexp = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
  if (a.dbg_stop == 3) return;
  // bucket j of lane j (every wave evaluates this identically: no further barrier): inclusive prefixes over the buckets of
  // the row total, of this wave's own count and of the counts of the waves before it
  int jstar = -1, g_sel = 0, c_me = 0, c_base = 0;
  if (a.short_path) {
    const int bj = lane & 31;
    const int b0 = sh.bkw[0][bj], b1 = sh.bkw[1][bj], b2 = sh.bkw[2][bj], b3 = sh.bkw[3][bj];
    int pg = b0 + b1 + b2 + b3;
    int pm = wave == 0 ? b0 : (wave == 1 ? b1 : (wave == 2 ? b2 : b3));
    int pb = wave == 0 ? 0 : (wave == 1 ? b0 : (wave == 2 ? b0 + b1 : b0 + b1 + b2));
    pg = incl_scan_dpp<32>(pg);   // (each half-wave scans the 32 buckets on its own: lanes j and j + 32 agree)
    pm = incl_scan_dpp<32>(pm);
    pb = incl_scan_dpp<32>(pb);
    const uint32_t reach = (uint32_t)(__ballot(pg >= k) & 0xffffffffull);   // lanes 0-31: buckets whose prefix holds k keys
    if (reach) {
      jstar = __ffs((int)reach) - 1;
      g_sel = __builtin_amdgcn_readlane(pg, jstar);
      c_me = __builtin_amdgcn_readlane(pm, jstar);
      c_base = __builtin_amdgcn_readlane(pb, jstar);
    }
  }
  // one: the whole row has at most 64 keys inside the distance -> ONE shared list, sorted once by wave 0 (no per-wave sorts,
  // no merges); per_wave: more than 64 in the row, at most 64 in this wave -> the wave's list holds them all (the merge
  // keeps the 64 largest); otherwise this wave falls back to the descent (ties by the thousand, k beyond the reach).
  const bool one = jstar >= 0 && g_sel <= 64;
  const bool per_wave = jstar >= 0 && !one && c_me <= 64;

  // ---- top-k without block-wide rounds.  Every wave picks the k largest of ITS keys by a radix-4 descent whose
  // counts meet inside the wave (DPP reductions, no LDS, no barrier), compacts them in index order (ties on the
  // k-th key: lowest indices first), sorts them with a 64-lane bitonic network on the 32-bit word
  // key << 16 | ~index (unique, and "larger word" == "larger logit, then lower index": the reference's stable
  // descending sort); wave 0 then merges the four sorted lists pairwise (max of one list against the reverse of the
  // other is bitonic and holds the 64 largest of both: six more stages sort it).  One barrier in total; lane r of
  // wave 0 ends up with the rank-r candidate.
  uint32_t thr = 0;
  if (one || per_wave) {   // keys >= kmax - (16 (jstar + 1) - 1), i.e. strictly above thr
    const int t0 = (int)kmax - (16 * (jstar + 1) - 1);
    thr = t0 > 1 ? (uint32_t)(t0 - 1) : 0u;
  } else {
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
  }
  const bool take_eq = !(one || per_wave);   // the short paths take every key above thr and none equal to it
  int my_gt = 0, my_eq = 0;
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    my_gt += kr[j] > thr;
    my_eq += (kr[j] == thr) && (thr != 0) && take_eq;
  }
  if (a.dbg_stop == 4) return;
  const int inc_gt = incl_scan_dpp<64>(my_gt), inc_eq = incl_scan_dpp<64>(my_eq);
  const int c_gt = __builtin_amdgcn_readlane(inc_gt, 63);     // this wave's keys above its threshold (< k; short paths: <= 64)
  const int need_eq = k - c_gt;
  int off_gt = inc_gt - my_gt + (one ? c_base : 0), off_eq = inc_eq - my_eq;
  uint32_t* wc = one ? sh.wc[0] : sh.wc[wave];                // `one`: every wave appends to the shared list (zeroed above)
  if (!one) {
    wc[lane] = 0;                                               // word 0 = empty place (real words have key >= 0x7f)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
#pragma unroll
  for (int j = 0; j < SMALL_EPT; ++j) {
    const uint32_t key = kr[j];
    const uint32_t word = (key << 16) | (uint32_t)(0xffff - (i0 + j));
    if (key > thr) {
      wc[off_gt++] = word;
    } else if (key == thr && thr != 0 && take_eq) {
      if (off_eq < need_eq) wc[c_gt + off_eq] = word;
      ++off_eq;
    }
  }
  // bitonic sort, descending over the 64 lanes
  auto sort64 = [&](uint32_t w) -> uint32_t {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const uint32_t o = (uint32_t)xor_lane_rt((int)w, stride, lane);
        const bool take_max = ((lane & stride) == 0) == ((lane & size) == 0);
        w = take_max ? max(w, o) : min(w, o);
      }
    return w;
  };
  uint32_t w = 0;
  if (!one) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    w = sort64(wc[lane]);
    wc[lane] = w;
  }
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
      const uint32_t o = (uint32_t)xor_lane_rt((int)m, stride, lane);
      m = ((lane & stride) == 0) ? max(m, o) : min(m, o);
    }
    return m;
  };
  uint32_t top;
  if (one) {
    top = sort64(sh.wc[0][lane]);
  } else {
    const uint32_t m01 = merge_desc(w, sh.wc[1][63 - lane]);
    const uint32_t m23 = merge_desc(sh.wc[2][lane], sh.wc[3][63 - lane]);
    const uint32_t m23_rev = (uint32_t)__shfl((int)m23, 63 - lane, 64);
    top = merge_desc(m01, m23_rev);
  }

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
    if (a.forced) tok = a.forced[(int64_t)slot * a.st.ncb1 + 1 + a.cb];
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
    if (a.forced) tok = a.forced[(int64_t)slot * a.st.ncb1];
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
    // (vector types, not HIP's uint4 struct: copies of the struct went through SCRATCH -- 112 bytes per lane, every row
    // piece stored right behind its load and re-loaded for the store, so the loads were not in flight together)
    const u32x4* src1 = reinterpret_cast<const u32x4*>(a.fast_emb + (int64_t)tok * a.fdim);
    const u32x4* src2 = a.qkv0_tab ? reinterpret_cast<const u32x4*>(a.qkv0_tab + (int64_t)tok * a.qkv0_dim) : nullptr;
    u32x4* dst1 = reinterpret_cast<u32x4*>(a.xf + (int64_t)b * a.fdim);
    u32x4* dst2 = a.qkv0_tab ? reinterpret_cast<u32x4*>(a.qkv0_out + (int64_t)b * a.qkv0_dim) : nullptr;
    const int nthr = gather_all ? 256 : 64, t0 = gather_all ? tid : lane;
    constexpr int GB = 6;
    for (int base = 0; base < n1 + n2; base += GB * nthr) {
      u32x4 gv[GB];
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        const int i = base + t0 + j * nthr;
        gv[j] = (u32x4){0u, 0u, 0u, 0u};
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
    static const bool descent = []() { const char* e = getenv("FMI_SAMPLE_DESCENT"); return e && atoi(e) != 0; }();
    SampleArgs b = a;
    b.short_path = descent ? 0 : 1;
    hipLaunchKernelGGL(sample_small_kernel, dim3(a.B), dim3(256), smem, s, b);
    FMI_CHECK_HIP(hipGetLastError());
    return FMI_OK;
  }
  size_t smem = sizeof(SamplerShared) + (size_t)a.n * 2 + 16;
  hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(256), smem, s, a);
  FMI_CHECK_HIP(hipGetLastError());
  return FMI_OK;
}

}  // namespace fmi
