// dualar_kernels.h -- launch wrappers of the Dual-AR decode kernels (dualar_kernels.hip).
#pragma once
#include "common.h"

namespace fmi {

constexpr int KV_PAGE = 64;       // tokens per KV page
constexpr int RAS_WIN = 10;       // inference.py:49
constexpr int SAMPLER_MAXK = 1024;

enum Epilogue { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SILU = 2 };

// Per-slot generation state, structure-of-arrays in device memory.
struct SlotState {
  int32_t* pos;        // [S] next input position (= tokens already in the KV cache)
  int32_t* frame;      // [S] frames generated so far (0 while the prefill frame is being built)
  int32_t* done;       // [S] 1 after <|im_end|>, 2 after the reservation ran out
  int32_t* limit;      // [S] max position (exclusive) the slot's pages cover
  int32_t* cur;        // [S][1+ncb] tokens of the last completed frame (next slow input)
  int32_t* window;     // [S][1+ncb][RAS_WIN]
  int32_t* out;        // [S][max_frames][1+ncb]
  float* temperature;  // [S] bf16-rounded
  float* top_p;        // [S] bf16-rounded
  int32_t* top_k;      // [S]
  uint32_t* seed;      // [S]
  int32_t* use_ras;    // [S]
  int32_t* block_table;  // [S][max_pages]
  int max_pages, max_frames, ncb1;
};

// Weight tile packing: row-major (N,K) bf16 -> [N/16][K/32][64 lanes][8] MFMA fragments; inside every pair of
// k-tiles the 8-element chunks are permuted (packed_k0 in dualar_kernels.hip) so that one all-lanes activation load
// serves both tiles.
// interleave: 0 identity; 1/2 = w1/w3 halves of the SwiGLU pair, 16-row blocks alternating.
int launch_pack_weight(const bf16_t* src, bf16_t* dst, int N, int K, int interleave, hipStream_t s);
// int8 row-major (N,K) -> [N/16][K/64][64 lanes][2 tiles][8 int8]: the two k-tiles of a pair side by side, same k
// permutation as the bf16 layout (one 16-byte load per lane per pair).  K must be a multiple of 64.
int launch_pack_weight_int8(const int8_t* src, int8_t* dst, int N, int K, int interleave, hipStream_t s);
// int8 row-major -> bf16 row-major (exact), for the bf16 copy the prefill / M > 8 paths use
int launch_dequant_int8(const int8_t* src, bf16_t* dst, int64_t n, hipStream_t s);
// scale vector into packed row order (interleave as launch_pack_weight)
int launch_pack_scale(const bf16_t* src, bf16_t* dst, int N, int interleave, hipStream_t s);
// gather `n` rows listed in ids_dev (int32 vocab ids, on device) then pack (live LM-head rows).
int launch_pack_rows_gather(const bf16_t* src, const int32_t* ids_dev, bf16_t* dst, int n, int n_pad,
                            int K, hipStream_t s);

struct LinearArgs {
  const bf16_t* wp;        // packed weights
  const bf16_t* x;         // [M][ldx]
  int ldx;
  const bf16_t* norm_w;    // fused RMSNorm weight (skinny path only) or nullptr
  float eps;
  const bf16_t* res;       // residual [M][ldr] for EPI_RESIDUAL
  int ldr;
  bf16_t* out;             // [M][ldo]
  int ldo;
  int M, N, K;             // N = packed rows (2*ffn for EPI_SILU)
  int epi;
  // weight-only int8 checkpoints (tools/llama/quantize.py:204-229): out = bf16(bf16(acc) * scale[row]) before the
  // epilogue.  `scale` is in PACKED row order (gate/up interleaved for SwiGLU); `wq` = the int8 tiles
  // (launch_pack_weight_int8) streamed by the M <= 8 decode GEMV instead of `wp` -- half the bytes.
  const bf16_t* scale;
  const int8_t* wq;
  // row-balanced decode copy of `wp` (launch_repack_rows; nullptr = none): streamed instead of `wp` by the M <= 8
  // decode GEMV, one work-group per CU-sized share of the rows.
  const bf16_t* wr;
  // nn.Linear bias [N] (only fast_project_in has one, llama.py:666): out = bf16(acc + bias), ONE rounding like torch's
  // addmm; skinny kernel, EPI_STORE only
  const bf16_t* bias;
  int late_epi;            // A/B measurements only (FMI_GEMV_LATE_EPI): epilogue operands loaded after the last barrier
  // tools/gemv_ksplit_probe.hip only (linear_skinny_kernel<..., KSL > 1>: the K range split over gridDim.y work-groups
  // per row block -- fewer activation requests per weight byte -- with a fixed-order cross-work-group reduction):
  float* part;             // [gridDim.y][gridDim.x][TILES][256] fp32 partial sums
                           // (launch_linear_tiled with linear_tiled_ksplit(N, K) > 1: its partial tiles, linear_tiled_part_floats)
  unsigned* cnt;           // [gridDim.x] arrival counters (zero between launches); nullptr = partials only (upper bound)
  int part_proto;          // 0: partials by plain stores + agent-scope release / acquire fences; 1: partials by agent-scope
                           //    (write-through / L2-bypassing) relaxed atomic stores and loads, no cache-wide fence
};
int launch_linear_skinny(const LinearArgs& a, hipStream_t s);  // M <= 16

// Row-balanced decode geometry of an (N, K) linear on the 256-CU part: `rows` weight rows per tile (< 16 when the
// 16-row tiling leaves CUs idle), `tiles` tiles per work-group, `wgs` work-groups.  ok == false: the 16-row layout is
// already balanced, the shape has no balanced tiling, or balancing was measured not to pay (SwiGLU pairs); the decode
// GEMV then streams the 16-row tiles.
struct RowPlan {
  bool ok;
  int rows, tiles, wgs;
  int64_t elems;   // bf16 elements of the row-balanced copy (wgs * tiles * rows * K)
};
RowPlan skinny_row_plan(int N, int K, int epi);
bool skinny_rows_supported(int N, int K, int epi, bool norm);   // a row-balanced kernel variant exists for the shape
// 16-row packed tiles (launch_pack_weight, with its SwiGLU interleave) -> the row-balanced copy of `plan`
int launch_repack_rows(const bf16_t* packed16, bf16_t* dst, int N, int K, int epi, const RowPlan& plan, hipStream_t s);
// any M, no fused norm.  variant: 0 = default (FMI_GEMM env: d / l / w, else the built-in default), 1 = LDS-staged
// 4-wave kernel, 2 = LDS-staged wave-specialised kernel; force_direct: operands straight from L2.  All bit-identical.
int launch_linear_tiled(const LinearArgs& a, hipStream_t s, bool force_direct = false, int variant = 0);
// Fixed split of the contraction (round 6) for the prefill GEMMs whose 256-column tiling leaves most CUs idle at a few
// hundred to a few thousand rows (N <= 2560: ten column tiles) and whose K is long enough to pay for a reduce pass (wo, w2:
// K >= 4096): S work-groups per output tile, each over 1 / S of the k-steps, fp32 partial tiles in LinearArgs::part, summed
// in range order by a second launch that also runs the epilogue.  S depends on (N, K) ONLY -- never on M -- so a row's
// bits do not depend on the rows it travels with (prefix reuse, ragged prefill).  1 = no split.
int linear_tiled_ksplit(int N, int K);
int64_t linear_tiled_part_floats(int M, int N, int K);   // floats LinearArgs::part must hold for a call (0: no split)
int launch_rmsnorm_rows(const bf16_t* x, int ldx, const bf16_t* w, float eps, bf16_t* out, int ldo,
                        int M, int K, hipStream_t s, bf16_t* out2 = nullptr);

struct EmbedArgs {
  const bf16_t* emb;      // [V][dim]
  const bf16_t* cb_emb;   // [ncb*cbs][dim]
  const int32_t* tokens;  // rows x (1+ncb) when row_slot==nullptr, else SlotState.cur
  const int32_t* row_slot;
  bf16_t* out;            // [rows][dim]
  int rows, dim, ncb, cbs, sem_begin, sem_end, scale;
};
int launch_embed(const EmbedArgs& a, hipStream_t s);
int launch_gather_rows(const bf16_t* src, int ld_src, const int32_t* row_idx, bf16_t* dst, int ld_dst,
                       int rows, int cols, hipStream_t s);

struct AttnArgs {
  const bf16_t* qkv;      // [rows][(H+2KVH)*D]
  bf16_t* q;              // [rows][H*D] scratch (normed+roped q)
  bf16_t* out;            // [rows][H*D]
  bf16_t* kpool;          // [pages][KVH][KV_PAGE][D]
  bf16_t* vpool;
  const bf16_t* qnw;      // q_norm weight or nullptr
  const bf16_t* knw;
  const bf16_t* rope;     // [max_seq][D/2][2]
  const int32_t* row_slot;  // [rows]
  const int32_t* row_pos;   // [rows] or nullptr -> SlotState.pos[slot]
  const int32_t* block_table;
  const int32_t* slot_pos;
  const int32_t* slot_done;   // decode only: a finished slot (SlotState.done != 0) no longer appends K/V
  const int4* qtiles;         // prefill (MFMA kernel): per query tile {row0, rows (<= qtile_rows), slot, first position}
  int n_qtiles;
  int qtile_rows;             // 16, 32 or 48: rows a work-group of the prefill kernel owns (1, 2 or 3 column groups)
  int max_pages;
  int rows, H, KVH, D;
  float eps;
  // decode at long contexts: rows whose position is >= long_thr are served by launch_attn_decode_long (MFMA, key ranges
  // split over work-groups, partial states in `part`), the others by launch_attn_decode_fused; 0 = every row VALU
  int long_thr;
  float* part;                // [rows][KVH][ATTN_Z][G][2 + D] fp32 (attn_decode_long_part_floats)
};
int launch_attn_prep(const AttnArgs& a, hipStream_t s);
int launch_attn(const AttnArgs& a, hipStream_t s);               // VALU kernel (kept for A/B parity runs)
int launch_attn_prefill_mfma(const AttnArgs& a, hipStream_t s);  // MFMA flash attention over the query tiles
int launch_attn_decode_fused(const AttnArgs& a, hipStream_t s);  // one row per slot only (rows below a.long_thr if that is set)
bool attn_decode_long_supported(int H, int KVH, int D);
int64_t attn_decode_long_part_floats(int rows, int H, int D);
int launch_attn_decode_long(const AttnArgs& a, hipStream_t s);   // rows at or beyond a.long_thr: MFMA kernel + merge

struct FastAttnArgs {
  const bf16_t* qkv;   // [B][(H+2KVH)*D]
  bf16_t* out;         // [B][H*D]
  bf16_t* kc;          // [slots][KVH][ncb][D]
  bf16_t* vc;
  const bf16_t* qnw;
  const bf16_t* knw;
  const bf16_t* rope;  // [ncb][D/2][2]
  const int32_t* row_slot;
  int B, H, KVH, D, ncb, pos;
  float eps;
  int merge;           // 1: rows [0, B) are position 0 and rows [B, 2B) position 1 of the same B utterances (pos ignored)
};
int launch_fast_attn(const FastAttnArgs& a, hipStream_t s);

struct SampleArgs {
  const bf16_t* logits;   // [B][ld]
  int B, n, ld;
  const int32_t* ids;     // row index -> vocab id (nullptr = identity)
  const int32_t* row_slot;  // nullptr = identity
  SlotState st;
  int mode;               // 0 slow (2 draws + RAS + cb0), 1 fast codebook, 2 bare op (tests)
  int cb;                 // fast: codebook index (1..ncb-1)
  int sem_begin, sem_end, im_end, cbs;
  const bf16_t* fast_emb; // [cbs][fdim] gathered into xf after the draw
  bf16_t* xf;             // [B][fdim]
  int fdim;
  // first fast layer's wqkv(rmsnorm(fast_emb[code])) precomputed per code (a pure function of the code): the row of
  // the drawn code is gathered next to the embedding, so the next fast position skips that GEMV
  const bf16_t* qkv0_tab; // [cbs][qkv0_dim] or nullptr
  bf16_t* qkv0_out;       // [B][qkv0_dim]
  int qkv0_dim;
  // mode 2 (op-level): explicit parameters
  float temperature, top_p;
  int top_k;
  uint32_t seed;
  int frame, draw;
  const int32_t* prev;    // [B][RAS_WIN] or nullptr
  int32_t* out_tok;       // [B]
  int small_k;            // 1 = every slot uses top_k <= 64 (fast single-wave finish)
  int dbg_stop;           // profiling only: return after stage N of sample_small_kernel (0 = run all)
  const int32_t* forced;  // test seam (fmi_dualar_fast_chain_forced): [slots][ncb1] frame that REPLACES every draw, or nullptr
  int short_path;         // set by launch_sample: 1 = bucket-count candidate selection (round 5), 0 = radix descent in every wave (FMI_SAMPLE_DESCENT=1; A/B)
};
int launch_sample(const SampleArgs& a, hipStream_t s);

}  // namespace fmi
