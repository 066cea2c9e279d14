// dualar.hip -- host side of the Dual-AR decoder behind the C ABI (include/fishmi.h):
// weight arena layout, paged KV allocator, per-slot state, the frame step and its hipGraph.
//
// Reference call path replaced here (fish_speech/models/text2semantic/):
//   generate / decode_n_tokens / decode_one_token_ar   inference.py:96-359
//   DualARTransformer.forward_generate(_fast)          llama.py:390-466,799-828
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "dualar_kernels.h"

using namespace fmi;

namespace {

struct LayerW {
  bf16_t *wqkv, *wo, *w13, *w2, *attn_norm, *ffn_norm, *q_norm, *k_norm;
  // weight-only int8 checkpoints (cfg.weight_int8): int8 tiles for the decode GEMV + per-row scales (packed order);
  // the bf16 pointers above then hold the exactly dequantised weights for the prefill / M > 8 paths
  int8_t *q_wqkv = nullptr, *q_wo = nullptr, *q_w13 = nullptr, *q_w2 = nullptr;
  bf16_t *s_wqkv = nullptr, *s_wo = nullptr, *s_w13 = nullptr, *s_w2 = nullptr;
  // row-balanced decode copies (outside the arena, derived on every rank after the weights are in place):
  // streamed by the M <= 8 decode GEMV instead of the 16-row tiles where those leave CUs idle (skinny_row_plan)
  bf16_t *r_wqkv = nullptr, *r_wo = nullptr, *r_w2 = nullptr;   // (w1|w3 keeps its 16-row tiles: skinny_row_plan)
};

struct Dims {
  int dim, H, KVH, D, ffn, qkv;
};

struct Workspace {
  int rows = 0;
  bf16_t *x = nullptr, *xn = nullptr, *qkv = nullptr, *q = nullptr, *ao = nullptr, *act = nullptr;
  int32_t *row_slot = nullptr, *row_pos = nullptr, *last_rows = nullptr;
  int4* qtiles = nullptr;   // prefill attention: query tiles {row0, rows, slot, first position}
  int n_qtiles = 0;
  int qtile_rows = 16;      // rows per tile: 16, 32 or 48 (chosen per prefill call from the row count)
};

}  // namespace

struct fmi_dualar {
  fmi_dualar_config cfg;
  Dims slow, fast;
  char* arena = nullptr;
  int64_t arena_bytes = 0;
  // arena regions
  std::vector<LayerW> L, FL;
  bf16_t *emb = nullptr, *cb_emb = nullptr, *norm = nullptr, *head_live = nullptr, *fast_emb = nullptr,
         *fast_norm = nullptr, *fast_out = nullptr, *rope = nullptr, *fast_rope = nullptr;
  int8_t* q_fast_out = nullptr;   // int8 checkpoints: fast_output is an nn.Linear too
  bf16_t* s_fast_out = nullptr;
  void* staging2 = nullptr;       // dequantisation scratch of the int8 loader
  size_t staging2_bytes = 0;
  int32_t* live_ids = nullptr;
  int n_live = 0, n_live_pad = 0;
  std::set<std::string> loaded;
  bool ready = false, rope_loaded = false, fast_rope_loaded = false;

  // runtime
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  int max_batch = 0, max_seq = 0, n_pages = 0, max_pages = 0, max_frames = 0;
  std::vector<bf16_t*> kpool, vpool, fkc, fvc;
  SlotState st{};
  std::vector<int> free_pages;
  std::vector<std::vector<int>> slot_pages;
  Workspace ws;
  bf16_t *hn = nullptr, *hf = nullptr, *xl = nullptr, *xf = nullptr, *logits = nullptr, *flogits = nullptr, *ftrace = nullptr;
  // fast positions 0 and 1 of a frame in one pass (tail): rows [0, B) = the hidden states, rows [B, 2B) = the
  // embeddings of the slow token's codes; [2 max_batch][fast_dim]
  bf16_t* x01 = nullptr;
  bool merge01 = true;                 // FMI_NO_MERGE01=1: positions 0 and 1 as two passes (rounds 1-3; A/B runs)
  // fast_dim != dim (llama.py:665-668): packed fast_project_in weight [fast_dim][dim] + bias in the arena, and the
  // projected hidden rows [max_batch][fast_dim] fast step 0 runs on
  bf16_t *fpi_w = nullptr, *fpi_b = nullptr, *hfp = nullptr;
  // fast layer 0 sees fast_embeddings[code] at every codebook position >= 1, so its wqkv(rmsnorm(.)) output is a
  // pure function of the code: tabulated once with the same GEMV kernel (batch-invariant bits), gathered by the
  // sampler; 9 of the 40 fast wqkv GEMVs of a frame (31.5 MB each at the S2 shape) become 8-row gathers
  bf16_t *qkv0_tab = nullptr, *qkv0_pre = nullptr;
  bool qkv0_tried = false;
  bool rows_tried = false;             // row-balanced decode copies (LayerW::r_*) derived
  std::vector<void*> row_copies;       // their allocations
  bool use_graph = true, ignore_eos = false;
  int trace = 0;   // fmi_dualar_set_trace: 0 off, 1 per-position fast logits kept + plain GEMV path, 2 kept on the frame loop's own path (table)
  float* gemm_part = nullptr;        // partial tiles of the split-contraction prefill GEMM (linear_tiled_ksplit), grown on demand
  int64_t gemm_part_floats = 0;
  bool skinny32 = false;             // linear(): rows 17-32 may take the decode GEMV (set by tail() around the merged fast pass)
  const int32_t* forced = nullptr;   // fmi_dualar_fast_chain_forced only: the frame that replaces the draws
  bool tail_in_normed = false;       //   "  : tail()'s input rows are already normed (the hidden the reference hands over)
  bool force_tiled = false;
  int attn_impl = 1;   // prefill attention: 1 = MFMA flash kernel with LDS-staged K/V tiles, 0 = VALU kernel (A/B parity)
  // decode attention: rows at or beyond this position run on the MFMA kernel + merge (launch_attn_decode_long), the
  // others on the fused VALU kernel; 0 = VALU for every row.  FMI_ATTN_THR overrides, fmi_dualar_set_attn_long_threshold
  // sets.  Which kernels a frame launches (attn_mask: 1 = VALU, 2 = MFMA) follows from the host's view of the slots'
  // positions (pos_host: exact for live slots, an over-estimate for finished ones, whose outputs nobody reads).
  int attn_long_thr = 1024;   // measured break-even at B = 8 (profiles/r03_attn_decode.txt): 19.2 vs 19.0 us at 1 k keys, 24.1 vs 32.3 at 2 k
  int attn_mask = 1;
  std::vector<int> pos_host;
  float* attn_part = nullptr;
  int max_top_k = 0;  // largest top_k over the LIVE slots (selects the sampler variant the graphs embed)
  std::vector<int> slot_top_k;  // per slot, 0 = released
  std::map<int, hipGraphExec_t> graphs;
  bool out_pending = false;             // frames of a decode call the caller's stream has not been ordered after
  std::vector<int32_t> row_slot_host;   // what ws.row_slot[0..n) holds on the device (empty: unknown)
  void* staging = nullptr;
  size_t staging_bytes = 0;
  float last_ms = 0.f;
  int launches = 0;        // counted while building the current eager sequence
  int frame_launches = 0;  // kernel launches of one decode frame (set whenever a frame is built)
};

namespace {

Dims slow_dims(const fmi_dualar_config& c) {
  return {c.dim, c.n_head, c.n_local_heads, c.head_dim, c.intermediate_size,
          (c.n_head + 2 * c.n_local_heads) * c.head_dim};
}
Dims fast_dims(const fmi_dualar_config& c) {
  return {c.fast_dim, c.fast_n_head, c.fast_n_local_heads, c.fast_head_dim, c.fast_intermediate_size,
          (c.fast_n_head + 2 * c.fast_n_local_heads) * c.fast_head_dim};
}

int count_live(const fmi_dualar_config& c) {
  int n = c.semantic_end_id - c.semantic_begin_id + 1;
  if (c.im_end_id < c.semantic_begin_id || c.im_end_id > c.semantic_end_id) n += 1;
  return n;
}

int validate(const fmi_dualar_config& c) {
  FMI_REQUIRE(c.dim > 0 && c.dim % 32 == 0, "dim=%d must be a positive multiple of 32", c.dim);
  FMI_REQUIRE(c.fast_dim > 0 && c.fast_dim % 32 == 0, "fast_dim=%d must be a positive multiple of 32", c.fast_dim);
  FMI_REQUIRE(c.fast_dim == c.dim || !c.weight_int8,
              "fast_dim != dim with weight-only int8: the reference's int8 Linear has no bias (quantize.py:204-229), "
              "its fast_project_in (llama.py:666) has one");
  FMI_REQUIRE(c.intermediate_size % 32 == 0 && c.fast_intermediate_size % 32 == 0, "intermediate_size %% 32");
  FMI_REQUIRE(c.head_dim == 32 || c.head_dim == 64 || c.head_dim == 128, "head_dim must be 32/64/128");
  FMI_REQUIRE(c.fast_head_dim == 32 || c.fast_head_dim == 64 || c.fast_head_dim == 128, "fast_head_dim 32/64/128");
  FMI_REQUIRE(c.n_head % c.n_local_heads == 0 && c.fast_n_head % c.fast_n_local_heads == 0, "GQA ratio");
  int g = c.n_head / c.n_local_heads;
  FMI_REQUIRE(g == 1 || g == 2 || g == 4, "n_head/n_local_heads must be 1, 2 or 4");
  FMI_REQUIRE((c.n_head * c.head_dim) % 32 == 0 && (c.fast_n_head * c.fast_head_dim) % 32 == 0, "H*D %% 32");
  FMI_REQUIRE(c.codebook_size % 16 == 0, "codebook_size %% 16");
  FMI_REQUIRE(c.num_codebooks >= 2 && c.num_codebooks <= 16, "num_codebooks in [2,16]");
  FMI_REQUIRE(c.semantic_begin_id >= 0 && c.semantic_end_id >= c.semantic_begin_id &&
                  c.semantic_end_id < c.vocab_size && c.im_end_id >= 0 && c.im_end_id < c.vocab_size,
              "semantic/im_end ids out of range");
  FMI_REQUIRE(c.max_seq_len > 0, "max_seq_len");
  return FMI_OK;
}

// Walk the arena layout.  With h == nullptr only the size is computed.
int64_t layout(const fmi_dualar_config& c, fmi_dualar* h) {
  int64_t off = 0;
  char* base = h ? h->arena : nullptr;
  auto take = [&](int64_t elems, int64_t esize) -> void* {
    void* p = base ? base + off : nullptr;
    off = align_up(off + elems * esize, 256);
    return p;
  };
  const Dims s = slow_dims(c), f = fast_dims(c);
  const int n_live = count_live(c), n_live_pad = (int)align_up(n_live, 32);  // even tile count
  auto layer = [&](const Dims& d) {
    LayerW w;
    w.wqkv = (bf16_t*)take((int64_t)d.qkv * d.dim, 2);
    w.wo = (bf16_t*)take((int64_t)d.dim * d.H * d.D, 2);
    w.w13 = (bf16_t*)take((int64_t)2 * d.ffn * d.dim, 2);
    w.w2 = (bf16_t*)take((int64_t)d.dim * d.ffn, 2);
    w.attn_norm = (bf16_t*)take(d.dim, 2);
    w.ffn_norm = (bf16_t*)take(d.dim, 2);
    w.q_norm = (bf16_t*)take(d.D, 2);
    w.k_norm = (bf16_t*)take(d.D, 2);
    if (c.weight_int8) {
      w.q_wqkv = (int8_t*)take((int64_t)d.qkv * d.dim, 1);
      w.q_wo = (int8_t*)take((int64_t)d.dim * d.H * d.D, 1);
      w.q_w13 = (int8_t*)take((int64_t)2 * d.ffn * d.dim, 1);
      w.q_w2 = (int8_t*)take((int64_t)d.dim * d.ffn, 1);
      w.s_wqkv = (bf16_t*)take(d.qkv, 2);
      w.s_wo = (bf16_t*)take(d.dim, 2);
      w.s_w13 = (bf16_t*)take((int64_t)2 * d.ffn, 2);
      w.s_w2 = (bf16_t*)take(d.dim, 2);
    }
    return w;
  };
  bf16_t* emb = (bf16_t*)take((int64_t)c.vocab_size * c.dim, 2);
  bf16_t* cb = (bf16_t*)take((int64_t)c.codebook_size * c.num_codebooks * c.dim, 2);
  bf16_t* norm = (bf16_t*)take(c.dim, 2);
  bf16_t* head = (bf16_t*)take((int64_t)n_live_pad * c.dim, 2);
  int32_t* ids = (int32_t*)take(n_live_pad, 4);
  bf16_t* femb = (bf16_t*)take((int64_t)c.codebook_size * c.fast_dim, 2);
  bf16_t* fnorm = (bf16_t*)take(c.fast_dim, 2);
  bf16_t* fout = (bf16_t*)take((int64_t)c.codebook_size * c.fast_dim, 2);
  bf16_t* rope = (bf16_t*)take((int64_t)c.max_seq_len * c.head_dim, 2);
  bf16_t* frope = (bf16_t*)take((int64_t)c.num_codebooks * c.fast_head_dim, 2);
  const bool proj = c.fast_dim != c.dim;
  bf16_t* fpiw = proj ? (bf16_t*)take((int64_t)c.fast_dim * c.dim, 2) : nullptr;
  bf16_t* fpib = proj ? (bf16_t*)take(c.fast_dim, 2) : nullptr;
  int8_t* qfout = c.weight_int8 ? (int8_t*)take((int64_t)c.codebook_size * c.fast_dim, 1) : nullptr;
  bf16_t* sfout = c.weight_int8 ? (bf16_t*)take(c.codebook_size, 2) : nullptr;
  std::vector<LayerW> L, FL;
  for (int i = 0; i < c.n_layer; ++i) L.push_back(layer(s));
  for (int i = 0; i < c.n_fast_layer; ++i) FL.push_back(layer(f));
  if (h) {
    h->emb = emb; h->cb_emb = cb; h->norm = norm; h->head_live = head; h->live_ids = ids;
    h->fast_emb = femb; h->fast_norm = fnorm; h->fast_out = fout; h->rope = rope; h->fast_rope = frope;
    h->L = L; h->FL = FL; h->n_live = n_live; h->n_live_pad = n_live_pad;
    h->q_fast_out = qfout; h->s_fast_out = sfout;
    h->fpi_w = fpiw; h->fpi_b = fpib;
  }
  return off;
}

int ensure_staging(fmi_dualar* h, size_t bytes) {
  if (h->staging_bytes >= bytes) return FMI_OK;
  if (h->staging) FMI_CHECK_HIP(hipFree(h->staging));
  h->staging = nullptr;
  h->staging_bytes = 0;
  FMI_CHECK_HIP(hipMalloc(&h->staging, bytes));
  h->staging_bytes = bytes;
  return FMI_OK;
}

int sync_in(fmi_dualar* h, void* user_stream) {
  FMI_CHECK_HIP(hipEventRecord(h->ev_in, (hipStream_t)user_stream));
  FMI_CHECK_HIP(hipStreamWaitEvent(h->stream, h->ev_in, 0));
  return FMI_OK;
}
int sync_out(fmi_dualar* h, void* user_stream) {
  FMI_CHECK_HIP(hipEventRecord(h->ev_out, h->stream));
  FMI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)user_stream, h->ev_out, 0));
  h->out_pending = false;
  return FMI_OK;
}

template <typename T>
int dev_alloc(T** p, int64_t n) {
  FMI_CHECK_HIP(hipMalloc((void**)p, (size_t)(n * (int64_t)sizeof(T))));
  FMI_CHECK_HIP(hipMemset(*p, 0, (size_t)(n * (int64_t)sizeof(T))));
  return FMI_OK;
}

int free_ws(Workspace& w) {
  void* ptrs[] = {w.x, w.xn, w.qkv, w.q, w.ao, w.act, w.row_slot, w.row_pos, w.last_rows, w.qtiles};
  for (void* p : ptrs)
    if (p) hipFree(p);
  w = Workspace();
  return FMI_OK;
}

void drop_graphs(fmi_dualar* h) {
  for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
  h->graphs.clear();
}

// Tables and copies DERIVED from the weights (row-balanced decode copies, fast layer-0 q|k|v table) are dropped when a
// tensor is (re)loaded after they were built; the next prefill rebuilds them.
int invalidate_derived(fmi_dualar* h) {
  if (!h->rows_tried && !h->qkv0_tried) return FMI_OK;
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  for (void* p : h->row_copies) hipFree(p);
  h->row_copies.clear();
  for (auto* LL : {&h->L, &h->FL})
    for (auto& w : *LL) w.r_wqkv = w.r_wo = w.r_w2 = nullptr;
  if (h->qkv0_tab) hipFree(h->qkv0_tab);
  if (h->qkv0_pre) hipFree(h->qkv0_pre);
  h->qkv0_tab = h->qkv0_pre = nullptr;
  h->rows_tried = h->qkv0_tried = false;
  return FMI_OK;
}

int ensure_rows(fmi_dualar* h, int rows) {
  if (h->ws.rows >= rows) return FMI_OK;
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  free_ws(h->ws);
  const Dims& s = h->slow;
  const Dims& f = h->fast;
  const int64_t dim = std::max(s.dim, f.dim), qkv = std::max(s.qkv, f.qkv), hd = std::max(s.H * s.D, f.H * f.D),
                ffn = std::max(s.ffn, f.ffn);
  Workspace& w = h->ws;
  FMI_CHECK(dev_alloc(&w.x, rows * dim));
  FMI_CHECK(dev_alloc(&w.xn, rows * dim));
  FMI_CHECK(dev_alloc(&w.qkv, rows * qkv));
  FMI_CHECK(dev_alloc(&w.q, rows * hd));
  FMI_CHECK(dev_alloc(&w.ao, rows * hd));
  FMI_CHECK(dev_alloc(&w.act, rows * ffn));
  FMI_CHECK(dev_alloc(&w.row_slot, rows));
  FMI_CHECK(dev_alloc(&w.row_pos, rows));
  FMI_CHECK(dev_alloc(&w.last_rows, std::max(rows, 1)));
  FMI_CHECK(dev_alloc(&w.qtiles, (int64_t)rows / 16 + h->max_batch + 1));
  w.rows = rows;
  return FMI_OK;
}

// out = linear(norm?(x)) for M rows; picks the skinny (fused norm) or tiled path.
int linear(fmi_dualar* h, const bf16_t* x, int ldx, const bf16_t* wp, const bf16_t* norm_w, const bf16_t* res,
           int ldr, bf16_t* out, int ldo, int M, int N, int K, int epi, hipStream_t s, const int8_t* wq = nullptr,
           const bf16_t* scale = nullptr, const bf16_t* wr = nullptr, const bf16_t* bias = nullptr) {
  LinearArgs a{};
  a.wp = wp; a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = h->cfg.norm_eps; a.res = res; a.ldr = ldr;
  a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.epi = epi; a.wq = wq; a.scale = scale; a.wr = wr;
  a.bias = bias;
  if (bias) {   // only the skinny kernel has the bias epilogue: 16 rows per launch
    FMI_REQUIRE(epi == EPI_STORE && !norm_w && !scale, "linear: bias goes with a plain store epilogue");
    for (int m0 = 0; m0 < M; m0 += 16) {
      a.x = x + (int64_t)m0 * ldx; a.out = out + (int64_t)m0 * ldo; a.M = M - m0 < 16 ? M - m0 : 16;
      h->launches += 1;
      FMI_CHECK(launch_linear_skinny(a, s));
    }
    return FMI_OK;
  }
  // h->force_tiled: the few suffix rows of a resumed prefill must go through the kernel a full prefill of the
  // whole prompt would have used for them (the tiled GEMM: a row's bits do not depend on how many rows run along)
  // (17-32 rows take the decode GEMV only inside tail(): the merged fast positions 0/1 of a batch of 9-16.  Everywhere
  // else the 16-row boundary between the GEMV and the tiled GEMM stays where dual_ar.py's prefix reuse expects it.)
  if ((M <= 16 || (M <= 32 && h->skinny32)) && !h->force_tiled) {
    h->launches += 1;
    return launch_linear_skinny(a, s);
  }
  if (norm_w) {
    FMI_CHECK(launch_rmsnorm_rows(x, ldx, norm_w, h->cfg.norm_eps, h->ws.xn, K, M, K, s));
    a.x = h->ws.xn;
    a.ldx = K;
    a.norm_w = nullptr;
    h->launches += 1;
  }
  h->launches += 1;
  if (const int64_t need = linear_tiled_part_floats(M, N, K); need > 0) {
    if (h->gemm_part_floats < need) {
      FMI_CHECK_HIP(hipStreamSynchronize(s));
      if (h->gemm_part) hipFree(h->gemm_part);
      h->gemm_part = nullptr;
      h->gemm_part_floats = 0;
      FMI_CHECK_HIP(hipMalloc((void**)&h->gemm_part, (size_t)need * 4));
      h->gemm_part_floats = need;
    }
    a.part = h->gemm_part;
    h->launches += 1;
  }
  return launch_linear_tiled(a, s);
}

// one transformer block (llama.py:839-844) over `rows` rows of the residual stream x (in place)
int block_slow(fmi_dualar* h, const LayerW& w, int layer, bf16_t* x, int rows, const int32_t* row_slot,
               const int32_t* row_pos, hipStream_t s) {
  const Dims& d = h->slow;
  Workspace& ws = h->ws;
  FMI_CHECK(linear(h, x, d.dim, w.wqkv, w.attn_norm, nullptr, 0, ws.qkv, d.qkv, rows, d.qkv, d.dim, EPI_STORE, s, w.q_wqkv, w.s_wqkv, w.r_wqkv));
  AttnArgs a{};
  a.qkv = ws.qkv; a.q = ws.q; a.out = ws.ao; a.kpool = h->kpool[layer]; a.vpool = h->vpool[layer];
  a.qnw = h->cfg.attention_qk_norm ? w.q_norm : nullptr;
  a.knw = h->cfg.attention_qk_norm ? w.k_norm : nullptr;
  a.rope = h->rope; a.row_slot = row_slot; a.row_pos = row_pos; a.block_table = h->st.block_table;
  a.slot_pos = h->st.pos; a.slot_done = row_pos == nullptr ? h->st.done : nullptr; a.max_pages = h->max_pages; a.rows = rows; a.H = d.H; a.KVH = d.KVH; a.D = d.D;
  a.eps = h->cfg.norm_eps;
  if (row_pos == nullptr) {  // decode: one row per slot -> fused prep + attention, by the row's own context length
    a.part = h->attn_part;
    a.long_thr = (h->attn_mask & 2) ? h->attn_long_thr : 0;
    if (h->attn_mask & 1) {
      FMI_CHECK(launch_attn_decode_fused(a, s));
      h->launches += 1;
    }
    if (h->attn_mask & 2) {
      FMI_CHECK(launch_attn_decode_long(a, s));
      h->launches += 2;
    }
  } else {
    FMI_CHECK(launch_attn_prep(a, s));
    a.qtiles = ws.qtiles; a.n_qtiles = ws.n_qtiles; a.qtile_rows = ws.qtile_rows;
    if (h->attn_impl == 1 && ws.n_qtiles > 0) FMI_CHECK(launch_attn_prefill_mfma(a, s));
    else FMI_CHECK(launch_attn(a, s));
    h->launches += 2;
  }
  FMI_CHECK(linear(h, ws.ao, d.H * d.D, w.wo, nullptr, x, d.dim, x, d.dim, rows, d.dim, d.H * d.D, EPI_RESIDUAL, s, w.q_wo, w.s_wo, w.r_wo));
  FMI_CHECK(linear(h, x, d.dim, w.w13, w.ffn_norm, nullptr, 0, ws.act, d.ffn, rows, 2 * d.ffn, d.dim, EPI_SILU, s, w.q_w13, w.s_w13));
  FMI_CHECK(linear(h, ws.act, d.ffn, w.w2, nullptr, x, d.dim, x, d.dim, rows, d.dim, d.ffn, EPI_RESIDUAL, s, w.q_w2, w.s_w2, w.r_w2));
  return FMI_OK;
}

// merge01 (tail): x holds 2 B rows -- rows [0, B) the B utterances at fast position 0, rows [B, 2B) the same utterances
// at position 1 -- and ONE pass over the layer's weights serves both (forward_generate_fast twice, inference.py:148-166:
// both inputs are known once the slow token is drawn, and position 1 attends only to itself and to position 0's K/V of
// the same layer).  qkv_rows1 != nullptr: the position-1 q|k|v rows are already in ws.qkv rows [B, 2B) (the tabulated
// first layer), only the position-0 rows go through the GEMV.  last_rows1: the layer's output feeds nothing at
// position 0 (its logits are discarded), so wo / FFN run on the position-1 rows only.
int block_fast(fmi_dualar* h, const LayerW& w, int layer, bf16_t* x, int B, int pos, const int32_t* row_slot,
               hipStream_t s, bool kv_only = false, const bf16_t* qkv_pre = nullptr, bool merge01 = false,
               bool qkv_rows1 = false, bool last_rows1 = false) {
  const Dims& d = h->fast;
  Workspace& ws = h->ws;
  const int M = merge01 ? 2 * B : B;
  if (!qkv_pre)
    FMI_CHECK(linear(h, x, d.dim, w.wqkv, w.attn_norm, nullptr, 0, ws.qkv, d.qkv, qkv_rows1 ? B : M, d.qkv, d.dim, EPI_STORE, s, w.q_wqkv, w.s_wqkv, w.r_wqkv));
  FastAttnArgs a{};
  a.qkv = qkv_pre ? qkv_pre : ws.qkv; a.out = ws.ao; a.kc = h->fkc[layer]; a.vc = h->fvc[layer];
  a.qnw = h->cfg.fast_attention_qk_norm ? w.q_norm : nullptr;
  a.knw = h->cfg.fast_attention_qk_norm ? w.k_norm : nullptr;
  a.rope = h->fast_rope; a.row_slot = row_slot; a.B = B; a.H = d.H; a.KVH = d.KVH; a.D = d.D;
  a.ncb = h->cfg.num_codebooks; a.pos = pos; a.eps = h->cfg.norm_eps; a.merge = merge01 ? 1 : 0;
  FMI_CHECK(launch_fast_attn(a, s));
  h->launches += 1;
  if (kv_only) return FMI_OK;  // only this layer's K/V at `pos` were needed
  if (merge01) {
    const int r0 = last_rows1 ? B : 0, m = M - r0;
    bf16_t* xr = x + (int64_t)r0 * d.dim;
    const bf16_t* ao = ws.ao + (int64_t)r0 * d.H * d.D;
    FMI_CHECK(linear(h, ao, d.H * d.D, w.wo, nullptr, xr, d.dim, xr, d.dim, m, d.dim, d.H * d.D, EPI_RESIDUAL, s, w.q_wo, w.s_wo, w.r_wo));
    FMI_CHECK(linear(h, xr, d.dim, w.w13, w.ffn_norm, nullptr, 0, ws.act, d.ffn, m, 2 * d.ffn, d.dim, EPI_SILU, s, w.q_w13, w.s_w13));
    FMI_CHECK(linear(h, ws.act, d.ffn, w.w2, nullptr, xr, d.dim, xr, d.dim, m, d.dim, d.ffn, EPI_RESIDUAL, s, w.q_w2, w.s_w2, w.r_w2));
    return FMI_OK;
  }
  FMI_CHECK(linear(h, ws.ao, d.H * d.D, w.wo, nullptr, x, d.dim, x, d.dim, B, d.dim, d.H * d.D, EPI_RESIDUAL, s, w.q_wo, w.s_wo, w.r_wo));
  FMI_CHECK(linear(h, x, d.dim, w.w13, w.ffn_norm, nullptr, 0, ws.act, d.ffn, B, 2 * d.ffn, d.dim, EPI_SILU, s, w.q_w13, w.s_w13));
  FMI_CHECK(linear(h, ws.act, d.ffn, w.w2, nullptr, x, d.dim, x, d.dim, B, d.dim, d.ffn, EPI_RESIDUAL, s, w.q_w2, w.s_w2, w.r_w2));
  return FMI_OK;
}

// everything after the slow transformer for B utterances whose last hidden rows are xl[B][dim]:
// final norm, restricted tied head, constrained sampling + RAS, the fast-AR chain
// (decode_one_token_ar, inference.py:108-181).
// final norm + restricted tied head (llama.py:450-457): hn = normed hidden, logits over the live rows
int tail_head(fmi_dualar* h, const bf16_t* xl, int B, hipStream_t s, bf16_t* hf_out = nullptr) {
  const fmi_dualar_config& c = h->cfg;
  const int dim = c.dim;
  // hn = normed hidden (head input, parity tap); hf = its copy that fast step 0 transforms in place
  if (h->tail_in_normed) {   // test seam: xl IS the normed hidden (what forward_generate returns, llama.py:459-461)
    FMI_CHECK_HIP(hipMemcpyAsync(h->hn, xl, (size_t)B * dim * 2, hipMemcpyDeviceToDevice, s));
    FMI_CHECK_HIP(hipMemcpyAsync(hf_out ? hf_out : h->hf, xl, (size_t)B * dim * 2, hipMemcpyDeviceToDevice, s));
  } else {
    FMI_CHECK(launch_rmsnorm_rows(xl, dim, h->norm, c.norm_eps, h->hn, dim, B, dim, s, hf_out ? hf_out : h->hf));
  }
  h->launches += 1;
  return linear(h, h->hn, dim, h->head_live, nullptr, nullptr, 0, h->logits, h->n_live_pad, B, h->n_live_pad, dim,
                EPI_STORE, s);
}

// llama.py:827: hidden_states = fast_project_in(hidden_states) -- Linear(dim, fast_dim) with bias, only when fast_dim != dim
int project_fast_in(fmi_dualar* h, const bf16_t* hid, bf16_t* out, int B, hipStream_t s) {
  const fmi_dualar_config& c = h->cfg;
  return linear(h, hid, c.dim, h->fpi_w, nullptr, nullptr, 0, out, c.fast_dim, B, c.fast_dim, c.dim, EPI_STORE, s, nullptr,
                nullptr, nullptr, h->fpi_b);
}

int tail(fmi_dualar* h, const bf16_t* xl, int B, const int32_t* row_slot, hipStream_t s) {
  const fmi_dualar_config& c = h->cfg;
  const int dim = c.dim;
  // positions 0 and 1 of the fast transformer in one pass over its weights (block_fast: merge01): 2 B <= 16 rows of
  // the decode GEMV; a row's bits do not depend on the rows it travels with, so nothing changes but the traffic
  // (-0.6 GB per frame at the S2 shape) and the launch count (-16)
  const bool merge = h->merge01 && !h->fpi_w && !c.weight_int8 && 2 * B <= 32 && c.num_codebooks >= 2 && c.fast_dim == c.dim &&
                     h->ws.rows >= 2 * B;
  struct Skinny32 {   // the merged pass of a batch of 9-16 runs 18-32 rows through the GEMV's two-column-set form
    fmi_dualar* h;
    explicit Skinny32(fmi_dualar* h_, bool on) : h(h_) { h->skinny32 = on; }
    ~Skinny32() { h->skinny32 = false; }
  } skinny32_guard(h, merge);
  bf16_t* const x0 = merge ? h->x01 : h->hf;                                   // position-0 rows
  bf16_t* const xf = merge ? h->x01 + (int64_t)B * c.fast_dim : h->xf;         // rows of positions >= 1
  FMI_CHECK(tail_head(h, xl, B, s, x0));
  SampleArgs sa{};
  sa.logits = h->logits; sa.B = B; sa.n = h->n_live; sa.ld = h->n_live_pad; sa.ids = h->live_ids;
  sa.row_slot = row_slot; sa.st = h->st; sa.mode = 0; sa.cb = 0; sa.sem_begin = c.semantic_begin_id;
  sa.sem_end = c.semantic_end_id; sa.im_end = h->ignore_eos ? -1 : c.im_end_id; sa.cbs = c.codebook_size; sa.fast_emb = h->fast_emb;
  sa.xf = xf; sa.fdim = c.fast_dim; sa.small_k = h->max_top_k <= 64; sa.forced = h->forced;
  const bool tab = h->qkv0_tab != nullptr && h->trace != 1;
  sa.qkv0_tab = tab ? h->qkv0_tab : nullptr; sa.qkv0_dim = h->fast.qkv;
  // merged pass: the tabulated q|k|v rows of the slow token's codes land where the GEMV would have put them
  sa.qkv0_out = merge ? h->ws.qkv + (int64_t)B * h->fast.qkv : h->qkv0_pre;
  FMI_CHECK(launch_sample(sa, s));
  h->launches += 1;
  sa.qkv0_out = h->qkv0_pre;
  // fast step 0 on the hidden state; its logits are discarded (inference.py:148-149)
  bf16_t* f0 = x0;
  if (h->fpi_w) {        // fast_dim != dim: step 0 runs on the projected hidden rows
    FMI_CHECK(project_fast_in(h, c.norm_fastlayer_input ? h->hn : xl, h->hfp, B, s));
    f0 = h->hfp + (int64_t)h->max_batch * c.fast_dim;   // step 0 transforms its input in place: keep the tap
    FMI_CHECK_HIP(hipMemcpyAsync(f0, h->hfp, (size_t)B * c.fast_dim * 2, hipMemcpyDeviceToDevice, s));
  } else if (!c.norm_fastlayer_input) {
    FMI_CHECK_HIP(hipMemcpyAsync(x0, xl, (size_t)B * dim * 2, hipMemcpyDeviceToDevice, s));
  }
  // Fast step 0 exists only to put the hidden state's K/V into slot 0 of every fast layer: its logits are
  // discarded (inference.py:148-149), so the last layer's wo / FFN output feeds nothing and is skipped.
  if (!merge)
    for (int i = 0; i < c.n_fast_layer; ++i)
      FMI_CHECK(block_fast(h, h->FL[i], i, f0, B, 0, row_slot, s, i == c.n_fast_layer - 1));
  for (int cb = 1; cb < c.num_codebooks; ++cb) {
    for (int i = 0; i < c.n_fast_layer; ++i) {
      if (merge && cb == 1)
        FMI_CHECK(block_fast(h, h->FL[i], i, h->x01, B, 0, row_slot, s, false, nullptr, true, i == 0 && tab, i == c.n_fast_layer - 1));
      else
        FMI_CHECK(block_fast(h, h->FL[i], i, xf, B, cb, row_slot, s, false, (i == 0 && tab) ? h->qkv0_pre : nullptr));
    }
    FMI_CHECK(linear(h, xf, c.fast_dim, h->fast_out, h->fast_norm, nullptr, 0, h->flogits, c.codebook_size, B,
                     c.codebook_size, c.fast_dim, EPI_STORE, s, h->q_fast_out, h->s_fast_out));
    if (h->trace) {
      FMI_CHECK_HIP(hipMemcpy2DAsync(h->ftrace + (int64_t)cb * c.codebook_size,
                                     (size_t)c.num_codebooks * c.codebook_size * 2, h->flogits,
                                     (size_t)c.codebook_size * 2, (size_t)c.codebook_size * 2, B,
                                     hipMemcpyDeviceToDevice, s));
    }
    SampleArgs fa = sa;
    fa.logits = h->flogits; fa.n = c.codebook_size; fa.ld = c.codebook_size; fa.ids = nullptr; fa.mode = 1;
    fa.cb = cb;
    FMI_CHECK(launch_sample(fa, s));
    h->launches += 1;
  }
  return FMI_OK;
}

// one decode frame for the B slots listed in ws.row_slot
int decode_frame(fmi_dualar* h, int B, hipStream_t s, bool head_only = false) {
  const fmi_dualar_config& c = h->cfg;
  h->launches = 0;
  EmbedArgs e{};
  e.emb = h->emb; e.cb_emb = h->cb_emb; e.tokens = h->st.cur; e.row_slot = h->ws.row_slot; e.out = h->ws.x;
  e.rows = B; e.dim = c.dim; e.ncb = c.num_codebooks; e.cbs = c.codebook_size; e.sem_begin = c.semantic_begin_id;
  e.sem_end = c.semantic_end_id; e.scale = c.scale_codebook_embeddings;
  FMI_CHECK(launch_embed(e, s));
  h->launches += 1;
  for (int i = 0; i < c.n_layer; ++i)
    FMI_CHECK(block_slow(h, h->L[i], i, h->ws.x, B, h->ws.row_slot, nullptr, s));
  if (head_only) return tail_head(h, h->ws.x, B, s);
  FMI_CHECK(tail(h, h->ws.x, B, h->ws.row_slot, s));
  h->frame_launches = h->launches;
  return FMI_OK;
}

// Row-balanced decode copies of the projection matrices whose 16-row tiling leaves CUs idle at the decode shapes
// (dualar_kernels.h: skinny_row_plan): derived from the packed arena once the weights are in place, on every rank
// (they are a permutation of arena bytes, so they do not travel with the broadcast).  FMI_NO_ROWS=1 keeps the 16-row
// tiles (A/B runs).  Costs one extra copy of the layer matrices in HBM (4.1 GB at the S2-Pro shape, of 288 GB).
int ensure_row_copies(fmi_dualar* h) {
  if (h->rows_tried || !h->ready) return FMI_OK;
  h->rows_tried = true;
  static const bool off = []() { const char* e = getenv("FMI_NO_ROWS"); return e && atoi(e) != 0; }();
  if (off || h->cfg.weight_int8) return FMI_OK;
  hipStream_t s = h->stream;
  auto derive = [&](const bf16_t* packed, bf16_t** dst, int N, int K, int epi, bool norm) -> int {
    if (!skinny_rows_supported(N, K, epi, norm)) return FMI_OK;
    const RowPlan p = skinny_row_plan(N, K, epi);
    void* mem = nullptr;
    FMI_CHECK_HIP(hipMalloc(&mem, (size_t)p.elems * 2));
    h->row_copies.push_back(mem);
    FMI_CHECK(launch_repack_rows(packed, (bf16_t*)mem, N, K, epi, p, s));
    *dst = (bf16_t*)mem;
    return FMI_OK;
  };
  auto layer = [&](LayerW& w, const Dims& d) -> int {
    FMI_CHECK(derive(w.wqkv, &w.r_wqkv, d.qkv, d.dim, EPI_STORE, true));
    FMI_CHECK(derive(w.wo, &w.r_wo, d.dim, d.H * d.D, EPI_RESIDUAL, false));
    return derive(w.w2, &w.r_w2, d.dim, d.ffn, EPI_RESIDUAL, false);
  };
  for (auto& w : h->L) FMI_CHECK(layer(w, h->slow));
  for (auto& w : h->FL) FMI_CHECK(layer(w, h->fast));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  drop_graphs(h);
  return FMI_OK;
}

// Tabulate fast layer 0's wqkv(rmsnorm(fast_embeddings[code])) for every code, 8 codes per launch of the decode
// GEMV itself (its row results do not depend on the batch they are computed in, tests/test_dualar_gpu.py).
int ensure_qkv0_table(fmi_dualar* h) {
  FMI_CHECK(ensure_row_copies(h));
  if (h->qkv0_tried || !h->ready || h->max_batch == 0) return FMI_OK;
  h->qkv0_tried = true;
  static const bool off = []() { const char* e = getenv("FMI_NO_QKV0"); return e && atoi(e) != 0; }();
  const fmi_dualar_config& c = h->cfg;
  if (off || c.n_fast_layer < 1 || c.codebook_size % 8 != 0) return FMI_OK;
  const Dims& d = h->fast;
  const LayerW& w = h->FL[0];
  FMI_CHECK(dev_alloc(&h->qkv0_tab, (int64_t)c.codebook_size * d.qkv));
  FMI_CHECK(dev_alloc(&h->qkv0_pre, (int64_t)h->max_batch * d.qkv));
  FMI_CHECK_HIP(hipDeviceSynchronize());   // dev_alloc clears on the null stream; h->stream does not wait for it
  hipStream_t s = h->stream;
  const int saved = h->launches;
  for (int code = 0; code < c.codebook_size; code += 8)
    FMI_CHECK(linear(h, h->fast_emb + (int64_t)code * d.dim, d.dim, w.wqkv, w.attn_norm, nullptr, 0,
                     h->qkv0_tab + (int64_t)code * d.qkv, d.qkv, 8, d.qkv, d.dim, EPI_STORE, s, w.q_wqkv, w.s_wqkv, w.r_wqkv));
  h->launches = saved;
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  drop_graphs(h);
  return FMI_OK;
}

int reserve_pages(fmi_dualar* h, int slot, int upto_pos_exclusive) {
  const int need = cdiv(std::max(upto_pos_exclusive, 1), KV_PAGE);
  FMI_REQUIRE(need <= h->max_pages, "slot %d needs %d pages > max %d (raise max_seq_len)", slot, need, h->max_pages);
  std::vector<int>& pg = h->slot_pages[slot];
  bool changed = false;
  while ((int)pg.size() < need) {
    if (h->free_pages.empty()) return set_error(FMI_ENOMEM, "KV page pool exhausted");
    pg.push_back(h->free_pages.back());
    h->free_pages.pop_back();
    changed = true;
  }
  if (changed)
    FMI_CHECK_HIP(hipMemcpyAsync(h->st.block_table + (int64_t)slot * h->max_pages, pg.data(), pg.size() * 4,
                                 hipMemcpyHostToDevice, h->stream));
  return FMI_OK;
}

int set_slot(fmi_dualar* h, int slot, int pos, int frame, int limit, const fmi_sampling& sp, bool zero_window) {
  hipStream_t s = h->stream;
  h->pos_host[slot] = pos;
  h->slot_top_k[slot] = (int)sp.top_k;
  int mk = 0;
  for (int k : h->slot_top_k) mk = std::max(mk, k);
  if ((mk > 64) != (h->max_top_k > 64)) {  // the captured graphs embed the sampler variant
    FMI_CHECK_HIP(hipStreamSynchronize(s));
    drop_graphs(h);
  }
  h->max_top_k = mk;
  const int32_t zero = 0;
  const float t = rbf(sp.temperature), p = rbf(sp.top_p);
#define PUT(arr, val) FMI_CHECK_HIP(hipMemcpyAsync((arr) + slot, &(val), 4, hipMemcpyHostToDevice, s))
  PUT(h->st.pos, pos);
  PUT(h->st.frame, frame);
  PUT(h->st.done, zero);
  PUT(h->st.limit, limit);
  PUT(h->st.temperature, t);
  PUT(h->st.top_p, p);
  PUT(h->st.top_k, sp.top_k);
  PUT(h->st.seed, sp.seed);
  PUT(h->st.use_ras, sp.use_ras);
#undef PUT
  if (zero_window)
    FMI_CHECK_HIP(hipMemsetAsync(h->st.window + (int64_t)slot * h->st.ncb1 * RAS_WIN, 0,
                                 (size_t)h->st.ncb1 * RAS_WIN * 4, s));
  return FMI_OK;
}

int check_slot(fmi_dualar* h, int slot) {
  FMI_REQUIRE(h->max_batch > 0, "setup_caches has not been called");
  FMI_REQUIRE(slot >= 0 && slot < h->max_batch, "slot %d out of range [0,%d)", slot, h->max_batch);
  return FMI_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI

extern "C" {

int fmi_version(void) { return 1; }
const char* fmi_last_error(void) { return g_last_error.c_str(); }

int fmi_device_arch(char* buf, size_t n) {
  int dev = 0;
  FMI_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  FMI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  snprintf(buf, n, "%s", p.gcnArchName);
  return FMI_OK;
}

int64_t fmi_dualar_arena_bytes(const fmi_dualar_config* cfg) {
  if (!cfg || validate(*cfg) != FMI_OK) return -1;
  return layout(*cfg, nullptr);
}

int fmi_dualar_create(const fmi_dualar_config* cfg, void* arena_dev, int64_t arena_bytes, fmi_dualar** out) {
  FMI_REQUIRE(cfg && arena_dev && out, "null argument");
  FMI_CHECK(validate(*cfg));
  const int64_t need = layout(*cfg, nullptr);
  FMI_REQUIRE(arena_bytes >= need, "arena too small: %lld < %lld", (long long)arena_bytes, (long long)need);
  FMI_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "arena must be 256-byte aligned");
  fmi_dualar* h = new fmi_dualar();
  h->cfg = *cfg;
  h->slow = slow_dims(*cfg);
  h->fast = fast_dims(*cfg);
  h->arena = (char*)arena_dev;
  h->arena_bytes = arena_bytes;
  layout(*cfg, h);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&h->ev_t0) != hipSuccess || hipEventCreate(&h->ev_t1) != hipSuccess) {
    delete h;
    return set_error(FMI_EHIP, "stream/event creation failed (no GPU?)");
  }
  *out = h;
  return FMI_OK;
}

void fmi_dualar_destroy(fmi_dualar* h) {
  if (!h) return;
  hipStreamSynchronize(h->stream);
  drop_graphs(h);
  free_ws(h->ws);
  for (auto p : h->kpool) hipFree(p);
  for (auto p : h->vpool) hipFree(p);
  for (auto p : h->fkc) hipFree(p);
  for (auto p : h->fvc) hipFree(p);
  void* ptrs[] = {h->st.pos, h->st.frame, h->st.done, h->st.limit, h->st.cur, h->st.window, h->st.out,
                  h->st.temperature, h->st.top_p, h->st.top_k, h->st.seed, h->st.use_ras, h->st.block_table,
                  h->hn, h->hf, h->xl, h->xf, h->logits, h->flogits, h->ftrace, h->staging, h->staging2,
                  h->qkv0_tab, h->qkv0_pre, h->attn_part, h->hfp, h->x01, h->gemm_part};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (void* p : h->row_copies) hipFree(p);
  hipEventDestroy(h->ev_in);
  hipEventDestroy(h->ev_out);
  hipEventDestroy(h->ev_t0);
  hipEventDestroy(h->ev_t1);
  hipStreamDestroy(h->stream);
  delete h;
}

int fmi_dualar_load_tensor(fmi_dualar* h, const char* name_c, const void* src, int64_t rows, int64_t cols,
                           int src_is_device, void* stream) {
  FMI_REQUIRE(h && name_c && src, "null argument");
  const std::string name(name_c);
  const fmi_dualar_config& c = h->cfg;
  FMI_CHECK(invalidate_derived(h));
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  const bf16_t* dsrc = (const bf16_t*)src;
  if (!src_is_device) {
    FMI_CHECK(ensure_staging(h, (size_t)(rows * cols * 2)));
    FMI_CHECK_HIP(hipMemcpyAsync(h->staging, src, (size_t)(rows * cols * 2), hipMemcpyHostToDevice, s));
    dsrc = (const bf16_t*)h->staging;
  }
  auto expect = [&](int64_t r, int64_t cc) -> int {
    if (rows != r || cols != cc)
      return set_error(FMI_EINVAL, "%s: shape (%lld,%lld) != expected (%lld,%lld)", name.c_str(), (long long)rows,
                       (long long)cols, (long long)r, (long long)cc);
    return FMI_OK;
  };
  auto copy = [&](bf16_t* dst) -> int {
    FMI_CHECK_HIP(hipMemcpyAsync(dst, dsrc, (size_t)(rows * cols * 2), hipMemcpyDeviceToDevice, s));
    return FMI_OK;
  };
  int rc = FMI_OK;
  if (name == "embeddings.weight") { FMI_CHECK(expect(c.vocab_size, c.dim)); rc = copy(h->emb); }
  else if (name == "codebook_embeddings.weight") { FMI_CHECK(expect((int64_t)c.codebook_size * c.num_codebooks, c.dim)); rc = copy(h->cb_emb); }
  else if (name == "norm.weight") { FMI_CHECK(expect(1, c.dim)); rc = copy(h->norm); }
  else if (name == "fast_embeddings.weight") { FMI_CHECK(expect(c.codebook_size, c.fast_dim)); rc = copy(h->fast_emb); }
  else if (name == "fast_norm.weight") { FMI_CHECK(expect(1, c.fast_dim)); rc = copy(h->fast_norm); }
  else if (name == "fast_output.weight") { FMI_CHECK(expect(c.codebook_size, c.fast_dim)); rc = launch_pack_weight(dsrc, h->fast_out, (int)rows, (int)cols, 0, s); }
  else if (name == "fast_project_in.weight" && h->fpi_w) { FMI_CHECK(expect(c.fast_dim, c.dim)); rc = launch_pack_weight(dsrc, h->fpi_w, (int)rows, (int)cols, 0, s); }
  else if (name == "fast_project_in.bias" && h->fpi_b) { FMI_CHECK(expect(1, c.fast_dim)); rc = copy(h->fpi_b); }
  else if (name == "freqs_cis") { FMI_CHECK(expect(c.max_seq_len, c.head_dim)); rc = copy(h->rope); h->rope_loaded = true; }
  else if (name == "fast_freqs_cis") { FMI_CHECK(expect(c.num_codebooks, c.fast_head_dim)); rc = copy(h->fast_rope); h->fast_rope_loaded = true; }
  else {
    const bool fastl = name.rfind("fast_layers.", 0) == 0;
    const bool slowl = name.rfind("layers.", 0) == 0;
    if (!fastl && !slowl) return set_error(FMI_EINVAL, "unknown tensor name '%s'", name.c_str());
    const size_t p0 = fastl ? 12 : 7;
    const size_t dot = name.find('.', p0);
    if (dot == std::string::npos) return set_error(FMI_EINVAL, "bad tensor name '%s'", name.c_str());
    const int idx = atoi(name.substr(p0, dot - p0).c_str());
    const std::string sub = name.substr(dot + 1);
    const Dims& d = fastl ? h->fast : h->slow;
    std::vector<LayerW>& LL = fastl ? h->FL : h->L;
    if (idx < 0 || idx >= (int)LL.size()) return set_error(FMI_EINVAL, "layer index out of range in '%s'", name.c_str());
    LayerW& w = LL[idx];
    if (sub == "attention.wqkv.weight") { FMI_CHECK(expect(d.qkv, d.dim)); rc = launch_pack_weight(dsrc, w.wqkv, d.qkv, d.dim, 0, s); }
    else if (sub == "attention.wo.weight") { FMI_CHECK(expect(d.dim, d.H * d.D)); rc = launch_pack_weight(dsrc, w.wo, d.dim, d.H * d.D, 0, s); }
    else if (sub == "feed_forward.w1.weight") { FMI_CHECK(expect(d.ffn, d.dim)); rc = launch_pack_weight(dsrc, w.w13, d.ffn, d.dim, 1, s); }
    else if (sub == "feed_forward.w3.weight") { FMI_CHECK(expect(d.ffn, d.dim)); rc = launch_pack_weight(dsrc, w.w13, d.ffn, d.dim, 2, s); }
    else if (sub == "feed_forward.w2.weight") { FMI_CHECK(expect(d.dim, d.ffn)); rc = launch_pack_weight(dsrc, w.w2, d.dim, d.ffn, 0, s); }
    else if (sub == "attention_norm.weight") { FMI_CHECK(expect(1, d.dim)); rc = copy(w.attn_norm); }
    else if (sub == "ffn_norm.weight") { FMI_CHECK(expect(1, d.dim)); rc = copy(w.ffn_norm); }
    else if (sub == "attention.q_norm.weight") { FMI_CHECK(expect(1, d.D)); rc = copy(w.q_norm); }
    else if (sub == "attention.k_norm.weight") { FMI_CHECK(expect(1, d.D)); rc = copy(w.k_norm); }
    else return set_error(FMI_EINVAL, "unknown tensor name '%s'", name.c_str());
  }
  FMI_CHECK(rc);
  if (!src_is_device) FMI_CHECK_HIP(hipStreamSynchronize(s));  // staging buffer is reused
  h->loaded.insert(name);
  return sync_out(h, stream);
}

int fmi_dualar_load_tensor_int8(fmi_dualar* h, const char* name_c, const void* weight_i8, const void* scales_bf16,
                                int64_t rows, int64_t cols, int src_is_device, void* stream) {
  FMI_REQUIRE(h && name_c && weight_i8 && scales_bf16, "null argument");
  FMI_REQUIRE(h->cfg.weight_int8, "the handle was not created with weight_int8=1");
  FMI_REQUIRE(cols % 64 == 0, "int8 linears need K %% 64 == 0 (got %lld)", (long long)cols);
  const std::string name(name_c);
  const fmi_dualar_config& c = h->cfg;
  FMI_CHECK(invalidate_derived(h));
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  const int64_t n = rows * cols;
  // device copies of the int8 weight and its scales, and the exact bf16 dequantisation (row-major scratch)
  const size_t need = (size_t)n + (size_t)rows * 2 + 256;
  FMI_CHECK(ensure_staging(h, need));
  int8_t* dq = (int8_t*)h->staging;
  bf16_t* dscale = (bf16_t*)((char*)h->staging + align_up(n, 256));
  const hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  FMI_CHECK_HIP(hipMemcpyAsync(dq, weight_i8, (size_t)n, kind, s));
  FMI_CHECK_HIP(hipMemcpyAsync(dscale, scales_bf16, (size_t)rows * 2, kind, s));
  if (h->staging2_bytes < (size_t)n * 2) {
    if (h->staging2) FMI_CHECK_HIP(hipFree(h->staging2));
    h->staging2 = nullptr;
    h->staging2_bytes = 0;
    FMI_CHECK_HIP(hipMalloc(&h->staging2, (size_t)n * 2));
    h->staging2_bytes = (size_t)n * 2;
  }
  bf16_t* deq = (bf16_t*)h->staging2;
  FMI_CHECK(launch_dequant_int8(dq, deq, n, s));
  auto expect = [&](int64_t r, int64_t cc) -> int {
    if (rows != r || cols != cc)
      return set_error(FMI_EINVAL, "%s: shape (%lld,%lld) != expected (%lld,%lld)", name.c_str(), (long long)rows,
                       (long long)cols, (long long)r, (long long)cc);
    return FMI_OK;
  };
  auto put = [&](bf16_t* wp, int8_t* wq, bf16_t* sc, int interleave) -> int {
    FMI_CHECK(launch_pack_weight(deq, wp, (int)rows, (int)cols, interleave, s));
    FMI_CHECK(launch_pack_weight_int8(dq, wq, (int)rows, (int)cols, interleave, s));
    return launch_pack_scale(dscale, sc, (int)rows, interleave, s);
  };
  int rc;
  if (name == "fast_output.weight") {
    FMI_CHECK(expect(c.codebook_size, c.fast_dim));
    rc = put(h->fast_out, h->q_fast_out, h->s_fast_out, 0);
  } else {
    const bool fastl = name.rfind("fast_layers.", 0) == 0;
    const bool slowl = name.rfind("layers.", 0) == 0;
    if (!fastl && !slowl) return set_error(FMI_EINVAL, "'%s' is not a quantised linear", name.c_str());
    const size_t p0 = fastl ? 12 : 7;
    const size_t dot = name.find('.', p0);
    if (dot == std::string::npos) return set_error(FMI_EINVAL, "bad tensor name '%s'", name.c_str());
    const int idx = atoi(name.substr(p0, dot - p0).c_str());
    const std::string sub = name.substr(dot + 1);
    const Dims& d = fastl ? h->fast : h->slow;
    std::vector<LayerW>& LL = fastl ? h->FL : h->L;
    if (idx < 0 || idx >= (int)LL.size()) return set_error(FMI_EINVAL, "layer index out of range in '%s'", name.c_str());
    LayerW& w = LL[idx];
    if (sub == "attention.wqkv.weight") { FMI_CHECK(expect(d.qkv, d.dim)); rc = put(w.wqkv, w.q_wqkv, w.s_wqkv, 0); }
    else if (sub == "attention.wo.weight") { FMI_CHECK(expect(d.dim, d.H * d.D)); rc = put(w.wo, w.q_wo, w.s_wo, 0); }
    else if (sub == "feed_forward.w1.weight") { FMI_CHECK(expect(d.ffn, d.dim)); rc = put(w.w13, w.q_w13, w.s_w13, 1); }
    else if (sub == "feed_forward.w3.weight") { FMI_CHECK(expect(d.ffn, d.dim)); rc = put(w.w13, w.q_w13, w.s_w13, 2); }
    else if (sub == "feed_forward.w2.weight") { FMI_CHECK(expect(d.dim, d.ffn)); rc = put(w.w2, w.q_w2, w.s_w2, 0); }
    else return set_error(FMI_EINVAL, "'%s' is not a quantised linear", name.c_str());
  }
  FMI_CHECK(rc);
  FMI_CHECK_HIP(hipStreamSynchronize(s));  // staging buffers are reused
  h->loaded.insert(name);
  return sync_out(h, stream);
}

int fmi_dualar_finalize_weights(fmi_dualar* h, void* stream) {
  FMI_REQUIRE(h, "null handle");
  const fmi_dualar_config& c = h->cfg;
  // completeness check
  std::vector<std::string> need = {"embeddings.weight", "codebook_embeddings.weight", "norm.weight",
                                   "fast_embeddings.weight", "fast_norm.weight", "fast_output.weight"};
  if (c.fast_dim != c.dim) {
    need.push_back("fast_project_in.weight");
    need.push_back("fast_project_in.bias");
  }
  auto add_layer = [&](const std::string& pre, bool qk) {
    for (const char* sfx : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight",
                            "feed_forward.w3.weight", "feed_forward.w2.weight", "attention_norm.weight",
                            "ffn_norm.weight"})
      need.push_back(pre + sfx);
    if (qk) {
      need.push_back(pre + "attention.q_norm.weight");
      need.push_back(pre + "attention.k_norm.weight");
    }
  };
  for (int i = 0; i < c.n_layer; ++i) add_layer("layers." + std::to_string(i) + ".", c.attention_qk_norm);
  for (int i = 0; i < c.n_fast_layer; ++i) add_layer("fast_layers." + std::to_string(i) + ".", c.fast_attention_qk_norm);
  for (const auto& n : need)
    if (!h->loaded.count(n)) return set_error(FMI_ESTATE, "tensor '%s' was never loaded", n.c_str());

  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  // live LM-head rows in ascending vocab order (ties resolve to the lowest vocab id)
  std::vector<int32_t> ids;
  const bool im_inside = c.im_end_id >= c.semantic_begin_id && c.im_end_id <= c.semantic_end_id;
  if (!im_inside && c.im_end_id < c.semantic_begin_id) ids.push_back(c.im_end_id);
  for (int v = c.semantic_begin_id; v <= c.semantic_end_id; ++v) ids.push_back(v);
  if (!im_inside && c.im_end_id > c.semantic_end_id) ids.push_back(c.im_end_id);
  ids.resize(h->n_live_pad, 0);
  FMI_CHECK_HIP(hipMemcpyAsync(h->live_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, s));
  FMI_CHECK(launch_pack_rows_gather(h->emb, h->live_ids, h->head_live, h->n_live, h->n_live_pad, c.dim, s));
  // RoPE tables (llama.py:1004-1023) unless the host supplied torch-built ones
  auto build_rope = [&](bf16_t* dst, int seq, int D) -> int {
    std::vector<bf16_t> tab((size_t)seq * D);
    for (int k = 0; k < D / 2; ++k) {
      const float freq = 1.0f / powf(c.rope_base, (float)(2 * k) / (float)D);
      for (int p = 0; p < seq; ++p) {
        const float ang = (float)p * freq;
        tab[((size_t)p * (D / 2) + k) * 2 + 0] = f2bf(cosf(ang));
        tab[((size_t)p * (D / 2) + k) * 2 + 1] = f2bf(sinf(ang));
      }
    }
    FMI_CHECK_HIP(hipMemcpyAsync(dst, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, s));
    FMI_CHECK_HIP(hipStreamSynchronize(s));
    return FMI_OK;
  };
  if (!h->rope_loaded) FMI_CHECK(build_rope(h->rope, c.max_seq_len, c.head_dim));
  if (!h->fast_rope_loaded) FMI_CHECK(build_rope(h->fast_rope, c.num_codebooks, c.fast_head_dim));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  h->ready = true;
  return sync_out(h, stream);
}

// The arena content was put in place by work enqueued on `stream` (a broadcast, a copy).  Everything this handle derives
// from it -- the first prefill rebuilds the row-balanced copies and the fast layer-0 q|k|v table FROM the arena -- runs
// on the handle's private non-blocking stream, which nothing else orders after that work: make it wait here.
int fmi_dualar_weights_ready(fmi_dualar* h, void* stream) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK(sync_in(h, stream));
  h->ready = true;
  return FMI_OK;
}

int fmi_dualar_setup_caches(fmi_dualar* h, int max_batch, int max_seq_len) {
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(max_batch >= 1 && max_seq_len >= 1, "bad cache size");
  FMI_REQUIRE(max_seq_len <= h->cfg.max_seq_len, "max_seq_len %d exceeds config.max_seq_len %d (RoPE table)",
              max_seq_len, h->cfg.max_seq_len);
  if (h->max_batch >= max_batch && h->max_seq >= max_seq_len) return FMI_OK;  // llama.py:310-311
  FMI_REQUIRE(h->max_batch == 0, "caches can only be set up once per handle");
  const fmi_dualar_config& c = h->cfg;
  const Dims &s = h->slow, &f = h->fast;
  h->max_batch = max_batch;
  h->max_seq = max_seq_len;
  h->max_pages = cdiv(max_seq_len, KV_PAGE);
  h->n_pages = h->max_pages * max_batch;
  h->max_frames = max_seq_len;
  const int ncb1 = c.num_codebooks + 1;
  for (int i = 0; i < c.n_layer; ++i) {
    bf16_t *k, *v;
    FMI_CHECK(dev_alloc(&k, (int64_t)h->n_pages * s.KVH * KV_PAGE * s.D));
    FMI_CHECK(dev_alloc(&v, (int64_t)h->n_pages * s.KVH * KV_PAGE * s.D));
    h->kpool.push_back(k);
    h->vpool.push_back(v);
  }
  for (int i = 0; i < c.n_fast_layer; ++i) {
    bf16_t *k, *v;
    FMI_CHECK(dev_alloc(&k, (int64_t)max_batch * f.KVH * c.num_codebooks * f.D));
    FMI_CHECK(dev_alloc(&v, (int64_t)max_batch * f.KVH * c.num_codebooks * f.D));
    h->fkc.push_back(k);
    h->fvc.push_back(v);
  }
  SlotState& st = h->st;
  st.max_pages = h->max_pages;
  st.max_frames = h->max_frames;
  st.ncb1 = ncb1;
  FMI_CHECK(dev_alloc(&st.pos, max_batch));
  FMI_CHECK(dev_alloc(&st.frame, max_batch));
  FMI_CHECK(dev_alloc(&st.done, max_batch));
  FMI_CHECK(dev_alloc(&st.limit, max_batch));
  FMI_CHECK(dev_alloc(&st.cur, (int64_t)max_batch * ncb1));
  FMI_CHECK(dev_alloc(&st.window, (int64_t)max_batch * ncb1 * RAS_WIN));
  FMI_CHECK(dev_alloc(&st.out, (int64_t)max_batch * h->max_frames * ncb1));
  FMI_CHECK(dev_alloc(&st.temperature, max_batch));
  FMI_CHECK(dev_alloc(&st.top_p, max_batch));
  FMI_CHECK(dev_alloc(&st.top_k, max_batch));
  FMI_CHECK(dev_alloc(&st.seed, max_batch));
  FMI_CHECK(dev_alloc(&st.use_ras, max_batch));
  FMI_CHECK(dev_alloc(&st.block_table, (int64_t)max_batch * h->max_pages));
  FMI_CHECK(dev_alloc(&h->hn, (int64_t)max_batch * c.dim));
  FMI_CHECK(dev_alloc(&h->xl, (int64_t)max_batch * c.dim));
  FMI_CHECK(dev_alloc(&h->hf, (int64_t)max_batch * c.dim));
  if (c.fast_dim != c.dim) FMI_CHECK(dev_alloc(&h->hfp, (int64_t)2 * max_batch * c.fast_dim));   // parity tap | step-0 work copy
  FMI_CHECK(dev_alloc(&h->xf, (int64_t)max_batch * c.fast_dim));
  FMI_CHECK(dev_alloc(&h->x01, (int64_t)2 * max_batch * c.fast_dim));
  {
    static const bool off = []() { const char* e = getenv("FMI_NO_MERGE01"); return e && atoi(e) != 0; }();
    h->merge01 = !off;
  }
  FMI_CHECK(dev_alloc(&h->logits, (int64_t)max_batch * h->n_live_pad));
  FMI_CHECK(dev_alloc(&h->flogits, (int64_t)max_batch * c.codebook_size));
  FMI_CHECK(dev_alloc(&h->ftrace, (int64_t)max_batch * c.num_codebooks * c.codebook_size));
  h->free_pages.clear();
  for (int p = h->n_pages - 1; p >= 0; --p) h->free_pages.push_back(p);
  h->slot_pages.assign(max_batch, {});
  h->slot_top_k.assign(max_batch, 0);
  h->pos_host.assign(max_batch, 0);
  h->max_top_k = 0;
  {
    static const int env_thr = []() { const char* e = getenv("FMI_ATTN_THR"); return e ? atoi(e) : -1; }();
    if (env_thr >= 0) h->attn_long_thr = env_thr;
    if (!attn_decode_long_supported(s.H, s.KVH, s.D)) h->attn_long_thr = 0;
    if (h->attn_long_thr > 0) FMI_CHECK(dev_alloc(&h->attn_part, attn_decode_long_part_floats(max_batch, s.H, s.D)));
  }
  return ensure_rows(h, std::max(max_batch, std::min(32, 2 * max_batch)));   // the merged fast pass runs 2 B <= 32 rows
}

int fmi_dualar_release(fmi_dualar* h, int slot) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK(check_slot(h, slot));
  for (int p : h->slot_pages[slot]) h->free_pages.push_back(p);
  h->slot_pages[slot].clear();
  h->slot_top_k[slot] = 0;   // the sampler variant is re-derived from the live slots at the next set_slot
  return FMI_OK;
}

static int prefill_impl(fmi_dualar* h, int n, const int32_t* slot_ids, const int32_t* tokens_dev, const int32_t* lens,
                        const int32_t* max_new, const fmi_sampling* samp, int frame_index, hipStream_t s,
                        bool head_only = false, const int32_t* pos0 = nullptr) {
  const fmi_dualar_config& c = h->cfg;
  FMI_CHECK(ensure_qkv0_table(h));
  int rows = 0;
  bool any_long = false;
  for (int i = 0; i < n; ++i) {
    FMI_CHECK(check_slot(h, slot_ids[i]));
    FMI_REQUIRE(lens[i] >= 1, "empty prompt for slot %d", slot_ids[i]);
    const int p0 = pos0 ? pos0[i] : 0;
    FMI_REQUIRE(p0 >= 0, "negative resume position");
    if (p0 > 0)
      FMI_REQUIRE((int)h->slot_pages[slot_ids[i]].size() * KV_PAGE >= p0,
                  "slot %d holds no K/V for positions below %d (released?)", slot_ids[i], p0);
    if (p0 + lens[i] >= h->max_seq)  // inference.py:263-266
      return set_error(FMI_EINVAL, "Input sequence length %d exceeds max_seq_len %d", p0 + lens[i], h->max_seq);
    rows += lens[i];
    any_long = any_long || p0 + lens[i] > 16;
  }
  struct TiledGuard {   // resumed prefill: suffix rows use the full prefill's kernels (see linear())
    fmi_dualar* h;
    ~TiledGuard() { h->force_tiled = false; }
  } guard{h};
  h->force_tiled = pos0 != nullptr && any_long;
  FMI_CHECK(ensure_rows(h, std::max(rows, std::max(h->max_batch, std::min(32, 2 * h->max_batch)))));
  std::vector<int32_t> row_slot(rows), row_pos(rows), last(n), slots(n);
  std::vector<int4> tiles;
  // rows per query tile of the prefill attention (1 / 2 / 3 column groups per work-group share the staged K/V blocks):
  // long prefills take the wide tiles, short ones keep the work-group count up.  Output bits do not depend on it.
  static const int env_mq = []() { const char* e = getenv("FMI_ATTN_MQ"); return e ? atoi(e) : 0; }();
  const int qrows = (env_mq >= 1 && env_mq <= 3) ? 16 * env_mq : rows >= 4096 ? 48 : rows >= 2048 ? 32 : 16;
  h->ws.qtile_rows = qrows;
  int r = 0;
  for (int i = 0; i < n; ++i) {
    const int p0 = pos0 ? pos0[i] : 0;
    const int full = p0 + lens[i];          // prompt length as generate() sees it
    int mn = max_new[i];
    if (mn <= 0 || full + mn > h->max_seq) mn = h->max_seq - full;  // inference.py:268-275
    const int limit = full + mn - 1;
    // positions 0..limit-1 receive K/V; a slot that ends AT its limit parks its position counter on `limit`, so
    // the block-table entry of that position must be the slot's own page too (the decode attention additionally
    // stops appending once SlotState.done is set)
    FMI_CHECK(reserve_pages(h, slot_ids[i], std::min(std::max(limit, full) + 1, h->max_seq)));
    FMI_CHECK(set_slot(h, slot_ids[i], full, frame_index, limit, samp[i], frame_index == 0));
    for (int t0 = 0; t0 < lens[i]; t0 += qrows)
      tiles.push_back(make_int4(r + t0, std::min(qrows, lens[i] - t0), slot_ids[i], p0 + t0));
    for (int t = 0; t < lens[i]; ++t, ++r) {
      row_slot[r] = slot_ids[i];
      row_pos[r] = p0 + t;
    }
    last[i] = r - 1;
    slots[i] = slot_ids[i];
  }
  Workspace& ws = h->ws;
  h->row_slot_host.clear();
  FMI_CHECK_HIP(hipMemcpyAsync(ws.row_slot, row_slot.data(), rows * 4, hipMemcpyHostToDevice, s));
  FMI_CHECK_HIP(hipMemcpyAsync(ws.row_pos, row_pos.data(), rows * 4, hipMemcpyHostToDevice, s));
  FMI_CHECK_HIP(hipMemcpyAsync(ws.last_rows, last.data(), n * 4, hipMemcpyHostToDevice, s));
  // causal work grows with the tile's position: heaviest query tiles first
  std::stable_sort(tiles.begin(), tiles.end(), [](const int4& x, const int4& y) { return x.w > y.w; });
  ws.n_qtiles = (int)tiles.size();
  FMI_CHECK_HIP(hipMemcpyAsync(ws.qtiles, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
  h->launches = 0;
  EmbedArgs e{};
  e.emb = h->emb; e.cb_emb = h->cb_emb; e.tokens = tokens_dev; e.row_slot = nullptr; e.out = ws.x; e.rows = rows;
  e.dim = c.dim; e.ncb = c.num_codebooks; e.cbs = c.codebook_size; e.sem_begin = c.semantic_begin_id;
  e.sem_end = c.semantic_end_id; e.scale = c.scale_codebook_embeddings;
  FMI_CHECK(launch_embed(e, s));
  for (int i = 0; i < c.n_layer; ++i) FMI_CHECK(block_slow(h, h->L[i], i, ws.x, rows, ws.row_slot, ws.row_pos, s));
  // llama.py:447-448: only the last position of each utterance feeds the head
  FMI_CHECK(launch_gather_rows(ws.x, c.dim, ws.last_rows, h->xl, c.dim, n, c.dim, s));
  // the tail addresses slots through row_slot[0..n)
  FMI_CHECK_HIP(hipMemcpyAsync(ws.row_slot, slots.data(), n * 4, hipMemcpyHostToDevice, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  h->row_slot_host = slots;
  h->force_tiled = false;   // the head and the fast chain see n <= 16 rows in a full prefill too
  if (head_only) return tail_head(h, h->xl, n, s);
  return tail(h, h->xl, n, ws.row_slot, s);
}

int fmi_dualar_prefill(fmi_dualar* h, int n, const int32_t* slot_ids, const int32_t* tokens_dev, const int32_t* lens,
                       const int32_t* max_new, const fmi_sampling* samp, void* stream) {
  FMI_REQUIRE(h && slot_ids && tokens_dev && lens && max_new && samp, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_REQUIRE(n >= 1 && n <= h->max_batch, "n=%d out of range", n);
  FMI_CHECK(sync_in(h, stream));
  FMI_CHECK(prefill_impl(h, n, slot_ids, tokens_dev, lens, max_new, samp, 0, h->stream));
  return sync_out(h, stream);
}

int fmi_dualar_prefill_resume(fmi_dualar* h, int n, const int32_t* slot_ids, const int32_t* tokens_dev,
                              const int32_t* lens, const int32_t* pos0, const int32_t* max_new,
                              const fmi_sampling* samp, void* stream) {
  FMI_REQUIRE(h && slot_ids && tokens_dev && lens && pos0 && max_new && samp, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_REQUIRE(n >= 1 && n <= h->max_batch, "n=%d out of range", n);
  FMI_CHECK(sync_in(h, stream));
  FMI_CHECK(prefill_impl(h, n, slot_ids, tokens_dev, lens, max_new, samp, 0, h->stream, false, pos0));
  return sync_out(h, stream);
}

int fmi_dualar_decode(fmi_dualar* h, int n, const int32_t* slot_ids, int n_frames, void* stream) {
  FMI_REQUIRE(h && slot_ids, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_REQUIRE(n >= 1 && n <= h->max_batch, "n=%d out of range", n);
  for (int i = 0; i < n; ++i) FMI_CHECK(check_slot(h, slot_ids[i]));
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  // the slot list of the frame graph: uploaded (and waited for: the source is the caller's memory) only when it
  // changes -- a call that continues the previous one's slots does not drain the queue, so a streaming caller that cuts
  // the frame loop into chunks keeps the GPU fed across its calls
  if (h->row_slot_host != std::vector<int32_t>(slot_ids, slot_ids + n)) {
    FMI_CHECK_HIP(hipMemcpyAsync(h->ws.row_slot, slot_ids, n * 4, hipMemcpyHostToDevice, s));
    FMI_CHECK_HIP(hipStreamSynchronize(s));
    h->row_slot_host.assign(slot_ids, slot_ids + n);
  }
  // which decode-attention kernels the frames of this call need: the VALU kernel if some slot is below the threshold
  // when the call starts, the MFMA pair if some slot reaches it before the call ends (each row then picks by its own
  // position; a launch whose rows are all in the other regime returns at once)
  int mask = 1;
  if (h->attn_long_thr > 0) {
    mask = 0;
    for (int i = 0; i < n; ++i) {
      const int p = h->pos_host[slot_ids[i]];
      if (p < h->attn_long_thr) mask |= 1;
      if (p + n_frames > h->attn_long_thr) mask |= 2;
    }
  }
  h->attn_mask = mask;
  hipGraphExec_t exec = nullptr;
  if (h->use_graph && !h->trace) {
    const int key = n * 4 + mask;
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
      hipGraph_t g = nullptr;
      FMI_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int rc = decode_frame(h, n, s);
      hipError_t e = hipStreamEndCapture(s, &g);
      if (rc != FMI_OK) return rc;
      FMI_CHECK_HIP(e);
      FMI_CHECK_HIP(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
      FMI_CHECK_HIP(hipGraphDestroy(g));
      h->graphs[key] = exec;
    } else {
      exec = it->second;
    }
  }
  FMI_CHECK_HIP(hipEventRecord(h->ev_t0, s));
  for (int f = 0; f < n_frames; ++f) {
    if (exec) FMI_CHECK_HIP(hipGraphLaunch(exec, s));
    else FMI_CHECK(decode_frame(h, n, s));
  }
  for (int i = 0; i < n; ++i)
    h->pos_host[slot_ids[i]] = std::min(h->pos_host[slot_ids[i]] + n_frames, h->max_seq);
  FMI_CHECK_HIP(hipEventRecord(h->ev_t1, s));
  // The caller's stream is NOT made to wait here.  A wait that stays pending on another hardware queue for the length
  // of the frame loop costs every kernel dispatch of that loop (measured, tools/decode_chunk_probe.py: +0.3 ms per frame
  // for the first ~30 frames of every call -- 4.81 -> 4.73 ms per frame for one 192-frame call, 5.10 -> 4.74 for calls
  // of 32 frames).  The frames are ordered before every later call on this handle; a caller that reads them from its
  // own stream first calls fmi_dualar_wait (stream order) or fmi_dualar_synchronize / poll_done / read (host).
  FMI_CHECK_HIP(hipEventRecord(h->ev_out, s));
  h->out_pending = true;
  return FMI_OK;
}

int fmi_dualar_wait(fmi_dualar* h, void* stream) {
  FMI_REQUIRE(h, "null handle");
  if (h->out_pending) FMI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_out, 0));
  return FMI_OK;
}

int fmi_dualar_synchronize(fmi_dualar* h) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  h->out_pending = false;
  return FMI_OK;
}

int fmi_dualar_last_decode_stats(fmi_dualar* h, float* ms, int* launches_per_frame) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipEventSynchronize(h->ev_t1));
  float t = 0.f;
  FMI_CHECK_HIP(hipEventElapsedTime(&t, h->ev_t0, h->ev_t1));
  if (ms) *ms = t;
  if (launches_per_frame) *launches_per_frame = h->frame_launches;
  return FMI_OK;
}

int fmi_dualar_out_ptr(fmi_dualar* h, void** out_dev, int* max_frames) {
  FMI_REQUIRE(h && out_dev, "null argument");
  FMI_REQUIRE(h->max_batch > 0, "setup_caches has not been called");
  *out_dev = h->st.out;
  if (max_frames) *max_frames = h->st.max_frames;
  return FMI_OK;
}

int fmi_dualar_read(fmi_dualar* h, int slot, int32_t* out_host, int max_frames, int* n_frames_out, int* done_out,
                    void* stream) {
  FMI_REQUIRE(h && out_host && n_frames_out, "null argument");
  FMI_CHECK(check_slot(h, slot));
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  int32_t nf = 0, dn = 0;
  FMI_CHECK_HIP(hipMemcpyAsync(&nf, h->st.frame + slot, 4, hipMemcpyDeviceToHost, s));
  FMI_CHECK_HIP(hipMemcpyAsync(&dn, h->st.done + slot, 4, hipMemcpyDeviceToHost, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  const int n = std::min(nf, max_frames);
  if (n > 0)
    FMI_CHECK_HIP(hipMemcpyAsync(out_host, h->st.out + (int64_t)slot * h->st.max_frames * h->st.ncb1,
                                 (size_t)n * h->st.ncb1 * 4, hipMemcpyDeviceToHost, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  *n_frames_out = n;
  if (done_out) *done_out = dn;
  return FMI_OK;
}

int fmi_dualar_poll_done(fmi_dualar* h, int n, const int32_t* slot_ids, int32_t* done_host, void* stream) {
  FMI_REQUIRE(h && slot_ids && done_host, "null argument");
  FMI_CHECK(sync_in(h, stream));
  std::vector<int32_t> all(h->max_batch);
  FMI_CHECK_HIP(hipMemcpyAsync(all.data(), h->st.done, h->max_batch * 4, hipMemcpyDeviceToHost, h->stream));
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    FMI_CHECK(check_slot(h, slot_ids[i]));
    done_host[i] = all[slot_ids[i]];
  }
  return FMI_OK;
}

int fmi_dualar_step(fmi_dualar* h, int slot, const int32_t* x_dev, int S, int pos0, const fmi_sampling* samp,
                    const int32_t* prev_dev, int32_t frame_index, int32_t* out_dev, void* stream) {
  FMI_REQUIRE(h && x_dev && samp && out_dev, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_CHECK(check_slot(h, slot));
  FMI_REQUIRE(S >= 1, "S must be >= 1");
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  const int ncb1 = h->st.ncb1;
  fmi_sampling sp = *samp;
  sp.use_ras = prev_dev != nullptr;
  if (S > 1 || pos0 == 0) {
    FMI_REQUIRE(pos0 == 0, "multi-token call must start at position 0");
    const int32_t len = S, mn = 0;
    FMI_CHECK(prefill_impl(h, 1, &slot, x_dev, &len, &mn, &sp, frame_index, s));
  } else {
    FMI_REQUIRE(pos0 < h->max_seq, "position %d beyond max_seq_len %d", pos0, h->max_seq);
    FMI_CHECK(ensure_qkv0_table(h));
    FMI_CHECK(reserve_pages(h, slot, h->max_seq));
    FMI_CHECK(set_slot(h, slot, pos0, frame_index, h->max_seq, sp, false));
    FMI_CHECK_HIP(hipMemcpyAsync(h->st.cur + (int64_t)slot * ncb1, x_dev, ncb1 * 4, hipMemcpyDeviceToDevice, s));
    h->row_slot_host.clear();
    FMI_CHECK_HIP(hipMemcpyAsync(h->ws.row_slot, &slot, 4, hipMemcpyHostToDevice, s));
    FMI_CHECK_HIP(hipStreamSynchronize(s));
    if (prev_dev)
      FMI_CHECK_HIP(hipMemcpyAsync(h->st.window + (int64_t)slot * ncb1 * RAS_WIN, prev_dev,
                                   (size_t)ncb1 * RAS_WIN * 4, hipMemcpyDeviceToDevice, s));
    h->attn_mask = (h->attn_long_thr > 0 && pos0 >= h->attn_long_thr) ? 2 : 1;
    FMI_CHECK(decode_frame(h, 1, s));
  }
  FMI_CHECK_HIP(hipMemcpyAsync(out_dev, h->st.cur + (int64_t)slot * ncb1, ncb1 * 4, hipMemcpyDeviceToDevice, s));
  return sync_out(h, stream);
}

// BaseTransformer.forward_generate (llama.py:390-466): slow transformer + final norm + tied head for one
// slot; logits over the live rows and the normed hidden state land in the library's tap buffers.
int fmi_dualar_forward_slow(fmi_dualar* h, int slot, const int32_t* x_dev, int S, int pos0, void* logits_out_dev,
                            void* hidden_out_dev, void* stream) {
  FMI_REQUIRE(h && x_dev, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_CHECK(check_slot(h, slot));
  FMI_REQUIRE(S >= 1, "S must be >= 1");
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  const int ncb1 = h->st.ncb1;
  fmi_sampling sp{1.0f, 1.0f, 1, 0u, 0};
  if (S > 1 || pos0 == 0) {
    FMI_REQUIRE(pos0 == 0, "multi-token call must start at position 0");
    const int32_t len = S, mn = 0;
    FMI_CHECK(prefill_impl(h, 1, &slot, x_dev, &len, &mn, &sp, 0, s, true));
  } else {
    FMI_REQUIRE(pos0 < h->max_seq, "position %d beyond max_seq_len %d", pos0, h->max_seq);
    FMI_CHECK(reserve_pages(h, slot, h->max_seq));
    FMI_CHECK(set_slot(h, slot, pos0, 1, h->max_seq, sp, false));
    FMI_CHECK_HIP(hipMemcpyAsync(h->st.cur + (int64_t)slot * ncb1, x_dev, ncb1 * 4, hipMemcpyDeviceToDevice, s));
    h->row_slot_host.clear();
    FMI_CHECK_HIP(hipMemcpyAsync(h->ws.row_slot, &slot, 4, hipMemcpyHostToDevice, s));
    FMI_CHECK_HIP(hipStreamSynchronize(s));
    h->attn_mask = (h->attn_long_thr > 0 && pos0 >= h->attn_long_thr) ? 2 : 1;
    FMI_CHECK(decode_frame(h, 1, s, true));
  }
  if (logits_out_dev)
    FMI_CHECK_HIP(hipMemcpyAsync(logits_out_dev, h->logits, (size_t)h->n_live * 2, hipMemcpyDeviceToDevice, s));
  if (hidden_out_dev) {
    // llama.py:459-461: hidden_states = slow_out (normed) if norm_fastlayer_input else the un-normed last row
    // (llama.py:827: DualARTransformer.forward_generate hands back fast_project_in(hidden): fast_dim values then)
    const bf16_t* hid = h->cfg.norm_fastlayer_input ? h->hn : ((S > 1 || pos0 == 0) ? h->xl : h->ws.x);
    if (h->fpi_w) {
      FMI_CHECK(project_fast_in(h, hid, h->hfp, 1, s));
      hid = h->hfp;
    }
    FMI_CHECK_HIP(hipMemcpyAsync(hidden_out_dev, hid, (size_t)h->cfg.fast_dim * 2, hipMemcpyDeviceToDevice, s));
  }
  return sync_out(h, stream);
}

// DualARTransformer.forward_generate_fast (llama.py:799-817): one fast-AR position for one slot.
int fmi_dualar_forward_fast(fmi_dualar* h, int slot, const void* hidden_in_dev, int pos, void* logits_out_dev,
                            void* stream) {
  FMI_REQUIRE(h && hidden_in_dev && logits_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_CHECK(check_slot(h, slot));
  const fmi_dualar_config& c = h->cfg;
  FMI_REQUIRE(pos >= 0 && pos < c.num_codebooks, "fast position %d out of range", pos);
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  FMI_CHECK_HIP(hipMemcpyAsync(h->xf, hidden_in_dev, (size_t)c.fast_dim * 2, hipMemcpyDeviceToDevice, s));
  h->row_slot_host.clear();
    FMI_CHECK_HIP(hipMemcpyAsync(h->ws.row_slot, &slot, 4, hipMemcpyHostToDevice, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < c.n_fast_layer; ++i) FMI_CHECK(block_fast(h, h->FL[i], i, h->xf, 1, pos, h->ws.row_slot, s));
  FMI_CHECK(linear(h, h->xf, c.fast_dim, h->fast_out, h->fast_norm, nullptr, 0, h->flogits, c.codebook_size, 1,
                   c.codebook_size, c.fast_dim, EPI_STORE, s, h->q_fast_out, h->s_fast_out));
  FMI_CHECK_HIP(hipMemcpyAsync(logits_out_dev, h->flogits, (size_t)c.codebook_size * 2, hipMemcpyDeviceToDevice, s));
  return sync_out(h, stream);
}

// Test seam for the float parity of the fast chain ON THE FRAME LOOP'S PATH: steps 6-8 of
// decode_one_token_ar (inference.py:148-176) for the B given slots from given NORMED hidden rows, run by tail() exactly
// as a decode frame runs it -- batch GEMV, merged positions 0/1 when enabled, the tabulated layer-0 q|k|v when
// table != 0 -- with every draw REPLACED by the forced frame (slow token, then codes 1..ncb-1 feed positions 2..), and
// every position's logits kept.  Frame bookkeeping runs too: the slots' frame counters advance like after a frame.
int fmi_dualar_fast_chain_forced(fmi_dualar* h, int B, const int32_t* slot_ids, const void* hidden_normed_dev,
                                 const int32_t* forced_dev, int table, void* fast_logits_out_dev, void* stream) {
  FMI_REQUIRE(h && slot_ids && hidden_normed_dev && forced_dev && fast_logits_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "weights not ready");
  FMI_REQUIRE(B >= 1 && B <= h->max_batch, "B=%d out of range", B);
  for (int i = 0; i < B; ++i) FMI_CHECK(check_slot(h, slot_ids[i]));
  const fmi_dualar_config& c = h->cfg;
  FMI_CHECK(ensure_rows(h, std::max(2 * B, 16)));
  FMI_CHECK(sync_in(h, stream));
  FMI_CHECK(ensure_qkv0_table(h));
  hipStream_t s = h->stream;
  h->row_slot_host.clear();
  FMI_CHECK_HIP(hipMemcpyAsync(h->ws.row_slot, slot_ids, (size_t)B * 4, hipMemcpyHostToDevice, s));
  // forced frame by SLOT (the samplers index it like the slot state)
  const int ncb1 = c.num_codebooks + 1;
  int32_t* forced_by_slot = nullptr;
  FMI_CHECK(dev_alloc(&forced_by_slot, (int64_t)h->max_batch * ncb1));
  FMI_CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < B; ++i)
    FMI_CHECK_HIP(hipMemcpyAsync(forced_by_slot + (int64_t)slot_ids[i] * ncb1, forced_dev + (int64_t)i * ncb1,
                                 (size_t)ncb1 * 4, hipMemcpyDeviceToDevice, s));
  const int saved_trace = h->trace;
  h->trace = table ? 2 : 1;
  h->forced = forced_by_slot;
  h->tail_in_normed = true;
  const int rc = tail(h, (const bf16_t*)hidden_normed_dev, B, h->ws.row_slot, s);
  h->trace = saved_trace;
  h->forced = nullptr;
  h->tail_in_normed = false;
  if (rc == FMI_OK)
    hipMemcpyAsync(fast_logits_out_dev, h->ftrace, (size_t)B * c.num_codebooks * c.codebook_size * 2,
                   hipMemcpyDeviceToDevice, s);
  hipStreamSynchronize(s);
  hipFree(forced_by_slot);
  FMI_CHECK(rc);
  return sync_out(h, stream);
}

// Row-major tables inside the arena that the host mirror may index itself (embedding lookups are
// tensor plumbing): which = 0 fast_embeddings (codebook_size x fast_dim bf16), 1 live ids (int32 n_live)
int fmi_dualar_table_ptr(fmi_dualar* h, int which, void** ptr, int* rows, int* cols) {
  FMI_REQUIRE(h && ptr, "null argument");
  if (which == 0) { *ptr = h->fast_emb; if (rows) *rows = h->cfg.codebook_size; if (cols) *cols = h->cfg.fast_dim; }
  else if (which == 1) { *ptr = h->live_ids; if (rows) *rows = h->n_live; if (cols) *cols = 1; }
  else return set_error(FMI_EINVAL, "unknown table %d", which);
  return FMI_OK;
}

// What this handle has derived from the arena so far (row-balanced decode copies, the fast layer-0 q|k|v table): both are
// built lazily by the first prefill on EVERY rank -- they are not part of the broadcast.
int fmi_dualar_derived_info(fmi_dualar* h, int* row_copies, int* table_rows, int* loaded_tensors) {
  FMI_REQUIRE(h, "null handle");
  if (row_copies) *row_copies = (int)h->row_copies.size();
  if (table_rows) *table_rows = h->qkv0_tab ? h->cfg.codebook_size : 0;
  if (loaded_tensors) *loaded_tensors = (int)h->loaded.size();
  return FMI_OK;
}

int fmi_dualar_debug_ptrs(fmi_dualar* h, void** slow_logits, int* n_live, int* ld_logits, void** live_ids,
                          void** hidden, void** fast_logits) {
  FMI_REQUIRE(h, "null handle");
  if (slow_logits) *slow_logits = h->logits;
  if (n_live) *n_live = h->n_live;
  if (ld_logits) *ld_logits = h->n_live_pad;
  if (live_ids) *live_ids = h->live_ids;
  if (hidden) *hidden = h->fpi_w ? h->hfp : h->hn;   // what the fast transformer is handed (llama.py:827)
  if (fast_logits) *fast_logits = h->flogits;
  return FMI_OK;
}

int fmi_dualar_set_trace(fmi_dualar* h, int enable, void** fast_trace) {
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(enable >= 0 && enable <= 2, "trace mode %d", enable);
  if (h->trace != enable) {   // the captured frames embed (or lack) the trace copies and the table's use
    FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
    drop_graphs(h);
  }
  h->trace = enable;
  if (fast_trace) *fast_trace = h->ftrace;
  return FMI_OK;
}

int fmi_dualar_set_ignore_eos(fmi_dualar* h, int enable) {
  FMI_REQUIRE(h, "null handle");
  if (h->ignore_eos != (enable != 0)) {
    hipStreamSynchronize(h->stream);
    drop_graphs(h);
  }
  h->ignore_eos = enable != 0;
  return FMI_OK;
}

int fmi_dualar_set_attn_impl(fmi_dualar* h, int impl) {
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(impl == 0 || impl == 1, "attn impl must be 0 (VALU) or 1 (MFMA)");
  h->attn_impl = impl;
  return FMI_OK;
}

// The handle's private stream re-created with a dispatch priority (-1 = highest, 0 = default, 1 = lowest): when another
// queue (the codec of the previous batch) is busy beside the frame loop, the loop's launches go first.
int fmi_dualar_set_stream_priority(fmi_dualar* h, int priority) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  int lo = 0, hi = 0;
  FMI_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const int pr = priority < 0 ? hi : (priority > 0 ? lo : (lo + hi) / 2);
  hipStream_t ns = nullptr;
  FMI_CHECK_HIP(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, pr));
  hipStreamDestroy(h->stream);
  h->stream = ns;
  return FMI_OK;
}

int fmi_dualar_set_fast_merge(fmi_dualar* h, int enable) {
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  h->merge01 = enable != 0;
  return FMI_OK;
}

int fmi_dualar_set_attn_long_threshold(fmi_dualar* h, int threshold) {
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(threshold >= 0, "threshold must be >= 0 (0 = VALU kernel for every row)");
  FMI_REQUIRE(threshold == 0 || h->attn_part, "the MFMA decode attention is not available for this shape / was disabled at setup_caches");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  h->attn_long_thr = threshold;
  return FMI_OK;
}

int fmi_dualar_set_graph(fmi_dualar* h, int enable) {
  FMI_REQUIRE(h, "null handle");
  h->use_graph = enable != 0;
  return FMI_OK;
}

// ------------------------------------------------------------------------- op-level entry points

int fmi_op_linear_bf16(const void* x_dev, const void* w_dev, const void* norm_w_dev, const void* residual_dev,
                       void* out_dev, int M, int N, int K, float eps, int epilogue, int force_path, void* stream) {
  FMI_REQUIRE(x_dev && w_dev && out_dev, "null argument");
  FMI_REQUIRE(epilogue >= 0 && epilogue <= 2, "bad epilogue");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* packed = nullptr;
  bf16_t* xn = nullptr;
  FMI_CHECK_HIP(hipMalloc((void**)&packed, (size_t)N * K * 2));
  int rc;
  const int n_out = (epilogue == EPI_SILU) ? N / 2 : N;
  if (epilogue == EPI_SILU) {
    rc = launch_pack_weight((const bf16_t*)w_dev, packed, N / 2, K, 1, s);
    if (rc == FMI_OK) rc = launch_pack_weight((const bf16_t*)w_dev + (int64_t)(N / 2) * K, packed, N / 2, K, 2, s);
  } else {
    rc = launch_pack_weight((const bf16_t*)w_dev, packed, N, K, 0, s);
  }
  LinearArgs a{};
  a.wp = packed; a.x = (const bf16_t*)x_dev; a.ldx = K; a.norm_w = (const bf16_t*)norm_w_dev; a.eps = eps;
  a.res = (const bf16_t*)residual_dev; a.ldr = n_out; a.out = (bf16_t*)out_dev; a.ldo = n_out; a.M = M; a.N = N;
  a.K = K; a.epi = epilogue;
  // 2: tiled (default variant), 5: tiled, operands straight from L2, 6: skinny on the row-balanced copy (must exist),
  // 7: tiled, LDS-staged 4-wave kernel, 8: tiled, LDS-staged wave-specialised kernel, 9: its 128 x 256-tile variant,
  // 10 / 11 / 14 / 15: linear_tiled_256p_kernel with 256 / 128 / 64 / 192-row tiles (round 4, what the shape selects among),
  // 12: the 16-wave and 13: the plain 8-wave 256 x 256 kernels (A/B)
  const bool skinny = force_path == 1 || force_path == 6 || (force_path == 0 && M <= 16);
  bf16_t* rowcopy = nullptr;
  if (rc == FMI_OK && force_path == 6) {
    if (M > 8 || !skinny_rows_supported(N, K, epilogue, norm_w_dev != nullptr)) {
      rc = set_error(FMI_EINVAL, "no row-balanced variant for M=%d N=%d K=%d epilogue=%d", M, N, K, epilogue);
    } else {
      const RowPlan p = skinny_row_plan(N, K, epilogue);
      if (hipMalloc((void**)&rowcopy, (size_t)p.elems * 2) != hipSuccess) rc = set_error(FMI_EHIP, "hipMalloc");
      if (rc == FMI_OK) rc = launch_repack_rows(packed, rowcopy, N, K, epilogue, p, s);
      a.wr = rowcopy;
    }
  }
  if (rc == FMI_OK) {
    if (skinny) {
      rc = launch_linear_skinny(a, s);
    } else {
      if (a.norm_w) {
        if (hipMalloc((void**)&xn, (size_t)M * K * 2) != hipSuccess) rc = set_error(FMI_EHIP, "hipMalloc");
        if (rc == FMI_OK) rc = launch_rmsnorm_rows(a.x, K, a.norm_w, eps, xn, K, M, K, s);
        a.x = xn;
        a.norm_w = nullptr;
      }
      float* part = nullptr;   // (the shape may run with a split contraction: linear_tiled_ksplit)
      if (rc == FMI_OK && (force_path == 2 || force_path == 0) && linear_tiled_part_floats(M, N, K) > 0) {
        if (hipMalloc((void**)&part, (size_t)linear_tiled_part_floats(M, N, K) * 4) != hipSuccess) rc = set_error(FMI_EHIP, "hipMalloc");
        a.part = part;
      }
      if (rc == FMI_OK) rc = launch_linear_tiled(a, s, force_path == 5, force_path == 7 ? 1 : force_path == 8 ? 2 : force_path == 9 ? 3 : force_path == 10 ? 7 : force_path == 11 ? 10 : force_path == 12 ? 9 : force_path == 13 ? 4 : force_path == 14 ? 11 : force_path == 15 ? 12 : 0);
      hipStreamSynchronize(s);
      if (part) hipFree(part);
    }
  }
  hipStreamSynchronize(s);
  hipFree(packed);
  if (xn) hipFree(xn);
  if (rowcopy) hipFree(rowcopy);
  return rc;
}

int fmi_op_linear_int8(const void* x_dev, const void* w_i8_dev, const void* scales_dev, const void* norm_w_dev,
                       const void* residual_dev, void* out_dev, int M, int N, int K, float eps, int epilogue,
                       int stream_int8, void* stream) {
  FMI_REQUIRE(x_dev && w_i8_dev && scales_dev && out_dev, "null argument");
  FMI_REQUIRE(epilogue >= 0 && epilogue <= 2 && M >= 1 && M <= 16 && K % 64 == 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  bf16_t *deq = nullptr, *packed = nullptr, *sc = nullptr;
  int8_t* q = nullptr;
  FMI_CHECK_HIP(hipMalloc((void**)&deq, (size_t)N * K * 2));
  FMI_CHECK_HIP(hipMalloc((void**)&packed, (size_t)N * K * 2));
  FMI_CHECK_HIP(hipMalloc((void**)&q, (size_t)N * K));
  FMI_CHECK_HIP(hipMalloc((void**)&sc, (size_t)N * 2));
  const int n_out = (epilogue == EPI_SILU) ? N / 2 : N;
  int rc = launch_dequant_int8((const int8_t*)w_i8_dev, deq, (int64_t)N * K, s);
  auto pack = [&](int64_t row0, int rows, int il) -> int {
    FMI_CHECK(launch_pack_weight(deq + row0 * K, packed, rows, K, il, s));
    FMI_CHECK(launch_pack_weight_int8((const int8_t*)w_i8_dev + row0 * K, q, rows, K, il, s));
    return launch_pack_scale((const bf16_t*)scales_dev + row0, sc, rows, il, s);
  };
  if (rc == FMI_OK) {
    if (epilogue == EPI_SILU) {
      rc = pack(0, N / 2, 1);
      if (rc == FMI_OK) rc = pack(N / 2, N / 2, 2);
    } else {
      rc = pack(0, N, 0);
    }
  }
  LinearArgs a{};
  a.wp = packed; a.x = (const bf16_t*)x_dev; a.ldx = K; a.norm_w = (const bf16_t*)norm_w_dev; a.eps = eps;
  a.res = (const bf16_t*)residual_dev; a.ldr = n_out; a.out = (bf16_t*)out_dev; a.ldo = n_out; a.M = M; a.N = N;
  a.K = K; a.epi = epilogue; a.scale = sc; a.wq = stream_int8 ? q : nullptr;
  if (rc == FMI_OK) rc = launch_linear_skinny(a, s);
  hipStreamSynchronize(s);
  hipFree(deq); hipFree(packed); hipFree(q); hipFree(sc);
  return rc;
}

int fmi_op_sample(const void* logits_dev, int B, int n, int ld, const int32_t* ids_dev, const fmi_sampling* samp,
                  int frame, int draw, const int32_t* prev_dev, int sem_begin, int sem_end, int32_t* out_dev,
                  void* stream) {
  FMI_REQUIRE(logits_dev && samp && out_dev, "null argument");
  SampleArgs a{};
  a.logits = (const bf16_t*)logits_dev; a.B = B; a.n = n; a.ld = ld; a.ids = ids_dev; a.row_slot = nullptr;
  a.mode = 2; a.sem_begin = sem_begin; a.sem_end = sem_end; a.temperature = rbf(samp->temperature);
  a.top_p = rbf(samp->top_p); a.top_k = samp->top_k; a.seed = samp->seed; a.frame = frame; a.draw = draw;
  a.prev = prev_dev; a.out_tok = out_dev; a.xf = nullptr; a.small_k = samp->top_k <= 64;
  return launch_sample(a, (hipStream_t)stream);
}

}  // extern "C"
