// dac.hip -- host side of the codec behind the C ABI: weight registry + arena, decode / encode graphs.
//
// Reference call path replaced (fish_speech/models/dac/):
//   DAC.from_indices / DAC.decode   modded_dac.py:925-946  (quantizer.decode rvq.py:352-366 + Decoder)
//   DAC.encode                      modded_dac.py:874-923  (Encoder + quantizer.forward rvq.py:293-316)
// Tensor names are the keys of codec.pth after weight-norm folding ("....conv.weight").
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "dac_kernels.h"

using namespace fmi;

namespace {

enum SlotKind { K_RAW, K_CONV, K_CONVTR, K_CONV_PART };

struct Slot {
  SlotKind kind;
  float* dst;
  std::vector<int64_t> dims;  // expected source dims
  int cout_pad = 0, cin_pad = 0, stride = 1, co_off = 0;
  bool loaded = false;
};

struct Conv {
  ConvW w;
  int k = 1, stride = 1, dil = 1;
  bool transposed = false;
};

struct ResUnit {
  float *a1 = nullptr, *a2 = nullptr;
  Conv c7, c1;
  int dil = 1;
};

struct TfLayer {
  float *attn_norm, *ffn_norm, *g_attn, *g_ffn;
  Conv wqkv, wo, w13, w2;
};
struct Tf {
  std::vector<TfLayer> layers;
  float* norm = nullptr;
  int dim = 0, ffn = 0, window = 0;
};

struct ConvNeXt {
  float *dw_w, *dw_b, *ln_w, *ln_b, *gamma;
  Conv pw1, pw2;
};

struct EncBlock {
  ResUnit ru[3];
  float* alpha;
  Conv down;
  bool has_tf = false;
  Tf tf;
};
struct DecBlock {
  float* alpha;
  Conv up;
  ResUnit ru[3];
};
struct VQ {
  float *in_w, *in_b, *cb, *out_w, *out_b;
  int n;
};

struct Buf {
  float* p = nullptr;
  int64_t n = 0;
};

}  // namespace

struct fmi_dac {
  fmi_dac_config cfg;
  char* arena = nullptr;
  int64_t arena_bytes = 0, off = 0;
  std::map<std::string, Slot> reg;
  bool ready = false, rope_loaded = false;
  int head_dim = 64;
  float eps = 1e-5f;
  // model
  float *first_w = nullptr, *first_b = nullptr;
  std::vector<EncBlock> enc;
  float* enc_alpha = nullptr;
  Conv enc_out;
  std::vector<Conv> down_conv, up_conv;
  std::vector<ConvNeXt> down_cnx, up_cnx;
  Tf pre, post;
  VQ sem;
  std::vector<VQ> rvq;
  float* lut = nullptr;
  int* lut_off = nullptr;
  Conv dec_in;
  std::vector<DecBlock> dec;
  float *dec_alpha = nullptr, *final_w = nullptr, *final_b = nullptr;
  bf16_t* rope = nullptr;
  int rope_pos = 0;
  // runtime
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool async_out = false;   // fmi_dac_set_async
  int lds_floor = 0;        // fmi_dac_set_background
  int* f16_ovf = nullptr;   // device word: an operand of the fp16-split arithmetic left the fp16 range (fmi_dac_fp16_overflow)
  Buf buf[6];
  void* staging = nullptr;
  size_t staging_bytes = 0;
  float* last_z = nullptr;
  int last_zC = 0, last_zL = 0;
  // decode-side contraction arithmetic (ConvArgs::planes): 2 = fp16 matrix cores on the scaled two-term split
  // (fp32-class, default), 0 = fp32 matrix cores, 1 = bf16 operands and results (autocast).  The encoder always runs
  // on the fp32 matrix cores.
  int decode_planes = 2;
  int cur_planes = 0;
  bf16_t* pbuf[2] = {nullptr, nullptr};   // operand-plane hand-off buffers of the decoder
  int64_t pbuf_bytes[2] = {0, 0};
  // one workspace per handle: request threads that share the codec object (inference_engine/__init__.py:179-192,
  // tools/api_server.py:115-122) are serialised here, whole call by whole call
  std::mutex mu;
  std::vector<ConvW> plane_jobs;
  // quantizer-side state of an incremental (streaming) decode: fmi_dac_decode_tail_cached
  struct StreamState {
    int B = 0, T = 0, cap = 0;            // utterances, frames covered so far, frame capacity of the buffers
    int64_t id = 0;                       // the caller's stream id the state belongs to
    int planes = -1;                      // arithmetic (fmi_dac_set_precision) the state was computed with
    std::vector<float*> qkv;              // per post-transformer layer: roped q|k|v of every frame [B][3C][cap]
    float* tf_out = nullptr;              // transformer output [B][C][cap]
    float* z = nullptr;                   // upsampled latents [B][latent][4 cap]
    // decoder left context (round 6): per plane-consuming conv of the decoder with k_eff > 1 -- block b's transposed conv
    // (1 column) and its three dilated k = 7 convs (6, 18, 54) -- the last k_eff - 1 columns of its input operand planes,
    // and the last 8 fp32 columns in front of the final conv; two copies, flipped per call (halo_swap reads one, writes
    // the other).  dec_valid: they belong to frames [0, T) of this stream.
    std::vector<bf16_t*> dh[2];           // (pointers into dslab)
    float* dx[2] = {nullptr, nullptr};
    char* dslab[2] = {nullptr, nullptr};  // one allocation per copy: cleared by one memset
    size_t dslab_bytes = 0;
    int dflip = 0;
    bool dec_valid = false;
    uint64_t used = 0;                    // LRU stamp
  } st;                                   // the state the kernels of the current call work on
  // states of OTHER streams that are still open (a serving loop interleaves the chunks of several utterances, each
  // with its own stream id): parked here between their calls, least recently used one dropped beyond MAX_PARKED
  std::vector<StreamState> parked;
  uint64_t st_clock = 0;
  static constexpr size_t MAX_PARKED = 15;
};

namespace {

constexpr int ROPE_POSITIONS = 32768;

int frame_length(const fmi_dac_config& c) {
  int h = 1;
  for (int i = 0; i < 4; ++i) h *= c.encoder_rates[i];
  return h * c.downsample[0] * c.downsample[1];
}

int validate(const fmi_dac_config& c) {
  FMI_REQUIRE(c.encoder_dim > 0 && c.decoder_dim > 0 && c.latent_dim == c.encoder_dim * 16, "latent_dim must be encoder_dim*16");
  FMI_REQUIRE(c.latent_dim % 64 == 0, "latent_dim %% 64 (transformer head_dim 64)");
  FMI_REQUIRE(c.codebook_dim >= 1 && c.codebook_dim <= 16, "codebook_dim in [1,16]");
  FMI_REQUIRE(c.n_codebooks >= 1 && c.n_codebooks <= 32, "n_codebooks");
  for (int i = 0; i < 4; ++i) FMI_REQUIRE(c.encoder_rates[i] >= 1 && c.decoder_rates[i] >= 1, "rates");
  FMI_REQUIRE(c.decoder_dim % 16 == 0, "decoder_dim %% 16");
  FMI_REQUIRE(c.tf_ffn > 0 && c.tf_layers >= 0 && c.tf_window > 0 && c.enc_tf_window > 0, "transformer config");
  return FMI_OK;
}

// ---- arena bump allocation + registry.  With h->arena == nullptr only the size is computed.
struct Builder {
  fmi_dac* h;
  int64_t off = 0;
  bool planes = false;   // also reserve the bf16 hi/mid/lo planes of the convs built from here on (decode side)
  void add_planes(Conv& c) {
    if (!planes) return;
    c.w.cin_pad16 = (int)align_up(c.w.cin, 16);
    c.w.wb = (const bf16_t*)take((int64_t)c.w.phases * c.w.taps * c.w.cin_pad16 * c.w.cout_pad * 3, 2);
    h->plane_jobs.push_back(c.w);
  }
  float* take(int64_t elems, int esize = 4) {
    float* p = h->arena ? (float*)(h->arena + off) : nullptr;
    off = align_up(off + elems * esize, 256);
    return p;
  }
  float* raw(const std::string& name, std::vector<int64_t> dims) {
    int64_t n = 1;
    for (auto d : dims) n *= d;
    float* p = take(n);
    Slot s;
    s.kind = K_RAW; s.dst = p; s.dims = dims;
    h->reg[name] = s;
    return p;
  }
  Conv conv(const std::string& prefix, int cout, int cin, int k, int stride, int dil, bool bias, bool transposed) {
    Conv c;
    c.k = k; c.stride = stride; c.dil = dil; c.transposed = transposed;
    c.w.cin = cin; c.w.cout = cout;
    c.w.cin_pad = (int)align_up(cin, 8);
    c.w.cout_pad = (int)align_up(cout, 32);
    if (transposed) {
      c.w.phases = stride;
      c.w.taps = k / stride;
    } else {
      c.w.phases = 1;
      c.w.taps = k;
    }
    float* wp = take((int64_t)c.w.phases * c.w.taps * c.w.cin_pad * c.w.cout_pad);
    c.w.w = wp;
    Slot s;
    s.kind = transposed ? K_CONVTR : K_CONV; s.dst = wp;
    s.dims = transposed ? std::vector<int64_t>{cin, cout, k} : std::vector<int64_t>{cout, cin, k};
    s.cout_pad = c.w.cout_pad; s.cin_pad = c.w.cin_pad; s.stride = stride;
    h->reg[prefix + ".weight"] = s;
    if (bias) c.w.bias = raw(prefix + ".bias", {cout});
    add_planes(c);
    return c;
  }
  // linear layer (k=1, no bias) whose rows come from `parts` checkpoint tensors stacked along cout
  Conv linear_parts(const std::vector<std::string>& names, int cout_each, int cin) {
    Conv c;
    const int cout = cout_each * (int)names.size();
    c.w.cin = cin; c.w.cout = cout;
    c.w.cin_pad = (int)align_up(cin, 8);
    c.w.cout_pad = (int)align_up(cout, 32);
    float* wp = take((int64_t)c.w.cin_pad * c.w.cout_pad);
    c.w.w = wp;
    for (size_t i = 0; i < names.size(); ++i) {
      Slot s;
      s.kind = K_CONV_PART; s.dst = wp; s.dims = {cout_each, cin};
      s.cout_pad = c.w.cout_pad; s.cin_pad = c.w.cin_pad; s.co_off = (int)i * cout_each;
      h->reg[names[i]] = s;
    }
    add_planes(c);
    return c;
  }
  ResUnit res_unit(const std::string& p, int dim, int dil) {
    ResUnit r;
    r.dil = dil;
    r.a1 = raw(p + ".block.0.alpha", {1, dim, 1});
    r.c7 = conv(p + ".block.1.conv", dim, dim, 7, 1, dil, true, false);
    r.a2 = raw(p + ".block.2.alpha", {1, dim, 1});
    r.c1 = conv(p + ".block.3.conv", dim, dim, 1, 1, 1, true, false);
    return r;
  }
  Tf transformer(const std::string& p, int dim, int n_layer, int ffn, int window) {
    Tf t;
    t.dim = dim; t.ffn = ffn; t.window = window;
    for (int i = 0; i < n_layer; ++i) {
      const std::string l = p + ".layers." + std::to_string(i);
      TfLayer L;
      L.wqkv = linear_parts({l + ".attention.wqkv.weight"}, 3 * dim, dim);
      L.wo = linear_parts({l + ".attention.wo.weight"}, dim, dim);
      L.w13 = linear_parts({l + ".feed_forward.w1.weight", l + ".feed_forward.w3.weight"}, ffn, dim);
      L.w2 = linear_parts({l + ".feed_forward.w2.weight"}, dim, ffn);
      L.ffn_norm = raw(l + ".ffn_norm.weight", {dim});
      L.attn_norm = raw(l + ".attention_norm.weight", {dim});
      L.g_attn = raw(l + ".attention_layer_scale.gamma", {dim});
      L.g_ffn = raw(l + ".ffn_layer_scale.gamma", {dim});
      t.layers.push_back(L);
    }
    t.norm = raw(p + ".norm.weight", {dim});
    return t;
  }
  ConvNeXt convnext(const std::string& p, int L) {
    ConvNeXt c;
    c.gamma = raw(p + ".gamma", {L});
    c.dw_w = raw(p + ".dwconv.conv.weight", {L, 1, 7});
    c.dw_b = raw(p + ".dwconv.conv.bias", {L});
    c.ln_w = raw(p + ".norm.weight", {L});
    c.ln_b = raw(p + ".norm.bias", {L});
    c.pw1 = linear_parts({p + ".pwconv1.weight"}, 4 * L, L);
    c.pw1.w.bias = raw(p + ".pwconv1.bias", {4 * L});
    c.pw2 = linear_parts({p + ".pwconv2.weight"}, L, 4 * L);
    c.pw2.w.bias = raw(p + ".pwconv2.bias", {L});
    return c;
  }
  VQ vq(const std::string& p, int L, int d, int n) {
    VQ v;
    v.n = n;
    v.in_b = raw(p + ".in_proj.bias", {d});
    v.in_w = raw(p + ".in_proj.weight", {d, L, 1});
    v.out_b = raw(p + ".out_proj.bias", {L});
    v.out_w = raw(p + ".out_proj.weight", {L, d, 1});
    v.cb = raw(p + ".codebook.weight", {n, d});
    return v;
  }
};

int64_t build(fmi_dac* h) {
  const fmi_dac_config& c = h->cfg;
  Builder b{h};
  h->reg.clear();
  h->plane_jobs.clear();
  h->enc.clear(); h->dec.clear(); h->rvq.clear();
  h->down_conv.clear(); h->up_conv.clear(); h->down_cnx.clear(); h->up_cnx.clear();
  const int L = c.latent_dim;
  // encoder (modded_dac.py:670-709)
  int d = c.encoder_dim;
  h->first_b = b.raw("encoder.block.0.conv.bias", {d});
  h->first_w = b.raw("encoder.block.0.conv.weight", {d, 1, 7});
  for (int bi = 0; bi < 4; ++bi) {
    d *= 2;
    const std::string p = "encoder.block." + std::to_string(bi + 1);
    EncBlock e;
    const int dils[3] = {1, 3, 9};
    for (int r = 0; r < 3; ++r) e.ru[r] = b.res_unit(p + ".block." + std::to_string(r), d / 2, dils[r]);
    e.alpha = b.raw(p + ".block.3.alpha", {1, d / 2, 1});
    const int s = c.encoder_rates[bi];
    e.down = b.conv(p + ".block.4.conv", d, d / 2, 2 * s, s, 1, true, false);
    if (bi == 3 && c.enc_tf_layers > 0) {
      e.has_tf = true;
      e.tf = b.transformer(p + ".block.5", d, c.enc_tf_layers, d * 3, c.enc_tf_window);
    }
    h->enc.push_back(e);
  }
  h->enc_alpha = b.raw("encoder.block.5.alpha", {1, d, 1});
  h->enc_out = b.conv("encoder.block.6.conv", L, d, 3, 1, 1, true, false);
  // quantizer (rvq.py:204-291)
  h->sem = b.vq("quantizer.semantic_quantizer.quantizers.0", L, c.codebook_dim, c.semantic_codebook_size);
  for (int i = 0; i < c.n_codebooks; ++i)
    h->rvq.push_back(b.vq("quantizer.quantizer.quantizers." + std::to_string(i), L, c.codebook_dim, c.codebook_size));
  for (int i = 0; i < 2; ++i) {
    const std::string p = "quantizer.downsample." + std::to_string(i);
    h->down_conv.push_back(b.conv(p + ".0.conv", L, L, c.downsample[i], c.downsample[i], 1, true, false));
    h->down_cnx.push_back(b.convnext(p + ".1", L));
  }
  b.planes = true;   // everything built from here on runs in from_indices / decode
  for (int i = 0; i < 2; ++i) {
    const std::string p = "quantizer.upsample." + std::to_string(i);
    const int f = c.downsample[1 - i];
    h->up_conv.push_back(b.conv(p + ".0.conv", L, L, f, f, 1, true, true));
    h->up_cnx.push_back(b.convnext(p + ".1", L));
  }
  b.planes = false;
  h->pre = b.transformer("quantizer.pre_module", L, c.tf_layers, c.tf_ffn, c.tf_window);
  b.planes = true;
  h->post = b.transformer("quantizer.post_module", L, c.tf_layers, c.tf_ffn, c.tf_window);
  const int64_t lut_rows = c.semantic_codebook_size + (int64_t)c.n_codebooks * c.codebook_size;
  h->lut = b.take(lut_rows * L);
  h->lut_off = (int*)b.take(c.n_codebooks + 2, 4);
  // decoder (modded_dac.py:760-801)
  h->dec_in = b.conv("decoder.model.0.conv", c.decoder_dim, L, 7, 1, 1, true, false);
  for (int i = 0; i < 4; ++i) {
    const int cin = c.decoder_dim >> i, cout = c.decoder_dim >> (i + 1);
    const std::string p = "decoder.model." + std::to_string(i + 1);
    DecBlock db;
    db.alpha = b.raw(p + ".block.0.alpha", {1, cin, 1});
    const int s = c.decoder_rates[i];
    db.up = b.conv(p + ".block.1.conv", cout, cin, 2 * s, s, 1, true, true);
    const int dils[3] = {1, 3, 9};
    for (int r = 0; r < 3; ++r) db.ru[r] = b.res_unit(p + ".block." + std::to_string(2 + r), cout, dils[r]);
    h->dec.push_back(db);
  }
  const int cl = c.decoder_dim >> 4;
  h->dec_alpha = b.raw("decoder.model.5.alpha", {1, cl, 1});
  h->final_b = b.raw("decoder.model.6.conv.bias", {1});
  h->final_w = b.raw("decoder.model.6.conv.weight", {1, cl, 7});
  h->rope = (bf16_t*)b.take((int64_t)ROPE_POSITIONS * h->head_dim, 2);
  h->rope_pos = ROPE_POSITIONS;
  return b.off;
}

int ensure_staging(fmi_dac* h, size_t bytes) {
  if (h->staging_bytes >= bytes) return FMI_OK;
  if (h->staging) hipFree(h->staging);
  h->staging = nullptr;
  h->staging_bytes = 0;
  FMI_CHECK_HIP(hipMalloc(&h->staging, bytes));
  h->staging_bytes = bytes;
  return FMI_OK;
}

int ensure_buf(fmi_dac* h, int i, int64_t n) {
  if (h->buf[i].n >= n) return FMI_OK;
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  if (h->buf[i].p) hipFree(h->buf[i].p);
  h->buf[i].p = nullptr;
  h->buf[i].n = 0;
  FMI_CHECK_HIP(hipMalloc((void**)&h->buf[i].p, (size_t)n * 4));
  h->buf[i].n = n;
  return FMI_OK;
}

int sync_in(fmi_dac* h, void* us) {
  set_f16_overflow_target(h->f16_ovf);   // every entry point comes through here, under h->mu: its kernels flag THIS handle
  set_conv_lds_floor(h->lds_floor);
  FMI_CHECK_HIP(hipEventRecord(h->ev_in, (hipStream_t)us));
  FMI_CHECK_HIP(hipStreamWaitEvent(h->stream, h->ev_in, 0));
  return FMI_OK;
}
int sync_out(fmi_dac* h, void* us) {
  FMI_CHECK_HIP(hipEventRecord(h->ev_out, h->stream));
  FMI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)us, h->ev_out, 0));
  return FMI_OK;
}
// encode / decode entry points: with fmi_dac_set_async(h, 1) the caller's stream is NOT made to wait for the call (it
// orders itself with fmi_dac_wait or waits with fmi_dac_synchronize) -- a codec call that runs beside the Dual-AR frame
// loop must not leave a cross-queue wait pending for its whole length (fishmi.h: fmi_dualar_decode, +0.3 ms per frame)
int sync_out_data(fmi_dac* h, void* us) {
  if (!h->async_out) return sync_out(h, us);
  FMI_CHECK_HIP(hipEventRecord(h->ev_out, h->stream));
  return FMI_OK;
}

// ---- layer runners (channel-major [B][C][L])

int run_conv(fmi_dac* h, const Conv& c, const float* x, float* out, int B, int lin, int* lout_p, const float* snake,
             const float* res, const float* gamma, int act, const bf16_t* xp = nullptr, bf16_t* outp = nullptr,
             const float* next_alpha = nullptr) {
  ConvArgs a{};
  a.w = c.w; a.x = x; a.out = out; a.snake_alpha = snake; a.res = res; a.gamma = gamma; a.B = B; a.lin = lin;
  a.act = act;
  a.planes = c.w.wb ? h->cur_planes : 0;
  a.xp = xp; a.outp = outp; a.next_alpha = next_alpha;
  FMI_REQUIRE((!xp && !outp) || a.planes > 0, "operand planes need the bf16-plane kernel");
  int lout;
  if (c.transposed) {  // CausalTransConvNet: (lin-1)*s + k - (k - s) = lin * s
    lout = lin * c.stride;
    a.x_stride = 1; a.tap_step = -1; a.tap_base = 0; a.out_stride = c.stride;
  } else {             // CausalConvNet: left pad k_eff - stride, output ceil(lin / stride)
    const int k_eff = (c.k - 1) * c.dil + 1;
    lout = cdiv(lin, c.stride);
    a.x_stride = c.stride; a.tap_step = c.dil; a.tap_base = -(k_eff - c.stride); a.out_stride = 1;
  }
  a.lout = lout;
  if (lout_p) *lout_p = lout;
  return launch_conv(a, h->stream);
}

// ResidualUnit in place on x, y = scratch of the same size
int run_res_unit(fmi_dac* h, const ResUnit& r, float* x, float* y, int B, int L) {
  FMI_CHECK(run_conv(h, r.c7, x, y, B, L, nullptr, r.a1, nullptr, nullptr, ACT_NONE));
  return run_conv(h, r.c1, y, x, B, L, nullptr, r.a2, x, nullptr, ACT_NONE);
}

// The codec transformer on operand planes (fp16 split arithmetic only): column norm / attention / SiLU-mul hand fp16
// hi/lo planes to linear_planes_kernel, which fetches both MFMA operands from L2 without any staging.
bool tf_on_planes(const fmi_dac* h, const Tf& t) {
  static const bool off = []() { const char* e = getenv("FMI_DAC_TF_PLANES"); return e && atoi(e) == 0; }();
  if (off || h->cur_planes != 2 || t.layers.empty() || h->head_dim != 64 || t.window > 128) return false;
  if (t.dim % 16 || t.ffn % 16) return false;
  for (const TfLayer& L : t.layers)
    for (const Conv* c : {&L.wqkv, &L.wo, &L.w13, &L.w2})
      if (!c->w.wb || c->w.taps != 1 || c->w.phases != 1 || c->w.cin_pad16 != c->w.cin) return false;
  return true;
}

// one layer on columns [q_lo, T) of qkv_hist (row stride ld; q_lo = 0, ld = T offline); x, nbp, qkv, ab, actp are compact
// over the n = T - q_lo new columns
int tf_layer_planes(fmi_dac* h, const Tf& t, const TfLayer& L, float* x, bf16_t* nbp, float* qkv, float* qkv_hist, int ld,
                    float* ab, bf16_t* actp, int B, int T, int q_lo) {
  const int C = t.dim, F = t.ffn, n = T - q_lo;
  hipStream_t s = h->stream;
  FMI_CHECK(launch_rmsnorm_cols_planes(x, L.attn_norm, h->eps, nbp, B, C, n, s));
  FMI_CHECK(launch_linear_planes(L.wqkv.w, nbp, qkv, nullptr, nullptr, ACT_NONE, B, n, s));
  FMI_CHECK(launch_rope_cols(qkv, h->rope + (int64_t)q_lo * h->head_dim, B, C, n, h->head_dim, s));
  if (qkv_hist != qkv)
    FMI_CHECK_HIP(hipMemcpy2DAsync(qkv_hist + q_lo, (size_t)ld * 4, qkv, (size_t)n * 4, (size_t)n * 4, (size_t)B * 3 * C,
                                   hipMemcpyDeviceToDevice, s));
  FMI_CHECK(launch_window_attn(qkv_hist, nullptr, B, C, T, h->head_dim, t.window, s, ld, q_lo, nbp));
  FMI_CHECK(launch_linear_planes(L.wo.w, nbp, x, x, L.g_attn, ACT_NONE, B, n, s));
  FMI_CHECK(launch_rmsnorm_cols_planes(x, L.ffn_norm, h->eps, nbp, B, C, n, s));
  FMI_CHECK(launch_linear_planes(L.w13.w, nbp, ab, nullptr, nullptr, ACT_NONE, B, n, s));
  FMI_CHECK(launch_silu_mul_planes(ab, actp, B, F, n, s));
  return launch_linear_planes(L.w2.w, actp, x, x, L.g_ffn, ACT_NONE, B, n, s);
}

// WindowLimitedTransformer on x [B][dim][T]; result written to out (may alias x)
int run_transformer(fmi_dac* h, const Tf& t, float* x, float* out, int B, int T) {
  const int C = t.dim, F = t.ffn;
  FMI_REQUIRE(T <= h->rope_pos, "sequence of %d frames exceeds the RoPE table (%d)", T, h->rope_pos);
  FMI_CHECK(ensure_buf(h, 2, (int64_t)B * C * T));       // norm / attention output
  FMI_CHECK(ensure_buf(h, 3, (int64_t)B * 3 * C * T));   // qkv
  FMI_CHECK(ensure_buf(h, 4, (int64_t)B * 2 * F * T));   // w1|w3 output
  FMI_CHECK(ensure_buf(h, 5, (int64_t)B * F * T));       // activation
  float *nb = h->buf[2].p, *qkv = h->buf[3].p, *ab = h->buf[4].p, *act = h->buf[5].p;
  hipStream_t s = h->stream;
  if (tf_on_planes(h, t)) {
    for (const TfLayer& L : t.layers)
      FMI_CHECK(tf_layer_planes(h, t, L, x, (bf16_t*)nb, qkv, qkv, T, ab, (bf16_t*)act, B, T, 0));
    return launch_rmsnorm_cols(x, t.norm, h->eps, out, B, C, T, s);
  }
  for (const TfLayer& L : t.layers) {
    FMI_CHECK(launch_rmsnorm_cols(x, L.attn_norm, h->eps, nb, B, C, T, s));
    FMI_CHECK(run_conv(h, L.wqkv, nb, qkv, B, T, nullptr, nullptr, nullptr, nullptr, ACT_NONE));
    FMI_CHECK(launch_rope_cols(qkv, h->rope, B, C, T, h->head_dim, s));
    FMI_CHECK(launch_window_attn(qkv, nb, B, C, T, h->head_dim, t.window, s));
    FMI_CHECK(run_conv(h, L.wo, nb, x, B, T, nullptr, nullptr, x, L.g_attn, ACT_NONE));
    FMI_CHECK(launch_rmsnorm_cols(x, L.ffn_norm, h->eps, nb, B, C, T, s));
    FMI_CHECK(run_conv(h, L.w13, nb, ab, B, T, nullptr, nullptr, nullptr, nullptr, ACT_NONE));
    FMI_CHECK(launch_silu_mul(ab, act, B, F, T, s));
    FMI_CHECK(run_conv(h, L.w2, act, x, B, T, nullptr, nullptr, x, L.g_ffn, ACT_NONE));
  }
  return launch_rmsnorm_cols(x, t.norm, h->eps, out, B, C, T, s);
}

// ConvNeXtBlock in place on x [B][L][T]
int run_convnext(fmi_dac* h, const ConvNeXt& c, float* x, int B, int C, int T) {
  FMI_CHECK(ensure_buf(h, 2, (int64_t)B * C * T));
  FMI_CHECK(ensure_buf(h, 3, (int64_t)B * C * T));
  FMI_CHECK(ensure_buf(h, 4, (int64_t)B * 4 * C * T));
  float *y = h->buf[2].p, *n = h->buf[3].p, *hid = h->buf[4].p;
  hipStream_t s = h->stream;
  FMI_CHECK(launch_dwconv7(x, c.dw_w, c.dw_b, y, B, C, T, s));
  FMI_CHECK(launch_layernorm_cols(y, c.ln_w, c.ln_b, 1e-6f, n, B, C, T, s));
  FMI_CHECK(run_conv(h, c.pw1, n, hid, B, T, nullptr, nullptr, nullptr, nullptr, ACT_GELU));
  return run_conv(h, c.pw2, hid, x, B, T, nullptr, nullptr, x, c.gamma, ACT_NONE);
}

// Decoder (modded_dac.py:760-801) on z = X [B][latent][len]; Y = scratch of the peak size
// skip_cols: leading latent columns whose audio is not written (left context of an incremental decode)
int64_t decode_peak_elems(const fmi_dac_config& c, int T);

// The same with activations handed from conv to conv as bf16 operand planes (ConvArgs::xp / outp): Snake and the
// three-way split run once, in the producer's epilogue.  X = z fp32 [B][latent][len]; X, Y double as the fp32
// residual stream of the ResidualUnits (in place in one of them).
int run_decoder_planes(fmi_dac* h, float* X, float* Y, int B, int len, float* audio_out_dev, int skip_cols) {
  const fmi_dac_config& c = h->cfg;
  const int64_t pbytes = (int64_t)B * decode_peak_elems(c, cdiv(len, 4)) * 2 * h->cur_planes;
  for (int i = 0; i < 2; ++i)
    if (h->pbuf_bytes[i] < pbytes) {
      FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
      if (h->pbuf[i]) hipFree(h->pbuf[i]);
      h->pbuf[i] = nullptr;
      h->pbuf_bytes[i] = 0;
      FMI_CHECK_HIP(hipMalloc((void**)&h->pbuf[i], (size_t)pbytes));
      h->pbuf_bytes[i] = pbytes;
    }
  bf16_t *cur = h->pbuf[0], *oth = h->pbuf[1];   // every conv reads `cur` and writes `oth`, then they swap
  int l2;
  // first conv reads fp32 z; its output only feeds block 0's Snake + transposed conv
  FMI_CHECK(run_conv(h, h->dec_in, X, nullptr, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE, nullptr, cur, h->dec[0].alpha));
  for (size_t bi = 0; bi < h->dec.size(); ++bi) {
    const DecBlock& db = h->dec[bi];
    // transposed conv: fp32 out = residual stream of the three ResidualUnits, planes = snake_a1(out) for ru[0].c7
    FMI_CHECK(run_conv(h, db.up, nullptr, X, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE, cur, oth, db.ru[0].a1));
    std::swap(cur, oth);
    len = l2;
    for (int r = 0; r < 3; ++r) {
      const ResUnit& ru = db.ru[r];
      FMI_CHECK(run_conv(h, ru.c7, nullptr, nullptr, B, len, nullptr, nullptr, nullptr, nullptr, ACT_NONE, cur, oth, ru.a2));
      std::swap(cur, oth);
      const bool last_ru = r == 2, last_block = bi + 1 == h->dec.size();
      // who reads this unit's output next: the next unit's first Snake, the next block's Snake, or the final conv (fp32)
      const float* na = !last_ru ? db.ru[r + 1].a1 : (last_block ? nullptr : h->dec[bi + 1].alpha);
      FMI_CHECK(run_conv(h, ru.c1, nullptr, X, B, len, nullptr, nullptr, X, nullptr, ACT_NONE, cur,
                         (last_ru && last_block) ? nullptr : oth, na));
      std::swap(cur, oth);
    }
  }
  int hop = 1;
  for (int i = 0; i < 4; ++i) hop *= c.decoder_rates[i];
  (void)Y;
  return launch_final_conv_tanh(X, h->dec_alpha, h->final_w, h->final_b, audio_out_dev, B, c.decoder_dim >> 4, len,
                                skip_cols * hop, h->stream);
}

int run_decoder(fmi_dac* h, float* X, float* Y, int B, int len, float* audio_out_dev, int skip_cols = 0) {
  const fmi_dac_config& c = h->cfg;
  static const bool fused_off = []() { const char* e = getenv("FMI_DAC_NO_PLANE_HANDOFF"); return e && atoi(e) != 0; }();
  bool chain16 = (c.decoder_dim >> 4) % 16 == 0;   // every stage's channel count is a multiple of 16
  if (h->cur_planes > 0 && chain16 && !fused_off && h->dec.size() == 4 && h->dec_in.w.wb)
    return run_decoder_planes(h, X, Y, B, len, audio_out_dev, skip_cols);
  int l2;
  FMI_CHECK(run_conv(h, h->dec_in, X, Y, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE));
  std::swap(X, Y);
  for (const DecBlock& db : h->dec) {
    FMI_CHECK(run_conv(h, db.up, X, Y, B, len, &l2, db.alpha, nullptr, nullptr, ACT_NONE));
    std::swap(X, Y);
    len = l2;
    for (int r = 0; r < 3; ++r) FMI_CHECK(run_res_unit(h, db.ru[r], X, Y, B, len));
  }
  int hop = 1;
  for (int i = 0; i < 4; ++i) hop *= c.decoder_rates[i];
  return launch_final_conv_tanh(X, h->dec_alpha, h->final_w, h->final_b, audio_out_dev, B, c.decoder_dim >> 4, len,
                                skip_cols * hop, h->stream);
}

// ---- streaming decoder (round 6): every conv sees the previous chunk's tail as real left context ------------------
// halo columns of the j-th stateful conv input: block b: [up (1), c7 d1 (6), c7 d3 (18), c7 d9 (54)]
constexpr int FINAL_HALO = 8;   // fp32 columns kept in front of the final k = 7 conv (6 needed; 8 keeps rows 16-byte aligned)
inline int dec_halo_cols(int j) {
  const int r = j & 3;
  return r == 0 ? 1 : r == 1 ? 6 : r == 2 ? 18 : 54;
}
bool decoder_streams(const fmi_dac* h) {
  const fmi_dac_config& c = h->cfg;
  static const bool off = []() { const char* e = getenv("FMI_DAC_NO_STREAM_HALO"); return e && atoi(e) != 0; }();   // (A/B: context recompute)
  if (off || h->cur_planes <= 0 || (c.decoder_dim >> 4) % 16 != 0 || h->dec.size() != 4 || !h->dec_in.w.wb) return false;
  if (h->dec_in.k != 7) return false;
  for (const DecBlock& db : h->dec) {
    if (!db.up.transposed || db.up.k != 2 * db.up.stride) return false;
    for (int r = 0; r < 3; ++r)
      if (db.ru[r].c7.k != 7 || db.ru[r].c7.dil != (r == 0 ? 1 : r == 1 ? 3 : 9) || db.ru[r].c1.k != 1) return false;
  }
  return true;
}

int alloc_dec_halos(fmi_dac* h, fmi_dac::StreamState& st, int B) {
  const fmi_dac_config& c = h->cfg;
  size_t off[17], total = 0;
  for (int j = 0; j < 16; ++j) {
    const int bi = j >> 2;
    const int cin = (j & 3) == 0 ? (c.decoder_dim >> bi) : (c.decoder_dim >> (bi + 1));   // up reads the block's input width
    off[j] = total;
    total += (size_t)B * (cin / 16) * 2 * dec_halo_cols(j) * 32;       // two planes at most
  }
  off[16] = total;
  total += (size_t)B * (c.decoder_dim >> 4) * FINAL_HALO * 4;
  for (int k = 0; k < 2; ++k) {
    FMI_CHECK_HIP(hipMalloc((void**)&st.dslab[k], total));
    st.dh[k].assign(16, nullptr);
    for (int j = 0; j < 16; ++j) st.dh[k][j] = reinterpret_cast<bf16_t*>(st.dslab[k] + off[j]);
    st.dx[k] = reinterpret_cast<float*>(st.dslab[k] + off[16]);
  }
  st.dslab_bytes = total;
  st.dec_valid = false;
  return FMI_OK;
}

int zero_dec_halos(fmi_dac* h, fmi_dac::StreamState& st, int B) {
  (void)B;
  for (int k = 0; k < 2; ++k) FMI_CHECK_HIP(hipMemsetAsync(st.dslab[k], 0, st.dslab_bytes, h->stream));
  st.dflip = 0;
  return FMI_OK;
}

// Zin: fp32 latents [B][latent][zl + len], its first zl columns = real left context of the first conv (0 at the start of an
// utterance); X: fp32 residual-stream buffer; audio of the last (len - skip_cols) latent columns is written.  Same kernels,
// same products per output element as run_decoder_planes: bit-identical to the offline decode.
int run_decoder_stream(fmi_dac* h, fmi_dac::StreamState& st, const float* Zin, int zl, float* X, int B, int len,
                       float* audio_out_dev, int skip_cols) {
  const fmi_dac_config& c = h->cfg;
  const int NP = h->cur_planes;
  hipStream_t s = h->stream;
  const int64_t pbytes = (int64_t)B * decode_peak_elems(c, cdiv(len, 4) + 8) * 2 * NP;
  for (int i = 0; i < 2; ++i)
    if (h->pbuf_bytes[i] < pbytes) {
      FMI_CHECK_HIP(hipStreamSynchronize(s));
      if (h->pbuf[i]) hipFree(h->pbuf[i]);
      h->pbuf[i] = nullptr;
      h->pbuf_bytes[i] = 0;
      FMI_CHECK_HIP(hipMalloc((void**)&h->pbuf[i], (size_t)pbytes));
      h->pbuf_bytes[i] = pbytes;
    }
  bf16_t *cur = h->pbuf[0], *oth = h->pbuf[1];
  const std::vector<bf16_t*>&old_h = st.dh[st.dflip], &new_h = st.dh[st.dflip ^ 1];
  auto conv = [&](const Conv& cv, const float* x, int x_ld, int x_left, const bf16_t* xp, int xh, float* out, int out_ld,
                  const float* res, bf16_t* outp, int oh, int lin, int* lout_p, const float* next_alpha) -> int {
    ConvArgs a{};
    a.w = cv.w; a.x = x; a.out = out; a.res = res; a.B = B; a.lin = lin; a.act = ACT_NONE; a.planes = NP;
    a.next_alpha = next_alpha;
    int lout;
    if (cv.transposed) {
      lout = lin * cv.stride;
      a.x_stride = 1; a.tap_step = -1; a.tap_base = 0; a.out_stride = cv.stride;
    } else {
      const int k_eff = (cv.k - 1) * cv.dil + 1;
      lout = cdiv(lin, cv.stride);
      a.x_stride = cv.stride; a.tap_step = cv.dil; a.tap_base = -(k_eff - cv.stride); a.out_stride = 1;
    }
    a.lout = lout;
    if (xp) { a.xp = xp + (int64_t)xh * 16; a.x_ld = xh + lin; a.x_left = xh; }
    else { a.x_ld = x_ld; a.x_left = x_left; }
    if (outp) { a.outp = outp + (int64_t)oh * 16; a.outp_ld = oh + lout; }
    a.out_ld = out_ld;
    if (lout_p) *lout_p = lout;
    return launch_conv(a, s);
  };
  auto swap_in = [&](bf16_t* buf, int j, int cin, int n) -> int {   // halo j of the buffer the next conv reads
    return launch_halo_swap(buf, old_h[j], new_h[j], (int64_t)B * (cin / 16) * NP, dec_halo_cols(j), n, 32, s);
  };
  int l2;
  // first conv: fp32 z with its own left context; its output only feeds block 0's Snake + transposed conv
  FMI_CHECK(conv(h->dec_in, Zin + zl, zl + len, zl, nullptr, 0, nullptr, 0, nullptr, cur, dec_halo_cols(0), len, &l2, h->dec[0].alpha));
  int hop = 1;
  for (int i = 0; i < 4; ++i) hop *= c.decoder_rates[i];
  const int len_final = len * hop;
  float* Xl = X;        // logical column 0 of the fp32 residual stream (moved behind the halo in the last block)
  int xld = 0;
  for (size_t bi = 0; bi < h->dec.size(); ++bi) {
    const DecBlock& db = h->dec[bi];
    const bool last_block = bi + 1 == h->dec.size();
    const int cin = c.decoder_dim >> bi, cout = c.decoder_dim >> (bi + 1);
    FMI_CHECK(swap_in(cur, (int)bi * 4 + 0, cin, len));
    if (last_block) { Xl = X + FINAL_HALO; xld = FINAL_HALO + len * db.up.stride; }
    FMI_CHECK(conv(db.up, nullptr, 0, 0, cur, 1, Xl, xld, nullptr, oth, 6, len, &l2, db.ru[0].a1));
    std::swap(cur, oth);
    len = l2;
    for (int r = 0; r < 3; ++r) {
      const ResUnit& ru = db.ru[r];
      const int j = (int)bi * 4 + 1 + r;
      FMI_CHECK(swap_in(cur, j, cout, len));
      FMI_CHECK(conv(ru.c7, nullptr, 0, 0, cur, dec_halo_cols(j), nullptr, 0, nullptr, oth, 0, len, nullptr, ru.a2));
      std::swap(cur, oth);
      const bool last_ru = r == 2;
      const float* na = !last_ru ? db.ru[r + 1].a1 : (last_block ? nullptr : h->dec[bi + 1].alpha);
      const int oh = !last_ru ? dec_halo_cols(j + 1) : (last_block ? 0 : dec_halo_cols((int)(bi + 1) * 4));
      FMI_CHECK(conv(ru.c1, nullptr, 0, 0, cur, 0, Xl, xld, Xl, (last_ru && last_block) ? nullptr : oth, oh, len, nullptr, na));
      std::swap(cur, oth);
    }
  }
  FMI_REQUIRE(len == len_final, "stream decoder: length bookkeeping");
  FMI_CHECK(launch_halo_swap(X, st.dx[st.dflip], st.dx[st.dflip ^ 1], (int64_t)B * (c.decoder_dim >> 4), FINAL_HALO, len, 4, s));
  st.dflip ^= 1;
  return launch_final_conv_tanh(X, h->dec_alpha, h->final_w, h->final_b, audio_out_dev, B, c.decoder_dim >> 4, FINAL_HALO + len,
                                FINAL_HALO + skip_cols * hop, s);
}

int64_t decode_peak_elems(const fmi_dac_config& c, int T) {
  int64_t peak = (int64_t)c.latent_dim * 4 * T;
  int64_t L = 4 * (int64_t)T;
  peak = std::max(peak, (int64_t)c.decoder_dim * L);
  for (int i = 0; i < 4; ++i) {
    L *= c.decoder_rates[i];
    peak = std::max(peak, (int64_t)(c.decoder_dim >> (i + 1)) * L);
  }
  return peak;
}

// Left context, in latent columns, after which a decoder output no longer depends on what preceded the cropped
// input: walk the decoder backwards (final k7 conv; per block three ResidualUnits k7 dil 1/3/9 and the
// k=2s transposed conv, which reads x[q] and x[q-1]; first k7 conv).  19 columns for rates (8,8,4,2).
int decoder_context_cols(const fmi_dac_config& c) {
  int ctx = 6;
  for (int i = 3; i >= 0; --i) {
    ctx += 6 * (1 + 3 + 9);
    ctx = cdiv(ctx, c.decoder_rates[i]) + 1;
  }
  return ctx + 6;
}

// quantizer.decode (rvq.py:352-366) for T frames; z ends up in h->buf[5] [B][latent][4T]
int run_quantizer_decode(fmi_dac* h, int64_t* indices_dev, int B, int T, float** Xp, float** Yp, int* len_p) {
  const fmi_dac_config& c = h->cfg;
  const int L0 = c.latent_dim;
  hipStream_t s = h->stream;
  const int64_t peak = (int64_t)B * decode_peak_elems(c, T);
  FMI_CHECK(ensure_buf(h, 0, peak));
  FMI_CHECK(ensure_buf(h, 1, peak));
  float *X = h->buf[0].p, *Y = h->buf[1].p;
  FMI_CHECK(launch_clamp_indices(indices_dev, B, c.n_codebooks + 1, T, c.semantic_codebook_size, c.codebook_size, s));
  FMI_CHECK(launch_lut_decode(indices_dev, h->lut, h->lut_off, c.n_codebooks, c.semantic_codebook_size, c.codebook_size,
                              X, B, L0, T, s));
  FMI_CHECK(run_transformer(h, h->post, X, X, B, T));
  int len = T;
  for (int i = 0; i < 2; ++i) {
    int l2;
    FMI_CHECK(run_conv(h, h->up_conv[i], X, Y, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE));
    std::swap(X, Y);
    len = l2;
    FMI_CHECK(run_convnext(h, h->up_cnx[i], X, B, L0, len));
  }
  // z must survive the decoder (debug tap, incremental decode): keep a copy in buffer 5
  FMI_CHECK(ensure_buf(h, 5, (int64_t)B * L0 * len));
  FMI_CHECK_HIP(hipMemcpyAsync(h->buf[5].p, X, (size_t)B * L0 * len * 4, hipMemcpyDeviceToDevice, s));
  h->last_z = h->buf[5].p; h->last_zC = L0; h->last_zL = len;
  *Xp = X; *Yp = Y; *len_p = len;
  return FMI_OK;
}

void free_one_state(fmi_dac::StreamState& st) {
  for (float* p : st.qkv)
    if (p) hipFree(p);
  st.qkv.clear();
  if (st.tf_out) hipFree(st.tf_out);
  if (st.z) hipFree(st.z);
  st.tf_out = st.z = nullptr;
  for (int k = 0; k < 2; ++k) {
    st.dh[k].clear();
    st.dx[k] = nullptr;
    if (st.dslab[k]) hipFree(st.dslab[k]);
    st.dslab[k] = nullptr;
  }
  st.dslab_bytes = 0;
  st.dec_valid = false;
  st.B = st.T = st.cap = 0;
  st.id = 0;
}

void free_stream_state(fmi_dac* h) { free_one_state(h->st); }

void free_all_stream_states(fmi_dac* h) {
  free_one_state(h->st);
  for (auto& p : h->parked) free_one_state(p);
  h->parked.clear();
}

// Make h->st the state of `stream_id`: the current one is parked if it belongs to another open stream, the wanted
// one is taken back from the parked set if it is there (else h->st stays empty and the call starts from frame 0).
int select_stream_state(fmi_dac* h, int64_t stream_id) {
  h->st.used = ++h->st_clock;
  if (h->st.id == stream_id || h->st.T == 0) {
    if (h->st.id != stream_id) {          // an empty / invalid state: reuse its buffers only for the same stream
      for (size_t i = 0; i < h->parked.size(); ++i)
        if (h->parked[i].id == stream_id) {
          FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
          free_one_state(h->st);
          h->st = h->parked[i];
          h->parked.erase(h->parked.begin() + i);
          h->st.used = h->st_clock;
          break;
        }
    }
    return FMI_OK;
  }
  fmi_dac::StreamState mine;
  for (size_t i = 0; i < h->parked.size(); ++i)
    if (h->parked[i].id == stream_id) {
      mine = h->parked[i];
      h->parked.erase(h->parked.begin() + i);
      break;
    }
  h->parked.push_back(h->st);
  h->st = mine;
  h->st.used = h->st_clock;
  if (h->parked.size() > fmi_dac::MAX_PARKED) {
    size_t lru = 0;
    for (size_t i = 1; i < h->parked.size(); ++i)
      if (h->parked[i].used < h->parked[lru].used) lru = i;
    FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
    free_one_state(h->parked[lru]);
    h->parked.erase(h->parked.begin() + lru);
  }
  return FMI_OK;
}

// frames of transformer output to the left of a frame that its upsampled latents depend on (two stages of
// [transposed conv k = stride, ConvNeXt with a causal depthwise k = 7])
int upsampler_context_frames(const fmi_dac_config& c) {
  int ctx = 0;
  for (int i = 1; i >= 0; --i) ctx = cdiv(ctx + 6, c.downsample[1 - i]);
  return ctx;
}

int copy_cols(fmi_dac* h, float* dst, int dst_ld, const float* src, int src_ld, int cols, int64_t rows) {
  FMI_CHECK_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * 4, src, (size_t)src_ld * 4, (size_t)cols * 4, (size_t)rows,
                                 hipMemcpyDeviceToDevice, h->stream));
  return FMI_OK;
}

// quantizer.decode for frames [t0, T) only, on top of the state of frames [0, t0) (t0 = 0: from scratch).  Same
// kernels as run_quantizer_decode on the new columns; every one of them computes a column from its own inputs only
// (k = 1 convs, column norms, RoPE by position, attention by query position), so z is bit-identical to the offline one.
int run_quantizer_decode_inc(fmi_dac* h, int64_t* indices_dev, int B, int T, int t0) {
  const fmi_dac_config& c = h->cfg;
  const Tf& tf = h->post;
  const int C = tf.dim, F = tf.ffn, L0 = c.latent_dim, n = T - t0, cap = h->st.cap;
  FMI_REQUIRE(C == L0, "post transformer width differs from the latent width");
  FMI_REQUIRE(T <= h->rope_pos, "sequence of %d frames exceeds the RoPE table (%d)", T, h->rope_pos);
  hipStream_t s = h->stream;
  const int64_t peak = (int64_t)B * decode_peak_elems(c, T);
  FMI_CHECK(ensure_buf(h, 0, peak));
  FMI_CHECK(ensure_buf(h, 1, peak));
  FMI_CHECK(ensure_buf(h, 2, (int64_t)B * C * n));
  FMI_CHECK(ensure_buf(h, 3, (int64_t)B * 3 * C * n));
  FMI_CHECK(ensure_buf(h, 4, (int64_t)B * 2 * F * n));
  FMI_CHECK(ensure_buf(h, 5, (int64_t)B * F * n));
  float *X = h->buf[0].p, *Y = h->buf[1].p;
  float *nb = h->buf[2].p, *qkv = h->buf[3].p, *ab = h->buf[4].p, *act = h->buf[5].p;
  FMI_CHECK(launch_clamp_indices(indices_dev, B, c.n_codebooks + 1, T, c.semantic_codebook_size, c.codebook_size, s));
  // code embeddings of every frame (a gather), then the new columns as a compact [B][C][n]
  FMI_CHECK(launch_lut_decode(indices_dev, h->lut, h->lut_off, c.n_codebooks, c.semantic_codebook_size, c.codebook_size,
                              Y, B, L0, T, s));
  FMI_CHECK(copy_cols(h, X, n, Y + t0, T, n, (int64_t)B * C));
  const bool on_planes = tf_on_planes(h, tf);
  for (size_t li = 0; li < tf.layers.size(); ++li) {
    const TfLayer& L = tf.layers[li];
    float* hist = h->st.qkv[li];
    if (on_planes) {
      FMI_CHECK(tf_layer_planes(h, tf, L, X, (bf16_t*)nb, qkv, hist, cap, ab, (bf16_t*)act, B, T, t0));
      continue;
    }
    FMI_CHECK(launch_rmsnorm_cols(X, L.attn_norm, h->eps, nb, B, C, n, s));
    FMI_CHECK(run_conv(h, L.wqkv, nb, qkv, B, n, nullptr, nullptr, nullptr, nullptr, ACT_NONE));
    FMI_CHECK(launch_rope_cols(qkv, h->rope + (int64_t)t0 * h->head_dim, B, C, n, h->head_dim, s));
    FMI_CHECK(copy_cols(h, hist + t0, cap, qkv, n, n, (int64_t)B * 3 * C));
    FMI_CHECK(launch_window_attn(hist, nb, B, C, T, h->head_dim, tf.window, s, cap, t0));
    FMI_CHECK(run_conv(h, L.wo, nb, X, B, n, nullptr, nullptr, X, L.g_attn, ACT_NONE));
    FMI_CHECK(launch_rmsnorm_cols(X, L.ffn_norm, h->eps, nb, B, C, n, s));
    FMI_CHECK(run_conv(h, L.w13, nb, ab, B, n, nullptr, nullptr, nullptr, nullptr, ACT_NONE));
    FMI_CHECK(launch_silu_mul(ab, act, B, F, n, s));
    FMI_CHECK(run_conv(h, L.w2, act, X, B, n, nullptr, nullptr, X, L.g_ffn, ACT_NONE));
  }
  FMI_CHECK(launch_rmsnorm_cols(X, tf.norm, h->eps, nb, B, C, n, s));
  FMI_CHECK(copy_cols(h, h->st.tf_out + t0, cap, nb, n, n, (int64_t)B * C));
  // upsampler over [t0 - cu, T): its first 4 cu output columns see a truncated left context and are dropped
  const int cu = std::min(t0, upsampler_context_frames(c));
  int len = cu + n;
  FMI_CHECK(copy_cols(h, X, len, h->st.tf_out + (t0 - cu), cap, len, (int64_t)B * C));
  for (int i = 0; i < 2; ++i) {
    int l2;
    FMI_CHECK(run_conv(h, h->up_conv[i], X, Y, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE));
    std::swap(X, Y);
    len = l2;
    FMI_CHECK(run_convnext(h, h->up_cnx[i], X, B, L0, len));
  }
  const int up = len / (cu + n);
  FMI_REQUIRE(up * (cu + n) == len && up == c.downsample[0] * c.downsample[1], "upsampler rate");
  FMI_CHECK(copy_cols(h, h->st.z + (int64_t)up * t0, up * cap, X + (int64_t)up * cu, len, up * n, (int64_t)B * L0));
  h->last_z = nullptr;
  return FMI_OK;
}

}  // namespace

extern "C" {

int64_t fmi_dac_arena_bytes(const fmi_dac_config* cfg) {
  if (!cfg || validate(*cfg) != FMI_OK) return -1;
  fmi_dac tmp;
  tmp.cfg = *cfg;
  return build(&tmp);
}

int fmi_dac_create(const fmi_dac_config* cfg, void* arena_dev, int64_t arena_bytes, fmi_dac** out) {
  FMI_REQUIRE(cfg && arena_dev && out, "null argument");
  FMI_CHECK(validate(*cfg));
  fmi_dac* h = new fmi_dac();
  h->cfg = *cfg;
  h->arena = (char*)arena_dev;
  h->arena_bytes = arena_bytes;
  const int64_t need = build(h);
  if (arena_bytes < need) {
    delete h;
    return set_error(FMI_EINVAL, "arena too small: %lld < %lld", (long long)arena_bytes, (long long)need);
  }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess ||
      hipMalloc((void**)&h->f16_ovf, sizeof(int)) != hipSuccess || hipMemset(h->f16_ovf, 0, sizeof(int)) != hipSuccess) {
    delete h;
    return set_error(FMI_EHIP, "stream/event creation failed (no GPU?)");
  }
  hipMemsetAsync(h->arena, 0, (size_t)need, h->stream);  // padded weight tiles must be zero
  hipStreamSynchronize(h->stream);
  *out = h;
  return FMI_OK;
}

void fmi_dac_destroy(fmi_dac* h) {
  if (!h) return;
  hipStreamSynchronize(h->stream);
  for (auto& b : h->buf)
    if (b.p) hipFree(b.p);
  if (h->staging) hipFree(h->staging);
  for (auto p : h->pbuf)
    if (p) hipFree(p);
  free_all_stream_states(h);
  set_f16_overflow_target(nullptr);   // (this thread's launch wrappers must not keep pointing at the freed word)
  if (h->f16_ovf) hipFree(h->f16_ovf);
  hipEventDestroy(h->ev_in);
  hipEventDestroy(h->ev_out);
  hipStreamDestroy(h->stream);
  delete h;
}

int fmi_dac_frame_length(const fmi_dac* h) { return h ? frame_length(h->cfg) : 0; }

int fmi_dac_load_tensor(fmi_dac* h, const char* name_c, const float* src, int ndim, const int64_t* dims,
                        int src_is_device, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && name_c && src && dims && ndim >= 1 && ndim <= 3, "bad argument");
  const std::string name(name_c);
  if (name == "rope_table") {  // bf16 (positions, head_dim) table built by the host with torch
    FMI_REQUIRE(ndim == 2 && dims[1] == h->head_dim && dims[0] <= h->rope_pos, "rope_table shape");
    FMI_CHECK(sync_in(h, stream));
    FMI_CHECK_HIP(hipMemcpyAsync(h->rope, src, (size_t)dims[0] * dims[1] * 2,
                                 src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
    h->rope_loaded = true;
    h->rope_pos = (int)dims[0];
    return sync_out(h, stream);
  }
  auto it = h->reg.find(name);
  if (it == h->reg.end()) return set_error(FMI_EINVAL, "unknown codec tensor '%s'", name.c_str());
  Slot& s = it->second;
  int64_t n = 1, ne = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  for (auto d : s.dims) ne *= d;
  bool ok = n == ne;
  if (ok && s.kind != K_RAW) {  // leading dims must match exactly for re-tiled tensors
    std::vector<int64_t> got(dims, dims + ndim);
    while (got.size() < s.dims.size()) got.push_back(1);
    ok = got == s.dims;
  }
  if (!ok) return set_error(FMI_EINVAL, "%s: unexpected shape (%lld elements, expected %lld)", name.c_str(), (long long)n, (long long)ne);
  FMI_CHECK(sync_in(h, stream));
  hipStream_t st = h->stream;
  const float* dsrc = src;
  if (!src_is_device) {
    FMI_CHECK(ensure_staging(h, (size_t)n * 4));
    FMI_CHECK_HIP(hipMemcpyAsync(h->staging, src, (size_t)n * 4, hipMemcpyHostToDevice, st));
    dsrc = (const float*)h->staging;
  }
  int rc = FMI_OK;
  switch (s.kind) {
    case K_RAW:
      FMI_CHECK_HIP(hipMemcpyAsync(s.dst, dsrc, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
      break;
    case K_CONV:
      rc = launch_pack_conv(dsrc, s.dst, (int)s.dims[0], (int)s.dims[1], (int)s.dims[2], s.cin_pad, s.cout_pad, st);
      break;
    case K_CONVTR:
      rc = launch_pack_convtr(dsrc, s.dst, (int)s.dims[0], (int)s.dims[1], (int)s.dims[2], s.stride, s.cin_pad,
                              s.cout_pad, st);
      break;
    case K_CONV_PART:  // rows [co_off, co_off+cout) of a stacked k=1 weight; padding stays zero (arena memset)
      rc = launch_pack_conv_part(dsrc, s.dst, (int)s.dims[0], (int)s.dims[1], s.cin_pad, s.cout_pad, s.co_off, st);
      break;
  }
  FMI_CHECK(rc);
  FMI_CHECK_HIP(hipStreamSynchronize(st));
  s.loaded = true;
  return sync_out(h, stream);
}

int fmi_dac_finalize_weights(fmi_dac* h, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  for (auto& kv : h->reg)
    if (!kv.second.loaded) return set_error(FMI_ESTATE, "codec tensor '%s' was never loaded", kv.first.c_str());
  FMI_CHECK(sync_in(h, stream));
  hipStream_t s = h->stream;
  const fmi_dac_config& c = h->cfg;
  const int L = c.latent_dim;
  // decode tables: out_proj of every code (from_codes = sum of table rows)
  std::vector<int> off(c.n_codebooks + 2);
  off[0] = 0;
  off[1] = c.semantic_codebook_size;
  for (int i = 1; i <= c.n_codebooks; ++i) off[i + 1] = off[i] + c.codebook_size;
  FMI_CHECK_HIP(hipMemcpyAsync(h->lut_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, s));
  FMI_CHECK(launch_build_lut(h->sem.cb, h->sem.out_w, h->sem.out_b, h->lut, h->sem.n, c.codebook_dim, L, s));
  for (int i = 0; i < c.n_codebooks; ++i)
    FMI_CHECK(launch_build_lut(h->rvq[i].cb, h->rvq[i].out_w, h->rvq[i].out_b, h->lut + (int64_t)off[i + 1] * L,
                               h->rvq[i].n, c.codebook_dim, L, s));
  if (!h->rope_loaded) {  // fallback: host cosf/sinf (the shim normally supplies torch's table)
    std::vector<bf16_t> tab((size_t)h->rope_pos * h->head_dim);
    const int hd = h->head_dim;
    for (int k = 0; k < hd / 2; ++k) {
      const float freq = 1.0f / powf(10000.0f, (float)(2 * k) / (float)hd);
      for (int p = 0; p < h->rope_pos; ++p) {
        tab[((size_t)p * (hd / 2) + k) * 2] = f2bf(cosf((float)p * freq));
        tab[((size_t)p * (hd / 2) + k) * 2 + 1] = f2bf(sinf((float)p * freq));
      }
    }
    FMI_CHECK_HIP(hipMemcpyAsync(h->rope, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, s));
  }
  // bf16 hi/mid/lo planes of the decode-side weights (inside the arena: they travel with the broadcast)
  for (const ConvW& w : h->plane_jobs)
    FMI_CHECK(launch_split_conv_planes(w.w, const_cast<bf16_t*>(w.wb), w.phases * w.taps, w.cin_pad, w.cin_pad16,
                                       w.cout_pad, s));
  FMI_CHECK_HIP(hipStreamSynchronize(s));
  h->ready = true;
  return sync_out(h, stream);
}

int fmi_dac_set_precision(fmi_dac* h, int planes) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(planes >= 0 && planes <= 2, "precision must be 0 (fp32 matrix cores), 1 (bf16, autocast) or 2 (fp16 split)");
  h->decode_planes = planes;
  return FMI_OK;
}

int fmi_dac_fp16_overflow(fmi_dac* h, int* overflowed) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && overflowed, "null argument");
  return read_clear_f16_overflow(h->f16_ovf, overflowed, h->stream);
}

int fmi_dac_set_async(fmi_dac* h, int enable) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  h->async_out = enable != 0;
  return FMI_OK;
}

int fmi_dac_set_background(fmi_dac* h, int lds_floor_bytes) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(lds_floor_bytes >= 0 && lds_floor_bytes <= 160 * 1024, "LDS floor must be in [0, 160 KiB]");
  h->lds_floor = lds_floor_bytes;
  return FMI_OK;
}

int fmi_dac_wait(fmi_dac* h, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_out, 0));
  return FMI_OK;
}

int fmi_dac_synchronize(fmi_dac* h) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  return FMI_OK;
}

// The handle's private stream re-created with a dispatch priority and / or a CU mask: a codec call that runs BESIDE the
// Dual-AR frame loop can be kept off most of the chip (mask) or behind the loop's launches (priority).
int fmi_dac_set_stream_options(fmi_dac* h, int priority, int cu_mask_words, const uint32_t* cu_mask) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_REQUIRE(cu_mask_words >= 0 && (cu_mask_words == 0 || cu_mask), "bad CU mask");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  hipStream_t ns = nullptr;
  if (cu_mask_words > 0) {
    FMI_CHECK_HIP(hipExtStreamCreateWithCUMask(&ns, (uint32_t)cu_mask_words, cu_mask));
  } else {
    int lo = 0, hi = 0;   // lo = least priority (numerically greatest), hi = greatest
    FMI_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int pr = priority < 0 ? hi : (priority > 0 ? lo : (lo + hi) / 2);
    FMI_CHECK_HIP(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, pr));
  }
  hipStreamDestroy(h->stream);
  h->stream = ns;
  return FMI_OK;
}

// as fmi_dualar_weights_ready: the handle's private stream waits for whatever filled the arena on `stream`
int fmi_dac_weights_ready(fmi_dac* h, void* stream) {
  FMI_REQUIRE(h, "null handle");
  std::unique_lock<std::mutex> lock_(h->mu);
  FMI_CHECK(sync_in(h, stream));
  h->ready = true;
  return FMI_OK;
}

int fmi_dac_decode(fmi_dac* h, int64_t* indices_dev, int B, int T, float* audio_out_dev, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && indices_dev && audio_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "codec weights not ready");
  FMI_REQUIRE(B >= 1 && T >= 1, "empty input");
  FMI_CHECK(sync_in(h, stream));
  h->cur_planes = h->decode_planes;
  float *X, *Y;
  int len;
  FMI_CHECK(run_quantizer_decode(h, indices_dev, B, T, &X, &Y, &len));
  FMI_CHECK(run_decoder(h, X, Y, B, len, audio_out_dev));
  return sync_out_data(h, stream);
}

int fmi_dac_context_frames(const fmi_dac* h) { return h ? cdiv(decoder_context_cols(h->cfg), 4) : 0; }

int fmi_dac_decode_tail(fmi_dac* h, int64_t* indices_dev, int B, int T, int t0, float* audio_out_dev, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && indices_dev && audio_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "codec weights not ready");
  FMI_REQUIRE(B >= 1 && T >= 1 && t0 >= 0 && t0 < T, "bad frame range [%d, %d)", t0, T);
  const fmi_dac_config& c = h->cfg;
  const int L0 = c.latent_dim;
  FMI_CHECK(sync_in(h, stream));
  h->cur_planes = h->decode_planes;
  hipStream_t s = h->stream;
  float *X, *Y;
  int len;
  // the quantizer side (LUT + 8-layer windowed transformer + x4 upsampler, 5 % of the per-frame work) is
  // recomputed over all frames so far: its receptive field (layers x window = 1016 frames) covers any utterance
  FMI_CHECK(run_quantizer_decode(h, indices_dev, B, T, &X, &Y, &len));
  const int ctx_frames = std::min(t0, cdiv(decoder_context_cols(c), 4));
  const int col_lo = 4 * (t0 - ctx_frames), w = len - col_lo;
  // crop z to [col_lo, 4T): rows are (b, channel)
  FMI_CHECK_HIP(hipMemcpy2DAsync(X, (size_t)w * 4, h->buf[5].p + col_lo, (size_t)len * 4, (size_t)w * 4,
                                 (size_t)B * L0, hipMemcpyDeviceToDevice, s));
  FMI_CHECK(run_decoder(h, X, Y, B, w, audio_out_dev, 4 * ctx_frames));
  return sync_out_data(h, stream);
}

int fmi_dac_decode_tail_cached(fmi_dac* h, int64_t* indices_dev, int B, int T, int t0, int64_t stream_id,
                               float* audio_out_dev, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && indices_dev && audio_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "codec weights not ready");
  FMI_REQUIRE(B >= 1 && T >= 1 && t0 >= 0 && t0 < T, "bad frame range [%d, %d)", t0, T);
  const fmi_dac_config& c = h->cfg;
  const int L0 = c.latent_dim, C = h->post.dim, up = c.downsample[0] * c.downsample[1];
  FMI_CHECK(sync_in(h, stream));
  h->cur_planes = h->decode_planes;
  hipStream_t s = h->stream;
  // continue the state of this stream's previous call, or start over (first call, other batch, a gap, buffers too small)
  FMI_CHECK(select_stream_state(h, stream_id));
  int from = t0;
  if (!(h->st.B == B && h->st.id == stream_id && h->st.planes == h->cur_planes && h->st.T == t0 && t0 > 0 &&
        T <= h->st.cap)) {
    from = 0;
    if (h->st.B != B || T > h->st.cap) {
      FMI_CHECK_HIP(hipStreamSynchronize(s));
      free_stream_state(h);
      int cap = 1024;
      while (cap < T) cap *= 2;
      h->st.qkv.assign(h->post.layers.size(), nullptr);
      for (auto& p : h->st.qkv) FMI_CHECK_HIP(hipMalloc((void**)&p, (size_t)B * 3 * C * cap * 4));
      FMI_CHECK_HIP(hipMalloc((void**)&h->st.tf_out, (size_t)B * C * cap * 4));
      FMI_CHECK_HIP(hipMalloc((void**)&h->st.z, (size_t)B * L0 * up * cap * 4));
      h->st.B = B;
      h->st.cap = cap;
    }
  }
  h->st.T = 0;   // invalid until this call has succeeded
  const bool stream_dec = decoder_streams(h);
  if (stream_dec && h->st.dh[0].empty()) FMI_CHECK(alloc_dec_halos(h, h->st, B));
  const bool halos_ok = stream_dec && from == t0 && t0 > 0 && h->st.dec_valid;   // the kept tails are those of frames [0, t0)
  h->st.dec_valid = false;
  FMI_CHECK(run_quantizer_decode_inc(h, indices_dev, B, T, from));
  // left context: none with the kept tails (every conv continues where the previous chunk ended); otherwise -- first call
  // of a stream at t0 > 0, a dropped state -- the receptive field is recomputed from zero tails and its audio discarded
  const int ctx_frames = halos_ok ? 0 : std::min(t0, cdiv(decoder_context_cols(c), up));
  const int col_lo = up * (t0 - ctx_frames), w = up * T - col_lo;
  if (stream_dec) {
    if (!halos_ok) FMI_CHECK(zero_dec_halos(h, h->st, B));
    const int zl = halos_ok ? std::min(6, col_lo) : 0;   // the first conv reads its context straight from the kept latents
    FMI_CHECK(ensure_buf(h, 0, (int64_t)B * decode_peak_elems(c, cdiv(w, up) + 8)));
    FMI_CHECK(ensure_buf(h, 1, (int64_t)B * L0 * (w + 8)));
    float *X = h->buf[0].p, *Zin = h->buf[1].p;
    FMI_CHECK(copy_cols(h, Zin, zl + w, h->st.z + col_lo - zl, up * h->st.cap, zl + w, (int64_t)B * L0));
    FMI_CHECK(run_decoder_stream(h, h->st, Zin, zl, X, B, w, audio_out_dev, up * ctx_frames));
    h->st.dec_valid = true;
  } else {
    float *X = h->buf[0].p, *Y = h->buf[1].p;
    FMI_CHECK(copy_cols(h, X, w, h->st.z + col_lo, up * h->st.cap, w, (int64_t)B * L0));
    FMI_CHECK(run_decoder(h, X, Y, B, w, audio_out_dev, up * ctx_frames));
  }
  h->st.T = T;
  h->st.id = stream_id;
  h->st.planes = h->cur_planes;
  return sync_out_data(h, stream);
}

int fmi_dac_stream_close(fmi_dac* h, int64_t stream_id) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  if (h->st.id == stream_id && h->st.T > 0) {   // the active state: its buffers stay for the next stream, its content is void
    h->st.T = 0;
    h->st.id = 0;
    return FMI_OK;
  }
  for (size_t i = 0; i < h->parked.size(); ++i)
    if (h->parked[i].id == stream_id) {
      FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
      free_one_state(h->parked[i]);
      h->parked.erase(h->parked.begin() + i);
      break;
    }
  return FMI_OK;   // (an unknown id is not an error: the state may have been dropped as least recently used)
}

int fmi_dac_stream_reset(fmi_dac* h) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h, "null handle");
  FMI_CHECK_HIP(hipStreamSynchronize(h->stream));
  free_all_stream_states(h);
  return FMI_OK;
}

int fmi_dac_decode_latent(fmi_dac* h, const float* z_dev, int B, int L, float* audio_out_dev, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && z_dev && audio_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "codec weights not ready");
  FMI_REQUIRE(B >= 1 && L >= 1, "empty input");
  const fmi_dac_config& c = h->cfg;
  FMI_CHECK(sync_in(h, stream));
  h->cur_planes = h->decode_planes;
  // L latent frames = L/4 code frames worth of decoder work
  const int64_t peak = (int64_t)B * decode_peak_elems(c, cdiv(L, 4));
  FMI_CHECK(ensure_buf(h, 0, peak));
  FMI_CHECK(ensure_buf(h, 1, peak));
  FMI_CHECK_HIP(hipMemcpyAsync(h->buf[0].p, z_dev, (size_t)B * c.latent_dim * L * 4, hipMemcpyDeviceToDevice, h->stream));
  FMI_CHECK(run_decoder(h, h->buf[0].p, h->buf[1].p, B, L, audio_out_dev));
  return sync_out_data(h, stream);
}

int fmi_dac_debug_z(fmi_dac* h, float** z_dev, int* C, int* L) {
  FMI_REQUIRE(h && h->last_z, "no decode has run");
  if (z_dev) *z_dev = h->last_z;
  if (C) *C = h->last_zC;
  if (L) *L = h->last_zL;
  return FMI_OK;
}

int fmi_dac_encode(fmi_dac* h, const float* audio_dev, int B, int N, int64_t* indices_out_dev, void* stream) {
  std::unique_lock<std::mutex> lock_;
  if (h) lock_ = std::unique_lock<std::mutex>(h->mu);
  FMI_REQUIRE(h && audio_dev && indices_out_dev, "null argument");
  FMI_REQUIRE(h->ready, "codec weights not ready");
  const fmi_dac_config& c = h->cfg;
  const int fl = frame_length(c);
  FMI_REQUIRE(B >= 1 && N >= fl && N % fl == 0, "audio length %d must be a positive multiple of frame_length %d", N, fl);
  FMI_CHECK(sync_in(h, stream));
  h->cur_planes = 0;   // encode: fp32 matrix cores (the codes are compared bit for bit with the reference's)
  hipStream_t s = h->stream;
  // peak activation: the block-1 residual units run at full rate on encoder_dim channels
  int64_t peak = (int64_t)c.encoder_dim * N;
  {
    int d = c.encoder_dim;
    int64_t len = N;
    for (int i = 0; i < 4; ++i) {
      d *= 2;
      peak = std::max(peak, (int64_t)(d / 2) * len);
      len /= c.encoder_rates[i];
      peak = std::max(peak, (int64_t)d * len);
    }
  }
  FMI_CHECK(ensure_buf(h, 0, (int64_t)B * peak));
  FMI_CHECK(ensure_buf(h, 1, (int64_t)B * peak));
  float *X = h->buf[0].p, *Y = h->buf[1].p;
  int len = N;
  FMI_CHECK(launch_first_conv(audio_dev, h->first_w, h->first_b, X, B, c.encoder_dim, len, s));
  int d = c.encoder_dim;
  for (const EncBlock& e : h->enc) {
    d *= 2;
    for (int r = 0; r < 3; ++r) FMI_CHECK(run_res_unit(h, e.ru[r], X, Y, B, len));
    int l2;
    FMI_CHECK(run_conv(h, e.down, X, Y, B, len, &l2, e.alpha, nullptr, nullptr, ACT_NONE));
    std::swap(X, Y);
    len = l2;
    if (e.has_tf) FMI_CHECK(run_transformer(h, e.tf, X, X, B, len));
  }
  FMI_CHECK(run_conv(h, h->enc_out, X, Y, B, len, nullptr, h->enc_alpha, nullptr, nullptr, ACT_NONE));
  std::swap(X, Y);
  // quantizer.forward up to the codes (rvq.py:293-316)
  const int L0 = c.latent_dim;
  for (int i = 0; i < 2; ++i) {
    int l2;
    FMI_CHECK(run_conv(h, h->down_conv[i], X, Y, B, len, &l2, nullptr, nullptr, nullptr, ACT_NONE));
    std::swap(X, Y);
    len = l2;
    FMI_CHECK(run_convnext(h, h->down_cnx[i], X, B, L0, len));
  }
  FMI_CHECK(run_transformer(h, h->pre, X, X, B, len));
  VqArgs v{};
  v.residual = X; v.B = B; v.C = L0; v.T = len; v.d = c.codebook_dim; v.books = c.n_codebooks + 1;
  v.codes = indices_out_dev;
  auto step = [&](const VQ& q, int book) {
    v.in_w = q.in_w; v.in_b = q.in_b; v.codebook = q.cb; v.out_w = q.out_w; v.out_b = q.out_b; v.n = q.n;
    v.book = book;
    return launch_vq_step(v, s);
  };
  FMI_CHECK(step(h->sem, 0));
  for (int i = 0; i < c.n_codebooks; ++i) FMI_CHECK(step(h->rvq[i], i + 1));
  return sync_out_data(h, stream);
}

}  // extern "C"
