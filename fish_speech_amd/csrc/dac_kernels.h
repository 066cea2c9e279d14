// dac_kernels.h -- launch wrappers of the codec kernels (dac_kernels.hip).  All fp32, channel-major
// activations [B][C][L] (time contiguous), like the reference's (B, C, T) tensors.
#pragma once
#include "common.h"

namespace fmi {

// Packed weight layout of one conv / linear / transposed conv, built at load time:
//   w[phase][tap][ci_pad/8][co_pad][8]  (zero padded; ci_pad % 8 == 0, co_pad % 32 == 0; the two 16-byte
//   halves of a [8] row are swapped on rows co with bit 3 set -- conv_w_index in dac_kernels.hip)
// regular conv:      phases = 1, taps = K
// transposed (k=2s): phases = s, taps = 2   (tap m reads x[q - m], weight index j + m*s)
// transposed (k=s):  phases = s, taps = 1
struct ConvW {
  const float* w = nullptr;
  const float* bias = nullptr;  // [cout] or nullptr
  int cin = 0, cout = 0, cin_pad = 0, cout_pad = 0;
  int phases = 1, taps = 1;
  // the same weights as 16-bit operand planes of the bf16 / fp16 matrix cores (launch_split_conv_planes):
  //   wb[phase][tap][ci_pad16/16][slot][co_pad][16]   (16-byte halves of a [16] row swapped on rows with bit 3 set)
  //   slot 0 = bf16(w); slots 1, 2 = the fp16 split f16(w), f16((w - hi) * 2048)
  // nullptr = this layer only has the fp32 path
  const bf16_t* wb = nullptr;
  int cin_pad16 = 0;
};

enum ConvAct { ACT_NONE = 0, ACT_GELU = 1 };

struct ConvArgs {
  ConvW w;
  const float* x;       // [B][cin][lin]
  float* out;           // [B][cout][lout]
  const float* snake_alpha;  // [cin] fused Snake1d on the input, or nullptr
  const float* res;     // [B][cout][lout] residual or nullptr
  const float* gamma;   // [cout] scale applied before the residual add (LayerScale / ConvNeXt gamma)
  int B, lin, lout;
  int x_stride;         // input column step per output column (conv stride; 1 for transposed)
  int tap_step;         // input column step per tap (dilation; -1 for transposed)
  int tap_base;         // input column of tap 0 for output column 0 (= -(left pad); 0 for transposed)
  int out_stride;       // output column step per computed column (1; stride for transposed)
  int act;
  // arithmetic of the contraction: 0 = fp32 matrix cores (v_mfma_f32_32x32x2_f32, an fmaf chain);
  // 2 = fp16 matrix cores on a two-term split of both operands with a 2^11-scaled low part: three products, two fp32
  //     accumulators (fp32-class: what is dropped is below 2^-22 of a product);
  // 1 = bf16 matrix cores, operands and result rounded to bf16 (what torch.autocast(bfloat16) makes of a conv / linear)
  int planes;
  // bf16-plane kernel only: activations handed from conv to conv as ready operand planes, so that Snake and the
  // three-way split are computed ONCE by the producer's epilogue instead of by every consumer work-group's staging:
  //   xp    input planes [B][cin/16][planes][lin][16] (replaces x / snake_alpha; cin % 16 == 0), or nullptr
  //   outp  if set, the epilogue also writes planes of snake(out, next_alpha) ([B][cout/16][planes][lout][16]);
  //         `out` may then be nullptr when nobody reads the fp32 tensor
  const bf16_t* xp;
  bf16_t* outp;
  const float* next_alpha;
  // Streaming decode (round 6, bf16-plane kernel only): buffers wider than the logical lengths, and real data LEFT of
  // input column 0 -- the tail of the previous chunk that a causal conv's first outputs read (k_eff - 1 columns; one for
  // the k = 2s transposed convs) -- instead of zero padding.  All pointers address logical column 0.  0 = as before.
  int x_ld;      // columns per row of x / xp (0: lin)
  int x_left;    // input columns -x_left .. -1 exist and hold data (0: columns < 0 are padding)
  int out_ld;    // columns per row of out / res (0: lout)
  int outp_ld;   // columns per row of outp (0: lout)
  // device word raised when an operand of the fp16 split leaves the fp16 range (owned by the codec handle);
  // nullptr = the one named by set_f16_overflow_target() on this host thread
  int* ovf;
};
// out[b][co][q*out_stride + phase] = res + gamma * act(bias + sum_ci sum_tap w[phase][tap][ci][co] *
//                                    snake(x)[b][ci][q*x_stride + tap_base + tap*tap_step])
int launch_conv(const ConvArgs& a, hipStream_t s);

// fp32 packed weights (layout above) -> the three bf16 planes of ConvW::wb
int launch_split_conv_planes(const float* w_packed, bf16_t* wb, int tapgroups /*phases*taps*/, int cin_pad,
                             int cin_pad16, int cout_pad, hipStream_t s);

// weight re-layouts (run once at load)
int launch_pack_conv(const float* w_src /*[cout][cin][k]*/, float* dst, int cout, int cin, int k, int cin_pad,
                     int cout_pad, hipStream_t s);
int launch_pack_conv_part(const float* w_src /*[cout][cin]*/, float* dst, int cout, int cin, int cin_pad, int cout_pad,
                          int co_off, hipStream_t s);
int launch_pack_convtr(const float* w_src /*[cin][cout][k]*/, float* dst, int cin, int cout, int k, int stride,
                       int cin_pad, int cout_pad, hipStream_t s);

// elementwise / small kernels
int launch_rmsnorm_cols(const float* x, const float* w, float eps, float* out, int B, int C, int L, hipStream_t s);
int launch_layernorm_cols(const float* x, const float* w, const float* b, float eps, float* out, int B, int C, int L,
                          hipStream_t s);
int launch_dwconv7(const float* x, const float* w /*[C][7]*/, const float* b, float* out, int B, int C, int L,
                   hipStream_t s);
int launch_silu_mul(const float* ab /*[B][2F][L]*/, float* out /*[B][F][L]*/, int B, int F, int L, hipStream_t s);
int launch_rope_cols(float* qkv /*[B][3C][L]*/, const bf16_t* table /*[L][hd/2][2]*/, int B, int C, int L, int hd,
                     hipStream_t s);
// ld: row stride of qkv (0 = L); only queries q_lo..L-1 are computed, out is the compact [B][C][L - q_lo]
int launch_window_attn(const float* qkv /*[B][3C][ld]*/, float* out /*[B][C][L - q_lo]*/, int B, int C, int L, int hd,
                       int window, hipStream_t s, int ld = 0, int q_lo = 0, bf16_t* outp = nullptr);
// operand-plane producers / consumer of the codec transformer (fp16 split arithmetic, C % 16 == 0):
//   planes [B][C/16][2][L][16] fp16 (hi, lo * 2^11), the layout of ConvArgs::xp
int launch_rmsnorm_cols_planes(const float* x, const float* w, float eps, bf16_t* outp, int B, int C, int L, hipStream_t s);
int launch_silu_mul_planes(const float* ab /*[B][2F][L]*/, bf16_t* outp /*planes of [B][F][L]*/, int B, int F, int L, hipStream_t s);
// out = res + gamma * act(W x + bias) for a k = 1 layer, x given as operand planes; every wave fetches both MFMA operands
// straight from L2 (no LDS staging): bit-identical to launch_conv with planes = 2 on the fp32 tensor
int launch_linear_planes(const ConvW& w, const bf16_t* xp, float* out, const float* res, const float* gamma, int act,
                         int B, int L, hipStream_t s);
int launch_lut_decode(const int64_t* idx /*[B][1+n][T]*/, const float* tables, const int* table_rows_off,
                      int n_books, int sem_size, int cb_size, float* out /*[B][C][T]*/, int B, int C, int T,
                      hipStream_t s);
int launch_clamp_indices(int64_t* idx, int B, int n_books1, int T, int sem_size, int cb_size, hipStream_t s);
int launch_build_lut(const float* codebook /*[n][d]*/, const float* w /*[C][d]*/, const float* bias, float* table,
                     int n, int d, int C, hipStream_t s);
// Streaming decode (round 6): a buffer row is [h halo columns | n new columns] (row stride h + n elements of ELEM bytes;
// `buf` addresses the row's first halo column).  Writes the previous chunk's tail (old_state, [rows][h]) into the halo and
// keeps the tail of [halo | new] -- the last h columns -- in new_state for the next chunk.  elem_bytes: 32 (one column of
// an operand plane: 16 x 16 bit) or 4 (fp32).
int launch_halo_swap(void* buf, const void* old_state, void* new_state, int64_t rows, int h, int n, int elem_bytes,
                     hipStream_t s);
int launch_final_conv_tanh(const float* x /*[B][C][L]*/, const float* alpha, const float* w /*[C][7]*/,
                           const float* bias, float* out /*[B][1][L-col0]*/, int B, int C, int L, int col0,
                           hipStream_t s);
int launch_first_conv(const float* x /*[B][1][L]*/, const float* w /*[C][7]*/, const float* bias, float* out,
                      int B, int C, int L, hipStream_t s);
struct VqArgs {
  float* residual;        // [B][C][T], updated in place (residual -= z_q_i)
  const float* in_w;      // [d][C]
  const float* in_b;      // [d]
  const float* codebook;  // [n][d]
  const float* out_w;     // [C][d]
  const float* out_b;     // [C]
  int64_t* codes;         // [B][books][T], this call writes book `book`
  int B, C, T, d, n, book, books;
};
int launch_vq_step(const VqArgs& a, hipStream_t s);

// Sticky fp16-split overflow flag, one device word per codec handle.  set_f16_overflow_target names the word the launch
// wrappers of THIS host thread hand to their kernels (an entry point sets it under its handle's mutex; nullptr = a
// process-wide word nobody reads); read_clear waits for `s`, returns the word and clears it.
void set_f16_overflow_target(int* dev_word);
// floor (bytes, <= 160 KiB) under the dynamic LDS of this host thread's conv_mfma_bf16 launches; 0 = none
void set_conv_lds_floor(int bytes);
int read_clear_f16_overflow(int* dev_word, int* flag, hipStream_t s);

}  // namespace fmi
