/* fishmi.h -- C ABI of libfishmi.so: the MI355X (gfx950) hot path of fish-speech S2 inference.
 *
 * The reference (fishaudio/fish-speech) has no FFI: its seams are Python duck-typing
 * (SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  All pointers marked `dev` are device
 * (HBM) addresses, e.g. torch `tensor.data_ptr()`; `stream` is a hipStream_t passed as
 * void* (torch.cuda.current_stream().cuda_stream); 0 = the null stream.  Every function
 * returns 0 on success or a negative FMI_E* code; fmi_last_error() gives the message
 * (thread-local).  A handle is owned by one host thread at a time, like the reference's
 * single llama worker thread (fish_speech/models/text2semantic/inference.py:748-799).
 */
#ifndef FISHMI_H
#define FISHMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMI_OK 0
#define FMI_EINVAL -1   /* bad argument / unsupported shape */
#define FMI_EHIP -2     /* HIP runtime error */
#define FMI_ESTATE -3   /* call out of order (weights not loaded, caches not set up ...) */
#define FMI_ENOMEM -4   /* arena / page pool exhausted */

int fmi_version(void);
const char* fmi_last_error(void);
/* gfx arch name of the current device ("gfx950") into buf. */
int fmi_device_arch(char* buf, size_t n);

/* ------------------------------------------------------------------ Dual-AR decoder ---- */

/* Mirrors the fields of DualARModelArgs that inference reads
 * (fish_speech/models/text2semantic/llama.py:27-193). */
typedef struct fmi_dualar_config {
  int32_t vocab_size, n_layer, n_head, n_local_heads, head_dim, dim, intermediate_size;
  int32_t n_fast_layer, fast_dim, fast_n_head, fast_n_local_heads, fast_head_dim,
      fast_intermediate_size;
  int32_t codebook_size, num_codebooks;
  int32_t semantic_begin_id, semantic_end_id, im_end_id;
  int32_t max_seq_len;
  int32_t attention_qk_norm, fast_attention_qk_norm;
  int32_t scale_codebook_embeddings, norm_fastlayer_input;
  float rope_base, norm_eps;
  int32_t weight_int8; /* 1: weight-only int8 checkpoint (tools/llama/quantize.py:186-229): every nn.Linear carries
                          an int8 weight + per-row bf16 scales; the arena then also holds the int8 tiles the
                          decode GEMV streams (half the bytes) */
} fmi_dualar_config;

typedef struct fmi_dualar fmi_dualar;

/* Bytes of device memory the packed weight arena needs for `cfg`. */
int64_t fmi_dualar_arena_bytes(const fmi_dualar_config* cfg);

/* Create a model over a caller-owned arena (one contiguous device allocation, so that a
 * single RCCL broadcast replicates the weights: SURVEY.md 8e).  Replaces
 * DualARTransformer.__init__ (llama.py:660-706). */
int fmi_dualar_create(const fmi_dualar_config* cfg, void* arena_dev, int64_t arena_bytes,
                      fmi_dualar** out);
void fmi_dualar_destroy(fmi_dualar* h);

/* Load one checkpoint tensor (row-major bf16, device or host pointer) by its state-dict
 * name after the reference's key remap (llama.py:229-246), e.g.
 * "layers.3.attention.wqkv.weight".  The library re-tiles it into the arena.  Replaces
 * load_state_dict in BaseTransformer.from_pretrained (llama.py:480-594). */
int fmi_dualar_load_tensor(fmi_dualar* h, const char* name, const void* src, int64_t rows,
                           int64_t cols, int src_is_device, void* stream);
/* Load one quantised linear of a weight-only-int8 checkpoint: `weight_i8` (rows, cols) int8 row-major and its
 * `scales_bf16` (rows,), as WeightOnlyInt8QuantHandler.create_quantized_state_dict writes them
 * (tools/llama/quantize.py:186-202).  name = the ".weight" key.  Arithmetic of the layer:
 * bf16(bf16(x @ W_int8^T) * scales) (WeightOnlyInt8Linear.forward, quantize.py:228-229). */
int fmi_dualar_load_tensor_int8(fmi_dualar* h, const char* name, const void* weight_i8, const void* scales_bf16,
                                int64_t rows, int64_t cols, int src_is_device, void* stream);
/* Call once after all tensors are loaded (rank 0 before the broadcast) to build derived
 * tables inside the arena (live LM-head rows, RoPE tables). */
int fmi_dualar_finalize_weights(fmi_dualar* h, void* stream);
/* Call on every rank whose arena was filled from outside (a broadcast, a device copy) instead of load_tensor +
 * finalize.  `stream` = the stream that work was enqueued on: the handle's private stream is made to wait for it
 * (an event, no host wait), so the derived tables the first prefill builds from the arena never read it early and
 * the caller needs no device synchronize between the collective and the first call.  Replaces nothing upstream:
 * the reference reads the checkpoint on every rank (tools/vqgan/extract_vq.py:161-207 starts one process per GPU). */
int fmi_dualar_weights_ready(fmi_dualar* h, void* stream);

/* KV-cache page pool + per-slot state for up to max_batch concurrent utterances of up to
 * max_seq_len positions.  Replaces DualARTransformer.setup_caches (llama.py:307-324,708-722). */
int fmi_dualar_setup_caches(fmi_dualar* h, int max_batch, int max_seq_len);

typedef struct fmi_sampling {
  float temperature, top_p; /* rounded to bf16 inside, like inference.py:305-306 */
  int32_t top_k;
  uint32_t seed;    /* counter-based uniform generator keyed by (seed, frame, draw, vocab id): slot-independent */
  int32_t use_ras;  /* 1 = repetition-aware sampling of inference.py:118-144 */
} fmi_sampling;

/* Start n utterances in slots slot_ids[i].  tokens_dev: int32, concatenated per utterance,
 * each (T_i, 1+num_codebooks) row-major (position-major); lens[i] = T_i (host array).
 * Runs the prompt through the slow transformer, then the first frame
 * (decode_one_token_ar with previous_tokens=None, inference.py:324-334).
 * max_new[i] bounds the pages reserved for the slot.  Replaces the prefill half of
 * generate() (inference.py:243-334). */
int fmi_dualar_prefill(fmi_dualar* h, int n, const int32_t* slot_ids, const int32_t* tokens_dev,
                       const int32_t* lens, const int32_t* max_new, const fmi_sampling* samp,
                       void* stream);

/* Prefix-KV reuse across the text chunks of generate_long (the reference re-prefills the whole conversation for every
 * chunk, inference.py:620-688): the slots still hold the K/V of positions [0, pos0[i]) from an earlier prefill of the
 * SAME tokens (not released since); only the lens[i] new columns in tokens_dev are run, at positions pos0[i].., with
 * attention over the cached prefix.  Bit-identical to fmi_dualar_prefill of the whole prompt as long as the reused
 * positions were themselves written by a prefill (the suffix rows are forced through the prefill kernels). */
int fmi_dualar_prefill_resume(fmi_dualar* h, int n, const int32_t* slot_ids, const int32_t* tokens_dev,
                              const int32_t* lens, const int32_t* pos0, const int32_t* max_new,
                              const fmi_sampling* samp, void* stream);

/* Advance the given slots by n_frames frames (one hipGraph replay per frame; no host sync
 * inside).  Replaces decode_n_tokens (inference.py:184-238) for a batch. A slot that emitted
 * <|im_end|> or exhausted its reservation stops advancing.
 * Ordering: the frames run after everything already enqueued on `stream` and before every later call on this handle;
 * `stream` itself is NOT made to wait for them (a cross-queue wait that stays pending for the length of the frame loop
 * slows every dispatch of the loop: +0.3 ms per frame measured).  A caller that consumes the frames on its own stream
 * (fmi_dualar_out_ptr) calls fmi_dualar_wait(h, stream) first; fmi_dualar_poll_done / _read / _synchronize wait on the
 * host. */
int fmi_dualar_decode(fmi_dualar* h, int n, const int32_t* slot_ids, int n_frames, void* stream);
int fmi_dualar_wait(fmi_dualar* h, void* stream);   /* order `stream` after the frames of the last decode call */
int fmi_dualar_synchronize(fmi_dualar* h);          /* host wait for everything enqueued on this handle */

/* Copy out what a slot generated so far: frames (n_frames, 1+num_codebooks) int32 into
 * out_host (capacity max_frames); *n_frames_out = count, *done_out = 1 if the slot ended
 * with <|im_end|>.  Synchronises the stream. */
int fmi_dualar_read(fmi_dualar* h, int slot, int32_t* out_host, int max_frames, int* n_frames_out,
                    int* done_out, void* stream);
/* Non-blocking-ish poll of done flags for slots (host array of n ints). Synchronises. */
int fmi_dualar_poll_done(fmi_dualar* h, int n, const int32_t* slot_ids, int32_t* done_host,
                         void* stream);
int fmi_dualar_release(fmi_dualar* h, int slot);
/* Device pointer of the generated-frames buffer, int32 [max_batch][max_frames][1+num_codebooks]
 * (slot-major), so that a consumer on the same GPU (the codec) reads the codes without a host copy.  Frames of a
 * decode call still in flight: order the consumer's stream with fmi_dualar_wait, or wait with fmi_dualar_synchronize. */
int fmi_dualar_out_ptr(fmi_dualar* h, void** out_dev, int* max_frames);

/* Drop-in single-step seam = the `decode_one_token` callable
 * (decode_one_token_ar, inference.py:96-181) for slot 0 semantics of the reference
 * (batch 1).  x_dev: int32 (S, 1+ncb) position-major; pos0 = input_pos[0]; S>1 is a
 * prefill call.  prev_dev: int32 (1+ncb, 10) RAS window or NULL.  out_dev: int32 (1+ncb).
 * logits_out_dev (optional, bf16, n_live) / hidden_out_dev (optional, bf16, dim) expose
 * forward_generate's results (llama.py:390-466) for the parity tests. */
int fmi_dualar_step(fmi_dualar* h, int slot, const int32_t* x_dev, int S, int pos0,
                    const fmi_sampling* samp, const int32_t* prev_dev, int32_t frame_index,
                    int32_t* out_dev, void* stream);

/* The model-object seam of SURVEY.md 8b, for callers that keep the reference's own
 * decode_one_token_ar: BaseTransformer.forward_generate (llama.py:390-466, slow transformer + final norm
 * + tied head; logits bf16 over the n_live constrained rows -- every other vocabulary row is -inf after
 * semantic_logit_bias, inference.py:310-320 -- hidden bf16 [fast_dim]: DualARTransformer.forward_generate's
 * fast_project_in Linear(dim, fast_dim) with bias is applied when fast_dim != dim, llama.py:665-668,827) and
 * DualARTransformer.forward_generate_fast (llama.py:799-817; hidden bf16 [fast_dim] at codebook
 * position pos -> logits bf16 [codebook_size]).  Batch 1 like the reference. */
int fmi_dualar_forward_slow(fmi_dualar* h, int slot, const int32_t* x_dev, int S, int pos0,
                            void* logits_out_dev, void* hidden_out_dev, void* stream);
int fmi_dualar_forward_fast(fmi_dualar* h, int slot, const void* hidden_in_dev, int pos,
                            void* logits_out_dev, void* stream);
/* Row-major tables inside the arena for host-side lookups: which = 0 fast_embeddings
 * (codebook_size x fast_dim bf16), 1 = vocab id of each live logits row (int32 [n_live]). */
int fmi_dualar_table_ptr(fmi_dualar* h, int which, void** ptr, int* rows, int* cols);

/* What the handle derived from the arena content on THIS process (ranks > 0 receive the arena by
 * broadcast and never see load_tensor / finalize): the number of row-balanced decode copies, the rows of the fast
 * layer-0 q|k|v table (0 = not built), and how many tensors went through fmi_dualar_load_tensor* on this handle
 * (0 on a rank that was fed by fmi_dualar_weights_ready alone).  Replaces nothing upstream: the reference loads the
 * checkpoint on every rank (tools/vqgan/extract_vq.py:161-207). */
int fmi_dualar_derived_info(fmi_dualar* h, int* row_copies, int* table_rows, int* loaded_tensors);

/* Debug / parity taps (device pointers owned by the library, valid until the next call):
 * live-row logits of the last slow step (bf16, [B][n_live_padded]), the vocab id of each
 * live row (int32 [n_live]), the hidden rows handed to the fast transformer (bf16 [B][fast_dim]: the normed hidden,
 * projected when fast_dim != dim) and the fast logits of the
 * last fast step (bf16 [B][codebook_size]). */
int fmi_dualar_debug_ptrs(fmi_dualar* h, void** slow_logits, int* n_live, int* ld_logits,
                          void** live_ids, void** hidden, void** fast_logits);
/* If enabled, every fast step's logits are also copied to a trace buffer
 * [B][num_codebooks][codebook_size] (bf16) returned here.  enable = 1: traced frames run the plain GEMV path (no
 * tabulated layer-0 q|k|v); 2: traced frames keep the frame loop's own path (table in use).  Eager either way. */
int fmi_dualar_set_trace(fmi_dualar* h, int enable, void** fast_trace);
/* Test seam (float parity of the fast chain at every codebook position, on the frame loop's own path): steps 6-8 of
 * decode_one_token_ar (fish_speech/models/text2semantic/inference.py:148-176 -- forward_generate_fast at position 0 on
 * the hidden state, then one position per codebook) for B slots (host array slot_ids) from the NORMED hidden rows
 * `hidden_normed_dev` (bf16 [B][dim], what forward_generate returns, llama.py:459-461), executed exactly as a decode
 * frame executes them (batch GEMV, merged positions 0/1 per fmi_dualar_set_fast_merge, the tabulated layer-0 q|k|v
 * when table != 0), with every draw REPLACED by `forced_dev` (int32 [B][1 + num_codebooks]: slow token id, then the
 * codes).  fast_logits_out_dev: bf16 [B][num_codebooks][codebook_size], rows 1.. = forward_generate_fast's logits at
 * positions 1..num_codebooks-1 (position 0's are never computed: the reference discards them).  The slots must have
 * been prefilled (their sampling parameters are read); their frame counters advance as after a decode frame. */
int fmi_dualar_fast_chain_forced(fmi_dualar* h, int B, const int32_t* slot_ids, const void* hidden_normed_dev,
                                 const int32_t* forced_dev, int table, void* fast_logits_out_dev, void* stream);
/* 1 = keep generating past <|im_end|> (fixed-length synthetic benchmarks). */
int fmi_dualar_set_ignore_eos(fmi_dualar* h, int enable);
/* Disable hipGraph replay (eager launches) -- used by tests/profiling. */
int fmi_dualar_set_graph(fmi_dualar* h, int enable);
/* Prefill attention implementation (F.scaled_dot_product_attention, llama.py:928-934): 1 = MFMA flash attention with
 * LDS-staged K/V tiles (default), 0 = the VALU kernel of round 1 (kept for A/B parity runs). */
int fmi_dualar_set_attn_impl(fmi_dualar* h, int impl);
/* Fast transformer positions 0 and 1 of a frame (the two forward_generate_fast calls of inference.py:148-149 and :166,
 * whose inputs are both known once the slow token is drawn): 1 = one pass over the fast weights with 2 x batch rows
 * (default when 2 x batch <= 32, bf16 weights, fast_dim == dim), 0 = two passes (rounds 1-3; kept for A/B parity runs).
 * Results are bit-identical either way, tokens and float taps (tests/test_s2_parity_gpu.py:
 * test_merged_fast_positions_equal_the_two_pass_path_bit_for_bit_at_batch_5_and_8). */
int fmi_dualar_set_fast_merge(fmi_dualar* h, int enable);
/* Dispatch priority of the handle's private stream (-1 highest, 0 default, 1 lowest); the stream is re-created, graphs
 * are re-captured.  For a frame loop that shares the GPU with another queue (the codec decode of the previous batch:
 * fmi_dac_set_async / fmi_dac_set_stream_options).  The reference has one CUDA stream and nothing to overlap
 * (fish_speech/inference_engine/__init__.py:73-140 decodes a segment between two generate calls). */
int fmi_dualar_set_stream_priority(fmi_dualar* h, int priority);
/* Decode attention (the per-frame step of llama.py:910-934 over the KV cache): rows whose position is >= threshold run
 * on the MFMA kernel (all query heads of a kv head in one work-group, key ranges split over work-groups, partial
 * softmax states merged), the others on the fused VALU kernel; which one depends only on the row's own position.
 * Default 1024, the measured break-even at batch 8 (FMI_ATTN_THR overrides at setup_caches); 0 = VALU kernel for every
 * row.  Call after setup_caches. */
int fmi_dualar_set_attn_long_threshold(fmi_dualar* h, int threshold);
/* Time of the last fmi_dualar_decode in ms measured with HIP events on `stream`, and the
 * number of kernel launches per frame. */
int fmi_dualar_last_decode_stats(fmi_dualar* h, float* ms, int* launches_per_frame);

/* -------------------- op-level entry points (parity tests call the kernels directly) ---- */

/* out[b][n] = sum_k xn[b][k] * W[n][k] with W row-major bf16 (re-tiled internally into
 * scratch).  norm_w != NULL fuses RMSNorm(x) (llama.py:990-1001) in the prologue.
 * epilogue: 0 = store bf16, 1 = bf16(residual + bf16(acc)) (llama.py:842),
 * 2 = SwiGLU pairing silu(w1 x)*w3 x with W = [w1;w3] stacked (llama.py:979-987).
 * force_path: 0 auto, 1 skinny (M<=16), 2 tiled GEMM (the variant the shape selects), 5 tiled GEMM with operands
 * straight from L2, 7 / 8 / 9 the LDS-staged 4-wave / wave-specialised 128x128 / wave-specialised 128x256 kernels,
 * 10 / 11 / 14 / 15 the 256 / 128 / 64 / 192-row x 256-column tiles of round 4 (what 2 selects among), 12 / 13 their
 * 16-wave / plain 8-wave forms (2, 5, 7 - 15 agree bit for bit), 6 skinny on the row-balanced weight copy. */
int fmi_op_linear_bf16(const void* x_dev, const void* w_dev, const void* norm_w_dev,
                       const void* residual_dev, void* out_dev, int M, int N, int K, float eps,
                       int epilogue, int force_path, void* stream);

/* The same for a weight-only int8 linear: W int8 (N,K) row-major + per-row bf16 scales;
 * out = epilogue(bf16(bf16(xn @ W^T) * scales)).  stream_int8: 1 = the decode GEMV streams the int8 tiles,
 * 0 = it streams the exactly dequantised bf16 tiles (the two agree bit for bit).  M <= 16, K % 64 == 0. */
int fmi_op_linear_int8(const void* x_dev, const void* w_i8_dev, const void* scales_dev, const void* norm_w_dev,
                       const void* residual_dev, void* out_dev, int M, int N, int K, float eps, int epilogue,
                       int stream_int8, void* stream);

/* One sampler call on bf16 logits [B][ld] (n valid).  ids_dev: optional int32 map from row
 * index to vocab id.  prev_dev: optional RAS window row (B x 10 int32) -- when given, the
 * second (high-temperature) draw + selection of inference.py:118-144 is applied with
 * [sem_begin, sem_end].  out_dev: int32 [B]. */
int fmi_op_sample(const void* logits_dev, int B, int n, int ld, const int32_t* ids_dev,
                  const fmi_sampling* samp, int frame, int draw, const int32_t* prev_dev,
                  int sem_begin, int sem_end, int32_t* out_dev, void* stream);

/* ------------------------------------------------------------------------- codec (DAC) -- */

typedef struct fmi_dac fmi_dac;

/* Shape of fish_speech/configs/modded_dac_vq.yaml, overridable for small test models. */
typedef struct fmi_dac_config {
  int32_t encoder_dim;          /* 64 */
  int32_t encoder_rates[4];     /* 2,4,8,8 */
  int32_t decoder_dim;          /* 1536 */
  int32_t decoder_rates[4];     /* 8,8,4,2 */
  int32_t latent_dim;           /* 1024 = encoder_dim * 16 */
  int32_t n_codebooks;          /* 9 residual */
  int32_t codebook_size;        /* 1024 */
  int32_t semantic_codebook_size; /* 4096 */
  int32_t codebook_dim;         /* 8 */
  int32_t downsample[2];        /* 2,2 */
  int32_t tf_layers;            /* 8  (quantizer pre/post modules) */
  int32_t tf_heads;             /* 16 */
  int32_t tf_ffn;               /* 3072 */
  int32_t tf_window;            /* 128 */
  int32_t enc_tf_layers;        /* 4  (last encoder block) */
  int32_t enc_tf_window;        /* 512 */
  int32_t sample_rate;          /* 44100 */
} fmi_dac_config;

int64_t fmi_dac_arena_bytes(const fmi_dac_config* cfg);
int fmi_dac_create(const fmi_dac_config* cfg, void* arena_dev, int64_t arena_bytes, fmi_dac** out);
void fmi_dac_destroy(fmi_dac* h);
/* fp32 tensors by codec.pth key, weight-norm pairs passed separately and
 * folded inside (w = g*v/||v||).  ndim<=3, dims[] row-major. */
int fmi_dac_load_tensor(fmi_dac* h, const char* name, const float* src, int ndim,
                        const int64_t* dims, int src_is_device, void* stream);
int fmi_dac_finalize_weights(fmi_dac* h, void* stream);
/* As fmi_dualar_weights_ready: marks an externally filled arena ready and orders the handle's stream after `stream`. */
int fmi_dac_weights_ready(fmi_dac* h, void* stream);
/* Arithmetic of the decode-side contractions (from_indices / decode / decode_tail; the encoder always uses the fp32
 * matrix cores so that its codes stay bit-comparable):
 *   2 (default) fp16 matrix cores on a two-term split of both operands, low part scaled by 2^11: three products, two
 *               fp32 accumulators -- fp32-class results (what is dropped is below 2^-22 of a product): the codec CLI's
 *               fp32 mode, fish_speech/models/dac/inference.py:52-112;
 *   0           fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain), the round-1 path;
 *   1           bf16 matrix cores, operands and result of every conv / linear rounded to bf16: what the engine's
 *               torch.autocast(bfloat16) computes (fish_speech/inference_engine/__init__.py:179-192). */
int fmi_dac_set_precision(fmi_dac* h, int planes);
/* Precision 2 needs every operand inside the fp16 range (|x| < 65504; the reference computes in fp32 / bf16, whose
 * range is 3e38).  The split saturates outside it -- finite, wrong samples instead of NaNs -- and raises a sticky
 * device flag.  This call waits for the handle's stream, returns the flag in *overflowed (0 / 1) and clears it; a
 * caller that sees 1 repeats the decode with fmi_dac_set_precision(h, 0).  fish_speech_amd.dac.MiDAC does so when
 * constructed with check_overflow=True. */
int fmi_dac_fp16_overflow(fmi_dac* h, int* overflowed);

/* Running a codec call BESIDE the Dual-AR frame loop (the MFMA-bound codec and the HBM-bound loop use different parts
 * of the chip; upstream decodes between generate calls, fish_speech/inference_engine/__init__.py:73-140,179-192):
 *   fmi_dac_set_async(h, 1): encode / decode entry points return once their kernels are enqueued on the handle's own
 *     stream and do NOT make the caller's `stream` wait (a cross-queue wait pending for the length of the call taxes
 *     every dispatch of the other queue); the caller orders a consumer with fmi_dac_wait(h, stream) or waits with
 *     fmi_dac_synchronize(h), and keeps the input / output buffers alive until then.
 *   fmi_dac_set_stream_options: re-creates the handle's stream with a dispatch priority (-1 highest, 0 default,
 *     1 lowest) or, when cu_mask_words > 0, with a CU mask (bit i = compute unit i may run this handle's kernels;
 *     hipExtStreamCreateWithCUMask) -- the codec can be confined to part of the chip while the loop keeps the rest.
 *     NOTE: the CU-mask form has no non-blocking flag, so unlike every other handle stream it is a BLOCKING stream: it
 *     synchronises implicitly with the legacy NULL stream (work torch enqueues on its default stream serialises with
 *     the codec's).  Give torch a side stream when the mask is in use.
 *   fmi_dac_set_background(h, bytes): a floor under the dynamic LDS of the decode-side conv kernels; above 80 KiB only one
 *     of their work-groups fits a CU, so the frame loop's work-groups can be co-resident instead of queueing behind them
 *     (0 = off, the default: two to three conv work-groups per CU). */
int fmi_dac_set_background(fmi_dac* h, int lds_floor_bytes);
int fmi_dac_set_async(fmi_dac* h, int enable);
int fmi_dac_wait(fmi_dac* h, void* stream);
int fmi_dac_synchronize(fmi_dac* h);
int fmi_dac_set_stream_options(fmi_dac* h, int priority, int cu_mask_words, const uint32_t* cu_mask);

/* DAC.from_indices (fish_speech/models/dac/modded_dac.py:925-927): indices int64
 * (B,1+n_codebooks,T) -> audio fp32 (B,1,T*frame_length).  Like the reference
 * (rvq.py:354-359) the indices are clamped IN PLACE. */
int fmi_dac_decode(fmi_dac* h, int64_t* indices_dev, int B, int T, float* audio_out_dev,
                   void* stream);
/* Incremental decode for streaming (BASELINE config 5; the consumer is the engine's segment streaming,
 * fish_speech/inference_engine/__init__.py:73-140, which today waits for a whole text chunk): indices
 * (B,1+n_codebooks,T) = all frames generated so far; writes the audio of frames [t0,T) only,
 * (B,1,(T-t0)*frame_length), bit-identical to the same samples of fmi_dac_decode over the final utterance
 * (every codec layer is causal: modded_dac.py:521-588, window mask 380-398).  The decoder conv stack runs on
 * frames [t0-ctx, T) with ctx = fmi_dac_context_frames() (its receptive field, 5 frames for rates 8,8,4,2). */
int fmi_dac_decode_tail(fmi_dac* h, int64_t* indices_dev, int B, int T, int t0, float* audio_out_dev,
                        void* stream);
int fmi_dac_context_frames(const fmi_dac* h);
/* The same samples with the quantizer side kept ACROSS calls (the engine's segment loop,
 * fish_speech/inference_engine/__init__.py:73-140, cut at frame granularity): the roped q|k|v of every layer of the
 * windowed post-transformer, its output and the upsampled latents of frames [0, t0) stay on the device, and only the
 * columns of frames [t0, T) run through LUT -> transformer (attention over the cached keys) -> upsampler (5 frames of
 * left context) -> decoder (fmi_dac_context_frames() of left context).  Bit-identical to fmi_dac_decode_tail.
 * The state is continued when (stream_id, B) equal the previous call's and t0 equals its T -- the caller promises that
 * the codes of frames [0, t0) are the ones it passed before; any other call recomputes frames [0, T) and restarts the
 * state (so interleaved streams stay correct, they only lose the saving).  State: 9 x B x 3 x latent x capacity x 4
 * bytes (capacity = 1024 frames, doubled on demand); fmi_dac_stream_reset frees it. */
int fmi_dac_decode_tail_cached(fmi_dac* h, int64_t* indices_dev, int B, int T, int t0, int64_t stream_id,
                               float* audio_out_dev, void* stream);
int fmi_dac_stream_reset(fmi_dac* h);
/* An utterance's stream has ended (or was cancelled): drop the state kept under `stream_id` now instead of waiting for it
 * to fall out as least recently used (a serving loop opens a stream per utterance; 16 states are kept).  Unknown ids are
 * ignored. */
int fmi_dac_stream_close(fmi_dac* h, int64_t stream_id);
/* DAC.decode (modded_dac.py:929-946): latent z fp32 (B, latent_dim, L) -> audio fp32 (B,1,L*hop_length). */
int fmi_dac_decode_latent(fmi_dac* h, const float* z_dev, int B, int L, float* audio_out_dev, void* stream);
/* DAC.encode (modded_dac.py:874-923): audio fp32 (B,1,N) (N already padded to a multiple of
 * frame_length by the caller shim) -> indices int64 (B,1+n_codebooks,N/frame_length). */
int fmi_dac_encode(fmi_dac* h, const float* audio_dev, int B, int N, int64_t* indices_out_dev,
                   void* stream);
int fmi_dac_frame_length(const fmi_dac* h);
/* Debug tap: copy the quantizer-decode output z (B, latent_dim, 4T) fp32 of the last decode. */
int fmi_dac_debug_z(fmi_dac* h, float** z_dev, int* C, int* L);

#ifdef __cplusplus
}
#endif
#endif /* FISHMI_H */
