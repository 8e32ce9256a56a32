/*
 * whisper_hip.h — C ABI of libwhisper_hip.so, the MI355X (gfx950) native Whisper inference path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a hipStream_t (passed as
 * void*), and returns an int status (WH_OK == 0).  No torch types cross this boundary.  Memory is
 * owned by the caller: the library computes how many bytes it needs (…_bytes functions), the caller
 * hands it one workspace pointer, and the library only carves it.
 *
 * Each function names the reference interface (openai/whisper @ v20250625, file:line under
 * /root/reference/) it stands in for.  The reference has no C FFI — its "operator API" for the hot
 * path is the Python seam `Inference.logits / rearrange_kv_cache / cleanup_caching`
 * (whisper/decoding.py:130-141), `model.encoder(mel)` (whisper/model.py:188), `log_mel_spectrogram`
 * (whisper/audio.py:110) and `median_filter` / `dtw` (whisper/timing.py:19,141).  INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add at each of those seams.
 */
#ifndef WHISPER_HIP_H
#define WHISPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WH_ABI_VERSION 1
#define WH_MEL_SCRATCH_BYTES 2048

/* status codes */
enum {
  WH_OK = 0,
  WH_ERR_ARG = 1,        /* invalid argument (shape / null pointer / unsupported size) */
  WH_ERR_WORKSPACE = 2,  /* workspace too small */
  WH_ERR_HIP = 3,        /* a HIP runtime call failed; see wh_last_hip_error() */
  WH_ERR_STATE = 4,      /* call sequence violation (e.g. step before prefill) */
  WH_ERR_LIMIT = 5,      /* exceeds a compiled-in limit (rows, LDS) */
  WH_ERR_HANDOFF = 6,    /* a bounded in-kernel hand-off spin of the fused decode-step launches ran out (csrc/xattn.hip):
                          * the results of the call are NOT valid.  Never seen on a healthy, unshared device; a GPU
                          * time-sliced between processes may stretch a spin past its bound.  wh_task_greedy /
                          * wh_task_beam do not return it: they check the counter before they return, and on a time-out
                          * move the task to the two-launch kernels for good (which wait for nothing) and re-run the loop
                          * from the prompt (wh_task_info(t, 4) counts these re-runs).  After host-driven wh_task_step
                          * calls ask wh_task_info(t, 1); create the task with WH_TASK_TWO_LAUNCH_SELF |
                          * WH_TASK_TWO_LAUNCH_CROSS to stay off the fused kernels from the start. */
  WH_RUNNING = 7         /* not an error: wh_task_poll — the loop begun with wh_task_greedy_begin / wh_task_beam_begin has not
                          * ended yet; call again */
};

/* element type of weights / activations / KV caches. Accumulation is always fp32. */
enum { WH_F32 = 0, WH_F16 = 1 };

/* ModelDimensions — whisper/model.py:25-36 (same ten integers, same order). */
typedef struct wh_dims {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wh_dims;

/*
 * Packed weights of one ResidualAttentionBlock — whisper/model.py:142-171.
 * Matrices are row-major [out][in] in the model element type (WH_F32 / WH_F16);
 * LayerNorm parameters and biases are always fp32.
 *   qkv_w  = rows [query; key; value] of block.attn      (3D x D); qkv_b has zeros for `key` (model.py:88)
 *   ckv_w  = rows [key; value] of block.cross_attn       (2D x D); ckv_b has zeros for `key`
 * Encoder blocks leave the cross-attention pointers NULL.
 */
typedef struct wh_layer_weights {
  const float *attn_ln_w, *attn_ln_b;
  const void *qkv_w;  const float *qkv_b;
  const void *out_w;  const float *out_b;
  const float *cross_ln_w, *cross_ln_b;
  const void *cq_w;   const float *cq_b;
  const void *ckv_w;  const float *ckv_b;
  const void *cout_w; const float *cout_b;
  const float *mlp_ln_w, *mlp_ln_b;
  const void *fc1_w;  const float *fc1_b;
  const void *fc2_w;  const float *fc2_b;
} wh_layer_weights;

/*
 * Whole-model weight table.
 *   conv1_w : [D][Kc1] with Kc1 = round_up(3*n_mels, 64), element (d, kk*n_mels + c) = conv1.weight[d][c][kk]
 *   conv2_w : [D][3*D], element (d, kk*D + c) = conv2.weight[d][c][kk]      (model.py:182-183)
 *   enc_pos : encoder.positional_embedding, fp32 [n_audio_ctx][D]           (model.py:184)
 *   tok_emb : decoder.token_embedding.weight [n_vocab][D] (element type; also the tied logits matrix, model.py:245)
 *   dec_pos : decoder.positional_embedding fp32 [n_text_ctx][D]             (model.py:214)
 */
typedef struct wh_model_weights {
  const void *conv1_w; const float *conv1_b;
  const void *conv2_w; const float *conv2_b;
  const float *enc_pos;
  const wh_layer_weights *enc_layers;   /* n_audio_layer entries (host array) */
  const float *enc_ln_post_w, *enc_ln_post_b;
  const void *tok_emb;
  const float *dec_pos;
  const wh_layer_weights *dec_layers;   /* n_text_layer entries (host array) */
  const float *dec_ln_w, *dec_ln_b;
  uint32_t flags;                       /* WH_WEIGHTS_* */
} wh_model_weights;

/* wh_model_weights.flags
 * WH_WEIGHTS_DEC_LN_FOLDED: in every DECODER block the affine part of attn_ln / cross_attn_ln / mlp_ln
 *   (whisper/model.py:39-41,150-157) has been folded into the Linear that consumes it,
 *     W (g * xhat + beta) + b == (W * g) xhat + (b + W beta),
 *   i.e. qkv_w/qkv_b, cq_w/cq_b, fc1_w/fc1_b already contain it and the stored ln_w / ln_b are (1, 0).  The decode-step
 *   projections then normalise in registers without reading gamma / beta.  decoder.ln (tied to the token embedding)
 *   and the encoder are never folded.  whisper_amd.hip.pack_weights does this for WH_F16 blobs. */
#define WH_WEIGHTS_DEC_LN_FOLDED 1u
/* WH_WEIGHTS_ENC_QK_SCALED (WH_F16 only): in every ENCODER block the query and key rows of qkv_w (rows [0, 2 n_state))
 *   and the query part of qkv_b are multiplied by sqrt(0.125 * log2 e) = 0.42466..., so that k.q is already the base-2
 *   exponent of the softmax: whisper/model.py:118-121 scales q and k by d_head ** -0.25 each (0.125 on the product),
 *   and exp(x) = 2 ** (x log2 e).  The encoder's flash-attention kernel then exponentiates the score accumulators as
 *   they are.  Without the flag the kernel multiplies by 0.125 log2 e itself.  whisper_amd.hip.pack_weights sets it
 *   for WH_F16 blobs; the decoder's attention is never pre-scaled. */
#define WH_WEIGHTS_ENC_QK_SCALED 2u
/* wh_model_create rejects (WH_ERR_ARG) any other bit in `flags`, and any flag with WH_F32: zero the struct before
 * filling it.  whisper_amd.hip.pack_weights records the flags in the last 64 bytes of the blob it returns
 * (int32 magic "WHB1", dtype, flags) so that they travel with the bytes. */

typedef struct wh_model wh_model;   /* opaque: dims + copies of the pointer tables */
typedef struct wh_task wh_task;     /* opaque: per-DecodingTask KV caches + workspace carve-up.  One call at a time per
                                     * handle: a second thread entering while a call runs gets WH_ERR_STATE (enforced) */
/* Concurrency.  A wh_model is immutable and may be used by any number of threads and tasks at once.  Tasks on DIFFERENT
 * streams run concurrently on the GPU, and for decode chains of few rows that is worth up to 1.5 x the throughput
 * (DESIGN.md §3 "Lanes").  Either every task is driven by its own host thread (wh_task_greedy / wh_task_beam return when
 * their loop has ended; the thread sleeps while it waits), or ONE thread drives them all through wh_task_greedy_begin /
 * wh_task_beam_begin + wh_task_poll, which never wait.  The library takes no lock: wh_encode calls that share a workspace
 * must be enqueued on one stream and not from two threads at once; every task owns its workspace. */

/* ---- library ------------------------------------------------------------------------------ */
int wh_abi_version(void);
const char *wh_status_string(int status);
int wh_last_hip_error(void);            /* hipError_t of the last failing runtime call on this thread */
const char *wh_last_hip_error_string(void);

/* ---- audio front end: log_mel_spectrogram — whisper/audio.py:110-157 ----------------------- */
/* audio: fp32 [batch][n_samples] (already zero-padded by `padding`, audio.py:145-146);
 * filters: fp32 [n_mels][201] (mel_filters(), audio.py:91-107); out: fp32 [batch][n_mels][n_samples/160].
 * The global max of audio.py:155 is taken over the whole [batch] tensor, as the reference does.
 * scratch: >= WH_MEL_SCRATCH_BYTES bytes of device memory. n_samples > 200 (reflect padding). */
int wh_log_mel(const float *audio, int64_t n_samples, int batch, int n_mels, const float *filters,
               float *out, void *scratch, void *stream);

/* ---- model handle -------------------------------------------------------------------------- */
/* Stands in for Whisper.__init__ + load_state_dict (whisper/__init__.py:154-156): the caller has
 * already packed the checkpoint into device memory; this only records dims and pointers. */
int wh_model_create(const wh_dims *dims, int dtype, const wh_model_weights *weights, wh_model **out);
void wh_model_destroy(wh_model *m);

/* ---- AudioEncoder.forward — whisper/model.py:188-204 --------------------------------------- */
size_t wh_encoder_workspace_bytes(const wh_model *m, int batch);
/* mel: [batch][n_mels][2*n_audio_ctx] in fp32 (mel_is_f16 == 0) or fp16 (1);
 * out: [batch][n_audio_ctx][D] in the model element type. */
int wh_encode(const wh_model *m, const void *mel, int mel_is_f16, int batch, void *out,
              void *workspace, size_t workspace_bytes, void *stream);

/* ---- decoding task: PyTorchInference + kv_cache — whisper/decoding.py:144-176, model.py:310-341 */
/* flags for wh_task_create */
enum {
  WH_TASK_CAPTURE_Q = 1,         /* keep cross-attention queries of every layer (word timestamps) */
  /* decode step: projection and attention as separate launches even where the fused kernels of csrc/xattn.hip apply
   * (A/B and tests: the results must agree) — for the self attention (the default since round 6, see
   * WH_TASK_FUSED_SELF; the flag is accepted and wins over it) / for the cross attention */
  WH_TASK_TWO_LAUNCH_SELF = 2,
  WH_TASK_TWO_LAUNCH_CROSS = 4,
  /* 8: reserved (development builds of the library only; ignored here) */
  /* Fault injection for the hand-off protocol of the fused step kernels: every consumer gives up after its FIRST poll, as
   * if its bounded spin had run out.  The step's result is then invalid by construction; wh_task_greedy / wh_task_beam
   * must notice (WH_ERR_HANDOFF internally), move the task to the two-launch kernels and re-run — what the tests check. */
  WH_TASK_EXPIRE_HANDOFFS = 16,
  /* decode step of <= 8 rows (fp16): LayerNorm + QKV projection + cache append + self attention as ONE launch
   * (sattn8_kernel, csrc/xattn.hip; bit-identical to the two launches).  Opt-in since round 6: alone on the chip the two
   * forms take the same time, beside other decode chains the fused one is slower (its consumers spin for q / k / v), so
   * the default step keeps only the cross attention fused (where the K/V stream hides the projection). */
  WH_TASK_FUSED_SELF = 32
};
/* The workspace holds the cross-attention K/V of n_audio segments, the self-attention cache of n_audio * n_group rows
 * and the step buffers.  WH_F16 tasks with n_group > 1 (beam search) additionally hold a transposed copy of the
 * cross-attention V per layer (the matrix-core form of the beam-group attention reads it): + n_text_layer * n_audio *
 * n_text_state * 1536 * 2 bytes at n_audio_ctx = 1500.  wh_task_workspace_bytes accounts for all of it. */
size_t wh_task_workspace_bytes(const wh_model *m, int n_audio, int n_group, int max_prefill_tokens,
                               int flags);
int wh_task_create(const wh_model *m, int n_audio, int n_group, int max_prefill_tokens, int flags,
                   void *workspace, size_t workspace_bytes, wh_task **out);
void wh_task_destroy(wh_task *t);
/* Cross-attention K/V of every decoder layer from the encoder output (the `key`/`value` Linear of
 * block.cross_attn applied to xa, model.py:101-105; cached by the hook at model.py:327-330).
 * features: [n_audio][n_audio_ctx][D] element type. */
int wh_task_set_audio(wh_task *t, const void *features, void *stream);
/* First Inference.logits call (decoding.py:155-163 with an empty kv_cache): feeds T0 tokens per row,
 * fills the self-attention caches at positions [offset, offset+T0) and returns fp32 logits.
 *   tokens: int64 [n_rows][token_stride] device; uses columns [0, T0)
 *   sel_pos/n_sel: host int array of positions in [0,T0) whose logits are wanted (NULL -> all T0);
 *   logits_out: fp32 [n_rows][n_sel][n_vocab].
 * May be called repeatedly (T0 >= 1) to append teacher-forced tokens; n_rows = n_audio * n_group. */
int wh_task_prefill(wh_task *t, const int64_t *tokens, int64_t token_stride, int T0,
                    const int32_t *sel_pos, int n_sel, float *logits_out, void *stream);
/* Later Inference.logits calls (decoding.py:159-163): one new token per row.
 *   last_tokens: int64 device, row r at last_tokens[r*token_stride]; logits_out: fp32 [n_rows][n_vocab]. */
int wh_task_step(wh_task *t, const int64_t *last_tokens, int64_t token_stride, float *logits_out,
                 void *stream);
/* Inference.rearrange_kv_cache (decoding.py:172-176): new row i takes the self-attention cache of
 * old row source_indices[i] (host array of n_rows ints). */
int wh_task_rearrange(wh_task *t, const int32_t *source_indices, void *stream);
/* Inference.cleanup_caching (decoding.py:165-170): forget cached positions (keeps the audio). */
int wh_task_reset(wh_task *t, void *stream);   /* stream-ordered (no host synchronisation) */
/* Ragged prompts (no counterpart in the reference, whose DecodingTask shares one initial_tokens tuple between all
 * rows, decoding.py:719).  Row r's token sequence is the longest row's shifted left by lag[r]
 * (host array of n_rows ints, 0 <= lag[r] < max_prefill_tokens; NULL = all zero): it has a shorter leading prompt.
 * Call before the prefill, with the position at 0.  The prefill still takes T0 tokens per row, of which the last
 * lag[r] are padding (any valid id); sel_pos, sot_index and sample_begin are given for the longest row and apply
 * lag[r] earlier in row r; each decode step appends row r at cache position (position - lag[r]) and
 * wh_task_greedy writes row r's sampled tokens from column sample_begin - lag[r].  Cleared by wh_task_reset. */
int wh_task_set_lag(wh_task *t, const int32_t *lag, void *stream);
/* number of cached self-attention positions of the longest row (the `offset` of model.py:234) */
int wh_task_position(const wh_task *t);
/* Introspection for tests and the benchmark.  what = 0: 1 when this task's decode step runs the cross attention with its
 * LayerNorm + query projection inside the same launch (csrc/xattn.hip: fp16, <= 8 rows, one row per audio), else 0.
 * what = 2: the same question for self attention + QKV projection + cache append (sattn8_kernel); what = 3: attn.out + the residual
 * add run inside one of the attention launches (development builds only: always 0 in the shipped library).
 * what = 1: number of bounded hand-off spins that ran out in that kernel since the task was created (always 0 on a
 * healthy device; reads device memory, i.e. synchronises `stream`).  what = 4: number of times wh_task_greedy /
 * wh_task_beam re-ran a loop on the two-launch kernels after such a time-out (the answers to 0, 2, 3 are 0 from then on).
 * Negative on error. */
int wh_task_info(wh_task *t, int what, void *stream);

/*
 * Fused sampling loop == DecodingTask._main_loop with GreedyDecoder (arg-max at temperature 0, otherwise one
 * categorical draw per row and step) and the SuppressBlank / SuppressTokens / ApplyTimestampRules filters
 * (decoding.py:272-298, 423-505, 680-710), run entirely on the device (no per-step host sync).  Rows are independent:
 * a task created with n_group = best_of decodes best_of samples per audio segment.
 */
typedef struct wh_greedy_params {
  int32_t sample_begin;          /* len(initial_tokens), decoding.py:536 */
  int32_t max_steps;             /* sample_len, decoding.py:529 (prefill step included) */
  int32_t n_ctx;                 /* loop ends once token count > n_ctx, decoding.py:705 */
  int32_t eot;                   /* tokenizer.eot */
  int32_t timestamp_begin;       /* tokenizer.timestamp_begin; < 0 disables ApplyTimestampRules */
  int32_t no_timestamps;         /* tokenizer.no_timestamps id or -1 */
  int32_t max_initial_timestamp_index; /* decoding.py:561-565 or -1 */
  int32_t suppress_blank;        /* 0/1, decoding.py:555-556 */
  int32_t blank_token;           /* tokenizer.encode(" ")[0] */
  const uint8_t *suppress_mask;  /* device [n_vocab] bytes: 1 = token in SuppressTokens list */
  float temperature;             /* 0: arg-max (decoding.py:278-279); > 0: Categorical(logits / temperature).sample()
                                    (:281-283) drawn on the device by Gumbel-max with counter-based noise — the same
                                    distribution, not torch's random stream; wh_task_greedy only */
  uint32_t reserved;
  uint64_t seed;                 /* noise key of this call (temperature > 0) */
} wh_greedy_params;
/*
 * tokens: int64 [n_rows][token_stride] device, columns [0,sample_begin) hold the initial tokens; the
 * loop appends sampled tokens in place.  sum_logprobs: fp32 [n_rows] (zeroed by the call).
 * no_speech_probs: fp32 [n_rows] out = softmax(logits at sot_index)[no_speech] (decoding.py:689-693),
 * skipped when no_speech_token < 0.  n_tokens_out (host): final token count per row (same for all rows).
 */
int wh_task_greedy(wh_task *t, const wh_greedy_params *p, int64_t *tokens, int64_t token_stride,
                   int sot_index, int no_speech_token, float *sum_logprobs, float *no_speech_probs,
                   int32_t *n_tokens_out, void *stream);
/*
 * The same loop without a blocked caller: DecodingTask._main_loop (decoding.py:680-710) split at the points where the host
 * would wait for the device.  wh_task_greedy_begin queues the prompt pass and the first sampling decision on `stream` and
 * returns at once (it never synchronises); wh_task_poll(t, &n) then queues further decode steps — never more than about
 * ten ahead of the device — and returns WH_RUNNING until the loop has ended, WH_OK (n = final token count per row,
 * results in the buffers given to _begin) once it has, or an error.  It never waits either: the host only asks whether an
 * event has been reached.  So ONE host thread can keep any number of tasks on different streams going by polling them in
 * turn (several decode chains in flight on a GPU: DESIGN.md §3 "Lanes"); call it at least every millisecond or two per
 * task, or the device runs out of queued steps.  Between _begin and the poll that reports the end, the task refuses every
 * other call (WH_ERR_STATE), and the buffers and `p->suppress_mask` must stay valid.  A hand-off time-out of the fused
 * step kernels is handled inside wh_task_poll exactly as inside wh_task_greedy (the loop is re-run on the two-launch
 * kernels; polls keep returning WH_RUNNING meanwhile).  wh_task_greedy == wh_task_greedy_begin + wh_task_poll until done,
 * with sleeping waits in place of the queries — the same sequence of device operations.
 */
int wh_task_greedy_begin(wh_task *t, const wh_greedy_params *p, int64_t *tokens, int64_t token_stride,
                         int sot_index, int no_speech_token, float *sum_logprobs, float *no_speech_probs,
                         void *stream);
int wh_task_poll(wh_task *t, int32_t *n_tokens_out);

/*
 * Fused beam search loop == DecodingTask._main_loop with BeamSearchDecoder (decoding.py:301-404) and the stock logit
 * filters, run on the device without a per-step host sync: filters + log_softmax + top-(beam+1) per row, the candidate
 * bookkeeping of every audio segment (stable descending order, first `beam` non-EOT sequences survive, EOT ones go
 * to the segment's finished list up to max_candidates), the KV-cache permutation (rearrange_kv_cache, :172-176).
 * The task must have been created with n_group == beam_size (2..8).
 * tokens: int64 [2][n_rows][token_stride] device (token_stride >= sample_begin + max_steps + 1); [0] holds the initial
 * tokens in columns [0, sample_begin) of every row and receives the live beams.  sum_logprobs: fp32 [n_rows] out.
 * Finished sequences of segment b, in the order the reference's dict holds them: fin_tokens [n_audio][max_candidates]
 * [token_stride], fin_len / fin_scores [n_audio][max_candidates], fin_count [n_audio] (all device).
 * n_tokens_out (host): length of the live beams' rows.  BeamSearchDecoder.finalize (:384-404) stays with the caller.
 */
typedef struct wh_beam_params {
  wh_greedy_params rules;        /* as for wh_task_greedy */
  int32_t beam_size;             /* decoding.py:303 */
  int32_t max_candidates;        /* round(beam_size * patience), decoding.py:313 */
} wh_beam_params;
int wh_task_beam(wh_task *t, const wh_beam_params *p, int64_t *tokens, int64_t token_stride, int sot_index,
                 int no_speech_token, float *sum_logprobs, float *no_speech_probs, int64_t *fin_tokens,
                 int32_t *fin_len, float *fin_scores, int32_t *fin_count, int32_t *n_tokens_out, void *stream);
/* wh_task_beam without a blocked caller: begin, then wh_task_poll until it stops returning WH_RUNNING (see
 * wh_task_greedy_begin; n_tokens_out of wh_task_poll = length of the live beams' rows). */
int wh_task_beam_begin(wh_task *t, const wh_beam_params *p, int64_t *tokens, int64_t token_stride, int sot_index,
                       int no_speech_token, float *sum_logprobs, float *no_speech_probs, int64_t *fin_tokens,
                       int32_t *fin_len, float *fin_scores, int32_t *fin_count, void *stream);

/* cross-attention QK of chosen heads for the cached positions — the `qk` captured by the hooks of
 * find_alignment (whisper/timing.py:186-197; model.py:130-137 manual path): for every pair
 * (layers[i], heads[i]) writes fp32 [n_tok][n_audio_ctx] = (q*scale)·(k*scale)ᵀ of row `row`.
 * Requires a task created with WH_TASK_CAPTURE_Q; tokens [tok_begin, tok_begin+n_tok) must have been
 * fed through wh_task_prefill. */
int wh_task_cross_qk(wh_task *t, int row, const int32_t *layers, const int32_t *heads, int n_pairs,
                     int tok_begin, int n_tok, float *out, void *stream);

/* ---- measurement hook (bench.py roofline leg; not part of the reference surface) ------------------
 * ONE kernel of the decode step, `iters` launches rotating over the decoder layers (every launch streams HBM-cold data
 * exactly like the real step does), captured into a hipGraph and replayed on `stream` (non-null) — the dependent-launch
 * boundaries of the real step — with HIP events around the replay; *ms_per_launch = best of 3 replays / iters.
 * kind: 0 = whole decode step (position not advanced beyond the cache), 1 = cross-attention decode kernel, 2 = self-attention decode kernel,
 * 3 = LN+QKV GEMV, 4 = LN+FC1 GEMV, 5 = FC2 GEMV, 6 = LN+logits GEMV, 7 = out-proj GEMV.
 * *bytes_per_launch receives the algorithmic HBM bytes of one launch (SURVEY.md §8d accounting). */
int wh_task_bench_kernel(wh_task *t, int kind, int iters, double *bytes_per_launch, float *ms_per_launch,
                         void *stream);

/* ---- word-timestamp kernels — whisper/timing.py:19-54 (median_filter), :82-151 (dtw) -------- */
/* x: fp32 [rows][n] -> out: fp32 [rows][n], reflect-padded sliding median of odd `width` along n. */
int wh_median_filter(const float *x, float *out, int64_t rows, int n, int width, void *stream);
/* x: fp32 [N][M] cost matrix.  trace_out: int8 [N+1][M+1] with dtw_cpu's codes (0 diag, 1 up, 2 left)
 * and dtw_cpu's tie rule (timing.py:95-100); row 0 / column 0 hold the codes backtrace() forces there
 * (timing.py:61-62).  N <= 8192. */
int wh_dtw_trace(const float *x, int N, int M, int8_t *trace_out, void *stream);
/* backtrace() of whisper/timing.py:57-79 for a batch of traces, on the device: clip b's trace is dense
 * int8 [(n_rows[b] + 1)][(n_cols[b] + 1)] at trace + b * trace_stride (as wh_dtw_trace / wh_task_align_batch write it);
 * d_n_rows / d_n_cols are DEVICE int32 arrays of n_clips entries, max_rows / max_cols their maxima.
 *   jumps_out (device int32 [n_clips][jump_stride], jump_stride >= max_rows, or NULL): entry [b][i] = the time index of the
 *     first path element whose text index is i — `time_indices[jumps]` of timing.py:226-228, what find_alignment turns
 *     into word boundaries (only these n_rows[b] ints per clip have to reach the host, not the trace);
 *   path_out (device int32 [n_clips][2][path_stride], path_stride >= max_rows + max_cols, or NULL) with path_len_out
 *     (device int32 [n_clips]): the (text, time) index pairs of `dtw()` (timing.py:141-151), RIGHT-aligned: row 0 =
 *     text indices, row 1 = time indices, both in [path_stride - len, path_stride); len = -1 if a trace code was invalid
 *     (the reference raises ValueError there). */
int wh_dtw_backtrace_batch(const int8_t *trace, int64_t trace_stride, const int32_t *d_n_rows, const int32_t *d_n_cols,
                           int n_clips, int max_rows, int max_cols, int32_t *jumps_out, int64_t jump_stride,
                           int32_t *path_out, int64_t path_stride, int32_t *path_len_out, void *stream);
/* find_alignment's attention post-processing — whisper/timing.py:207-216: qk fp32 [n_heads][n_tok][n_audio_ctx]
 * (from wh_task_cross_qk) -> crop to the first n_frames frames, softmax(qk * qk_scale) over frames, z-normalise
 * over the token axis (biased std), median filter of odd `width` along frames, mean over heads, keep token rows
 * [row_begin, row_end) and negate -> out fp32 [row_end-row_begin][n_frames], the cost matrix handed to dtw.
 * scratch: >= 2*n_heads*n_tok*n_frames*4 + 16 bytes. */
int wh_align_matrix(const float *qk, int n_heads, int n_tok, int n_audio_ctx, int n_frames, int width,
                    int row_begin, int row_end, float qk_scale, float *out, void *scratch, void *stream);

/* find_alignment for every row of a task at once (timing.py:186-216 per clip; BASELINE configs[4]: a batch of clips
 * with word_timestamps).  The task (WH_TASK_CAPTURE_Q, one row per clip) has been teacher-forced with
 * [sot sequence, <|notimestamps|>, text tokens, <|endoftext|>] of every clip (shorter rows padded on the right).  For row r
 * the alignment covers tokens [0, n_tok[r]) and frames [0, n_frames[r]):
 *   QK of the (layer, head) pairs -> softmax over frames -> z-norm over tokens -> median(width) -> -mean over heads,
 *   rows [row_begin, n_tok[r] - 1) -> cost_out [n_rows][Nmax][Fmax] fp32 (Nmax = max n_tok - 1 - row_begin, Fmax = max
 *   n_frames) -> dtw: trace_out row r at byte r * trace_stride, dense [(N_r + 1)][(n_frames[r] + 1)] int8 (dtw_cpu codes).
 * n_tok / n_frames / layers / heads are host arrays.  Results are bit-identical to wh_task_cross_qk + wh_align_matrix +
 * wh_dtw_trace clip by clip (the single-clip entry points run the same kernels as a batch of one). */
size_t wh_align_batch_scratch_bytes(int n_rows, int n_pairs, int max_tok, int n_audio_ctx, int max_frames);
int wh_task_align_batch(wh_task *t, const int32_t *layers, const int32_t *heads, int n_pairs, const int32_t *n_tok,
                        const int32_t *n_frames, int width, int row_begin, float qk_scale, float *cost_out,
                        int8_t *trace_out, int64_t trace_stride, void *scratch, size_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_HIP_H */
