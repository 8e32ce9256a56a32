// kernels.h — internal launcher interface between api.cpp and the .hip kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// developer probe (tools/probe_decode.cpp builds the kernels with -DWH_PROBE): thread 0 of every
// workgroup records s_memtime at phase boundaries.  Compiled out of libwhisper_hip.so.
#ifdef WH_PROBE
#define WH_PROBE_FIELD long long* probe;
#define WH_PROBE_KVHS_FIELD int64_t kv_hs;   /* probe builds only: head stride of the cross K / V (0 = 64: heads side by side in a key row) */
#define WH_PROBE_AT(args, wg, i) \
  do { if ((args).probe && threadIdx.x == 0) (args).probe[(size_t)(wg) * 8 + (i)] = clock64(); } while (0)
#else
#define WH_PROBE_FIELD
#define WH_PROBE_KVHS_FIELD
#define WH_PROBE_AT(args, wg, i) do {} while (0)
#endif

// Developer A/B switches (tools/README.md): environment variables are read ONLY by the -DWH_DEV build
// (`make dev` -> libwhisper_hip_dev.so, loaded by the tools through WHISPER_AMD_LIB); in the shipped library every
// switch is the constant of the measured-faster form and its name does not exist in the binary.  The one variable the
// shipped library reads is WH_NO_GRAPH (api.cpp: decode steps launched eagerly — a support switch, not an experiment).
#ifdef WH_DEV
#include <stdlib.h>
#define WH_DEV_FLAG(name) ([] { static const bool v = [] { const char* e = getenv(name); return e && e[0] == '1'; }(); return v; }())
#define WH_DEV_INT(name) ([] { static const int v = [] { const char* e = getenv(name); return e ? atoi(e) : 0; }(); return v; }())
#else
#define WH_DEV_FLAG(name) false
#define WH_DEV_INT(name) 0
#endif

#include <atomic>
namespace whk {

// Dynamic LDS above 64 KB is a per-DEVICE function attribute (hipFuncSetAttribute): raise it once on every device this
// process launches `fn` on.  `c` is the call site's own cache (function-local static).
struct LdsAttr { std::atomic<int> done[64]; };
inline hipError_t raise_dynamic_lds(LdsAttr& c, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (c.done[dev].load(std::memory_order_acquire) == 1) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) c.done[dev].store(1, std::memory_order_release);
  else (void)hipGetLastError();
  return e;
}

// dtype: 0 = fp32, 1 = fp16 (matches WH_F32 / WH_F16)

// ---- mel.hip -------------------------------------------------------------------------------
hipError_t launch_log_mel(const float* audio, int64_t n_samples, int batch, int n_mels,
                          const float* filters, const float* tables, float* out, void* scratch,
                          hipStream_t stream);

// ---- gemm.hip ------------------------------------------------------------------------------
struct GemmArgs {
  const void* A; int64_t lda; int64_t a_bs;     // [M][K], row stride lda (elements), batch stride
  const void* W; int64_t ldw; int64_t w_bs;     // [N][K]
  void* C; int64_t ldc; int64_t c_bs;           // [M][N]
  const float* bias; int bias_on_m;             // per-n (default) or per-m bias
  const float* res; int64_t ldr; int64_t r_bs; int res_mod;  // fp32 residual, row index m % res_mod if res_mod>0
  int act;                                      // 0 none, 1 exact GELU
  int M, N, K;
  int tiles_m, tiles_n, persistent;             // filled by the launcher
  int serial_epilogue;                          // filled by the launcher: A/B switch WH_GEMM_SERIAL_EPILOGUE=1 (gemm.hip)
  WH_PROBE_FIELD
};
hipError_t launch_gemm(const GemmArgs& a, int dtype, int out_f32, int batch, hipStream_t stream);

// ---- elementwise.hip -----------------------------------------------------------------------
// mel [B][n_mels][F] (fp32 or fp16) -> [B][F+2][n_mels] element type with zero rows 0 and F+1
hipError_t launch_mel_transpose(const void* mel, int mel_is_f16, int B, int n_mels, int F, void* out,
                                int dtype, hipStream_t stream);
// fp32 x [rows][D] (row stride ldx) -> element-type out [rows][D] (row stride ldo)
hipError_t launch_layernorm(const float* x, int64_t ldx, const float* w, const float* b, void* out,
                            int64_t ldo, int64_t rows, int D, int dtype, hipStream_t stream);
// x[r*T0+t][:] = tok_emb[tokens[r*stride+t]][:] + pos[(offset - lag[r] + t)][:]   (fp32 out; lag may be null)
hipError_t launch_embed(const int64_t* tokens, int64_t stride, int R, int T0, const void* tok_emb,
                        const float* pos, const int* d_offset, const int* lag, int D, int n_vocab, float* x,
                        int dtype, hipStream_t stream);
// qkv [R*T0][3D] -> q [R*T0][D] is left in place (ld 3D); k,v rows scattered to caches [R][n_ctx][D]
hipError_t launch_scatter_kv(const void* qkv, int R, int T0, int D, const int* d_offset, int n_ctx,
                             void* kcache, void* vcache, int dtype, hipStream_t stream);
// out[i][:] = x[sel[i]][:] (fp32 rows of width D); sel device int array
hipError_t launch_gather_rows(const float* x, const int* sel, int n_sel, int D, float* out,
                              hipStream_t stream);
// dst rows <- src rows[src_idx[i]] : self-attention cache gather for beam search (rows of row_elems elements)
hipError_t launch_gather_cache(const void* src, void* dst, const int* src_idx, int R, int64_t row_bytes,
                               int64_t used_bytes, hipStream_t stream);
// in-place beam reorder of every layer's self-attention K and V cache in one launch; src_idx[i] must lie in the
// beam group of row i (groups of G consecutive rows), G <= 8.  copy_from (may be null): row i only needs the cache
// positions >= copy_from[i] of its source — below that the two rows already hold the same bytes (pos_bytes per position)
hipError_t launch_permute_groups(void* k_base, void* v_base, int n_layers, int64_t layer_bytes, int n_audio, int G,
                                 int64_t row_bytes, int64_t used_bytes, const int* src_idx, const int* copy_from,
                                 int64_t pos_bytes, hipStream_t stream);
// the first used_bytes of row src_row -> rows [dst_row0, dst_row0 + G) in each of n_layers slabs (layer_bytes apart)
hipError_t launch_replicate_row(void* base, int64_t layer_bytes, int n_layers, int64_t row_bytes, int src_row, int dst_row0,
                                int G, int64_t used_bytes, hipStream_t stream);
hipError_t launch_add_int(int* p, int v, hipStream_t stream);

// ---- attention.hip -------------------------------------------------------------------------
struct AttnArgs {
  const void* q; int64_t q_ld; int64_t q_bs;      // q rows [Tq] per batch row (element type), head h at cols h*64
  const void* k; int64_t k_ld; int64_t k_bs;      // keys  [Tk] per kv-batch
  const void* v; int64_t v_ld; int64_t v_bs;
  void* out; int64_t o_ld; int64_t o_bs;          // [Tq][H*64] element type
  int H; int Tq; int Tk;                          // Tk used when d_len == nullptr
  const int* d_len;                               // device: cached length BEFORE this call (Tk = *d_len + Tq), or null
  int causal;                                     // query t sees keys <= (Tk - Tq) + t
  int kv_group;                                   // kv batch index = batch / kv_group
  float* qk_out;                                  // unused (reserved)
};
// generic (any Tq/Tk, VALU) attention, used for fp32 parity mode and decoder prefill
hipError_t launch_attn_generic(const AttnArgs& a, int batch, int dtype, hipStream_t stream);
// MFMA flash attention for the encoder (fp16 only): q,k [B][T][..] rows, vt = V transposed [B][H*64][ldv];
// prescaled: q and k each carry sqrt(0.125 * log2 e) already (WH_WEIGHTS_ENC_QK_SCALED)
hipError_t launch_attn_flash_f16(const void* q, int64_t q_ld, int64_t q_bs, const void* k, int64_t k_ld,
                                 int64_t k_bs, const void* vt, int64_t vt_ld, int64_t vt_bs, void* out,
                                 int64_t o_ld, int64_t o_bs, int B, int H, int T, int prescaled, hipStream_t stream, int Tq = 0);
// single-query decode attention with split-K partials
struct DecAttnArgs {
  const void* q; int64_t q_ld;                    // [R][H*64]
  const void* k; int64_t k_ld; int64_t k_bs;      // key j of head h at k + b*k_bs + j*k_ld + h*kv_hs
  const void* v; int64_t v_ld; int64_t v_bs;
  int64_t kv_hs;                                  // head stride in elements: 0 means 64 (heads side by side in a row)
  int H; int R; int kv_group;
  int Tk; const int* d_len; int len_plus;         // Tk fixed, or *d_len + len_plus
  const int* lag;                                 // with d_len: row r holds lag[r] fewer keys (ragged prompts); may be null
  int splits;                                     // key-range splits
  void* out; int64_t o_ld;                        // splits==1: normalized output, element type
  int o_frag;                                     // splits==1, fp16: write `out` in fragment order (GemvArgs::x_frag below); o_ld ignored
  void* part_o; float* part_ml;                   // splits>1: [S][R][H][64] element type, normalised (o / l); (m, l) fp32 [S][R][H][2]
  // fp16, kv_group == 1, 2 <= splits <= 4: [R][H] tickets, all 0 between launches.  With it the last workgroup of a (row, head)
  // to finish merges the partials and writes `out` (o_ld / o_frag as for splits == 1): no merge launch, no PRO_COMBINE after it
  int* merge_cnt;
  // beam groups, fp16: V transposed per audio, row h*64+d holds the keys of head dim d (row stride vt_ld >= the
  // padded key count, pad columns finite).  With it the group kernel runs on the matrix cores; null: vector ALU form
  const void* vt; int64_t vt_ld; int64_t vt_bs;
  WH_PROBE_FIELD
};
constexpr int DEC_ATTN_MAX_SPLITS = 16;
// keys one workgroup of attn_decode can hold in registers (per split)
int attn_decode_capacity(int dtype);
hipError_t launch_attn_decode(const DecAttnArgs& a, int dtype, hipStream_t stream);
// scores of selected (layer, head) pairs: out[pair][t][j] = scale^2 * q[t]·k[j]
// batched: row r (grid.z = r * n_pairs + pair), queries qcap [layer][R][C][D] at positions [0, Tmax), keys of audio
// r / kv_group in cross_kv [layer][B * Tk][2D]; out [R][n_pairs][Tmax][Tk]; positions >= ntok[r] are skipped
hipError_t launch_cross_qk_batch(const void* qcap, int64_t q_layer_stride, int64_t q_row_stride, int D, const void* cross_kv,
                                 int64_t kv_layer_stride, int64_t kv_audio_stride, int kv_group, const int* d_layers,
                                 const int* d_heads, int n_pairs, const int* d_ntok, int R, int Tmax, int Tk, float* out,
                                 int dtype, hipStream_t stream);
hipError_t launch_cross_qk(const void* q, int64_t q_ld, const void* k, int64_t k_ld, int head, int n_tok,
                           int Tk, float* out, int dtype, hipStream_t stream);

// ---- xattn.hip: cross attention of one decode step with its LayerNorm + query projection inside the launch ----------
struct XAttnArgs {
  // projection: q = (W . LN(x) + bias) * d_head^-0.5, LayerNorm affine folded into W / bias (WH_WEIGHTS_DEC_LN_FOLDED)
  const float* xf; int64_t xf_ld;                 // fp32 residual rows [R][D]
  const void* W; const float* bias;               // fp16 [D][D], fp32 [D]
  int D, H, R;
  // cross K/V of row r (one audio per row): key j of head h at k + r*k_bs + j*k_ld + h*64
  const void* k; int64_t k_ld; int64_t k_bs;
  const void* v; int64_t v_ld; int64_t v_bs;
  int Tk, splits;
  void* out; int64_t o_ld;                        // splits == 1
  void* part_o; float* part_ml;                   // splits > 1 (layouts of DecAttnArgs)
  // hand-off of q between workgroups: 8-byte granules {2 x fp16, tag} [R][D/2]; tag = ((*d_tick + 1 + epoch) << 6) | (layer + 1)
  unsigned long long* qg; const int* d_tick; int epoch, layer;
  int* err;                                       // counts bounded spins that ran out (never on a healthy device)
  int mode;                                       // bit 0: granules fetched through the scalar memory path (fused_mode())
  // optional phase 0 (out_w != null): x_io[r][:] += out_w . att_in[r] + out_b (attn.out + residual, model.py:153) by the
  // producer workgroups before their LayerNorm; x_io == xf (in place), pflags = D / 8 eight-byte flag words (zero-initialised
  // once; tags never repeat), att_in = fp16 [R][D] attention rows of the self attention
  const void* att_in; const void* out_w; const float* out_b; float* x_io; unsigned long long* pflags;
  WH_PROBE_FIELD
  WH_PROBE_KVHS_FIELD
};
int fused_mode(int kind);
bool xattn_supported(int D, int H, int R, int kv_group, int Tk, int splits);
hipError_t launch_xattn8(const XAttnArgs& a, hipStream_t stream);
// self attention of one decode step with LayerNorm + QKV projection + KV-cache append inside the launch
struct SAttnArgs {
  const float* xf; int64_t xf_ld;                 // fp32 residual rows [R][D]
  const void* W; const float* bias;               // fp16 [3D][D] = [query; key; value], fp32 [3D] (LayerNorm folded in)
  int D, H, R;
  void* kcache; void* vcache; int64_t cache_bs;   // this layer's self K / V [R][n_ctx][D]; row r appends at *d_pos - lag[r]
  const int* d_pos; const int* lag;               // lag may be null
  void* q_out;                                    // optional: the unscaled q rows [R][D] (what the two-launch form leaves)
  void* out; int64_t o_ld;                        // attention output [R][D] (when x_out is null)
  // optional third stage, attn.out + residual in the same launch: x_out[r][:] = xf[r][:] + out_w . attention[r] + out_b
  // (x_out must not alias xf); og = granules [R][D/2] the attention workgroups hand their outputs over in
  const void* out_w; const float* out_b; float* x_out; unsigned long long* og;
  unsigned long long* qg; const int* d_tick; int epoch, layer;   // granules [R][3D/2]; tag as in XAttnArgs
  int* err;
  int mode;                                       // bit 0: scalar-path polls
  WH_PROBE_FIELD
};
bool sattn_supported(int D, int H, int R, int n_ctx);
hipError_t launch_sattn8(const SAttnArgs& a, hipStream_t stream);

// ---- gemv.hip ------------------------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_LN = 1, PRO_COMBINE = 2 };
enum { EPI_STORE = 0, EPI_QKV = 1, EPI_RESID = 2, EPI_GELU = 3, EPI_F32 = 4 };
struct GemvArgs {
  // prologue
  int pro;
  const void* x; int64_t x_ld;                    // PRO_PLAIN: element type [R][K]
  const float* xf; int64_t xf_ld;                 // PRO_LN: fp32 residual rows [R][K]
  const float* ln_w; const float* ln_b;
  int ln_folded;                                  // PRO_LN: ln_w == 1, ln_b == 0 (folded into W / bias at load time)
  const void* part_o; const float* part_ml; int splits; int H;  // PRO_COMBINE: [S][R][H][64] element type, [S][R][H][2]
  // weights
  const void* W; const float* bias; int N; int K; int R;
  // gemv8_kernel only (fp16, R <= 24; every other kernel ignores them — set them only where gemv8_will_run says so):
  int x_frag;                                     // PRO_PLAIN: x holds FRAGMENT-ORDER rows (below), K columns; x_ld ignored
  int y_frag;                                     // EPI_STORE / EPI_GELU: write y in fragment order (N % 64 == 0); y_ld ignored
  int w_ordered;                                  // set by the launcher (gemv.hip): weight wave-loads in memory order + a turn through LDS
  // epilogue
  int epi;
  void* y; int64_t y_ld;                          // EPI_STORE / EPI_GELU (element type), EPI_F32 (float)
  float* resid; int64_t resid_ld;                 // EPI_RESID: resid[r][n] += y
  void* kcache; void* vcache; int64_t cache_bs; const int* d_pos; int D;  // EPI_QKV: row r appends at *d_pos - lag[r]
  const int* lag;                                 // EPI_QKV: per-row position lag (ragged prompts); may be null
  int* bump; int bump_by;                         // optional: *bump += bump_by once per launch (position counter)
  int* bump2;                                     // optional: *bump2 += 1 once per launch (the step tick of xattn.hip)
  int variant;                                    // 0 = shape heuristic; > 0 forces a kernel shape (tools/probe_decode)
  WH_PROBE_FIELD
};
hipError_t launch_gemv(const GemvArgs& a, int dtype, hipStream_t stream);
// Fragment order (fp16 activations that only travel from one launch of a decode step to the next: self-attention output ->
// attn.out, merged cross-attention output -> cross_attn.out, FC1 -> FC2).  A wave-load is 64 lanes x 16 bytes, and the memory
// pipeline of a CU takes ~58 cycles for one whose CONSECUTIVE lanes sit in different cache lines — the MFMA operand map read
// straight from row-major rows: lane 16 c + 8 half + i = row i, bytes 64 half + 16 c of a 128-byte block — against ~19.5 when
// lane l reads bytes [16 l, +16) of one contiguous KB (tools/ubench_ta.cpp, profiles/r06_fragment_order.txt: same 8 lines, same
// bytes).  With 2 - 3 row tiles a weight wave of a PRO_PLAIN launch issues 10 - 15 such x loads behind its 5 weight loads, so
// the PRODUCER writes the rows in the order the consumer's lanes want them: unit (row tile r / 8, K block k / 64) = 1 KB,
// lane 16 ((k & 31) >> 3) + 8 ((k >> 5) & 1) + r % 8 holds elements [k & ~7, +8).  Rows up to the next multiple of 8 must exist
// in the buffer (never written, read into output columns that are dropped).  FC2 of a 24-row step 11.0 -> 7.7 us, the D x D
// projections 4.9 -> 3.9; the same order for the WEIGHTS was measured too and is not used (D x D 4.9 -> 5.6 us: the 640-byte
// pieces of a 5-feature workgroup come off fewer HBM channels than five 128-byte lines 2.5 KB apart).
// element offset of x[r][k] in a fragment-order activation of K columns (K % 64 == 0)
static inline int64_t frag_index(int r, int k, int K) {
  return ((((int64_t)(r >> 3) * (K >> 6) + (k >> 6)) * 64 + 16 * ((k & 31) >> 3) + 8 * ((k >> 5) & 1) + (r & 7)) << 3) + (k & 7);
}
bool gemv8_will_run(int R, int N, int K, int pro);      // launch_gemv(R rows of fp16) goes to gemv8_kernel for this shape
// PRO_COMBINE's merge of the decode-attention partials as a launch of its own ([rows][H*64] in the element type)
hipError_t launch_merge_partials(const void* part_o, const float* part_ml, int splits, int R, int H, void* out,
                                 int64_t o_ld, int dtype, hipStream_t stream, int o_frag = 0);

// ---- sampling.hip --------------------------------------------------------------------------
struct SampleArgs {
  const float* logits; int64_t logits_ld;   // row r at logits + r*logits_ld, V entries
  int R, V;
  int64_t* tokens; int64_t token_stride;          // [R][stride], current length *d_ntok
  const int* d_ntok;          // device: number of tokens currently in the longest row (== cached positions)
  const int* lag;             // row r holds lag[r] fewer tokens, its sample_begin is lag[r] earlier; may be null
  int sample_begin, eot, timestamp_begin, no_timestamps, max_initial_ts, suppress_blank, blank_token;
  const uint8_t* suppress_mask;
  float* sum_logprobs;        // [R]
  int64_t* step_tokens;       // [R] next-step input tokens (may be null)
  int* d_alive_step;          // device: set to ntok by every row whose sampled token is not EOT
  float* partials;            // scratch: greedy_sample_scratch_bytes(R, V)
  float inv_temperature;      // 0: arg-max; > 0: sample from softmax(logits / T) (Gumbel-max, counter-based noise)
  uint32_t seed_lo, seed_hi;
  // optional: the final kernel also writes the next step's decoder input x_next[r][:] = tok_emb[token] + pos_emb[index of
  // the token] (model.py:236-240), so the step that follows needs no embedding launch; skipped when x_next is null
  float* x_next; const void* tok_emb; const float* pos_emb; int D, emb_f16, n_pos;
  // optional: per-row timestamp-rule state kept by the final kernel, {last sampled token is a timestamp, the one before
  // it is, value of the last sampled timestamp + 1 (0 = none), unused}, zeroed before the first sample of a sequence.
  // With it the partial kernel does not walk the row's sampled tokens (three dependent round trips per step).
  int* row_state;
};
size_t greedy_sample_scratch_bytes(int R, int V);
hipError_t launch_greedy_sample(const SampleArgs& a, hipStream_t stream);
hipError_t launch_no_speech(const float* logits_row0, int64_t row_stride, int R, int V, int no_speech,
                            float* out, hipStream_t stream);
// dst[r] = src[r*stride]
hipError_t launch_gather_tokens(const int64_t* src, int64_t stride, int R, int64_t* dst, hipStream_t stream);

// ---- beam.hip -------------------------------------------------------------------------------
constexpr int BEAM_KMAX = 9;   // candidates per row = beam + 1, beam <= 8
struct BeamArgs {
  const float* logits; int64_t logits_ld;   // row r at logits + r*logits_ld, V entries
  int R, V, G, K;                           // rows = segments x G beams; K = G + 1
  const int64_t* tokens_in; int64_t* tokens_out; int64_t token_stride;   // [R][stride] each; rows hold *d_ntok tokens
  const int* d_ntok;
  const int* lag;                           // row r holds lag[r] fewer tokens and its sample_begin is lag[r] earlier (ragged
                                            // prompts; equal within a beam group); may be null
  int sample_begin, eot, timestamp_begin, no_timestamps, max_initial_ts, suppress_blank, blank_token;
  const uint8_t* suppress_mask;
  float* sum_logprobs;                      // [R] in/out
  float* part_stat; float* part_val; int* part_idx;   // scratch (beam_scratch_carve)
  float* cand_lp; int* cand_tok;            // [R][BEAM_KMAX] top-K log-probabilities / tokens of every row
  int64_t* fin_tok; int* fin_len; float* fin_score; int* fin_count;   // [B][max_candidates][stride], [B][mc], [B][mc], [B]
  int max_candidates;
  int* src;                                 // [R] out: row whose KV cache the new beam continues
  // optional (both or neither): lcp [B][8][8] in/out = number of leading cache positions at which two rows of a segment
  // hold identical K/V (initialise every entry to INT_MAX-like: "everything so far"); copy_from [R] out = first position
  // row r has to take from src[r] (below it the bytes are already the same)
  int* lcp; int* copy_from;
  int64_t* step_tokens;                     // [R] out: next step's input tokens
  const int* done_prev; int* done_next;     // [B] completion flags written by the previous / this update
  int* d_applied;                           // number of updates applied (not frozen)
  int first;                                // 1: first update — all beams of a segment hold the same prefix
};
size_t beam_scratch_bytes(int R, int V);
void beam_scratch_carve(BeamArgs& a, void* base, int R, int V);
// filters + log_softmax + top-(G+1) of every row, then the candidate bookkeeping of every segment (3 launches)
hipError_t launch_beam_step(const BeamArgs& a, int B, hipStream_t stream);

// ---- timing.hip ----------------------------------------------------------------------------
hipError_t launch_median_filter(const float* x, float* out, int64_t rows, int n, int width,
                                hipStream_t stream);
hipError_t launch_dtw(const float* x, int N, int M, int8_t* trace, hipStream_t stream);
// batched find_alignment post-processing: clip b has ntok[b] token rows and nfr[b] frames (device arrays) inside slabs
// qk [clips][H][Tmax][Tk]; cost [clips][Nmax][Fmax] = -mean over heads of rows [row_begin, ntok[b] - row_tail);
// scratch: 2 * clips * H * Tmax * Fmax floats
hipError_t launch_align_batch(const float* qk, const int* d_ntok, const int* d_nfr, int clips, int H, int Tmax, int Tk,
                              int Fmax, int width, int row_begin, int row_tail, float qk_scale, float* cost, int Nmax,
                              float* scratch, hipStream_t stream);
// dtw of every clip's cost matrix; trace of clip b dense [(N_b + 1)][(F_b + 1)] at trace + b * trace_bs
hipError_t launch_dtw_batch(const float* cost, const int* d_ntok, const int* d_nfr, int clips, int Tmax, int Fmax,
                            int row_begin, int row_tail, int Nmax, int8_t* trace, int64_t trace_bs, hipStream_t stream);
// back-trace of every clip's trace (dense [(rows_b + 1)][(cols_b + 1)] at trace + b * trace_bs; d_rows / d_cols device
// arrays): jumps [clips][jump_stride] = frame at which each text index is first reached; optional right-aligned path
// [clips][2][path_stride] + path_len [clips]
hipError_t launch_dtw_backtrace_batch(const int8_t* trace, int64_t trace_bs, const int* d_rows, const int* d_cols, int clips,
                                      int max_rows, int max_cols, int* jumps, int64_t jump_stride, int* path,
                                      int64_t path_stride, int* path_len, hipStream_t stream);
// qk [H][T][Tk] -> softmax over first F frames, z-norm over tokens, median(width), -mean over heads of rows
// [row_begin,row_end) -> out [rows][F]; scratch: 2*H*T*F floats + 2 ints
hipError_t launch_align_matrix(const float* qk, int H, int T, int Tk, int F, int width, int row_begin,
                               int row_end, float qk_scale, float* out, float* scratch, hipStream_t stream);

}  // namespace whk
