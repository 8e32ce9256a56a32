// api.cpp — the C ABI of libwhisper_hip.so (see include/whisper_hip.h) and the host-side orchestration
// of the encoder pass, the decoder prefill, the hipGraph-captured decode step and the fused greedy loop.
// Host code only: all device work is in the .hip files behind kernels.h.
#include "../../include/whisper_hip.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

using namespace whk;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local hipError_t g_last_hip = hipSuccess;

#define HIPCHK(expr)                                   \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) { g_last_hip = _e; return WH_ERR_HIP; } \
  } while (0)

extern "C" int wh_abi_version(void) { return WH_ABI_VERSION; }
extern "C" const char* wh_status_string(int s) {
  switch (s) {
    case WH_OK: return "ok";
    case WH_ERR_ARG: return "invalid argument";
    case WH_ERR_WORKSPACE: return "workspace too small";
    case WH_ERR_HIP: return "HIP runtime error";
    case WH_ERR_STATE: return "invalid call sequence";
    case WH_ERR_LIMIT: return "compiled-in limit exceeded";
    case WH_ERR_HANDOFF: return "in-kernel hand-off timed out (results invalid)";
    case WH_RUNNING: return "loop still running (call wh_task_poll again)";
    default: return "unknown status";
  }
}
extern "C" int wh_last_hip_error(void) { return (int)g_last_hip; }
extern "C" const char* wh_last_hip_error_string(void) { return hipGetErrorString(g_last_hip); }

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {   // bump allocator over the caller's workspace; with base == nullptr it only sizes
  char* base; size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  void* take(size_t bytes) {
    off = align_up(off, 256);
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

struct wh_model {
  wh_dims d;
  int dtype;         // WH_F32 / WH_F16
  int esize;
  int kc1;           // padded K of conv1 GEMM
  wh_model_weights w;
  std::vector<wh_layer_weights> enc, dec;
};

// ------------------------------------------------------------------------------------------------
// log-mel
// ------------------------------------------------------------------------------------------------
static std::mutex g_tab_mutex;
static float* g_mel_tables[64] = {nullptr};

static int mel_tables(float** out) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return WH_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  if (!g_mel_tables[dev]) {
    std::vector<float> h(1200);
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < 400; ++i) {
      h[i] = (float)cos(2.0 * PI * i / 400.0);
      h[400 + i] = (float)sin(2.0 * PI * i / 400.0);
      h[800 + i] = (float)(0.5 - 0.5 * cos(2.0 * PI * i / 400.0));   // torch.hann_window(400), periodic
    }
    float* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 1200 * sizeof(float)));
    HIPCHK(hipMemcpy(d, h.data(), 1200 * sizeof(float), hipMemcpyHostToDevice));
    g_mel_tables[dev] = d;
  }
  *out = g_mel_tables[dev];
  return WH_OK;
}

extern "C" int wh_log_mel(const float* audio, int64_t n_samples, int batch, int n_mels,
                          const float* filters, float* out, void* scratch, void* stream) {
  if (!audio || !filters || !out || !scratch) return WH_ERR_ARG;
  if (n_samples <= 200 || batch <= 0 || n_mels <= 0 || n_mels > 128) return WH_ERR_ARG;
  if (n_samples / 160 <= 0) return WH_ERR_ARG;
  float* tables = nullptr;
  int rc = mel_tables(&tables);
  if (rc != WH_OK) return rc;
  HIPCHK(launch_log_mel(audio, n_samples, batch, n_mels, filters, tables, out, scratch, (hipStream_t)stream));
  return WH_OK;
}

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
extern "C" int wh_model_create(const wh_dims* dims, int dtype, const wh_model_weights* weights, wh_model** out) {
  if (!dims || !weights || !out) return WH_ERR_ARG;
  if (dtype != WH_F32 && dtype != WH_F16) return WH_ERR_ARG;
  const wh_dims& d = *dims;
  if (d.n_audio_state != d.n_text_state) return WH_ERR_ARG;
  if (d.n_audio_state % 64 != 0 || d.n_audio_state / 64 != d.n_audio_head || d.n_text_state / 64 != d.n_text_head)
    return WH_ERR_ARG;   // d_head == 64 for every Whisper checkpoint
  if (d.n_audio_state > 2048 || d.n_mels > 128 || d.n_audio_ctx > 1536 || d.n_text_ctx > 1536) return WH_ERR_LIMIT;
  // `flags` changes how the kernels read the blob: a caller built against an older struct (or one that left the field
  // uninitialised) must fail here, not produce silently wrong LayerNorms / attention scores
  if (weights->flags & ~(WH_WEIGHTS_DEC_LN_FOLDED | WH_WEIGHTS_ENC_QK_SCALED)) return WH_ERR_ARG;
  if (dtype == WH_F32 && weights->flags != 0) return WH_ERR_ARG;     // the strict-parity engine keeps the reference's order
  wh_model* m = new (std::nothrow) wh_model();
  if (!m) return WH_ERR_ARG;
  m->d = d;
  m->dtype = dtype;
  m->esize = dtype == WH_F16 ? 2 : 4;
  m->kc1 = (int)align_up((size_t)3 * d.n_mels, 64);
  m->w = *weights;
  m->enc.assign(weights->enc_layers, weights->enc_layers + d.n_audio_layer);
  m->dec.assign(weights->dec_layers, weights->dec_layers + d.n_text_layer);
  m->w.enc_layers = m->enc.data();
  m->w.dec_layers = m->dec.data();
  *out = m;
  return WH_OK;
}
extern "C" void wh_model_destroy(wh_model* m) { delete m; }

// ------------------------------------------------------------------------------------------------
// GEMM convenience
// ------------------------------------------------------------------------------------------------
static hipError_t gemm(const wh_model* m, const void* A, int64_t lda, const void* W, int K, void* C, int64_t ldc,
                       int M, int N, const float* bias, int act, const float* res, int64_t ldr, bool out_f32,
                       hipStream_t s) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = ldc;
  g.bias = bias; g.act = act; g.res = res; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K;
  return launch_gemm(g, m->dtype, out_f32 || m->dtype == WH_F32, 1, s);
}

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
struct EncWs {
  void *melT, *c1, *xn, *qkv, *vt, *att, *h;
  float* x;
  size_t total;
};
static const int VT_LD = 1536;

static EncWs enc_carve(const wh_model* m, int B, void* base) {
  const wh_dims& d = m->d;
  const size_t es = m->esize, T = d.n_audio_ctx, F = 2 * T, D = d.n_audio_state;
  Carver c(base);
  EncWs w;
  w.melT = c.take(((size_t)B * (F + 2) * d.n_mels + 256) * es);
  w.c1 = c.take(((size_t)B * (F + 2) * D + 256) * es);
  w.x = (float*)c.take((size_t)B * T * D * 4);
  w.xn = c.take((size_t)B * T * D * es);
  w.qkv = c.take((size_t)B * T * 3 * D * es);
  w.vt = m->dtype == WH_F16 ? c.take((size_t)B * D * VT_LD * es) : nullptr;
  w.att = c.take((size_t)B * T * D * es);
  w.h = c.take((size_t)B * T * 4 * D * es);
  w.total = align_up(c.off, 256);
  return w;
}

extern "C" size_t wh_encoder_workspace_bytes(const wh_model* m, int batch) {
  if (!m || batch <= 0) return 0;
  return enc_carve(m, batch, nullptr).total;
}

extern "C" int wh_encode(const wh_model* m, const void* mel, int mel_is_f16, int B, void* out, void* workspace,
                         size_t workspace_bytes, void* stream_) {
  if (!m || !mel || !out || !workspace || B <= 0) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  const wh_dims& d = m->d;
  const int T = d.n_audio_ctx, F = 2 * T, D = d.n_audio_state, H = d.n_audio_head, NM = d.n_mels;
  const size_t es = m->esize;
  EncWs w = enc_carve(m, B, workspace);
  if (w.total > workspace_bytes) return WH_ERR_WORKSPACE;
  const bool f16 = m->dtype == WH_F16;

  // ---- conv1 / conv2 as GEMMs over overlapping rows (whisper/model.py:193-194) -------------
  HIPCHK(hipMemsetAsync((char*)w.melT + (size_t)B * (F + 2) * NM * es, 0, 256 * es, s));   // slack read by padded K
  HIPCHK(launch_mel_transpose(mel, mel_is_f16, B, NM, F, w.melT, m->dtype, s));
  for (int b = 0; b < B; ++b) {   // zero pad rows 0 and F+1 of the conv1 output
    char* base = (char*)w.c1 + (size_t)b * (F + 2) * D * es;
    HIPCHK(hipMemsetAsync(base, 0, (size_t)D * es, s));
    HIPCHK(hipMemsetAsync(base + (size_t)(F + 1) * D * es, 0, (size_t)D * es, s));
  }
  {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = w.melT; g.lda = NM; g.a_bs = (int64_t)(F + 2) * NM;
    g.W = m->w.conv1_w; g.ldw = m->kc1;
    g.C = (char*)w.c1 + (size_t)D * es; g.ldc = D; g.c_bs = (int64_t)(F + 2) * D;
    g.bias = m->w.conv1_b; g.act = 1;
    g.M = F; g.N = D; g.K = m->kc1;
    HIPCHK(launch_gemm(g, m->dtype, !f16, B, s));
  }
  {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = w.c1; g.lda = 2 * D; g.a_bs = (int64_t)(F + 2) * D;
    g.W = m->w.conv2_w; g.ldw = 3 * D;
    g.C = w.x; g.ldc = D; g.c_bs = (int64_t)T * D;
    g.bias = m->w.conv2_b; g.act = 1;
    g.res = m->w.enc_pos; g.ldr = D; g.r_bs = 0;      // + positional_embedding (model.py:198)
    g.M = T; g.N = D; g.K = 3 * D;
    HIPCHK(launch_gemm(g, m->dtype, 1, B, s));
  }
  if (f16) HIPCHK(hipMemsetAsync(w.vt, 0, (size_t)B * D * VT_LD * es, s));   // zero key padding of V^T

  const int M = B * T;
  for (int l = 0; l < d.n_audio_layer; ++l) {
    const wh_layer_weights& L = m->enc[l];
    HIPCHK(launch_layernorm(w.x, D, L.attn_ln_w, L.attn_ln_b, w.xn, D, M, D, m->dtype, s));
    if (f16) {
      // Q,K row-major; V produced transposed (V^T = Wv · X^T) for the PV MFMA operand
      HIPCHK(gemm(m, w.xn, D, L.qkv_w, D, w.qkv, 2 * D, M, 2 * D, L.qkv_b, 0, nullptr, 0, false, s));
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = (const char*)L.qkv_w + (size_t)2 * D * D * es; g.lda = D; g.a_bs = 0;
      g.W = w.xn; g.ldw = D; g.w_bs = (int64_t)T * D;
      g.C = w.vt; g.ldc = VT_LD; g.c_bs = (int64_t)D * VT_LD;
      g.bias = L.qkv_b + 2 * D; g.bias_on_m = 1;
      g.M = D; g.N = T; g.K = D;
      HIPCHK(launch_gemm(g, m->dtype, 0, B, s));
      HIPCHK(launch_attn_flash_f16(w.qkv, 2 * D, (int64_t)T * 2 * D, (const char*)w.qkv + (size_t)D * es, 2 * D,
                                   (int64_t)T * 2 * D, w.vt, VT_LD, (int64_t)D * VT_LD, w.att, D,
                                   (int64_t)T * D, B, H, T, (m->w.flags & WH_WEIGHTS_ENC_QK_SCALED) ? 1 : 0, s));
    } else {
      HIPCHK(gemm(m, w.xn, D, L.qkv_w, D, w.qkv, 3 * D, M, 3 * D, L.qkv_b, 0, nullptr, 0, false, s));
      AttnArgs a; memset(&a, 0, sizeof(a));
      a.q = w.qkv; a.q_ld = 3 * D; a.q_bs = (int64_t)T * 3 * D;
      a.k = (const char*)w.qkv + (size_t)D * es; a.k_ld = 3 * D; a.k_bs = a.q_bs;
      a.v = (const char*)w.qkv + (size_t)2 * D * es; a.v_ld = 3 * D; a.v_bs = a.q_bs;
      a.out = w.att; a.o_ld = D; a.o_bs = (int64_t)T * D;
      a.H = H; a.Tq = T; a.Tk = T; a.causal = 0; a.kv_group = 1;
      HIPCHK(launch_attn_generic(a, B, m->dtype, s));
    }
    HIPCHK(gemm(m, w.att, D, L.out_w, D, w.x, D, M, D, L.out_b, 0, w.x, D, true, s));
    HIPCHK(launch_layernorm(w.x, D, L.mlp_ln_w, L.mlp_ln_b, w.xn, D, M, D, m->dtype, s));
    HIPCHK(gemm(m, w.xn, D, L.fc1_w, D, w.h, 4 * D, M, 4 * D, L.fc1_b, 1, nullptr, 0, false, s));
    HIPCHK(gemm(m, w.h, 4 * D, L.fc2_w, 4 * D, w.x, D, M, D, L.fc2_b, 0, w.x, D, true, s));
  }
  HIPCHK(launch_layernorm(w.x, D, m->w.enc_ln_post_w, m->w.enc_ln_post_b, out, D, M, D, m->dtype, s));
  return WH_OK;
}

// ------------------------------------------------------------------------------------------------
// decoding task
// ------------------------------------------------------------------------------------------------
struct wh_loop;
struct wh_task {
  const wh_model* m;
  int B, G, R, Tmax, flags;
  int pos;                 // host mirror of *d_pos
  bool audio_set;
  // the decode step exists in two captured forms: [0] starts with the token embedding of `step_tokens` (host-driven
  // steps, beam search), [1] starts at layer 0 because the greedy sampler of the previous step already wrote x
  int steps_eager[2];      // decode steps launched without a graph (first one warms up attributes)
  int handoff_fallbacks;   // times a fused loop was re-run on the two-launch kernels after a hand-off time-out
  hipGraph_t graph[2]; hipGraphExec_t graph_exec[2];
  // device buffers (carved from the caller's workspace)
  void* cross_kv;          // [L][B*Ta][2D]
  void* cross_vt;          // beam groups, fp16: [L][B][D][vt_ld], V transposed (row = head dim, column = key) for the
  int vt_ld;               //   matrix-core group attention; vt_ld = splits x the kernel's rounded key chunk >= Ta
  void* self_k; void* self_v;   // [L][R][n_ctx][D]
  void* spare_k; void* spare_v; // beam reorder staging (G > 1)
  float* x; void* xn; void* qkv; void* att; void* h; void* qbuf;
  void* part_o; float* part_ml;
  float* logits;           // [R][V] step logits / [R][2][V] greedy prefill logits
  float* xsel; void* xseln;
  int* d_pos; int* d_alive; int* d_sel; int* d_src;
  int* d_lag;              // [R] ragged prompts: row r is lag[r] tokens shorter than the longest row (zeros otherwise)
  int* h_lag;              // host copy
  int* h_poll;             // pinned host words the fused loops copy the completion counter / flags / results into
  hipEvent_t poll_event;   // recorded behind that copy: the loop waits for it two steps later, with work already queued
  hipEvent_t done_event;   // recorded behind a fused loop's last copies
  struct wh_loop* loop;    // state of the fused greedy / beam loop (wh_task_*_begin / wh_task_poll)
  int loop_kind;           // LOOP_IDLE, or the loop that is running: every entry point but wh_task_poll is refused meanwhile
  int* h_sel;              // pinned: row indices of the prefill's selected positions (source of an async copy)
  hipEvent_t sel_event;    // recorded behind that copy; waited for before h_sel is rewritten
  bool lag_on;
  bool needs_reset;        // created, position counter / lag not zeroed yet
  std::atomic<int> busy;   // handles are not thread-safe: a second thread entering while a call runs gets WH_ERR_STATE
  int64_t* step_tokens;
  float* samp_part;        // greedy sampler stage-1 partials
  int* samp_state;         // [R][4] timestamp-rule state of the fused greedy loop
  void* beam_scratch;      // beam search partials / candidates (G > 1)
  int* beam_flags;         // [2][B] completion flags + [1] applied-update counter
  int* beam_lcp;           // [B][8][8] shared-history lengths of the rows of a segment + [R] first position to copy
  void* qcap;              // [L][R*Tcap][D] captured cross-attention queries
  int cross_splits, self_splits;
  // fused query projection + cross attention (xattn.hip; fp16, <= 8 rows): q granules, the step tick, the error word
  unsigned long long* xq_gran; int* d_tick; int* d_err;
  unsigned long long* sq_gran;   // the same for self attention + QKV projection (q, new k, new v)
  unsigned long long* so_gran;   // ... and for the attention outputs handed to attn.out inside the same launch
  int* merge_cnt;                // [R][H] tickets of the cross attention's in-launch merge (attention.hip), 0 between launches
  float* x2;                     // second residual-stream buffer: the fused self-attention launch reads x and writes x2 (or back)
  bool fused_xattn, fused_sattn, fused_out;   // fused_out: attn.out + residual inside the self-attention launch (dev builds)
  bool fused_xout;               // attn.out + residual as phase 0 of the fused cross-attention launch (xattn.hip, OUT0)
  size_t total;
};

// state of a fused greedy / beam loop between wh_task_*_begin and the wh_task_poll that reports its end
enum { LOOP_IDLE = 0, LOOP_GREEDY = 1, LOOP_BEAM = 2 };
enum { PH_STEPPING = 0, PH_DRAINING = 1 };
struct wh_loop {
  int phase;
  hipStream_t s;
  // the caller's arguments (kept for the hand-off fallback's second run)
  wh_beam_params bp;                 // .rules are the greedy parameters
  int64_t* tokens; int64_t token_stride; int sot_index, no_speech_token;
  float* sum_logprobs; float* no_speech_probs;
  int64_t* fin_tokens; int32_t* fin_len; float* fin_scores; int32_t* fin_count;
  // loop state
  SampleArgs sa; BeamArgs ba;
  bool fused_embed, pending, wait_due;
  int ntok, steps, ntok_at_copy, cur;
  int n_tokens;                      // result
};

// One call at a time per task handle (include/whisper_hip.h: "not thread-safe per handle"), enforced instead of only
// documented: the second caller is refused, nothing is corrupted.
struct TaskGuard {
  wh_task* t; bool ok;
  // loop_call: wh_task_poll, the one call a task takes while a fused loop begun with wh_task_*_begin is running
  explicit TaskGuard(wh_task* task, bool loop_call = false) : t(task), ok(false) {
    if (t) {
      int expected = 0; ok = t->busy.compare_exchange_strong(expected, 1);
      if (ok && !loop_call && t->loop_kind != 0) { t->busy.store(0); ok = false; }
    }
  }
  ~TaskGuard() { if (t && ok) t->busy.store(0); }
};
#define TASK_ENTER(task) TaskGuard _guard(task); if ((task) && !_guard.ok) return WH_ERR_STATE

// the few-row prefill (see prefill_impl): bounds used by the workspace carve-up as well
static const int SKINNY_ROWS = 96;          // rows x prompt tokens (round 6: 48 -> 96, so that the prompt pass of a 24-row chain — 24 x 4 tokens —
                                            // stays off 128 x 128 GEMM tiles that fill 10 workgroups: 10 -> ~4 ms per chain)
static const int SKINNY_LOGIT_ROWS = 48;    // selected rows the streaming logits launch takes

// key-range splits of the decode attention: enough for the register-resident tile to hold a split
// (attn_decode_capacity) and enough workgroups (>= ~640) to cover the 256 CUs with loads in flight
static int pick_splits(int R, int H, int max_keys, int dtype) {
  const int cap = attn_decode_capacity(dtype);
  const int unit = dtype == WH_F16 ? 32 : 16;          // chunk rounding inside the kernel (keys per 4-wave round)
  int s = (480 + R * H - 1) / (R * H);      // measured: 480 workgroups of 512 keys beat 640 x 384 (probe_decode)
  if (s < 1) s = 1;
  if (s > DEC_ATTN_MAX_SPLITS) s = DEC_ATTN_MAX_SPLITS;
  while (s < DEC_ATTN_MAX_SPLITS && ((max_keys + s - 1) / s + unit - 1) / unit * unit > cap) ++s;
  while (s > 1 && (max_keys + s - 2) / (s - 1) <= unit) --s;   // never more splits than 32-key chunks
  return s;
}

static void task_carve(wh_task* t, void* base) {
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const size_t es = m->esize, D = d.n_text_state, L = d.n_text_layer, Ta = d.n_audio_ctx, C = d.n_text_ctx;
  const size_t R = t->R, Mx = (size_t)t->R * t->Tmax, V = d.n_vocab, H = d.n_text_head;
  Carver c(base);
  t->cross_kv = c.take(L * t->B * Ta * 2 * D * es);
  t->cross_vt = nullptr; t->vt_ld = 0;
  if (t->G > 1 && m->dtype == WH_F16) {
    const int S = pick_splits(t->R, d.n_text_head, d.n_audio_ctx, m->dtype);
    const int chunk = (((int)Ta + S - 1) / S + 127) / 128 * 128;       // attn_decode_group_mfma_kernel's key chunk
    t->vt_ld = S * chunk;
    t->cross_vt = c.take(L * t->B * D * (size_t)t->vt_ld * es);
  } else if ((t->flags & WH_TASK_CAPTURE_Q) && m->dtype == WH_F16) {
    // word-timestamp tasks teacher-force ~200 tokens per clip: their prefill's cross attention (T0 x 1500 per row) runs
    // on the matrix-core flash kernel, which wants V transposed as well
    t->vt_ld = ((int)Ta + 127) / 128 * 128;
    t->cross_vt = c.take(L * t->B * D * (size_t)t->vt_ld * es);
  }
  t->self_k = c.take(L * R * C * D * es);
  t->self_v = c.take(L * R * C * D * es);
  t->spare_k = t->G > 1 ? c.take(R * C * D * es) : nullptr;
  t->spare_v = nullptr;
  t->x = (float*)c.take(Mx * D * 4);
  t->xn = c.take(Mx * D * es);
  t->qkv = c.take(Mx * 3 * D * es);
  // (att, h may be written in fragment order — kernels.h — whose units span 8 rows: whole row tiles must exist)
  const size_t Mx8 = std::max(Mx, (R + 7) / 8 * 8);
  t->att = c.take(Mx8 * D * es);
  t->h = c.take(Mx8 * 4 * D * es);
  t->qbuf = c.take(R * D * es);
  // the few-row prefill (<= SKINNY_ROWS rows x tokens) runs its cross attention through the decode kernel: one partial per (row, token)
  size_t Rp = Mx < (size_t)SKINNY_ROWS ? Mx : (size_t)SKINNY_ROWS;
  if (Rp < R) Rp = R;
  if (Rp < 48) Rp = 48;
  t->part_o = c.take(Rp * H * DEC_ATTN_MAX_SPLITS * 64 * 4);
  t->part_ml = (float*)c.take(Rp * H * DEC_ATTN_MAX_SPLITS * 2 * 4);
  t->logits = (float*)c.take(R * 2 * V * 4);
  t->xsel = (float*)c.take(Mx * D * 4);
  t->xseln = c.take(Mx * D * es);
  t->d_pos = (int*)c.take(256);
  t->d_alive = (int*)c.take(256);
  t->d_sel = (int*)c.take(Mx * 4);
  t->d_src = (int*)c.take(R * 4);
  t->d_lag = (int*)c.take(R * 4);
  t->step_tokens = (int64_t*)c.take(R * 8);
  t->samp_part = (float*)c.take(greedy_sample_scratch_bytes((int)R, (int)V));
  t->samp_state = (int*)c.take(R * 16);
  t->beam_scratch = t->G > 1 ? c.take(beam_scratch_bytes((int)R, (int)V)) : nullptr;
  t->beam_flags = t->G > 1 ? (int*)c.take((2 * (size_t)t->B + 1) * 4) : nullptr;
  t->beam_lcp = t->G > 1 ? (int*)c.take(((size_t)t->B * 64 + R) * 4) : nullptr;
  t->qcap = (t->flags & WH_TASK_CAPTURE_Q) ? c.take(L * R * C * D * es) : nullptr;
  t->xq_gran = (unsigned long long*)c.take(R * (D / 2) * 8);
  t->sq_gran = (unsigned long long*)c.take(R * (3 * D / 2) * 8);
  t->so_gran = (unsigned long long*)c.take(R * (D / 2) * 8);
  t->x2 = (float*)c.take(R * D * 4);
  t->d_tick = (int*)c.take(256);
  t->d_err = t->d_tick + 16;
  t->merge_cnt = (int*)c.take(R * H * 4);
  t->total = align_up(c.off, 256);
}

extern "C" size_t wh_task_workspace_bytes(const wh_model* m, int n_audio, int n_group, int max_prefill_tokens,
                                          int flags) {
  if (!m || n_audio <= 0 || n_group <= 0 || max_prefill_tokens <= 0) return 0;
  wh_task t; memset((void*)&t, 0, sizeof(t));
  t.m = m; t.B = n_audio; t.G = n_group; t.R = n_audio * n_group; t.Tmax = max_prefill_tokens; t.flags = flags;
  task_carve(&t, nullptr);
  return t.total;
}

extern "C" int wh_task_create(const wh_model* m, int n_audio, int n_group, int max_prefill_tokens, int flags,
                              void* workspace, size_t workspace_bytes, wh_task** out) {
  if (!m || !workspace || !out || n_audio <= 0 || n_group <= 0 || max_prefill_tokens <= 0) return WH_ERR_ARG;
  if (max_prefill_tokens > m->d.n_text_ctx) return WH_ERR_ARG;
  wh_task* t = new (std::nothrow) wh_task();
  if (!t) return WH_ERR_ARG;
  memset((void*)t, 0, sizeof(*t));
  t->m = m; t->B = n_audio; t->G = n_group; t->R = n_audio * n_group; t->Tmax = max_prefill_tokens; t->flags = flags;
  task_carve(t, workspace);
  if (t->total > workspace_bytes) { delete t; return WH_ERR_WORKSPACE; }
  t->cross_splits = pick_splits(t->R, m->d.n_text_head, m->d.n_audio_ctx, m->dtype);
  {  // self-attention: the split count must hold for every cached length up to n_text_ctx
    const int cap = attn_decode_capacity(m->dtype);
    t->self_splits = (m->d.n_text_ctx + cap - 1) / cap;
  }
  t->fused_xattn = !(flags & WH_TASK_TWO_LAUNCH_CROSS) && m->dtype == WH_F16 && (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) &&
                   xattn_supported(m->d.n_text_state, m->d.n_text_head, t->R, t->G, m->d.n_audio_ctx, t->cross_splits);
  t->fused_sattn = (flags & WH_TASK_FUSED_SELF) && !(flags & WH_TASK_TWO_LAUNCH_SELF) && m->dtype == WH_F16 && (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) && t->self_splits == 1 &&
                   sattn_supported(m->d.n_text_state, m->d.n_text_head, t->R, m->d.n_text_ctx);
  // attn.out + residual as phase 0 of the cross-attention launch (needs plain attention rows: one key split): a rejected
  // experiment, slower than the two launches (xattn.hip, out0_issue) — development builds only, WH_FUSED_XOUT=1
#ifdef WH_DEV
  t->fused_xout = t->fused_xattn && t->self_splits == 1 && WH_DEV_FLAG("WH_FUSED_XOUT");
#else
  t->fused_xout = false;
#endif
  {
    // attn.out inside the self-attention launch: built, bit-identical, and slower than its own launch (the gather of 160
    // (row, head) outputs by the projection workgroups costs ~4 us after the attention, and its polling slows the
    // projection phase of the same launch): a development-build experiment (-DWH_DEV: WH_FUSED_OUT=1 / flag bit 8)
#ifdef WH_DEV
    t->fused_out = t->fused_sattn && (WH_DEV_FLAG("WH_FUSED_OUT") || (flags & 8 /* development builds: WH_TASK_FUSE_OUT */));
#else
    t->fused_out = false;                            // the stage is not compiled into the shipped kernels
#endif
  }
  t->h_lag = (int*)calloc((size_t)t->R, sizeof(int));
  if (!t->h_lag) { delete t; return WH_ERR_ARG; }
  // no device work here: the position counter and the lag array are zeroed by the first wh_task_reset, which the
  // caller issues on the stream the workspace is valid on
  t->needs_reset = true;
  t->busy.store(0);
  *out = t;
  return WH_OK;
}

extern "C" void wh_task_destroy(wh_task* t) {
  if (!t) return;
  for (int i = 0; i < 2; ++i) {
    if (t->graph_exec[i]) (void)hipGraphExecDestroy(t->graph_exec[i]);
    if (t->graph[i]) (void)hipGraphDestroy(t->graph[i]);
  }
  free(t->h_lag);
  if (t->h_poll) (void)hipHostFree(t->h_poll);
  if (t->h_sel) (void)hipHostFree(t->h_sel);
  if (t->poll_event) (void)hipEventDestroy(t->poll_event);
  if (t->done_event) (void)hipEventDestroy(t->done_event);
  if (t->sel_event) (void)hipEventDestroy(t->sel_event);
  delete t->loop;
  delete t;
}

extern "C" int wh_task_position(const wh_task* t) { return t ? t->pos : -1; }

extern "C" int wh_task_info(wh_task* t, int what, void* stream) {
  if (!t) return -1;
  if (what == 0) return t->fused_xattn ? 1 : 0;
  if (what == 2) return t->fused_sattn ? 1 : 0;
  if (what == 3) return (t->fused_out || t->fused_xout) ? 1 : 0;
  if (what == 4) return t->handoff_fallbacks;
  if (what == 1) {                       // hand-off timeouts of the fused cross attention since the task was created
    if (t->needs_reset) return 0;
    int v = 0;
    if (hipMemcpyAsync(&v, t->d_err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return v;
  }
  return -1;
}

static int task_reset_impl(wh_task* t, void* stream_);
extern "C" int wh_task_reset(wh_task* t, void* stream_) {
  TASK_ENTER(t);
  return task_reset_impl(t, stream_);
}
static int task_reset_impl(wh_task* t, void* stream_) {
  if (!t) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  HIPCHK(hipMemsetAsync(t->d_pos, 0, 4, s));     // stream-ordered: no host or device-wide synchronisation
  t->pos = 0;
  if (t->needs_reset) {
    // the workspace arrives uninitialised: no granule may carry a tag that looks valid, and the tick starts at 0 (it
    // only ever counts up afterwards, so tags never repeat while the task lives)
    HIPCHK(hipMemsetAsync(t->xq_gran, 0, (size_t)t->R * (t->m->d.n_text_state / 2) * 8, s));
    HIPCHK(hipMemsetAsync(t->sq_gran, 0, (size_t)t->R * (3 * t->m->d.n_text_state / 2) * 8, s));
    HIPCHK(hipMemsetAsync(t->so_gran, 0, (size_t)t->R * (t->m->d.n_text_state / 2) * 8, s));
    HIPCHK(hipMemsetAsync(t->d_tick, 0, 256, s));
    HIPCHK(hipMemsetAsync(t->merge_cnt, 0, (size_t)t->R * t->m->d.n_text_head * 4, s));
    if (t->cross_vt)        // pad columns of the transposed cross-attention V: finite forever after (wh_task_set_audio)
      HIPCHK(hipMemsetAsync(t->cross_vt, 0, (size_t)t->m->d.n_text_layer * t->B * t->m->d.n_text_state * t->vt_ld * t->m->esize, s));
  }
  if (t->lag_on || t->needs_reset) {
    HIPCHK(hipMemsetAsync(t->d_lag, 0, (size_t)t->R * 4, s));
    memset(t->h_lag, 0, (size_t)t->R * sizeof(int));
    t->lag_on = false;
  }
  t->needs_reset = false;
  return WH_OK;
}

// Ragged prompts: row r's token sequence is the longest row's shifted left by lag[r] (a shorter leading prompt).
// Set before the prefill; the prefill then reads T0 tokens per row of which the last lag[r] are padding (any valid
// id; causality keeps them out of the real positions and the decode steps overwrite their cache slots), and every
// later step appends row r at position - lag[r].  Cleared by wh_task_reset.
extern "C" int wh_task_set_lag(wh_task* t, const int32_t* lag, void* stream) {
  TASK_ENTER(t);
  if (!t) return WH_ERR_ARG;
  if (t->pos != 0) return WH_ERR_STATE;
  if (t->flags & WH_TASK_CAPTURE_Q) return WH_ERR_STATE;      // captured queries are indexed by the common position
  if (t->needs_reset) { const int rc = task_reset_impl(t, stream); if (rc != WH_OK) return rc; }
  bool any = false;
  for (int r = 0; r < t->R; ++r) {
    const int v = lag ? lag[r] : 0;
    if (v < 0 || v >= t->Tmax) return WH_ERR_ARG;
    t->h_lag[r] = v;
    any = any || v != 0;
  }
  HIPCHK(hipMemcpyAsync(t->d_lag, t->h_lag, (size_t)t->R * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  t->lag_on = any;
  return WH_OK;
}

extern "C" int wh_task_set_audio(wh_task* t, const void* features, void* stream_) {
  TASK_ENTER(t);
  if (!t || !features) return WH_ERR_ARG;
  if (t->needs_reset) { const int rc = task_reset_impl(t, stream_); if (rc != WH_OK) return rc; }
  hipStream_t s = (hipStream_t)stream_;
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state, Ta = d.n_audio_ctx, M = t->B * Ta;
  const size_t es = m->esize;
  for (int l = 0; l < d.n_text_layer; ++l) {
    const wh_layer_weights& L = m->dec[l];
    void* C = (char*)t->cross_kv + (size_t)l * M * 2 * D * es;
    HIPCHK(gemm(m, features, D, L.ckv_w, D, C, 2 * D, M, 2 * D, L.ckv_b, 0, nullptr, 0, false, s));
  }
  if (t->cross_vt) {
    // V^T = W_v . X^T per audio (bias along rows), as in the encoder; the pad columns [Ta, vt_ld) only have to be finite:
    // the whole buffer is zeroed once, when the workspace is first used (task_reset_impl) — the GEMM below never writes
    // them (a 2-D memset of 36 columns x 330 k rows per call cost 28 ms on large-v3 x 8 clips)
    for (int l = 0; l < d.n_text_layer; ++l) {
      const wh_layer_weights& L = m->dec[l];
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = (const char*)L.ckv_w + (size_t)D * D * es; g.lda = D; g.a_bs = 0;
      g.W = features; g.ldw = D; g.w_bs = (int64_t)Ta * D;
      g.C = (char*)t->cross_vt + (size_t)l * t->B * D * t->vt_ld * es; g.ldc = t->vt_ld; g.c_bs = (int64_t)D * t->vt_ld;
      g.bias = L.ckv_b + D; g.bias_on_m = 1;
      g.M = D; g.N = Ta; g.K = D;
      HIPCHK(launch_gemm(g, m->dtype, 0, t->B, s));
    }
  }
  t->audio_set = true;
  return WH_OK;
}

static inline void* self_k_layer(const wh_task* t, int l) {
  const wh_dims& d = t->m->d;
  return (char*)t->self_k + (size_t)l * t->R * d.n_text_ctx * d.n_text_state * t->m->esize;
}
static inline void* self_v_layer(const wh_task* t, int l) {
  const wh_dims& d = t->m->d;
  return (char*)t->self_v + (size_t)l * t->R * d.n_text_ctx * d.n_text_state * t->m->esize;
}
static inline void* cross_layer(const wh_task* t, int l) {
  const wh_dims& d = t->m->d;
  return (char*)t->cross_kv + (size_t)l * t->B * d.n_audio_ctx * 2 * d.n_text_state * t->m->esize;
}

// ---- skinny projections of the prefill -------------------------------------------------------------
// The usual prefill is a handful of rows (batch x 3-4 initial tokens): a 128x128 MFMA tile would be > 80 % padding
// and its launch fills 10-40 workgroups, so up to SKINNY_ROWS rows go through the decode-step projection kernels
// (LayerNorm / residual fused, weights streamed once); long prompts keep the GEMM path.

static hipError_t proj_ln(const wh_model* m, const float* x, int rows, const float* ln_w, const float* ln_b, const void* W,
                          const float* bias, int N, int K, void* y, int64_t y_ld, bool gelu, hipStream_t s) {
  GemvArgs g; memset(&g, 0, sizeof(g));
  g.pro = PRO_LN; g.xf = x; g.xf_ld = K; g.ln_w = ln_w; g.ln_b = ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
  g.W = W; g.bias = bias; g.N = N; g.K = K; g.R = rows;
  g.epi = gelu ? EPI_GELU : EPI_STORE; g.y = y; g.y_ld = y_ld;
  return launch_gemv(g, m->dtype, s);
}
static hipError_t proj_resid(const wh_model* m, const void* x, int64_t x_ld, int rows, const void* W, const float* bias,
                             int N, int K, float* resid, hipStream_t s) {
  GemvArgs g; memset(&g, 0, sizeof(g));
  g.pro = PRO_PLAIN; g.x = x; g.x_ld = x_ld;
  g.W = W; g.bias = bias; g.N = N; g.K = K; g.R = rows;
  g.epi = EPI_RESID; g.resid = resid; g.resid_ld = N;
  return launch_gemv(g, m->dtype, s);
}

// ---- prefill: T0 tokens per row through the GEMM path ------------------------------------------
// leaders (beam search / best-of sampling, G > 1): the G rows of a segment start from the same tokens, so only ONE row
// per segment — row b G of the caller's token matrix — runs through the decoder; it uses cache rows / logits rows
// 0 .. B-1, and the caller replicates them to the rows of each group afterwards (replicate_leader_rows).  The reference
// feeds all n_audio x n_group identical rows (decoding.py:734 repeat_interleave): same values, G times the work.
static int prefill_impl(wh_task* t, const int64_t* tokens, int64_t token_stride, int T0, const int32_t* sel_pos,
                        int n_sel, float* logits_out, int64_t logits_row_ld, hipStream_t s, bool leaders = false) {
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state, H = d.n_text_head, C = d.n_text_ctx, Ta = d.n_audio_ctx, V = d.n_vocab;
  const int R = leaders ? t->B : t->R, M = R * T0;
  const int Gp = leaders ? 1 : t->G;               // rows per audio in this pass
  const int lag_step = leaders ? t->G : 1;         // row r of this pass is row r * lag_step of the task
  if (leaders) token_stride *= t->G;
  const size_t es = m->esize;
  if (!t->audio_set) return WH_ERR_STATE;
  if (T0 <= 0 || T0 > t->Tmax || t->pos + T0 > C) return WH_ERR_ARG;
  if (leaders && (t->pos != 0 || t->qcap)) return WH_ERR_STATE;
  if (t->lag_on) {
    if (t->pos != 0) return WH_ERR_STATE;
    for (int r = 0; r < R; ++r) if (t->h_lag[r * lag_step] >= T0) return WH_ERR_ARG;
  }

  const bool skinny = M <= SKINNY_ROWS && D <= 2048;
  // one row per audio (alignment tasks; beam search through its leader rows) and a transposed V: flash cross attention
  const bool no_flash = WH_DEV_FLAG("WH_NO_PREFILL_FLASH");   // developer A/B switch
  const bool flash_cross = !skinny && !no_flash && m->dtype == WH_F16 && t->cross_vt && Gp == 1 && R == t->B &&
                           t->vt_ld >= (Ta + 63) / 64 * 64;
  HIPCHK(launch_embed(tokens, token_stride, R, T0, m->w.tok_emb, m->w.dec_pos, t->d_pos, nullptr, D, V, t->x, m->dtype, s));
  for (int l = 0; l < d.n_text_layer; ++l) {
    const wh_layer_weights& L = m->dec[l];
    // self attention (causal over cached + new positions)
    if (skinny) {
      HIPCHK(proj_ln(m, t->x, M, L.attn_ln_w, L.attn_ln_b, L.qkv_w, L.qkv_b, 3 * D, D, t->qkv, 3 * D, false, s));
    } else {
      HIPCHK(launch_layernorm(t->x, D, L.attn_ln_w, L.attn_ln_b, t->xn, D, M, D, m->dtype, s));
      HIPCHK(gemm(m, t->xn, D, L.qkv_w, D, t->qkv, 3 * D, M, 3 * D, L.qkv_b, 0, nullptr, 0, false, s));
    }
    HIPCHK(launch_scatter_kv(t->qkv, R, T0, D, t->d_pos, C, self_k_layer(t, l), self_v_layer(t, l), m->dtype, s));
    {
      AttnArgs a; memset(&a, 0, sizeof(a));
      a.q = t->qkv; a.q_ld = 3 * D; a.q_bs = (int64_t)T0 * 3 * D;
      a.k = self_k_layer(t, l); a.k_ld = D; a.k_bs = (int64_t)C * D;
      a.v = self_v_layer(t, l); a.v_ld = D; a.v_bs = (int64_t)C * D;
      a.out = t->att; a.o_ld = D; a.o_bs = (int64_t)T0 * D;
      a.H = H; a.Tq = T0; a.d_len = t->d_pos; a.causal = 1; a.kv_group = 1;
      HIPCHK(launch_attn_generic(a, R, m->dtype, s));
    }
    if (skinny) {
      HIPCHK(proj_resid(m, t->att, D, M, L.out_w, L.out_b, D, D, t->x, s));
      HIPCHK(proj_ln(m, t->x, M, L.cross_ln_w, L.cross_ln_b, L.cq_w, L.cq_b, D, D, t->qkv, D, false, s));
    } else {
      HIPCHK(gemm(m, t->att, D, L.out_w, D, t->x, D, M, D, L.out_b, 0, t->x, D, true, s));
      // cross attention over the cached audio K/V
      HIPCHK(launch_layernorm(t->x, D, L.cross_ln_w, L.cross_ln_b, t->xn, D, M, D, m->dtype, s));
      HIPCHK(gemm(m, t->xn, D, L.cq_w, D, t->qkv, D, M, D, L.cq_b, 0, nullptr, 0, false, s));
    }
    if (t->qcap) {   // keep q rows at their cache positions: [l][r][pos+t][D]
      for (int r = 0; r < R; ++r) {
        char* dst = (char*)t->qcap + (((size_t)l * R + r) * C + t->pos) * D * es;
        HIPCHK(hipMemcpyAsync(dst, (char*)t->qkv + (size_t)r * T0 * D * es, (size_t)T0 * D * es,
                              hipMemcpyDeviceToDevice, s));
      }
    }
    if (skinny) {
      // rows (r, t) are M independent single-query rows of the decode cross attention; the T0 * G rows of one
      // audio share its K/V (one workgroup per (split, head, audio) when T0 * G <= 8)
      const int ps = pick_splits(M, H, Ta, m->dtype);
      DecAttnArgs a; memset(&a, 0, sizeof(a));
      a.q = t->qkv; a.q_ld = D;
      a.k = cross_layer(t, l); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
      a.v = (char*)cross_layer(t, l) + (size_t)D * es; a.v_ld = 2 * D; a.v_bs = a.k_bs;
      a.H = H; a.R = M; a.kv_group = T0 * Gp; a.Tk = Ta; a.splits = ps;
      a.out = t->att; a.o_ld = D; a.part_o = t->part_o; a.part_ml = t->part_ml;
      HIPCHK(launch_attn_decode(a, m->dtype, s));
      GemvArgs g; memset(&g, 0, sizeof(g));
      if (ps > 1) { g.pro = PRO_COMBINE; g.part_o = t->part_o; g.part_ml = t->part_ml; g.splits = ps; g.H = H; }
      else { g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D; }
      g.W = L.cout_w; g.bias = L.cout_b; g.N = D; g.K = D; g.R = M;
      g.epi = EPI_RESID; g.resid = t->x; g.resid_ld = D;
      HIPCHK(launch_gemv(g, m->dtype, s));
    } else if (flash_cross) {
      // T0 queries x 1500 keys per row on the encoder's flash-attention kernel (MFMA; unscaled q / k: it applies
      // d_head^-0.5 itself): the generic kernel took 1.4 ms per layer for 32 rows x 205 tokens (profiles/r03_kernel_stats_extras.csv)
      HIPCHK(launch_attn_flash_f16(t->qkv, D, (int64_t)T0 * D, cross_layer(t, l), 2 * D, (int64_t)Ta * 2 * D,
                                   (char*)t->cross_vt + (size_t)l * t->B * D * t->vt_ld * es, t->vt_ld, (int64_t)D * t->vt_ld,
                                   t->att, D, (int64_t)T0 * D, R, H, Ta, 0, s, T0));
    } else {
      AttnArgs a; memset(&a, 0, sizeof(a));
      a.q = t->qkv; a.q_ld = D; a.q_bs = (int64_t)T0 * D;
      a.k = cross_layer(t, l); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
      a.v = (char*)cross_layer(t, l) + (size_t)D * es; a.v_ld = 2 * D; a.v_bs = a.k_bs;
      a.out = t->att; a.o_ld = D; a.o_bs = (int64_t)T0 * D;
      a.H = H; a.Tq = T0; a.Tk = Ta; a.causal = 0; a.kv_group = Gp;
      HIPCHK(launch_attn_generic(a, R, m->dtype, s));
    }
    if (skinny) {
      HIPCHK(proj_ln(m, t->x, M, L.mlp_ln_w, L.mlp_ln_b, L.fc1_w, L.fc1_b, 4 * D, D, t->h, 4 * D, true, s));
      HIPCHK(proj_resid(m, t->h, 4 * D, M, L.fc2_w, L.fc2_b, D, 4 * D, t->x, s));
    } else {
      HIPCHK(gemm(m, t->att, D, L.cout_w, D, t->x, D, M, D, L.cout_b, 0, t->x, D, true, s));
      // MLP
      HIPCHK(launch_layernorm(t->x, D, L.mlp_ln_w, L.mlp_ln_b, t->xn, D, M, D, m->dtype, s));
      HIPCHK(gemm(m, t->xn, D, L.fc1_w, D, t->h, 4 * D, M, 4 * D, L.fc1_b, 1, nullptr, 0, false, s));
      HIPCHK(gemm(m, t->h, 4 * D, L.fc2_w, 4 * D, t->x, D, M, D, L.fc2_b, 0, t->x, D, true, s));
    }
  }
  // logits of the selected positions
  if (logits_out && n_sel > 0) {
    // the row indices go through pinned memory the task owns, so that the copy needs no host synchronisation behind it (the
    // fused loops' begin calls must not wait for the device); an earlier call's copy has to have executed before the words
    // are rewritten — it has, unless calls follow each other faster than the stream drains
    if (!t->h_sel) HIPCHK(hipHostMalloc((void**)&t->h_sel, (size_t)t->R * t->Tmax * sizeof(int), hipHostMallocDefault));
    if (!t->sel_event) HIPCHK(hipEventCreateWithFlags(&t->sel_event, hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(t->sel_event));
    int* sel = t->h_sel;
    for (int r = 0; r < R; ++r)
      for (int i = 0; i < n_sel; ++i) {
        const int p = (sel_pos ? sel_pos[i] : i) - (sel_pos ? t->h_lag[r * lag_step] : 0);   // selected positions shift with the row
        if (p < 0 || p >= T0) return WH_ERR_ARG;
        sel[(size_t)r * n_sel + i] = r * T0 + p;
      }
    const int Ms = R * n_sel;
    HIPCHK(hipMemcpyAsync(t->d_sel, sel, (size_t)Ms * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(t->sel_event, s));
    HIPCHK(launch_gather_rows(t->x, t->d_sel, Ms, D, t->xsel, s));
    if (Ms <= SKINNY_LOGIT_ROWS && D <= 2048) {    // LayerNorm + tied logits projection as one streaming launch
      GemvArgs g; memset(&g, 0, sizeof(g));
      g.pro = PRO_LN; g.xf = t->xsel; g.xf_ld = D; g.ln_w = m->w.dec_ln_w; g.ln_b = m->w.dec_ln_b;
      g.W = m->w.tok_emb; g.N = V; g.K = D; g.R = Ms;
      g.epi = EPI_F32; g.y = logits_out; g.y_ld = logits_row_ld;
      HIPCHK(launch_gemv(g, m->dtype, s));
    } else {
      HIPCHK(launch_layernorm(t->xsel, D, m->w.dec_ln_w, m->w.dec_ln_b, t->xseln, D, Ms, D, m->dtype, s));
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.A = t->xseln; g.lda = D; g.W = m->w.tok_emb; g.ldw = D;
      g.C = logits_out; g.ldc = logits_row_ld;
      g.M = Ms; g.N = V; g.K = D;
      HIPCHK(launch_gemm(g, m->dtype, 1, 1, s));
    }
  }
  HIPCHK(launch_add_int(t->d_pos, T0, s));
  t->pos += T0;
  return WH_OK;
}

extern "C" int wh_task_prefill(wh_task* t, const int64_t* tokens, int64_t token_stride, int T0,
                               const int32_t* sel_pos, int n_sel, float* logits_out, void* stream) {
  TASK_ENTER(t);
  if (!t || !tokens) return WH_ERR_ARG;
  if (!sel_pos) n_sel = T0;
  return prefill_impl(t, tokens, token_stride, T0, sel_pos, n_sel, logits_out, t->m->d.n_vocab, (hipStream_t)stream);
}

static inline void* cross_layer(const wh_task* t, int l);
// arguments of the fused LN -> cross query -> cross attention launch of layer l (xattn.hip)
static XAttnArgs xattn_args(const wh_task* t, int l, int epoch, float* x_in, bool with_out = false) {
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state, Ta = d.n_audio_ctx;
  const wh_layer_weights& L = m->dec[l];
  XAttnArgs a; memset(&a, 0, sizeof(a));
  a.xf = x_in; a.xf_ld = D; a.W = L.cq_w; a.bias = L.cq_b; a.D = D; a.H = d.n_text_head; a.R = t->R;
  a.k = cross_layer(t, l); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
  a.v = (char*)cross_layer(t, l) + (size_t)D * m->esize; a.v_ld = 2 * D; a.v_bs = a.k_bs;
  a.Tk = Ta; a.splits = t->cross_splits;
  a.out = t->att; a.o_ld = D; a.part_o = t->part_o; a.part_ml = t->part_ml;
  a.qg = t->xq_gran; a.d_tick = t->d_tick; a.epoch = epoch; a.layer = l; a.err = t->d_err; a.mode = fused_mode(0) | ((t->flags & WH_TASK_EXPIRE_HANDOFFS) ? 4 : 0);
  if (with_out) {      // phase 0: x_in += attn.out(t->att) + bias, in place, before the LayerNorm of this launch
    a.att_in = t->att; a.out_w = L.out_w; a.out_b = L.out_b; a.x_io = x_in; a.pflags = t->so_gran;
  }
  return a;
}

static inline void* self_k_layer(const wh_task* t, int l);
static inline void* self_v_layer(const wh_task* t, int l);
// arguments of the fused LN -> QKV -> cache append -> self attention launch of layer l (xattn.hip)
// x_in: the residual stream the layer starts from; x_out (or null): where x_in + attn.out(attention) goes when the output
// projection runs inside the same launch
static SAttnArgs sattn_args(const wh_task* t, int l, int epoch, const float* x_in, float* x_out) {
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state;
  const wh_layer_weights& L = m->dec[l];
  SAttnArgs a; memset(&a, 0, sizeof(a));
  a.xf = x_in; a.xf_ld = D; a.W = L.qkv_w; a.bias = L.qkv_b; a.D = D; a.H = d.n_text_head; a.R = t->R;
  a.out_w = L.out_w; a.out_b = L.out_b; a.x_out = x_out; a.og = t->so_gran;
  a.kcache = self_k_layer(t, l); a.vcache = self_v_layer(t, l); a.cache_bs = (int64_t)d.n_text_ctx * D;
  a.d_pos = t->d_pos; a.lag = t->d_lag; a.q_out = t->qbuf;
  a.out = t->att; a.o_ld = D;
  a.qg = t->sq_gran; a.d_tick = t->d_tick; a.epoch = epoch; a.layer = l; a.err = t->d_err; a.mode = fused_mode(1) | ((t->flags & WH_TASK_EXPIRE_HANDOFFS) ? 4 : 0);
  return a;
}

// ---- one decode step (all kernels read the position from *d_pos: graph-replayable) ---------------
// How the launches of one decode step hand their activations on (decided once per task shape; step_launch and the per-kernel
// bench below read the same plan).
//  * Fragment-order hand-offs (kernels.h): an activation that goes from one launch straight into a PRO_PLAIN projection is written
//    by its producer in the order that projection's lanes read it — only where BOTH ends are launches that know the order (fp16,
//    <= 24 rows: gemv8_kernel).  A/B: WH_NO_FRAGMENT_ORDER=1.
//  * Up to 16 rows the per-row cross attention (no beam groups) with 2 - 4 key splits merges its partials in the launch
//    (attention.hip: the last workgroup of a (row, head) to finish), so cross_attn.out is a plain projection without a merge
//    prologue: 9 / 12 / 16 rows 1753 / 1816 / 1999 -> 1701 / 1770 / 1973 us per step.  At 17 - 24 rows the tickets of 1440
//    workgroups cost more than the merge launch they replace (24 rows 2448 -> 2471): that launch stays.  A/B: WH_NO_TAIL_MERGE=1.
struct StepPlan { bool frag_att, frag_self, frag_mlp, tail_merge; };
static StepPlan step_plan(const wh_task* t) {
  const wh_model* m = t->m;
  const int D = m->d.n_text_state, R = t->R;
  StepPlan p;
  const bool frag_on = m->dtype == WH_F16 && !WH_DEV_FLAG("WH_NO_FRAGMENT_ORDER");
  p.frag_att = frag_on && gemv8_will_run(R, D, D, PRO_PLAIN);                              // attention output -> D x D projection
  p.frag_self = p.frag_att && !t->fused_sattn && !t->fused_xout && t->self_splits <= 1;
  p.frag_mlp = frag_on && (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) && gemv8_will_run(R, 4 * D, D, PRO_LN) &&
               gemv8_will_run(R, D, 4 * D, PRO_PLAIN);                                    // FC1 -> FC2
  p.tail_merge = m->dtype == WH_F16 && !t->fused_xattn && t->G == 1 && R <= 16 && t->cross_splits >= 2 &&
                 t->cross_splits <= 4 && !WH_DEV_FLAG("WH_NO_TAIL_MERGE");
  return p;
}

static int step_launch(wh_task* t, hipStream_t s, bool embedded = false) {
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state, H = d.n_text_head, C = d.n_text_ctx, Ta = d.n_audio_ctx, V = d.n_vocab;
  const int R = t->R;
  const size_t es = m->esize;
  if (!embedded)
    HIPCHK(launch_embed(t->step_tokens, 1, R, 1, m->w.tok_emb, m->w.dec_pos, t->d_pos, t->d_lag, D, V, t->x, m->dtype, s));
  // the residual stream of the step: starts in t->x (embedding / the sampler's x_next); a self-attention launch that also
  // applies attn.out writes the OTHER buffer (its LayerNorm input is still being read), so the pointer alternates
  float* xc = t->x;
  float* xo = t->x2;
  const StepPlan plan = step_plan(t);
  const bool frag_att = plan.frag_att, frag_self = plan.frag_self, frag_mlp = plan.frag_mlp, tail_merge = plan.tail_merge;
  for (int l = 0; l < d.n_text_layer; ++l) {
    const wh_layer_weights& L = m->dec[l];
    GemvArgs g;
    bool out_done = false;
    if (t->fused_sattn) {
      // LN -> QKV -> cache append -> self attention [-> attn.out + residual] as ONE launch (xattn.hip, sattn8_kernel)
      HIPCHK(launch_sattn8(sattn_args(t, l, 0, xc, t->fused_out ? xo : nullptr), s));
      if (t->fused_out) { float* tmp = xc; xc = xo; xo = tmp; out_done = true; }
    } else {
    // LN -> QKV, K/V appended in place at *d_pos
    memset(&g, 0, sizeof(g));
    g.pro = PRO_LN; g.xf = xc; g.xf_ld = D; g.ln_w = L.attn_ln_w; g.ln_b = L.attn_ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
    g.W = L.qkv_w; g.bias = L.qkv_b; g.N = 3 * D; g.K = D; g.R = R;
    g.epi = EPI_QKV; g.y = t->qbuf; g.y_ld = D;
    g.kcache = self_k_layer(t, l); g.vcache = self_v_layer(t, l); g.cache_bs = (int64_t)C * D; g.d_pos = t->d_pos; g.D = D;
    g.lag = t->d_lag;
    HIPCHK(launch_gemv(g, m->dtype, s));
    {
      DecAttnArgs a; memset(&a, 0, sizeof(a));
      a.q = t->qbuf; a.q_ld = D;
      a.k = self_k_layer(t, l); a.k_ld = D; a.k_bs = (int64_t)C * D;
      a.v = self_v_layer(t, l); a.v_ld = D; a.v_bs = (int64_t)C * D;
      a.H = H; a.R = R; a.kv_group = 1; a.d_len = t->d_pos; a.len_plus = 1; a.splits = t->self_splits;
      a.lag = t->d_lag;
      a.out = t->att; a.o_ld = D; a.o_frag = frag_self; a.part_o = t->part_o; a.part_ml = t->part_ml;
      HIPCHK(launch_attn_decode(a, m->dtype, s));
    }
    }
    if (t->fused_xout) out_done = true;       // attn.out + residual run as phase 0 of the cross-attention launch below
    if (!out_done) {
    memset(&g, 0, sizeof(g));
    if (t->self_splits > 1) {
      g.pro = PRO_COMBINE; g.part_o = t->part_o; g.part_ml = t->part_ml; g.splits = t->self_splits; g.H = H;
    } else {
      g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D; g.x_frag = frag_self;
    }
    g.W = L.out_w; g.bias = L.out_b; g.N = D; g.K = D; g.R = R;
    g.epi = EPI_RESID; g.resid = xc; g.resid_ld = D;
    HIPCHK(launch_gemv(g, m->dtype, s));
    }
    if (t->fused_xattn) {
      // LN -> cross query -> cross attention as ONE launch: the K/V stream starts at kernel entry, the projection runs
      // under it and reaches the K/V waves through tagged granules (xattn.hip)
      HIPCHK(launch_xattn8(xattn_args(t, l, 0, xc, t->fused_xout), s));
    } else {
    // LN -> cross query
    memset(&g, 0, sizeof(g));
    g.pro = PRO_LN; g.xf = xc; g.xf_ld = D; g.ln_w = L.cross_ln_w; g.ln_b = L.cross_ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
    g.W = L.cq_w; g.bias = L.cq_b; g.N = D; g.K = D; g.R = R;
    g.epi = EPI_STORE; g.y = t->qbuf; g.y_ld = D;
    HIPCHK(launch_gemv(g, m->dtype, s));
    {
      DecAttnArgs a; memset(&a, 0, sizeof(a));
      a.q = t->qbuf; a.q_ld = D;
      a.k = cross_layer(t, l); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
      a.v = (char*)cross_layer(t, l) + (size_t)D * es; a.v_ld = 2 * D; a.v_bs = a.k_bs;
      a.H = H; a.R = R; a.kv_group = t->G; a.Tk = Ta; a.splits = t->cross_splits;
      a.out = t->att; a.o_ld = D; a.part_o = t->part_o; a.part_ml = t->part_ml;
      if (tail_merge) { a.merge_cnt = t->merge_cnt; a.o_frag = frag_att; }
      if (t->cross_vt) {
        a.vt = (char*)t->cross_vt + (size_t)l * t->B * D * t->vt_ld * es; a.vt_ld = t->vt_ld; a.vt_bs = (int64_t)D * t->vt_ld;
      }
      HIPCHK(launch_attn_decode(a, m->dtype, s));
    }
    }
    memset(&g, 0, sizeof(g));
    // 17+ rows (beam search): the projection runs as 16-row workgroups that would each merge their rows' partials
    // again, so the merge is a launch of its own there (A/B: WH_NO_MERGE_KERNEL=1)
    const bool merge_kernel = !WH_DEV_FLAG("WH_NO_MERGE_KERNEL");   // developer A/B switch
    if (tail_merge) {                 // merged inside the attention launch by the last workgroup of each (row, head)
      g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D; g.x_frag = frag_att;
    } else if (t->cross_splits > 1 && R > 16 && m->dtype == WH_F16 && merge_kernel) {   // the fp32 engine keeps one code path
      HIPCHK(launch_merge_partials(t->part_o, t->part_ml, t->cross_splits, R, H, t->att, D, m->dtype, s, frag_att));
      g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D; g.x_frag = frag_att;
    } else if (t->cross_splits > 1) {
      g.pro = PRO_COMBINE; g.part_o = t->part_o; g.part_ml = t->part_ml; g.splits = t->cross_splits; g.H = H;
    } else {
      g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D;
    }
    g.W = L.cout_w; g.bias = L.cout_b; g.N = D; g.K = D; g.R = R;
    g.epi = EPI_RESID; g.resid = xc; g.resid_ld = D;
    HIPCHK(launch_gemv(g, m->dtype, s));
    // LN -> MLP
    memset(&g, 0, sizeof(g));
    g.pro = PRO_LN; g.xf = xc; g.xf_ld = D; g.ln_w = L.mlp_ln_w; g.ln_b = L.mlp_ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
    g.W = L.fc1_w; g.bias = L.fc1_b; g.N = 4 * D; g.K = D; g.R = R;
    g.epi = EPI_GELU; g.y = t->h; g.y_ld = 4 * D; g.y_frag = frag_mlp;
    HIPCHK(launch_gemv(g, m->dtype, s));
    memset(&g, 0, sizeof(g));
    g.pro = PRO_PLAIN; g.x = t->h; g.x_ld = 4 * D; g.x_frag = frag_mlp;
    g.W = L.fc2_w; g.bias = L.fc2_b; g.N = D; g.K = 4 * D; g.R = R;
    g.epi = EPI_RESID; g.resid = xc; g.resid_ld = D;
    HIPCHK(launch_gemv(g, m->dtype, s));
  }
  {
    GemvArgs g; memset(&g, 0, sizeof(g));
    g.pro = PRO_LN; g.xf = xc; g.xf_ld = D; g.ln_w = m->w.dec_ln_w; g.ln_b = m->w.dec_ln_b;
    g.W = m->w.tok_emb; g.bias = nullptr; g.N = V; g.K = D; g.R = R;
    g.epi = EPI_F32; g.y = t->logits; g.y_ld = V;
    g.bump = t->d_pos; g.bump_by = 1;         // the last kernel of the step advances the position counter
    g.bump2 = t->d_tick;                      // ... and the step tick the granule tags of xattn.hip are made of
    HIPCHK(launch_gemv(g, m->dtype, s));
  }
  return WH_OK;
}

static bool graphs_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("WH_NO_GRAPH"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}

// runs one step from t->step_tokens into t->logits
static int step_run(wh_task* t, hipStream_t s, bool embedded = false) {
  if (t->pos <= 0) return WH_ERR_STATE;
  if (t->pos + 1 > t->m->d.n_text_ctx) return WH_ERR_ARG;
  int rc;
  const int gi = embedded ? 1 : 0;
  if (t->graph_exec[gi]) {
    HIPCHK(hipGraphLaunch(t->graph_exec[gi], s));
  } else if (s == nullptr || !graphs_enabled() || t->steps_eager[gi] < 1) {
    rc = step_launch(t, s, embedded);
    if (rc != WH_OK) return rc;
    t->steps_eager[gi]++;
  } else {
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rc = step_launch(t, s, embedded);
    hipError_t e = hipStreamEndCapture(s, &t->graph[gi]);
    if (rc != WH_OK) return rc;
    HIPCHK(e);
    HIPCHK(hipGraphInstantiate(&t->graph_exec[gi], t->graph[gi], nullptr, nullptr, 0));
    HIPCHK(hipGraphLaunch(t->graph_exec[gi], s));
  }
  t->pos += 1;
  return WH_OK;
}

extern "C" int wh_task_step(wh_task* t, const int64_t* last_tokens, int64_t token_stride, float* logits_out,
                            void* stream_) {
  TASK_ENTER(t);
  if (!t || !last_tokens || !logits_out) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  HIPCHK(launch_gather_tokens(last_tokens, token_stride, t->R, t->step_tokens, s));
  int rc = step_run(t, s);
  if (rc != WH_OK) return rc;
  if (logits_out != t->logits)
    HIPCHK(hipMemcpyAsync(logits_out, t->logits, (size_t)t->R * t->m->d.n_vocab * 4, hipMemcpyDeviceToDevice, s));
  return WH_OK;
}

extern "C" int wh_task_rearrange(wh_task* t, const int32_t* source_indices, void* stream_) {
  TASK_ENTER(t);
  if (!t || !source_indices) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  const wh_dims& d = t->m->d;
  bool identity = true;
  for (int i = 0; i < t->R; ++i) {
    if (source_indices[i] < 0 || source_indices[i] >= t->R) return WH_ERR_ARG;
    if (source_indices[i] != i) identity = false;
  }
  if (identity || t->pos == 0) return WH_OK;     // decoding.py:173
  if (!t->spare_k) return WH_ERR_STATE;
  HIPCHK(hipMemcpyAsync(t->d_src, source_indices, (size_t)t->R * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  const size_t es = t->m->esize;
  const int64_t row_bytes = (int64_t)d.n_text_ctx * d.n_text_state * es;
  const int64_t used_bytes = (int64_t)t->pos * d.n_text_state * es;
  bool in_group = t->G <= 8;                       // beam search never takes a source from another audio
  for (int i = 0; i < t->R && in_group; ++i) in_group = source_indices[i] / t->G == i / t->G;
  if (in_group) {
    HIPCHK(launch_permute_groups(t->self_k, t->self_v, d.n_text_layer, (int64_t)t->R * row_bytes, t->B, t->G, row_bytes,
                                 used_bytes, t->d_src, nullptr, 0, s));
    return WH_OK;
  }
  for (int l = 0; l < d.n_text_layer; ++l) {
    void* caches[2] = {self_k_layer(t, l), self_v_layer(t, l)};
    for (int c = 0; c < 2; ++c) {
      HIPCHK(launch_gather_cache(caches[c], t->spare_k, t->d_src, t->R, row_bytes, used_bytes, s));
      HIPCHK(hipMemcpy2DAsync(caches[c], row_bytes, t->spare_k, row_bytes, used_bytes, t->R,
                              hipMemcpyDeviceToDevice, s));
    }
  }
  return WH_OK;
}

// ---- fused greedy / beam loops as resumable state machines ---------------------------------------------------------
// wh_task_greedy_begin / wh_task_beam_begin queue the prompt pass and the first sampling decision and return;
// wh_task_poll queues more decode steps — never more than ~10 ahead of the device, the look-ahead the blocking loops have
// always had — and returns WH_RUNNING until the loop has ended, without ever waiting for the device.  wh_task_greedy /
// wh_task_beam are begin + poll-until-done with blocking waits in place of the event queries: one code path, the same
// sequence of device operations either way.
//
// A bounded hand-off spin of the fused step launches that ran out during a loop (never observed on an unshared device; a
// GPU time-sliced between processes could stretch a spin past its bound) makes what the loop produced invalid.  The task
// then leaves the fused kernels for good — its step graphs are dropped and rebuilt from the two-launch kernels, which wait
// for nothing — and the loop is re-run from the prompt, which is still in place: the prefill starts at position 0 again,
// the cross K/V and the rows' lags are untouched.  The caller only ever sees the second result.
static int handoff_fallback(wh_task* t, hipStream_t s) {
  t->fused_xattn = t->fused_sattn = t->fused_out = t->fused_xout = false;
  for (int i = 0; i < 2; ++i) {
    if (t->graph_exec[i]) { (void)hipGraphExecDestroy(t->graph_exec[i]); t->graph_exec[i] = nullptr; }
    if (t->graph[i]) { (void)hipGraphDestroy(t->graph[i]); t->graph[i] = nullptr; }
    t->steps_eager[i] = 0;
  }
  HIPCHK(hipMemsetAsync(t->d_pos, 0, 4, s));
  t->pos = 0;
  t->handoff_fallbacks++;
  return WH_OK;
}

// pinned words behind the B per-segment completion flags of h_poll
enum { HP_ALIVE = 0, HP_ERR = 8, HP_ERR0 = 9, HP_ALIVE_INIT = 10, HP_APPLIED = 11, HP_WORDS = 32 };

static int loop_resources(wh_task* t) {
  if (!t->h_poll) HIPCHK(hipHostMalloc((void**)&t->h_poll, ((size_t)t->B + HP_WORDS) * 4, hipHostMallocDefault));
  // waits on these events SLEEP (hipEventBlockingSync) instead of spinning: a host thread that drives a loop costs no core
  if (!t->poll_event) HIPCHK(hipEventCreateWithFlags(&t->poll_event, hipEventDisableTiming | hipEventBlockingSync));
  if (!t->done_event) HIPCHK(hipEventCreateWithFlags(&t->done_event, hipEventDisableTiming | hipEventBlockingSync));
  if (!t->loop) { t->loop = new (std::nothrow) wh_loop(); if (!t->loop) return WH_ERR_ARG; memset((void*)t->loop, 0, sizeof(wh_loop)); }
  return WH_OK;
}

// wait for (block) or ask about (!block) an event: WH_OK = reached, WH_RUNNING = not yet
static int event_reached(hipEvent_t e, bool block) {
  if (block) { HIPCHK(hipEventSynchronize(e)); return WH_OK; }
  const hipError_t q = hipEventQuery(e);
  if (q == hipSuccess) return WH_OK;
  if (q == hipErrorNotReady) { (void)hipGetLastError(); return WH_RUNNING; }
  g_last_hip = q;
  return WH_ERR_HIP;
}

static int greedy_start(wh_task* t) {
  wh_loop* L = t->loop;
  hipStream_t s = L->s;
  const wh_greedy_params* p = &L->bp.rules;
  const wh_dims& d = t->m->d;
  const int V = d.n_vocab, R = t->R, T0 = p->sample_begin;
  int* hp = t->h_poll + t->B;
  // hand-off time-outs are judged per call: the counter as it stands when this call's work starts (stream-ordered copy)
  // is the baseline, so time-outs of earlier host-driven wh_task_step calls on a cached task do not trigger a fallback here
  HIPCHK(hipMemcpyAsync(hp + HP_ERR0, t->d_err, 4, hipMemcpyDeviceToHost, s));

  int32_t sel[2]; int n_sel;
  const bool want_ns = L->no_speech_token >= 0 && L->no_speech_probs != nullptr;
  if (want_ns && L->sot_index != T0 - 1) { sel[0] = L->sot_index; sel[1] = T0 - 1; n_sel = 2; }
  else { sel[0] = T0 - 1; n_sel = 1; }
  int rc = prefill_impl(t, L->tokens, L->token_stride, T0, sel, n_sel, t->logits, V, s);
  if (rc != WH_OK) return rc;
  if (want_ns) HIPCHK(launch_no_speech(t->logits, (int64_t)n_sel * V, R, V, L->no_speech_token, L->no_speech_probs, s));
  HIPCHK(hipMemsetAsync(L->sum_logprobs, 0, (size_t)R * 4, s));
  hp[HP_ALIVE_INIT] = T0 - 1;                          // pinned: read when the copy executes; rewritten only by a later loop
  HIPCHK(hipMemcpyAsync(t->d_alive, hp + HP_ALIVE_INIT, 4, hipMemcpyHostToDevice, s));

  SampleArgs& sa = L->sa; memset(&sa, 0, sizeof(sa));
  sa.R = R; sa.V = V; sa.tokens = L->tokens; sa.token_stride = L->token_stride; sa.d_ntok = t->d_pos; sa.lag = t->d_lag;
  sa.sample_begin = T0; sa.eot = p->eot; sa.timestamp_begin = p->timestamp_begin; sa.no_timestamps = p->no_timestamps;
  sa.max_initial_ts = p->max_initial_timestamp_index; sa.suppress_blank = p->suppress_blank;
  sa.blank_token = p->blank_token; sa.suppress_mask = p->suppress_mask; sa.sum_logprobs = L->sum_logprobs;
  sa.step_tokens = t->step_tokens; sa.d_alive_step = t->d_alive; sa.partials = t->samp_part;
  sa.row_state = t->samp_state;
  HIPCHK(hipMemsetAsync(t->samp_state, 0, (size_t)R * 16, s));
  // the sampler also writes the next step's input row (token embedding + position): the step graph starts at layer 0
  L->fused_embed = !WH_DEV_FLAG("WH_NO_FUSED_EMBED");   // developer A/B switch
  if (L->fused_embed) {
    sa.x_next = t->x; sa.tok_emb = t->m->w.tok_emb; sa.pos_emb = t->m->w.dec_pos; sa.D = d.n_text_state;
    sa.emb_f16 = t->m->dtype == WH_F16 ? 1 : 0; sa.n_pos = d.n_text_ctx;
  }
  if (p->temperature > 0.f) {
    sa.inv_temperature = 1.0f / p->temperature;
    sa.seed_lo = (uint32_t)(p->seed & 0xffffffffu); sa.seed_hi = (uint32_t)(p->seed >> 32);
  }
  sa.logits = t->logits + (size_t)(n_sel - 1) * V; sa.logits_ld = (int64_t)n_sel * V;
  HIPCHK(launch_greedy_sample(sa, s));
  sa.logits = t->logits; sa.logits_ld = V;
  L->ntok = T0 + 1; L->steps = 1;
  L->pending = L->wait_due = false; L->ntok_at_copy = 0;
  L->phase = PH_STEPPING;
  return WH_OK;
}

// after a `leaders` prefill: cache rows 0 .. B-1 hold the T0 prompt positions of the B segments and logits rows 0 .. B-1
// their selected logits; row b goes to rows b G .. b G + G - 1.  Descending b: the destination rows of segment b only
// overlap source rows of segments above b, which have been replicated by then.
static int replicate_leader_rows(wh_task* t, int T0, float* logits, int n_sel, hipStream_t s) {
  const wh_dims& d = t->m->d;
  const size_t es = t->m->esize;
  const int64_t row_bytes = (int64_t)d.n_text_ctx * d.n_text_state * es;
  const int64_t layer_bytes = (int64_t)t->R * row_bytes;
  for (int b = t->B - 1; b >= 0; --b) {
    HIPCHK(launch_replicate_row(t->self_k, layer_bytes, d.n_text_layer, row_bytes, b, b * t->G, t->G,
                                (int64_t)T0 * d.n_text_state * es, s));
    HIPCHK(launch_replicate_row(t->self_v, layer_bytes, d.n_text_layer, row_bytes, b, b * t->G, t->G,
                                (int64_t)T0 * d.n_text_state * es, s));
    if (logits)
      HIPCHK(launch_replicate_row(logits, 0, 1, (int64_t)n_sel * d.n_vocab * 4, b, b * t->G, t->G,
                                  (int64_t)n_sel * d.n_vocab * 4, s));
  }
  return WH_OK;
}

// one BeamSearchDecoder.update on the device (3 launches) + rearrange_kv_cache (decoding.py:172-176): new beam i continues
// the cache of row src[i]
static int beam_update(wh_task* t, const float* logits, int64_t logits_ld, int first) {
  wh_loop* L = t->loop;
  const wh_dims& d = t->m->d;
  const size_t es = t->m->esize;
  const int R = t->R, B = t->B, G = t->G;
  const int64_t row_bytes = (int64_t)d.n_text_ctx * d.n_text_state * es;
  int64_t* buf[2] = {L->tokens, L->tokens + (int64_t)R * L->token_stride};
  int* done[2] = {t->beam_flags, t->beam_flags + B};
  BeamArgs& a = L->ba;
  a.logits = logits; a.logits_ld = logits_ld; a.first = first;
  a.tokens_in = buf[L->cur]; a.tokens_out = buf[L->cur ^ 1];
  a.done_prev = done[L->cur]; a.done_next = done[L->cur ^ 1];
  HIPCHK(launch_beam_step(a, B, L->s));
  L->cur ^= 1;
  HIPCHK(launch_permute_groups(t->self_k, t->self_v, d.n_text_layer, (int64_t)R * row_bytes, B, G, row_bytes,
                               (int64_t)t->pos * d.n_text_state * es, t->d_src, a.copy_from,
                               (int64_t)d.n_text_state * es, L->s));
  return WH_OK;
}

static int beam_start(wh_task* t) {
  wh_loop* L = t->loop;
  hipStream_t s = L->s;
  const wh_greedy_params* p = &L->bp.rules;
  const wh_dims& d = t->m->d;
  const int V = d.n_vocab, R = t->R, B = t->B, G = t->G, T0 = p->sample_begin;
  int* hp = t->h_poll + B;
  HIPCHK(hipMemcpyAsync(hp + HP_ERR0, t->d_err, 4, hipMemcpyDeviceToHost, s));      // time-out counter at call entry (see greedy_start)

  int32_t sel[2]; int n_sel;
  const bool want_ns = L->no_speech_token >= 0 && L->no_speech_probs != nullptr;
  if (want_ns && L->sot_index != T0 - 1) { sel[0] = L->sot_index; sel[1] = T0 - 1; n_sel = 2; }
  else { sel[0] = T0 - 1; n_sel = 1; }
  // the G beams of a segment hold the same prompt: one row per segment through the decoder, then replicate (A/B:
  // WH_BEAM_FULL_PREFILL=1 feeds all R rows as the reference does)
  const bool leaders = !WH_DEV_FLAG("WH_BEAM_FULL_PREFILL");
  int rc = prefill_impl(t, L->tokens, L->token_stride, T0, sel, n_sel, t->logits, V, s, leaders);
  if (rc != WH_OK) return rc;
  if (leaders) { rc = replicate_leader_rows(t, T0, t->logits, n_sel, s); if (rc != WH_OK) return rc; }
  if (want_ns) HIPCHK(launch_no_speech(t->logits, (int64_t)n_sel * V, R, V, L->no_speech_token, L->no_speech_probs, s));
  HIPCHK(hipMemsetAsync(L->sum_logprobs, 0, (size_t)R * 4, s));
  HIPCHK(hipMemsetAsync(L->fin_count, 0, (size_t)B * 4, s));
  HIPCHK(hipMemsetAsync(t->beam_flags, 0, (2 * (size_t)B + 1) * 4, s));

  BeamArgs& a = L->ba; memset(&a, 0, sizeof(a));
  a.R = R; a.V = V; a.G = G; a.K = G + 1; a.token_stride = L->token_stride; a.d_ntok = t->d_pos; a.lag = t->d_lag;
  a.sample_begin = T0; a.eot = p->eot; a.timestamp_begin = p->timestamp_begin; a.no_timestamps = p->no_timestamps;
  a.max_initial_ts = p->max_initial_timestamp_index; a.suppress_blank = p->suppress_blank;
  a.blank_token = p->blank_token; a.suppress_mask = p->suppress_mask; a.sum_logprobs = L->sum_logprobs;
  beam_scratch_carve(a, t->beam_scratch, R, V);
  a.fin_tok = L->fin_tokens; a.fin_len = L->fin_len; a.fin_score = L->fin_scores; a.fin_count = L->fin_count;
  a.max_candidates = L->bp.max_candidates;
  a.src = t->d_src; a.step_tokens = t->step_tokens; a.d_applied = t->beam_flags + 2 * B;
  // shared-history bookkeeping for the cache permutation (beams that descend from one ancestor hold the same K/V up to
  // the point where they split: those positions are never copied).  "Everything so far" to start with: all beams of a
  // segment hold the same prompt.  Not used with ragged prompts (positions are row-local there).
  const bool lcp_on = !WH_DEV_FLAG("WH_BEAM_FULL_PERMUTE");   // developer A/B switch
  if (lcp_on && !t->lag_on) {
    a.lcp = t->beam_lcp; a.copy_from = t->beam_lcp + (size_t)B * 64;
    HIPCHK(hipMemsetAsync(t->beam_lcp, 0x7f, (size_t)B * 64 * 4, s));
  }
  L->cur = 0;
  rc = beam_update(t, t->logits + (size_t)(n_sel - 1) * V, (int64_t)n_sel * V, 1);
  if (rc != WH_OK) return rc;
  L->ntok = T0 + 1; L->steps = 1;
  L->pending = L->wait_due = false;
  L->phase = PH_STEPPING;
  return WH_OK;
}

// Queue what can be queued and report: WH_OK = the loop has ended and its results are in place (L->n_tokens), WH_RUNNING
// = call again (never returned with block), anything else = error (the loop is abandoned).
// Completion is polled every 8 tokens WITHOUT draining the queue: the counter (greedy: index of the last live token; beam:
// the per-segment completion flags) is copied to pinned memory behind step k, an event is recorded, two more steps are
// queued, and only then is the event waited for / asked about — the device is two steps behind the host at that point and
// never idles for a launch.  At most two steps run past completion (they change nothing: finished rows keep emitting EOT
// without accumulating, a beam update whose segments are all done leaves the state untouched).
static int loop_pump(wh_task* t, bool block) {
  wh_loop* L = t->loop;
  hipStream_t s = L->s;
  const wh_greedy_params* p = &L->bp.rules;
  const wh_dims& d = t->m->d;
  const int B = t->B;
  int* hp = t->h_poll + B;
  for (;;) {
    if (L->phase == PH_STEPPING) {
      bool finished = false;
      for (;;) {
        if (L->pending && L->wait_due) {
          const int rc = event_reached(t->poll_event, block);
          if (rc != WH_OK) return rc;
          L->pending = L->wait_due = false;
          if (t->loop_kind == LOOP_GREEDY) {
            if (t->h_poll[0] < L->ntok_at_copy - 1) { finished = true; break; }
          } else {
            bool fin = true;
            for (int b = 0; b < B; ++b) fin = fin && t->h_poll[b] != 0;
            if (fin) { finished = true; break; }
          }
        }
        if (!(L->steps < p->max_steps && L->ntok <= p->n_ctx && L->ntok <= d.n_text_ctx)) break;
        int rc = step_run(t, s, t->loop_kind == LOOP_GREEDY && L->fused_embed);
        if (rc != WH_OK) return rc;
        if (t->loop_kind == LOOP_GREEDY) HIPCHK(launch_greedy_sample(L->sa, s));
        else { rc = beam_update(t, t->logits, d.n_vocab, 0); if (rc != WH_OK) return rc; }
        ++L->ntok; ++L->steps;
        if (L->pending && (L->steps & 7) == 2) L->wait_due = true;
        if ((L->steps & 7) == 0) {
          if (t->loop_kind == LOOP_GREEDY) HIPCHK(hipMemcpyAsync(t->h_poll, t->d_alive, 4, hipMemcpyDeviceToHost, s));
          else HIPCHK(hipMemcpyAsync(t->h_poll, (L->cur ? t->beam_flags + B : t->beam_flags), (size_t)B * 4, hipMemcpyDeviceToHost, s));
          HIPCHK(hipEventRecord(t->poll_event, s));
          L->pending = true; L->wait_due = false; L->ntok_at_copy = L->ntok;
        }
      }
      (void)finished;
      // the loop's results, behind everything queued so far
      if (t->loop_kind == LOOP_GREEDY) {
        HIPCHK(hipMemcpyAsync(hp + HP_ALIVE, t->d_alive, 4, hipMemcpyDeviceToHost, s));
      } else {
        HIPCHK(hipMemcpyAsync(hp + HP_APPLIED, t->beam_flags + 2 * B, 4, hipMemcpyDeviceToHost, s));
        if (L->cur == 1)
          HIPCHK(hipMemcpyAsync(L->tokens, L->tokens + (int64_t)t->R * L->token_stride, (size_t)t->R * L->token_stride * 8,
                                hipMemcpyDeviceToDevice, s));
      }
      HIPCHK(hipMemcpyAsync(hp + HP_ERR, t->d_err, 4, hipMemcpyDeviceToHost, s));       // fused launches: hand-off time-outs
      HIPCHK(hipEventRecord(t->done_event, s));
      L->phase = PH_DRAINING;
    }
    {
      const int rc = event_reached(t->done_event, block);
      if (rc != WH_OK) return rc;
    }
    if ((t->fused_xattn || t->fused_sattn) && hp[HP_ERR] != hp[HP_ERR0]) {
      // a hand-off spin ran out: once more from the prompt, on the two-launch kernels
      int rc = handoff_fallback(t, s);
      if (rc == WH_OK) rc = t->loop_kind == LOOP_GREEDY ? greedy_start(t) : beam_start(t);
      if (rc != WH_OK) return rc;
      continue;
    }
    if (t->loop_kind == LOOP_GREEDY) {
      // the sampler that appended token index c ran with ntok == c; "completed" first holds at c = alive + 1
      int final_len = hp[HP_ALIVE] + 2;
      if (final_len > L->ntok) final_len = L->ntok;
      L->n_tokens = final_len;
    } else {
      L->n_tokens = p->sample_begin + hp[HP_APPLIED];
    }
    return WH_OK;
  }
}

static int greedy_check(const wh_task* t, const wh_greedy_params* p, const int64_t* tokens, int64_t token_stride,
                        const float* sum_logprobs) {
  if (!t || !p || !tokens || !sum_logprobs) return WH_ERR_ARG;
  const wh_dims& d = t->m->d;
  const int T0 = p->sample_begin;
  if (t->pos != 0 || T0 <= 0 || T0 > t->Tmax || p->max_steps <= 0) return WH_ERR_ARG;
  if (token_stride < (int64_t)T0 + p->max_steps) return WH_ERR_ARG;
  // ragged rows share one step counter: no row may reach the context limit before the step budget runs out
  if (t->lag_on && (T0 + p->max_steps > p->n_ctx || T0 + p->max_steps > d.n_text_ctx)) return WH_ERR_ARG;
  return WH_OK;
}

static int beam_check(const wh_task* t, const wh_beam_params* bp, const int64_t* tokens, int64_t token_stride,
                      const float* sum_logprobs, const int64_t* fin_tokens, const int32_t* fin_len, const float* fin_scores,
                      const int32_t* fin_count) {
  if (!t || !bp || !tokens || !sum_logprobs || !fin_tokens || !fin_len || !fin_scores || !fin_count) return WH_ERR_ARG;
  const wh_greedy_params* p = &bp->rules;
  const wh_dims& d = t->m->d;
  const int R = t->R, G = t->G, T0 = p->sample_begin;
  if (G < 2 || G > 8 || bp->beam_size != G || bp->max_candidates < 1 || !t->beam_scratch) return WH_ERR_ARG;
  if (t->pos != 0 || T0 <= 0 || T0 > t->Tmax || p->max_steps <= 0) return WH_ERR_ARG;
  if (token_stride < (int64_t)T0 + p->max_steps + 1) return WH_ERR_ARG;
  // ragged prompts: rows share one step counter (no row may reach the context limit before the budget ends), and the
  // beams of one segment share its prompt
  if (t->lag_on) {
    if (T0 + p->max_steps > p->n_ctx || T0 + p->max_steps > d.n_text_ctx) return WH_ERR_ARG;
    for (int r = 0; r < R; ++r) if (t->h_lag[r] != t->h_lag[r / G * G]) return WH_ERR_ARG;
  }
  return WH_OK;
}

static int loop_begin(wh_task* t, int kind, const wh_beam_params* bp, int64_t* tokens, int64_t token_stride, int sot_index,
                      int no_speech_token, float* sum_logprobs, float* no_speech_probs, int64_t* fin_tokens,
                      int32_t* fin_len, float* fin_scores, int32_t* fin_count, void* stream_) {
  int rc = loop_resources(t);
  if (rc != WH_OK) return rc;
  wh_loop* L = t->loop;
  memset((void*)L, 0, sizeof(*L));
  L->s = (hipStream_t)stream_;
  L->bp = *bp;
  L->tokens = tokens; L->token_stride = token_stride; L->sot_index = sot_index; L->no_speech_token = no_speech_token;
  L->sum_logprobs = sum_logprobs; L->no_speech_probs = no_speech_probs;
  L->fin_tokens = fin_tokens; L->fin_len = fin_len; L->fin_scores = fin_scores; L->fin_count = fin_count;
  t->loop_kind = kind;
  rc = kind == LOOP_GREEDY ? greedy_start(t) : beam_start(t);
  if (rc != WH_OK) t->loop_kind = LOOP_IDLE;
  return rc;
}

extern "C" int wh_task_greedy_begin(wh_task* t, const wh_greedy_params* p, int64_t* tokens, int64_t token_stride,
                                    int sot_index, int no_speech_token, float* sum_logprobs, float* no_speech_probs,
                                    void* stream_) {
  TASK_ENTER(t);
  int rc = greedy_check(t, p, tokens, token_stride, sum_logprobs);
  if (rc != WH_OK) return rc;
  wh_beam_params bp; memset(&bp, 0, sizeof(bp)); bp.rules = *p;
  return loop_begin(t, LOOP_GREEDY, &bp, tokens, token_stride, sot_index, no_speech_token, sum_logprobs, no_speech_probs,
                    nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int wh_task_beam_begin(wh_task* t, const wh_beam_params* bp, int64_t* tokens, int64_t token_stride, int sot_index,
                                  int no_speech_token, float* sum_logprobs, float* no_speech_probs, int64_t* fin_tokens,
                                  int32_t* fin_len, float* fin_scores, int32_t* fin_count, void* stream_) {
  TASK_ENTER(t);
  int rc = beam_check(t, bp, tokens, token_stride, sum_logprobs, fin_tokens, fin_len, fin_scores, fin_count);
  if (rc != WH_OK) return rc;
  return loop_begin(t, LOOP_BEAM, bp, tokens, token_stride, sot_index, no_speech_token, sum_logprobs, no_speech_probs,
                    fin_tokens, fin_len, fin_scores, fin_count, stream_);
}

extern "C" int wh_task_poll(wh_task* t, int32_t* n_tokens_out) {
  if (!t) return WH_ERR_ARG;
  TaskGuard guard(t, /*loop_call=*/true);
  if (!guard.ok) return WH_ERR_STATE;
  if (!t->loop || t->loop_kind == LOOP_IDLE) return WH_ERR_STATE;
  const int rc = loop_pump(t, false);
  if (rc == WH_RUNNING) return rc;
  if (rc == WH_OK && n_tokens_out) *n_tokens_out = t->loop->n_tokens;
  t->loop_kind = LOOP_IDLE;                       // ended (or failed): the task takes other calls again
  return rc;
}

extern "C" int wh_task_greedy(wh_task* t, const wh_greedy_params* p, int64_t* tokens, int64_t token_stride,
                              int sot_index, int no_speech_token, float* sum_logprobs, float* no_speech_probs,
                              int32_t* n_tokens_out, void* stream_) {
  TASK_ENTER(t);
  if (!n_tokens_out) return WH_ERR_ARG;
  int rc = greedy_check(t, p, tokens, token_stride, sum_logprobs);
  if (rc != WH_OK) return rc;
  wh_beam_params bp; memset(&bp, 0, sizeof(bp)); bp.rules = *p;
  rc = loop_begin(t, LOOP_GREEDY, &bp, tokens, token_stride, sot_index, no_speech_token, sum_logprobs, no_speech_probs,
                  nullptr, nullptr, nullptr, nullptr, stream_);
  if (rc == WH_OK) rc = loop_pump(t, true);
  if (rc == WH_OK) *n_tokens_out = t->loop->n_tokens;
  if (t->loop) t->loop_kind = LOOP_IDLE;
  return rc;
}

extern "C" int wh_task_beam(wh_task* t, const wh_beam_params* bp, int64_t* tokens, int64_t token_stride, int sot_index,
                            int no_speech_token, float* sum_logprobs, float* no_speech_probs, int64_t* fin_tokens,
                            int32_t* fin_len, float* fin_scores, int32_t* fin_count, int32_t* n_tokens_out,
                            void* stream_) {
  TASK_ENTER(t);
  if (!n_tokens_out) return WH_ERR_ARG;
  int rc = beam_check(t, bp, tokens, token_stride, sum_logprobs, fin_tokens, fin_len, fin_scores, fin_count);
  if (rc != WH_OK) return rc;
  rc = loop_begin(t, LOOP_BEAM, bp, tokens, token_stride, sot_index, no_speech_token, sum_logprobs, no_speech_probs,
                  fin_tokens, fin_len, fin_scores, fin_count, stream_);
  if (rc == WH_OK) rc = loop_pump(t, true);
  if (rc == WH_OK) *n_tokens_out = t->loop->n_tokens;
  if (t->loop) t->loop_kind = LOOP_IDLE;
  return rc;
}

// ---- measurement hook ----------------------------------------------------------------------------------
static int bench_issue(wh_task* t, int kind, int iters, double* bytes_per_launch, void* stream_) {
  if (!t || iters <= 0) return WH_ERR_ARG;
  if (!t->audio_set || t->pos <= 0) return WH_ERR_STATE;
  hipStream_t s = (hipStream_t)stream_;
  const wh_model* m = t->m;
  const wh_dims& d = m->d;
  const int D = d.n_text_state, H = d.n_text_head, C = d.n_text_ctx, Ta = d.n_audio_ctx, V = d.n_vocab;
  const int R = t->R, Ln = d.n_text_layer;
  const double es = m->esize;
  const StepPlan plan = step_plan(t);            // the kernels are timed in the form the step launches them in
  double bytes = 0;
  for (int i = 0; i < iters; ++i) {
    const int l = i % Ln;
    const wh_layer_weights& L = m->dec[l];
    GemvArgs g; memset(&g, 0, sizeof(g));
    switch (kind) {
      case 0: {
        int rc = step_launch(t, s);
        if (rc != WH_OK) return rc;
        HIPCHK(launch_add_int(t->d_pos, -1, s));   // stay at the same position
        bytes = es * ((double)Ln * 14.0 * D * D + (double)V * D) + (double)t->B * Ln * 2.0 * Ta * D * es +
                (double)R * (t->pos + 1) * Ln * 2.0 * D * es + (double)R * V * 4.0;
      } break;
      case 1: if (t->fused_xattn) {
        // the fused launch: the tag of launch i is unique within the graph (epoch = i); one add behind the chain keeps
        // replays apart
        // (with attn.out + residual as its phase 0 the launch also streams that D x D matrix and rewrites t->x in place:
        // the values drift over the chain, the timing does not depend on them)
        HIPCHK(launch_xattn8(xattn_args(t, l, i, t->x, t->fused_xout), s));
        if (i == iters - 1) HIPCHK(launch_add_int(t->d_tick, iters, s));
        bytes = (double)t->B * 2.0 * Ta * D * es + (t->fused_xout ? 2.0 : 1.0) * D * D * es;
      } else {
        DecAttnArgs a; memset(&a, 0, sizeof(a));
        a.q = t->qbuf; a.q_ld = D;
        a.k = cross_layer(t, l); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
        a.v = (char*)cross_layer(t, l) + (size_t)D * m->esize; a.v_ld = 2 * D; a.v_bs = a.k_bs;
        a.H = H; a.R = R; a.kv_group = t->G; a.Tk = Ta; a.splits = t->cross_splits;
        a.out = t->att; a.o_ld = D; a.part_o = t->part_o; a.part_ml = t->part_ml;
        if (plan.tail_merge) { a.merge_cnt = t->merge_cnt; a.o_frag = plan.frag_att; }
        HIPCHK(launch_attn_decode(a, m->dtype, s));
        bytes = (double)t->B * 2.0 * Ta * D * es;
      } break;
      case 2: if (t->fused_sattn) {
        HIPCHK(launch_sattn8(sattn_args(t, l, i, t->x, t->fused_out ? t->x2 : nullptr), s));
        if (i == iters - 1) HIPCHK(launch_add_int(t->d_tick, iters, s));
        bytes = (double)R * t->pos * 2.0 * D * es + (t->fused_out ? 4.0 : 3.0) * D * D * es;
      } else {
        DecAttnArgs a; memset(&a, 0, sizeof(a));
        a.q = t->qbuf; a.q_ld = D;
        a.k = self_k_layer(t, l); a.k_ld = D; a.k_bs = (int64_t)C * D;
        a.v = self_v_layer(t, l); a.v_ld = D; a.v_bs = (int64_t)C * D;
        a.H = H; a.R = R; a.kv_group = 1; a.d_len = t->d_pos; a.len_plus = 0; a.splits = t->self_splits;
        a.out = t->att; a.o_ld = D; a.o_frag = plan.frag_self; a.part_o = t->part_o; a.part_ml = t->part_ml;
        HIPCHK(launch_attn_decode(a, m->dtype, s));
        bytes = (double)R * t->pos * 2.0 * D * es;
      } break;
      case 3:
        g.pro = PRO_LN; g.xf = t->x; g.xf_ld = D; g.ln_w = L.attn_ln_w; g.ln_b = L.attn_ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
        g.W = L.qkv_w; g.bias = L.qkv_b; g.N = 3 * D; g.K = D; g.R = R;
        g.epi = EPI_STORE; g.y = t->qkv; g.y_ld = 3 * D;
        HIPCHK(launch_gemv(g, m->dtype, s));
        bytes = 3.0 * D * D * es;
        break;
      case 4:
        g.pro = PRO_LN; g.xf = t->x; g.xf_ld = D; g.ln_w = L.mlp_ln_w; g.ln_b = L.mlp_ln_b; g.ln_folded = (m->w.flags & WH_WEIGHTS_DEC_LN_FOLDED) ? 1 : 0;
        g.W = L.fc1_w; g.bias = L.fc1_b; g.N = 4 * D; g.K = D; g.R = R;
        g.epi = EPI_GELU; g.y = t->h; g.y_ld = 4 * D; g.y_frag = plan.frag_mlp;
        HIPCHK(launch_gemv(g, m->dtype, s));
        bytes = 4.0 * D * D * es;
        break;
      case 5:
        g.pro = PRO_PLAIN; g.x = t->h; g.x_ld = 4 * D; g.x_frag = plan.frag_mlp;
        g.W = L.fc2_w; g.bias = L.fc2_b; g.N = D; g.K = 4 * D; g.R = R;
        g.epi = EPI_STORE; g.y = t->att; g.y_ld = D;
        HIPCHK(launch_gemv(g, m->dtype, s));
        bytes = 4.0 * D * D * es;
        break;
      case 6:
        g.pro = PRO_LN; g.xf = t->x; g.xf_ld = D; g.ln_w = m->w.dec_ln_w; g.ln_b = m->w.dec_ln_b;
        g.W = m->w.tok_emb; g.N = V; g.K = D; g.R = R;
        g.epi = EPI_F32; g.y = t->logits; g.y_ld = V;
        HIPCHK(launch_gemv(g, m->dtype, s));
        bytes = (double)V * D * es + (double)R * V * 4.0;
        break;
      case 7:
        g.pro = PRO_PLAIN; g.x = t->att; g.x_ld = D; g.x_frag = plan.frag_self;
        g.W = L.out_w; g.bias = L.out_b; g.N = D; g.K = D; g.R = R;
        g.epi = EPI_STORE; g.y = t->qbuf; g.y_ld = D;
        HIPCHK(launch_gemv(g, m->dtype, s));
        bytes = 1.0 * D * D * es;
        break;
      default: return WH_ERR_ARG;
    }
  }
  if (bytes_per_launch) *bytes_per_launch = bytes;
  return WH_OK;
}

// Average duration of one launch of a decode-step kernel, measured with HIP events on the launch stream: `iters`
// launches (rotating over the layers, so every launch is HBM-cold) are captured into a hipGraph and replayed — the
// same dependent-launch boundaries the kernels see inside the decode step (an eager loop of short kernels measures the
// host's launch rate instead).  kind 0 = the whole step (257 launches + the counter reset), captured the same way.
extern "C" int wh_task_bench_kernel(wh_task* t, int kind, int iters, double* bytes_per_launch, float* ms_per_launch,
                                    void* stream_) {
  TASK_ENTER(t);
  if (!t || iters <= 0 || !ms_per_launch) return WH_ERR_ARG;
  if (!t->audio_set || t->pos <= 0) return WH_ERR_STATE;
  hipStream_t s = (hipStream_t)stream_;
  if (s == nullptr) return WH_ERR_ARG;            // stream capture needs a real stream
  int rc = bench_issue(t, kind, 2, bytes_per_launch, stream_);      // warm-up: function attributes, code objects
  if (rc != WH_OK) return rc;
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  rc = bench_issue(t, kind, iters, bytes_per_launch, stream_);
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != WH_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  HIPCHK(e);
  HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipGraphLaunch(exec, s));                // first replay untimed
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    HIPCHK(hipEventRecord(e0, s));
    HIPCHK(hipGraphLaunch(exec, s));
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
  *ms_per_launch = best / (float)iters;
  return WH_OK;
}

// ---- word timestamps ---------------------------------------------------------------------------------
extern "C" int wh_task_cross_qk(wh_task* t, int row, const int32_t* layers, const int32_t* heads, int n_pairs,
                                int tok_begin, int n_tok, float* out, void* stream_) {
  TASK_ENTER(t);
  if (!t || !layers || !heads || !out || n_pairs <= 0) return WH_ERR_ARG;
  if (!t->qcap) return WH_ERR_STATE;
  const wh_dims& d = t->m->d;
  if (row < 0 || row >= t->R || tok_begin < 0 || n_tok <= 0 || tok_begin + n_tok > t->pos) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  const int D = d.n_text_state, C = d.n_text_ctx, Ta = d.n_audio_ctx;
  const size_t es = t->m->esize;
  for (int i = 0; i < n_pairs; ++i) {
    const int l = layers[i], h = heads[i];
    if (l < 0 || l >= d.n_text_layer || h < 0 || h >= d.n_text_head) return WH_ERR_ARG;
    const char* q = (const char*)t->qcap + ((((size_t)l * t->R + row) * C) + tok_begin) * D * es;
    const char* k = (const char*)cross_layer(t, l) + (size_t)(row / t->G) * Ta * 2 * D * es;
    HIPCHK(launch_cross_qk(q, D, k, 2 * D, h, n_tok, Ta, out + (size_t)i * n_tok * Ta, t->m->dtype, s));
  }
  return WH_OK;
}

// ---- batched find_alignment core --------------------------------------------------------------------------------
static size_t align_batch_carve(int R, int P, int Tmax, int Ta, int Fmax, void* base, float** qk, float** w, int** ints) {
  Carver c(base);
  *qk = (float*)c.take((size_t)R * P * Tmax * Ta * 4);
  *w = (float*)c.take((size_t)2 * R * P * Tmax * Fmax * 4);
  *ints = (int*)c.take((size_t)(2 * R + 2 * P) * 4);
  return align_up(c.off, 256);
}

extern "C" size_t wh_align_batch_scratch_bytes(int n_rows, int n_pairs, int max_tok, int n_audio_ctx, int max_frames) {
  if (n_rows <= 0 || n_pairs <= 0 || max_tok <= 0 || n_audio_ctx <= 0 || max_frames <= 0) return 0;
  float *a, *b; int* i;
  return align_batch_carve(n_rows, n_pairs, max_tok, n_audio_ctx, max_frames, nullptr, &a, &b, &i);
}

extern "C" int wh_task_align_batch(wh_task* t, const int32_t* layers, const int32_t* heads, int n_pairs,
                                   const int32_t* n_tok, const int32_t* n_frames, int width, int row_begin, float qk_scale,
                                   float* cost_out, int8_t* trace_out, int64_t trace_stride, void* scratch,
                                   size_t scratch_bytes, void* stream_) {
  TASK_ENTER(t);
  if (!t || !layers || !heads || !n_tok || !n_frames || !cost_out || !trace_out || !scratch || n_pairs <= 0) return WH_ERR_ARG;
  if (!t->qcap) return WH_ERR_STATE;
  if (width <= 0 || (width & 1) == 0 || width > 63 || row_begin < 0) return WH_ERR_ARG;
  hipStream_t s = (hipStream_t)stream_;
  const wh_dims& d = t->m->d;
  const int R = t->R, D = d.n_text_state, C = d.n_text_ctx, Ta = d.n_audio_ctx;
  int Tmax = 0, Fmax = 0;
  for (int r = 0; r < R; ++r) {
    if (n_tok[r] <= row_begin + 1 || n_tok[r] > t->pos || n_frames[r] <= 0 || n_frames[r] > Ta) return WH_ERR_ARG;
    if (n_tok[r] > Tmax) Tmax = n_tok[r];
    if (n_frames[r] > Fmax) Fmax = n_frames[r];
  }
  for (int i = 0; i < n_pairs; ++i)
    if (layers[i] < 0 || layers[i] >= d.n_text_layer || heads[i] < 0 || heads[i] >= d.n_text_head) return WH_ERR_ARG;
  const int Nmax = Tmax - 1 - row_begin;
  if (Nmax > 8192) return WH_ERR_LIMIT;
  if (trace_stride < (int64_t)(Nmax + 1) * (Fmax + 1)) return WH_ERR_ARG;
  float *qk, *w; int* ints;
  if (align_batch_carve(R, n_pairs, Tmax, Ta, Fmax, scratch, &qk, &w, &ints) > scratch_bytes) return WH_ERR_WORKSPACE;
  std::vector<int> h((size_t)2 * R + 2 * n_pairs);
  for (int r = 0; r < R; ++r) { h[r] = n_tok[r]; h[R + r] = n_frames[r]; }
  for (int i = 0; i < n_pairs; ++i) { h[2 * R + i] = layers[i]; h[2 * R + n_pairs + i] = heads[i]; }
  HIPCHK(hipMemcpyAsync(ints, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));       // `h` is host stack memory
  const int *d_ntok = ints, *d_nfr = ints + R, *d_layers = ints + 2 * R, *d_heads = ints + 2 * R + n_pairs;
  const size_t es = t->m->esize;
  (void)es;
  HIPCHK(launch_cross_qk_batch(t->qcap, (int64_t)R * C * D, (int64_t)C * D, D, t->cross_kv, (int64_t)t->B * Ta * 2 * D,
                               (int64_t)Ta * 2 * D, t->G, d_layers, d_heads, n_pairs, d_ntok, R, Tmax, Ta, qk,
                               t->m->dtype, s));
  HIPCHK(launch_align_batch(qk, d_ntok, d_nfr, R, n_pairs, Tmax, Ta, Fmax, width, row_begin, 1, qk_scale, cost_out, Nmax, w, s));
  HIPCHK(launch_dtw_batch(cost_out, d_ntok, d_nfr, R, Tmax, Fmax, row_begin, 1, Nmax, trace_out, trace_stride, s));
  return WH_OK;
}

extern "C" int wh_median_filter(const float* x, float* out, int64_t rows, int n, int width, void* stream) {
  if (!x || !out || rows < 0 || n <= 0) return WH_ERR_ARG;
  if (width <= 0 || (width & 1) == 0 || width > 63) return WH_ERR_ARG;
  HIPCHK(launch_median_filter(x, out, rows, n, width, (hipStream_t)stream));
  return WH_OK;
}

extern "C" int wh_dtw_trace(const float* x, int N, int M, int8_t* trace_out, void* stream) {
  if (!x || !trace_out || N <= 0 || M <= 0) return WH_ERR_ARG;
  if (N > 8192) return WH_ERR_LIMIT;
  HIPCHK(launch_dtw(x, N, M, trace_out, (hipStream_t)stream));
  return WH_OK;
}

extern "C" int wh_dtw_backtrace_batch(const int8_t* trace, int64_t trace_stride, const int32_t* d_n_rows,
                                      const int32_t* d_n_cols, int n_clips, int max_rows, int max_cols, int32_t* jumps_out,
                                      int64_t jump_stride, int32_t* path_out, int64_t path_stride, int32_t* path_len_out,
                                      void* stream) {
  if (!trace || !d_n_rows || !d_n_cols || n_clips <= 0 || max_rows <= 0 || max_cols <= 0) return WH_ERR_ARG;
  if (!jumps_out && !path_out) return WH_ERR_ARG;
  if (jumps_out && jump_stride < max_rows) return WH_ERR_ARG;
  if (path_out && (path_stride < (int64_t)max_rows + max_cols || !path_len_out)) return WH_ERR_ARG;
  if (trace_stride < (int64_t)(max_rows + 1) * (max_cols + 1)) return WH_ERR_ARG;
  HIPCHK(launch_dtw_backtrace_batch(trace, trace_stride, d_n_rows, d_n_cols, n_clips, max_rows, max_cols, jumps_out,
                                    jump_stride, path_out, path_stride, path_len_out, (hipStream_t)stream));
  return WH_OK;
}

extern "C" int wh_align_matrix(const float* qk, int n_heads, int n_tok, int n_audio_ctx, int n_frames, int width,
                               int row_begin, int row_end, float qk_scale, float* out, void* scratch, void* stream) {
  if (!qk || !out || !scratch || n_heads <= 0 || n_tok <= 0 || n_frames <= 0 || n_frames > n_audio_ctx) return WH_ERR_ARG;
  if (row_begin < 0 || row_end > n_tok || row_end <= row_begin) return WH_ERR_ARG;
  if (width <= 0 || (width & 1) == 0 || width > 63) return WH_ERR_ARG;
  HIPCHK(launch_align_matrix(qk, n_heads, n_tok, n_audio_ctx, n_frames, width, row_begin, row_end, qk_scale, out,
                             (float*)scratch, (hipStream_t)stream));
  return WH_OK;
}
