// sampling.hip — the logits epilogue of DecodingTask._main_loop (whisper/decoding.py:696-703) for
// greedy decoding, as two small kernels per step (vocabulary-parallel partials, per-row decision) with
// no host synchronisation:
//   SuppressBlank (decoding.py:423-430) -> SuppressTokens (:433-438) -> ApplyTimestampRules (:441-505)
//   -> GreedyDecoder.update at temperature 0 (:277-293): argmax, log_softmax of the *filtered* logits,
//   sum_logprobs accumulation for rows not yet at EOT, EOT stickiness.
// A single pass over the fp32 logits keeps separate online (max, sum-exp, arg-max) statistics for the text range [0, timestamp_begin) and the timestamp range, from which the
// "timestamp probability mass > best text token" rule (:498-505) and the final normaliser follow
// without re-reading the row.  The row's token history (needed by the pairing / monotonicity rules)
// is scanned in parallel.
#include "common.h"
#include "kernels.h"

namespace {

struct Stat {           // online softmax statistics + first-index argmax of a range
  float m, s; int idx;
};
__device__ __forceinline__ void stat_add(Stat& a, float x, int i) {
  if (x == WH_NEG_INF) return;
  if (x > a.m) { a.s = a.s * expf(a.m - x) + 1.0f; a.m = x; a.idx = i; }
  else { a.s += expf(x - a.m); }   // x == a.m keeps the earlier (smaller) index: thread scan is ascending
}
__device__ __forceinline__ void stat_merge(Stat& a, float m, float s, int idx) {
  if (m == WH_NEG_INF) return;
  if (a.m == WH_NEG_INF) { a.m = m; a.s = s; a.idx = idx; return; }
  if (m > a.m || (m == a.m && idx < a.idx)) {
    a.s = a.s * expf(a.m - m) + s;
    a.m = m; a.idx = idx;
  } else {
    a.s += s * expf(m - a.m);
  }
}
__device__ __forceinline__ void stat_wave_reduce(Stat& a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m = __shfl_xor(a.m, o, 64);
    const float s = __shfl_xor(a.s, o, 64);
    const int idx = __shfl_xor(a.idx, o, 64);
    stat_merge(a, m, s, idx);
  }
}

constexpr int SCHUNK = 1024;    // vocabulary entries per stage-1 workgroup (256 threads x 4)

// Temperature sampling (GreedyDecoder.update at temperature > 0, decoding.py:281-283: Categorical(logits / T).sample())
// as a Gumbel-max draw: argmax over the filtered entries of x / T + g, g = -log(-log(u)), u from a counter-based hash of
// (seed, step, row, token) — one uniform per entry, no state, the same partial / final kernel structure as the arg-max.
// The log-probability that is accumulated is the one of the UNSCALED filtered logits (decoding.py:285-287).
struct Pick {           // best perturbed key of a range: key, the entry's logit, its index
  float key, x; int idx;
};
__device__ __forceinline__ void pick_merge(Pick& a, float key, float x, int idx) {
  if (idx == 0x7fffffff) return;
  if (a.idx == 0x7fffffff || key > a.key || (key == a.key && idx < a.idx)) { a.key = key; a.x = x; a.idx = idx; }
}
__device__ __forceinline__ void pick_wave_reduce(Pick& a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float key = __shfl_xor(a.key, o, 64);
    const float x = __shfl_xor(a.x, o, 64);
    const int idx = __shfl_xor(a.idx, o, 64);
    pick_merge(a, key, x, idx);
  }
}
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// standard Gumbel noise for (seed, step, row, token); 24-bit uniform strictly inside (0, 1)
__device__ __forceinline__ float gumbel(uint32_t seed_lo, uint32_t seed_hi, int step, int row, int v) {
  uint32_t h = fmix32((uint32_t)row * 0x9E3779B1u + (uint32_t)step) ^ seed_lo;
  h = fmix32(h ^ (uint32_t)v * 0x85EBCA77u);
  h = fmix32(h + seed_hi);
  const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return -logf(-logf(u));
}
constexpr int PSTRIDE = 8;      // floats per (row, chunk, range) partial: m, s, idx | key, x (sampling)

// Stage 1: grid (chunks, rows).  Applies the filters to one 1024-entry slice of the row and reduces it
// to two partial statistics (text range, timestamp range): {max, sum exp(x - max), first arg-max}.
// All four logits of a thread are requested before any is used (one L2 round trip per workgroup).
template <bool SAMPLE>
__global__ __launch_bounds__(256) void greedy_partial_kernel(whk::SampleArgs a, int nchunk) {
  pin_kernargs(a);
  asm volatile("" ::"s"(nchunk));
  __shared__ int sh_last_ts;
  __shared__ float sh_m[2][4], sh_s[2][4];
  __shared__ int sh_i[2][4];
  __shared__ float sh_k[2][4], sh_x[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, k = blockIdx.y;
  const float* x = a.logits + (int64_t)k * a.logits_ld;
  float xv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = c * SCHUNK + j * 256 + tid;
    xv[j] = v < a.V ? x[v] : WH_NEG_INF;
  }
  const int vl = load_agent_int(a.lag ? a.lag + k : a.d_ntok), vn = load_agent_int(a.d_ntok);
  const int lag = a.lag ? uniform(vl) : 0;   // ragged prompts: this row's indices sit lag earlier than the longest row's
  const int ntok = uniform(vn) - lag;
  const int sample_begin = a.sample_begin - lag;
  const int64_t* row = a.tokens + (int64_t)k * a.token_stride;
  const int L = ntok - sample_begin;
  const int TB = a.timestamp_begin;       // < 0: timestamp rules disabled (without_timestamps)
  const bool ts_rules = TB >= 0;

  bool last_ts = false, pen_ts = false;
  int ts_lo = 0, ts_hi = 0;              // forbidden timestamp interval [ts_lo, ts_hi)
  if (ts_rules && a.row_state) {
    // the final kernel of the previous step left what the rules need about this row's sampled tokens
    const int s0 = load_agent_int(a.row_state + 4 * k), s1 = load_agent_int(a.row_state + 4 * k + 1),
              s2 = load_agent_int(a.row_state + 4 * k + 2);
    last_ts = (L >= 1) && uniform(s0) != 0;
    pen_ts = (L < 2) || uniform(s1) != 0;
    const int tv = uniform(s2);
    if (tv > 0) { ts_lo = TB; ts_hi = (last_ts && !pen_ts) ? tv - 1 : tv; }
  } else if (ts_rules) {
    if (tid == 0) sh_last_ts = -1;
    __syncthreads();
    for (int t = tid; t < L; t += 256)
      if (row[sample_begin + t] >= TB) atomicMax(&sh_last_ts, t);
    __syncthreads();
    last_ts = (L >= 1) && (row[ntok - 1] >= TB);
    pen_ts = (L < 2) || (row[ntok - 2] >= TB);
    if (sh_last_ts >= 0) {
      const int t = (int)row[sample_begin + sh_last_ts];
      ts_lo = TB;
      ts_hi = (last_ts && !pen_ts) ? t : t + 1;
    }
  }
  const int split = ts_rules ? TB : a.V;   // text range [0, split), timestamp range [split, V)

  Stat st[2];
  st[0] = Stat{WH_NEG_INF, 0.f, 0x7fffffff};
  st[1] = Stat{WH_NEG_INF, 0.f, 0x7fffffff};
  Pick pk[2];
  pk[0] = Pick{WH_NEG_INF, WH_NEG_INF, 0x7fffffff};
  pk[1] = pk[0];
  const int step = ntok + lag;             // the common position counter: the same for every row of a step
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = c * SCHUNK + j * 256 + tid;
    if (v >= a.V) continue;
    bool masked = a.suppress_mask && a.suppress_mask[v];
    if (a.suppress_blank && L == 0 && (v == a.blank_token || v == a.eot)) masked = true;
    if (ts_rules) {
      if (v == a.no_timestamps) masked = true;
      if (last_ts) {
        if (pen_ts) { if (v >= TB) masked = true; }
        else { if (v < a.eot) masked = true; }
      }
      if (v >= ts_lo && v < ts_hi) masked = true;
      if (L == 0) {
        if (v < TB) masked = true;
        if (a.max_initial_ts >= 0 && v > TB + a.max_initial_ts) masked = true;
      }
    }
    if (masked) continue;
    if (v < split) stat_add(st[0], xv[j], v); else stat_add(st[1], xv[j], v);
    if constexpr (SAMPLE) {
      if (xv[j] != WH_NEG_INF) {
        const float key = xv[j] * a.inv_temperature + gumbel(a.seed_lo, a.seed_hi, step, k, v);
        pick_merge(pk[v < split ? 0 : 1], key, xv[j], v);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    stat_wave_reduce(st[g]);
    if constexpr (SAMPLE) pick_wave_reduce(pk[g]);
    if (lane == 0) {
      sh_m[g][wave] = st[g].m; sh_s[g][wave] = st[g].s; sh_i[g][wave] = SAMPLE ? pk[g].idx : st[g].idx;
      if constexpr (SAMPLE) { sh_k[g][wave] = pk[g].key; sh_x[g][wave] = pk[g].x; }
    }
  }
  __syncthreads();
  if (tid < 2) {
    Stat t = Stat{WH_NEG_INF, 0.f, 0x7fffffff};
    float* o = a.partials + (((int64_t)k * nchunk + c) * 2 + tid) * PSTRIDE;   // nchunk == gridDim.x, passed as an argument:
                                                                               // gridDim sits in the implicit arguments (a second kernarg round trip)
    if constexpr (SAMPLE) {
      Pick p = Pick{WH_NEG_INF, WH_NEG_INF, 0x7fffffff};
      for (int w = 0; w < 4; ++w) {
        stat_merge(t, sh_m[tid][w], sh_s[tid][w], 0);
        pick_merge(p, sh_k[tid][w], sh_x[tid][w], sh_i[tid][w]);
      }
      o[0] = t.m; o[1] = t.s; o[2] = __int_as_float(p.idx); o[3] = p.key; o[4] = p.x;
    } else {
      for (int w = 0; w < 4; ++w) stat_merge(t, sh_m[tid][w], sh_s[tid][w], sh_i[tid][w]);
      o[0] = t.m; o[1] = t.s; o[2] = __int_as_float(t.idx);
    }
  }
}

// Stage 2: one workgroup per row merges the chunk partials, applies the "timestamp mass" rule and
// GreedyDecoder.update, and appends the token.
template <bool SAMPLE>
__global__ __launch_bounds__(256) void greedy_final_kernel(whk::SampleArgs a, int nchunk) {
  pin_kernargs(a);
  asm volatile("" ::"s"(nchunk));
  __shared__ float sh_m[2][4], sh_s[2][4];
  __shared__ int sh_i[2][4];
  __shared__ float sh_k[2][4], sh_x[2][4];
  __shared__ int sh_next;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x;
  const int vl = load_agent_int(a.lag ? a.lag + k : a.d_ntok), vn = load_agent_int(a.d_ntok);
  const int lag = a.lag ? uniform(vl) : 0;
  const int ntok = uniform(vn) - lag;
  int64_t* row = a.tokens + (int64_t)k * a.token_stride;
  const bool ts_rules = a.timestamp_begin >= 0;
  Stat st[2];
  st[0] = Stat{WH_NEG_INF, 0.f, 0x7fffffff};
  st[1] = Stat{WH_NEG_INF, 0.f, 0x7fffffff};
  Pick pk[2];
  pk[0] = Pick{WH_NEG_INF, WH_NEG_INF, 0x7fffffff};
  pk[1] = pk[0];
  for (int c = tid; c < nchunk; c += 256) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* o = a.partials + (((int64_t)k * nchunk + c) * 2 + g) * PSTRIDE;
      if constexpr (SAMPLE) {
        stat_merge(st[g], o[0], o[1], 0);
        pick_merge(pk[g], o[3], o[4], __float_as_int(o[2]));
      } else {
        stat_merge(st[g], o[0], o[1], __float_as_int(o[2]));
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    stat_wave_reduce(st[g]);
    if constexpr (SAMPLE) pick_wave_reduce(pk[g]);
    if (lane == 0) {
      sh_m[g][wave] = st[g].m; sh_s[g][wave] = st[g].s; sh_i[g][wave] = SAMPLE ? pk[g].idx : st[g].idx;
      if constexpr (SAMPLE) { sh_k[g][wave] = pk[g].key; sh_x[g][wave] = pk[g].x; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int64_t last_tok = row[ntok - 1];
    Stat tx = Stat{WH_NEG_INF, 0.f, 0x7fffffff}, ts = tx;
    Pick ptx = Pick{WH_NEG_INF, WH_NEG_INF, 0x7fffffff}, pts = ptx;
    for (int w = 0; w < 4; ++w) {
      stat_merge(tx, sh_m[0][w], sh_s[0][w], SAMPLE ? 0 : sh_i[0][w]);
      stat_merge(ts, sh_m[1][w], sh_s[1][w], SAMPLE ? 0 : sh_i[1][w]);
      if constexpr (SAMPLE) {
        pick_merge(ptx, sh_k[0][w], sh_x[0][w], sh_i[0][w]);
        pick_merge(pts, sh_k[1][w], sh_x[1][w], sh_i[1][w]);
      }
    }
    bool text_masked = false;
    if (ts_rules) {
      // decoding.py:498-505 — logsumexp of timestamp logprobs vs max text logprob (same normaliser)
      Stat all = tx;
      stat_merge(all, ts.m, ts.s, ts.idx);
      if (all.m != WH_NEG_INF) {
        const float lse_all = logf(all.s);
        const float ts_lp_max = (ts.m - all.m) - lse_all;
        const float ts_lse = (ts.m == WH_NEG_INF) ? WH_NEG_INF : ts_lp_max + logf(ts.s);
        const float text_lp_max = (tx.m == WH_NEG_INF) ? WH_NEG_INF : (tx.m - all.m) - lse_all;
        if (ts_lse > text_lp_max) text_masked = true;
      }
    }
    Stat fin = ts;
    if (!text_masked) { fin = tx; stat_merge(fin, ts.m, ts.s, ts.idx); }
    int next = fin.idx;
    // log_softmax(filtered)[next] = x - max - log(sum exp(x - max)); x[next] == max
    float lp = -logf(fin.s);
    if constexpr (SAMPLE) {                  // the Gumbel-max draw among the allowed entries; unscaled log-probability
      Pick p = pts;
      if (!text_masked) pick_merge(p, ptx.key, ptx.x, ptx.idx);
      next = p.idx;
      lp = (p.x - fin.m) - logf(fin.s);
    }
    if (fin.m == WH_NEG_INF) { next = 0; lp = __builtin_nanf(""); }   // every logit filtered: argmax of all -inf
    if (last_tok != a.eot) a.sum_logprobs[k] += lp;
    else next = a.eot;
    row[ntok] = next;
    if (a.step_tokens) a.step_tokens[k] = next;
    if (next != a.eot) *a.d_alive_step = ntok + lag;      // benign race: every writer stores the same value
    sh_next = next;
    if (a.row_state && ts_rules) {                         // what the next step's timestamp rules need about this row
      int* st = a.row_state + 4 * k;
      const bool is_ts = next >= a.timestamp_begin;
      st[1] = st[0];
      st[0] = is_ts ? 1 : 0;
      if (is_ts) st[2] = next + 1;
    }
  }
  if (a.x_next) {                             // the next step's input row: embedding of the token just chosen, at its index
    __syncthreads();
    int tok = sh_next;
    if (tok < 0) tok = 0;
    if (tok > a.V - 1) tok = a.V - 1;
    if (ntok < a.n_pos) {
      const float* pp = a.pos_emb + (int64_t)ntok * a.D;
      float* xr = a.x_next + (int64_t)k * a.D;
      if (a.emb_f16) {
        const half_t* e = (const half_t*)a.tok_emb + (int64_t)tok * a.D;
        for (int dd = tid; dd < a.D; dd += 256) xr[dd] = (float)e[dd] + pp[dd];
      } else {
        const float* e = (const float*)a.tok_emb + (int64_t)tok * a.D;
        for (int dd = tid; dd < a.D; dd += 256) xr[dd] = e[dd] + pp[dd];
      }
    }
  }
}

// softmax(logits at <|startoftranscript|>)[no_speech] per row (decoding.py:689-693).  One workgroup of 1024 threads per
// row; every thread requests 8 logits before it folds any of them (the first version walked the row with one dependent
// L2 round trip per 256 entries: 72 us for 51866 entries; this form takes a handful of round trips).
__global__ __launch_bounds__(1024) void no_speech_kernel(const float* __restrict__ logits, int64_t row_stride,
                                                         int V, int no_speech, float* __restrict__ out) {
  __shared__ float sh_m[16], sh_s[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)blockIdx.x * row_stride;
  float m = WH_NEG_INF, s = 0.f;
  constexpr int UN = 8;
  for (int v0 = 0; v0 < V; v0 += 1024 * UN) {
    float xv[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      int v = v0 + j * 1024 + tid; if (v > V - 1) v = V - 1;       // branch-free loads, masked at use
      xv[j] = x[v];
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      if (v0 + j * 1024 + tid < V) {
        if (xv[j] > m) { s = s * expf(m - xv[j]) + 1.0f; m = xv[j]; } else s += expf(xv[j] - m);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (mn == WH_NEG_INF) ? 0.f : s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
  }
  if (lane == 0) { sh_m[wave] = m; sh_s[wave] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = sh_m[0];
    for (int w = 1; w < 16; ++w) M = fmaxf(M, sh_m[w]);
    float S = 0.f;
    for (int w = 0; w < 16; ++w) S += (sh_m[w] == WH_NEG_INF) ? 0.f : sh_s[w] * expf(sh_m[w] - M);
    out[blockIdx.x] = expf(x[no_speech] - M) / S;
  }
}

__global__ void gather_tokens_kernel(const int64_t* __restrict__ src, int64_t stride, int R,
                                     int64_t* __restrict__ dst) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) dst[r] = src[(int64_t)r * stride];
}

}  // namespace

namespace whk {

size_t greedy_sample_scratch_bytes(int R, int V) { return (size_t)R * ((V + SCHUNK - 1) / SCHUNK) * 2 * PSTRIDE * sizeof(float); }

hipError_t launch_greedy_sample(const SampleArgs& a, hipStream_t stream) {
  const int nchunk = (a.V + SCHUNK - 1) / SCHUNK;
  if (!a.partials) return hipErrorInvalidValue;
  if (a.inv_temperature > 0.f) {
    hipLaunchKernelGGL(greedy_partial_kernel<true>, dim3(nchunk, a.R), dim3(256), 0, stream, a, nchunk);
    hipLaunchKernelGGL(greedy_final_kernel<true>, dim3(a.R), dim3(256), 0, stream, a, nchunk);
  } else {
    hipLaunchKernelGGL(greedy_partial_kernel<false>, dim3(nchunk, a.R), dim3(256), 0, stream, a, nchunk);
    hipLaunchKernelGGL(greedy_final_kernel<false>, dim3(a.R), dim3(256), 0, stream, a, nchunk);
  }
  return hipGetLastError();
}

hipError_t launch_no_speech(const float* logits, int64_t row_stride, int R, int V, int no_speech,
                            float* out, hipStream_t stream) {
  hipLaunchKernelGGL(no_speech_kernel, dim3(R), dim3(1024), 0, stream, logits, row_stride, V, no_speech, out);
  return hipGetLastError();
}

hipError_t launch_gather_tokens(const int64_t* src, int64_t stride, int R, int64_t* dst, hipStream_t stream) {
  hipLaunchKernelGGL(gather_tokens_kernel, dim3((R + 63) / 64), dim3(64), 0, stream, src, stride, R, dst);
  return hipGetLastError();
}

}  // namespace whk
