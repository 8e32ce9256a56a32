// beam.hip — the logits epilogue and candidate bookkeeping of beam search, on the device:
//   DecodingTask._main_loop (whisper/decoding.py:696-703) with BeamSearchDecoder.update (:323-382) and the
//   SuppressBlank / SuppressTokens / ApplyTimestampRules filters (:423-505).
// Three kernels per step, no host synchronisation:
//   beam_partial_kernel  grid (vocabulary chunks, rows): filters one 1024-entry slice of a row and reduces it to online
//                        softmax statistics and its K = beam + 1 best entries, separately for the text range
//                        [0, timestamp_begin) and the timestamp range (the "timestamp mass" rule :498-505 decides
//                        later whether text is allowed at all);
//   beam_row_kernel      grid (rows): merges the chunk partials -> log_softmax normaliser of the filtered row and its
//                        top K (log-probability, token) pairs == F.log_softmax(logits).topk(beam + 1) (:333-345);
//   beam_update_kernel   grid (audio segments): the candidate bookkeeping of one segment — scores = sum_logprobs + logprob
//                        in fp32, stable descending order, the first `beam` sequences not ending in EOT survive (their
//                        token rows are gathered from the source beam into the other token buffer), the EOT ones met
//                        on the way are appended to the segment's finished list while it holds fewer than
//                        max_candidates (:347-382); writes the source row of every new beam for the KV-cache
//                        permutation (rearrange_kv_cache, :172-176) and the next step's input tokens.
// The reference keys candidates by their token tuple in a dict ("later duplicates overwrite").  Two live beams of one
// segment never hold the same prefix except at the first update, where all of them hold the initial tokens and produce
// the same candidates: there only the last beam's candidates count (the dict keeps the first beam's order and the last
// beam's values and source).  Once every segment has max_candidates finished sequences the reference stops; the
// device cannot stop a queue of launches, so later updates see the completion flags of the previous launch and leave
// everything untouched until the host polls them.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BCHUNK = 1024;     // vocabulary entries per partial workgroup (256 threads x 4)
constexpr int KMAX = whk::BEAM_KMAX;
constexpr int NO_IDX = 0x7fffffff;

struct Stat {           // online softmax statistics of a range
  float m, s;
};
__device__ __forceinline__ void stat_add(Stat& a, float x) {
  if (x == WH_NEG_INF) return;
  if (x > a.m) { a.s = a.s * expf(a.m - x) + 1.0f; a.m = x; }
  else { a.s += expf(x - a.m); }
}
__device__ __forceinline__ void stat_merge(Stat& a, float m, float s) {
  if (m == WH_NEG_INF) return;
  if (a.m == WH_NEG_INF) { a.m = m; a.s = s; return; }
  if (m > a.m) { a.s = a.s * expf(a.m - m) + s; a.m = m; }
  else { a.s += s * expf(m - a.m); }
}
__device__ __forceinline__ void stat_wave_reduce(Stat& a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m = __shfl_xor(a.m, o, 64);
    const float s = __shfl_xor(a.s, o, 64);
    stat_merge(a, m, s);
  }
}

// larger value wins; equal values: smaller index wins
__device__ __forceinline__ void best_merge(float& v, int& i, float v2, int i2) {
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
// arg-max over the 256 threads of a workgroup; every thread receives the winner.  sh_v/sh_i: [4] scratch.
__device__ __forceinline__ void block_best(float& v, int& i, float* sh_v, int* sh_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o, 64);
    const int i2 = __shfl_xor(i, o, 64);
    best_merge(v, i, v2, i2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                        // scratch free (previous round read)
  if (lane == 0) { sh_v[wave] = v; sh_i[wave] = i; }
  __syncthreads();
  v = sh_v[0]; i = sh_i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) best_merge(v, i, sh_v[w], sh_i[w]);
}

__global__ __launch_bounds__(256) void beam_partial_kernel(whk::BeamArgs a, int nchunk) {
  pin_kernargs(a);
  asm volatile("" ::"s"(nchunk));      // == gridDim.x, as an argument: gridDim sits in the implicit arguments
  __shared__ int sh_last_ts;
  __shared__ float sh_m[2][4], sh_s[2][4];
  __shared__ float sh_bv[4];
  __shared__ int sh_bi[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, k = blockIdx.y;
  const float* x = a.logits + (int64_t)k * a.logits_ld;
  float xv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = c * BCHUNK + j * 256 + tid;
    xv[j] = v < a.V ? x[v] : WH_NEG_INF;
  }
  const int lag = a.lag ? a.lag[k] : 0;                   // ragged prompts: this row's sequence is `lag` tokens shorter
  const int ntok = load_uniform_int(a.d_ntok) - lag;
  const int sample_begin = a.sample_begin - lag;
  const int64_t* row = a.tokens_in + (int64_t)k * a.token_stride;
  const int L = ntok - sample_begin;
  const int TB = a.timestamp_begin;       // < 0: timestamp rules disabled (without_timestamps)
  const bool ts_rules = TB >= 0;

  if (tid == 0) sh_last_ts = -1;
  __syncthreads();
  if (ts_rules) {
    for (int t = tid; t < L; t += 256)
      if (row[sample_begin + t] >= TB) atomicMax(&sh_last_ts, t);
  }
  __syncthreads();

  bool last_ts = false, pen_ts = false;
  int ts_lo = 0, ts_hi = 0;              // forbidden timestamp interval [ts_lo, ts_hi)
  if (ts_rules) {
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
  st[0] = Stat{WH_NEG_INF, 0.f};
  st[1] = Stat{WH_NEG_INF, 0.f};
  unsigned live = 0;                       // bit j: entry j of this thread passes the filters
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = c * BCHUNK + j * 256 + tid;
    bool masked = v >= a.V;
    if (!masked) {
      masked = a.suppress_mask && a.suppress_mask[v];
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
    }
    // the survivors are tracked in a bit mask and selected at every use: a conditional `xv[j] = -inf` into the
    // register array was dropped by the compiler here (empty if-body in the ISA; every filter silently off)
    live |= (masked ? 0u : 1u) << j;
    if (!masked) { if (v < split) stat_add(st[0], xv[j]); else stat_add(st[1], xv[j]); }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    stat_wave_reduce(st[g]);
    if (lane == 0) { sh_m[g][wave] = st[g].m; sh_s[g][wave] = st[g].s; }
  }
  __syncthreads();
  if (tid < 2) {
    Stat t = Stat{WH_NEG_INF, 0.f};
    for (int w = 0; w < 4; ++w) stat_merge(t, sh_m[tid][w], sh_s[tid][w]);
    float* o = a.part_stat + (((int64_t)k * nchunk + c) * 2 + tid) * 2;
    o[0] = t.m; o[1] = t.s;
  }

  // the K best entries of the slice, per range
  const int K = a.K;
  const int lo = c * BCHUNK, hi = (lo + BCHUNK < a.V) ? lo + BCHUNK : a.V;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float* pv = a.part_val + (((int64_t)k * nchunk + c) * 2 + g) * KMAX;
    int* pi = a.part_idx + (((int64_t)k * nchunk + c) * 2 + g) * KMAX;
    const bool has = g == 0 ? lo < split : hi > split;        // uniform
    if (!has) {
      if (tid < K) { pv[tid] = WH_NEG_INF; pi[tid] = NO_IDX; }
      continue;
    }
    unsigned avail = 0;                    // live entries of this range not yet taken
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = lo + j * 256 + tid;
      avail |= ((((live >> j) & 1u) != 0 && (v < split) == (g == 0) && xv[j] != WH_NEG_INF) ? 1u : 0u) << j;
    }
    for (int kk = 0; kk < K; ++kk) {
      float bv = WH_NEG_INF; int bi = NO_IDX;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool on = ((avail >> j) & 1u) != 0;
        best_merge(bv, bi, on ? xv[j] : WH_NEG_INF, on ? lo + j * 256 + tid : NO_IDX);
      }
      block_best(bv, bi, sh_bv, sh_bi);
      if (tid == 0) { pv[kk] = bv; pi[kk] = bi; }
#pragma unroll
      for (int j = 0; j < 4; ++j) avail &= ~(((lo + j * 256 + tid == bi) ? 1u : 0u) << j);
    }
  }
}

constexpr int POOL_MAX = 2 * 64 * KMAX;     // both ranges x up to 64 chunks x K

__global__ __launch_bounds__(256) void beam_row_kernel(whk::BeamArgs a, int nchunk) {
  pin_kernargs(a);
  asm volatile("" ::"s"(nchunk));
  __shared__ float sh_m[2][4], sh_s[2][4];
  __shared__ float sh_bv[4];
  __shared__ int sh_bi[4];
  __shared__ float pool_v[POOL_MAX];
  __shared__ int pool_i[POOL_MAX];
  __shared__ int sh_text_masked;
  __shared__ float sh_norm[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x;
  const bool ts_rules = a.timestamp_begin >= 0;
  Stat st[2];
  st[0] = Stat{WH_NEG_INF, 0.f};
  st[1] = Stat{WH_NEG_INF, 0.f};
  for (int c = tid; c < nchunk; c += 256) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* o = a.part_stat + (((int64_t)k * nchunk + c) * 2 + g) * 2;
      stat_merge(st[g], o[0], o[1]);
    }
  }
  const int K = a.K;
  const int npool = 2 * nchunk * K;
  for (int e = tid; e < npool; e += 256) {
    const int cg = e / K, kk = e - cg * K;
    pool_v[e] = a.part_val[((int64_t)k * nchunk * 2 + cg) * KMAX + kk];
    pool_i[e] = a.part_idx[((int64_t)k * nchunk * 2 + cg) * KMAX + kk];
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    stat_wave_reduce(st[g]);
    if (lane == 0) { sh_m[g][wave] = st[g].m; sh_s[g][wave] = st[g].s; }
  }
  __syncthreads();
  if (tid == 0) {
    Stat tx = Stat{WH_NEG_INF, 0.f}, ts = tx;
    for (int w = 0; w < 4; ++w) {
      stat_merge(tx, sh_m[0][w], sh_s[0][w]);
      stat_merge(ts, sh_m[1][w], sh_s[1][w]);
    }
    bool text_masked = false;
    if (ts_rules) {
      // decoding.py:498-505 — logsumexp of timestamp logprobs vs max text logprob (same normaliser)
      Stat all = tx;
      stat_merge(all, ts.m, ts.s);
      if (all.m != WH_NEG_INF) {
        const float lse_all = logf(all.s);
        const float ts_lp_max = (ts.m - all.m) - lse_all;
        const float ts_lse = (ts.m == WH_NEG_INF) ? WH_NEG_INF : ts_lp_max + logf(ts.s);
        const float text_lp_max = (tx.m == WH_NEG_INF) ? WH_NEG_INF : (tx.m - all.m) - lse_all;
        if (ts_lse > text_lp_max) text_masked = true;
      }
    }
    Stat fin = ts;
    if (!text_masked) { fin = tx; stat_merge(fin, ts.m, ts.s); }
    sh_text_masked = text_masked ? 1 : 0;
    sh_norm[0] = fin.m;
    sh_norm[1] = (fin.m == WH_NEG_INF) ? 0.f : logf(fin.s);
  }
  __syncthreads();
  const bool text_masked = sh_text_masked != 0;
  const float M = sh_norm[0], LS = sh_norm[1];
  // drop the text pool when the timestamp mass rule fired: entry e belongs to range (e / K) & 1
  if (text_masked) {
    for (int e = tid; e < npool; e += 256)
      if (((e / K) & 1) == 0) pool_v[e] = WH_NEG_INF;
  }
  __syncthreads();
  for (int kk = 0; kk < K; ++kk) {
    float bv = WH_NEG_INF; int bi = NO_IDX;
    for (int e = tid; e < npool; e += 256)
      if (pool_v[e] != WH_NEG_INF) best_merge(bv, bi, pool_v[e], pool_i[e]);
    block_best(bv, bi, sh_bv, sh_bi);
    if (tid == 0) {
      a.cand_lp[(int64_t)k * KMAX + kk] = (bv == WH_NEG_INF) ? WH_NEG_INF : (bv - M) - LS;
      a.cand_tok[(int64_t)k * KMAX + kk] = (bi == NO_IDX) ? 0 : bi;
    }
    for (int e = tid; e < npool; e += 256)
      if (pool_i[e] == bi) pool_v[e] = WH_NEG_INF;       // a token sits in exactly one (chunk, range)
    // block_best's leading barrier orders these writes before the next round's reads
  }
}

constexpr int NC_MAX = 8 * KMAX;     // candidates of one segment: beam (<= 8) x K

__global__ __launch_bounds__(64) void beam_update_kernel(whk::BeamArgs a, int B) {
  pin_kernargs(a);
  asm volatile("" ::"s"(B));
  __shared__ float score[NC_MAX];
  __shared__ int ctok[NC_MAX], csrc[NC_MAX], order[NC_MAX];
  __shared__ int kept[8], nfin_list[NC_MAX];
  __shared__ int sh_n[3];               // candidates in order, kept, newly finished
  __shared__ int sh_frozen;
  __shared__ int sh_lcp[64], sh_src[8];
  const int tid = threadIdx.x;
  const int au = blockIdx.x;
  const int G = a.G, K = a.K;
  const int r0 = au * G;
  const int len = load_uniform_int(a.d_ntok) - (a.lag ? a.lag[r0] : 0);    // the beams of a segment share its prompt

  // completed at the previous update (every segment full): leave the state as it is
  int not_done = 0;
  for (int b = tid; b < B; b += 64) not_done |= a.done_prev[b] ? 0 : 1;
  not_done = __any(not_done);
  if (tid == 0) sh_frozen = not_done ? 0 : 1;
  __syncthreads();
  if (sh_frozen) {
    for (int g = 0; g < G; ++g) {
      const int64_t* s = a.tokens_in + (int64_t)(r0 + g) * a.token_stride;
      int64_t* d = a.tokens_out + (int64_t)(r0 + g) * a.token_stride;
      for (int t = tid; t < len; t += 64) d[t] = s[t];
      if (tid == 0) { a.src[r0 + g] = r0 + g; if (a.copy_from) a.copy_from[r0 + g] = 0; }
    }
    if (tid == 0) a.done_next[au] = 1;
    return;
  }
  if (a.lcp) sh_lcp[tid] = a.lcp[au * 64 + tid];

  const int N = G * K;
  for (int c = tid; c < N; c += 64) {
    const int j = c / K, kk = c - j * K;
    const bool valid = a.first ? (j == G - 1) : true;
    const float lp = a.cand_lp[(int64_t)(r0 + j) * KMAX + kk];
    score[c] = valid ? a.sum_logprobs[r0 + j] + lp : __builtin_nanf("");
    ctok[c] = a.cand_tok[(int64_t)(r0 + j) * KMAX + kk];
    csrc[c] = r0 + j;
  }
  __syncthreads();
  // stable descending order: rank = number of candidates that come before this one (NaN marks "not a candidate").
  // A row with fewer than K finite logits yields (-inf, token 0) candidates; they sort behind every finite one (in
  // index order, so the outcome is deterministic) and are only taken when a segment has nothing else to continue —
  // the reference's torch.topk picks arbitrary -inf entries in that case.
  for (int c = tid; c < N; c += 64) {
    const float s = score[c];
    if (s != s) continue;
    int rank = 0;
    for (int o = 0; o < N; ++o) {
      const float so = score[o];
      if (so != so) continue;
      if (so > s || (so == s && o < c)) ++rank;
    }
    order[rank] = c;
  }
  __syncthreads();
  if (tid == 0) {
    const int nvalid = a.first ? K : N;
    int kk = 0, nf = 0;
    for (int i = 0; i < nvalid && kk < G; ++i) {
      const int c = order[i];
      if (ctok[c] == a.eot) nfin_list[nf++] = c;
      else kept[kk++] = c;
    }
    sh_n[0] = nvalid; sh_n[1] = kk; sh_n[2] = nf;
  }
  __syncthreads();
  const int nkept = sh_n[1], nf = sh_n[2];
  // finished sequences (already in descending order), while the list holds fewer than max_candidates
  int count = a.fin_count[au];
  for (int i = 0; i < nf && count < a.max_candidates; ++i, ++count) {
    const int c = nfin_list[i];
    const int64_t* s = a.tokens_in + (int64_t)csrc[c] * a.token_stride;
    int64_t* d = a.fin_tok + ((int64_t)au * a.max_candidates + count) * a.token_stride;
    for (int t = tid; t < len; t += 64) d[t] = s[t];
    if (tid == 0) {
      d[len] = a.eot;
      a.fin_len[au * a.max_candidates + count] = len + 1;
      a.fin_score[au * a.max_candidates + count] = score[c];
    }
  }
  // the surviving beams
  for (int b = 0; b < nkept; ++b) {
    const int c = kept[b];
    const int64_t* s = a.tokens_in + (int64_t)csrc[c] * a.token_stride;
    int64_t* d = a.tokens_out + (int64_t)(r0 + b) * a.token_stride;
    for (int t = tid; t < len; t += 64) d[t] = s[t];
    if (tid == 0) {
      d[len] = ctok[c];
      a.src[r0 + b] = csrc[c];
      a.step_tokens[r0 + b] = ctok[c];
    }
  }
  __syncthreads();                 // every score has been read before the sums are replaced
  if (a.lcp) {
    // Shared history.  Rows i and j of this segment hold identical K/V at their first lcp[i][j] cache positions.  The
    // new row i continues old row s_i = src[i]: it only has to copy the positions from lcp_old[i][s_i] on (below that
    // its own bytes are already those of s_i), and new rows i, j share what their sources shared — everything so far
    // if they have the same source.  Rows that were not kept hold nothing anyone relies on: 0.
    if (tid < 8) sh_src[tid] = tid < nkept ? csrc[kept[tid]] - r0 : -1;
    __syncthreads();
    const int i = tid >> 3, j = tid & 7;
    const int si = sh_src[i], sj = sh_src[j];
    int v = 0;
    if (si >= 0 && sj >= 0) { v = si == sj ? len : sh_lcp[si * 8 + sj]; if (v > len) v = len; }
    a.lcp[au * 64 + tid] = v;
    if (j == 0 && i < G) {
      int cf = 0;
      if (si >= 0) { cf = si == i ? len : sh_lcp[i * 8 + si]; if (cf > len) cf = len; }
      a.copy_from[r0 + i] = cf;
    }
  }
  if (tid < nkept) a.sum_logprobs[r0 + tid] = score[kept[tid]];
  if (tid == 0) {
    a.fin_count[au] = count;
    a.done_next[au] = count >= a.max_candidates ? 1 : 0;
    if (au == 0) atomicAdd(a.d_applied, 1);
  }
}

}  // namespace

namespace whk {

size_t beam_scratch_bytes(int R, int V) {
  const size_t nchunk = (V + BCHUNK - 1) / BCHUNK;
  // part_stat [R][nchunk][2][2] f32, part_val / part_idx [R][nchunk][2][KMAX], cand_lp / cand_tok [R][KMAX]
  return (size_t)R * nchunk * 2 * 2 * 4 + 2 * (size_t)R * nchunk * 2 * KMAX * 4 + 2 * (size_t)R * KMAX * 4 + 256;
}

void beam_scratch_carve(BeamArgs& a, void* base, int R, int V) {
  const size_t nchunk = (V + BCHUNK - 1) / BCHUNK;
  char* p = (char*)base;
  a.part_stat = (float*)p; p += (size_t)R * nchunk * 2 * 2 * 4;
  a.part_val = (float*)p; p += (size_t)R * nchunk * 2 * KMAX * 4;
  a.part_idx = (int*)p; p += (size_t)R * nchunk * 2 * KMAX * 4;
  a.cand_lp = (float*)p; p += (size_t)R * KMAX * 4;
  a.cand_tok = (int*)p;
}

hipError_t launch_beam_step(const BeamArgs& a, int B, hipStream_t stream) {
  const int nchunk = (a.V + BCHUNK - 1) / BCHUNK;
  if (a.K != a.G + 1 || a.K > KMAX || a.G > 8 || nchunk > 64 || a.R != B * a.G) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_partial_kernel, dim3(nchunk, a.R), dim3(256), 0, stream, a, nchunk);
  hipLaunchKernelGGL(beam_row_kernel, dim3(a.R), dim3(256), 0, stream, a, nchunk);
  hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, stream, a, B);
  return hipGetLastError();
}

}  // namespace whk
