// gemv.hip — the decode-step projection: y[r][n] = sum_k x[r][k] * W[n][k] (+ bias, + fused epilogue)
// for a handful of rows r (batch x beam <= a few dozen).  This is `Linear.forward`
// (whisper/model.py:44-50) at T = 1, i.e. TextDecoder.forward (model.py:227-249) inside
// DecodingTask._main_loop (whisper/decoding.py:686-687) — the HBM-bound part of the path: every
// decoder weight is streamed from HBM exactly once per step for the whole batch.  No MFMA here: the
// arithmetic intensity is R FLOP/byte.
//
// A decode-step matrix is tiny next to the machine (D x D fp16 = 12.8 KB per CU at D = 1280), so a
// launch is a LATENCY chain, not a bandwidth problem.  The kernel is built around that:
//   * loads come back in issue order (vmcnt), so the issue order IS the schedule: first the few
//     L2-resident operands the prologue and the epilogue need (x rows, LayerNorm parameters, attention
//     partials, bias, residual), then the wave's whole share of the weight tile (up to NU = 10 x 16 B
//     per lane, non-temporal).  The prologue arithmetic then runs while the weights are in flight;
//   * the LayerNorm prologue is single-pass: a wave holds a whole fp32 row in registers
//     (K / 64 floats per lane), so mean / variance / normalise cost one L2 round trip;
//   * x sits in LDS in the element type and is read back with conflict-free broadcast ds_read_b128;
//     products are v_dot2_f32_f16 with fp32 accumulation;
//   * a wave-load covers 64 / LPR weight rows x LPR*16 contiguous bytes (LPR = 8 -> 128 B lines,
//     LPR = 16 -> 256 B for K >= 2048 so that N = D matrices still give >= 320 workgroups);
//     the four waves of a workgroup split K; partial sums meet in LDS;
//   * a workgroup may walk several feature groups (logits: V = 51866 rows) re-using the staged x,
//     with the next group's weights already in flight (double-buffered registers).
//
// Fused prologues: plain copy | LayerNorm of the fp32 residual stream (model.py:39-41) |
//                  merge of split-K attention partials (attention.hip, attn_decode).
// Fused epilogues: store | q + in-place KV-cache append (replaces torch.cat, model.py:332) |
//                  residual add | exact GELU | fp32 logits.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#ifndef WH_GEMV8_MAX_ROWS
#define WH_GEMV8_MAX_ROWS 24
#endif

namespace {

// compiler-only fence: the machine scheduler otherwise hoists the (independent) weight loads above the
// prologue's own loads, and loads return in issue order
#define ISSUE_FENCE() asm volatile("" ::: "memory")

constexpr int CS = 4;          // PRO_COMBINE fast path: splits held in registers

// 4 consecutive attention-partial values (element type T) as fp32
__device__ __forceinline__ float4v load_part4(const float* p) { return *(const float4v*)p; }
__device__ __forceinline__ float4v load_part4(const half_t* p) {
  const half4v h = *(const half4v*)p;
  return float4v{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

template <typename T> struct Pack4;
template <> struct Pack4<float> {
  static __device__ __forceinline__ void store(float* p, float a, float b, float c, float d) {
    *(float4v*)p = float4v{a, b, c, d};
  }
};
template <> struct Pack4<half_t> {
  static __device__ __forceinline__ void store(half_t* p, float a, float b, float c, float d) {
    *(half4v*)p = half4v{(half_t)a, (half_t)b, (half_t)c, (half_t)d};
  }
};

// PRO: prologue kind (whk::PRO_*); J: float4 per lane per row held by the LayerNorm prologue (K <= 256*J) or
// 16-byte x units per thread per row staged by PRO_PLAIN; MULTI: the workgroup walks several (feature group,
// K batch) items with double-buffered weights (WAVES == 4, GS == 1 only);
// WAVES x 64 threads per workgroup, arranged as GS feature-group slots x KS = WAVES / GS splits of K.
// A single wave issues one VALU instruction every ~8.6 cycles (tools/ubench_valu), so the per-wave
// instruction chain — not FLOPs — bounds these kernels: wide workgroups cut the chain per wave.
// MF (fp16 only, EXPERIMENT — instantiated only by tools/probe_decode): the dot products of a wave-load run
// as ONE v_mfma_f32_16x16x32_f16 instead of 8 x 4 v_dot2 per lane — a wave-load is then 16 weight rows x 64 B
// (lane -> row lane % 16, k-chunk lane / 16, i.e. exactly the MFMA A fragment), x is the B fragment.  Measured on
// MI355X at R = 8: no faster than the v_dot2 forms (launch + memory latency dominate, not the dot products), so
// the product path keeps the decode projections on the VALU as BASELINE.json's north_star asks.
template <typename T, int RT, int LPR, int PRO, int J, bool MULTI, int WAVES, int GS, bool MF>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(whk::GemvArgs a, int gp, int red_alias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pin_kernargs(a);
  asm volatile("" ::"s"(gp), "s"(red_alias));
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  constexpr int NT = WAVES * 64;
  constexpr int KS = WAVES / GS;         // waves splitting K for one feature group
  constexpr int NU = (WAVES == 4 || MF) ? 10 : 5;   // weight units (16 B) in flight per lane per buffer
  constexpr int NB = 64 / LPR;           // output features per group (= weight rows per wave-load); MF: LPR == 4
  constexpr int BLK = LPR * UNIT;        // K elements covered by one wave-load
  static_assert(!MF || (sizeof(T) == 2 && LPR == 4 && RT >= 8), "MFMA form: fp16, 16 rows x 32 k per wave-load");
  constexpr int NR = (RT + WAVES - 1) / WAVES;   // LayerNorm rows per wave
  constexpr int TPR = NT / RT;           // PRO_PLAIN: threads staging one row
  constexpr int CT = (RT * 20 * 16 + NT - 1) / NT;             // PRO_COMBINE fast path: tasks per thread (<= 20 heads)
  static_assert(!MULTI || (WAVES == 4 && GS == 1), "MULTI is the 4-wave streaming form");
  const int K = a.K;
  const int xld = K;
  // MF: the B-fragment read touches 16 x-rows at the same column, and a row is a multiple of 256 B, so rows would
  // collide on the LDS banks; 16-byte unit u of row r is therefore stored at unit u ^ (r & 15) (K is a multiple of
  // 16 units for every Whisper width), which makes the read conflict-free without padding the rows — padding would
  // push 16 rows x K = 5120 past the 160 KB of LDS.
  auto xcol = [&](int r, int k) -> int { return MF ? ((((k >> 3) ^ (r & 15)) << 3) | (k & 7)) : k; };
  T* xs = (T*)smem;                                              // [RT][xld]
  // cross-wave partial sums [WAVES][NB][RT]; aliases xs when the workgroup makes a single pass
  float* red = red_alias ? (float*)smem : (float*)(smem + (size_t)RT * xld * sizeof(T));
  float* csc = red + WAVES * NB * RT;                            // PRO_COMBINE: [RT*H][CS] merge weights

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave / KS, kw = wave % KS;
  // VALU form: LPR lanes walk one weight row (sub = 16-byte unit, fr = row); MFMA form: the A-fragment map
  const int sub = MF ? lane / 16 : lane % LPR, fr = MF ? lane % 16 : lane / LPR;
  const int r0 = blockIdx.y * RT;
  int R = a.R - r0; if (R > RT) R = RT;
  const int nblk = K / BLK;
  const int nbatch = ((nblk + KS - 1) / KS + NU - 1) / NU;  // batches of NU wave-loads per group (uniform)
  const int ngroups = (a.N + NB - 1) / NB;
  const int g0 = MULTI ? blockIdx.x * gp : blockIdx.x * GS;
  int g1 = g0 + (MULTI ? gp : GS); if (g1 > ngroups) g1 = ngroups;
  const int total = MULTI ? (g1 - g0) * nbatch : 1;
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
  (void)wgid;
  WH_PROBE_AT(a, wgid, 0);

  auto load_item = [&](int it, unit_t* w) {
    const int g = MULTI ? g0 + it / nbatch : g0 + slot, b = MULTI ? it - (it / nbatch) * nbatch : 0;
    int n = g * NB + fr; if (n > a.N - 1) n = a.N - 1;
    const T* base = (const T*)a.W + (int64_t)n * K + sub * UNIT;
    // branch-free: a predicated load with a zero-fill else-arm makes the compiler drain vmcnt at the join,
    // which would serialise the whole prologue behind the weight stream.  Out-of-range slots re-read the
    // last block (valid address) and are skipped by compute_item.
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int ub = kw + KS * (b * NU + u); if (ub > nblk - 1) ub = nblk - 1;
      w[u] = __builtin_nontemporal_load((const unit_t*)(base + (size_t)ub * BLK));
    }
  };
  unit_t wa[NU], wb[MULTI ? NU : 1];

  // epilogue operands of the first pass, requested ahead of the weights: thread -> (slot, row, feature)
  const int es = tid / (NB * RT), er = (tid / NB) % RT, ej = tid % NB;
  const bool e_on = es < GS && er < R;
  float e_bias = 0.f, e_res = 0.f;
  // EPI_QKV: cache position and this row's lag (ragged prompts).  The position is requested here and first USED in the
  // epilogue (every lane reads the same word; no readfirstlane, which would wait for the load on the spot)
  int e_lag = 0, e_pos = 0;
  if (a.epi == whk::EPI_QKV) e_pos = load_agent_int(a.d_pos);
  if (e_on && a.epi == whk::EPI_QKV && a.lag) e_lag = a.lag[r0 + er];
  if (e_on) {
    const int n = (g0 + es) * NB + ej;
    if (n < a.N) {
      if (a.bias) e_bias = a.bias[n];
      if (a.epi == whk::EPI_RESID) e_res = a.resid[(int64_t)(r0 + er) * a.resid_ld + n];
    }
  }

  // ---------------------------------------------------------------------- prologue -> xs
  if (PRO == whk::PRO_PLAIN) {
    // row-major staging without integer divisions: TPR threads per row, thread moves units c, c+TPR, ...
    const int upr = K / UNIT;
    const int r = tid / TPR, c = tid % TPR;
    const T* src = (const T*)a.x + (int64_t)(r0 + (r < R ? r : R - 1)) * a.x_ld;     // padded rows re-read a valid row
    unit_t xv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      int u = j * TPR + c; if (u > upr - 1) u = upr - 1;                              // branch-free, clamped
      xv[j] = *(const unit_t*)(src + u * UNIT);
    }
    ISSUE_FENCE(); load_item(0, wa); ISSUE_FENCE(); WH_PROBE_AT(a, wgid, 1);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int u = j * TPR + c;
      if (u < upr) {
        unit_t v = xv[j];
        if (r >= R) {
#pragma unroll
          for (int e = 0; e < UNIT; ++e) v[e] = 0;
        }
        *(unit_t*)(xs + (size_t)r * xld + xcol(r, u * UNIT)) = v;
      }
    }
  } else if (PRO == whk::PRO_LN) {
    // single pass: row(s) wave, wave + WAVES, ... live in registers; waves beyond RT only stream weights
    float4v v[NR][J], w4[J], b4[J];
    const bool ln_wave = wave < RT;
    if (ln_wave) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = wave + WAVES * i;
        const float* src = a.xf + (int64_t)(r0 + (r < R ? r : 0)) * a.xf_ld;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;      // branch-free; masked at use
          v[i][j] = *(const float4v*)(src + k);
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
        w4[j] = *(const float4v*)(a.ln_w + k);
        b4[j] = *(const float4v*)(a.ln_b + k);
      }
    }
    ISSUE_FENCE(); load_item(0, wa); ISSUE_FENCE(); WH_PROBE_AT(a, wgid, 1);
    if (ln_wave) {
      const float invK = 1.0f / (float)K;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = wave + WAVES * i;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
          s += ((j * 64 + lane) * 4 < K) ? t : 0.f;
        }
        const float mean = wave_sum(s) * invK;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int k = (j * 64 + lane) * 4;
          if (k < K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
          }
        }
        const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
        T* xr = xs + (size_t)r * xld;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int k = (j * 64 + lane) * 4;
          if (k < K) {
            if (r < R)
              Pack4<T>::store(xr + xcol(r, k), (v[i][j][0] - mean) * rstd * w4[j][0] + b4[j][0],
                              (v[i][j][1] - mean) * rstd * w4[j][1] + b4[j][1],
                              (v[i][j][2] - mean) * rstd * w4[j][2] + b4[j][2],
                              (v[i][j][3] - mean) * rstd * w4[j][3] + b4[j][3]);
            else
              Pack4<T>::store(xr + xcol(r, k), 0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
  } else {  // PRO_COMBINE: merge split-K attention partials (m, l, o[64]) per (row, head); K == H * 64
    const int S = a.splits, H = a.H;
    const int per_row = H * 16, ntask = RT * per_row;
    const bool fast = S <= CS && ntask <= NT * CT && RT * H <= NT;
    if (fast) {
      // all partial sums requested before the weights; scale factors exp(m_s - M) / den via LDS
      float4v o[CT][CS];
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        int i = j * NT + tid; if (i > ntask - 1) i = ntask - 1;        // branch-free loads, clamped
        int r = i / per_row; const int rem = i - r * per_row;
        if (r > R - 1) r = R - 1;
        const int h = rem >> 4, d4 = rem & 15;
        // partials are split-major [S][rows][H][64], normalised (attention.hip)
#pragma unroll
        for (int s = 0; s < CS; ++s)
          o[j][s] = load_part4((const T*)a.part_o + (((int64_t)(s < S ? s : S - 1) * a.R + (r0 + r)) * H + h) * 64 + d4 * 4);
      }
      float2v ml[CS];
      const int pt = tid < RT * H ? tid : RT * H - 1;
      const int pr = pt / H, ph = pt - pr * H;                   // one (row, head) pair per thread
      const bool pair = tid < RT * H && pr < R;
#pragma unroll
      for (int s = 0; s < CS; ++s)
        ml[s] = *(const float2v*)(a.part_ml + (((int64_t)(s < S ? s : S - 1) * a.R + (r0 + (pr < R ? pr : R - 1))) * H + ph) * 2);
      ISSUE_FENCE(); load_item(0, wa); ISSUE_FENCE(); WH_PROBE_AT(a, wgid, 1);
#pragma unroll
      for (int s = 0; s < CS; ++s)
        if (s >= S) ml[s] = float2v{WH_NEG_INF, 0.f};
      if (tid < RT * H) {
        float M = WH_NEG_INF;
#pragma unroll
        for (int s = 0; s < CS; ++s) M = fmaxf(M, ml[s][0]);
        float w[CS], den = 0.f;
#pragma unroll
        for (int s = 0; s < CS; ++s) { w[s] = (ml[s][0] == WH_NEG_INF) ? 0.f : __expf(ml[s][0] - M); den = __builtin_fmaf(w[s], ml[s][1], den); }
        const float inv = pair ? 1.0f / den : 0.f;
#pragma unroll
        for (int s = 0; s < CS; ++s) csc[tid * CS + s] = w[s] * ml[s][1] * inv;      // partial o is o_s / l_s
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int i = j * NT + tid;
        if (i < ntask) {
          const int r = i / per_row, rem = i - r * per_row;
          const int h = rem >> 4, d4 = rem & 15;
          const float* f = csc + (r * H + h) * CS;
          float4v num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < CS; ++s) {
            const float w = f[s];
            num[0] = __builtin_fmaf(w, o[j][s][0], num[0]); num[1] = __builtin_fmaf(w, o[j][s][1], num[1]);
            num[2] = __builtin_fmaf(w, o[j][s][2], num[2]); num[3] = __builtin_fmaf(w, o[j][s][3], num[3]);
          }
          Pack4<T>::store(xs + (size_t)r * xld + xcol(r, h * 64 + d4 * 4), num[0], num[1], num[2], num[3]);
        }
      }
    } else {
      // generic path (many splits / wide rows): merge first, then start the weight stream
      for (int i = tid; i < ntask; i += NT) {
        const int r = i / per_row, rem = i - r * per_row;
        const int h = rem >> 4, d4 = rem & 15;
        float4v num = {0.f, 0.f, 0.f, 0.f};
        float den = 1.f;
        if (r < R) {
          const int64_t pb = (int64_t)(r0 + r) * H + h, ps = (int64_t)a.R * H;     // [S][rows][H], normalised partials
          float M = WH_NEG_INF;
          for (int s = 0; s < S; ++s) M = fmaxf(M, a.part_ml[(pb + s * ps) * 2]);
          den = 0.f;
          for (int s = 0; s < S; ++s) {
            const float2v ml = *(const float2v*)(a.part_ml + (pb + s * ps) * 2);
            const float w = (ml[0] == WH_NEG_INF ? 0.f : __expf(ml[0] - M)) * ml[1];
            const float4v o = load_part4((const T*)a.part_o + (pb + s * ps) * 64 + d4 * 4);
            num[0] = __builtin_fmaf(w, o[0], num[0]); num[1] = __builtin_fmaf(w, o[1], num[1]);
            num[2] = __builtin_fmaf(w, o[2], num[2]); num[3] = __builtin_fmaf(w, o[3], num[3]);
            den += w;
          }
        }
        const float inv = 1.0f / den;
        Pack4<T>::store(xs + (size_t)r * xld + xcol(r, h * 64 + d4 * 4), num[0] * inv, num[1] * inv, num[2] * inv, num[3] * inv);
      }
      load_item(0, wa);
    }
  }
  WH_PROBE_AT(a, wgid, 2);
  __syncthreads();
  WH_PROBE_AT(a, wgid, 3);

  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;
  float4v macc = {0.f, 0.f, 0.f, 0.f};   // MF: out[x row lane % 16][feature 4 * (lane / 16) + e]

  // x of weight unit u for all rows, read one unit ahead of the dot products
  auto compute_item = [&](int it, const unit_t* w) {
    const int b = MULTI ? it - (it / nbatch) * nbatch : 0;
    if constexpr (MF) {
      const int xr_ = fr % RT;                                         // B fragment: column = x row (duplicated)
      const T* xrow = xs + (size_t)xr_ * xld;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int ub = kw + KS * (b * NU + u);
        if (ub < nblk) {
          const unit_t xf = *(const unit_t*)(xrow + xcol(xr_, ub * BLK + sub * UNIT));
          macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], xf, macc, 0, 0, 0);
        }
      }
      return;
    }
    unit_t xc[RT], xn[RT];
    auto fetch = [&](int u, unit_t* x) {
      int ub = kw + KS * (b * NU + u); if (ub > nblk - 1) ub = nblk - 1;
      const T* xb = xs + (size_t)ub * BLK + sub * UNIT;
#pragma unroll
      for (int r = 0; r < RT; ++r) x[r] = *(const unit_t*)(xb + (size_t)r * xld);
    };
    fetch(0, xc);
#pragma unroll
    for (int u = 0; u < NU; u += 2) {
      if (u + 1 < NU) fetch(u + 1, xn);
      if (kw + KS * (b * NU + u) < nblk) {
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = dot_unit(w[u], xc[r], acc[r]);
      }
      if (u + 2 < NU) fetch(u + 2, xc);
      if (u + 1 < NU && kw + KS * (b * NU + u + 1) < nblk) {
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = dot_unit(w[u + 1], xn[r], acc[r]);
      }
    }
  };

  auto finish_group = [&](int it) {
    const int gbase = MULTI ? g0 + it / nbatch : g0;     // group of slot 0 in this pass
    if constexpr (!MF) {
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = LPR == 8 ? group8_sum(acc[r]) : group16_sum(acc[r]);
    }
    if (red_alias) __syncthreads();           // every wave is done reading xs before `red` overwrites it
    if constexpr (MF) {
      if (fr < RT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * NB + sub * 4 + e) * RT + fr] = macc[e];
      }
      macc = float4v{0.f, 0.f, 0.f, 0.f};
    } else if (sub == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r) red[(wave * NB + fr) * RT + r] = acc[r];
    }
    __syncthreads();
    if (e_on) {
      const int g = gbase + es;
      const int n = g * NB + ej;
      if (g < ngroups && n < a.N) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) v += red[((es * KS + k) * NB + ej) * RT + er];
        const int64_t rr = r0 + er;
        if (gbase == g0) v += e_bias;
        else if (a.bias) v += a.bias[n];
        switch (a.epi) {
          case whk::EPI_STORE: ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(v); break;
          case whk::EPI_GELU: ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(gelu_erf(v)); break;
          case whk::EPI_F32: ((float*)a.y)[rr * a.y_ld + n] = v; break;
          case whk::EPI_RESID:
            a.resid[rr * a.resid_ld + n] = (gbase == g0 ? e_res : a.resid[rr * a.resid_ld + n]) + v;
            break;
          case whk::EPI_QKV: {
            const int D = a.D;
            if (n < D) ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(v);
            else {
              const int64_t pos = e_pos - e_lag;
              if (n < 2 * D) ((T*)a.kcache)[rr * a.cache_bs + pos * D + (n - D)] = from_f32<T>(v);
              else ((T*)a.vcache)[rr * a.cache_bs + pos * D + (n - 2 * D)] = from_f32<T>(v);
            }
          } break;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
    if (it + 1 < total) __syncthreads();      // `red` is reused by the next group
  };

  if (MULTI) {
    for (int it = 0; it < total; it += 2) {
      if (it + 1 < total) load_item(it + 1, wb);
      compute_item(it, wa);
      if (it == 0) WH_PROBE_AT(a, wgid, 4);
      if ((it + 1) % nbatch == 0) finish_group(it);
      if (it == 0) WH_PROBE_AT(a, wgid, 5);
      if (it + 1 < total) {
        if (it + 2 < total) load_item(it + 2, wa);
        compute_item(it + 1, wb);
        if ((it + 2) % nbatch == 0) finish_group(it + 1);
      }
    }
  } else {
    compute_item(0, wa);
    WH_PROBE_AT(a, wgid, 4);
    finish_group(0);
    WH_PROBE_AT(a, wgid, 5);
  }

  WH_PROBE_AT(a, wgid, 6);
  if (a.bump && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump, a.bump_by);
  if (a.bump2 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump2, 1);
}

template <typename T, int RT, int LPR, int PRO, int J, bool MULTI, int WAVES, int GS, bool MF = false>
hipError_t launch_cfg(const whk::GemvArgs& a, int gp, hipStream_t stream) {
  constexpr int NB = 64 / LPR;
  const int ngroups = (a.N + NB - 1) / NB;
  const int red_alias = (!MULTI || gp == 1) && PRO != whk::PRO_COMBINE ? 1 : 0;
  size_t lds = (size_t)RT * a.K * sizeof(T);
  if (MF && a.K % 128 != 0) return hipErrorInvalidValue;          // the unit swizzle stays inside a row
  const size_t red_bytes = (size_t)WAVES * NB * RT * sizeof(float);
  if (!red_alias) lds += red_bytes;
  else if (lds < red_bytes) lds = red_bytes;
  if (PRO == whk::PRO_COMBINE) lds += (size_t)RT * a.H * CS * sizeof(float);   // merge weights behind `red`
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static whk::LdsAttr attr;
  { hipError_t e = whk::raise_dynamic_lds(attr, (const void*)gemv_kernel<T, RT, LPR, PRO, J, MULTI, WAVES, GS, MF>, 160 * 1024); if (e != hipSuccess) return e; }
  constexpr int KS_ = WAVES / GS, NU_ = (WAVES == 4 || MF) ? 10 : 5, BLK_ = LPR * ET<T>::UNIT;
  if (!MULTI && (a.K / BLK_ + KS_ - 1) / KS_ > NU_) return hipErrorInvalidValue;   // one batch per wave only
  const int per_wg = MULTI ? gp : GS;
  dim3 grid((ngroups + per_wg - 1) / per_wg, (a.R + RT - 1) / RT), block(WAVES * 64);
  hipLaunchKernelGGL((gemv_kernel<T, RT, LPR, PRO, J, MULTI, WAVES, GS, MF>), grid, block, lds, stream, a, gp, red_alias);
  return hipGetLastError();
}

template <typename T, int RT, int LPR, bool MULTI, int WAVES, int GS, bool MF = false>
hipError_t launch_pro(const whk::GemvArgs& a, int gp, hipStream_t stream) {
  constexpr int TPR = WAVES * 64 / RT;
  switch (a.pro) {
    case whk::PRO_PLAIN: {
      const int upr = a.K / ET<T>::UNIT;            // J = 16-byte units per thread per row
      if (upr <= TPR) return launch_cfg<T, RT, LPR, whk::PRO_PLAIN, 1, MULTI, WAVES, GS, MF>(a, gp, stream);
      if (upr <= 3 * TPR) return launch_cfg<T, RT, LPR, whk::PRO_PLAIN, 3, MULTI, WAVES, GS, MF>(a, gp, stream);
      if (upr <= 6 * TPR) return launch_cfg<T, RT, LPR, whk::PRO_PLAIN, 6, MULTI, WAVES, GS, MF>(a, gp, stream);
      if (upr <= 12 * TPR) return launch_cfg<T, RT, LPR, whk::PRO_PLAIN, 12, MULTI, WAVES, GS, MF>(a, gp, stream);
      if constexpr (WAVES == 4) {                    // narrow workgroups stage long rows with more units per thread (a
                                                     // 16-wave instantiation would spill: 128 VGPRs per lane)
        if (upr <= 24 * TPR) return launch_cfg<T, RT, LPR, whk::PRO_PLAIN, 24, MULTI, WAVES, GS, MF>(a, gp, stream);
      }
      return hipErrorInvalidValue;
    }
    case whk::PRO_LN:
      if (a.K <= 256 * 5) return launch_cfg<T, RT, LPR, whk::PRO_LN, 5, MULTI, WAVES, GS, MF>(a, gp, stream);
      if constexpr (WAVES <= 8) {                    // rows wider than 1280 (no released Whisper width): 8 float4 per lane per
                                                     // row only fit the register budget of <= 8-wave workgroups
        if (a.K <= 256 * 8) return launch_cfg<T, RT, LPR, whk::PRO_LN, 8, MULTI, WAVES, GS, MF>(a, gp, stream);
      }
      return hipErrorInvalidValue;
    case whk::PRO_COMBINE:
      if (a.K != a.H * 64) return hipErrorInvalidValue;
      return launch_cfg<T, RT, LPR, whk::PRO_COMBINE, 1, MULTI, WAVES, GS, MF>(a, gp, stream);
  }
  return hipErrorInvalidValue;
}

// shape heuristics (large-v3, R = 8, every candidate measured with tools/probe_decode; see profiles/).
// Fewer, fatter workgroups win whenever a prologue is shared: a CU sustains >= 50 B/clk on this stream, so
// covering all 256 CUs is not the constraint — the redundant LayerNorm / merge prologues and the per-wave
// instruction chain are.
//   V x D logits ................ wave-private streaming kernel (gemv_stream_kernel): no barriers after the prologue
//   D x 4D (fc2) ................ one 16-wave workgroup per 8-feature group (K split 16 ways)
//   LN + 4D x D (fc1) ........... 16 waves = 4 groups x 4 K-splits   (160 workgroups)
//   LN + 3D x D (qkv) ........... 8 waves = 2 groups x 4 K-splits    (240 workgroups)
//   LN / merge + D x D .......... 8 waves, one group
//   plain D x D (out) ........... 4 waves, one group
constexpr int PRO_LN_ = whk::PRO_LN, PRO_PLAIN_ = whk::PRO_PLAIN, EPI_F32_ = whk::EPI_F32;

template <typename T, int RT> hipError_t launch_stream(const whk::GemvArgs& a, hipStream_t stream);

template <typename T, int RT>
hipError_t launch_rt(const whk::GemvArgs& a, hipStream_t stream) {
  constexpr int UNIT = ET<T>::UNIT;
  if (a.K % (8 * UNIT) != 0) return hipErrorInvalidValue;
  const int ngroups8 = (a.N + 7) / 8;
  const int nblk8 = a.K / (8 * UNIT);
  const int force = a.variant > 0 ? a.variant : 0;       // developer override (probe tool); 0 = heuristic, < 0 = heuristic
                                                         // among the v_dot2 forms only (skips the MFMA diagonal kernel)
  if (force == 0) {
    if (ngroups8 > 1024 && a.pro == PRO_LN_ && a.epi == EPI_F32_ && a.K <= 2048) return launch_stream<T, RT>(a, stream);
    if (ngroups8 > 1024) return launch_pro<T, RT, 8, true, 4, 1>(a, (ngroups8 + 1023) / 1024, stream);
    if (nblk8 >= 64 && (nblk8 + 15) / 16 <= 5) return launch_pro<T, RT, 8, false, 16, 1>(a, 1, stream);
    if (a.pro == PRO_LN_ && ngroups8 >= 600 && (nblk8 + 3) / 4 <= 5) return launch_pro<T, RT, 8, false, 16, 4>(a, 1, stream);
    if (a.pro == PRO_LN_ && ngroups8 >= 400 && (nblk8 + 3) / 4 <= 5) return launch_pro<T, RT, 8, false, 8, 2>(a, 1, stream);
    if (a.pro != PRO_PLAIN_ && (nblk8 + 7) / 8 <= 5) return launch_pro<T, RT, 8, false, 8, 1>(a, 1, stream);
    if ((nblk8 + 3) / 4 <= 10) return launch_pro<T, RT, 8, false, 4, 1>(a, 1, stream);
    return launch_pro<T, RT, 8, true, 4, 1>(a, 1, stream);
  }
#ifdef WH_PROBE   // MFMA forms: measured, not faster on these latency-bound launches (profiles/r01_gemv_shape_sweep.txt);
                  // compiled only into tools/probe_decode so the comparison stays reproducible
  if constexpr (sizeof(T) == 2 && RT >= 8) {
    if (a.K % 32 == 0) {
      switch (force) {
        case 11: return launch_pro<T, RT, 4, false, 4, 1, true>(a, 1, stream);
        case 12: return launch_pro<T, RT, 4, false, 8, 1, true>(a, 1, stream);
        case 13: return launch_pro<T, RT, 4, false, 8, 2, true>(a, 1, stream);
        case 14: return launch_pro<T, RT, 4, false, 16, 4, true>(a, 1, stream);
        case 15: return launch_pro<T, RT, 4, false, 16, 1, true>(a, 1, stream);
        case 16: return launch_pro<T, RT, 4, false, 16, 2, true>(a, 1, stream);
        case 17: return launch_pro<T, RT, 4, true, 4, 1, true>(a, ((a.N + 15) / 16 + 1023) / 1024, stream);
        default: break;
      }
    }
  }
#endif
#ifdef WH_PROBE
  switch (force) {
    case 1: return launch_pro<T, RT, 8, false, 4, 1>(a, 1, stream);
    case 2: return launch_pro<T, RT, 16, false, 4, 1>(a, 1, stream);
    case 3: return launch_pro<T, RT, 8, false, 8, 1>(a, 1, stream);
    case 4: return launch_pro<T, RT, 8, false, 16, 1>(a, 1, stream);
    case 5: return launch_pro<T, RT, 16, false, 8, 1>(a, 1, stream);
    case 6: return launch_pro<T, RT, 8, false, 8, 2>(a, 1, stream);
    case 7: return launch_pro<T, RT, 8, false, 16, 2>(a, 1, stream);
    case 8: return launch_pro<T, RT, 8, false, 16, 4>(a, 1, stream);
    default: return hipErrorInvalidValue;
  }
#else
  return hipErrorInvalidValue;
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// More than 8 rows (beam search: rows = audios x beams, e.g. 8 x 5 = 40): the step is a skinny GEMM, not a matvec.
// Row tiles of 16 on grid.y and the MFMA form of the kernel (one v_mfma_f32_16x16x32_f16 per wave-load, all 16
// columns live): 3 passes over the weights for 40 rows instead of 5 passes of 64 v_dot2 per 16 bytes.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
hipError_t launch_rows16_mf(const whk::GemvArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    const int ngroups16 = (a.N + 15) / 16;
    const int nks = a.K / 32;
    if (nks > 40) {                                    // D x 4D: K split 16 ways, x fills the LDS exactly at K = 5120
      if ((nks + 15) / 16 <= 10) return launch_pro<T, 16, 4, false, 16, 1, true>(a, 1, stream);
      return hipErrorInvalidValue;
    }
    if (ngroups16 > 1024) return launch_pro<T, 16, 4, true, 4, 1, true>(a, (ngroups16 + 1023) / 1024, stream);
    if (a.pro == whk::PRO_LN && ngroups16 >= 300) return launch_pro<T, 16, 4, false, 16, 4, true>(a, 1, stream);
    if (a.pro == whk::PRO_LN && ngroups16 >= 200) return launch_pro<T, 16, 4, false, 8, 2, true>(a, 1, stream);
    if (a.pro != whk::PRO_PLAIN) return launch_pro<T, 16, 4, false, 8, 1, true>(a, 1, stream);
    return launch_pro<T, 16, 4, false, 4, 1, true>(a, 1, stream);
  } else {
    return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-private streaming form for the tied logits projection (V x D, 133 MB at large-v3): after the shared
// LayerNorm prologue every wave owns whole 8-feature groups — it walks all of K itself, so there is no cross-wave
// reduction, no barrier and no LDS traffic besides the broadcast x reads; waves of a workgroup drift apart freely
// and the matrix streams at the HBM rate.  Epilogue: fp32 logits (EPI_F32) only; 8 waves per workgroup.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int RT, int J>
__global__ __launch_bounds__(512) void gemv_stream_kernel(whk::GemvArgs a, int groups_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pin_kernargs(a);
  asm volatile("" ::"s"(groups_per_wave));
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT, WAVES = 8, LPR = 8, NB = 8, BLK = LPR * UNIT, NU = 10;
  constexpr int NR = (RT + WAVES - 1) / WAVES;
  const int K = a.K;
  T* xs = (T*)smem;                                              // [RT][K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane % LPR, fr = lane / LPR;
  const int r0 = blockIdx.y * RT;
  int R = a.R - r0; if (R > RT) R = RT;
  const int nblk = K / BLK;
  const int ngroups = (a.N + NB - 1) / NB;
  const int gfirst = (blockIdx.x * WAVES + wave) * groups_per_wave;
  int glast = gfirst + groups_per_wave; if (glast > ngroups) glast = ngroups;

  auto load_units = [&](int g, int b0, unit_t* w) {
    int n = g * NB + fr; if (n > a.N - 1) n = a.N - 1;
    const T* base = (const T*)a.W + (int64_t)n * K + sub * UNIT;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int ub = b0 + u; if (ub > nblk - 1) ub = nblk - 1;
      w[u] = __builtin_nontemporal_load((const unit_t*)(base + (size_t)ub * BLK));
    }
  };
  unit_t w[NU];

  // ---- LayerNorm prologue (rows wave, wave + 8, ...), weights of the first batch issued behind its loads
  {
    float4v v[NR][J], w4[J], b4[J];
    const bool ln_wave = wave < RT;
    if (ln_wave) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = wave + WAVES * i;
        const float* src = a.xf + (int64_t)(r0 + (r < R ? r : 0)) * a.xf_ld;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
          v[i][j] = *(const float4v*)(src + k);
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
        w4[j] = *(const float4v*)(a.ln_w + k);
        b4[j] = *(const float4v*)(a.ln_b + k);
      }
    }
    ISSUE_FENCE();
    if (gfirst < glast) load_units(gfirst, 0, w);
    ISSUE_FENCE();
    if (ln_wave) {
      const float invK = 1.0f / (float)K;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = wave + WAVES * i;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
          s += ((j * 64 + lane) * 4 < K) ? t : 0.f;
        }
        const float mean = wave_sum(s) * invK;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int k = (j * 64 + lane) * 4;
          if (k < K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
          }
        }
        const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
        T* xr = xs + (size_t)r * K;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int k = (j * 64 + lane) * 4;
          if (k < K) {
            if (r < R)
              Pack4<T>::store(xr + k, (v[i][j][0] - mean) * rstd * w4[j][0] + b4[j][0],
                              (v[i][j][1] - mean) * rstd * w4[j][1] + b4[j][1],
                              (v[i][j][2] - mean) * rstd * w4[j][2] + b4[j][2],
                              (v[i][j][3] - mean) * rstd * w4[j][3] + b4[j][3]);
            else
              Pack4<T>::store(xr + k, 0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- each wave: its groups, all of K, NU wave-loads per batch
  for (int g = gfirst; g < glast; ++g) {
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
    for (int b0 = 0; b0 < nblk; b0 += NU) {
      if (!(g == gfirst && b0 == 0)) load_units(g, b0, w);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (b0 + u < nblk) {
          const T* xb = xs + (size_t)(b0 + u) * BLK + sub * UNIT;
#pragma unroll
          for (int r = 0; r < RT; ++r) acc[r] = dot_unit(w[u], *(const unit_t*)(xb + (size_t)r * K), acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = group8_sum(acc[r]);
    const int n = g * NB + fr;
    if (n < a.N && sub < R) {
      // lane `sub` of each feature's 8-lane group stores row `sub`: 8 stores per feature spread over 8 lanes
      float v = acc[0];
#pragma unroll
      for (int r = 1; r < RT; ++r) v = (sub == r) ? acc[r] : v;
      if (a.bias) v += a.bias[n];
      ((float*)a.y)[(int64_t)(r0 + sub) * a.y_ld + n] = v;
    }
  }
  if (a.bump && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump, a.bump_by);
  if (a.bump2 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump2, 1);
}

template <typename T, int RT>
hipError_t launch_stream(const whk::GemvArgs& a, hipStream_t stream) {
  constexpr int UNIT = ET<T>::UNIT;
  static_assert(RT <= 8, "one lane per row in the store epilogue");
  if (a.pro != whk::PRO_LN || a.epi != whk::EPI_F32 || a.K % (8 * UNIT) != 0 || a.K > 256 * 8) return hipErrorInvalidValue;
  const size_t lds = (size_t)RT * a.K * sizeof(T);
  const int ngroups = (a.N + 7) / 8;
  // ~2 waves per SIMD over the whole chip: 256 CUs x 2 workgroups x 8 waves
  int gpw = (ngroups + 256 * 2 * 8 - 1) / (256 * 2 * 8);
  if (gpw < 1) gpw = 1;
  dim3 grid((ngroups + 8 * gpw - 1) / (8 * gpw), (a.R + RT - 1) / RT), block(512);
  if (a.K <= 256 * 5) {
    static whk::LdsAttr attr5;
    { hipError_t e = whk::raise_dynamic_lds(attr5, (const void*)gemv_stream_kernel<T, RT, 5>, 160 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((gemv_stream_kernel<T, RT, 5>), grid, block, lds, stream, a, gpw);
  } else {
    static whk::LdsAttr attr8;
    { hipError_t e = whk::raise_dynamic_lds(attr8, (const void*)gemv_stream_kernel<T, RT, 8>, 160 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((gemv_stream_kernel<T, RT, 8>), grid, block, lds, stream, a, gpw);
  }
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// R <= 8 rows, fp16 — the decode step proper (batch of 8 clips): MFMA "diagonal" form, no LDS staging of x rows.
//
// tools/probe_floor showed that a dependent launch which only streams a D x D matrix and reduces costs 2.9 us
// (3.3 MB, 160 workgroups), while the v_dot2 kernels above take 5.0-6.7 us for the same bytes: the gap is the
// per-wave instruction chain (32 v_dot2 + 8 ds_read_b128 per 16 bytes of weights per lane), the 40 KB x block every
// workgroup pushes through the texture addresser and LDS, and the weight requests waiting behind the prologue's own
// loads.  Here one v_mfma_f32_16x16x32_f16 consumes a whole wave-load (64 lanes x 16 B of weights), the x operand is
// produced directly in fragment layout, and the waves of a workgroup are specialised:
//   * weight waves (GS feature-group slots x KS splits of K): request their NU wave-loads of weights as their very
//     first memory instructions, then wait for the x fragments, run NU MFMAs, and exchange 64 partial sums through LDS;
//   * prologue waves (KS of them; none for PRO_PLAIN): read the fp32 residual rows / the attention partials for the
//     K range kw of all 8 rows, LayerNorm or merge them in registers and publish fp16 fragments to LDS in fragment
//     order (lane-linear 16-byte units: conflict-free both ways).  Their L2 round trip and arithmetic run under the
//     HBM latency of the weight stream instead of in front of it.
//
// Fragment map.  A 16x16x32 MFMA wants 16 A rows x 32 k; with 16 distinct weight rows a wave-load would be 16 rows x
// 64 B — half cache lines.  Instead the 16 A rows are 8 weight rows x 2 halves of a 64-element K block, and the 16 B
// columns are the 8 batch rows x the same 2 halves: lane l = 16 c + 8 half + i holds weight row i (A) / batch row i
// (B), elements [64 blk + 32 half + 8 c, +8).  A wave-load is then 8 weight rows x 128 contiguous bytes, and of the
// 16 x 16 products only the two diagonal 8 x 8 blocks (same half on both sides) are meaningful:
// y[n][r] = C[n][r] + C[8 + n][8 + r].  Off-diagonal lanes are zeroed and the pair is summed with one DPP row_ror:8
// and one v_permlane32_swap.  Half of the MFMA work is discarded — irrelevant next to the HBM stream.
//
// LayerNorm: a prologue wave holds, for each of the 8 rows, 64 NU elements; per-wave (sum, sum of squares) meet in 64
// floats of LDS.  The LayerNorm weight / bias are folded into the projection at load time
// (WH_WEIGHTS_DEC_LN_FOLDED, whisper_hip.h), so the prologue is (x - mean) * rstd.
// Addresses are a wave-uniform base (SGPR pair, per K block) plus one 32-bit per-lane offset, so that a load costs
// no vector ALU work beyond its issue slot.
// CSm: PRO_COMBINE — exactly the number of attention splits (2..4), so that no partial is requested twice.
// ---------------------------------------------------------------------------------------------------------------
// fw: output features per workgroup (<= 8 GS; slot s covers features [8 s, min(8 s + 8, fw)) of the workgroup's range):
// the launcher picks it so that a projection spreads over all 256 CUs (1280 features -> 256 workgroups x 5) — the time
// of these kernels is set by the line requests of the busiest CU.
// NRT: row tiles of 8 handled by ONE workgroup against the same weight fragments (beam search: 8 clips x 5 beams = 40
// rows -> NRT = 6; the few-row prefill likewise): the weights are streamed once for all rows, every further tile costs
// NU more MFMAs and its x fragments.
template <int PRO, int GS, int KS, int NU, int CSm, int XW, int NRT>
__global__ __launch_bounds__((GS * KS + XW) * 64) void gemv8_kernel(whk::GemvArgs a, int fw) {
  pin_kernargs(a);
  asm volatile("" ::"s"(fw));
  static_assert((PRO == whk::PRO_PLAIN) == (XW == 0), "prologue waves exist exactly when there is a prologue");
  static_assert(PRO != whk::PRO_LN || KS == 4 || KS == 2, "LayerNorm prologue: one wave-load of fp32 = 256 elements = 4 K blocks");
  constexpr int MW = GS * KS;                              // weight (MFMA) waves; XW prologue waves in front of them
  constexpr int NT = (MW + XW) * 64;
  constexpr int RW = 8 * NRT;                              // rows per workgroup
  constexpr int FRAG = KS * NU * 64;                       // x fragment units of one row tile
  __shared__ float red[MW][NRT][8][8];                     // [weight wave][row tile][feature][row] partial sums
  __shared__ __attribute__((aligned(16))) half8v xfrag[XW ? NRT * FRAG : 1];   // [row tile][kw][u][lane]
  // memory-order weight loads pay only where a workgroup issues MANY of them next to as many x loads: the 16-wave K = 5120 shape
  // (FC2: 7.65 -> 7.30 us at 24 rows, 5.80 -> 5.44 at 8; the 4-wave and LayerNorm shapes lose 0.1 - 0.3 us, tools/probe_gemv8)
  constexpr bool W_ORDERED_OK = PRO == whk::PRO_PLAIN && KS == 16;
  __shared__ __attribute__((aligned(16))) half8v turn[W_ORDERED_OK ? MW : 1][64];   // one wave-load per weight wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_x = wave < XW;                             // wave-uniform role
  const int K = a.K, nblk = K >> 6;
  const int r0 = blockIdx.y * RW;
  int R = a.R - r0; if (R > RW) R = RW;
  const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
  (void)wgid;
  WH_PROBE_AT(a, wgid, 0);

  // x fragments of a weight wave: all row tiles at once when they come from global memory (PRO_PLAIN: requested up front); with
  // prologue waves they are read from LDS one row tile at a time, right before that tile's MFMAs (registers: NU, not NRT x NU)
  half8v wa[NU], xb[XW ? 1 : NRT][XW ? 1 : NU];
  const int mw = wave - XW;
  const int kw = mw % KS;                                  // weight waves: split of K

  if (!is_x) {
    // ================= weight waves: the whole share of the weight tile as the very first memory instructions
    // (8 rows x 128 contiguous bytes per wave-load, non-temporal); lane l = 16 c + 8 half + i
    const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;
    // rows beyond the workgroup's range re-read its last row (same cache lines: no extra requests); their outputs are dropped
    int n_last = (blockIdx.x + 1) * fw - 1; if (n_last > a.N - 1) n_last = a.N - 1;
    // W_ORDERED: the lanes in MEMORY order — lane 8 r + s requests piece s ^ r (16 bytes) of row r's 128-byte block — and the
    // operand order restored through 1 KB of wave-private LDS before the MFMAs (turn[]): the same 8 lines, a third of the
    // memory pipeline's cycles per wave-load (kernels.h "fragment order")
    const bool w_ordered = W_ORDERED_OK && a.w_ordered;
    const int wrow = w_ordered ? (lane >> 3) : idx;
    const int wpiece = w_ordered ? ((lane & 7) ^ (lane >> 3)) : (4 * ((lane >> 3) & 1) + (lane >> 4));
    int n = blockIdx.x * fw + (mw / KS) * 8 + wrow; if (n > n_last) n = n_last;
    const uint32_t lane_off = ((uint32_t)n * (uint32_t)K + (uint32_t)wpiece * 8u) * 2u;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int blk = kw + KS * u; if (blk > nblk - 1) blk = nblk - 1;          // wave-uniform; clamped, masked through x == 0
      wa[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)a.W + (size_t)blk * 128 + lane_off));
    }
    if (PRO == whk::PRO_PLAIN) {
      if (a.x_frag) {
        // fragment-order rows: unit (row tile, K block) = 1 KB, lane-linear (rows beyond R hold whatever the producer left:
        // they only reach output columns that are dropped)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const char* tile = (const char*)a.x + (size_t)((r0 >> 3) + rt) * nblk * 1024 + (uint32_t)lane * 16u;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            int blk = kw + KS * u; if (blk > nblk - 1) blk = nblk - 1;
            xb[rt][u] = *(const half8v*)(tile + (size_t)blk * 1024);
          }
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          int rr = rt * 8 + idx; if (rr > R - 1) rr = R - 1;  // padded rows re-read a valid row; their outputs are dropped
          const uint32_t xoff = ((uint32_t)(r0 + rr) * (uint32_t)a.x_ld + (uint32_t)koff) * 2u;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            int blk = kw + KS * u; if (blk > nblk - 1) blk = nblk - 1;
            xb[rt][u] = *(const half8v*)((const char*)a.x + (size_t)blk * 128 + xoff);
          }
        }
      }
    }
  } else if (PRO == whk::PRO_LN) {
    // ================= LayerNorm waves.  Loads are row-contiguous (a wave-load = 1 KB of one row = 8 full cache
    // lines: the number of line requests a CU keeps in flight is what bounds these kernels); wave w owns rows w,
    // w + XW, ... whole, so the statistics never leave the wave; results are scattered to LDS in MFMA fragment order:
    // element k of row r -> tile r / 8, unit ((k >> 6) % KS, (k >> 6) / KS), lane 16 ((k & 31) >> 3) + 8 ((k & 63) >> 5) + r % 8.
    constexpr int NR = XW ? (RW + XW - 1) / XW : 1;
    const float invK = 1.0f / (float)K;
    // LDS byte address of this lane's 4 elements of wave-load j, row r: + j * 1024 + (r / 8) * FRAG * 16 + (r % 8) * 16
    // (a wave-load of fp32 covers 4 K blocks: block 4 j + q, q = lane >> 4, is unit ((4 j + q) % KS, (4 j + q) / KS) — for KS = 4
    // unit (q, j), for KS = 2 unit (q & 1, 2 j + (q >> 1)): a per-lane base plus j * JSTRIDE bytes)
    constexpr int NJ = NU * KS / 4;                        // wave-loads of fp32 per row
    constexpr int JSTRIDE = (4 / KS) * 1024;
    const int q4 = lane >> 4;
    const uint32_t fbase = (uint32_t)(((q4 % KS) * NU + q4 / KS) * 64 + 16 * ((lane >> 1) & 3) + 8 * ((lane >> 3) & 1)) * 16u + (uint32_t)(lane & 1) * 8u;
    constexpr int RB = NR > 3 ? 3 : NR;                    // rows in flight per wave (register budget: 3 x NJ float4 = 60 VGPRs)
#pragma unroll
    for (int i0 = 0; i0 < NR; i0 += RB) {
      float4v v[RB][NJ];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = wave + XW * (i0 + i);
        const char* src = (const char*)a.xf + (size_t)(r0 + (r < R ? r : R - 1)) * (size_t)a.xf_ld * 4;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;          // branch-free, masked at use
          v[i][j] = *(const float4v*)(src + (uint32_t)k * 4u);
        }
      }
      if (i0 == 0) WH_PROBE_AT(a, wgid, 1);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = wave + XW * (i0 + i);
        if (i0 + i < NR && r < RW) {
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
            sum += ((j * 64 + lane) * 4 < K) ? t : 0.f;
          }
          const float mean = wave_sum(sum) * invK;
          float ss = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if ((j * 64 + lane) * 4 < K) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
            }
          }
          const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
          const uint32_t rbase = fbase + (uint32_t)((r >> 3) * FRAG * 16 + (r & 7) * 16);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const bool on = (j * 64 + lane) * 4 < K;
            half4v o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = on ? (half_t)((v[i][j][e] - mean) * rstd) : (half_t)0.f;
            *(half4v*)((char*)xfrag + rbase + (uint32_t)(j * JSTRIDE)) = o4;
          }
        }
      }
    }
    WH_PROBE_AT(a, wgid, 2);
  } else if (PRO == whk::PRO_COMBINE) {
    // ================= merge waves: partials are split-major [S][rows][H][64] fp16, normalised (attention.hip):
    // wave-load i of a split covers 8 consecutive (row, head) pairs x 64 dims; lane = 8 (pair in the load) + dim / 8
    const int H = a.H, npair = RW * H;                    // pairs of this row block
    const int nload = (npair + 7) >> 3;                   // wave-loads per split
    constexpr int NLD = XW ? (20 * NRT + XW - 1) / XW : 1;   // wave-loads per merge wave (H <= 20)
    constexpr int LB = NLD > 5 ? 5 : NLD;                 // wave-loads in flight per merge wave (register budget)
    const size_t split_stride = (size_t)a.R * H;          // pairs per split
#pragma unroll 1
    for (int i0 = 0; i0 < NLD; i0 += LB) {
      half8v po[LB][CSm];
      float2v pml[LB][CSm];
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        int ld = wave + XW * (i0 + i); if (ld > nload - 1) ld = nload - 1;
        int pair = ld * 8 + (lane >> 3); if (pair > npair - 1) pair = npair - 1;
        const int prow = pair / H, ph = pair - prow * H;
        const int grow = r0 + (prow < R ? prow : R - 1);
        const uint32_t pidx = (uint32_t)grow * (uint32_t)H + (uint32_t)ph;
#pragma unroll
        for (int s = 0; s < CSm; ++s) {
          pml[i][s] = *(const float2v*)((const char*)a.part_ml + ((size_t)s * split_stride) * 8 + pidx * 8u);
          po[i][s] = *(const half8v*)((const char*)a.part_o + ((size_t)s * split_stride) * 128 + (pidx * 64u + (uint32_t)(lane & 7) * 8u) * 2u);
        }
      }
      if (i0 == 0) WH_PROBE_AT(a, wgid, 1);
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int ld = wave + XW * (i0 + i);
        const int pair = ld * 8 + (lane >> 3);
        if (i0 + i < NLD && ld < nload && pair < npair) {
          const int prow = pair / H, ph = pair - prow * H;
          float M = pml[i][0][0];
#pragma unroll
          for (int s = 1; s < CSm; ++s) M = fmaxf(M, pml[i][s][0]);
          float w[CSm], den = 0.f;
#pragma unroll
          for (int s = 0; s < CSm; ++s) {
            w[s] = (pml[i][s][0] != WH_NEG_INF) ? __expf(pml[i][s][0] - M) * pml[i][s][1] : 0.f;   // partial o is o_s / l_s
            den += w[s];
          }
          const float inv = 1.0f / den;
          half8v xo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float num = 0.f;
#pragma unroll
            for (int s = 0; s < CSm; ++s) num = __builtin_fmaf(w[s], (float)po[i][s][e], num);
            xo[e] = (half_t)(num * inv);
          }
          // head ph = K block: unit (ph % KS, ph / KS); dims [8 (lane & 7), +8): half = (lane & 7) >> 2, c = lane & 3
          const int dl = lane & 7;
          xfrag[(prow >> 3) * FRAG + ((ph % KS) * NU + ph / KS) * 64 + 16 * (dl & 3) + 8 * (dl >> 2) + (prow & 7)] = xo;
        }
      }
    }
    // K blocks beyond the last head (NU * KS > H) multiply clamped weights: zero them
    for (int blk = H + wave; blk < NU * KS; blk += XW) {
      half8v z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (half_t)0.f;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) xfrag[rt * FRAG + ((blk % KS) * NU + blk / KS) * 64 + lane] = z;
    }
    WH_PROBE_AT(a, wgid, 2);
  }
  ISSUE_FENCE();

  // ---- epilogue operands (L2 hits, needed last; requested behind every wave's own loads): output o = tid + NT j ->
  // (slot, row tile, row, feature)
  constexpr int NOUT = (GS * NRT * 64 + NT - 1) / NT;
  float e_bias[NOUT], e_res[NOUT];
  int e_lag[NOUT];
  int e_pos = 0;
  if (a.epi == whk::EPI_QKV) e_pos = load_agent_int(a.d_pos);
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    const int o = tid + NT * j;
    const int es = o / (NRT * 64), ert = (o >> 6) % NRT, er = (o >> 3) & 7, ej = o & 7;
    const int en = blockIdx.x * fw + es * 8 + ej;
    const bool on = o < GS * NRT * 64 && ert * 8 + er < R && es * 8 + ej < fw && en < a.N;
    e_bias[j] = 0.f; e_res[j] = 0.f; e_lag[j] = 0;
    if (on) {
      const int64_t rr = r0 + ert * 8 + er;
      if (a.epi == whk::EPI_QKV && a.lag) e_lag[j] = a.lag[rr];
      if (a.bias) e_bias[j] = a.bias[en];
      if (a.epi == whk::EPI_RESID) e_res[j] = a.resid[rr * a.resid_ld + en];
    }
  }
  if (!XW) WH_PROBE_AT(a, wgid, 1);

  if constexpr (XW != 0) {
    __syncthreads();
  } else {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!(kw + KS * u < nblk)) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
          for (int e = 0; e < 8; ++e) xb[rt][u][e] = (half_t)0.f;
      }
    }
  }
  WH_PROBE_AT(a, wgid, 3);

  // ---- weight waves: NU MFMAs per row tile, C[m][n] with m = weight row (+8: second half), n = batch row (+8: second half)
  if (!is_x) {
    if (W_ORDERED_OK && a.w_ordered) {
      const int i8 = lane & 7, rd = 8 * i8 + ((4 * ((lane >> 3) & 1) + (lane >> 4)) ^ i8);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        turn[mw][lane] = wa[u];
        __builtin_amdgcn_wave_barrier();
        wa[u] = turn[mw][rd];
        __builtin_amdgcn_wave_barrier();
      }
    }
    const bool diag = (lane >> 5) == ((lane >> 3) & 1);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      float4v acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (XW != 0) {
        half8v xt[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) xt[u] = xfrag[rt * FRAG + (kw * NU + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xt[u], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xb[rt][u], acc, 0, 0, 0);
      }
      // lane holds C[m = 4 (lane >> 4) + e][n = lane & 15]; valid where (m >> 3) == (n >> 3)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = diag ? acc[e] : 0.f;
        z += lane_xor8(z);
        float p, q; lane_swap32(z, p, q);
        acc[e] = p + q;
      }
      if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[mw][rt][4 * (lane >> 4) + e][lane & 7] = acc[e];
      }
    }
  }
  WH_PROBE_AT(a, wgid, 4);
  __syncthreads();
  WH_PROBE_AT(a, wgid, 5);
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    const int o = tid + NT * j;
    const int es = o / (NRT * 64), ert = (o >> 6) % NRT, er = (o >> 3) & 7, ej = o & 7;
    const int en = blockIdx.x * fw + es * 8 + ej;
    const bool on = o < GS * NRT * 64 && ert * 8 + er < R && es * 8 + ej < fw && en < a.N;
    if (on) {
      float v = e_bias[j];
#pragma unroll
      for (int k = 0; k < KS; ++k) v += red[es * KS + k][ert][ej][er];
      const int64_t rr = r0 + ert * 8 + er;
      const int n = en;
      switch (a.epi) {
        case whk::EPI_STORE: ((half_t*)a.y)[a.y_frag ? frag_elem(rr, n, a.N) : rr * a.y_ld + n] = (half_t)v; break;
        case whk::EPI_GELU: ((half_t*)a.y)[a.y_frag ? frag_elem(rr, n, a.N) : rr * a.y_ld + n] = (half_t)gelu_erf(v); break;
        case whk::EPI_F32: ((float*)a.y)[rr * a.y_ld + n] = v; break;
        case whk::EPI_RESID: a.resid[rr * a.resid_ld + n] = e_res[j] + v; break;
        case whk::EPI_QKV: {
          const int D = a.D;
          if (n < D) ((half_t*)a.y)[rr * a.y_ld + n] = (half_t)v;
          else {
            const int64_t pos = e_pos - e_lag[j];
            if (n < 2 * D) ((half_t*)a.kcache)[rr * a.cache_bs + pos * D + (n - D)] = (half_t)v;
            else ((half_t*)a.vcache)[rr * a.cache_bs + pos * D + (n - 2 * D)] = (half_t)v;
          }
        } break;
      }
    }
  }
  WH_PROBE_AT(a, wgid, 6);
  if (a.bump && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump, a.bump_by);
  if (a.bump2 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicAdd(a.bump2, 1);
}

template <int PRO, int GS, int KS, int CSm, int XW, int NRT>
hipError_t launch_gemv8_nrt(const whk::GemvArgs& a, int fw, hipStream_t stream) {
  constexpr int WAVES = GS * KS + XW;
  static_assert(WAVES <= 16, "at most 1024 threads per workgroup");
  if (fw < 1 || fw > 8 * GS) return hipErrorInvalidValue;
  const int nblk = a.K / 64;
  const int nu = (nblk + KS - 1) / KS;
  dim3 grid((a.N + fw - 1) / fw, (a.R + 8 * NRT - 1) / (8 * NRT)), block(WAVES * 64);
  if constexpr (KS == 2) {         // the 3-slot LayerNorm projection of 9+ rows: K <= 1280 as 2 splits x 10 wave-loads
    if (nu > 10) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemv8_kernel<PRO, GS, KS, 10, CSm, XW, NRT>), grid, block, 0, stream, a, fw);
  } else {
    if (nu <= 3) hipLaunchKernelGGL((gemv8_kernel<PRO, GS, KS, 3, CSm, XW, NRT>), grid, block, 0, stream, a, fw);
    else if (nu <= 5) hipLaunchKernelGGL((gemv8_kernel<PRO, GS, KS, 5, CSm, XW, NRT>), grid, block, 0, stream, a, fw);
    else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// rows per workgroup: 8 (the decode step of up to 8 clips), 24 or 48 (beam search / best_of rows, the few-row prefill)
template <int PRO, int GS, int KS, int CSm, int XW>
hipError_t launch_gemv8_cfg(const whk::GemvArgs& a, int fw, hipStream_t stream) {
  if (a.R <= 8) return launch_gemv8_nrt<PRO, GS, KS, CSm, XW, 1>(a, fw, stream);
  if (a.R <= 16) return launch_gemv8_nrt<PRO, GS, KS, CSm, XW, 2>(a, fw, stream);
  // 16-wave workgroups have 128 VGPRs per lane: 3 row tiles of x fragments (60) + the weights (20) fit, 6 would spill
  if constexpr (GS * KS + XW < 16) {
    if (a.R > 24) return launch_gemv8_nrt<PRO, GS, KS, CSm, XW, 6>(a, fw, stream);
  }
  return launch_gemv8_nrt<PRO, GS, KS, CSm, XW, 3>(a, fw, stream);
}

// features per workgroup: spread N over all 256 CUs when that leaves >= 4 features per workgroup, never more than 8 per slot
static int gemv8_fw(int N, int gs) {
  static int ncu = 256;
  int fw = (N + ncu - 1) / ncu;
  if (fw > 8 * gs) fw = 8 * gs;
  if (fw < 4) fw = N < 8 ? N : 8 < 8 * gs ? 8 : 8 * gs;      // small matrices: whole 8-feature groups
  return fw;
}

// shapes (large-v3 in brackets; every choice measured with tools/probe_gemv8, profiles/r02_probe_gemv8.txt):
//   K <= 1280 -> 4 splits of K, K <= 5120 -> 16 (PRO_PLAIN only);
//   D x D [160 workgroups] and 3D x D [2 feature-group slots, 240 workgroups]: 8 prologue waves, one row each;
//   4D x D: 3 slots [214 workgroups of 4 + 12 waves] — a launch costs about
//   1.5 us + (12.6 cycles x HBM lines + 3.7 cycles x L2 lines requested by the busiest CU) / clock.
// feature-group slots per workgroup for (N, prologue)
static int gemv8_slots(int N, int pro) {
  const int ngroups = (N + 7) / 8;
  if (pro == whk::PRO_LN) return ngroups >= 600 ? 3 : ngroups >= 400 ? 2 : 1;
  return 1;
}

template <int PRO, int CSm>
hipError_t launch_gemv8_pro(const whk::GemvArgs& a, hipStream_t stream) {
  const int nblk = a.K / 64, gs = gemv8_slots(a.N, PRO), fw = gemv8_fw(a.N, gs);
  if constexpr (PRO == whk::PRO_PLAIN) {
    if (nblk > 20) {
      whk::GemvArgs b = a;
      b.w_ordered = !WH_DEV_FLAG("WH_GEMV8_W_OPERAND_ORDER");      // developer A/B: 1 = operand-order weight loads as everywhere else
      return launch_gemv8_cfg<PRO, 1, 16, CSm, 0>(b, fw, stream);
    }
    return launch_gemv8_cfg<PRO, 1, 4, CSm, 0>(a, fw, stream);
  } else {
    if (nblk > 20) return hipErrorNotSupported;              // the prologue waves cover K <= 1280
    if constexpr (PRO == whk::PRO_LN) {
      // 3 slots (4D x D of large-v3: 256 workgroups x 20 features).  Up to 8 rows: 12 weight waves + 4 LayerNorm waves of 2 rows.
      // From 9 rows on the 4 LayerNorm waves (6 rows each at 24 rows, two round trips) were the launch's long pole (9280 of its
      // 13 984 cycles, tools/probe_gemv8 24): 6 weight waves of 10 wave-loads + 8 LayerNorm waves of 3 rows, one round trip.
      if (gs == 3 && a.R > 8 && !WH_DEV_FLAG("WH_GEMV8_LN3_KS4")) {
        if (a.R <= 16) return launch_gemv8_nrt<PRO, 3, 2, CSm, 8, 2>(a, fw, stream);
        return launch_gemv8_nrt<PRO, 3, 2, CSm, 8, 3>(a, fw, stream);
      }
      if (gs == 3) return launch_gemv8_cfg<PRO, 3, 4, CSm, 4>(a, fw, stream);
      if (gs == 2) return launch_gemv8_cfg<PRO, 2, 4, CSm, 8>(a, fw, stream);
    }
    return launch_gemv8_cfg<PRO, 1, 4, CSm, 8>(a, fw, stream);
  }
}

static bool gemv8_covers(int N, int K, int pro) {
  if (K % 64 != 0) return false;
  const int nblk = K / 64;
  if (nblk > 20 ? (nblk + 15) / 16 > 5 : (nblk + 3) / 4 > 5) return false;
  if (pro != whk::PRO_PLAIN && nblk > 20) return false;
  if ((int64_t)N * K >= (1ll << 31)) return false;
  return true;
}
bool gemv8_enabled() {
  return !WH_DEV_FLAG("WH_GEMV_DOT2");      // developer switch: the round-1 v_dot2 kernels instead
}

// returns hipErrorNotSupported when the shape / mode is not covered (the caller falls back to the v_dot2 kernels)
hipError_t launch_gemv8(const whk::GemvArgs& a, hipStream_t stream) {
  if (!gemv8_enabled() || a.variant != 0) return hipErrorNotSupported;
  if (a.K % 64 != 0) return hipErrorNotSupported;
  const int nblk = a.K / 64;
  if (nblk > 20 ? (nblk + 15) / 16 > 5 : (nblk + 3) / 4 > 5) return hipErrorNotSupported;
  if (a.epi == whk::EPI_F32 && a.N > 16384) return hipErrorNotSupported;      // logits: the streaming kernel
  if ((int64_t)a.N * a.K >= (1ll << 31)) return hipErrorNotSupported;         // 32-bit lane offsets
  switch (a.pro) {
    case whk::PRO_PLAIN:
      if (a.x_ld % 8 != 0 || (int64_t)a.R * a.x_ld >= (1ll << 30)) return hipErrorNotSupported;
      return launch_gemv8_pro<whk::PRO_PLAIN, 1>(a, stream);
    case whk::PRO_LN:
      if (!a.ln_folded || a.xf_ld % 4 != 0 || (int64_t)a.R * a.xf_ld >= (1ll << 29)) return hipErrorNotSupported;
      return launch_gemv8_pro<whk::PRO_LN, 1>(a, stream);
    case whk::PRO_COMBINE:
      if (a.K != a.H * 64 || a.H > 20 || (int64_t)a.R * a.H * a.splits * 64 >= (1ll << 29)) return hipErrorNotSupported;
      if (a.splits == 2) return launch_gemv8_pro<whk::PRO_COMBINE, 2>(a, stream);
      if (a.splits == 3) return launch_gemv8_pro<whk::PRO_COMBINE, 3>(a, stream);
      if (a.splits == 4) return launch_gemv8_pro<whk::PRO_COMBINE, 4>(a, stream);
      return hipErrorNotSupported;
  }
  return hipErrorNotSupported;
}


// ---------------------------------------------------------------------------------------------------------------
// 17..48 rows (beam search: 8 clips x 5 beams = 40) behind a LayerNorm: ALL rows in one workgroup.
// The 16-row tiles above put three row blocks on grid.y, so every 16 output features are handled by three workgroups
// that each repeat a LayerNorm prologue and re-read the weights: 960 workgroups for FC1, 3.75 rounds per CU,
// 14.9 us.  Here a 16-wave workgroup normalises all 48 rows once (3 per wave, in registers, fp16 into a bank-swizzled
// LDS tile of 48 x K <= 123 KB), then the waves split into NFT feature tiles x 16 / NFT slices of K and run
// 3 MFMAs (row tiles) per weight fragment; partial sums meet in LDS (aliasing the x tile).  One workgroup per 16 or 32
// features: a single round on 256 CUs.  LayerNorm affine parameters must be folded into W (WH_WEIGHTS_DEC_LN_FOLDED).
// ---------------------------------------------------------------------------------------------------------------
template <int NFT>      // feature tiles of 16 per workgroup (1 or 2)
__global__ __launch_bounds__(1024) void gemv_rows48_kernel(whk::GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pin_kernargs(a);
  constexpr int WAVES = 16, RT = 48, KSP = WAVES / NFT, J = 5, NU = (40 + KSP - 1) / KSP;   // K <= 1280: 40 steps of 32
  const int K = a.K, nks = K / 32;
  half_t* xs = (half_t*)smem;                                   // [48][K], 16-byte unit u of row r stored at u ^ (r & 15)
  float* red = (float*)smem;                                    // [KSP][NFT][3][16 rows][16 features], after the MFMAs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ft = wave % NFT, ksp = wave / NFT;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int n0 = blockIdx.x * (NFT * 16);
  const int R = a.R;

  // ---- requests: the three fp32 rows of this wave first (L2), then its weight fragments (HBM)
  float4v v[3][J];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = wave + WAVES * i;
    const float* src = a.xf + (int64_t)(r < R ? r : 0) * a.xf_ld;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;           // branch-free; masked at use
      v[i][j] = *(const float4v*)(src + k);
    }
  }
  ISSUE_FENCE();
  half8v w[NU];
  {
    int n = n0 + ft * 16 + i16; if (n > a.N - 1) n = a.N - 1;
    const half_t* base = (const half_t*)a.W + (int64_t)n * K + g4 * 8;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int st = ksp + KSP * u; if (st > nks - 1) st = nks - 1;          // clamped: skipped below
      w[u] = __builtin_nontemporal_load((const half8v*)(base + st * 32));
    }
  }
  ISSUE_FENCE();
  int e_pos = 0;
  if (a.epi == whk::EPI_QKV) e_pos = load_agent_int(a.d_pos);
  float e_bias[(NFT * 768 + 1023) / 1024];           // the bias of this thread's outputs, requested now
#pragma unroll
  for (int q = 0; q < (NFT * 768 + 1023) / 1024; ++q) {
    const int o = tid + q * 1024;
    int n = n0 + ((o >> 8) / 3) * 16 + (o & 15); if (n > a.N - 1) n = a.N - 1;
    e_bias[q] = a.bias ? a.bias[n] : 0.f;
  }

  // ---- LayerNorm (two-pass, a wave owns whole rows), fp16 into the swizzled tile
  {
    const float invK = 1.0f / (float)K;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = wave + WAVES * i;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
        sum += ((j * 64 + lane) * 4 < K) ? t : 0.f;
      }
      const float mean = wave_sum(sum) * invK;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if ((j * 64 + lane) * 4 < K) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
        }
      }
      const float rstd = r < R ? rsqrtf(wave_sum(ss) * invK + 1e-5f) : 0.f;     // rows beyond R become zeros
      half_t* xr = xs + (size_t)r * K;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int k = (j * 64 + lane) * 4;
        if (k < K)
          Pack4<half_t>::store(xr + ((((k >> 3) ^ (r & 15)) << 3) | (k & 7)), (v[i][j][0] - mean) * rstd,
                               (v[i][j][1] - mean) * rstd, (v[i][j][2] - mean) * rstd, (v[i][j][3] - mean) * rstd);
      }
    }
  }
  __syncthreads();

  // ---- out[feature][row] += W fragment . x fragments of the three row tiles
  float4v acc[3];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) acc[rt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int st = ksp + KSP * u;
    if (st < nks) {
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        const half8v xf = *(const half8v*)(xs + (size_t)(rt * 16 + i16) * K + (((st * 4 + g4) ^ i16) << 3));
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], xf, acc[rt], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                              // every wave is done with the x tile: `red` takes its place
  // lane (row j = i16, feature group g4) holds features 4 g4 + e of row j
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) *(float4v*)(red + ((((ksp * NFT + ft) * 3 + rt) * 16 + i16) * 16 + g4 * 4)) = acc[rt];
  __syncthreads();

  // ---- epilogue: NFT x 3 x 16 x 16 outputs over 1024 threads
#pragma unroll
  for (int q = 0; q < (NFT * 768 + 1023) / 1024; ++q) {
    const int o = tid + q * 1024;
    if (o >= NFT * 768) continue;
    const int f = o & 15, j = (o >> 4) & 15, rt = (o >> 8) % 3, oft = (o >> 8) / 3;
    const int row = rt * 16 + j, n = n0 + oft * 16 + f;
    if (row >= R || n >= a.N) continue;
    float val = e_bias[q];
#pragma unroll
    for (int k = 0; k < KSP; ++k) val += red[((((k * NFT + oft) * 3 + rt) * 16 + j) * 16) + f];
    const int64_t rr = row;
    switch (a.epi) {
      case whk::EPI_STORE: ((half_t*)a.y)[rr * a.y_ld + n] = (half_t)val; break;
      case whk::EPI_GELU: ((half_t*)a.y)[rr * a.y_ld + n] = (half_t)gelu_erf(val); break;
      case whk::EPI_QKV: {
        const int D = a.D;
        if (n < D) ((half_t*)a.y)[rr * a.y_ld + n] = (half_t)val;
        else {
          const int64_t pos = e_pos - (a.lag ? a.lag[row] : 0);
          if (n < 2 * D) ((half_t*)a.kcache)[rr * a.cache_bs + pos * D + (n - D)] = (half_t)val;
          else ((half_t*)a.vcache)[rr * a.cache_bs + pos * D + (n - 2 * D)] = (half_t)val;
        }
      } break;
      default: break;
    }
  }
  if (a.bump && blockIdx.x == 0 && tid == 0) atomicAdd(a.bump, a.bump_by);
  if (a.bump2 && blockIdx.x == 0 && tid == 0) atomicAdd(a.bump2, 1);
}

// applies to: fp16, LayerNorm prologue with folded affine part, 17..48 rows, K a multiple of 256 up to 1280,
// store / GELU / QKV epilogues
bool rows48_applies(const whk::GemvArgs& a) {
  const bool off = WH_DEV_FLAG("WH_NO_ROWS48");   // developer A/B switch
  // measured at 40 rows, large-v3 (rocprof, real beam step): FC1 14.9 -> 11.4 us, QKV 11.7 -> 11.4 us; the D x D
  // cross-attention query got slower (7.9 -> 8.6 us: 80 workgroups each normalising all 48 rows), so N >= 2048 only
  return !off && a.R > 16 && a.R <= 48 && a.pro == whk::PRO_LN && a.ln_folded && a.K % 256 == 0 && a.K <= 1280 &&
         a.N >= 2048 && a.variant <= 0 && (a.epi == whk::EPI_STORE || a.epi == whk::EPI_GELU || a.epi == whk::EPI_QKV);
}

hipError_t launch_rows48(const whk::GemvArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)48 * a.K * 2 > 49152 ? (size_t)48 * a.K * 2 : 49152;     // x tile, later the partial sums
  static whk::LdsAttr attr2;
  { hipError_t e = whk::raise_dynamic_lds(attr2, (const void*)gemv_rows48_kernel<2>, 160 * 1024); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL((gemv_rows48_kernel<2>), dim3((a.N + 31) / 32), dim3(1024), lds, stream, a);   // 32 features per workgroup
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// The tied logits projection (V x D, 133 MB at large-v3) for 17..48 rows.  The 16-row tiles put 3 row blocks on
// grid.y and each streams the whole matrix: 106 us per launch at 40 rows.  Here the 48-row x tile is normalised once
// per workgroup (decoder.ln is never folded: gamma / beta go through LDS) and every WAVE then owns whole 16-feature
// tiles: it walks all of K itself in two batches of 20 weight fragments, 3 MFMAs (row tiles) per fragment, and stores
// its fp32 logits straight from the accumulators — no cross-wave reduction, the matrix is read once.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gemv_rows48_stream_kernel(whk::GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pin_kernargs(a);
  constexpr int WAVES = 16, J = 5, NB = 20;                     // K <= 1280: 40 steps of 32 in two batches
  const int K = a.K, nks = K / 32;
  half_t* xs = (half_t*)smem;                                   // [48][K], 16-byte unit u of row r stored at u ^ (r & 15)
  float* gb = (float*)(smem + (size_t)48 * K * 2);              // gamma [K], beta [K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  const int R = a.R;
  const int ntiles = (a.N + 15) / 16;

  float4v v[3][J];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = wave + WAVES * i;
    const float* src = a.xf + (int64_t)(r < R ? r : 0) * a.xf_ld;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;           // branch-free; masked at use
      v[i][j] = *(const float4v*)(src + k);
    }
  }
  for (int k = tid * 4; k < K; k += 4096) {
    *(float4v*)(gb + k) = *(const float4v*)(a.ln_w + k);
    *(float4v*)(gb + K + k) = *(const float4v*)(a.ln_b + k);
  }
  __syncthreads();
  {
    const float invK = 1.0f / (float)K;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = wave + WAVES * i;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
        sum += ((j * 64 + lane) * 4 < K) ? t : 0.f;
      }
      const float mean = wave_sum(sum) * invK;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if ((j * 64 + lane) * 4 < K) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
        }
      }
      const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
      half_t* xr = xs + (size_t)r * K;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int k = (j * 64 + lane) * 4;
        if (k < K) {
          const float4v w4 = *(const float4v*)(gb + k), b4 = *(const float4v*)(gb + K + k);
          half_t* dst = xr + ((((k >> 3) ^ (r & 15)) << 3) | (k & 7));
          if (r < R)
            Pack4<half_t>::store(dst, (v[i][j][0] - mean) * rstd * w4[0] + b4[0], (v[i][j][1] - mean) * rstd * w4[1] + b4[1],
                                 (v[i][j][2] - mean) * rstd * w4[2] + b4[2], (v[i][j][3] - mean) * rstd * w4[3] + b4[3]);
          else
            Pack4<half_t>::store(dst, 0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
  __syncthreads();

  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    int n = tile * 16 + i16; if (n > a.N - 1) n = a.N - 1;
    const half_t* base = (const half_t*)a.W + (int64_t)n * K + g4 * 8;
    float4v acc[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) acc[rt] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int st0 = 0; st0 < nks; st0 += NB) {
      half8v w[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        int st = st0 + u; if (st > nks - 1) st = nks - 1;
        w[u] = __builtin_nontemporal_load((const half8v*)(base + st * 32));
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int st = st0 + u;
        if (st < nks) {
#pragma unroll
          for (int rt = 0; rt < 3; ++rt) {
            const half8v xf = *(const half8v*)(xs + (size_t)(rt * 16 + i16) * K + (((st * 4 + g4) ^ i16) << 3));
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], xf, acc[rt], 0, 0, 0);
          }
        }
      }
    }
    // lane (row j = i16, feature group g4) holds features 16 tile + 4 g4 + e of rows 16 rt + j
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      const int row = rt * 16 + i16;
      if (row < R) {
        float* y = (float*)a.y + (int64_t)row * a.y_ld + tile * 16 + g4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (tile * 16 + g4 * 4 + e < a.N) y[e] = acc[rt][e];
      }
    }
  }
  if (a.bump && blockIdx.x == 0 && tid == 0) atomicAdd(a.bump, a.bump_by);
  if (a.bump2 && blockIdx.x == 0 && tid == 0) atomicAdd(a.bump2, 1);
}

// fp16, LayerNorm prologue, fp32 output without bias, 17..48 rows, K a multiple of 256 up to 1280, a long N
bool rows48_stream_applies(const whk::GemvArgs& a) {
  const bool off = WH_DEV_FLAG("WH_NO_ROWS48");   // developer A/B switch
  return !off && a.R > 16 && a.R <= 48 && a.pro == whk::PRO_LN && a.epi == whk::EPI_F32 && !a.bias && a.K % 256 == 0 &&
         a.K <= 1280 && a.N >= 16384 && a.variant <= 0;
}

hipError_t launch_rows48_stream(const whk::GemvArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)48 * a.K * 2 + (size_t)2 * a.K * 4;
  static whk::LdsAttr attr;
  { hipError_t e = whk::raise_dynamic_lds(attr, (const void*)gemv_rows48_stream_kernel, 160 * 1024); if (e != hipSuccess) return e; }
  const int ntiles = (a.N + 15) / 16;
  int wgs = (ntiles + 15) / 16;                        // one 16-feature tile per wave ...
  if (wgs > 256) wgs = 256;                            // ... or several when there are more tiles than 256 x 16 waves
  hipLaunchKernelGGL(gemv_rows48_stream_kernel, dim3(wgs), dim3(1024), lds, stream, a);
  return hipGetLastError();
}


// The merge of PRO_COMBINE as its own launch, for 17+ rows: fused into the projection every 16-row workgroup merges
// its rows x 20 heads x S splits again (480 workgroups, 59 MB of L2 reads at 40 rows: 11.3 us against 4.9 us for the
// same matrix on merged input); here every (row, head, 4 head dims) is merged once and the projection runs PRO_PLAIN.
// Same operations in the same order as the fused fast path above (S <= 4): the fp16 values are identical.
template <typename T, int MS>      // MS: splits held in registers (>= S)
__global__ __launch_bounds__(256) void merge_partials_kernel(const T* __restrict__ part_o, const float* __restrict__ part_ml,
                                                             int S, int R, int H, T* __restrict__ out, int64_t o_ld, int o_frag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * H * 16) return;
  const int r = i / (H * 16), rem = i - r * (H * 16);
  const int h = rem >> 4, d4 = rem & 15;
  const int64_t pb = (int64_t)r * H + h, ps = (int64_t)R * H;          // [S][rows][H]
  float2v ml[MS];
  float4v o[MS];
#pragma unroll
  for (int s = 0; s < MS; ++s) {
    const int sc = s < S ? s : S - 1;                                   // branch-free loads, clamped
    ml[s] = *(const float2v*)(part_ml + (pb + sc * ps) * 2);
    o[s] = load_part4(part_o + (pb + sc * ps) * 64 + d4 * 4);
  }
  float M = WH_NEG_INF;
#pragma unroll
  for (int s = 0; s < MS; ++s) { if (s >= S) ml[s] = float2v{WH_NEG_INF, 0.f}; M = fmaxf(M, ml[s][0]); }
  float w[MS], den = 0.f;
#pragma unroll
  for (int s = 0; s < MS; ++s) {
    w[s] = (ml[s][0] == WH_NEG_INF) ? 0.f : __expf(ml[s][0] - M);
    den = __builtin_fmaf(w[s], ml[s][1], den);
  }
  const float inv = 1.0f / den;
  float4v num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < MS; ++s) {
    const float f = w[s] * ml[s][1] * inv;
    num[0] = __builtin_fmaf(f, o[s][0], num[0]); num[1] = __builtin_fmaf(f, o[s][1], num[1]);
    num[2] = __builtin_fmaf(f, o[s][2], num[2]); num[3] = __builtin_fmaf(f, o[s][3], num[3]);
  }
  // o_frag: fragment order for the projection that follows (kernels.h); the 4 values stay inside one 16-byte piece
  Pack4<T>::store(out + (o_frag ? frag_elem(r, h * 64 + d4 * 4, H * 64) : (int64_t)r * o_ld + h * 64 + d4 * 4), num[0], num[1], num[2], num[3]);
}

}  // namespace

namespace whk {

hipError_t launch_gemv(const GemvArgs& a, int dtype, hipStream_t stream) {
  if (a.R <= 0) return hipErrorInvalidValue;
  if (a.x_frag || a.y_frag) {
    // only gemv8_kernel knows the fragment order (kernels.h): a caller that asks for it where that kernel does not run must hear
    // about it (api.cpp::step_plan asks gemv8_will_run first) — never row-major arithmetic on fragment-order bytes
    if (dtype != 1 || a.R > WH_GEMV8_MAX_ROWS) return hipErrorInvalidValue;
    const hipError_t e = launch_gemv8(a, stream);
    return e == hipErrorNotSupported ? hipErrorInvalidValue : e;
  }
  if (dtype == 1) {
    // the decode step (and the few-row prefill, and beam-search rows): MFMA diagonal form
    // More rows (beam search: 8 clips x 5 beams = 40): every workgroup reads ALL x rows, so beyond 8 rows the x
    // fragments (40 rows x K) outweigh its share of the weights; measured at 40 rows, large-v3 (profiles/r02_beam_*):
    // the 24 / 48-row forms of gemv8 take 6.5 / 8.6 / 14.3 / 11.4 / 14.9 / 18.4 us (out, cq, cout, qkv, fc1, fc2) against
    // 4.8 / 7.9 / 11.2 / 14.9 / 11.7 / 11.3 us for the 16-row LDS-staged MFMA tiles below — those keep the job wherever
    // they apply; gemv8's row blocks serve the remaining shapes (row counts 9..96 outside the 16-row form's limits).
    if (a.R > WH_GEMV8_MAX_ROWS || a.epi == whk::EPI_F32) {
      if (rows48_applies(a)) return launch_rows48(a, stream);
      if (rows48_stream_applies(a)) return launch_rows48_stream(a, stream);
    }
    bool rows16 = a.R > 8 && a.variant <= 0 && a.K % 128 == 0 && a.K <= 5120 && (a.K / 32 <= 40 || a.pro == whk::PRO_PLAIN);
    // Round 5: 9 - 24 rows run on gemv8_kernel with TWO / THREE row tiles per weight fragment (the weights are streamed once,
    // every weight wave holds that many sets of x fragments) instead of the 16-row LDS-staged tiles / the 48-row LayerNorm
    // kernels: the whole step of large-v3 at 16 rows 2680 -> 2073 us, 12 rows 2472 -> 1855, 2 x 5 beam rows 2276 -> 1646,
    // 3 x 5 2355 -> 1740, 4 x 5 2228 -> 1949, 24 rows 2862 -> 2609.  Beyond 24 rows it loses: 4 / 5 / 6 tiles at 32 / 40 rows
    // 3526 / 3030 vs 3207 / 2571 us (every weight wave fetches all tiles' x fragments itself, the LayerNorm waves take 5 rows
    // each), so 25 - 48 rows keep the LDS-staged forms (profiles/r05_rows24.txt).  WH_GEMV8_MAX_ROWS is the knob of that A/B.
    if (a.R <= WH_GEMV8_MAX_ROWS) rows16 = false;
    if (a.R <= 96 && !rows16) {
      const hipError_t e = launch_gemv8(a, stream);
      if (e != hipErrorNotSupported) return e;
    }
    if (a.R <= 4) return launch_rt<half_t, 4>(a, stream);
    // beam-search row counts: row tiles of 16 through the matrix cores while x (16 rows) fits in LDS
    if (a.R > 8 && a.variant <= 0 && a.K % 128 == 0 && a.K <= 5120 && (a.K / 32 <= 40 || a.pro == whk::PRO_PLAIN))
      return launch_rows16_mf<half_t>(a, stream);
    return launch_rt<half_t, 8>(a, stream);           // R <= 8, or long K: row tiles of 8 on grid.y
  }
  // fp32 (strict-parity mode): x rows are twice as wide in LDS, and the 4-wave staging moves at most 24 units per
  // thread per row (K <= 3072 at 8 rows, K <= 6144 at 4 rows)
  if (a.R <= 4 || (size_t)a.K * 8 * sizeof(float) > 128 * 1024 || a.K > 3072) return launch_rt<float, 4>(a, stream);
  return launch_rt<float, 8>(a, stream);
}

// true when launch_gemv hands (R rows, fp16) x (N, K, pro) to gemv8_kernel — the only kernel that reads x_frag and writes y_frag
bool gemv8_will_run(int R, int N, int K, int pro) {
  return gemv8_enabled() && R >= 1 && R <= WH_GEMV8_MAX_ROWS && gemv8_covers(N, K, pro);
}

hipError_t launch_merge_partials(const void* part_o, const float* part_ml, int splits, int R, int H, void* out,
                                 int64_t o_ld, int dtype, hipStream_t stream, int o_frag) {
  if (splits < 1 || splits > DEC_ATTN_MAX_SPLITS || R <= 0 || H <= 0) return hipErrorInvalidValue;
  const int n = R * H * 16;
  const dim3 grid((n + 255) / 256), block(256);
  if (dtype == 1) {
    if (splits <= 4)
      hipLaunchKernelGGL((merge_partials_kernel<half_t, 4>), grid, block, 0, stream, (const half_t*)part_o, part_ml, splits, R, H, (half_t*)out, o_ld, o_frag);
    else
      hipLaunchKernelGGL((merge_partials_kernel<half_t, DEC_ATTN_MAX_SPLITS>), grid, block, 0, stream, (const half_t*)part_o, part_ml, splits, R, H, (half_t*)out, o_ld, o_frag);
  } else {
    if (splits <= 4)
      hipLaunchKernelGGL((merge_partials_kernel<float, 4>), grid, block, 0, stream, (const float*)part_o, part_ml, splits, R, H, (float*)out, o_ld, o_frag);
    else
      hipLaunchKernelGGL((merge_partials_kernel<float, DEC_ATTN_MAX_SPLITS>), grid, block, 0, stream, (const float*)part_o, part_ml, splits, R, H, (float*)out, o_ld, o_frag);
  }
  return hipGetLastError();
}

}  // namespace whk
