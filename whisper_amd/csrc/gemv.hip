// gemv.hip — the decode-step projection: y[r][n] = sum_k x[r][k] * W[n][k] (+ bias, + fused epilogue)
// for a handful of rows r (batch x beam <= a few dozen).  This is `Linear.forward`
// (whisper/model.py:44-50) at T = 1, i.e. TextDecoder.forward (model.py:227-249) inside
// DecodingTask._main_loop (whisper/decoding.py:686-687) — the HBM-bound part of the path: every
// decoder weight is streamed from HBM exactly once per step for the whole batch.  No MFMA here: the
// arithmetic intensity is R FLOP/byte; the job is to keep >= 10 KB of 16-byte loads in flight per CU.
//
// Workgroup = 256 threads (4 waves) owns 8 consecutive output features.  A wave instruction loads
// 8 rows x 128 contiguous bytes of W (8 lanes x 16 B per row: whole cache lines, fully coalesced); the
// four waves split K; five such loads are issued back-to-back before any is consumed.  x (after the
// fused prologue) sits in LDS in the element type and is read with broadcast ds_read_b128; products are
// v_dot2_f32_f16 (fp32 accumulate).  Partial sums are reduced over the 8 lanes of a row with
// xor-shuffles, over the 4 waves through LDS, and the epilogue is applied by 8*R threads.
//
// Fused prologues: plain copy | LayerNorm of the fp32 residual stream (model.py:39-41) |
//                  merge of split-K cross-attention partials (attention.hip, attn_decode).
// Fused epilogues: store | q + in-place KV-cache append (replaces torch.cat, model.py:332) |
//                  residual add | exact GELU | fp32 logits.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NB = 8;          // output features per workgroup
constexpr int UNROLL = 5;      // weight loads in flight per lane
constexpr int XS_BUDGET = 96 * 1024;

template <typename T, int RT>
__global__ __launch_bounds__(256) void gemv_kernel(whk::GemvArgs a, int kseg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  T* xs = (T*)smem;                                  // [RT][kseg]
  float* red = (float*)(smem + (size_t)RT * kseg * sizeof(T));   // [4][NB][RT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NB;
  const int r0 = blockIdx.y * RT;
  int R = a.R - r0; if (R > RT) R = RT;
  const int K = a.K;

  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;

  int wrow = n0 + (lane >> 3); if (wrow > a.N - 1) wrow = a.N - 1;
  const T* wp = (const T*)a.W + (int64_t)wrow * K + (lane & 7) * UNIT;

  for (int ks = 0; ks < K; ks += kseg) {
    const int klen = (K - ks) < kseg ? (K - ks) : kseg;
    if (ks > 0) __syncthreads();
    // ------------------------------------------------------------------ prologue -> xs
    if (a.pro == whk::PRO_PLAIN) {
      const int upr = klen / UNIT;
      for (int i = tid; i < RT * upr; i += 256) {
        const int r = i / upr, u = i - r * upr;
        unit_t v;
        if (r < R) v = *(const unit_t*)((const T*)a.x + (int64_t)(r0 + r) * a.x_ld + ks + u * UNIT);
        else {
#pragma unroll
          for (int e = 0; e < UNIT; ++e) v[e] = 0;
        }
        *(unit_t*)(xs + (size_t)r * kseg + u * UNIT) = v;
      }
    } else if (a.pro == whk::PRO_LN) {
      // one wave per row: fp32 statistics exactly as layer_norm (mean, biased variance, eps 1e-5)
      for (int r = wave; r < RT; r += 4) {
        T* xr = xs + (size_t)r * kseg;
        if (r >= R) {
          for (int k = lane; k < K; k += 64) xr[k] = from_f32<T>(0.f);
          continue;
        }
        const float* src = a.xf + (int64_t)(r0 + r) * a.xf_ld;
        float s = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
          const float4v t = *(const float4v*)(src + k);
          s += (t[0] + t[1]) + (t[2] + t[3]);
        }
        s = wave_sum(s);
        const float mean = s / (float)K;
        float ss = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
          const float4v t = *(const float4v*)(src + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = t[e] - mean; ss = __builtin_fmaf(d, d, ss); }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)K + 1e-5f);
        for (int k = lane * 4; k < K; k += 256) {
          const float4v t = *(const float4v*)(src + k);
          const float4v w4 = *(const float4v*)(a.ln_w + k);
          const float4v b4 = *(const float4v*)(a.ln_b + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[k + e] = from_f32<T>((t[e] - mean) * rstd * w4[e] + b4[e]);
        }
      }
    } else {  // PRO_COMBINE: merge split-K attention partials (m, l, o[64]) per (row, head)
      const int S = a.splits;
      for (int i = tid; i < RT * K; i += 256) {
        const int r = i / K, c = i - r * K;
        float v = 0.f;
        if (r < R) {
          const int h = c >> 6, d = c & 63;
          const int64_t pb = ((int64_t)(r0 + r) * a.H + h) * S;
          float M = WH_NEG_INF;
          for (int s = 0; s < S; ++s) M = fmaxf(M, a.part_ml[(pb + s) * 2]);
          float num = 0.f, den = 0.f;
          for (int s = 0; s < S; ++s) {
            const float w = __expf(a.part_ml[(pb + s) * 2] - M);
            num = __builtin_fmaf(w, a.part_o[(pb + s) * 64 + d], num);
            den = __builtin_fmaf(w, a.part_ml[(pb + s) * 2 + 1], den);
          }
          v = num / den;
        }
        xs[(size_t)r * kseg + c] = from_f32<T>(v);
      }
    }
    __syncthreads();

    // ------------------------------------------------------------------ stream W
    const int nub = klen / (8 * UNIT);            // blocks of 8 units (128 bytes per row)
    const T* wseg = wp + ks;
    for (int ub0 = wave; ub0 < nub; ub0 += 4 * UNROLL) {
      unit_t w[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int ub = ub0 + 4 * u;
        if (ub < nub) w[u] = __builtin_nontemporal_load((const unit_t*)(wseg + (size_t)ub * 8 * UNIT));
        else {
#pragma unroll
          for (int e = 0; e < UNIT; ++e) w[u][e] = 0;
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int ub = ub0 + 4 * u;
        if (ub < nub) {
          const T* xb = xs + (size_t)(ub * 8 + (lane & 7)) * UNIT;
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const unit_t xu = *(const unit_t*)(xb + (size_t)r * kseg);
            acc[r] = dot_unit(w[u], xu, acc[r]);
          }
        }
      }
    }
  }

  // ---------------------------------------------------------------------- reduce
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    float v = acc[r];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    acc[r] = v;
  }
  if ((lane & 7) == 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r) red[(wave * NB + (lane >> 3)) * RT + r] = acc[r];
  }
  __syncthreads();

  // ---------------------------------------------------------------------- epilogue
  for (int t = tid; t < NB * R; t += 256) {
    const int j = t & (NB - 1), r = t >> 3;
    const int n = n0 + j;
    if (n >= a.N) continue;
    float v = red[(0 * NB + j) * RT + r] + red[(1 * NB + j) * RT + r] + red[(2 * NB + j) * RT + r] +
              red[(3 * NB + j) * RT + r];
    if (a.bias) v += a.bias[n];
    const int64_t rr = r0 + r;
    switch (a.epi) {
      case whk::EPI_STORE: ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(v); break;
      case whk::EPI_GELU: ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(gelu_erf(v)); break;
      case whk::EPI_F32: ((float*)a.y)[rr * a.y_ld + n] = v; break;
      case whk::EPI_RESID: a.resid[rr * a.resid_ld + n] += v; break;
      case whk::EPI_QKV: {
        const int D = a.D;
        if (n < D) ((T*)a.y)[rr * a.y_ld + n] = from_f32<T>(v);
        else {
          const int64_t pos = *a.d_pos;
          if (n < 2 * D) ((T*)a.kcache)[rr * a.cache_bs + pos * D + (n - D)] = from_f32<T>(v);
          else ((T*)a.vcache)[rr * a.cache_bs + pos * D + (n - 2 * D)] = from_f32<T>(v);
        }
      } break;
    }
  }
}

template <typename T, int RT>
hipError_t launch_rt(const whk::GemvArgs& a, hipStream_t stream) {
  constexpr int UNIT = ET<T>::UNIT;
  const int quantum = 32 * UNIT;                       // 4 waves x 8 units
  if (a.K % (8 * UNIT) != 0) return hipErrorInvalidValue;
  int kseg = XS_BUDGET / (RT * (int)sizeof(T));
  kseg = (kseg / quantum) * quantum;
  if (kseg >= a.K) kseg = a.K;
  else if (a.pro != whk::PRO_PLAIN) return hipErrorInvalidValue;   // LN / combine need the whole row
  const size_t lds = (size_t)RT * kseg * sizeof(T) + 4 * NB * RT * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemv_kernel<T, RT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, XS_BUDGET + 8192);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid((a.N + NB - 1) / NB, (a.R + RT - 1) / RT), block(256);
  hipLaunchKernelGGL((gemv_kernel<T, RT>), grid, block, lds, stream, a, kseg);
  return hipGetLastError();
}

}  // namespace

namespace whk {

hipError_t launch_gemv(const GemvArgs& a, int dtype, hipStream_t stream) {
  if (a.R <= 0) return hipErrorInvalidValue;
  if (dtype == 1) {
    if (a.R <= 4) return launch_rt<half_t, 4>(a, stream);
    if (a.R <= 8) return launch_rt<half_t, 8>(a, stream);
    if (a.R <= 16) return launch_rt<half_t, 16>(a, stream);
    return launch_rt<half_t, 32>(a, stream);
  }
  if (a.R <= 4) return launch_rt<float, 4>(a, stream);
  if (a.R <= 8) return launch_rt<float, 8>(a, stream);
  return launch_rt<float, 16>(a, stream);
}

}  // namespace whk
