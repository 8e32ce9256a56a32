// timing.hip — word-timestamp kernels.
//   median filter : whisper/timing.py:19-54 (reflect pad + sliding median along the last axis); the
//                   reference's GPU version is the Triton bubble-sort kernel whisper/triton_ops.py:43-117.
//                   Here: one thread per output element, the window lives in registers, the median is
//                   found with (w/2+1) bubble passes — exact, order statistics only, no arithmetic.
//   dtw           : whisper/timing.py:82-105 (dtw_cpu) semantics — including its tie rule, which differs
//                   from the Triton kernel's (triton_ops.py:38-40) — evaluated as an anti-diagonal
//                   wavefront by one workgroup: three rolling diagonals of the fp32 cost matrix live in
//                   LDS, only the int8 trace goes to HBM.  fp32 adds of fp32 inputs are bit-identical to
//                   dtw_cpu's float64-add-then-store-float32.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <int W>
__global__ void median_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int n) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * n) return;
  const int64_t row = gid / n;
  const int i = (int)(gid - row * n);
  const float* xr = x + row * n;
  float v[W];
#pragma unroll
  for (int j = 0; j < W; ++j) v[j] = xr[reflect_idx(i - W / 2 + j, n)];
#pragma unroll
  for (int p = 0; p < W / 2 + 1; ++p) {
#pragma unroll
    for (int j = 0; j < W - 1 - p; ++j) {
      const float lo = fminf(v[j], v[j + 1]), hi = fmaxf(v[j], v[j + 1]);
      v[j] = lo; v[j + 1] = hi;
    }
  }
  out[gid] = v[W / 2];
}

// generic odd width <= 63 (window in scratch memory; correctness path only)
__global__ void median_generic_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                      int n, int W) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * n) return;
  const int64_t row = gid / n;
  const int i = (int)(gid - row * n);
  const float* xr = x + row * n;
  float v[63];
  for (int j = 0; j < W; ++j) v[j] = xr[reflect_idx(i - W / 2 + j, n)];
  for (int p = 0; p < W / 2 + 1; ++p)
    for (int j = 0; j < W - 1 - p; ++j) {
      const float lo = fminf(v[j], v[j + 1]), hi = fmaxf(v[j], v[j + 1]);
      v[j] = lo; v[j + 1] = hi;
    }
  out[gid] = v[W / 2];
}

__global__ void copy_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) out[gid] = x[gid];
}

// one workgroup; thread t owns rows i = t+1, t+1+blockDim, ... (i in 1..N)
__global__ __launch_bounds__(1024) void dtw_kernel(const float* __restrict__ x, int N, int M,
                                                   int8_t* __restrict__ trace) {
  extern __shared__ float diag[];          // 3 x (N+1)
  float* d[3] = {diag, diag + (N + 1), diag + 2 * (N + 1)};
  const int tid = threadIdx.x, nt = blockDim.x;
  // trace borders as dtw_cpu leaves them (-1) are overwritten by backtrace (timing.py:61-62); store codes
  // 2 on row 0 and 1 on column 0 directly.
  for (int j = tid + 1; j <= M; j += nt) trace[j] = 2;
  for (int i = tid; i <= N; i += nt) trace[(int64_t)i * (M + 1)] = 1;
  // diagonal k holds cost[i][k-i]; k = 0: cost[0][0] = 0, k = 1: cost[0][1] = cost[1][0] = inf
  for (int i = tid; i <= N; i += nt) {
    d[0][i] = (i == 0) ? 0.f : __builtin_huge_valf();
    d[1][i] = __builtin_huge_valf();
  }
  __syncthreads();
  for (int k = 2; k <= N + M; ++k) {
    float* d2 = d[(k - 2) % 3];
    float* d1 = d[(k - 1) % 3];
    float* d0 = d[k % 3];
    for (int i = tid + 1; i <= N; i += nt) {
      const int j = k - i;
      float c = __builtin_huge_valf();
      if (j >= 1 && j <= M) {
        const float c0 = d2[i - 1], c1 = d1[i - 1], c2 = d1[i];
        float cm; int8_t t;
        if (c0 < c1 && c0 < c2) { cm = c0; t = 0; }
        else if (c1 < c0 && c1 < c2) { cm = c1; t = 1; }
        else { cm = c2; t = 2; }
        c = x[(int64_t)(i - 1) * M + (j - 1)] + cm;
        trace[(int64_t)i * (M + 1) + j] = t;
      }
      d0[i] = c;
    }
    if (tid == 0) d0[0] = __builtin_huge_valf();
    __syncthreads();
  }
}

// ---- find_alignment post-processing (whisper/timing.py:207-216) ---------------------------------
// qk [H][T][Tk] -> w [H][T][F]: softmax over the first F frames of (qk * qk_scale)   (timing.py:208-209)
__global__ __launch_bounds__(256) void align_softmax_kernel(const float* __restrict__ qk, int T, int Tk, int F,
                                                            float qk_scale, float* __restrict__ w) {
  __shared__ float red[4];
  const int t = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = qk + ((int64_t)h * T + t) * Tk;
  float* dst = w + ((int64_t)h * T + t) * F;
  float m = WH_NEG_INF;
  for (int j = tid; j < F; j += 256) m = fmaxf(m, src[j] * qk_scale);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = tid; j < F; j += 256) {
    const float e = expf(src[j] * qk_scale - m);
    dst[j] = e;
    s += e;
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = tid; j < F; j += 256) dst[j] = dst[j] / s;
}

// z-normalise every (head, frame) column over the token axis: std_mean(dim=-2, unbiased=False), timing.py:210-211
__global__ void align_znorm_kernel(float* __restrict__ w, int T, int F) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y;
  if (j >= F) return;
  float* col = w + (int64_t)h * T * F + j;
  float mean = 0.f;
  for (int t = 0; t < T; ++t) mean += col[(int64_t)t * F];
  mean /= (float)T;
  float var = 0.f;
  for (int t = 0; t < T; ++t) { const float d = col[(int64_t)t * F] - mean; var = __builtin_fmaf(d, d, var); }
  const float sd = sqrtf(var / (float)T);
  for (int t = 0; t < T; ++t) col[(int64_t)t * F] = (col[(int64_t)t * F] - mean) / sd;
}

// out[t - row_begin][j] = -mean_h w[h][t][j]   (timing.py:214-216: mean over heads, row slice, negation for dtw)
__global__ void align_head_mean_kernel(const float* __restrict__ w, int H, int T, int F, int row_begin, int rows,
                                       float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)rows * F) return;
  const int r = (int)(gid / F), j = (int)(gid - (int64_t)r * F);
  float s = 0.f;
  for (int h = 0; h < H; ++h) s += w[((int64_t)h * T + row_begin + r) * F + j];
  out[gid] = -(s / (float)H);
}

}  // namespace

namespace whk {

hipError_t launch_align_matrix(const float* qk, int H, int T, int Tk, int F, int width, int row_begin,
                               int row_end, float qk_scale, float* out, float* scratch, hipStream_t stream) {
  float* w0 = scratch;
  float* w1 = scratch + (size_t)H * T * F;
  hipLaunchKernelGGL(align_softmax_kernel, dim3(T, H), dim3(256), 0, stream, qk, T, Tk, F, qk_scale, w0);
  hipLaunchKernelGGL(align_znorm_kernel, dim3((F + 63) / 64, H), dim3(64), 0, stream, w0, T, F);
  hipError_t e = launch_median_filter(w0, w1, (int64_t)H * T, F, width, stream);
  if (e != hipSuccess) return e;
  const int rows = row_end - row_begin;
  const int64_t total = (int64_t)rows * F;
  hipLaunchKernelGGL(align_head_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w1, H, T, F,
                     row_begin, rows, out);
  return hipGetLastError();
}

hipError_t launch_median_filter(const float* x, float* out, int64_t rows, int n, int width,
                                hipStream_t stream) {
  if (width <= 0 || (width & 1) == 0 || width > 63) return hipErrorInvalidValue;
  const int64_t total = rows * n;
  if (total <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (n <= width / 2) {   // timing.py:22-24: input returned unchanged
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, stream, x, out, total);
    return hipGetLastError();
  }
  switch (width) {
#define MED_CASE(W) case W: hipLaunchKernelGGL((median_kernel<W>), dim3(blocks), dim3(256), 0, stream, x, out, rows, n); break;
    MED_CASE(1) MED_CASE(3) MED_CASE(5) MED_CASE(7) MED_CASE(9) MED_CASE(11) MED_CASE(13)
#undef MED_CASE
    default:
      hipLaunchKernelGGL(median_generic_kernel, dim3(blocks), dim3(256), 0, stream, x, out, rows, n, width);
  }
  return hipGetLastError();
}

hipError_t launch_dtw(const float* x, int N, int M, int8_t* trace, hipStream_t stream) {
  if (N <= 0 || M <= 0 || N > 8192) return hipErrorInvalidValue;
  int threads = ((N + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const size_t lds = 3 * (size_t)(N + 1) * sizeof(float);
  hipLaunchKernelGGL(dtw_kernel, dim3(1), dim3(threads), lds, stream, x, N, M, trace);
  return hipGetLastError();
}

}  // namespace whk
