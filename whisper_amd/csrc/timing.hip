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

// ---- find_alignment post-processing (whisper/timing.py:207-216), batched over clips ---------------------------
// Clip b (blockIdx.z) holds T_b = ntok[b] token rows and F_b = nfr[b] frames inside slabs sized for (Tmax, Fmax); a
// single clip is a batch of one through the same kernels, so batched and clip-by-clip results are bit-identical.
struct AlignBatch {
  const int* ntok; const int* nfr;      // device [clips]
  int H, Tmax, Tk, Fmax;                // heads (pairs), slab rows, keys per qk row, slab columns
};

// qk [clip][H][Tmax][Tk] -> w [clip][H][Tmax][Fmax]: softmax over the first F frames of (qk * qk_scale)   (timing.py:208-209)
__global__ __launch_bounds__(256) void align_softmax_kernel(const float* __restrict__ qk, AlignBatch ab, float qk_scale,
                                                            float* __restrict__ w) {
  __shared__ float red[4];
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = ab.ntok[b], F = ab.nfr[b];
  if (t >= T) return;                    // workgroup-uniform
  const float* src = qk + (((int64_t)b * ab.H + h) * ab.Tmax + t) * ab.Tk;
  float* dst = w + (((int64_t)b * ab.H + h) * ab.Tmax + t) * ab.Fmax;
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
__global__ void align_znorm_kernel(float* __restrict__ w, AlignBatch ab) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int T = ab.ntok[b], F = ab.nfr[b];
  if (j >= F) return;
  float* col = w + ((int64_t)b * ab.H + h) * ab.Tmax * ab.Fmax + j;
  const int64_t ld = ab.Fmax;
  float mean = 0.f;
  for (int t = 0; t < T; ++t) mean += col[(int64_t)t * ld];
  mean /= (float)T;
  float var = 0.f;
  for (int t = 0; t < T; ++t) { const float d = col[(int64_t)t * ld] - mean; var = __builtin_fmaf(d, d, var); }
  const float sd = sqrtf(var / (float)T);
  for (int t = 0; t < T; ++t) col[(int64_t)t * ld] = (col[(int64_t)t * ld] - mean) / sd;
}

// sliding median of width W along the first F_b entries of every slab row (rows >= T_b are skipped)
template <int W>
__global__ void align_median_kernel(const float* __restrict__ x, float* __restrict__ out, AlignBatch ab) {
  const int b = blockIdx.z, F = ab.nfr[b], T = ab.ntok[b];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over H * Tmax * Fmax of this clip
  const int64_t per = (int64_t)ab.H * ab.Tmax * ab.Fmax;
  if (gid >= per) return;
  const int i = (int)(gid % ab.Fmax);
  const int t = (int)((gid / ab.Fmax) % ab.Tmax);
  if (i >= F || t >= T) return;
  const float* xr = x + b * per + (gid - i);
  float* o = out + b * per + gid;
  if (F <= W / 2) { *o = xr[i]; return; }                                 // timing.py:22-24: returned unchanged
  float v[W];
#pragma unroll
  for (int j = 0; j < W; ++j) v[j] = xr[reflect_idx(i - W / 2 + j, F)];
#pragma unroll
  for (int p = 0; p < W / 2 + 1; ++p) {
#pragma unroll
    for (int j = 0; j < W - 1 - p; ++j) {
      const float lo = fminf(v[j], v[j + 1]), hi = fmaxf(v[j], v[j + 1]);
      v[j] = lo; v[j + 1] = hi;
    }
  }
  *o = v[W / 2];
}

// any odd width <= 63 (window in scratch memory; correctness path only)
__global__ void align_median_generic_kernel(const float* __restrict__ x, float* __restrict__ out, AlignBatch ab, int W) {
  const int b = blockIdx.z, F = ab.nfr[b], T = ab.ntok[b];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)ab.H * ab.Tmax * ab.Fmax;
  if (gid >= per) return;
  const int i = (int)(gid % ab.Fmax);
  const int t = (int)((gid / ab.Fmax) % ab.Tmax);
  if (i >= F || t >= T) return;
  const float* xr = x + b * per + (gid - i);
  float* o = out + b * per + gid;
  if (F <= W / 2) { *o = xr[i]; return; }
  float v[63];
  for (int j = 0; j < W; ++j) v[j] = xr[reflect_idx(i - W / 2 + j, F)];
  for (int p = 0; p < W / 2 + 1; ++p)
    for (int j = 0; j < W - 1 - p; ++j) {
      const float lo = fminf(v[j], v[j + 1]), hi = fmaxf(v[j], v[j + 1]);
      v[j] = lo; v[j + 1] = hi;
    }
  *o = v[W / 2];
}

__global__ void set_two_ints_kernel(int* p, int a, int b) { p[0] = a; p[1] = b; }

// out[clip][t - row_begin][j] = -mean_h w[clip][h][t][j], rows t in [row_begin, T_b - row_tail)   (timing.py:214-216)
__global__ void align_head_mean_kernel(const float* __restrict__ w, AlignBatch ab, int row_begin, int row_tail, int Nmax,
                                       float* __restrict__ out) {
  const int b = blockIdx.z, F = ab.nfr[b], rows = ab.ntok[b] - row_tail - row_begin;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)Nmax * ab.Fmax) return;
  const int r = (int)(gid / ab.Fmax), j = (int)(gid - (int64_t)r * ab.Fmax);
  if (r >= rows || j >= F) return;
  float s = 0.f;
  for (int h = 0; h < ab.H; ++h) s += w[(((int64_t)b * ab.H + h) * ab.Tmax + row_begin + r) * ab.Fmax + j];
  out[((int64_t)b * Nmax + r) * ab.Fmax + j] = -(s / (float)ab.H);
}

// dtw of every clip's cost matrix [N_b][F_b] (rows of stride Fmax); trace of clip b at trace + b * trace_bs, dense
// [(N_b + 1)][(F_b + 1)].  One workgroup per clip, same wavefront as dtw_kernel.
__global__ __launch_bounds__(1024) void dtw_batch_kernel(const float* __restrict__ x, AlignBatch ab, int row_begin,
                                                         int row_tail, int Nmax, int8_t* __restrict__ trace_all,
                                                         int64_t trace_bs) {
  extern __shared__ float diag[];          // 3 x (Nmax+1)
  const int b = blockIdx.x;
  const int N = ab.ntok[b] - row_tail - row_begin, M = ab.nfr[b];
  if (N <= 0 || M <= 0) return;
  const float* xb = x + (int64_t)b * Nmax * ab.Fmax;
  const int64_t ldx = ab.Fmax;
  int8_t* trace = trace_all + b * trace_bs;
  float* d[3] = {diag, diag + (Nmax + 1), diag + 2 * (Nmax + 1)};
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = tid + 1; j <= M; j += nt) trace[j] = 2;
  for (int i = tid; i <= N; i += nt) trace[(int64_t)i * (M + 1)] = 1;
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
        c = xb[(int64_t)(i - 1) * ldx + (j - 1)] + cm;
        trace[(int64_t)i * (M + 1) + j] = t;
      }
      d0[i] = c;
    }
    if (tid == 0) d0[0] = __builtin_huge_valf();
    __syncthreads();
  }
}

}  // namespace

namespace whk {

hipError_t launch_align_batch(const float* qk, const int* d_ntok, const int* d_nfr, int clips, int H, int Tmax, int Tk,
                              int Fmax, int width, int row_begin, int row_tail, float qk_scale, float* cost, int Nmax,
                              float* scratch, hipStream_t stream) {
  if (width <= 0 || (width & 1) == 0 || width > 63) return hipErrorInvalidValue;
  AlignBatch ab; ab.ntok = d_ntok; ab.nfr = d_nfr; ab.H = H; ab.Tmax = Tmax; ab.Tk = Tk; ab.Fmax = Fmax;
  const int64_t per = (int64_t)H * Tmax * Fmax;
  float* w0 = scratch;
  float* w1 = scratch + (size_t)clips * per;
  hipLaunchKernelGGL(align_softmax_kernel, dim3(Tmax, H, clips), dim3(256), 0, stream, qk, ab, qk_scale, w0);
  hipLaunchKernelGGL(align_znorm_kernel, dim3((Fmax + 63) / 64, H, clips), dim3(64), 0, stream, w0, ab);
  const dim3 mg((unsigned)((per + 255) / 256), 1, clips);
  switch (width) {
#define AMED(W) case W: hipLaunchKernelGGL((align_median_kernel<W>), mg, dim3(256), 0, stream, w0, w1, ab); break;
    AMED(1) AMED(3) AMED(5) AMED(7) AMED(9) AMED(11) AMED(13)
#undef AMED
    default: hipLaunchKernelGGL(align_median_generic_kernel, mg, dim3(256), 0, stream, w0, w1, ab, width);
  }
  const int64_t total = (int64_t)Nmax * Fmax;
  hipLaunchKernelGGL(align_head_mean_kernel, dim3((unsigned)((total + 255) / 256), 1, clips), dim3(256), 0, stream, w1, ab,
                     row_begin, row_tail, Nmax, cost);
  return hipGetLastError();
}

// single clip (wh_align_matrix): a batch of one; scratch = 2 * H * T * F floats followed by 2 ints
hipError_t launch_align_matrix(const float* qk, int H, int T, int Tk, int F, int width, int row_begin,
                               int row_end, float qk_scale, float* out, float* scratch, hipStream_t stream) {
  int* sizes = (int*)(scratch + (size_t)2 * H * T * F);
  hipLaunchKernelGGL(set_two_ints_kernel, dim3(1), dim3(1), 0, stream, sizes, T, F);
  return launch_align_batch(qk, sizes, sizes + 1, 1, H, T, Tk, F, width, row_begin, T - row_end, qk_scale, out,
                            row_end - row_begin, scratch, stream);
}

hipError_t launch_dtw_batch(const float* cost, const int* d_ntok, const int* d_nfr, int clips, int Tmax, int Fmax,
                            int row_begin, int row_tail, int Nmax, int8_t* trace, int64_t trace_bs, hipStream_t stream) {
  if (Nmax <= 0 || Nmax > 8192) return hipErrorInvalidValue;
  AlignBatch ab; ab.ntok = d_ntok; ab.nfr = d_nfr; ab.H = 0; ab.Tmax = Tmax; ab.Tk = 0; ab.Fmax = Fmax;
  int threads = ((Nmax + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  const size_t lds = 3 * (size_t)(Nmax + 1) * sizeof(float);
  hipLaunchKernelGGL(dtw_batch_kernel, dim3(clips), dim3(threads), lds, stream, cost, ab, row_begin, row_tail, Nmax, trace, trace_bs);
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
