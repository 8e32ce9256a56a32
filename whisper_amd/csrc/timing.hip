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


// ---- back-trace of every clip's trace matrix on the device (timing.py:57-79) ------------------------------------------
// One workgroup per clip.  The clip's int8 trace [(N+1)][(M+1)] is first packed to 2 bits per entry in LDS by all
// threads (coalesced reads; a 230 x 1501 trace is 86 KB packed), then ONE lane walks it from (N, M) to the origin: an
// LDS load per step instead of a dependent global load (the reference walks a numpy array on the host; round 2 copied
// every clip's whole trace to the host for that).  Traces too large for the LDS are walked in global memory.
// Outputs, per clip:
//   jumps[i], i in [0, N): time index (frame) of the FIRST path entry whose text index is i — exactly
//     `time_indices[jumps]` of timing.py:226-228 (the path visits every text index, in order);
//   path (optional): the (text, time) pairs right-aligned in [2][path_stride] (entries [path_stride - len, path_stride)),
//     path_len[clip] = len — the (2, len) array `dtw()` returns (timing.py:141-151).
// Border rule as the reference: row 0 moves left, column 0 moves up (timing.py:61-62); a code outside {0,1,2} stops the
// walk and reports length -1 (the reference raises ValueError).
__global__ __launch_bounds__(1024) void dtw_backtrace_batch_kernel(const int8_t* __restrict__ trace_all, int64_t trace_bs,
                                                                   const int* __restrict__ n_rows, const int* __restrict__ n_cols,
                                                                   int lds_bytes, int* __restrict__ jumps, int64_t jump_stride,
                                                                   int* __restrict__ path, int64_t path_stride,
                                                                   int* __restrict__ path_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char packed[];
  const int b = blockIdx.x;
  const int N = n_rows[b], M = n_cols[b];
  if (N <= 0 || M <= 0) { if (path_len && threadIdx.x == 0) path_len[b] = 0; return; }
  const int8_t* trace = trace_all + b * trace_bs;
  const int W = M + 1, RS = (W + 3) >> 2;                     // packed row stride in bytes
  const bool in_lds = (int64_t)(N + 1) * RS <= lds_bytes;
  if (in_lds) {
    const int total = (N + 1) * RS;
    for (int q = threadIdx.x; q < total; q += blockDim.x) {
      const int i = q / RS, j0 = (q - i * RS) << 2;
      const int8_t* src = trace + (int64_t)i * W + j0;
      unsigned v = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j0 + e < W) v |= ((unsigned)src[e] & 3u) << (2 * e);      // codes 0..2; anything else was never written (-1 -> 3)
      packed[q] = (unsigned char)v;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  int i = N, j = M, len = 0;
  int* jp = jumps ? jumps + b * jump_stride : nullptr;
  int* pt = path ? path + b * 2 * path_stride : nullptr;
  bool bad = false;
  while (i > 0 || j > 0) {
    if (jp && i >= 1) jp[i - 1] = j - 1;                       // the last write per i is the first entry in path order
    if (pt && len < path_stride) { pt[path_stride - 1 - len] = i - 1; pt[2 * path_stride - 1 - len] = j - 1; }
    ++len;
    int t;
    if (i == 0) t = 2;
    else if (j == 0) t = 1;
    else if (in_lds) t = (packed[i * RS + (j >> 2)] >> (2 * (j & 3))) & 3;
    else t = trace[(int64_t)i * W + j] & 3;
    if (t == 0) { --i; --j; }
    else if (t == 1) --i;
    else if (t == 2) --j;
    else { bad = true; break; }
  }
  if (path_len) path_len[b] = bad ? -1 : len;
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

hipError_t launch_dtw_backtrace_batch(const int8_t* trace, int64_t trace_bs, const int* d_rows, const int* d_cols, int clips,
                                      int max_rows, int max_cols, int* jumps, int64_t jump_stride, int* path,
                                      int64_t path_stride, int* path_len, hipStream_t stream) {
  if (clips <= 0) return hipSuccess;
  const int64_t need = (int64_t)(max_rows + 1) * ((max_cols + 4) >> 2);
  const int LDS_MAX = 156 * 1024;
  int lds = need <= LDS_MAX ? (int)((need + 15) & ~15ll) : 0;                // 0: walk the global trace
  // where the dynamic-LDS limit cannot be raised on this device the walk runs on the global trace (lds = 0)
  static whk::LdsAttr attr;
  if (lds > 64 * 1024 && whk::raise_dynamic_lds(attr, (const void*)dtw_backtrace_batch_kernel, LDS_MAX) != hipSuccess) lds = 0;
  hipLaunchKernelGGL(dtw_backtrace_batch_kernel, dim3(clips), dim3(lds ? 1024 : 64), (size_t)lds, stream, trace, trace_bs,
                     d_rows, d_cols, lds, jumps, jump_stride, path, path_stride, path_len);
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
