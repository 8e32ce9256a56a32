// tools/probe_floor.cpp — developer probe (not part of the product): what does ONE link of a dependent chain of
// decode-step kernels cost on MI355X, feature by feature?  64 dependent launches replayed from a hipGraph; the
// kernel body is assembled from template flags so that each line of the table adds one ingredient of a real
// decode GEMV (kernel-argument fetch, activation read from L2, weight stream from HBM, LDS staging + barrier,
// cross-wave reduction, output store).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_floor.cpp -o tools/probe_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

struct Args {
  const float4v* w;     // weights: [nwg][threads][WU] units of 16 B
  const float* xin;     // [8][1280] fp32 activations written by the previous kernel
  float* xout;
  int n_out;            // outputs per workgroup (<= 64)
};

// ACT: 0 none, 1 read the 40 KB activation block (every workgroup reads all of it), 2 read only 5 KB (one row)
// WU : 16-byte weight units per thread (0 = no weight stream)
// LDS: stage activations through LDS + barrier, reduce across waves through LDS + barrier
template <int THREADS, int ACT, int WU, bool LDS>
__global__ __launch_bounds__(THREADS) void link_kernel(Args a) {
  __shared__ float xs[8 * 1280];
  __shared__ float red[16];
  const int tid = threadIdx.x, wg = blockIdx.x;
  float4v w[WU > 0 ? WU : 1];
  float acc = 0.f;
  constexpr int NX = ACT == 1 ? (8 * 1280 / 4 + THREADS - 1) / THREADS : ACT == 2 ? (1280 / 4 + THREADS - 1) / THREADS : 0;
  float4v xv[NX > 0 ? NX : 1];
  if (ACT) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      int i = j * THREADS + tid; const int lim = (ACT == 1 ? 8 * 1280 : 1280) / 4;
      if (i > lim - 1) i = lim - 1;
      xv[j] = *(const float4v*)(a.xin + i * 4);
    }
  }
  asm volatile("" ::: "memory");
  if (WU > 0) {
#pragma unroll
    for (int u = 0; u < WU; ++u) w[u] = __builtin_nontemporal_load(a.w + ((size_t)wg * WU + u) * THREADS + tid);   // 1 KB contiguous per wave-load
  }
  asm volatile("" ::: "memory");
  if (ACT) {
    if (LDS) {
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int i = j * THREADS + tid; const int lim = (ACT == 1 ? 8 * 1280 : 1280) / 4;
        if (i < lim) *(float4v*)(xs + i * 4) = xv[j];
      }
      __syncthreads();
      acc += xs[(tid * 7) % (ACT == 1 ? 10240 : 1280)];
    } else {
#pragma unroll
      for (int j = 0; j < NX; ++j) acc += xv[j][0] + xv[j][3];
    }
  }
  if (WU > 0) {
#pragma unroll
    for (int u = 0; u < WU; ++u) acc += w[u][0] * 0.5f + w[u][1] + w[u][2] + w[u][3];
  }
  if (LDS) {
    // wave reduction (DPP-free stand-in: shuffles) + cross-wave through LDS
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    float v = 0.f;
    for (int k = 0; k < THREADS / 64; ++k) v += red[k];
    acc = v;
  }
  if (tid < a.n_out) a.xout[(tid >> 3) * 1280 + (wg * 8 + (tid & 7)) % 1280] = acc * 1e-6f;
}

static hipStream_t st;
static hipEvent_t e0, e1;

template <int THREADS, int ACT, int WU, bool LDS>
static void run(const char* name, int nwg, const float4v* w, float** x) {
  const int N = 64, LAYERS = 8;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) {
    Args a; a.w = w + (size_t)(k % LAYERS) * 4 * 1024 * 1024; a.xin = x[k & 1]; a.xout = x[(k + 1) & 1]; a.n_out = 64;
    hipLaunchKernelGGL((link_kernel<THREADS, ACT, WU, LDS>), dim3(nwg), dim3(THREADS), 0, st, a);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 1 && ms < best) best = ms;
  }
  const double wbytes = (double)nwg * THREADS * WU * 16;
  printf("%-64s wgs %4d x %4d thr  weights %6.2f MB  %6.2f us per link\n", name, nwg, THREADS, wbytes / 1e6, best * 1e3f / N);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float4v* w; CK(hipMalloc(&w, (size_t)8 * 4 * 1024 * 1024 * 16)); CK(hipMemset(w, 0, (size_t)8 * 4 * 1024 * 1024 * 16));   // 8 x 64 MB: beyond the 256 MB MALL
  float* x[2]; CK(hipMalloc(&x[0], 8 * 1280 * 4)); CK(hipMalloc(&x[1], 8 * 1280 * 4));
  CK(hipMemset(x[0], 0, 8 * 1280 * 4)); CK(hipMemset(x[1], 0, 8 * 1280 * 4));

  run<256, 0, 0, false>("store only", 160, w, x);
  run<256, 2, 0, false>("+ 5 KB activations (registers)", 160, w, x);
  run<256, 1, 0, false>("+ 40 KB activations (registers)", 160, w, x);
  run<256, 1, 0, true>("+ 40 KB activations -> LDS, barrier, wave+LDS reduction", 160, w, x);
  run<256, 0, 5, false>("weights 5 units/thread only", 160, w, x);
  run<256, 0, 5, true>("weights 5 units/thread + reduction", 160, w, x);
  run<256, 1, 5, true>("40 KB act -> LDS + weights 5 units (D x D shape)", 160, w, x);
  run<256, 2, 5, true>("5 KB act -> LDS + weights 5 units", 160, w, x);
  run<256, 1, 10, true>("40 KB act -> LDS + weights 10 units", 160, w, x);
  run<512, 1, 5, true>("40 KB act -> LDS + weights 5 units, 512 thr", 160, w, x);
  run<512, 1, 5, true>("40 KB act -> LDS + weights 5 units, 512 thr (qkv bytes)", 240, w, x);
  run<1024, 1, 5, true>("40 KB act -> LDS + weights 5 units, 1024 thr (fc1 bytes)", 160, w, x);
  run<256, 1, 10, true>("40 KB act -> LDS + weights 10 units, 256 thr, 320 wgs", 320, w, x);
  run<256, 1, 5, true>("40 KB act -> LDS + weights 5 units, 256 thr, 640 wgs", 640, w, x);
  run<256, 1, 16, true>("40 KB act -> LDS + weights 16 units, 256 thr, 256 wgs (16 MB)", 256, w, x);
  run<256, 0, 16, true>("weights 16 units only, 256 thr, 256 wgs (16 MB)", 256, w, x);
  run<256, 0, 16, true>("weights 16 units only, 256 thr, 512 wgs (32 MB)", 512, w, x);
  run<256, 0, 16, true>("weights 16 units only, 256 thr, 1024 wgs (64 MB)", 1024, w, x);
  return 0;
}
