// developer microbenchmark: does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=16, scalar kernel
// arguments delivered in SGPRs at wave launch instead of through an s_load) shorten a short dependent kernel?
// Build twice (with / without the flag) and compare: 200 dependent launches replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(const float* __restrict__ p, float* __restrict__ o, const float* __restrict__ w, int n, float s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = p[i] * s + w[i];
}
int main() {
  const int n = 160 * 256;
  float *a, *b, *w; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&w, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(w, 0, n * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(160), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, w, n, 1.0f);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipGraphLaunch(e, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < 10; ++i) hipGraphLaunch(e, st);
  hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%.3f us per dependent launch\n", ms * 1e3f / 2000);
  return 0;
}
