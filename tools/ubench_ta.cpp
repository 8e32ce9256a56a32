// tools/ubench_ta.cpp — developer probe (not part of the product): what does one 16-byte-per-lane wave-load cost a CU as a function of
// WHICH lanes share a cache line?  256 workgroups x 16 waves, every wave issues NL loads of 1 KB from an L2-resident buffer and waits:
//   diag   : the decode projections' fragment map — lane 16 c + 8 half + i reads 16 B at row i, byte 64 half + 16 c (8 rows x 128 B)
//   rows8  : the same 8 rows x 128 B, lanes in memory order — lane 8 r + u reads 16 B at row r, byte 16 u
//   linear : 1 KB contiguous, lane l reads bytes [16 l, +16)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_ta.cpp -o tools/ubench_ta
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE, int NL>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, int row_stride, int nunits, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  float4v v[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int unit = (wave * NL + j + blockIdx.x * 7) % nunits;          // which 1 KB unit / K block
    size_t off;
    if (MODE == 0) off = (size_t)(lane & 7) * row_stride + (size_t)unit * 128 + 64 * ((lane >> 3) & 1) + 16 * (lane >> 4);
    else if (MODE == 1) off = (size_t)(lane >> 3) * row_stride + (size_t)unit * 128 + 16 * (lane & 7);
    else off = (size_t)unit * 1024 + 16 * lane;
    v[j] = __builtin_nontemporal_load((const float4v*)(buf + off));
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NL> void run(const char* name, const char* buf, int row_stride, int nunits, float* out, long long* cyc) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NL>), dim3(256), dim3(1024), 0, 0, buf, row_stride, nunits, out, cyc);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NL>), dim3(256), dim3(1024), 0, 0, buf, row_stride, nunits, out, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-8s NL=%2d  %7.2f us per launch   workgroup cycles med %6lld max %6lld  -> %5.1f cycles per wave-load (16 waves x NL)\n", name, NL, ms * 1e3 / 20,
         h[128], h[255], (double)h[128] / (16.0 * NL));
}

int main() {
  const int row_stride = 10240, nunits = 80;                 // 8 rows x 10 KB = the FC2 x block of one row tile
  char* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  float* out; CK(hipMalloc(&out, 4096)); long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
  run<0, 5>("diag", buf, row_stride, nunits, out, cyc);   run<1, 5>("rows8", buf, row_stride, nunits, out, cyc);   run<2, 5>("linear", buf, row_stride, nunits, out, cyc);
  run<0, 10>("diag", buf, row_stride, nunits, out, cyc);  run<1, 10>("rows8", buf, row_stride, nunits, out, cyc);  run<2, 10>("linear", buf, row_stride, nunits, out, cyc);
  run<0, 20>("diag", buf, row_stride, nunits, out, cyc);  run<1, 20>("rows8", buf, row_stride, nunits, out, cyc);  run<2, 20>("linear", buf, row_stride, nunits, out, cyc);
  return 0;
}
