// tools/probe_gemm.cpp — developer microbenchmark of gemm.hip on the large-v3 encoder shapes (B = 8 -> M = 12000).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_gemm.cpp -o tools/probe_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../whisper_amd/csrc/gemm.hip"
#include "../whisper_amd/csrc/attention.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
  const int M = 12000, D = 1280;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  half_t *A, *W, *C16; float *C32, *bias;
  CK(hipMalloc(&A, (size_t)M * 4 * D * 2)); CK(hipMalloc(&W, (size_t)4 * D * D * 2 * 4));
  CK(hipMalloc(&C16, (size_t)M * 4 * D * 2)); CK(hipMalloc(&C32, (size_t)M * D * 4)); CK(hipMalloc(&bias, 4 * D * 4));
  {
    std::vector<half_t> h((size_t)M * 4 * D);
    for (auto& v : h) v = (half_t)(((rand() & 0xffff) / 65536.0f - 0.5f));
    CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    h.resize((size_t)4 * D * D * 4);
    for (auto& v : h) v = (half_t)(((rand() & 0xffff) / 65536.0f - 0.5f) * 0.05f);
    CK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 4 * D * 4)); CK(hipMemset(C32, 0, (size_t)M * D * 4));
  }
  struct Case { const char* name; int N, K, act, f32, res; };
  Case cases[] = {{"qkv   N=2560 K=1280      ", 2 * D, D, 0, 0, 0}, {"fc1   N=5120 K=1280 gelu ", 4 * D, D, 1, 0, 0},
                  {"fc1   N=5120 K=1280 noact", 4 * D, D, 0, 0, 0}, {"out   N=1280 K=1280 res32", D, D, 0, 1, 1},
                  {"fc2   N=1280 K=5120 res32", D, 4 * D, 0, 1, 1}};
  for (const Case& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) {
        whk::GemmArgs g; memset(&g, 0, sizeof(g));
        g.A = A; g.lda = c.K; g.W = W + (size_t)(i % 4) * 4 * D * D; g.ldw = c.K;
        g.C = c.f32 ? (void*)C32 : (void*)C16; g.ldc = c.N; g.bias = bias; g.act = c.act;
        if (c.res) { g.res = C32; g.ldr = c.N; }
        g.M = M; g.N = c.N; g.K = c.K;
        CK(whk::launch_gemm(g, 1, c.f32, 1, st));
      }
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 10, tf = 2.0 * M * c.N * c.K / us * 1e-6;
    printf("%s %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)\n", c.name, us, tf, tf / 25.0);
  }
  {  // encoder flash attention: B = 8, H = 20, T = 1500 (q,k row-major [T][2D], V^T [D][1536])
    const int B = 8, H = 20, T = 1500;
    half_t* qk = A; half_t* vt = W; half_t* o = C16;
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i)
        CK(whk::launch_attn_flash_f16(qk, 2 * D, (int64_t)T * 2 * D, qk + D, 2 * D, (int64_t)T * 2 * D, vt, 1536, (int64_t)D * 1536,
                                      o, D, (int64_t)T * D, B, H, T, st));
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 10, tf = 4.0 * T * T * 64 * H * B / us * 1e-6;
    printf("flash attention B=8 H=20 T=1500 %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)\n", us, tf, tf / 25.0);
  }
  return 0;
}
