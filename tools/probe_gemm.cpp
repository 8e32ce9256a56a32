// tools/probe_gemm.cpp — developer microbenchmark of gemm.hip on the large-v3 encoder shapes (B = 8 -> M = 12000).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_gemm.cpp -o tools/probe_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
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
    CK(hipMemset(C32, 0, (size_t)M * D * 4));
  }
  std::vector<float> hbias(4 * D);
  for (auto& v : hbias) v = ((rand() & 0xffff) / 65536.0f - 0.5f) * 0.5f;
  CK(hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
  std::vector<half_t> hA((size_t)M * 4 * D), hW((size_t)4 * D * D * 4), hC((size_t)M * 4 * D);
  CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
#ifdef WH_PROBE
  long long* d_probe; CK(hipMalloc(&d_probe, 256 * 8 * 8)); CK(hipMemset(d_probe, 0, 256 * 8 * 8));
#endif
  struct Case { const char* name; int N, K, act, f32, res; };
  Case cases[] = {{"qkv   N=2560 K=1280      ", 2 * D, D, 0, 0, 0}, {"fc1   N=5120 K=1280 gelu ", 4 * D, D, 1, 0, 0},
                  {"fc1   N=5120 K=1280 noact", 4 * D, D, 0, 0, 0}, {"out   N=1280 K=1280 res32", D, D, 0, 1, 1},
                  {"fc2   N=1280 K=5120 res32", D, 4 * D, 0, 1, 1}};
  for (const Case& c : cases) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) {
        whk::GemmArgs g; memset(&g, 0, sizeof(g));
        g.A = A; g.lda = c.K; g.W = W + (size_t)(i % 4) * 4 * D * D; g.ldw = c.K;
        g.C = c.f32 ? (void*)C32 : (void*)C16; g.ldc = c.N; g.bias = bias; g.act = c.act;
        if (c.res) { g.res = C32; g.ldr = c.N; }
        g.M = M; g.N = c.N; g.K = c.K;
#ifdef WH_PROBE
        g.probe = (rep == 5 && i == 9) ? d_probe : nullptr;
#endif
        CK(whk::launch_gemm(g, 1, c.f32, 1, st));
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rep > 0 && t < best) best = t;
    }
    const float ms = best;
    const double us = ms * 1e3 / 10, tf = 2.0 * M * c.N * c.K / us * 1e-6;
    printf("%s %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)", c.name, us, tf, tf / 25.0);
    if (!c.f32) {   // spot check against host dot products (the last launch used weight slice 9 % 4 = 1)
      CK(hipMemcpy(hC.data(), C16, (size_t)M * c.N * 2, hipMemcpyDeviceToHost));
      const half_t* w = hW.data() + (size_t)1 * 4 * D * D;
      double worst = 0;
      for (int t = 0; t < 4000; ++t) {
        const int m = t < 64 ? M - 1 - t : rand() % M, n = t < 64 ? c.N - 1 - (t * 7) % c.N : rand() % c.N;
        double acc = hbias[n];
        for (int k = 0; k < c.K; ++k) acc += (double)(float)hA[(size_t)m * c.K + k] * (double)(float)w[(size_t)n * c.K + k];
        if (c.act) acc = 0.5 * acc * (1.0 + erf(acc * 0.7071067811865476));
        const double d = fabs(acc - (double)(float)hC[(size_t)m * c.N + n]);
        if (d > worst) worst = d;
      }
      printf("   max |err| over 4000 samples %.2e", worst);
    }
    printf("\n");
#ifdef WH_PROBE
    {
      std::vector<long long> pr(256 * 8); CK(hipMemcpy(pr.data(), d_probe, pr.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemset(d_probe, 0, 256 * 8 * 8));
      printf("     phase medians over workgroups (cycles from kernel entry):");
      for (int ph = 1; ph < 8; ++ph) {
        std::vector<long long> d;
        for (int w = 0; w < 256; ++w) if (pr[w * 8] && pr[w * 8 + ph]) d.push_back(pr[w * 8 + ph] - pr[w * 8]);
        if (d.empty()) { printf("  [%d] -", ph); continue; }
        std::sort(d.begin(), d.end());
        printf("  [%d] %lld", ph, d[d.size() / 2]);
      }
      printf("\n");
    }
#endif
  }
  {  // encoder flash attention: B = 8, H = 20, T = 1500 (q,k row-major [T][2D], V^T [D][1536])
    const int B = 8, H = 20, T = 1500;
    half_t* qk = A; half_t* vt = W; half_t* o = C16;
    std::vector<half_t> ref((size_t)B * T * D), got((size_t)B * T * D);
    const int modes[4] = {2, 3, 1, 3};             // bit 0: pre-scaled q,k; bit 1: plain V^T tile layout (the product stores it in P order)
    for (int mi = 0; mi < 4; ++mi) {
      const int pre = modes[mi];
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i)
          CK(whk::launch_attn_flash_f16(qk, 2 * D, (int64_t)T * 2 * D, qk + D, 2 * D, (int64_t)T * 2 * D, vt, 1536, (int64_t)D * 1536,
                                        o, D, (int64_t)T * D, B, H, T, pre, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep > 0 && t < best) best = t;
      }
      const double us = best * 1e3 / 10, tf = 4.0 * T * T * 64 * H * B / us * 1e-6;
      printf("flash attention B=8 H=20 T=1500 %s%s %8.1f us  %7.1f TFLOP/s  (%.1f %% of 2500)\n", (pre & 1) ? "prescaled q,k" : "unscaled q,k ",
             (pre & 2) ? ", plain V^T tile" : "                 ", us, tf, tf / 25.0);
      if (pre == 3 && mi == 1) CK(hipMemcpy(ref.data(), o, ref.size() * 2, hipMemcpyDeviceToHost));
      if (pre == 1) {
        CK(hipMemcpy(got.data(), o, got.size() * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, nan = 0; double mx = 0;
        for (size_t i = 0; i < got.size(); ++i) {
          const float a = (float)ref[i], b = (float)got[i];
          if (a != a || b != b) { ++nan; continue; }
          if (memcmp(&ref[i], &got[i], 2) != 0) { ++diff; if (fabs(a - b) > mx) mx = fabs(a - b); }
        }
        printf("  P-order V^T tile vs plain layout: %zu of %zu outputs differ (max |diff| %.3g), %zu NaN\n", diff, got.size(), mx, nan);
      }
    }
  }
  return 0;
}
