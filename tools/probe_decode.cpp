// tools/probe_decode.cpp — developer microbenchmark for the decode-step kernels (not part of the product).
// Includes the kernel sources directly with -DWH_PROBE so that thread 0 of every workgroup records
// s_memtime at phase boundaries; prints per-launch HIP-event time and the median phase durations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWH_PROBE -I include tools/probe_decode.cpp -o tools/probe_decode
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../whisper_amd/csrc/gemv.hip"
#include "../whisper_amd/csrc/attention.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static void fill_half(half_t* d, size_t n, float scale) {
  std::vector<half_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (half_t)(((rand() & 0xffff) / 65536.0f - 0.5f) * scale);
  CK(hipMemcpy(d, h.data(), n * sizeof(half_t), hipMemcpyHostToDevice));
}
static void fill_float(float* d, size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = ((rand() & 0xffff) / 65536.0f - 0.5f) * scale;
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}

static void report(const char* name, float us, double bytes, long long* d_probe, int nwg, int npts) {
  std::vector<long long> p((size_t)nwg * 8);
  CK(hipMemcpy(p.data(), d_probe, p.size() * 8, hipMemcpyDeviceToHost));
  printf("%-28s %7.2f us/launch  %7.0f GB/s  wgs=%d | phase medians (cycles):", name, us, bytes / us * 1e-3, nwg);
  for (int i = 1; i < npts; ++i) {
    std::vector<long long> d;
    for (int w = 0; w < nwg; ++w) d.push_back(p[(size_t)w * 8 + i] - p[(size_t)w * 8 + i - 1]);
    std::sort(d.begin(), d.end());
    printf(" %lld", d[d.size() / 2]);
  }
  long long t0 = p[0], t1 = p[npts - 1];
  std::vector<long long> tot;
  for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, p[(size_t)w * 8]); t1 = std::max(t1, p[(size_t)w * 8 + npts - 1]); tot.push_back(p[(size_t)w * 8 + npts - 1] - p[(size_t)w * 8]); }
  std::sort(tot.begin(), tot.end());
  printf(" | wg total med %lld max %lld | first-start..last-end %lld\n", tot[tot.size() / 2], tot.back(), t1 - t0);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main(int argc, char** argv) {
  const int D = 1280, H = 20, R = argc > 1 ? atoi(argv[1]) : 8, L = 8, V = 51866, Ta = 1500;
  const int LR = argc > 2 ? atoi(argv[2]) : L;      // layers actually rotated over (1-2: MALL / L2 resident)
  printf("rows %d, rotating over %d layer copies\n", R, LR);
  const int iters = 40;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long long* d_probe; CK(hipMalloc(&d_probe, 8192 * 8 * 8)); CK(hipMemset(d_probe, 0, 8192 * 8 * 8));

  half_t *W, *xh, *y, *kv, *q, *att; float *xf, *lnw, *lnb, *bias, *resid, *logits, *part_o, *part_ml;
  const size_t wl = (size_t)14 * D * D;                 // one layer's worth, rotated over L layers
  CK(hipMalloc(&W, (wl * L + (size_t)V * D) * 2)); fill_half(W, wl * L + (size_t)V * D, 0.05f);
  CK(hipMalloc(&xh, (size_t)R * 4 * D * 2)); fill_half(xh, (size_t)R * 4 * D, 1.0f);
  CK(hipMalloc(&y, (size_t)R * 4 * D * 2));
  CK(hipMalloc(&xf, (size_t)R * D * 4)); fill_float(xf, (size_t)R * D, 2.0f);
  CK(hipMalloc(&resid, (size_t)R * D * 4)); fill_float(resid, (size_t)R * D, 2.0f);
  CK(hipMalloc(&lnw, D * 4)); fill_float(lnw, D, 1.0f);
  CK(hipMalloc(&lnb, D * 4)); fill_float(lnb, D, 1.0f);
  CK(hipMalloc(&bias, (size_t)V * 4)); fill_float(bias, V, 0.1f);
  CK(hipMalloc(&logits, (size_t)R * V * 4));
  CK(hipMalloc(&q, (size_t)R * D * 2)); fill_half(q, (size_t)R * D, 1.0f);
  CK(hipMalloc(&att, (size_t)R * D * 2));
  const size_t kvl = (size_t)R * Ta * 2 * D;
  CK(hipMalloc(&kv, kvl * L * 2)); fill_half(kv, kvl * L, 1.0f);
  CK(hipMalloc(&part_o, (size_t)R * H * 16 * 64 * 4)); fill_float(part_o, (size_t)R * H * 16 * 64, 1.0f);
  CK(hipMalloc(&part_ml, (size_t)R * H * 16 * 2 * 4)); fill_float(part_ml, (size_t)R * H * 16 * 2, 1.0f);

  for (int wgs : {160, 640}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, st, (int*)nullptr);
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel %d wgs: %.2f us/launch\n", wgs, ms * 1e3f / 200);
  }
  {  // the same empty launches replayed from a hipGraph
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(320), dim3(256), 0, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel 320 wgs, hipGraph replay: %.2f us/launch\n", ms * 1e3f / 1000);
  }
  struct Case { const char* name; int pro, epi, N, K; size_t woff; };
  Case cases[] = {
    {"gemv LN->qkv (3D x D)", whk::PRO_LN, whk::EPI_STORE, 3 * D, D, 0},
    {"gemv LN->cq store (DxD)", whk::PRO_LN, whk::EPI_STORE, D, D, (size_t)5 * D * D},
    {"gemv plain->store (DxD)", whk::PRO_PLAIN, whk::EPI_STORE, D, D, (size_t)3 * D * D},
    {"gemv plain->store (Dx4D)", whk::PRO_PLAIN, whk::EPI_STORE, D, 4 * D, (size_t)10 * D * D},
    {"gemv plain->out resid (DxD)", whk::PRO_PLAIN, whk::EPI_RESID, D, D, (size_t)3 * D * D},
    {"gemv combine->cout resid", whk::PRO_COMBINE, whk::EPI_RESID, D, D, (size_t)4 * D * D},
    {"gemv LN->fc1 gelu (4D x D)", whk::PRO_LN, whk::EPI_GELU, 4 * D, D, (size_t)6 * D * D},
    {"gemv plain->fc2 resid (Dx4D)", whk::PRO_PLAIN, whk::EPI_RESID, D, 4 * D, (size_t)10 * D * D},
  };
  const char* vnames[] = {"heuristic", "4w LPR8", "4w LPR16", "8w LPR8", "16w LPR8", "8w LPR16", "8w GS2", "16w GS2", "16w GS4", "", "",
                          "MF 4w", "MF 8w", "MF 8w GS2", "MF 16w GS4", "MF 16w", "MF 16w GS2", "MF stream"};
  for (const Case& c : cases) for (int variant = 0; variant <= 17; ++variant) {
    if (variant == 9 || variant == 10 || variant == 2 || variant == 5 || variant == 7) continue;
    if (variant == 4 && c.K < 4 * D) { }
    bool ok = true;
    for (int rep = 0; rep < 2 && ok; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) {
        whk::GemvArgs g; memset(&g, 0, sizeof(g));
        g.pro = c.pro; g.x = xh; g.x_ld = c.K; g.xf = xf; g.xf_ld = D; g.ln_w = lnw; g.ln_b = lnb;
        g.part_o = part_o; g.part_ml = part_ml; g.splits = 3; g.H = H;
        g.W = W + wl * (i % LR) + c.woff; g.bias = bias; g.N = c.N; g.K = c.K; g.R = R;
        g.epi = c.epi; g.y = y; g.y_ld = c.N; g.resid = resid; g.resid_ld = D;
        g.probe = (argc > 4 && atoi(argv[4]) == variant) ? d_probe : nullptr; g.variant = variant;
        if (whk::launch_gemv(g, 1, st) != hipSuccess) { ok = false; (void)hipGetLastError(); break; }
      }
      if (rep == 1 && ok) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    if (!ok) continue;
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    char verdict[64] = "";
    if (c.epi == whk::EPI_STORE && (c.pro == whk::PRO_LN || c.pro == whk::PRO_PLAIN)) {
      // numerics of the last launch (layer (iters-1) % LR) against a double-precision host reference
      const size_t woff = wl * ((iters - 1) % LR) + c.woff;
      std::vector<half_t> hw((size_t)c.N * c.K), hy((size_t)R * c.N), hx((size_t)R * c.K);
      std::vector<float> hxf((size_t)R * D), hlw(D), hlb(D), hb(c.N);
      CK(hipMemcpy(hw.data(), W + woff, hw.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx.data(), xh, hx.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hxf.data(), xf, hxf.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hlw.data(), lnw, D * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hlb.data(), lnb, D * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), bias, c.N * 4, hipMemcpyDeviceToHost));
      double maxerr = 0;
      for (int r = 0; r < R; ++r) {
        std::vector<double> xin(c.K);
        if (c.pro == whk::PRO_LN) {
          double m = 0, v = 0;
          for (int k = 0; k < c.K; ++k) m += hxf[(size_t)r * D + k];
          m /= c.K;
          for (int k = 0; k < c.K; ++k) v += (hxf[(size_t)r * D + k] - m) * (hxf[(size_t)r * D + k] - m);
          const double rstd = 1.0 / sqrt(v / c.K + 1e-5);
          for (int k = 0; k < c.K; ++k) xin[k] = (double)(half_t)(float)((hxf[(size_t)r * D + k] - m) * rstd * hlw[k] + hlb[k]);
        } else {
          for (int k = 0; k < c.K; ++k) xin[k] = (double)hx[(size_t)r * c.K + k];
        }
        for (int n = 0; n < c.N; n += 7) {
          double acc = hb[n];
          for (int k = 0; k < c.K; ++k) acc += xin[k] * (double)hw[(size_t)n * c.K + k];
          maxerr = fmax(maxerr, fabs(acc - (double)hy[(size_t)r * c.N + n]));
        }
      }
      snprintf(verdict, sizeof verdict, "  max|err| %.4f %s", maxerr, maxerr < 0.02 ? "ok" : "MISMATCH");
    }
    printf("%-30s %-10s %7.2f us/launch %7.0f GB/s%s\n", c.name, vnames[variant], ms * 1e3f / iters, (double)c.N * c.K * 2 / (ms * 1e3f / iters) * 1e-3, verdict);
    if (argc > 4 && atoi(argv[4]) == variant) report("   phases", ms * 1e3f / iters, (double)c.N * c.K * 2, d_probe, 80, 7);
  }
  {  // logits
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) {
        whk::GemvArgs g; memset(&g, 0, sizeof(g));
        g.pro = whk::PRO_LN; g.xf = xf; g.xf_ld = D; g.ln_w = lnw; g.ln_b = lnb;
        g.W = W + wl * L; g.N = V; g.K = D; g.R = R; g.epi = whk::EPI_F32; g.y = logits; g.y_ld = V; g.probe = d_probe;
        g.variant = argc > 3 ? atoi(argv[3]) : 0;
        CK(whk::launch_gemv(g, 1, st));
      }
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int ngroups = (V + 7) / 8, gp = (ngroups + 1023) / 1024;
    report("gemv LN->logits (V x D)", ms * 1e3f / 10, (double)V * D * 2, d_probe, (ngroups + gp - 1) / gp, 7);
  }
  for (int S = 3; S <= 8; ++S) {   // cross attention
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) {
        whk::DecAttnArgs a; memset(&a, 0, sizeof(a));
        a.q = q; a.q_ld = D; a.k = kv + kvl * (i % LR); a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
        a.v = kv + kvl * (i % LR) + D; a.v_ld = 2 * D; a.v_bs = a.k_bs;
        a.H = H; a.R = R; a.kv_group = 1; a.Tk = Ta; a.splits = S; a.out = att; a.o_ld = D; a.part_o = part_o; a.part_ml = part_ml;
        a.probe = d_probe;
        CK(whk::launch_attn_decode(a, 1, st));
      }
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    char nm[64]; snprintf(nm, sizeof nm, "attn_decode cross S=%d", S);
    report(nm, ms * 1e3f / iters, (double)R * Ta * 2 * D * 2, d_probe, S * H * R, 5);
  }
  for (int S = 3; S <= 6; S += 3) {   // head-major cross K/V: [row][head][key][64] for K, then the same for V
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1) CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) {
        whk::DecAttnArgs a; memset(&a, 0, sizeof(a));
        half_t* base = kv + kvl * (i % LR);
        a.q = q; a.q_ld = D; a.k = base; a.k_ld = 64; a.k_bs = (int64_t)Ta * D; a.kv_hs = (int64_t)Ta * 64;
        a.v = base + (size_t)R * Ta * D; a.v_ld = 64; a.v_bs = a.k_bs;
        a.H = H; a.R = R; a.kv_group = 1; a.Tk = Ta; a.splits = S; a.out = att; a.o_ld = D; a.part_o = part_o; a.part_ml = part_ml;
        a.probe = d_probe;
        CK(whk::launch_attn_decode(a, 1, st));
      }
      if (rep == 1) CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    char nm[64]; snprintf(nm, sizeof nm, "attn cross head-major S=%d", S);
    report(nm, ms * 1e3f / iters, (double)R * Ta * 2 * D * 2, d_probe, S * H * R, 5);
  }
  return 0;
}
