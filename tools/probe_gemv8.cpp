// tools/probe_gemv8.cpp — developer probe (not part of the product): the six decode-step projections of one large-v3
// decoder layer (R = 8 rows, fp16), each as a chain of 64 dependent launches replayed from a hipGraph, rotating over 8
// layer copies (HBM-cold).  Reports us per link and, with -DWH_PROBE, the s_memtime phase medians per workgroup:
//   thread 0 of the workgroup (a prologue wave when the kernel has them, else weight wave 0):
//   0 entry | 1 own loads issued | 2 prologue arithmetic done, fragments written | 3 barrier, fragments read |
//   4 MFMAs + diagonal sum + partial sums written | 5 barrier | 6 epilogue stores issued
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWH_PROBE -I include tools/probe_gemv8.cpp -o tools/probe_gemv8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../whisper_amd/csrc/gemv.hip"

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

int main(int argc, char** argv) {
  const int D = 1280, H = 20, R = argc > 1 ? atoi(argv[1]) : 8, N = 64;
  const int L = getenv("PROBE_L") ? atoi(getenv("PROBE_L")) : 8;   // distinct layers of weights the chain rotates through (1: cache-resident)
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long long* d_probe; CK(hipMalloc(&d_probe, 8192 * 8 * 8)); CK(hipMemset(d_probe, 0, 8192 * 8 * 8));
  half_t *W, *xh, *y, *part_o; float *xf, *lnw, *lnb, *bias, *resid, *part_ml;
  const size_t wl = (size_t)14 * D * D;
  CK(hipMalloc(&W, wl * L * 2)); fill_half(W, wl * L, 0.05f);
  CK(hipMalloc(&xh, (size_t)R * 4 * D * 2)); fill_half(xh, (size_t)R * 4 * D, 1.0f);
  CK(hipMalloc(&y, (size_t)R * 4 * D * 2));
  CK(hipMalloc(&xf, (size_t)R * D * 4)); fill_float(xf, (size_t)R * D, 2.0f);
  CK(hipMalloc(&resid, (size_t)R * D * 4)); fill_float(resid, (size_t)R * D, 2.0f);
  CK(hipMalloc(&lnw, D * 4)); fill_float(lnw, D, 1.0f);
  CK(hipMalloc(&lnb, D * 4)); fill_float(lnb, D, 1.0f);
  CK(hipMalloc(&bias, (size_t)4 * D * 4)); fill_float(bias, 4 * D, 0.1f);
  CK(hipMalloc(&part_o, (size_t)R * H * 16 * 64 * 2)); fill_half(part_o, (size_t)R * H * 16 * 64, 1.0f);
  CK(hipMalloc(&part_ml, (size_t)R * H * 16 * 2 * 4)); fill_float(part_ml, (size_t)R * H * 16 * 2, 1.0f);

  struct Case { const char* name; int pro, epi, Nn, K; size_t woff; };
  Case cases[] = {
    {"LN->qkv (3D x D)", whk::PRO_LN, whk::EPI_STORE, 3 * D, D, 0},
    {"plain->out resid (DxD)", whk::PRO_PLAIN, whk::EPI_RESID, D, D, (size_t)3 * D * D},
    {"LN->cq store (DxD)", whk::PRO_LN, whk::EPI_STORE, D, D, (size_t)5 * D * D},
    {"combine->cout resid", whk::PRO_COMBINE, whk::EPI_RESID, D, D, (size_t)4 * D * D},
    {"LN->fc1 gelu (4D x D)", whk::PRO_LN, whk::EPI_GELU, 4 * D, D, (size_t)6 * D * D},
    {"plain->fc2 resid (Dx4D)", whk::PRO_PLAIN, whk::EPI_RESID, D, 4 * D, (size_t)10 * D * D},
  };
  // PROBE_FRAG=1: the PRO_PLAIN x rows in fragment order (kernels.h); checked once against the row-major launch
  // (the weights in fragment order were measured with an earlier form of this probe: profiles/r06_fragment_order.txt)
  const bool frag = getenv("PROBE_FRAG") && atoi(getenv("PROBE_FRAG")), fragx = frag;
  half_t* xh_frag = nullptr;
  if (frag) {
    const int Rp = (R + 7) / 8 * 8;
    CK(hipMalloc(&xh_frag, (size_t)Rp * 4 * D * 2)); CK(hipMemset(xh_frag, 0, (size_t)Rp * 4 * D * 2));
  }
  for (const Case& c : cases) for (int variant : {0, 99}) {
    if (variant == 99 && (getenv("PROBE_MFMA_ONLY") || frag)) continue;
    if (frag) {
      if (c.pro == whk::PRO_PLAIN) {
        std::vector<half_t> h((size_t)R * c.K), hf((size_t)((R + 7) / 8 * 8) * c.K, (half_t)0.f);
        CK(hipMemcpy(h.data(), xh, h.size() * 2, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r) for (int k = 0; k < c.K; ++k) hf[whk::frag_index(r, k, c.K)] = h[(size_t)r * c.K + k];
        CK(hipMemcpy(xh_frag, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
      }
      // one launch each way from the same inputs
      std::vector<float> res0((size_t)R * D), o0, o1;
      CK(hipMemcpy(res0.data(), resid, res0.size() * 4, hipMemcpyDeviceToHost));
      for (int way = 0; way < 2; ++way) {
        whk::GemvArgs a; memset(&a, 0, sizeof(a));
        a.pro = c.pro; a.x = way && fragx && c.pro == whk::PRO_PLAIN ? xh_frag : xh; a.x_frag = way && fragx && c.pro == whk::PRO_PLAIN; a.x_ld = c.K; a.xf = xf; a.xf_ld = D; a.ln_w = lnw; a.ln_b = lnb; a.ln_folded = 1;
        a.part_o = part_o; a.part_ml = part_ml; a.splits = 3; a.H = H;
        a.w_ordered = way && getenv("PROBE_WORD") ? atoi(getenv("PROBE_WORD")) : 0;
        a.W = W + c.woff; a.bias = bias; a.N = c.Nn; a.K = c.K; a.R = R;
        a.epi = c.epi; a.y = y; a.y_ld = c.Nn; a.resid = resid; a.resid_ld = D;
        CK(hipMemcpy(resid, res0.data(), res0.size() * 4, hipMemcpyHostToDevice));
        CK(whk::launch_gemv(a, 1, st)); CK(hipStreamSynchronize(st));
        std::vector<float>& o = way ? o1 : o0;
        if (c.epi == whk::EPI_RESID) { o.resize((size_t)R * D); CK(hipMemcpy(o.data(), resid, o.size() * 4, hipMemcpyDeviceToHost)); }
        else { std::vector<half_t> hy((size_t)R * c.Nn); CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost)); o.assign(hy.begin(), hy.end()); }
      }
      size_t nd = 0; for (size_t i = 0; i < o0.size(); ++i) nd += !(o0[i] == o1[i]);
      printf("%-26s fragment order vs row-major: %zu of %zu outputs differ\n", c.name, nd, o0.size());
      CK(hipMemcpy(resid, res0.data(), res0.size() * 4, hipMemcpyHostToDevice));
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    bool ok = true;
    for (int i = 0; i < N; ++i) {
      whk::GemvArgs a; memset(&a, 0, sizeof(a));
      a.pro = c.pro; a.x = xh; a.x_ld = c.K; a.xf = xf; a.xf_ld = D; a.ln_w = lnw; a.ln_b = lnb; a.ln_folded = 1;
      a.part_o = part_o; a.part_ml = part_ml; a.splits = 3; a.H = H;
      a.W = W + wl * (i % L) + c.woff; a.bias = bias; a.N = c.Nn; a.K = c.K; a.R = R;
      if (fragx && c.pro == whk::PRO_PLAIN) { a.x = xh_frag; a.x_frag = 1; }
      a.w_ordered = getenv("PROBE_WORD") ? atoi(getenv("PROBE_WORD")) : 0;
      a.epi = c.epi; a.y = y; a.y_ld = c.Nn; a.resid = resid; a.resid_ld = D;
      a.probe = i == N - 1 ? d_probe : nullptr;
      a.variant = variant == 99 ? -1 : variant;      // -1: the v_dot2 kernels
      hipError_t e = whk::launch_gemv(a, 1, st);
      if (e != hipSuccess) ok = false;
    }
    CK(hipStreamEndCapture(st, &g));
    if (!ok) { printf("%s: launch failed\n", c.name); continue; }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    const int nwg = 160;
    std::vector<long long> p((size_t)nwg * 8);
    CK(hipMemcpy(p.data(), d_probe, p.size() * 8, hipMemcpyDeviceToHost));
    printf("%-26s %-6s %6.2f us/link | phases:", c.name, variant == 99 ? "dot2" : "mfma8", best * 1e3f / N);
    const int npt = 7;
    for (int i = 1; i < npt; ++i) {
      std::vector<long long> d;
      for (int w = 0; w < nwg; ++w) d.push_back(p[(size_t)w * 8 + i] - p[(size_t)w * 8 + i - 1]);
      std::sort(d.begin(), d.end());
      printf(" %5lld", d[d.size() / 2]);
    }
    std::vector<long long> tot;
    for (int w = 0; w < nwg; ++w) tot.push_back(p[(size_t)w * 8 + npt - 1] - p[(size_t)w * 8]);
    std::sort(tot.begin(), tot.end());
    printf(" | wg total med %lld max %lld\n", tot[tot.size() / 2], tot.back());
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
