// tools/probe_fused.cpp — developer probe (not part of the product): the fused attention launches of csrc/xattn.hip on
// large-v3 shapes (R = 8 rows, fp16), each as a chain of 32 dependent launches replayed from a hipGraph (rotating over 8
// layer copies: HBM-cold), against the two-launch forms on the same buffers; with -DWH_PROBE the per-role time line of the
// LAST launch (clock64 stamps, cycles from the auxiliary wave's entry; medians over workgroups):
//   aux wave 0:   0 entry | 1 published (producers) | 2 q fetched
//   K/V wave 0:   4 K/V requests issued | 5 past the hand-off barrier | 6 scores done (keys have arrived) | 7 output stored
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWH_PROBE -I include tools/probe_fused.cpp -o tools/probe_fused
//   WH_XATTN_VECTOR_POLL=1 / WH_SATTN_SCALAR_POLL=1 flip the poll paths
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../whisper_amd/csrc/xattn.hip"
#include "../whisper_amd/csrc/attention.hip"
#include "../whisper_amd/csrc/gemv.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void add_int_k(int* p, int v) { *p += v; }

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

static long long med(std::vector<long long>& v) { if (v.empty()) return -1; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int D = 1280, H = 20, R = 8, N = 32, Ta = 1500, C = 448, S = 3;
  const int L = getenv("PROBE_L") ? atoi(getenv("PROBE_L")) : 8;   // distinct layers the chain rotates through (8: 0.5 GB of cross K/V; 32: 2 GB, as the step)
  const int pos = argc > 1 ? atoi(argv[1]) : 112;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long long* d_probe; CK(hipMalloc(&d_probe, 4096 * 8 * 8)); CK(hipMemset(d_probe, 0, 4096 * 8 * 8));
  half_t *Wq, *Wqkv, *ckv, *sk, *sv, *att, *part_o, *qbuf; float *xf, *bias, *part_ml;
  unsigned long long *xg, *sg; int *d_tick, *d_pos, *d_err;
  CK(hipMalloc(&Wq, (size_t)L * D * D * 2)); fill_half(Wq, (size_t)L * D * D, 0.05f);
  CK(hipMalloc(&Wqkv, (size_t)L * 3 * D * D * 2)); fill_half(Wqkv, (size_t)L * 3 * D * D, 0.05f);
  CK(hipMalloc(&ckv, (size_t)L * R * Ta * 2 * D * 2)); fill_half(ckv, (size_t)L * R * Ta * 2 * D, 1.0f);
  CK(hipMalloc(&sk, (size_t)L * R * C * D * 2)); fill_half(sk, (size_t)L * R * C * D, 1.0f);
  CK(hipMalloc(&sv, (size_t)L * R * C * D * 2)); fill_half(sv, (size_t)L * R * C * D, 1.0f);
  CK(hipMalloc(&att, (size_t)R * D * 2)); CK(hipMalloc(&qbuf, (size_t)R * 3 * D * 2));
  CK(hipMalloc(&part_o, (size_t)48 * H * 16 * 64 * 4)); CK(hipMalloc(&part_ml, (size_t)48 * H * 16 * 2 * 4));
  CK(hipMalloc(&xf, (size_t)R * D * 4)); fill_float(xf, (size_t)R * D, 2.0f);
  CK(hipMalloc(&bias, (size_t)3 * D * 4)); fill_float(bias, 3 * D, 0.1f);
  CK(hipMalloc(&xg, (size_t)R * (D / 2) * 8)); CK(hipMemset(xg, 0, (size_t)R * (D / 2) * 8));
  CK(hipMalloc(&sg, (size_t)R * (3 * D / 2) * 8)); CK(hipMemset(sg, 0, (size_t)R * (3 * D / 2) * 8));
  unsigned long long* og; CK(hipMalloc(&og, (size_t)R * (D / 2) * 8)); CK(hipMemset(og, 0, (size_t)R * (D / 2) * 8));
  half_t* Wo; CK(hipMalloc(&Wo, (size_t)L * D * D * 2)); fill_half(Wo, (size_t)L * D * D, 0.05f);
  float* x2; CK(hipMalloc(&x2, (size_t)R * D * 4));
  const bool with_out = getenv("PROBE_NO_OUT") == nullptr;
  CK(hipMalloc(&d_tick, 256)); CK(hipMemset(d_tick, 0, 256)); d_err = d_tick + 16;
  CK(hipMalloc(&d_pos, 256)); CK(hipMemcpy(d_pos, &pos, 4, hipMemcpyHostToDevice));

  auto xargs = [&](int i) {
    whk::XAttnArgs a; memset(&a, 0, sizeof(a));
    const int l = i % L;
    a.xf = xf; a.xf_ld = D; a.W = Wq + (size_t)l * D * D; a.bias = bias; a.D = D; a.H = H; a.R = R;
    a.k = ckv + (size_t)l * R * Ta * 2 * D; a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
    a.v = ckv + (size_t)l * R * Ta * 2 * D + D; a.v_ld = 2 * D; a.v_bs = a.k_bs;
    a.Tk = Ta; a.splits = S; a.out = att; a.o_ld = D; a.part_o = part_o; a.part_ml = part_ml;
    a.qg = xg; a.d_tick = d_tick; a.epoch = i; a.layer = l; a.err = d_err; a.mode = whk::fused_mode(0);
    a.probe = i == N - 1 ? d_probe : nullptr;
    return a;
  };
  auto sargs = [&](int i) {
    whk::SAttnArgs a; memset(&a, 0, sizeof(a));
    const int l = i % L;
    a.xf = xf; a.xf_ld = D; a.W = Wqkv + (size_t)l * 3 * D * D; a.bias = bias; a.D = D; a.H = H; a.R = R;
    a.kcache = sk + (size_t)l * R * C * D; a.vcache = sv + (size_t)l * R * C * D; a.cache_bs = (int64_t)C * D;
    a.d_pos = d_pos; a.lag = nullptr; a.q_out = qbuf; a.out = att; a.o_ld = D;
    a.qg = sg; a.d_tick = d_tick; a.epoch = i; a.layer = l; a.err = d_err; a.mode = whk::fused_mode(1);
    if (with_out) { a.out_w = Wo + (size_t)l * D * D; a.out_b = bias; a.x_out = x2; a.og = og; }
    a.probe = i == N - 1 ? d_probe : nullptr;
    return a;
  };
  auto two_cross = [&](int i) {      // LN -> cq (gemv8) + attn_decode_kernel
    const int l = i % L;
    whk::GemvArgs g; memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_LN; g.xf = xf; g.xf_ld = D; g.ln_folded = 1; g.W = Wq + (size_t)l * D * D; g.bias = bias; g.N = D; g.K = D; g.R = R;
    g.epi = whk::EPI_STORE; g.y = qbuf; g.y_ld = D;
    if (whk::launch_gemv(g, 1, st) != hipSuccess) return false;
    whk::DecAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = qbuf; a.q_ld = D; a.k = ckv + (size_t)l * R * Ta * 2 * D; a.k_ld = 2 * D; a.k_bs = (int64_t)Ta * 2 * D;
    a.v = ckv + (size_t)l * R * Ta * 2 * D + D; a.v_ld = 2 * D; a.v_bs = a.k_bs;
    a.H = H; a.R = R; a.kv_group = 1; a.Tk = Ta; a.splits = S; a.out = att; a.o_ld = D; a.part_o = part_o; a.part_ml = part_ml;
    return whk::launch_attn_decode(a, 1, st) == hipSuccess;
  };
  auto two_self = [&](int i) {       // LN -> qkv + cache append (gemv8) + attn_decode_kernel over the cache
    const int l = i % L;
    whk::GemvArgs g; memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_LN; g.xf = xf; g.xf_ld = D; g.ln_folded = 1; g.W = Wqkv + (size_t)l * 3 * D * D; g.bias = bias; g.N = 3 * D; g.K = D; g.R = R;
    g.epi = whk::EPI_QKV; g.y = qbuf; g.y_ld = D; g.kcache = sk + (size_t)l * R * C * D; g.vcache = sv + (size_t)l * R * C * D;
    g.cache_bs = (int64_t)C * D; g.d_pos = d_pos; g.D = D;
    if (whk::launch_gemv(g, 1, st) != hipSuccess) return false;
    whk::DecAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = qbuf; a.q_ld = D; a.k = sk + (size_t)l * R * C * D; a.k_ld = D; a.k_bs = (int64_t)C * D;
    a.v = sv + (size_t)l * R * C * D; a.v_ld = D; a.v_bs = (int64_t)C * D;
    a.H = H; a.R = R; a.kv_group = 1; a.d_len = d_pos; a.len_plus = 1; a.splits = 1; a.out = att; a.o_ld = D;
    a.part_o = part_o; a.part_ml = part_ml;
    return whk::launch_attn_decode(a, 1, st) == hipSuccess;
  };

  auto out_proj = [&](int i) {       // attn.out + residual (gemv8 PRO_PLAIN, EPI_RESID) on the attention rows
    const int l = i % L;
    whk::GemvArgs g; memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_PLAIN; g.x = att; g.x_ld = D; g.W = Wo + (size_t)l * D * D; g.bias = bias; g.N = D; g.K = D; g.R = R;
    g.epi = whk::EPI_RESID; g.resid = x2; g.resid_ld = D;
    return whk::launch_gemv(g, 1, st) == hipSuccess;
  };
  struct Case { const char* name; int kind; };
  Case cases[] = {{"cross: two launches (LN->q + attention)", 0}, {"cross: fused xattn8", 1},
                  {"self:  two launches (LN->qkv + attention)", 2}, {"self:  fused sattn8", 3},
                  {"attn.out + residual alone", 4}, {"pair: fused sattn8, then attn.out", 5},
                  {"pair: attn.out, then fused xattn8", 6},
                  {"ONE launch: attn.out as phase 0 of xattn8", 7},
                  {"cross: fused xattn8, HEAD-MAJOR K / V layout", 8}};
  for (const Case& c : cases) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipMemset(d_probe, 0, 4096 * 8 * 8));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    bool ok = true;
    for (int i = 0; i < N; ++i) {
      if (c.kind == 0) ok = ok && two_cross(i);
      if (c.kind == 1) ok = ok && whk::launch_xattn8(xargs(i), st) == hipSuccess;
      if (c.kind == 2) ok = ok && two_self(i);
      if (c.kind == 3) ok = ok && whk::launch_sattn8(sargs(i), st) == hipSuccess;
      if (c.kind == 4) ok = ok && out_proj(i);
      if (c.kind == 5) { whk::SAttnArgs sa = sargs(i); sa.probe = nullptr; ok = ok && whk::launch_sattn8(sa, st) == hipSuccess && out_proj(i); }
      if (c.kind == 6) { whk::XAttnArgs xa = xargs(i); xa.probe = nullptr; ok = ok && out_proj(i) && whk::launch_xattn8(xa, st) == hipSuccess; }
      if (c.kind == 8) {          // K [row][head][key][64] and V likewise in the second half of the layer's block: contiguous 64 KB per split
        whk::XAttnArgs xa = xargs(i);
        half_t* base = ckv + (size_t)(i % L) * R * Ta * 2 * D;
        xa.k = base; xa.v = base + (size_t)R * Ta * D;
        xa.k_ld = 64; xa.v_ld = 64; xa.k_bs = (int64_t)H * Ta * 64; xa.v_bs = xa.k_bs; xa.kv_hs = (int64_t)Ta * 64;
        ok = ok && whk::launch_xattn8(xa, st) == hipSuccess;
      }
      if (c.kind == 7) {
        whk::XAttnArgs xa = xargs(i);
        xa.att_in = att; xa.out_w = Wo + (size_t)(i % L) * D * D; xa.out_b = bias; xa.x_io = xf; xa.pflags = og;
        ok = ok && whk::launch_xattn8(xa, st) == hipSuccess;
      }
    }
    hipLaunchKernelGGL(add_int_k, dim3(1), dim3(1), 0, st, d_tick, N);
    CK(hipStreamEndCapture(st, &g));
    if (!ok) { printf("%s: launch failed\n", c.name); continue; }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    int err = 0; CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
    printf("%-46s %6.2f us per link%s (pos %d, hand-off timeouts %d)\n", c.name, best * 1e3f / N, c.kind >= 5 ? " (= per PAIR)" : "", pos, err);
    if (c.kind == 1 || c.kind == 3 || c.kind == 7 || c.kind == 8) {
      const int nwg = c.kind != 3 ? S * H * R : 3 * D / 8;
      std::vector<long long> p((size_t)nwg * 8);
      CK(hipMemcpy(p.data(), d_probe, p.size() * 8, hipMemcpyDeviceToHost));
      const int nprod = c.kind != 3 ? D / 8 : nwg;              // producers: first D/8 (cross) / all (self)
      const int cons0 = c.kind != 3 ? 0 : nwg - H * R;           // consumers: all (cross) / last H*R (self)
      // stamps are wall_clock64() (100 MHz, chip-wide): absolute times from the first workgroup's entry, in microseconds
      long long t0 = 0;
      for (int w = 0; w < nwg; ++w) { const long long e = p[(size_t)w * 8]; if (e && (!t0 || e < t0)) t0 = e; }
      std::vector<long long> ent, pub, fetch, kvis, bar, sco, end;
      for (int w = 0; w < nwg; ++w) {
        const long long* q = &p[(size_t)w * 8];
        if (!q[0]) continue;
        ent.push_back(q[0] - t0);
        if (w < nprod && q[1]) pub.push_back(q[1] - t0);
        if (w >= cons0 && q[2]) fetch.push_back(q[2] - t0);
        if (w >= cons0 && q[4]) kvis.push_back(q[4] - t0);
        if (w >= cons0 && q[5]) bar.push_back(q[5] - t0);
        if (w >= cons0 && q[6] && !(c.kind == 3 && with_out && w < D / 8)) sco.push_back(q[6] - t0);
        if (w >= cons0 && q[7]) end.push_back(q[7] - t0);
      }
      auto row = [&](const char* name, std::vector<long long>& v) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        const size_t n = v.size();
        printf("    %-26s min %5.2f  p10 %5.2f  median %5.2f  p90 %5.2f  max %5.2f us  (%zu workgroups)\n", name, v[0] * 0.01,
               v[n / 10] * 0.01, v[n / 2] * 0.01, v[n * 9 / 10] * 0.01, v[n - 1] * 0.01, n);
      };
      if (c.kind == 3) {                                          // self attention: publish time by third of the grid
        for (int part = 0; part < 3; ++part) {
          std::vector<long long> e3, p3;
          for (int w = part * (nwg / 3); w < (part + 1) * (nwg / 3); ++w) {
            const long long* q = &p[(size_t)w * 8];
            if (!q[0]) continue;
            e3.push_back(q[0] - t0);
            if (q[1]) p3.push_back(q[1] - t0);
          }
          char nm[64];
          snprintf(nm, sizeof(nm), "  entry, wg third %d", part); row(nm, e3);
          snprintf(nm, sizeof(nm), "  published, third %d", part); row(nm, p3);
        }
      }
      if (c.kind == 7) {
        std::vector<long long> p0;
        for (int w = 0; w < D / 8; ++w) { const long long* q = &p[(size_t)w * 8]; if (q[0] && q[3]) p0.push_back(q[3] - t0); }
        row("attn.out rows published", p0);
      }
      row("entry", ent); row("q published", pub); row("q fetched", fetch); row("K/V requested", kvis);
      row("past hand-off barrier", bar); row("scores done", sco); row("stored", end);
      if (c.kind == 3 && with_out) {
        std::vector<long long> outp;
        for (int w = 0; w < D / 8; ++w) { const long long* q = &p[(size_t)w * 8]; if (q[0] && q[6]) outp.push_back(q[6] - t0); }
        row("output projection stored", outp);
      }
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
