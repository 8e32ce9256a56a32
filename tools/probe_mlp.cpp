// tools/probe_mlp.cpp — developer probe (not part of the product): the MLP of one large-v3 decoder layer at 8 rows
// (LN + FC1 + GELU, then FC2 + residual; whisper/model.py:155-157,170) as ONE launch against the product's TWO launches.
//
// Why.  FC2's 13.1 MB of weights do not depend on FC1's output.  In the two-launch form they are requested only after
// FC1 has ended AND the launch boundary has passed (~1.9 us) AND the new launch has ramped up.  Here every workgroup asks
// for its FC2 slice at kernel entry, right behind its FC1 slice, so that the FC2 stream runs under FC1's LayerNorm, MFMAs
// and GELU; what is left on the critical path between the two matrix products is the hand-off of FC1's 8 x 5120 fp16
// outputs (80 KB) from all 256 workgroups to all 256 workgroups:
//   producer  40 lanes store the workgroup's 8 x 20 outputs as 8-byte write-through (agent-scope, sc1) stores, the storing
//             wave drains vmcnt, one lane stores the workgroup's flag {tag} (MI355X_MICROARCH.md "valid forms": sc1 payload
//             -> asm vmcnt(0) -> sc1 flag);
//   consumer  4 waves poll the 256 flags (one 8-byte agent-scope load per lane per pass, bounded), workgroup barrier,
//             then every wave fetches its x fragments with `sc0 sc1` 16-byte loads (L1-bypassing: the stores were
//             write-through, so no acquire fence is needed) straight into MFMA fragment layout.
// The tag is (*d_tick + link + 1): a device counter bumped once per graph replay, so a flag of an earlier replay never
// matches.  Same MFMA fragment maps, same order of sums as gemv8_kernel: the result must be BIT-IDENTICAL to the two
// launches (checked).  Reports us per MLP for both forms as 32-link hipGraph chains rotating over 8 layers of weights,
// and the fused kernel's time line (wall_clock64, 100 MHz) over workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_mlp.cpp -o tools/probe_mlp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../whisper_amd/csrc/gemv.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;

struct MlpArgs {
  const float* xf; int64_t xf_ld;        // residual stream in, fp32 [R][D]; also FC2's residual operand
  float* x_out;                          // residual stream out (may alias xf: written only after every workgroup's flag is up)
  const half_t* W1; const float* b1;     // [4D][D] (LayerNorm affine folded in), [4D]
  const half_t* W2; const float* b2;     // [D][4D], [D]
  half_t* h;                             // [8][4D] hand-off buffer
  u64* flags;                            // [256] one tag per producer workgroup
  const int* d_tick; int epoch;
  int* err;
  int R, D;
  long long* probe;                      // [256][8] wall_clock64 stamps or null
  int late_w2;                           // 1: request the FC2 slice only after FC1's flag is up (A/B: what the early request buys)
};

#define STAMP(i) do { if (a.probe && tid == 0) a.probe[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
constexpr int MLP_MAX_SPINS = 1 << 14;

__device__ __forceinline__ half8v load_b128_sc(const void* p) {
  half8v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// D = 1280, 256 workgroups x 16 waves.  Phase 1 = gemv8_kernel<PRO_LN, GS 3, KS 4, NU 5, XW 4> with fw = 20; phase 2 =
// gemv8_kernel<PRO_PLAIN, GS 1, KS 16, NU 5> with fw = 5.
__global__ __launch_bounds__(1024) void mlp8_kernel(MlpArgs a) {
  pin_kernargs(a);
  constexpr int XW = 4, KS = 4, NU = 5, MW = 12, FRAG = KS * NU * 64;
  constexpr int KS2 = 16, N1 = 5120, FW1 = 20, FW2 = 5;
  __shared__ float red[MW][8][8];
  __shared__ __attribute__((aligned(16))) half8v xfrag[FRAG];
  __shared__ __attribute__((aligned(16))) half_t ysh[8][24];
  __shared__ float red2[KS2][8][8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.D, nblk = K >> 6, R = a.R;
  const int w = blockIdx.x;
  STAMP(0);
  const int vtick = load_agent_int(a.d_tick);
  const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;

  half8v wa[NU], wa2[NU];
  const int mw = wave - XW, kw = mw & 3;
  // the FC2 slice of this workgroup (5 rows x 5120, all 16 waves) and the epilogue operands: requested by every wave right
  // behind its own phase-1 requests.  Each role keeps its requests and its first use in ONE straight-line region, so that
  // the compiler can count (vmcnt(n)) instead of draining everything (vmcnt(0)) before the LayerNorm.
  int n2 = w * FW2 + idx; if (n2 > w * FW2 + FW2 - 1) n2 = w * FW2 + FW2 - 1;
  const uint32_t lane_off2 = ((uint32_t)n2 * (uint32_t)N1 + (uint32_t)koff) * 2u;
  float e_b1 = 0.f, e_b2 = 0.f, e_res = 0.f;
#define ISSUE_PHASE2_REQUESTS()                                                                                          \
  do {                                                                                                                   \
    ISSUE_FENCE();                                                                                                       \
    if (!a.late_w2) {                                                                                                    \
      _Pragma("unroll") for (int u = 0; u < NU; ++u)                                                                     \
        wa2[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)a.W2 + (size_t)(wave + KS2 * u) * 128 + lane_off2));       \
    }                                                                                                                    \
    ISSUE_FENCE();                                                                                                       \
    if (tid < 192) {                                                                                                     \
      const int es = tid >> 6, ej = tid & 7;                                                                             \
      if (es * 8 + ej < FW1) e_b1 = a.b1[w * FW1 + es * 8 + ej];                                                         \
    }                                                                                                                    \
    if (tid < 64) {                                                                                                      \
      const int er = tid >> 3, ej = tid & 7;                                                                             \
      if (ej < FW2) { e_b2 = a.b2[w * FW2 + ej]; e_res = a.xf[(int64_t)(er < R ? er : R - 1) * a.xf_ld + w * FW2 + ej]; } \
    }                                                                                                                    \
    ISSUE_FENCE();                                                                                                       \
  } while (0)

  if (wave >= XW) {
    int n = w * FW1 + (mw >> 2) * 8 + idx; if (n > w * FW1 + FW1 - 1) n = w * FW1 + FW1 - 1;
    const uint32_t lane_off = ((uint32_t)n * (uint32_t)K + (uint32_t)koff) * 2u;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int blk = kw + KS * u; if (blk > nblk - 1) blk = nblk - 1;
      wa[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)a.W1 + (size_t)blk * 128 + lane_off));
    }
    ISSUE_PHASE2_REQUESTS();
  } else {
    // ---- phase 1: LayerNorm (waves 0-3: rows wave, wave + 4) -> fragments
    float4v v[2][NU];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave + XW * i;
      const char* src = (const char*)a.xf + (size_t)(r < R ? r : R - 1) * (size_t)a.xf_ld * 4;
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
        v[i][j] = *(const float4v*)(src + (uint32_t)k * 4u);
      }
    }
    ISSUE_PHASE2_REQUESTS();
    const float invK = 1.0f / (float)K;
    const uint32_t fbase = (uint32_t)((lane >> 4) * NU * 64 + 16 * ((lane >> 1) & 3) + 8 * ((lane >> 3) & 1)) * 16u + (uint32_t)(lane & 1) * 8u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave + XW * i;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
        sum += ((j * 64 + lane) * 4 < K) ? t : 0.f;
      }
      const float mean = wave_sum(sum) * invK;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        if ((j * 64 + lane) * 4 < K) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
        }
      }
      const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
      const uint32_t rbase = fbase + (uint32_t)((r & 7) * 16);
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const bool on = (j * 64 + lane) * 4 < K;
        half4v o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = on ? (half_t)((v[i][j][e] - mean) * rstd) : (half_t)0.f;
        *(half4v*)((char*)xfrag + rbase + (uint32_t)(j * 1024)) = o4;
      }
    }
  }
  __syncthreads();                                                   // B1
  if (wave >= XW) {
    const bool diag = (lane >> 5) == ((lane >> 3) & 1);
    float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xfrag[(kw * NU + u) * 64 + lane], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z = diag ? acc[e] : 0.f;
      z += lane_xor8(z);
      float p, q; lane_swap32(z, p, q);
      acc[e] = p + q;
    }
    if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[mw][4 * (lane >> 4) + e][lane & 7] = acc[e];
    }
  }
  __syncthreads();                                                   // B2
  if (tid < 192) {
    const int es = tid >> 6, er = (tid >> 3) & 7, ej = tid & 7;
    if (es * 8 + ej < FW1) {
      float s = e_b1;
#pragma unroll
      for (int k = 0; k < KS; ++k) s += red[es * KS + k][ej][er];
      ysh[er][es * 8 + ej] = (half_t)gelu_erf(s);
    }
  }
  __syncthreads();                                                   // B3
  const unsigned tag = (unsigned)(uniform(vtick) + a.epoch + 1);
  STAMP(1);
  // ---- publish: 8 rows x 20 features = 40 eight-byte write-through stores, drained, then the flag
  if (wave == 0) {
    if (tid < 40) {
      const int r = tid / 5, ch = tid - r * 5;
      if (r < R) {
        const u64 x = *(const u64*)&ysh[r][4 * ch];
        __hip_atomic_store((u64*)(a.h + (size_t)r * N1 + w * FW1 + 4 * ch), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) __hip_atomic_store(a.flags + w, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  STAMP(2);
  if (a.late_w2) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
      wa2[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)a.W2 + (size_t)(wave + KS2 * u) * 128 + lane_off2));
    ISSUE_FENCE();
  }
  // ---- consume: 4 waves poll the 256 flags
  if (wave < 4) {
    const u64* fp = a.flags + wave * 64 + lane;
    int spins = 0;
    for (;;) {
      const u64 f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((unsigned)f == tag)) break;
      if (++spins >= MLP_MAX_SPINS) { if (lane == 0 && a.err) atomicAdd(a.err, 1); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();                                                   // B4: every producer has published
  STAMP(3);
  half8v xb2[NU];
  {
    const int rr = idx < R ? idx : R - 1;
    const char* base = (const char*)a.h + ((size_t)rr * N1 + koff) * 2;
#pragma unroll
    for (int u = 0; u < NU; ++u) xb2[u] = load_b128_sc(base + (size_t)(wave + KS2 * u) * 128);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  STAMP(4);
  {
    const bool diag = (lane >> 5) == ((lane >> 3) & 1);
    float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa2[u], xb2[u], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z = diag ? acc[e] : 0.f;
      z += lane_xor8(z);
      float p, q; lane_swap32(z, p, q);
      acc[e] = p + q;
    }
    if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red2[wave][4 * (lane >> 4) + e][lane & 7] = acc[e];
    }
  }
  __syncthreads();                                                   // B5
  if (tid < 64) {
    const int er = tid >> 3, ej = tid & 7;
    if (ej < FW2 && er < R) {
      float s = e_b2;
#pragma unroll
      for (int k = 0; k < KS2; ++k) s += red2[k][ej][er];
      a.x_out[(int64_t)er * a.xf_ld + w * FW2 + ej] = e_res + s;
    }
  }
  STAMP(5);
}

__global__ void bump_kernel(int* p, int by) { if (threadIdx.x == 0) atomicAdd(p, by); }

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
  const int D = 1280, R = argc > 1 ? atoi(argv[1]) : 8, N = 32;
  const int L = getenv("PROBE_L") ? atoi(getenv("PROBE_L")) : 8;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t wl = (size_t)8 * D * D;                     // FC1 (4 D^2) + FC2 (4 D^2) per layer
  half_t *W, *h, *h2; float *x0, *xa, *xb, *b1, *b2; u64* flags; int *tick, *err; long long* probe;
  CK(hipMalloc(&W, wl * L * 2)); fill_half(W, wl * L, 0.05f);
  CK(hipMalloc(&h, (size_t)8 * 4 * D * 2)); CK(hipMemset(h, 0, (size_t)8 * 4 * D * 2));
  CK(hipMalloc(&h2, (size_t)8 * 4 * D * 2));
  CK(hipMalloc(&x0, (size_t)8 * D * 4)); fill_float(x0, (size_t)8 * D, 2.0f);
  CK(hipMalloc(&xa, (size_t)8 * D * 4)); CK(hipMalloc(&xb, (size_t)8 * D * 4));
  CK(hipMalloc(&b1, (size_t)4 * D * 4)); fill_float(b1, 4 * D, 0.1f);
  CK(hipMalloc(&b2, (size_t)D * 4)); fill_float(b2, D, 0.1f);
  CK(hipMalloc(&flags, 256 * 8)); CK(hipMemset(flags, 0, 256 * 8));
  CK(hipMalloc(&tick, 4)); CK(hipMemset(tick, 0, 4));
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  CK(hipMalloc(&probe, 256 * 8 * 8)); CK(hipMemset(probe, 0, 256 * 8 * 8));

  auto two_launch = [&](int i, float* x) -> bool {
    whk::GemvArgs g; memset(&g, 0, sizeof(g));
    const half_t* w1 = W + wl * (i % L); const half_t* w2 = w1 + (size_t)4 * D * D;
    g.pro = whk::PRO_LN; g.xf = x; g.xf_ld = D; g.ln_folded = 1; g.W = w1; g.bias = b1; g.N = 4 * D; g.K = D; g.R = R;
    g.epi = whk::EPI_GELU; g.y = h2; g.y_ld = 4 * D;
    if (whk::launch_gemv(g, 1, st) != hipSuccess) return false;
    memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_PLAIN; g.x = h2; g.x_ld = 4 * D; g.W = w2; g.bias = b2; g.N = D; g.K = 4 * D; g.R = R;
    g.epi = whk::EPI_RESID; g.resid = x; g.resid_ld = D;
    return whk::launch_gemv(g, 1, st) == hipSuccess;
  };
  auto fused = [&](int i, float* x, int late, long long* pr) {
    MlpArgs a; memset(&a, 0, sizeof(a));
    a.xf = x; a.xf_ld = D; a.x_out = x; a.W1 = W + wl * (i % L); a.b1 = b1; a.W2 = a.W1 + (size_t)4 * D * D; a.b2 = b2;
    a.h = h; a.flags = flags; a.d_tick = tick; a.epoch = i; a.err = err; a.R = R; a.D = D; a.probe = pr; a.late_w2 = late;
    hipLaunchKernelGGL(mlp8_kernel, dim3(256), dim3(1024), 0, st, a);
  };

  // ---- numerics: one MLP, both forms, from the same residual rows
  CK(hipMemcpyAsync(xa, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice, st));
  CK(hipMemcpyAsync(xb, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice, st));
  if (!two_launch(3, xa)) { printf("two-launch form failed to launch\n"); return 1; }
  fused(3, xb, 0, nullptr);
  hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, st, tick, N);
  CK(hipStreamSynchronize(st));
  {
    std::vector<float> ha((size_t)8 * D), hb((size_t)8 * D), h0((size_t)8 * D);
    CK(hipMemcpy(ha.data(), xa, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), xb, hb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), x0, h0.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, mv = 0; size_t nd = 0;
    for (size_t i = 0; i < (size_t)R * D; ++i) { mx = std::max(mx, (double)fabsf(ha[i] - hb[i])); mv = std::max(mv, (double)fabsf(ha[i] - h0[i])); nd += ha[i] != hb[i]; }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("numerics (R = %d): fused vs two launches: max |d| %.3g, %zu of %d values differ (MLP moved the rows by up to %.3g); hand-off time-outs %d\n",
           R, mx, nd, R * D, mv, herr);
  }

  // ---- timing: 32-link chains
  struct Form { const char* name; int kind; };
  Form forms[] = {{"two launches (product: LN+FC1+GELU | FC2+residual)", 0}, {"one launch, FC2 slice requested at entry", 1},
                  {"one launch, FC2 slice requested after the publish", 2}};
  for (const Form& f : forms) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipMemcpy(xa, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    bool ok = true;
    for (int i = 0; i < N; ++i) {
      if (f.kind == 0) ok = ok && two_launch(i, xa);
      else fused(i, xa, f.kind == 2, i == N - 1 ? probe : nullptr);
    }
    if (f.kind != 0) hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, st, tick, N);
    CK(hipStreamEndCapture(st, &g));
    if (!ok) { printf("%s: launch failed\n", f.name); continue; }
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-56s %6.2f us per MLP (32-link graph chain, %d layers of weights)", f.name, best * 1e3f / N, L);
    if (f.kind != 0) {
      std::vector<long long> p((size_t)256 * 8);
      CK(hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost));
      long long t0 = p[0];
      for (int wg = 0; wg < 256; ++wg) t0 = std::min(t0, p[(size_t)wg * 8]);
      printf(" | time-outs %d\n    time line, us after the first workgroup's entry (min / median / max over 256 workgroups):\n", herr);
      const char* names[] = {"entry", "FC1 + GELU done", "published (flag up)", "all 256 flags seen", "x fragments landed", "FC2 + residual stored"};
      for (int s = 0; s < 6; ++s) {
        std::vector<double> d;
        for (int wg = 0; wg < 256; ++wg) d.push_back((p[(size_t)wg * 8 + s] - t0) / 100.0);
        std::sort(d.begin(), d.end());
        printf("      %-24s %6.2f / %6.2f / %6.2f\n", names[s], d[0], d[128], d[255]);
      }
    } else printf("\n");
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
