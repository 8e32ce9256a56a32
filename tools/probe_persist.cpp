// tools/probe_persist.cpp — developer probe (not part of the product): ONE persistent launch per decode layer, with an
// XCD-hierarchical grid barrier between the projections and the next projection's weights requested before the barrier
// (run-ahead), against the same six projections as six dependent launches replayed from a hipGraph.
//
// The six phases are the projections of one large-v3 decoder layer at 8 rows (QKV 3D x D, out D x D, cross-Q D x D,
// merge+out D x D, FC1 4D x D, FC2 D x 4D; the two attention kernels are left out on both sides): every workgroup
// streams its N/256 weight rows (nt loads), stages the full x vector of the 8 rows (produced by ALL workgroups of the
// previous phase) in LDS, multiplies with v_dot2, writes its slice of y.  That is the all-to-all edge every fusion of
// two decode-step kernels has to cross.  Both forms run the same body; the persistent one replaces the kernel boundary
// by   release fence -> per-XCD counter -> top counter -> per-XCD generation word -> acquire fence   (the barrier-xcd
// recipe of MI355X_MICROARCH.md) and may issue the weight loads of phase p + 1 before it arrives at the barrier.
// All spins are bounded; the probe aborts instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_persist.cpp -o tools/probe_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#include "../whisper_amd/csrc/common.h"     // wave_sum (DPP / permlane reductions), half types

constexpr int D = 1280, ROWS = 8, NWG = 256, NT = 256, NPH = 6;
struct Phase { int N, K; };
__constant__ Phase c_ph[NPH];
static const Phase h_ph[NPH] = {{3 * D, D}, {D, D}, {D, D}, {D, D}, {4 * D, D}, {D, 4 * D}};

struct Sync {                             // all words monotonic; zeroed once per launch batch
  unsigned grp_cnt[8]; unsigned top_cnt; unsigned top_gen; unsigned grp_gen[8]; unsigned abort_flag;
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread per workgroup; returns false when a spin ran out (the kernel then leaves)
__device__ bool grid_barrier(Sync* s, unsigned gen, int grp, int members, int ngroups) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const unsigned prev = atomicAdd(&s->grp_cnt[grp], 1u);
  bool ok = true;
  if (prev == gen * members + members - 1) {          // last of the group
    const unsigned p2 = atomicAdd(&s->top_cnt, 1u);
    if (p2 == gen * ngroups + ngroups - 1) __hip_atomic_store(&s->top_gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (ld_agent(&s->top_gen) < gen + 1) { __builtin_amdgcn_s_sleep(1); if (++spins > 2000000) { ok = false; break; } }
    __hip_atomic_store(&s->grp_gen[grp], gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    int spins = 0;
    while (ld_agent(&s->grp_gen[grp]) < gen + 1) { __builtin_amdgcn_s_sleep(1); if (++spins > 2000000) { ok = false; break; } }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (!ok) atomicExch(&s->abort_flag, 1u);
  return ok;
}

// The same barrier without fences, for the tuned form: the caller has already made its stores visible (sc1 write-through
// stores + a counted vmcnt wait that leaves the run-ahead loads in flight); the poll result is consumed before
// `buffer_inv sc1` invalidates the non-coherent lines, so no wait on the vector-memory counter is needed here either.
__device__ bool grid_barrier_nofence(Sync* s, unsigned gen, int grp, int members, int ngroups) {
  const unsigned prev = __hip_atomic_fetch_add(&s->grp_cnt[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = true;
  if (prev == gen * members + members - 1) {
    const unsigned p2 = __hip_atomic_fetch_add(&s->top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p2 == gen * ngroups + ngroups - 1) __hip_atomic_store(&s->top_gen, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (ld_agent(&s->top_gen) < gen + 1) { if (++spins > 20000000) { ok = false; break; } }
    __hip_atomic_store(&s->grp_gen[grp], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    int spins = 0;
    while (ld_agent(&s->grp_gen[grp]) < gen + 1) { if (++spins > 20000000) { ok = false; break; } }
  }
  asm volatile("buffer_inv sc1" ::: "memory");
  if (!ok) atomicExch(&s->abort_flag, 1u);
  return ok;
}

// A phase in the form of the real kernels: the 4 waves of a workgroup own whole features (wave w: features w, w + 4, ...),
// the 64 lanes of a wave split K in 16-byte units, partial sums meet by a wave reduction.  FW = features per
// workgroup, K = reduction length; FPW x UPL weight units of 16 bytes per lane, all requested up front.
template <int FW, int K> struct Shape {
  static constexpr int FPW = (FW + 3) / 4, UPR = K / 8, UPL = (UPR + 63) / 64, NU = FPW * UPL;
};
constexpr int MAXU = 20;
template <int FW, int K>
__device__ __forceinline__ void issue_weights_t(const half_t* W, int wg, half8v* w) {
  typedef Shape<FW, K> S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < S::FPW; ++i) {
    int f = wave + 4 * i; if (f > FW - 1) f = FW - 1;
    const half_t* row = W + ((size_t)wg * FW + f) * K;
#pragma unroll
    for (int j = 0; j < S::UPL; ++j) {
      int u = lane + 64 * j; if (u > S::UPR - 1) u = S::UPR - 1;
      w[i * S::UPL + j] = __builtin_nontemporal_load((const half8v*)(row + (size_t)u * 8));
    }
  }
}
template <int FW, int K, bool SC1>
__device__ __forceinline__ void phase_body_t(int wg, const half8v* w, const half_t* x, int xld, half_t* y, int yld, half_t* xs) {
  typedef Shape<FW, K> S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ROWS * S::UPR; i += NT) {
    const int r = i / S::UPR, u = i - r * S::UPR;
    *(half8v*)(xs + (size_t)r * K + u * 8) = *(const half8v*)(x + (size_t)r * xld + u * 8);
  }
  __syncthreads();
  const float scale = 1.0f / (0.029f * __builtin_sqrtf((float)K));     // keeps |y| ~ |x| over a long chain
#pragma unroll
  for (int i = 0; i < S::FPW; ++i) {
    const int f = wave + 4 * i;
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < S::UPL; ++j) {
      const int u = lane + 64 * j;
      if (u < S::UPR) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const half8v xv = *(const half8v*)(xs + (size_t)r * K + u * 8);
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            acc[r] = __builtin_amdgcn_fdot2(half2v{w[i * S::UPL + j][e], w[i * S::UPL + j][e + 1]}, half2v{xv[e], xv[e + 1]}, acc[r], false);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
    if (f < FW && lane < ROWS) {
      float v = acc[0];
#pragma unroll
      for (int r = 1; r < ROWS; ++r) v = lane == r ? acc[r] : v;
      half_t* dst = y + (size_t)lane * yld + wg * FW + f;
      const half_t hv = (half_t)(v * scale);
      if (SC1) {                                       // write-through to the coherence point: visible once acknowledged
        const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, hv);
        asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(dst), "v"(bits) : "memory");
      } else {
        *dst = hv;
      }
    }
  }
}
__device__ __forceinline__ void issue_weights(const half_t* W, int ph, int wg, half8v* w) {
  switch (ph) {
    case 0: issue_weights_t<15, D>(W, wg, w); break;
    case 4: issue_weights_t<20, D>(W, wg, w); break;
    case 5: issue_weights_t<5, 4 * D>(W, wg, w); break;
    default: issue_weights_t<5, D>(W, wg, w); break;
  }
}
template <bool SC1 = false>
__device__ __forceinline__ void phase_body(int ph, int wg, const half8v* w, const half_t* x, int xld, half_t* y, int yld, half_t* xs) {
  switch (ph) {
    case 0: phase_body_t<15, D, SC1>(wg, w, x, xld, y, yld, xs); break;
    case 4: phase_body_t<20, D, SC1>(wg, w, x, xld, y, yld, xs); break;
    case 5: phase_body_t<5, 4 * D, SC1>(wg, w, x, xld, y, yld, xs); break;
    default: phase_body_t<5, D, SC1>(wg, w, x, xld, y, yld, xs); break;
  }
}
// wait until this wave's stores are acknowledged while the `ph` weight requests issued after them stay in flight
__device__ __forceinline__ void wait_stores_keep_weights(int ph) {
  switch (ph) {
    case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Shape<15, D>::NU) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Shape<20, D>::NU) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Shape<5, 4 * D>::NU) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Shape<5, D>::NU) : "memory"); break;
  }
}

// one phase per launch (the chain of today's design)
__global__ __launch_bounds__(NT) void phase_kernel(const half_t* W, int ph, const half_t* x, half_t* y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = (half_t*)smem;
  half8v w[MAXU];
  issue_weights(W, ph, blockIdx.x, w);
  phase_body(ph, blockIdx.x, w, x, 4 * D, y, 4 * D, xs);
}

// the whole layer (x layers) in one launch; RUNAHEAD: the next phase's weights are requested before the barrier
template <bool RUNAHEAD>
__global__ __launch_bounds__(NT) void persistent_kernel(const half_t* const* Wl, int layers, half_t* buf0, half_t* buf1, Sync* s) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = (half_t*)smem;
  __shared__ int sh_ok;
  const int wg = blockIdx.x, grp = wg & 7;
  half8v w[MAXU];
  const int total = layers * NPH;
  if (RUNAHEAD) issue_weights(Wl[0], 0, wg, w);
  for (int p = 0; p < total; ++p) {
    const int ph = p % NPH;
    const half_t* x = (p & 1) ? buf1 : buf0;
    half_t* y = (p & 1) ? buf0 : buf1;
    if (!RUNAHEAD) issue_weights(Wl[p], ph, wg, w);
    phase_body(ph, wg, w, x, 4 * D, y, 4 * D, xs);
    if (p + 1 == total) break;
    if (RUNAHEAD) issue_weights(Wl[p + 1], (p + 1) % NPH, wg, w);   // in flight across the barrier
    __syncthreads();                                   // every thread's y stores are issued
    if (threadIdx.x == 0) sh_ok = grid_barrier(s, (unsigned)p, grp, NWG / 8, 8) ? 1 : 0;
    __syncthreads();
    if (!sh_ok) return;
  }
}

// tuned form: sc1 stores, counted wait, fence-free barrier, run-ahead weights kept in flight across it
__global__ __launch_bounds__(NT) void persistent_tuned_kernel(const half_t* const* Wl, int layers, half_t* buf0, half_t* buf1, Sync* s) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = (half_t*)smem;
  __shared__ int sh_ok;
  const int wg = blockIdx.x, grp = wg & 7;
  half8v w[MAXU];
  const int total = layers * NPH;
  issue_weights(Wl[0], 0, wg, w);
  for (int p = 0; p < total; ++p) {
    const int ph = p % NPH;
    const half_t* x = (p & 1) ? buf1 : buf0;
    half_t* y = (p & 1) ? buf0 : buf1;
    phase_body<true>(ph, wg, w, x, 4 * D, y, 4 * D, xs);
    if (p + 1 == total) break;
    asm volatile("" ::: "memory");
    issue_weights(Wl[p + 1], (p + 1) % NPH, wg, w);
    asm volatile("" ::: "memory");
    wait_stores_keep_weights((p + 1) % NPH);
    __syncthreads();                                   // every wave's y stores are acknowledged
    if (threadIdx.x == 0) sh_ok = grid_barrier_nofence(s, (unsigned)p, grp, NWG / 8, 8) ? 1 : 0;
    __syncthreads();
    if (!sh_ok) return;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_ph), h_ph, sizeof(h_ph)));
  const int LAYERS = 16;                               // weights of 16 layers rotate: every phase streams from HBM
  std::vector<half_t*> h_W(LAYERS * NPH);
  size_t bytes_layer = 0;
  for (int p = 0; p < NPH; ++p) bytes_layer += (size_t)h_ph[p].N * h_ph[p].K * 2;
  {
    std::vector<half_t> h((size_t)4 * D * D);
    for (auto& v : h) v = (half_t)(((rand() & 0xffff) / 65536.0f - 0.5f) * 0.1f);
    for (int i = 0; i < LAYERS * NPH; ++i) {
      const size_t n = (size_t)h_ph[i % NPH].N * h_ph[i % NPH].K;
      CK(hipMalloc(&h_W[i], n * 2));
      CK(hipMemcpy(h_W[i], h.data(), n * 2, hipMemcpyHostToDevice));
    }
  }
  const half_t** d_Wl; CK(hipMalloc(&d_Wl, h_W.size() * sizeof(half_t*)));
  CK(hipMemcpy(d_Wl, h_W.data(), h_W.size() * sizeof(half_t*), hipMemcpyHostToDevice));
  half_t *buf0, *buf1; CK(hipMalloc(&buf0, (size_t)ROWS * 4 * D * 2)); CK(hipMalloc(&buf1, (size_t)ROWS * 4 * D * 2));
  std::vector<half_t> hx((size_t)ROWS * 4 * D);
  for (auto& v : hx) v = (half_t)((rand() & 0xffff) / 65536.0f - 0.5f);
  Sync* d_sync; CK(hipMalloc(&d_sync, sizeof(Sync)));
  const size_t lds = (size_t)ROWS * 4 * D * 2;
  CK(hipFuncSetAttribute((const void*)phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)persistent_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)persistent_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<true>, NT, lds));
  printf("persistent kernel: %d workgroup(s) per CU possible, grid %d (one per CU)\n", occ, NWG);

  std::vector<half_t> ref((size_t)ROWS * 4 * D), got((size_t)ROWS * 4 * D);
  // ---- chain: LAYERS x 6 dependent launches, replayed from a graph
  {
    CK(hipMemcpy(buf0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < LAYERS * NPH; ++p)
      hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(NT), lds, st, h_W[p], p % NPH, (p & 1) ? buf1 : buf0, (p & 1) ? buf0 : buf1);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemcpyAsync(buf0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
    }
    CK(hipMemcpy(ref.data(), ((LAYERS * NPH) & 1) ? buf1 : buf0, ref.size() * 2, hipMemcpyDeviceToHost));
    printf("chain of launches (hipGraph)        : %7.2f us per layer of 6 projections (%.2f us per launch, %.2f TB/s of weights)\n",
           best * 1e3 / LAYERS, best * 1e3 / (LAYERS * NPH), bytes_layer * LAYERS / (best * 1e-3) * 1e-12);
  }
  // ---- persistent forms
  CK(hipFuncSetAttribute((const void*)persistent_tuned_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int ra = 0; ra < 3; ++ra) {
    float best = 1e30f; unsigned aborted = 0;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemcpyAsync(buf0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice, st));
      CK(hipMemsetAsync(d_sync, 0, sizeof(Sync), st));
      CK(hipEventRecord(e0, st));
      if (ra == 2) hipLaunchKernelGGL(persistent_tuned_kernel, dim3(NWG), dim3(NT), lds, st, (const half_t* const*)d_Wl, LAYERS, buf0, buf1, d_sync);
      else if (ra) hipLaunchKernelGGL(persistent_kernel<true>, dim3(NWG), dim3(NT), lds, st, (const half_t* const*)d_Wl, LAYERS, buf0, buf1, d_sync);
      else hipLaunchKernelGGL(persistent_kernel<false>, dim3(NWG), dim3(NT), lds, st, (const half_t* const*)d_Wl, LAYERS, buf0, buf1, d_sync);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
      Sync hs; CK(hipMemcpy(&hs, d_sync, sizeof(Sync), hipMemcpyDeviceToHost)); aborted |= hs.abort_flag;
    }
    CK(hipMemcpy(got.data(), ((LAYERS * NPH) & 1) ? buf1 : buf0, got.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r = 0; r < ROWS; ++r)
      for (int n = 0; n < D; ++n) { const double dd = fabs((double)(float)got[(size_t)r * 4 * D + n] - (double)(float)ref[(size_t)r * 4 * D + n]); if (dd > worst) worst = dd; }
    printf("one persistent launch, %s: %7.2f us per layer (%.2f us per phase incl. barrier)%s   max |y - chain| %.1e\n",
           ra == 2 ? "run-ahead, sc1 stores, no fences     " : ra ? "weights requested before the barrier" : "weights requested after the barrier ", best * 1e3 / LAYERS,
           best * 1e3 / (LAYERS * NPH), aborted ? "  [A SPIN RAN OUT]" : "", worst);
  }
  return 0;
}
