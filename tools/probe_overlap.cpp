// tools/probe_overlap.cpp — developer probe (not part of the product): can the head of decode-step kernel k+1
// (dispatch, kernel-argument fetch, weight requests) run under the tail of kernel k on gfx950, and what does the
// software hand-off that replaces the kernel boundary cost?
//
//   1. hipExtAnyOrderLaunch on one stream (eager and captured into a hipGraph): does kernel B start before A ends?
//   2. the same pair on two streams (the known-good way to overlap) as the yardstick;
//   3. a chain of N dependent GEMV-like kernels (160 workgroups x 256 threads; each streams its own 20 KB of
//      "weights", reads the 40 KB activation vector the previous kernel wrote, writes 8 x 8 outputs):
//        a) plain launches on one stream (kernel boundary = the dependency);
//        b) "soft" dependencies: kernels alternate between two streams, every workgroup requests its weights first,
//           then polls the previous kernel's arrival counter (bounded spin), reads activations with sc0 sc1 loads;
//           producers store sc0 sc1, drain vmcnt, and bump the counter (MI355X_MICROARCH.md, valid hand-off forms);
//        both eager and as a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_overlap.cpp -o tools/probe_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long wall() { return __builtin_readcyclecounter(); }   // s_memtime

// ---- 1/2: overlap detection ---------------------------------------------------------------------
__global__ void spin_kernel(long long* t, long long cycles) {
  const long long t0 = wall();
  if (threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
  while (wall() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(long long* t) {
  if (threadIdx.x == 0) t[2] = __builtin_amdgcn_s_memrealtime();
}

// ---- 3: dependent chain -----------------------------------------------------------------------------
struct ChainArgs {
  const float4v* w;        // [nwg][256][5] 16-byte units: 20 KB per workgroup
  const float* xin;        // [8][1280] activations written by the previous kernel
  float* xout;             // [8][1280]
  unsigned* counter_prev;  // arrivals of the previous kernel (soft mode)
  unsigned* counter_mine;
  unsigned expect;         // counter_prev value that means "previous kernel done"
  int soft;
  int* err;
};

__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a) {
  __shared__ float xs[8 * 1280];
  const int tid = threadIdx.x, wg = blockIdx.x;
  float4v w[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) w[u] = __builtin_nontemporal_load(a.w + ((size_t)wg * 256 + tid) * 5 + u);
  if (a.soft) {
    if (tid == 0) {
      const long long t0 = wall();
      while (__hip_atomic_load(a.counter_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect) {
        __builtin_amdgcn_s_sleep(2);
        if (wall() - t0 > 4000000) { *a.err = 1; break; }   // ~2 ms: give up, never hang the box
      }
    }
    __syncthreads();
    // agent-scope relaxed 8-byte loads (global_load_dwordx2 sc1): served below the non-coherent L1, counted by the compiler
    for (int i = tid; i < 8 * 1280 / 2; i += 256) {
      const unsigned long long v = __hip_atomic_load((const unsigned long long*)a.xin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *(unsigned long long*)(xs + i * 2) = v;
    }
  } else {
    for (int i = tid; i < 8 * 1280 / 4; i += 256) *(float4v*)(xs + i * 4) = *(const float4v*)(a.xin + i * 4);
  }
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 5; ++u)
    acc += w[u][0] * xs[(tid * 5 + u) % 10240] + w[u][1] + w[u][2] + w[u][3];
  // 8 features x 8 rows per workgroup: 64 outputs
  if (tid < 64) {
    float* dst = a.xout + (tid >> 3) * 1280 + wg * 8 + (tid & 7);
    const float v = acc * 1e-3f + xs[(tid >> 3) * 1280 + wg * 8 + (tid & 7)];
    if (a.soft) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = v;
  }
  if (a.soft) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.counter_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  long long* t; CK(hipMalloc(&t, 64));
  long long h[3];
  const long long spin = 100000;   // ~40-50 us
  auto show = [&](const char* name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
    printf("%-46s A: %lld..%lld  B at %lld  -> B %s (B - A.end = %.2f us)\n", name, 0LL, h[1] - h[0], h[2] - h[0],
           h[2] < h[1] ? "OVERLAPS A" : "after A", (h[2] - h[1]) / 100.0);
  };
  // plain, one stream
  CK(hipMemset(t, 0, 64));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, t, spin);
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, t);
  show("one stream, plain launches");
  // any-order flag, one stream
  CK(hipMemset(t, 0, 64));
  hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, t, spin);
  hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, t);
  show("one stream, hipExtAnyOrderLaunch");
  // two streams
  CK(hipMemset(t, 0, 64));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, t, spin);
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, t);
  show("two streams");
  // any-order captured into a graph
  {
    CK(hipMemset(t, 0, 64));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, t, spin);
    hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, t);
    hipError_t e = hipStreamEndCapture(s1, &g);
    if (e == hipSuccess && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
      CK(hipGraphLaunch(ge, s1));
      show("graph, captured hipExtAnyOrderLaunch");
    } else printf("graph capture of hipExtAnyOrderLaunch failed: %s\n", hipGetErrorString(e));
  }
  // graph with two parallel branches (fork / join through events)
  {
    CK(hipMemset(t, 0, 64));
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t ef, ej; CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, t, spin);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, t);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s1));
    show("graph, two parallel branches");
  }

  // ---- dependent chain ------------------------------------------------------------------------------
  const int NWG = 160, N = 64, LAYERS = 16;
  float4v* w; CK(hipMalloc(&w, (size_t)LAYERS * NWG * 256 * 5 * 16)); CK(hipMemset(w, 0, (size_t)LAYERS * NWG * 256 * 5 * 16));
  float* x[2]; CK(hipMalloc(&x[0], 8 * 1280 * 4)); CK(hipMalloc(&x[1], 8 * 1280 * 4));
  CK(hipMemset(x[0], 0, 8 * 1280 * 4)); CK(hipMemset(x[1], 0, 8 * 1280 * 4));
  unsigned* cnt; CK(hipMalloc(&cnt, (N + 1) * 256));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto launch_chain = [&](int soft, int two_streams, unsigned epoch) {
    for (int k = 0; k < N; ++k) {
      ChainArgs a;
      a.w = w + (size_t)(k % LAYERS) * NWG * 256 * 5; a.xin = x[k & 1]; a.xout = x[(k + 1) & 1];
      a.counter_prev = cnt + (size_t)k * 64; a.counter_mine = cnt + (size_t)(k + 1) * 64;
      a.expect = k == 0 ? 0u : (unsigned)NWG * epoch; a.soft = soft; a.err = err;
      hipStream_t s = (two_streams && (k & 1)) ? s2 : s1;
      hipLaunchKernelGGL(chain_kernel, dim3(NWG), dim3(256), 0, s, a);
    }
  };
  auto time_eager = [&](const char* name, int soft, int two) {
    CK(hipMemset(cnt, 0, (N + 1) * 256));
    unsigned epoch = 0;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      ++epoch;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      if (two) { CK(hipStreamWaitEvent(s2, e0, 0)); }
      launch_chain(soft, two, epoch);
      if (two) { hipEvent_t ej; CK(hipEventCreate(&ej)); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-46s %7.2f us per kernel (best of 5, %d kernels)%s\n", name, best * 1e3f / N, N, herr ? "  [SPIN TIMEOUT]" : "");
    CK(hipMemset(err, 0, 4));
  };
  time_eager("chain eager, plain boundaries", 0, 0);
  time_eager("chain eager, soft deps, ONE stream", 1, 0);
  time_eager("chain eager, soft deps, two streams", 1, 1);

  auto time_graph = [&](const char* name, int soft, int two) {
    CK(hipMemset(cnt, 0, (N + 1) * 256));
    CK(hipDeviceSynchronize());
    hipGraph_t g; hipGraphExec_t ge; hipEvent_t ef, ej; CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    // the graph is replayed with a fixed `expect`, so the counters are reset between replays
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    if (two) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
    launch_chain(soft, two, 1);
    if (two) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemsetAsync(cnt, 0, (N + 1) * 256, s1));
      CK(hipEventRecord(e0, s1));
      CK(hipGraphLaunch(ge, s1));
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-46s %7.2f us per kernel (best of 5, %d kernels)%s\n", name, best * 1e3f / N, N, herr ? "  [SPIN TIMEOUT]" : "");
    CK(hipMemset(err, 0, 4));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  };
  time_graph("chain graph, plain boundaries", 0, 0);
  time_graph("chain graph, soft deps, two branches", 1, 1);
  return 0;
}
