// tools/probe_actread.cpp — developer probe (not part of the product): how fast can EVERY CU fetch the same small
// activation block that the previous launch has just written?
//
// Why.  Round 5's engine probe (tools/probe_engine.cpp) and the three launches it was compared with are both explained by
// ONE number: a CU moves ~10 B/clk (24 GB/s) through its vector-memory pipe, whatever it asks for.  The decode step's
// projection launches are then not bound by their weights (12.8 / 51 KB per CU) but by the ACTIVATIONS every CU re-reads
// in full: 61 KB of attention partials (merge + cross_attn.out), 40 KB of fp32 residual rows (every LayerNorm prologue),
// 80 KB of MLP activations (FC2) — 280 KB of 690 KB per CU per layer.  If those (L2 / Infinity-Cache resident, identical
// for all CUs) could be fetched faster than the HBM stream, every launch of the step gets shorter without any restructuring.
//
// What.  Link = producer launch (256 workgroups write their 1/256 slice of a B-byte block, plain stores, as FC1's epilogue
// does) -> consumer launch (256 workgroups x 1024 threads, each reads the WHOLE block with method M, XORs it down and
// stores one word so nothing is dead).  32 links per hipGraph; reported: us per link minus the same chain with a consumer
// that reads nothing ("empty").  Methods:
//   plain   global_load_dwordx4, B / 16 KB loads per lane in flight, fragment-style addressing (what gemv8_kernel does)
//   sc1     the same with sc1 (L1 bypass; what a tagged hand-off uses)
//   nt      non-temporal
//   glds    global_load_lds_dwordx4 (LDS-DMA) by all 16 waves, then one ds_read pass
//   glds1   LDS-DMA by ONE wave (the engine's loader)
//   half    plain loads, but only HALF of the block per workgroup (what a 2-way K split would read)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_actread.cpp -o tools/probe_actread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void producer(uint4v* buf, int units, unsigned seed) {
  // workgroup w writes units [w * units / 256, ...): 16-byte plain stores
  const int per = units / 256;
  for (int i = threadIdx.x; i < per; i += 256) {
    const unsigned v = seed + blockIdx.x * 977u + i;
    buf[(size_t)blockIdx.x * per + i] = uint4v{v, v ^ 0x55u, v + 3u, v * 7u};
  }
}

__device__ __forceinline__ uint4v ld_sc1(const uint4v* p) {
  uint4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// NL = 16-byte loads per lane (units = 1024 * NL for the full block); METHOD as in the header
template <int NL, int METHOD>
__global__ __launch_bounds__(1024) void consumer(const uint4v* __restrict__ buf, unsigned* out, long long* stamp) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long t0 = 0;
  if (stamp && tid == 0) t0 = wall_clock64();
  unsigned acc = 0;
  if (METHOD == 0 || METHOD == 2 || METHOD == 5) {
    constexpr int N = METHOD == 5 ? NL / 2 : NL;
    uint4v v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint4v* p = buf + (size_t)(i * 16 + wave) * 64 + lane;      // wave-load = 1 KB contiguous, K blocks interleaved over waves
      v[i] = METHOD == 2 ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  } else if (METHOD == 1) {
    // sc1 loads, all in flight, one wait
    uint4v v[NL];
    const uint4v* p = buf + (size_t)wave * 64 + lane;
#pragma unroll
    for (int i = 0; i < NL; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(p + (size_t)i * 1024) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NL; ++i) { asm volatile("" : "+v"(v[i])); acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3]; }
  } else {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    if (METHOD == 3) {
#pragma unroll
      for (int i = 0; i < NL; ++i) glds16(buf + (size_t)(i * 16 + wave) * 64 + lane, lds0 + (unsigned)(i * 16 + wave) * 1024u);
    } else if (wave == 0) {
#pragma unroll 8
      for (int i = 0; i < NL * 16; ++i) glds16(buf + (size_t)i * 64 + lane, lds0 + (unsigned)i * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint4v v = *(const uint4v*)(smem + ((size_t)(i * 16 + wave) * 64 + lane) * 16);
      acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  }
  // wave XOR through LDS-free DPP-less path: just let lane 0 of each wave store (keeps every load alive per lane via the xor below)
  for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
  if (lane == 0) out[blockIdx.x * 16 + wave] = acc;
  if (stamp && tid == 0) stamp[blockIdx.x] = wall_clock64() - t0;
}

__global__ __launch_bounds__(1024) void consumer_empty(unsigned* out) {
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = 1;
}

template <int NL, int METHOD>
static void run(const char* name, uint4v* buf, unsigned* out, long long* stamp, hipStream_t st, float empty_us, int rotate) {
  const int units = NL * 1024, N = 32;
  const size_t lds = (METHOD == 3 || METHOD == 4) ? (size_t)units * 16 : 0;
  if (lds) CK(hipFuncSetAttribute((const void*)consumer<NL, METHOD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) {
    uint4v* b = buf + (size_t)(i % rotate) * units;
    hipLaunchKernelGGL(producer, dim3(256), dim3(256), 0, st, b, units, (unsigned)i);
    hipLaunchKernelGGL((consumer<NL, METHOD>), dim3(256), dim3(1024), lds, st, b, out, i == N - 1 ? stamp : nullptr);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 1 && ms < best) best = ms;
  }
  std::vector<long long> s(256);
  CK(hipMemcpy(s.data(), stamp, 256 * 8, hipMemcpyDeviceToHost));
  std::sort(s.begin(), s.end());
  const float us = best * 1e3f / N;
  const float kb = units * 16 / 1024.f * (METHOD == 5 ? 0.5f : 1.f);
  printf("  %-34s %4.0f KB per CU: %6.2f us per link, %6.2f over the empty consumer = %5.1f GB/s per CU | in-kernel (entry -> stored) %5.2f / %5.2f / %5.2f us\n",
         name, kb, us, us - empty_us, kb * 1024 / ((us - empty_us) * 1e3f), s[0] / 100.0, s[128] / 100.0, s[255] / 100.0);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint4v* buf; unsigned* out; long long* stamp;
  CK(hipMalloc(&buf, (size_t)8 * 10 * 1024 * 16)); CK(hipMalloc(&out, 256 * 16 * 4)); CK(hipMalloc(&stamp, 256 * 8));
  // baseline: producer + empty consumer
  float empty_us[2];
  for (int nl : {5, 10}) {
    const int units = nl * 1024, N = 32;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(producer, dim3(256), dim3(256), 0, st, buf, units, (unsigned)i);
      hipLaunchKernelGGL(consumer_empty, dim3(256), dim3(1024), 0, st, out);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    empty_us[nl == 10] = best * 1e3f / N;
    printf("producer (%d KB) + empty consumer: %.2f us per link\n", units * 16 / 1024, empty_us[nl == 10]);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  printf("80 KB block (FC2's activations, 8 x 5120 fp16):\n");
  run<5, 0>("plain dwordx4, 16 waves x 5", buf, out, stamp, st, empty_us[0], 1);
  run<5, 1>("sc1 dwordx4", buf, out, stamp, st, empty_us[0], 1);
  run<5, 2>("non-temporal dwordx4", buf, out, stamp, st, empty_us[0], 1);
  run<5, 3>("LDS-DMA by 16 waves + ds_read", buf, out, stamp, st, empty_us[0], 1);
  run<5, 4>("LDS-DMA by one wave + ds_read", buf, out, stamp, st, empty_us[0], 1);
  run<5, 0>("plain, 8 rotating blocks", buf, out, stamp, st, empty_us[0], 8);
  printf("160 KB block (the same as 8-byte {data, tag} granules):\n");
  run<10, 0>("plain dwordx4, 16 waves x 10", buf, out, stamp, st, empty_us[1], 1);
  run<10, 1>("sc1 dwordx4", buf, out, stamp, st, empty_us[1], 1);
  run<10, 5>("plain, HALF the block per workgroup", buf, out, stamp, st, empty_us[1], 1);
  return 0;
}
