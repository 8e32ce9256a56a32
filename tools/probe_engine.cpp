// tools/probe_engine.cpp — developer probe (not part of the product): the attention-free half of one large-v3 decoder
// layer at 8 rows as ONE launch on a loader/consumer engine, against the product's THREE launches
//   [merge of the cross-attention splits + cross_attn.out + residual]  ->  [LN + FC1 + GELU]  ->  [FC2 + residual]
// (whisper/model.py:44-50, 142-171).
//
// What is different from round 4's mlp8_kernel (tools/probe_mlp.cpp, rejected at 12.7 vs 11.7 us):
//   * the weights do not travel through the consumers' registers.  ONE loader wave per workgroup streams the workgroup's
//     three weight slices (cout 12.8 KB, FC1 51.2 KB, FC2 51.2 KB = 115 KB of the CU's 160 KB of LDS — the whole triple fits,
//     so the loader never waits for a dependency) with `global_load_lds_dwordx4 ... nt` straight into MFMA fragment order
//     (a DMA instruction writes 64 x 16 B lane-linear; the per-lane SOURCE address is chosen so that the image in LDS is the
//     fragment the consumer's ds_read_b128 wants: no swizzle, no holes — 5-row and 4-row feature groups are packed 40 / 32
//     lanes per fragment).  It keeps DEPTH groups of 8 KB in flight (counted vmcnt) and publishes a `landed` counter in LDS;
//   * so no consumer wave has a weight request of its own outstanding when it polls a hand-off: its vmcnt is its own;
//   * the two all-to-all edges are data-tagged 8-byte granules written by ONE write-through (sc1) store each
//     (x' = 8 x 1280 fp32 -> {fp32, tag}; h = 8 x 5120 fp16 -> {2 x fp16, tag}), swept by the consumers with 16-byte sc1
//     loads (two granules per load, 10 loads per lane in flight) straight into the registers that need them: no flag, no
//     drain between payload and flag, no second round trip for the payload after the flag;
//   * no workgroup barrier after the role split: the 15 consumer waves synchronise through an LDS arrival counter, the
//     loader joins them as the 16th FC2 wave once its last DMA is issued.
// Same MFMA fragment maps, same chains of accumulation and the same order of the cross-wave sums as gemv8_kernel
// (PRO_COMBINE/GS1/KS4/XW8 fw 5, PRO_LN/GS3/KS4/XW4 fw 20, PRO_PLAIN/GS1/KS16 fw 5): the result must be BIT-IDENTICAL to the
// three launches (checked, and the probe says so).  Every spin is bounded and counts in *err.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_engine.cpp -o tools/probe_engine
//   ./tools/probe_engine          (PROBE_L = distinct layers of weights the 32-link chain rotates through, default 8)
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

struct EngArgs {
  const half_t* part_o; const float* part_ml;   // [3][R][H][64] fp16 (o / l), [3][R][H][2] fp32 (m, l)
  const half_t* Wc; const float* bc;            // cross_attn.out [D][D], [D]
  const half_t* W1; const float* b1;            // [4D][D] (LayerNorm affine folded in), [4D]
  const half_t* W2; const float* b2;            // [D][4D], [D]
  float* x; int64_t x_ld;                       // residual stream, in place: workgroup w owns features [5 w, 5 w + 5)
  u64* xg;                                      // [8][D]   granules {fp32 x', tag}
  u64* hg;                                      // [8][2 D] granules {2 x fp16 h, tag}
  const int* d_tick; int epoch;
  int* err;
  int R;
  long long* probe;                             // [256][16] wall_clock64 stamps or null
};

namespace eng {
constexpr int D = 1280, H = 20, N1 = 5120;
constexpr int KS = 4, NU = 5, KS2 = 16;
constexpr int FWC = 5, FW1 = 20, FW2 = 5;
constexpr int FRAG = KS * NU * 64;                         // x fragment units (16 B) of one row tile
// DMA plan: instruction t writes LDS bytes [1024 t, 1024 t + 1024)
constexpr int T_C = 13;                                    // cout: 20 fragments x 40 lanes x 16 B = 12 800 B
constexpr int T_1 = 50;                                    // FC1: 40 fragments x 1 KB + 20 fragments x 512 B
constexpr int T_2 = 50;                                    // FC2: 80 fragments x 40 lanes x 16 B
constexpr int T_ALL = T_C + T_1 + T_2;                     // 113
constexpr int GRP = 8;                                     // DMA instructions per published group
constexpr int NGRP = (T_ALL + GRP - 1) / GRP;              // 15
constexpr int OFF_WC = 0, OFF_W1 = T_C * 1024, OFF_W2 = (T_C + T_1) * 1024, OFF_XF = T_ALL * 1024;
constexpr int OFF_RED = OFF_XF + FRAG * 16;                // [12][8][8] floats: cout (4 slots), FC1 (12)
constexpr int OFF_RED2 = OFF_RED + 12 * 64 * 4;            // FC2's own [16][8][8] (its first writers do not wait for FC1's readers)
constexpr int OFF_CTL = OFF_RED2 + 16 * 64 * 4;            // ints: [0] landed instructions, [1] consumer arrival counter, [2] tag
constexpr int LDS_BYTES = OFF_CTL + 64;
constexpr int MAX_SPINS = 1 << 14;
// `landed` a consumer needs before it may read: all instructions of a segment
constexpr int NEED_C = T_C, NEED_1A = T_C + 20, NEED_1B = T_C + 40, NEED_1C = T_C + 50, NEED_2 = T_ALL;
}  // namespace eng

#define ESTAMP(i) do { if (a.probe && lane == 0) a.probe[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)

__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void eng_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ten 16-byte write-through-coherent (sc1: L1-bypassing) loads of one lane, issued back to back and waited for inside ONE
// statement (hipcc does not count asm loads: cdna_hip_programming.md §5.7 form (i))
struct Gather10 { uint4v v[10]; };
// OFF2: byte distance of a lane's second 16 bytes (granules 2, 3) from its first (granules 0, 1).  The hand-off buffers are laid
// out so that one load INSTRUCTION reads whole 128-byte lines: the first halves of 8 (h) / 64 (x') neighbouring lanes are
// contiguous, the second halves follow 128 bytes / 1 KB further on (with the natural layout, 32 contiguous bytes per lane, every
// instruction touched every line of its span and used half of it: twice the L2 requests per byte).
template <int OFF2>
__device__ __forceinline__ void gather10(Gather10& g, const char* p0, const char* p1, const char* p2, const char* p3, const char* p4) {
  asm volatile(
      "global_load_dwordx4 %0, %10, off sc1\n\t"
      "global_load_dwordx4 %1, %10, off offset:%c15 sc1\n\t"
      "global_load_dwordx4 %2, %11, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off offset:%c15 sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %12, off offset:%c15 sc1\n\t"
      "global_load_dwordx4 %6, %13, off sc1\n\t"
      "global_load_dwordx4 %7, %13, off offset:%c15 sc1\n\t"
      "global_load_dwordx4 %8, %14, off sc1\n\t"
      "global_load_dwordx4 %9, %14, off offset:%c15 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(g.v[0]), "=&v"(g.v[1]), "=&v"(g.v[2]), "=&v"(g.v[3]), "=&v"(g.v[4]), "=&v"(g.v[5]), "=&v"(g.v[6]), "=&v"(g.v[7]),
        "=&v"(g.v[8]), "=&v"(g.v[9])
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "i"(OFF2)
      : "memory");
}
__device__ __forceinline__ bool tags_ok(const Gather10& g, uint32_t tag) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 10; ++i) ok = ok && g.v[i][1] == tag && g.v[i][3] == tag;
  return ok;
}
// every ACTIVE lane fetches its own 5 x 32 bytes of granules until all of ITS tags match; lanes that are done issue nothing more
template <int OFF2>
__device__ __forceinline__ void gather_until(Gather10& g, const char* p0, const char* p1, const char* p2, const char* p3, const char* p4,
                                             uint32_t tag, int lane, int* err, bool active = true) {
  bool done = !active;
  int spins = 0;
  for (;;) {
    if (!done) {
      gather10<OFF2>(g, p0, p1, p2, p3, p4);
      done = tags_ok(g, tag);
    }
    if (__all(done)) break;
    if (++spins >= eng::MAX_SPINS) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(1);
  }
}

// barrier among the consumer waves through an LDS arrival counter (cumulative target), as xattn.hip's aux_barrier
__device__ __forceinline__ void cbarrier(int* cnt, int target, int lane, int* err) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    if (++spins >= (1 << 20)) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void wait_landed(const int* landed, int need, int lane, int* err) {
  int spins = 0;
  while (__hip_atomic_load(landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
    if (++spins >= (1 << 20)) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// five chained MFMAs of one (feature group, K split) + the diagonal fold of gemv8_kernel -> red[slot][feature][row]
__device__ __forceinline__ void chain5(const half8v* wa, const half8v* xb, float* red_slot, int lane) {
  const bool diag = (lane >> 5) == ((lane >> 3) & 1);
  float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < eng::NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xb[u], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float z = diag ? acc[e] : 0.f;
    z += lane_xor8(z);
    float p, q; lane_swap32(z, p, q);
    acc[e] = p + q;
  }
  if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red_slot[(4 * (lane >> 4) + e) * 8 + (lane & 7)] = acc[e];
  }
}

// 256 workgroups (one per CU) x 16 waves: waves 0-14 consumers, wave 15 the loader (and the 16th FC2 wave).
__device__ __forceinline__ void wait_flag(const int* f, int lane, int* err) {
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
    if (++spins >= (1 << 20)) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(2);
  }
}

// DEPTH: groups of 8 DMA instructions (8 KB) the loader keeps in flight.  MODE bit 0: sampled polling (one row of an edge is polled,
// the rest is gathered once it is complete) instead of every wave sweeping its whole share from the moment it is ready; bit 1: the
// loader holds FC2's slice back until the x' gather of its workgroup is over.
// (An intermediate version announced arrivals on 8 sharded device-scope counters instead: the 256 / 768 atomics of an edge took
// 5 / 12 us to complete — ~16 ns each, serialised — profiles/r05_probe_engine.txt.)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void tail3_kernel(EngArgs a) {
  using namespace eng;
  pin_kernargs(a);
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  half8v* xfrag = (half8v*)(smem + OFF_XF);
  float* red = (float*)(smem + OFF_RED);
  float* red2 = (float*)(smem + OFF_RED2);
  int* ctl = (int*)(smem + OFF_CTL);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x, R = a.R;
  const int idx = lane & 7, hlf = (lane >> 3) & 1, cq = lane >> 4;       // MFMA lane map: row idx, K half, 16-byte column
  const int gq = hlf * 4 + cq;                                           // 16-byte piece of the 128-byte K block
  if (tid < 16) ctl[tid] = 0;
  __syncthreads();                                                       // the only workgroup barrier: before the role split
  if (wave == 0) ESTAMP(0);
  int vt = 0;                                                            // the step tick: requested BEHIND the merge's loads (consumers)
  unsigned tag = 0;

  if (wave == 15) {
    // ================================================= loader
    // source of DMA lane `lane` of instruction t, segment by segment; LDS image = fragments in consumer lane order
    auto issue = [&](int t) {
      const char* src; bool on = true;
      if (t < T_C) {
        const uint32_t p = 64u * t + lane, f = p / 40u, q = p - f * 40u, g = q / 5u, i = q - g * 5u;
        on = p < 800u;
        src = (const char*)a.Wc + ((size_t)(w * FWC + i) * D + f * 64u + g * 8u) * 2;
      } else if (t < T_C + 40) {
        const uint32_t f = t - T_C, grp = f / 20u, blk = f - grp * 20u;
        src = (const char*)a.W1 + ((size_t)(w * FW1 + grp * 8 + idx) * D + blk * 64u + gq * 8u) * 2;
      } else if (t < T_C + T_1) {
        const uint32_t p = 64u * (t - T_C - 40) + lane, f = p >> 5, q = p & 31u, g = q >> 2, i = q & 3u;
        src = (const char*)a.W1 + ((size_t)(w * FW1 + 16 + i) * D + f * 64u + g * 8u) * 2;
      } else {
        const uint32_t p = 64u * (t - T_C - T_1) + lane, f = p / 40u, q = p - f * 40u, g = q / 5u, i = q - g * 5u;
        src = (const char*)a.W2 + ((size_t)(w * FW2 + i) * N1 + f * 64u + g * 8u) * 2;
      }
      if (on) glds16_nt(src, lds0 + (uint32_t)t * 1024u);
    };
    // [T0, T1) in groups of GRP instructions: after issuing a group, wait until all but the DEPTH - 1 youngest groups have
    // landed and publish the count; a full drain at the end of the range
    auto stream = [&](auto t0_tag, auto t1_tag) {
      constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
      constexpr int NG = (T1 - T0 + GRP - 1) / GRP, LAST = (T1 - T0) - (NG - 1) * GRP;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < GRP; ++i) if (T0 + g * GRP + i < T1) issue(T0 + g * GRP + i);
        if (g >= DEPTH - 1 && g < NG - 1) {
          eng_wait_vmcnt<(DEPTH - 1) * GRP>();
          if (lane == 0) __hip_atomic_store(&ctl[0], T0 + (g - DEPTH + 2) * GRP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (T0 == 0 && g == 1) ESTAMP(8);
      }
      // drain: the youngest groups one by one (vmcnt immediates: k - 1 full groups + the last, shorter one)
#pragma unroll
      for (int k = (DEPTH - 1 < NG ? DEPTH - 1 : NG) - 1; k >= 0; --k) {
        switch (k) {
          case 0: eng_wait_vmcnt<0>(); break;
          case 1: eng_wait_vmcnt<LAST>(); break;
          case 2: eng_wait_vmcnt<GRP + LAST>(); break;
          case 3: eng_wait_vmcnt<2 * GRP + LAST>(); break;
          case 4: eng_wait_vmcnt<3 * GRP + LAST>(); break;
          default: eng_wait_vmcnt<4 * GRP + LAST>(); break;
        }
        if (lane == 0) __hip_atomic_store(&ctl[0], k == 0 ? T1 : T0 + (NG - k) * GRP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    if (MODE & 2) {
      // cout + FC1 first; FC2's slice only once this workgroup's x' gather is over: a gather issued while 48 KB of the CU's own
      // DMA requests are in flight comes back behind them (the CU's vector-memory pipe returns in issue order)
      stream(std::integral_constant<int, 0>(), std::integral_constant<int, T_C + T_1>());
      int spins = 0;
      while (__hip_atomic_load(&ctl[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8) {
        if (++spins >= (1 << 20)) { if (lane == 0 && a.err) atomicAdd(a.err, 1); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      stream(std::integral_constant<int, T_C + T_1>(), std::integral_constant<int, T_ALL>());
    } else {
      stream(std::integral_constant<int, 0>(), std::integral_constant<int, T_ALL>());
    }
    ESTAMP(9);
    // the launch's tag, left in LDS by consumer wave 0 long ago (never 0)
    int spins = 0, tg;
    while ((tg = __hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) {
      if (++spins >= (1 << 20)) { if (lane == 0 && a.err) atomicAdd(a.err, 1); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    tag = (unsigned)uniform(tg);
  }

  // the tick was requested at entry; it is READ only after the merge (consumers) / the last DMA (loader), so that its round
  // trip hides under the first requests instead of standing in front of them
  float xprime = 0.f;                                                    // wave 0: x'[er][5 w + ej] of lane (er, ej)
  float e_bc = 0.f, e_res = 0.f, e_b1 = 0.f, e_b2 = 0.f;

  if (wave < 15) {
    // ================================================= consumers
    // ---- phase 0a: merge of the 3 attention splits (gemv8_kernel PRO_COMBINE, CSm = 3) -> fragments; wave-loads wave, wave + 15
    {
      constexpr int CSm = 3;
      const int npair = 8 * H, nload = (npair + 7) >> 3;                 // 20 wave-loads of 8 (row, head) pairs
      const size_t split_stride = (size_t)R * H;
      half8v po[2][CSm]; float2v pml[2][CSm];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int ld = wave + 15 * i; if (ld > nload - 1) ld = nload - 1;
        const int pair = ld * 8 + (lane >> 3);
        const int prow = pair / H, ph = pair - prow * H;
        const int grow = prow < R ? prow : R - 1;
        const uint32_t pidx = (uint32_t)grow * (uint32_t)H + (uint32_t)ph;
#pragma unroll
        for (int s = 0; s < CSm; ++s) {
          pml[i][s] = *(const float2v*)((const char*)a.part_ml + ((size_t)s * split_stride) * 8 + pidx * 8u);
          po[i][s] = *(const half8v*)((const char*)a.part_o + ((size_t)s * split_stride) * 128 + (pidx * 64u + (uint32_t)(lane & 7) * 8u) * 2u);
        }
      }
      ISSUE_FENCE();
      // epilogue operands of all three phases (L2 hits), behind the merge's own requests
      if (wave == 0) {
        const int er = lane >> 3, ej = lane & 7;
        if (ej < FWC) {
          e_bc = a.bc[w * FWC + ej]; e_b2 = a.b2[w * FW2 + ej];
          e_res = a.x[(int64_t)(er < R ? er : R - 1) * a.x_ld + w * FWC + ej];
        }
      }
      if (wave < 3) { const int ej = lane & 7; if (wave * 8 + ej < FW1) e_b1 = a.b1[w * FW1 + wave * 8 + ej]; }
      vt = load_agent_int(a.d_tick);
      ISSUE_FENCE();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ld = wave + 15 * i;
        const int pair = ld * 8 + (lane >> 3);
        if (ld < nload) {
          const int prow = pair / H, ph = pair - prow * H;
          float M = pml[i][0][0];
#pragma unroll
          for (int s = 1; s < CSm; ++s) M = fmaxf(M, pml[i][s][0]);
          float wgt[CSm], den = 0.f;
#pragma unroll
          for (int s = 0; s < CSm; ++s) {
            wgt[s] = (pml[i][s][0] != WH_NEG_INF) ? __expf(pml[i][s][0] - M) * pml[i][s][1] : 0.f;
            den += wgt[s];
          }
          const float inv = 1.0f / den;
          half8v xo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float num = 0.f;
#pragma unroll
            for (int s = 0; s < CSm; ++s) num = __builtin_fmaf(wgt[s], (float)po[i][s][e], num);
            xo[e] = (half_t)(num * inv);
          }
          const int dl = lane & 7;
          xfrag[((ph % KS) * NU + ph / KS) * 64 + 16 * (dl & 3) + 8 * (dl >> 2) + (prow & 7)] = xo;
        }
      }
    }
    cbarrier(&ctl[1], 15, lane, a.err);                                  // #1: merged fragments complete
    tag = (unsigned)(uniform(vt) + a.epoch + 1);
    if (wave == 0) {
      if (lane == 0) __hip_atomic_store(&ctl[2], (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ESTAMP(1);
    }
    // ---- phase 0b: cout chains (waves 0-3 = K splits)
    if (wave < 4) {
      wait_landed(&ctl[0], NEED_C, lane, a.err);
      half8v wa[NU], xb[NU];
      const int ii = idx < FWC ? idx : FWC - 1;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int blk = wave + KS * u;
        wa[u] = *(const half8v*)(smem + OFF_WC + (blk * 40 + gq * 5 + ii) * 16);
        xb[u] = xfrag[(wave * NU + u) * 64 + lane];
      }
      chain5(wa, xb, red + wave * 64, lane);
    }
    cbarrier(&ctl[1], 30, lane, a.err);                                  // #2
    // ---- phase 0c: bias + residual, x' published as granules {fp32, tag}
    if (wave == 0) {
      const int er = lane >> 3, ej = lane & 7;
      float v = e_bc;
#pragma unroll
      for (int k = 0; k < KS; ++k) v += red[k * 64 + ej * 8 + er];
      xprime = e_res + v;
      if (ej < FWC && er < R) {
        const u64 g = ((u64)tag << 32) | (u64)__float_as_uint(xprime);
        const int f = w * FWC + ej;                                      // granule slot of feature f: see gather10
        const int slot = (f >> 8) * 256 + ((f >> 1) & 1) * 128 + ((f & 255) >> 2) * 2 + (f & 1);
        __hip_atomic_store(a.xg + (size_t)er * D + slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ESTAMP(2);
    }
    // ---- phase 1a: waves 0-7 gather one row of x' each (20 granules per lane) and LayerNorm it into fragments
    if (wave < 8) {
      const int r = wave;
      const char* rowp = (const char*)(a.xg + (size_t)(r < R ? r : R - 1) * D);
      Gather10 g;
      // SAMPLED POLLING (MODE bit 0): every producer stores its granules of all 8 rows with ONE instruction, so row 0 complete
      // means (as good as) everything complete.  Wave 0 polls row 0 — 10 KB per round instead of the 80 KB all eight waves
      // would sweep — and releases the others, which then gather once; the tags still decide (a straggling granule is re-fetched).
      if ((MODE & 1) && wave != 0) wait_flag(&ctl[3], lane, a.err);
      // lane's elements of wave-load j: k = (64 j + lane) * 4 .. + 3: granules 0, 1 at 2 KB j + 16 lane, granules 2, 3 1 KB further
      gather_until<1024>(g, rowp + (size_t)0 * 2048 + lane * 16, rowp + (size_t)1 * 2048 + lane * 16, rowp + (size_t)2 * 2048 + lane * 16,
                         rowp + (size_t)3 * 2048 + lane * 16, rowp + (size_t)4 * 2048 + lane * 16, tag, lane, a.err);
      if ((MODE & 1) && wave == 0 && lane == 0) __hip_atomic_store(&ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) __hip_atomic_fetch_add(&ctl[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the loader may go on
      if (wave == 0) ESTAMP(3);
      float4v v[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j)
        v[j] = float4v{__uint_as_float(g.v[2 * j][0]), __uint_as_float(g.v[2 * j][2]), __uint_as_float(g.v[2 * j + 1][0]), __uint_as_float(g.v[2 * j + 1][2])};
      const float invK = 1.0f / (float)D;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NU; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      const float mean = wave_sum(sum) * invK;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NU; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
      }
      const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
      const uint32_t fbase = (uint32_t)((lane >> 4) * NU * 64 + 16 * ((lane >> 1) & 3) + 8 * ((lane >> 3) & 1)) * 16u + (uint32_t)(lane & 1) * 8u;
      const uint32_t rbase = fbase + (uint32_t)((r & 7) * 16);
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        half4v o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (half_t)((v[j][e] - mean) * rstd);
        *(half4v*)((char*)xfrag + rbase + (uint32_t)(j * 1024)) = o4;
      }
    }
    cbarrier(&ctl[1], 45, lane, a.err);                                  // #3: LayerNorm fragments complete
    if (wave == 0) ESTAMP(4);
    // ---- phase 1b: FC1 chains: wave = 4 grp + K split, grp 0-2 (8, 8, 4 features)
    if (wave < 12) {
      const int grp = wave >> 2, kw = wave & 3;
      wait_landed(&ctl[0], grp == 0 ? NEED_1A : grp == 1 ? NEED_1B : NEED_1C, lane, a.err);
      half8v wa[NU], xb[NU];
      const int i4 = idx < 4 ? idx : 3;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int blk = kw + KS * u;
        if (grp < 2) wa[u] = *(const half8v*)(smem + OFF_W1 + (grp * 20 + blk) * 1024 + lane * 16);
        else wa[u] = *(const half8v*)(smem + OFF_W1 + 40 * 1024 + (blk * 32 + gq * 4 + i4) * 16);
        xb[u] = xfrag[(kw * NU + u) * 64 + lane];
      }
      chain5(wa, xb, red + wave * 64, lane);
    }
    cbarrier(&ctl[1], 60, lane, a.err);                                  // #4
    // ---- phase 1c: bias + GELU, h published as granules {2 x fp16, tag}
    if (wave < 3) {
      const int er = (lane >> 3) & 7, ej = lane & 7;
      float s = e_b1;
#pragma unroll
      for (int k = 0; k < KS; ++k) s += red[(wave * KS + k) * 64 + ej * 8 + er];
      const half_t hv = (half_t)gelu_erf(s);
      const bool on = wave * 8 + ej < FW1 && er < R;
      const uint32_t mine = (uint32_t)__builtin_bit_cast(unsigned short, hv);
      const uint32_t other = __float_as_uint(lane_xor1(__uint_as_float(mine)));
      if ((lane & 1) == 0 && on) {
        const u64 g = ((u64)tag << 32) | (u64)(mine | (other << 16));
        const int n = w * FW1 + wave * 8 + ej;                           // even feature; slot of its granule: see gather10
        const int wi = n & 63, slot = (n >> 6) * 32 + ((wi >> 2) & 1) * 16 + (wi >> 3) * 2 + ((wi >> 1) & 1);
        __hip_atomic_store(a.hg + (size_t)er * (N1 / 2) + slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wave == 0) ESTAMP(5);
    }
  }

  // ================================================= all 16 waves: FC2, K split = wave (blocks wave + 16 u)
  {
    const int rr = idx < R ? idx : R - 1;
    const char* rowp = (const char*)(a.hg + (size_t)rr * (N1 / 2));
    Gather10 g;
    // lane's B fragment of block blk: k = 64 blk + 32 half + 8 c .. + 7: granules 0, 1 at 256 blk + 16 (4 half + c), 2, 3 128 bytes on
    const size_t lo = (size_t)gq * 16;
    const char *q0 = rowp + (size_t)(wave + 16 * 0) * 256 + lo, *q1 = rowp + (size_t)(wave + 16 * 1) * 256 + lo,
               *q2 = rowp + (size_t)(wave + 16 * 2) * 256 + lo, *q3 = rowp + (size_t)(wave + 16 * 3) * 256 + lo,
               *q4 = rowp + (size_t)(wave + 16 * 4) * 256 + lo;
    // sampled polling: the row-0 lanes (8 of 64) poll until the producers of this wave's five K blocks have published, then
    // everyone gathers once (a producer stores all 8 rows with one instruction)
    if (MODE & 1) gather_until<128>(g, q0, q1, q2, q3, q4, tag, lane, a.err, idx == 0);
    if (wave == 0) ESTAMP(10);
    gather_until<128>(g, q0, q1, q2, q3, q4, tag, lane, a.err);
    if (wave == 0) ESTAMP(6);
    wait_landed(&ctl[0], NEED_2, lane, a.err);
    half8v wa[NU], xb[NU];
    const int ii = idx < FW2 ? idx : FW2 - 1;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int blk = wave + KS2 * u;
      wa[u] = *(const half8v*)(smem + OFF_W2 + (blk * 40 + gq * 5 + ii) * 16);
      const uint4v d = uint4v{g.v[2 * u][0], g.v[2 * u][2], g.v[2 * u + 1][0], g.v[2 * u + 1][2]};
      xb[u] = __builtin_bit_cast(half8v, d);
    }
    chain5(wa, xb, red2 + wave * 64, lane);
  }
  cbarrier(&ctl[1], 76, lane, a.err);                                    // #5: all 16 waves
  if (wave == 0) ESTAMP(11);
  if (wave == 0) {
    const int er = lane >> 3, ej = lane & 7;
    float s = e_b2;
#pragma unroll
    for (int k = 0; k < KS2; ++k) s += red2[k * 64 + ej * 8 + er];
    if (ej < FW2 && er < R) a.x[(int64_t)er * a.x_ld + w * FW2 + ej] = xprime + s;
    ESTAMP(7);
  }
}

__global__ void bump_kernel(int* p, int by) { if (threadIdx.x == 0) atomicAdd(p, by); }

static void fill_half(half_t* d, size_t n, float scale) {
  std::vector<half_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (half_t)(((rand() & 0xffff) / 65536.0f - 0.5f) * scale);
  CK(hipMemcpy(d, h.data(), n * sizeof(half_t), hipMemcpyHostToDevice));
}
static void fill_float(float* d, size_t n, float scale, float offset = 0.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = ((rand() & 0xffff) / 65536.0f - 0.5f) * scale + offset;
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}

template <int DEPTH, int MODE>
static void launch_tail3(const EngArgs& a, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    CK(hipFuncSetAttribute((const void*)tail3_kernel<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, eng::LDS_BYTES));
    raised = true;
  }
  hipLaunchKernelGGL((tail3_kernel<DEPTH, MODE>), dim3(256), dim3(1024), eng::LDS_BYTES, st, a);
}

int main(int argc, char** argv) {
  const int D = 1280, H = 20, R = argc > 1 ? atoi(argv[1]) : 8, N = 32, S = 3;
  const int L = getenv("PROBE_L") ? atoi(getenv("PROBE_L")) : 8;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t wl = (size_t)9 * D * D;                     // cout (D^2) + FC1 (4 D^2) + FC2 (4 D^2) per layer
  half_t *W, *h2, *po; float *x0, *xa, *xb, *bc, *b1, *b2, *pml; u64 *xg, *hg; int *tick, *err; long long* probe;
  CK(hipMalloc(&W, wl * L * 2)); fill_half(W, wl * L, 0.05f);
  CK(hipMalloc(&h2, (size_t)8 * 4 * D * 2));
  CK(hipMalloc(&po, (size_t)S * 8 * D * 2)); fill_half(po, (size_t)S * 8 * D, 2.0f);
  CK(hipMalloc(&pml, (size_t)S * 8 * H * 2 * 4));
  {
    std::vector<float> ml((size_t)S * 8 * H * 2);
    for (size_t i = 0; i < ml.size(); i += 2) { ml[i] = ((rand() & 0xffff) / 65536.0f - 0.5f) * 4.f; ml[i + 1] = 1.f + (rand() & 0xffff) / 65536.0f * 400.f; }
    CK(hipMemcpy(pml, ml.data(), ml.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&x0, (size_t)8 * D * 4)); fill_float(x0, (size_t)8 * D, 2.0f);
  CK(hipMalloc(&xa, (size_t)8 * D * 4)); CK(hipMalloc(&xb, (size_t)8 * D * 4));
  CK(hipMalloc(&bc, (size_t)D * 4)); fill_float(bc, D, 0.1f);
  CK(hipMalloc(&b1, (size_t)4 * D * 4)); fill_float(b1, 4 * D, 0.1f);
  CK(hipMalloc(&b2, (size_t)D * 4)); fill_float(b2, D, 0.1f);
  CK(hipMalloc(&xg, (size_t)8 * D * 8)); CK(hipMemset(xg, 0, (size_t)8 * D * 8));
  CK(hipMalloc(&hg, (size_t)8 * 2 * D * 8)); CK(hipMemset(hg, 0, (size_t)8 * 2 * D * 8));
  CK(hipMalloc(&tick, 4)); CK(hipMemset(tick, 0, 4));
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  CK(hipMalloc(&probe, 256 * 16 * 8)); CK(hipMemset(probe, 0, 256 * 16 * 8));

  auto three_launch = [&](int i, float* x) -> bool {
    const half_t* wc = W + wl * (i % L); const half_t* w1 = wc + (size_t)D * D; const half_t* w2 = w1 + (size_t)4 * D * D;
    whk::GemvArgs g; memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_COMBINE; g.part_o = po; g.part_ml = pml; g.splits = S; g.H = H; g.W = wc; g.bias = bc; g.N = D; g.K = D; g.R = R;
    g.epi = whk::EPI_RESID; g.resid = x; g.resid_ld = D;
    if (whk::launch_gemv(g, 1, st) != hipSuccess) return false;
    memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_LN; g.xf = x; g.xf_ld = D; g.ln_folded = 1; g.W = w1; g.bias = b1; g.N = 4 * D; g.K = D; g.R = R;
    g.epi = whk::EPI_GELU; g.y = h2; g.y_ld = 4 * D;
    if (whk::launch_gemv(g, 1, st) != hipSuccess) return false;
    memset(&g, 0, sizeof(g));
    g.pro = whk::PRO_PLAIN; g.x = h2; g.x_ld = 4 * D; g.W = w2; g.bias = b2; g.N = D; g.K = 4 * D; g.R = R;
    g.epi = whk::EPI_RESID; g.resid = x; g.resid_ld = D;
    return whk::launch_gemv(g, 1, st) == hipSuccess;
  };
  auto fused = [&](int variant, int i, float* x, long long* pr) {      // variant = 10 * DEPTH + MODE
    EngArgs a; memset(&a, 0, sizeof(a));
    a.part_o = po; a.part_ml = pml; a.Wc = W + wl * (i % L); a.bc = bc; a.W1 = a.Wc + (size_t)D * D; a.b1 = b1;
    a.W2 = a.W1 + (size_t)4 * D * D; a.b2 = b2; a.x = x; a.x_ld = D; a.xg = xg; a.hg = hg; a.d_tick = tick; a.epoch = i;
    a.err = err; a.R = R; a.probe = pr;
    switch (variant) {
      case 60: launch_tail3<6, 0>(a, st); break;
      case 61: launch_tail3<6, 1>(a, st); break;
      case 63: launch_tail3<6, 3>(a, st); break;
      case 43: launch_tail3<4, 3>(a, st); break;
      default: launch_tail3<6, 2>(a, st); break;
    }
  };

  // ---- numerics: one triple, both forms, from the same residual rows (and a second link, so that stale granules of the
  // first one are in the buffers)
  for (int nlinks = 1; nlinks <= 2; ++nlinks)
  for (int depth : {60, 61, 63, 43, 62}) {
    if (nlinks == 1 && depth != 60) continue;
    CK(hipMemcpyAsync(xa, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice, st));
    CK(hipMemcpyAsync(xb, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice, st));
    bool ok = three_launch(3, xa) && (nlinks == 1 || three_launch(4, xa));
    if (!ok) { printf("three-launch form failed to launch\n"); return 1; }
    fused(depth, 3, xb, nullptr); if (nlinks == 2) fused(depth, 4, xb, nullptr);
    printf("[%d link(s)] ", nlinks);
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, st, tick, N);
    CK(hipStreamSynchronize(st));
    std::vector<float> ha((size_t)8 * D), hb((size_t)8 * D), h0((size_t)8 * D);
    CK(hipMemcpy(ha.data(), xa, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), xb, hb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), x0, h0.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, mv = 0; size_t nd = 0, nbad = 0;
    for (size_t i = 0; i < (size_t)R * D; ++i) {
      if (!std::isfinite(hb[i])) { ++nbad; continue; }
      mx = std::max(mx, (double)fabsf(ha[i] - hb[i])); mv = std::max(mv, (double)fabsf(ha[i] - h0[i])); nd += ha[i] != hb[i];
    }
    if (nd) {
      printf("    rows that differ:");
      for (int r = 0; r < R; ++r) {
        size_t c = 0; double m = 0;
        for (int n = 0; n < D; ++n) { c += ha[(size_t)r * D + n] != hb[(size_t)r * D + n]; m = std::max(m, (double)fabsf(ha[(size_t)r * D + n] - hb[(size_t)r * D + n])); }
        if (c) printf(" row %d: %zu values, max %.3g;", r, c, m);
      }
      printf("\n");
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("numerics (R = %d, variant %d = 10 x loader depth + mode): one launch vs three, two links: max |d| %.3g, %zu of %d values differ, %zu not finite "
           "(the two links moved the rows by up to %.3g); spins that ran out %d\n", R, depth, mx, nd, R * D, nbad, mv, herr);
  }

  // ---- timing: 32-link chains
  struct Form { const char* name; int depth; };
  Form forms[] = {{"three launches (product: merge+cout | LN+FC1+GELU | FC2)", 0},
                  {"one launch, depth 6, sweeps from the start (round-5 v1)", 60},
                  {"one launch, depth 6, sampled polling", 61},
                  {"one launch, depth 6, sampled polling + loader holds FC2 back", 63},
                  {"one launch, depth 4, sampled polling + loader holds FC2 back", 43},
                  {"one launch, depth 6, loader holds FC2 back", 62}};
  for (const Form& f : forms) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipMemcpy(xa, x0, (size_t)8 * D * 4, hipMemcpyDeviceToDevice));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    bool ok = true;
    for (int i = 0; i < N; ++i) {
      if (f.depth == 0) ok = ok && three_launch(i, xa);
      else fused(f.depth, i, xa, i == N - 1 ? probe : nullptr);
    }
    if (f.depth != 0) hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, st, tick, N);
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
    printf("%-60s %6.2f us per triple (32-link graph chain, %d layers of weights)", f.name, best * 1e3f / N, L);
    if (f.depth != 0) {
      std::vector<long long> p((size_t)256 * 16);
      CK(hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost));
      long long t0 = p[0];
      for (int wg = 0; wg < 256; ++wg) t0 = std::min(t0, p[(size_t)wg * 16]);
      printf(" | spins that ran out %d\n    time line, us after the first workgroup's entry (min / median / max over 256 workgroups):\n", herr);
      const char* names[] = {"entry", "merged fragments complete", "x' published", "x' row gathered (wave 0)", "LayerNorm complete", "h published",
                             "h gathered (wave 0)", "FC2 + residual stored", "loader: 16 DMA issued", "loader: all landed",
                             "h: row-0 sample complete (wave 0)", "FC2 partial sums complete"};
      for (int s = 0; s < 12; ++s) {
        std::vector<double> d;
        for (int wg = 0; wg < 256; ++wg) d.push_back((p[(size_t)wg * 16 + s] - t0) / 100.0);
        std::sort(d.begin(), d.end());
        printf("      %-28s %6.2f / %6.2f / %6.2f\n", names[s], d[0], d[128], d[255]);
      }
    } else printf("\n");
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
