// xattn.hip — the two attention blocks of one decode step with their LayerNorm + projection INSIDE the launch (fp16, <= 8
// rows): xattn8_kernel (cross attention) and sattn8_kernel (self attention + KV-cache append).
//
// Reference ops: `cross_attn_ln` + `cross_attn.query` (whisper/model.py:39-50, 160-161) and `qkv_attention` over the
// cached cross K/V at T = 1 (model.py:106-139); `attn_ln` + `attn.query/key/value` + the cache append (model.py:152-153,
// 310-341) + the single-query attention over the cache.  The decode step used to run each as two dependent launches
// (LN -> projection, then attention).  The cross-attention K/V stream (61 MB at 8 rows of large-v3, 11.7 us) does not
// depend on the query — only the arithmetic after it does — so here every workgroup requests its K/V slice at entry and
// the projection runs UNDER that stream:
//
//   * grid (split, head, row) as before, 12 waves per workgroup: waves 0-3 are auxiliary, waves 4-11 hold the K/V tile
//     ("KV waves");
//   * the first D / 8 workgroups (dispatched first) are also PRODUCERS: their auxiliary waves compute one 8-feature
//     group of q for all rows with the MFMA diagonal tile of gemv8_kernel (weights requested first, LayerNorm of two
//     rows per wave in registers -> fp16 fragments in LDS -> 5 MFMAs -> cross-wave sum -> bias), scale by
//     d_head^-0.5 (0.125, exact in fp16: model.py:118-121 applies d_head^-0.25 to q and k) and publish the 64 values
//     as 32 GRANULES: 8-byte {2 x fp16, tag} words written by one write-through (sc1) store each;
//   * every workgroup's first auxiliary wave polls the 32 granules of its (row, head) — bounded — and hands q to the KV
//     waves through LDS and the workgroup barrier.  A granule is valid when its tag equals this launch's tag, so no flag,
//     fence or counter reset is needed (MI355X_MICROARCH.md "handoff-1to1").
//     tag = ((tick + 1 + epoch) << 6) | (layer + 1): `tick` is a device counter bumped once per decode step by the
//     step's last kernel and never reset, so tags never repeat during the life of a task.
//
// What the GPU taught (profiles/r03_probe_tail_fusion.txt): a CU serves its vector-memory requests in ISSUE ORDER.  With
// the auxiliary waves behind the KV waves (first version) the projection's loads and the polls sat behind 256 KB of K/V
// requests per CU and the fused launch took exactly as long as its two halves (15.4 = 11.7 + 3.7 us).  Hence: the
// auxiliary waves are the FIRST waves of the workgroup (their requests go out ahead of the K/V tile), the bias is
// requested with the weights, and the cross attention polls through the SCALAR memory path (s_load ... glc: its own port
// to L2, not queued behind the tile): 13.4 us per launch, and the step loses two launch boundaries per layer
// (1532 -> ~1460 us per token).
//
// Producers never wait for anything and are the lowest-numbered workgroups (cross) / consumers are the highest-numbered
// (self), so a resident consumer implies dispatched producers; every spin is bounded (a timeout counts in *err and lets
// the workgroup finish with whatever it has — bench.py and the tests assert the count is 0).
// q is bit-identical to the two-launch form; the fused self attention is bit-identical altogether; the fused cross
// attention sums each key range with 8 instead of 4 waves' partial sums (fp32 association only).
// A/B: WH_NO_FUSED_XATTN=1 / WH_NO_FUSED_SATTN=1 (process) or WH_TASK_TWO_LAUNCH_CROSS / _SELF (per task).
#include "common.h"
#include "kernels.h"

// developer probe (tools/probe_fused.cpp, -DWH_PROBE): lane 0 of the calling wave stamps slot `i` of its workgroup's record —
// slots 0-3 by auxiliary wave 0, slots 4-7 by the first K/V wave
#ifdef WH_PROBE
#define XPROBE(args, wg, i) do { if ((args).probe && lane == 0) (args).probe[(size_t)(wg) * 8 + (i)] = wall_clock64(); } while (0)   /* 100 MHz, one clock for the whole chip */
#else
#define XPROBE(args, wg, i) do {} while (0)
#endif

namespace {

typedef unsigned long long u64;
constexpr float XSCALE = 0.125f;      // d_head^-0.5 for d_head = 64
// Bound of every hand-off spin (one poll is 0.3 - 1 us: 2.5 - 8 ms in all).  A hang guard, not a scheduling tool: producers never
// wait, so a consumer only ever waits for a producer workgroup of its own launch to be DISPATCHED, which on a chip that is not held
// by something else for milliseconds takes microseconds.  (Round 5 ran 65 536 for a while: with three decode chains of fused self-
// AND cross-attention launches plus the encoder's stream on the GPU's four hardware queues, 8192 polls ran out in 3 of ~25 fresh
// processes.  Since round 6 the self attention of the default step is two plain launches and the headline is one wide chain, so the
// only spinning kernel left on the default path is xattn8_kernel of <= 8-row tasks; profiles/r06_lanes.txt counts its time-outs over
// 25 fresh processes with three such chains in flight.)  When a spin does run out the loop is re-run on the two-launch kernels (api.cpp).
constexpr int X_MAX_SPINS = 1 << 13;

__device__ __forceinline__ float qk_unit8(half8v q, half8v k) {
  float d = __builtin_amdgcn_fdot2(half2v{q[0], q[1]}, half2v{k[0], k[1]}, 0.f, false);
  d = __builtin_amdgcn_fdot2(half2v{q[2], q[3]}, half2v{k[2], k[3]}, d, false);
  d = __builtin_amdgcn_fdot2(half2v{q[4], q[5]}, half2v{k[4], k[5]}, d, false);
  d = __builtin_amdgcn_fdot2(half2v{q[6], q[7]}, half2v{k[6], k[7]}, d, false);
  return d;
}

// ---- the projection of ONE 8-feature group by 4 auxiliary waves (the MFMA diagonal tile of gemv8_kernel<PRO_LN>) -------
constexpr int P_NU = 5, P_KS = 4;       // K <= 1280: 20 blocks of 64 = 4 waves x 5 wave-loads

// stage 1a: request the wave's 5 wave-loads of weights (8 rows x 128 contiguous bytes each, non-temporal; lane l = 16 c +
// 8 half + i) and rows aw, aw + 4 of the fp32 residual stream (row-contiguous: a wave-load = 1 KB of one row) — all the
// memory requests of the projection, issued back to back
__device__ __forceinline__ void proj_issue(int aw, int lane, int g, const void* W, int K, const float* xf, int64_t xf_ld,
                                           int R, half8v (&wa)[P_NU], float4v (&v)[2][P_NU]) {
  const int nblk = K >> 6;
  const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;
  const uint32_t lane_off = ((uint32_t)(g * 8 + idx) * (uint32_t)K + (uint32_t)koff) * 2u;
#pragma unroll
  for (int u = 0; u < P_NU; ++u) {
    int blk = aw + P_KS * u; if (blk > nblk - 1) blk = nblk - 1;          // clamped; masked through x == 0
    wa[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)W + (size_t)blk * 128 + lane_off));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = aw + 4 * i;
    const char* src = (const char*)xf + (size_t)(row < R ? row : R - 1) * (size_t)xf_ld * 4;
#pragma unroll
    for (int j = 0; j < P_NU; ++j) {
      int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
      v[i][j] = *(const float4v*)(src + (uint32_t)k * 4u);
    }
  }
}

// stage 1b: LayerNorm of the two rows whole in registers (the affine part is folded into W / bias) and scatter to LDS
// in MFMA fragment order
__device__ __forceinline__ void proj_ln(int aw, int lane, int K, const float4v (&v)[2][P_NU], half8v* xfrag) {
  const float invK = 1.0f / (float)K;
  const uint32_t fbase = (uint32_t)((lane >> 4) * P_NU * 64 + 16 * ((lane >> 1) & 3) + 8 * ((lane >> 3) & 1)) * 16u + (uint32_t)(lane & 1) * 8u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = aw + 4 * i;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < P_NU; ++j) {
      const float t = (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
      sum += ((j * 64 + lane) * 4 < K) ? t : 0.f;
    }
    const float mean = wave_sum(sum) * invK;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < P_NU; ++j) {
      if ((j * 64 + lane) * 4 < K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean; ss = __builtin_fmaf(d, d, ss); }
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) * invK + 1e-5f);
    const uint32_t rbase = fbase + (uint32_t)(row * 16);
#pragma unroll
    for (int j = 0; j < P_NU; ++j) {
      const bool on = (j * 64 + lane) * 4 < K;
      half4v o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = on ? (half_t)((v[i][j][e] - mean) * rstd) : (half_t)0.f;
      *(half4v*)((char*)xfrag + rbase + (uint32_t)(j * 1024)) = o4;
    }
  }
}

__device__ __forceinline__ void proj_stage1(int aw, int lane, int g, const void* W, int K, const float* xf, int64_t xf_ld,
                                            int R, half8v (&wa)[P_NU], half8v* xfrag) {
  float4v v[2][P_NU];
  proj_issue(aw, lane, g, W, K, xf, xf_ld, R, wa, v);
  proj_ln(aw, lane, K, v, xfrag);
}

// stage 2 (between the two barriers): 5 MFMAs against the x fragments, the two meaningful diagonal blocks summed, the
// wave's 8 x 8 partial sums to pred[aw][feature][row]
__device__ __forceinline__ void proj_stage2(int aw, int lane, const half8v (&wa)[P_NU], const half8v* xfrag, float (*pred)[8][8]) {
  const bool diag = (lane >> 5) == ((lane >> 3) & 1);
  float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < P_NU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xfrag[(aw * P_NU + u) * 64 + lane], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float z = diag ? acc[e] : 0.f;
    z += lane_xor8(z);
    float p, q; lane_swap32(z, p, q);
    acc[e] = p + q;
  }
  if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pred[aw][4 * (lane >> 4) + e][lane & 7] = acc[e];
  }
}

// one wave fetches 32 granules (64 fp16) into LDS: relaxed agent-scope 8-byte loads until every tag matches; bounded
__device__ __forceinline__ void fetch_granules(const u64* gp, uint32_t tag, int lane, uint32_t* dst, int* err, int max_spins) {
  uint32_t data = 0;
  int spins = 0;
  for (;;) {
    const u64 g = __hip_atomic_load(gp + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    data = (uint32_t)g;
    const bool ok = (uint32_t)(g >> 32) == tag;
    if (__all(ok)) break;
    if (++spins >= max_spins) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(4);
  }
  if (lane < 32) dst[lane] = data;
}

// The same fetch through the SCALAR memory path (mode bit 0).  A CU serves its vector-memory requests in issue order, so
// a poll issued behind a workgroup's K/V tile (128 KB per workgroup, two workgroups per CU) only returns when that stream
// has drained — measured: the fused cross attention took exactly as long as its two launches.  s_load goes through the
// scalar cache's own port to L2; `glc` makes every poll miss the scalar cache.  32 granules = 256 contiguous bytes =
// four s_load_dwordx16 into 64 SGPRs; the tags are compared on the scalar ALU and lane j picks data dword 2 j.
typedef int int16s __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void fetch_granules_scalar(const u64* gp, uint32_t tag, int lane, uint32_t* dst, int* err, int max_spins) {
  const u64* p = gp;                     // wave-uniform by construction (block index, wave index)
  int16s r0, r1, r2, r3;
  int spins = 0;
  for (;;) {
    asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\t"
                 "s_load_dwordx16 %1, %4, 0x40 glc\n\t"
                 "s_load_dwordx16 %2, %4, 0x80 glc\n\t"
                 "s_load_dwordx16 %3, %4, 0xc0 glc\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "s"(p) : "memory");
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      ok = ok && (uint32_t)r0[2 * e + 1] == tag && (uint32_t)r1[2 * e + 1] == tag && (uint32_t)r2[2 * e + 1] == tag &&
           (uint32_t)r3[2 * e + 1] == tag;
    if (ok) break;
    if (++spins >= max_spins) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(2);
  }
  uint32_t val = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    val = lane == e ? (uint32_t)r0[2 * e] : val;
    val = lane == 8 + e ? (uint32_t)r1[2 * e] : val;
    val = lane == 16 + e ? (uint32_t)r2[2 * e] : val;
    val = lane == 24 + e ? (uint32_t)r3[2 * e] : val;
  }
  if (lane < 32) dst[lane] = val;
}

// a pair of fp16 values (this lane's and its xor-1 neighbour's) as one granule, written through to L2 by ONE 8-byte store
__device__ __forceinline__ void publish_pair(u64* slot, half_t mine_h, int lane, bool even, bool on, uint32_t tag) {
  const uint32_t mine = (uint32_t)__builtin_bit_cast(unsigned short, mine_h);
  const uint32_t other = __float_as_uint(lane_xor1(__uint_as_float(mine)));
  if (even && on) {
    const u64 g = ((u64)tag << 32) | (u64)(mine | (other << 16));
    __hip_atomic_store(slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// barrier among `n_waves_total / per-phase` auxiliary waves through an LDS arrival counter: lane 0 of each wave adds 1 after
// the wave's own LDS writes (in-order in the LDS pipe; release fence for the compiler), everyone spins until the counter
// reaches `target` (cumulative over the phases), acquire fence before the reads that follow
// — a spin that runs out (the waves of a resident workgroup always make progress, so: never) counts in *err like the
// hand-off spins do, and the step's result is then discarded by the caller (WH_ERR_HANDOFF)
__device__ __forceinline__ void aux_barrier(int* cnt, int target, int lane, int* err) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  int spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    if (++spins >= (1 << 20)) { if (lane == 0 && err) atomicAdd(err, 1); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- phase 0 of the cross-attention launch (OUT0): `attn.out` + the residual add (model.py:153) by the producers' auxiliary
// waves, handed to the LayerNorm of the SAME launch — a round-4 experiment, REJECTED, compiled into development builds
// (-DWH_DEV) only.  Measured (tools/probe_fused, profiles/r04_probe_out0_fusion.txt): 19.7 us for the one launch against
// 17.0 us for attn.out (3.85) followed by the fused cross attention (13.0).  The time line says why: the rows are
// published 4.4 us into the launch, but the producers' LayerNorm input arrives 8 us later — their flag polls and row
// loads are vector-memory requests and queue behind the 256 KB of K/V requests of their own CU (a CU serves them in
// issue order: the lesson of round 3, which is why the q hand-off polls through the scalar path), and 40 KB of rows per
// producer is too much for the scalar path.  The first build was also not bit-identical to the separate launch
// (asm loads without a data dependence on their wait); the loads below now carry one.  Kept for the record: ---------------
// The output projection of the self attention used to be a launch of its own between the two attention launches (3.3 MB of
// weights: 3.85 us per link, most of it boundary and ramp).  Its weights do not depend on anything, and the cross
// attention's 61 MB K/V stream is requested at entry and takes ~10 us to land whatever the auxiliary waves do meanwhile —
// so the projection moves UNDER that stream: producer workgroup g (g < D / 8) computes x'[:, 8g .. 8g+8) = x + W_out att + b
// for all rows with the MFMA diagonal tile (PRO_PLAIN form of gemv8_kernel: same fragments, same order of sums — the rows
// are bit-identical to the separate launch), stores them IN PLACE with write-through (sc1) stores, drains, and raises its
// flag {tag}; every producer then waits for all D / 8 flags (bounded), reads the finished rows with L1-bypassing loads
// (MI355X_MICROARCH.md, valid forms: sc1 payload -> vmcnt(0) -> sc1 flag; sc1 loads after the poll) and goes on with the
// LayerNorm + query projection as before.  Producers wait for producers only, and those are the first workgroups of the
// grid: on an idle device all are resident at once; if some are not (a shared GPU), the bounded polls run out, the count
// in *err moves and the caller re-runs the step on the two-launch kernels, as for every other hand-off here.
__device__ __forceinline__ float4v load_f4_sc(const void* p) {
  float4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// requests of phase 0: the wave's 5 wave-loads of W_out rows 8g .. 8g+7 (non-temporal) and the matching fragments of the
// attention rows (fp16 [R][D], written by the previous launch: plain loads); K blocks beyond D / 64 are zeroed at use
__device__ __forceinline__ void out0_issue(int aw, int lane, int g, const void* Wo, const void* att, int K, int R,
                                           half8v (&wo)[P_NU], half8v (&xo)[P_NU]) {
  const int nblk = K >> 6;
  const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;
  const uint32_t w_off = ((uint32_t)(g * 8 + idx) * (uint32_t)K + (uint32_t)koff) * 2u;
  const uint32_t x_off = ((uint32_t)(idx < R ? idx : R - 1) * (uint32_t)K + (uint32_t)koff) * 2u;
#pragma unroll
  for (int u = 0; u < P_NU; ++u) {
    int blk = aw + P_KS * u; if (blk > nblk - 1) blk = nblk - 1;
    wo[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)Wo + (size_t)blk * 128 + w_off));
  }
#pragma unroll
  for (int u = 0; u < P_NU; ++u) {
    int blk = aw + P_KS * u; if (blk > nblk - 1) blk = nblk - 1;
    xo[u] = *(const half8v*)((const char*)att + (size_t)blk * 128 + x_off);
  }
}

// 5 MFMAs, the two meaningful diagonal blocks summed, the wave's 8 x 8 partial sums to pred[aw][feature][row]
__device__ __forceinline__ void out0_mfma(int aw, int lane, int K, const half8v (&wo)[P_NU], half8v (&xo)[P_NU], float (*pred)[8][8]) {
  const int nblk = K >> 6;
  const bool diag = (lane >> 5) == ((lane >> 3) & 1);
  float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < P_NU; ++u) {
    if (!(aw + P_KS * u < nblk)) {
#pragma unroll
      for (int e = 0; e < 8; ++e) xo[u][e] = (half_t)0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo[u], xo[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float z = diag ? acc[e] : 0.f;
    z += lane_xor8(z);
    float p, q; lane_swap32(z, p, q);
    acc[e] = p + q;
  }
  if (lane < 32 && (lane & 15) < 8) {
#pragma unroll
    for (int e = 0; e < 4; ++e) pred[aw][4 * (lane >> 4) + e][lane & 7] = acc[e];
  }
}

// rows aw, aw + 4 of the finished residual stream, row-contiguous, past the L1 (the producers stored write-through)
__device__ __forceinline__ void out0_rows(int aw, int lane, int K, const float* x, int64_t x_ld, int R, float4v (&v)[2][P_NU]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = aw + 4 * i;
    const char* src = (const char*)x + (size_t)(row < R ? row : R - 1) * (size_t)x_ld * 4;
#pragma unroll
    for (int j = 0; j < P_NU; ++j) {
      int k = (j * 64 + lane) * 4; if (k > K - 4) k = K - 4;
      v[i][j] = load_f4_sc(src + (uint32_t)k * 4u);
    }
  }
  // the compiler does not know that an asm load's result is pending: every register is an in/out operand of the wait, so
  // that no use can be scheduled above it
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < P_NU; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i][j]) : : "memory");
}

// Both kernels below are written as TWO role bodies under one wave-uniform branch — the KV waves and the auxiliary waves
// never share a basic block — so that the register allocator sees each role's pressure on its own (the K/V tile of the
// KV waves is 128 VGPRs; merged control flow carried it through the projection code: 209 VGPRs, one workgroup per CU).
// Every role executes the same number of s_barrier instructions on every path (counted in the comments: B0 ... B5); the
// two synchronisation points inside the cross-attention projection are LDS arrival counters among the auxiliary waves.

// 8 KV waves x NL rounds of 64 keys (NL = 4 / 6 / 8: up to 512 keys per split) + 4 auxiliary waves.  The two-launch form
// holds the same 512 keys in 4 waves x 16 rounds; here the tile must stay live across the hand-off barrier (the compiler
// cannot retire K registers into scores early), and 16 rounds (156 VGPRs) would leave room for one workgroup per CU only:
// with 8 rounds the kernel needs 80 VGPRs, two 12-wave workgroups fit a CU and all 480 workgroups of large-v3 x 8 rows are
// resident at once — which is what keeps 61 MB of loads in flight.
template <int NL, bool OUT0>
__global__ __launch_bounds__(768, 6) void xattn8_kernel(whk::XAttnArgs a) {
  pin_kernargs(a);
  constexpr int WAVES = 8;               // KV waves (waves 4 .. 11); the auxiliary waves are waves 0 .. 3: launched first
  constexpr int AUX = 4;
  constexpr int KPR = WAVES * 8;         // keys per round
  __shared__ __attribute__((aligned(16))) half8v xfrag[P_KS * P_NU * 64];   // LayerNorm output, 8 rows, MFMA fragment order
  __shared__ float pred[P_KS][8][8];     // producer: [auxiliary wave][feature][row] partial sums
  __shared__ __attribute__((aligned(16))) uint32_t qsh[32];                 // q of this (row, head): 64 fp16, scaled
  __shared__ float red[WAVES][64];
  __shared__ float redm[WAVES], reds[WAVES];
  __shared__ int aux_cnt;                // arrival counter of the auxiliary waves (producer workgroups)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
  const int S = a.splits, D = a.D;
  const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const bool producer = wgid < (D >> 3); // workgroup-uniform: the first D / 8 workgroups (dispatched first)
  // Producer workgroups: the two synchronisation points INSIDE the projection involve the 4 auxiliary waves only (an LDS
  // arrival counter, aux_barrier) — probe_fused showed the K/V waves still issuing their 16 loads 4 us into the launch
  // (the CU's request queue pushes back), and a workgroup barrier there held the MFMA stage until then: q was published
  // after 5.9 us instead of ~2.5.  The counter needs a defined start: one workgroup barrier at entry, before any request.
  if (producer) {
    if (tid == 0) aux_cnt = 0;
    __syncthreads();                                 // B0
  }

  if (wave < AUX) {
    // ================= auxiliary waves =================
    const int aw = wave;
    if (aw == 0) XPROBE(a, wgid, 0);
    const int vtick = load_agent_int(a.d_tick);      // requested first, read (below) behind the projection's requests
    uint32_t tag;
    if (producer) {
      half8v wa[P_NU];
      float4v xv[2][P_NU];
      int bar = 0;                                   // cumulative target of the auxiliary waves' arrival counter
      if constexpr (OUT0) {
        // ---- phase 0: x' = x + W_out att + b for feature group `wgid`, all rows; requests first, the query weights behind
        half8v wo[P_NU], xo[P_NU];
        out0_issue(aw, lane, wgid, a.out_w, a.att_in, D, a.R, wo, xo);
        float ob = 0.f, ores = 0.f;
        if (aw == 0) {                               // lane = 8 row + feature
          ob = a.out_b[wgid * 8 + (lane & 7)];
          ores = a.x_io[(int64_t)((lane >> 3) < a.R ? (lane >> 3) : a.R - 1) * a.xf_ld + wgid * 8 + (lane & 7)];
        }
        {
          const int nblk = D >> 6;
          const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;
          const uint32_t lane_off = ((uint32_t)(wgid * 8 + idx) * (uint32_t)D + (uint32_t)koff) * 2u;
#pragma unroll
          for (int u = 0; u < P_NU; ++u) {
            int blk = aw + P_KS * u; if (blk > nblk - 1) blk = nblk - 1;
            wa[u] = WH_WEIGHT_LOAD((const half8v*)((const char*)a.W + (size_t)blk * 128 + lane_off));
          }
        }
        const float bias0 = a.bias[wgid * 8 + (lane & 7)];
        asm volatile("" ::: "memory");
        out0_mfma(aw, lane, D, wo, xo, pred);
        tag = ((uint32_t)(uniform(vtick) + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
        aux_barrier(&aux_cnt, bar += AUX, lane, a.err);
        if (aw == 0) {
          const int er = lane >> 3, ej = lane & 7;
          float val = ob;
#pragma unroll
          for (int k = 0; k < P_KS; ++k) val += pred[k][ej][er];
          if (er < a.R)
            __hip_atomic_store(a.x_io + (int64_t)er * a.xf_ld + wgid * 8 + ej, ores + val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the storing wave drains, THEN the flag
          if (lane == 0) __hip_atomic_store(a.pflags + wgid, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          XPROBE(a, wgid, 3);                        // phase 0 published
        }
        {
          // every auxiliary wave polls its share of the D / 8 flags (one 8-byte agent-scope load per lane and pass)
          const int nprod = D >> 3;
          const int fi = aw * 64 + lane;
          const u64* fp = a.pflags + (fi < nprod ? fi : nprod - 1);
          const int max_spins = (a.mode & 4) ? 1 : X_MAX_SPINS;
          int spins = 0;
          if (aw * 64 < nprod) {
            for (;;) {
              const u64 f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__all((uint32_t)f == tag)) break;
              if (++spins >= max_spins) { if (lane == 0 && a.err) atomicAdd(a.err, 1); break; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
        }
        aux_barrier(&aux_cnt, bar += AUX, lane, a.err);       // all flags seen by all four waves (and pred is free again)
        out0_rows(aw, lane, D, a.x_io, a.xf_ld, a.R, xv);
        const float bias = bias0;
        proj_ln(aw, lane, D, xv, xfrag);
        aux_barrier(&aux_cnt, bar += AUX, lane, a.err);
        proj_stage2(aw, lane, wa, xfrag, pred);
        aux_barrier(&aux_cnt, bar += AUX, lane, a.err);
        if (aw == 0) {
          const int er = lane >> 3, ej = lane & 7;
          const int n = wgid * 8 + ej;
          float val = bias;
#pragma unroll
          for (int k = 0; k < P_KS; ++k) val += pred[k][ej][er];
          const half_t qh = (half_t)val;
          const half_t qsc = (half_t)((float)qh * XSCALE);
          publish_pair(a.qg + (size_t)er * (D >> 1) + (n >> 1), qsc, lane, (ej & 1) == 0, er < a.R, tag);
        }
      } else {
      proj_issue(aw, lane, wgid, a.W, D, a.xf, a.xf_ld, a.R, wa, xv);
      const float bias = a.bias[wgid * 8 + (lane & 7)];          // requested with the rest, used after the MFMAs
      tag = ((uint32_t)(uniform(vtick) + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
      proj_ln(aw, lane, D, xv, xfrag);
      aux_barrier(&aux_cnt, AUX, lane, a.err);              // the 4 auxiliary waves only: the K/V waves are still issuing
      proj_stage2(aw, lane, wa, xfrag, pred);
      aux_barrier(&aux_cnt, 2 * AUX, lane, a.err);
      if (aw == 0) {                                 // 64 outputs: lane = 8 row + feature
        const int er = lane >> 3, ej = lane & 7;
        const int n = wgid * 8 + ej;
        float val = bias;
#pragma unroll
        for (int k = 0; k < P_KS; ++k) val += pred[k][ej][er];
        const half_t qh = (half_t)val;                 // what the two-launch form stores ...
        const half_t qsc = (half_t)((float)qh * XSCALE);   // ... and what its attention kernel makes of it (exact)
        publish_pair(a.qg + (size_t)er * (D >> 1) + (n >> 1), qsc, lane, (ej & 1) == 0, er < a.R, tag);
      }
      }
    } else {
      tag = ((uint32_t)(uniform(vtick) + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
    }
    if (aw == 0) XPROBE(a, wgid, 1);                 // producers: published
    // every workgroup: auxiliary wave 0 fetches the q granules of (row r, head h)
    if (aw == 0) {
      const u64* gp = a.qg + (size_t)r * (D >> 1) + h * 32;
      if (a.mode & 1) fetch_granules_scalar(gp, tag, lane, qsh, a.err, (a.mode & 4) ? 1 : X_MAX_SPINS);
      else fetch_granules(gp, tag, lane, qsh, a.err, (a.mode & 4) ? 1 : X_MAX_SPINS);
      XPROBE(a, wgid, 2);                            // q fetched
    }
    __syncthreads();                                 // B3: q is in LDS
    __syncthreads();                                 // B4
    __syncthreads();                                 // B5
    return;
  }

  // ================= KV waves: the whole K / V slice of the split, requested before anything else =================
  const int kw = wave - AUX;             // KV wave index 0 .. 7
  const int Tk = a.Tk;
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + KPR - 1) / KPR * KPR;
  const int k0 = s * chunk;
  int k1 = k0 + chunk; if (k1 > Tk) k1 = Tk;
  const int nkeys = k1 > k0 ? k1 - k0 : 0;
  const int cu = lane & 7, ks = lane >> 3;
  const int kk0 = kw * 8 + ks;
  half8v ku[NL], vu[NL];
  {
#ifdef WH_PROBE      // tools/probe_fused: the same kernel on a head-major K / V layout (a key row = 128 contiguous bytes per head)
    const int64_t hs = a.kv_hs ? a.kv_hs : 64;
    const half_t* kp = (const half_t*)a.k + (int64_t)r * a.k_bs + h * hs + cu * 8;
    const half_t* vp = (const half_t*)a.v + (int64_t)r * a.v_bs + h * hs + cu * 8;
#else
    const half_t* kp = (const half_t*)a.k + (int64_t)r * a.k_bs + h * 64 + cu * 8;
    const half_t* vp = (const half_t*)a.v + (int64_t)r * a.v_bs + h * 64 + cu * 8;
#endif
    const int klast = nkeys > 0 ? nkeys - 1 : 0;
    const uint32_t ldk = (uint32_t)a.k_ld, ldv = (uint32_t)a.v_ld;
    const uint32_t ok0 = (uint32_t)(k0 + kk0) * ldk, okl = (uint32_t)(k0 + klast) * ldk;
    const uint32_t ov0 = (uint32_t)(k0 + kk0) * ldv, ovl = (uint32_t)(k0 + klast) * ldv;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      uint32_t o = ok0 + (uint32_t)(i * KPR) * ldk; if (o > okl) o = okl;
      ku[i] = __builtin_nontemporal_load((const half8v*)(kp + o));
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      uint32_t o = ov0 + (uint32_t)(i * KPR) * ldv; if (o > ovl) o = ovl;
      vu[i] = __builtin_nontemporal_load((const half8v*)(vp + o));
    }
  }
  if (kw == 0) XPROBE(a, wgid, 4);                   // K/V requests issued
  __syncthreads();                                   // B3: q is in LDS
  if (kw == 0) XPROBE(a, wgid, 5);

  // scores, softmax statistics, p . V (attn_decode_kernel, SKIP = false)
  float sc[NL];
  float mx = WH_NEG_INF;
  {
    const half8v qs = *(const half8v*)((const char*)qsh + cu * 16);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float d = qk_unit8(qs, ku[i]);
      d = group8_sum(d);
      sc[i] = (kk0 + i * KPR < nkeys) ? d : WH_NEG_INF;
      mx = fmaxf(mx, sc[i]);
    }
    mx = across_groups8_max(mx);
    if (lane == 0) redm[kw] = mx;
  }
  if (kw == 0) XPROBE(a, wgid, 6);                   // scores done: the keys have arrived
  __syncthreads();                                   // B4
  mx = redm[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, redm[w]);
  float acc[8];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const float p = (sc[i] == WH_NEG_INF) ? 0.f : __expf(sc[i] - mx);
    sum += p;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)vu[i][e], acc[e]);
  }
  sum = across_groups8_sum(sum);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = across_groups8_sum(acc[e]);
  if (ks == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[kw][cu * 8 + e] = acc[e];
  }
  if (lane == 0) reds[kw] = sum;
  __syncthreads();                                   // B5
  if (kw == 0) {
    float o = red[0][lane], l = reds[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red[w][lane]; l += reds[w]; }
    if (S == 1) {
      ((half_t*)a.out)[(int64_t)r * a.o_ld + h * 64 + lane] = (half_t)(o / l);
    } else {
      const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
      ((half_t*)a.part_o)[pi * 64 + lane] = (half_t)(nkeys > 0 ? o / l : 0.f);
      if (lane == 0) {
        a.part_ml[pi * 2 + 0] = nkeys > 0 ? mx : WH_NEG_INF;
        a.part_ml[pi * 2 + 1] = nkeys > 0 ? l : 0.f;
      }
    }
    XPROBE(a, wgid, 7);
  }
}

// =====================================================================================================================
// self attention of one decode step with its LayerNorm + QKV projection inside the launch (fp16, <= 8 rows).
// `attn_ln` + `attn.query / key / value` (model.py:39-50, 152-153) + the KV-cache append (model.py:310-341) + the
// single-query attention over the cache (model.py:114-139).  One workgroup per 8-feature group of the 3D outputs
// (480 for large-v3): 4 auxiliary waves project (proj_stage1 / 2), auxiliary wave 0 writes the outputs — q scaled by
// d_head^-0.5 as a granule; k, v unscaled as granules AND into the cache row of this step (for the steps to come).
// The LAST H x R workgroups are also CONSUMERS of one (head, row): their 8 KV waves request the cached keys / values
// of the earlier positions at entry (they do not depend on this step), three auxiliary waves then fetch the 3 x 32
// granules of q and of the NEW key / value of their head, which take the place of the cache row in the register tile.
// Consumers sit at the end of the dispatch order: when one is resident every producer has been dispatched, so no
// occupancy can starve them.  In the other workgroups the 8 KV waves exit at once.
// Arithmetic and order as gemv8_kernel<PRO_LN> (EPI_QKV) + attn_decode_kernel<half, 8, 8, true>: bit-identical.
// 7 rounds x 64 keys cover n_text_ctx = 448 (the register tile has to stay live across the hand-off barrier; with 8
// rounds the kernel spilled at the 80 VGPRs that let two 12-wave workgroups share a CU).
// =====================================================================================================================
__global__ __launch_bounds__(768, 6) void sattn8_kernel(whk::SAttnArgs a) {
  pin_kernargs(a);
  constexpr int NL = 7, WAVES = 8, KPR = WAVES * 8;      // 7 rounds x 64 keys = 448 cached positions = n_text_ctx
  __shared__ __attribute__((aligned(16))) half8v xfrag[P_KS * P_NU * 64];
  __shared__ float pred[P_KS][8][8];
  __shared__ __attribute__((aligned(16))) uint32_t qkv_sh[3][32];           // q (scaled), new k, new v of (row, head)
  __shared__ float red[WAVES][64];
  __shared__ float redm[WAVES], reds[WAVES];
#ifdef WH_DEV
  __shared__ __attribute__((aligned(16))) half8v xfrag2[P_KS * P_NU * 64];  // output projection: attention rows, fragment order
  __shared__ float pred2[P_KS][8][8];
  __shared__ int cnt_aux, cnt_out;                 // arrival counters (output-projection workgroups)
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = a.D, H = a.H, R = a.R;
  const int wgid = blockIdx.x, nwg = gridDim.x;
  const int cidx = wgid - (nwg - H * R);           // consumer index, >= 0 in the last H * R workgroups
  const bool consumer = cidx >= 0;                 // workgroup-uniform
  const int r = consumer ? cidx / H : 0, h = consumer ? cidx - (cidx / H) * H : 0;
  // Third stage — DEVELOPMENT BUILDS ONLY (-DWH_DEV; a rejected experiment kept reproducible: bit-identical, 16.3 us per
  // launch against 10.5 + 4.8 for the two launches, profiles/r03_probe_fused_out_stage.txt; the shipped kernel does not
  // contain it) — (x_out != null): `attn.out` + the residual add (model.py:153) in the same launch.  The FIRST D / 8
  // workgroups — never consumers — use their 8 otherwise idle K/V waves for it: waves 0-3 request the 8 x D weight rows
  // of feature group `wgid` at entry, each of the 8 waves gathers the attention output of ONE row (all heads) from the
  // granules the consumers publish (a.og) into LDS in MFMA fragment order; 5 MFMAs per weight wave, bias + residual, fp32 store into the
  // OTHER residual buffer (x_out; the launch's own LayerNorm input a.xf is still being read by late workgroups).
  // Inside these workgroups nothing uses the workgroup barrier after B0: the two groups of waves synchronise through LDS
  // arrival counters (aux_barrier), so neither waits for the other.
#ifdef WH_DEV
  const bool out_wg = a.x_out != nullptr && wgid < (D >> 3);
  if (out_wg) {
    if (tid == 0) { cnt_aux = 0; cnt_out = 0; }
    __syncthreads();                               // B0
  }
#endif

  if (wave >= WAVES) {
    // ================= auxiliary waves: the projection of feature group `wgid` (every workgroup) =================
    const int aw = wave - WAVES;
    if (aw == 0) XPROBE(a, wgid, 0);
    // everything the epilogue needs is requested here, ahead of the weights: the tick, the position, this lane's bias
    // and its row's lag (a load issued in the epilogue would be one more dependent L2 round trip at the end)
    const int vtick = load_agent_int(a.d_tick);
    const int vpos = load_agent_int(a.d_pos);                         // used by the cache append only
    const float bias = a.bias[wgid * 8 + (lane & 7)];
    const int rlag = (a.lag && (lane >> 3) < R) ? a.lag[lane >> 3] : 0;
    half8v wa[P_NU];
    proj_stage1(aw, lane, wgid, a.W, D, a.xf, a.xf_ld, R, wa, xfrag);
    const uint32_t tag = ((uint32_t)(uniform(vtick) + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
#ifdef WH_DEV
    if (out_wg) aux_barrier(&cnt_aux, 4, lane, a.err); else
#endif
    __syncthreads();                                 // B1
    proj_stage2(aw, lane, wa, xfrag, pred);
#ifdef WH_DEV
    if (out_wg) aux_barrier(&cnt_aux, 8, lane, a.err); else
#endif
    __syncthreads();                                 // B2
    if (aw == 0) {                                   // 64 outputs: lane = 8 row + feature
      const int er = lane >> 3, ej = lane & 7;
      const int n = wgid * 8 + ej;                   // output feature in [0, 3D)
      float val = bias;
#pragma unroll
      for (int k = 0; k < P_KS; ++k) val += pred[k][ej][er];
      const half_t vh = (half_t)val;
      const bool on = er < R;
      half_t pub = vh;
      if (n < D) {
        pub = (half_t)((float)vh * XSCALE);          // q: scaled for the attention (exact in fp16)
        if (on && a.q_out) ((half_t*)a.q_out)[(int64_t)er * D + n] = vh;
      } else if (on) {
        const int64_t pos = (int64_t)vpos - rlag;
        if (n < 2 * D) ((half_t*)a.kcache)[(int64_t)er * a.cache_bs + pos * D + (n - D)] = vh;
        else ((half_t*)a.vcache)[(int64_t)er * a.cache_bs + pos * D + (n - 2 * D)] = vh;
      }
      publish_pair(a.qg + (size_t)er * (3 * D >> 1) + (n >> 1), pub, lane, (ej & 1) == 0, on, tag);
      XPROBE(a, wgid, 1);                            // published
    }
    if (!consumer) return;
    // consumers: three auxiliary waves fetch the granules of q, new k, new v of (row r, head h)
    if (aw < 3) {
      const u64* gp = a.qg + (size_t)r * (3 * D >> 1) + aw * (D >> 1) + h * 32;
      if (a.mode & 1) fetch_granules_scalar(gp, tag, lane, qkv_sh[aw], a.err, (a.mode & 4) ? 1 : X_MAX_SPINS);
      else fetch_granules(gp, tag, lane, qkv_sh[aw], a.err, (a.mode & 4) ? 1 : X_MAX_SPINS);
      if (aw == 0) XPROBE(a, wgid, 2);               // q fetched
    }
    __syncthreads();                                 // B3
    __syncthreads();                                 // B4
    __syncthreads();                                 // B5
    return;
  }

#ifdef WH_DEV
  // ================= waves 0-7 of the first D / 8 workgroups: output projection + residual =================
  if (out_wg) {
    const int ow = wave;                             // also the attention row this wave gathers
    const int vtick = load_agent_int(a.d_tick);
    half8v wa[P_NU];
    float e_res = 0.f, e_bias = 0.f;
    if (ow < 4) {
      // weight waves: the 8 x D rows of feature group `wgid` of attn.out, requested at entry (lane l = 16 c + 8 half + i)
      const int nblk = D >> 6;
      const int idx = lane & 7, koff = ((lane >> 3) & 1) * 32 + (lane >> 4) * 8;
      const uint32_t lane_off = ((uint32_t)(wgid * 8 + idx) * (uint32_t)D + (uint32_t)koff) * 2u;
#pragma unroll
      for (int u = 0; u < P_NU; ++u) {
        int blk = ow + P_KS * u; if (blk > nblk - 1) blk = nblk - 1;
        wa[u] = __builtin_nontemporal_load((const half8v*)((const char*)a.out_w + (size_t)blk * 128 + lane_off));
      }
      if (ow == 0) {                                 // epilogue operands: lane = 8 row + feature
        const int er = lane >> 3, n = wgid * 8 + (lane & 7);
        e_bias = a.out_b[n];
        e_res = a.xf[(int64_t)(er < R ? er : R - 1) * a.xf_ld + n];
      }
    }
    const uint32_t tag = ((uint32_t)(uniform(vtick) + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
    // ---- every one of the 8 waves gathers ONE attention row (row = wave).  Its 160 fragment units are zeroed first (a row
    // >= R and K blocks >= D / 64 stay zero), then every granule {2 fp16 of (row, k), tag} lands as one ds_write_b32 at the
    // fragment position of k: unit ((k >> 6) % 4) * 5 + (k >> 6) / 4, lane 16 ((k & 31) >> 3) + 8 ((k & 63) >> 5) + row,
    // element k & 7.  The 10 granules of a lane (10 sweeps of 64 over the row) are requested back to back and checked
    // together: a pass costs one L2 round trip, not ten (the first version polled sweep by sweep: + 9 us per launch).
    {
      half8v z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (half_t)0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int j = lane + 64 * i;                 // 0 .. 159: (K-block unit, (c, half) slot)
        if (j < 160) xfrag2[(j >> 3) * 64 + (j & 7) * 8 + ow] = z;
      }
    }
    if (ow < R) {                                    // wave-uniform
      // no polling while nothing can have been published: wait for this workgroup's own projection (LDS counter), then
      // about as long again — the consumers still have their attention to do
      int spins = 0;
      while (__hip_atomic_load(&cnt_aux, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 && ++spins < (1 << 20))
        __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_s_sleep(48);
      const int ngr = D >> 1;                        // granules per row
      constexpr int NSW = 10;                        // sweeps of 64 granules per row (D <= 1280)
      const u64* gp = a.og + (size_t)ow * ngr;
      uint32_t data[NSW];
      int sp = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NSW; ++i) {
          const int gi = i * 64 + lane;
          const u64 gv = __hip_atomic_load(gp + (gi < ngr ? gi : ngr - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          data[i] = (uint32_t)gv;
          ok = ok && (gi >= ngr || (uint32_t)(gv >> 32) == tag);
        }
        if (__all(ok)) break;
        if (++sp >= X_MAX_SPINS) { if (lane == 0 && a.err) atomicAdd(a.err, 1); break; }
        __builtin_amdgcn_s_sleep(4);
      }
#pragma unroll
      for (int i = 0; i < NSW; ++i) {
        const int gi = i * 64 + lane;
        if (gi < ngr) {
          const int k = 2 * gi, blk = k >> 6;
          const uint32_t unit = (uint32_t)(((blk & 3) * P_NU + (blk >> 2)) * 64 + 16 * ((k & 31) >> 3) + 8 * ((k & 63) >> 5) + ow);
          *(uint32_t*)((char*)xfrag2 + unit * 16u + (uint32_t)(k & 7) * 2u) = data[i];
        }
      }
    }
    aux_barrier(&cnt_out, 8, lane, a.err);                  // fragments of all 8 rows are in LDS
    if (ow >= 4) return;
    proj_stage2(ow, lane, wa, xfrag2, pred2);
    aux_barrier(&cnt_out, 12, lane, a.err);                 // the 4 weight waves
    if (ow == 0) {
      const int er = lane >> 3, ej = lane & 7;
      float v = e_bias;
#pragma unroll
      for (int k = 0; k < P_KS; ++k) v += pred2[k][ej][er];
      if (er < R) a.x_out[(int64_t)er * a.xf_ld + wgid * 8 + ej] = e_res + v;
      XPROBE(a, wgid, 6);                            // output projection stored
    }
    return;
  }
#endif

  // ================= KV waves: only in consumer workgroups =================
  if (!consumer) return;                             // (ended waves are not waited for by the barriers of the others)
  const int vpos = load_agent_int(a.d_pos);
  const int vlag = load_agent_int(a.lag ? a.lag + r : a.d_pos);
  const int vtk = load_agent_int(a.d_tick);          // for the tag of the output granules: requested now, read at the end
  const int cu = lane & 7, ks = lane >> 3;
  const int kk0 = wave * 8 + ks;
  const int Tk = uniform(vpos) + 1 - (a.lag ? uniform(vlag) : 0);     // keys incl. the new one at index Tk - 1
  const int nround = (Tk + KPR - 1) / KPR;
  half8v ku[NL], vu[NL];
  {
    // the cached K / V of the earlier positions, requested before anything else
    const half_t* kp = (const half_t*)a.kcache + (int64_t)r * a.cache_bs + h * 64 + cu * 8;
    const half_t* vp = (const half_t*)a.vcache + (int64_t)r * a.cache_bs + h * 64 + cu * 8;
    const uint32_t ld = (uint32_t)D;
    const uint32_t last_old = Tk >= 2 ? (uint32_t)(Tk - 2) * ld : 0u;  // slots at or past the new key re-read an old row
    const uint32_t o0 = (uint32_t)kk0 * ld;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (i < nround) {
        uint32_t o = o0 + (uint32_t)(i * KPR) * ld; if (o > last_old) o = last_old;
        ku[i] = __builtin_nontemporal_load((const half8v*)(kp + o));
      }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (i < nround) {
        uint32_t o = o0 + (uint32_t)(i * KPR) * ld; if (o > last_old) o = last_old;
        vu[i] = __builtin_nontemporal_load((const half8v*)(vp + o));
      }
    }
  }
  if (wave == 0) XPROBE(a, wgid, 4);                 // cached K/V requested
  __syncthreads();                                   // B1
  __syncthreads();                                   // B2
  __syncthreads();                                   // B3: q, new k, new v are in LDS
  if (wave == 0) XPROBE(a, wgid, 5);

  float sc[NL];
  float mx = WH_NEG_INF;
  {
    // the key this step appends sits at index Tk - 1: exactly one (round, key slot) of one wave takes it from the
    // granules instead of the register tile (a select per round; the tile itself stays untouched)
    const half8v qs = *(const half8v*)((const char*)qkv_sh[0] + cu * 16);
    const half8v knew = *(const half8v*)((const char*)qkv_sh[1] + cu * 16);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      sc[i] = WH_NEG_INF;
      if (i < nround) {
        const int kidx = kk0 + i * KPR;
        float d = qk_unit8(qs, kidx == Tk - 1 ? knew : ku[i]);
        d = group8_sum(d);
        sc[i] = (kidx < Tk) ? d : WH_NEG_INF;
        mx = fmaxf(mx, sc[i]);
      }
    }
    mx = across_groups8_max(mx);
    if (lane == 0) redm[wave] = mx;
  }
  __syncthreads();                                   // B4
  mx = redm[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, redm[w]);
  float acc[8];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const half8v vnew = *(const half8v*)((const char*)qkv_sh[2] + cu * 16);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (i < nround) {
      const float p = (sc[i] == WH_NEG_INF) ? 0.f : __expf(sc[i] - mx);
      sum += p;
      const half8v vv = (kk0 + i * KPR == Tk - 1) ? vnew : vu[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)vv[e], acc[e]);
    }
  }
  sum = across_groups8_sum(sum);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = across_groups8_sum(acc[e]);
  if (ks == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][cu * 8 + e] = acc[e];
  }
  if (lane == 0) reds[wave] = sum;
  __syncthreads();                                   // B5
  if (tid < 64) {
    float o = red[0][tid], l = reds[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red[w][tid]; l += reds[w]; }
    const half_t ov = (half_t)(o / l);
#ifdef WH_DEV
    if (a.x_out) {                                   // to the output-projection workgroups of this launch
      const uint32_t tag2 = ((uint32_t)(vtk + 1 + a.epoch) << 6) | (uint32_t)(a.layer + 1);
      publish_pair(a.og + (size_t)r * (D >> 1) + h * 32 + (tid >> 1), ov, lane, (tid & 1) == 0, true, tag2);
    } else
#endif
    {
      ((half_t*)a.out)[(int64_t)r * a.o_ld + h * 64 + tid] = ov;
    }
    XPROBE(a, wgid, 7);
  }
}

}  // namespace

namespace whk {

bool xattn_enabled() { return !WH_DEV_FLAG("WH_NO_FUSED_XATTN"); }      // developer switch; per task: WH_TASK_TWO_LAUNCH_CROSS

// how the granules are fetched: through the scalar memory path (default for the cross attention, whose consumer CUs are
// saturated with vector-memory requests: 13.4 vs 14.1 us per launch) or with vector loads (default for the self attention:
// 10.5 vs 10.9 us).  Developer switches: WH_XATTN_VECTOR_POLL=1, WH_SATTN_SCALAR_POLL=1.
int fused_mode(int kind) {      // kind 0: cross attention, 1: self attention; bit 0 = scalar-path polls
  const bool flip = kind == 0 ? WH_DEV_FLAG("WH_XATTN_VECTOR_POLL") : WH_DEV_FLAG("WH_SATTN_SCALAR_POLL");
  return kind == 0 ? (flip ? 0 : 1) : (flip ? 1 : 0);
}

// the shapes the fused form takes: fp16 (checked by the caller), <= 8 rows, one row per audio, K = D <= 1280 in blocks
// of 64, enough workgroups to host the D / 8 producers, a split that fits 16 rounds of 32 keys
bool xattn_supported(int D, int H, int R, int kv_group, int Tk, int splits) {
  if (!xattn_enabled()) return false;
  if (R < 1 || R > 8 || kv_group != 1 || D % 64 != 0 || D > 1280 || H * 64 != D) return false;
  if (splits < 1 || splits * H * R < D / 8) return false;
  const int chunk = (Tk + splits - 1) / splits;
  return (chunk + 63) / 64 <= 8;
}

// self attention + QKV projection: <= 8 rows, the cache fits the 512-key register tile, enough workgroups for the consumers
bool sattn_supported(int D, int H, int R, int n_ctx) {
  if (WH_DEV_FLAG("WH_NO_FUSED_SATTN")) return false;      // developer switch; per task: WH_TASK_TWO_LAUNCH_SELF
  if (R < 1 || R > 8 || D % 64 != 0 || D > 1280 || H * 64 != D || n_ctx > 448) return false;
  return H * R <= 3 * D / 8;
}

hipError_t launch_sattn8(const SAttnArgs& a, hipStream_t stream) {
  if ((int64_t)3 * a.D * a.D >= (1ll << 30) || (int64_t)a.D * 448 > 0x7fffffff) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sattn8_kernel, dim3(3 * a.D / 8), dim3(768), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_xattn8(const XAttnArgs& a, hipStream_t stream) {
  if (!xattn_supported(a.D, a.H, a.R, 1, a.Tk, a.splits)) return hipErrorNotSupported;
  if ((int64_t)a.k_ld * 2048 > 0x7fffffff || (int64_t)a.v_ld * 2048 > 0x7fffffff) return hipErrorInvalidValue;
  if ((int64_t)a.D * a.D >= (1ll << 30)) return hipErrorInvalidValue;          // 32-bit lane offsets into W
  const int chunk = (a.Tk + a.splits - 1) / a.splits;
  const int rounds = (chunk + 63) / 64;
  dim3 grid(a.splits, a.H, a.R), block(768);
  if (a.out_w) {                                       // phase 0: attn.out + residual inside this launch
#ifdef WH_DEV                                          // a REJECTED experiment (see out0_issue): development builds only
    if (!a.att_in || !a.out_b || !a.x_io || !a.pflags || a.x_io != a.xf) return hipErrorInvalidValue;
    if (rounds <= 4) hipLaunchKernelGGL((xattn8_kernel<4, true>), grid, block, 0, stream, a);
    else if (rounds <= 6) hipLaunchKernelGGL((xattn8_kernel<6, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((xattn8_kernel<8, true>), grid, block, 0, stream, a);
    return hipGetLastError();
#else
    return hipErrorNotSupported;
#endif
  }
  if (rounds <= 4) hipLaunchKernelGGL((xattn8_kernel<4, false>), grid, block, 0, stream, a);
  else if (rounds <= 6) hipLaunchKernelGGL((xattn8_kernel<6, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((xattn8_kernel<8, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace whk
