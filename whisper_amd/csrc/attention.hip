// attention.hip — MultiHeadAttention.qkv_attention (whisper/model.py:114-139), d_head = 64 always.
//
//   attn_flash_f16   encoder self-attention (1500 x 1500, non-causal) on MFMA: flash-style online
//                    softmax, S^T = K·Q^T and O^T = V^T·P^T with v_mfma_f32_32x32x16_f16 so that a
//                    lane owns one query column end-to-end (row max / rescale are lane-local, P feeds
//                    the second MFMA straight from registers).  K tile and V^T tile are staged through
//                    LDS (register staging, double buffered, XOR-swizzled 128-byte rows).
//   attn_generic     any Tq x Tk, optional causal mask (decoder prefill, fp32 strict-parity encoder)
//   attn_decode      the single-query step: one workgroup per (key split, head, row), 16 B/lane coalesced
//                    K/V streaming, wave-shuffle reductions; split-K partials are merged by the consumer
//                    GEMV's prologue.  This is where the cross-attention KV bytes (245.8 MB/row/step
//                    for large-v3) are read — HBM-bound.
//   cross_qk         raw scaled QK^T of chosen heads for word timestamps (whisper/timing.py:186-208)
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

constexpr float SCALE = 0.125f;                       // (64 ** -0.25) ** 2, model.py:118
constexpr float SCALE_LOG2E = 0.125f * 1.4426950408889634f;

// =============================================================================================
// generic attention (VALU)
// =============================================================================================
constexpr int GQ = 16;   // queries per workgroup (4 waves x 4)

template <typename T>
__global__ __launch_bounds__(256) void attn_generic_kernel(whk::AttnArgs a) {
  __shared__ float Ks[64][65];
  __shared__ float Vs[64][64];
  __shared__ float Qs[GQ][64];
  __shared__ float Ps[4][64];
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  constexpr int UPR = 64 / UNIT;   // units per head row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * GQ;
  const int Tq = a.Tq;
  const int Tk = a.d_len ? (*a.d_len + Tq) : a.Tk;
  const int kvb = b / a.kv_group;

  const T* qp = (const T*)a.q + (int64_t)b * a.q_bs + h * 64;
  const T* kp = (const T*)a.k + (int64_t)kvb * a.k_bs + h * 64;
  const T* vp = (const T*)a.v + (int64_t)kvb * a.v_bs + h * 64;

  for (int i = tid; i < GQ * 64; i += 256) {
    const int qi = i >> 6, d = i & 63;
    float v = 0.f;
    if (q0 + qi < Tq) v = to_f32(qp[(int64_t)(q0 + qi) * a.q_ld + d]);
    Qs[qi][d] = v;
  }

  float m[4], l[4], o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m[i] = WH_NEG_INF; l[i] = 0.f; o[i] = 0.f; }

  // keys needed by this block: causal blocks stop early
  int k_end = Tk;
  if (a.causal) {
    int last_q = q0 + GQ - 1; if (last_q > Tq - 1) last_q = Tq - 1;
    k_end = (Tk - Tq) + last_q + 1;
    if (k_end > Tk) k_end = Tk;
  }
  const int nkt = (k_end + 63) / 64;

  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    for (int u = tid; u < 64 * UPR; u += 256) {
      const int row = u / UPR, cu = u - row * UPR;
      const int key = kt * 64 + row;
      unit_t kv, vv;
      if (key < Tk) {
        kv = *(const unit_t*)(kp + (int64_t)key * a.k_ld + cu * UNIT);
        vv = *(const unit_t*)(vp + (int64_t)key * a.v_ld + cu * UNIT);
      } else {
#pragma unroll
        for (int e = 0; e < UNIT; ++e) { kv[e] = 0; vv[e] = 0; }
      }
#pragma unroll
      for (int e = 0; e < UNIT; ++e) {
        Ks[row][cu * UNIT + e] = to_f32(kv[e]);
        Vs[row][cu * UNIT + e] = to_f32(vv[e]);
      }
    }
    __syncthreads();

#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      const int qq = wave * 4 + qi;       // query within block
      const int qg = q0 + qq;
      if (qg >= Tq) continue;             // wave-uniform
      const int key = kt * 64 + lane;
      float s = 0.f;
#pragma unroll 16
      for (int d = 0; d < 64; ++d) s = __builtin_fmaf(Qs[qq][d], Ks[lane][d], s);
      s *= SCALE;
      bool valid = key < Tk;
      if (a.causal) valid = valid && (key <= (Tk - Tq) + qg);
      if (!valid) s = WH_NEG_INF;
      const float mx = wave_max(s);
      const float mn = fmaxf(m[qi], mx);
      float alpha, p;
      if (mn == WH_NEG_INF) { alpha = 1.f; p = 0.f; }
      else { alpha = __expf(m[qi] - mn); p = __expf(s - mn); }
      const float ps = wave_sum(p);
      l[qi] = l[qi] * alpha + ps;
      m[qi] = mn;
      Ps[wave][lane] = p;
      // wave-local LDS hand-off: the same wave writes and reads Ps[wave]; DS ops of one wave execute in order
      __builtin_amdgcn_wave_barrier();
      float acc = o[qi] * alpha;
#pragma unroll 16
      for (int j = 0; j < 64; ++j) acc = __builtin_fmaf(Ps[wave][j], Vs[j][lane], acc);
      o[qi] = acc;
    }
  }

  T* op = (T*)a.out + (int64_t)b * a.o_bs + h * 64;
#pragma unroll
  for (int qi = 0; qi < 4; ++qi) {
    const int qg = q0 + wave * 4 + qi;
    if (qg < Tq) op[(int64_t)qg * a.o_ld + lane] = from_f32<T>(o[qi] / l[qi]);
  }
}

// =============================================================================================
// encoder flash attention, fp16 MFMA
// =============================================================================================
#ifndef WH_FLASH_WAVES
#define WH_FLASH_WAVES 4
#endif
constexpr int FW = WH_FLASH_WAVES;   // waves per workgroup
constexpr int FQ = FW * 32;          // queries per workgroup (32 per wave)

#ifndef WH_FLASH_WAVES_PER_SIMD
#define WH_FLASH_WAVES_PER_SIMD 2
#endif
// PRESCALED: q and k arrive multiplied by sqrt(0.125 * log2 e) each (WH_WEIGHTS_ENC_QK_SCALED: the factor sits in the
// encoder's query / key projection weights), so K·Q^T is already the exp2 argument.  The score accumulators then
// start at -m_ref instead of 0 and p = exp2(acc) needs no multiply-subtract, and the row sums are taken from the fp16
// P two at a time (v_dot2_f32_f16).  Per 32 x 64 score tile that leaves 32 v_exp + 16 v_cvt_pk + 16 v_dot2 + 16 v_max3
// on the vector ALU (the bound of this loop) where the unscaled form has 32 v_fma + 32 v_add instead of the v_dot2.
// VTP: the V^T tile is stored in LDS with the 4-key halves of neighbouring 16-byte units exchanged — unit 2m holds keys
// {0-3, 8-11} of its 16-key group, unit 2m+1 keys {4-7, 12-15} — which is the order the score MFMA leaves P in, so a
// lane's V^T fragment is ONE ds_read_b128 (4 LDS cycles, conflict-free under the row swizzle) instead of a
// ds_read2_b64 (8 cycles; profiles/r02_pmc_lds.csv: a third of this kernel's LDS cycles were counted as conflicts).
// The tile store becomes two ds_write_b64 per unit.  Pure data movement: the output is bit-identical to the plain
// layout (tools/probe_gemm compares all 15.4 M outputs of the large-v3 shape), 143.9 -> 134.3 us per launch.
// VTP = false keeps the plain layout for that comparison (launch mode bit 1).
template <bool PRESCALED, bool VTP = true>
__global__ __launch_bounds__(FW * 64, WH_FLASH_WAVES_PER_SIMD) void attn_flash_f16_kernel(
    const half_t* __restrict__ q, int64_t q_ld, int64_t q_bs, const half_t* __restrict__ k, int64_t k_ld,
    int64_t k_bs, const half_t* __restrict__ vt, int64_t vt_ld, int64_t vt_bs, half_t* __restrict__ out,
    int64_t o_ld, int64_t o_bs, int T, int Tq) {      // T keys, Tq queries (encoder: Tq == T; decoder prefill: T0 x 1500)
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];   // [buf][K | Vt][64 rows x 128 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qrow = blockIdx.x * FQ + wave * 32 + (lane & 31);
  const int hi = lane >> 5;

  const half_t* qp = q + (int64_t)b * q_bs + h * 64;
  const half_t* kp = k + (int64_t)b * k_bs + h * 64;
  const half_t* vp = vt + (int64_t)b * vt_bs + (int64_t)h * 64 * vt_ld;

  // Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds Q[q][16 s + 8 hi .. +7], s = 0..3
  half8v qf[4];
  {
    const int qr = qrow < Tq ? qrow : Tq - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const half8v*)(qp + (int64_t)qr * q_ld + 16 * s + 8 * hi);
  }

  float16v oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
  float m_run = PRESCALED ? 0.f : WH_NEG_INF, l_run = 0.f;

  const int nkt = (T + 63) / 64;
  // staging: thread handles units u = tid, tid + threads, ... of each 512-unit tile.  Addresses are a uniform base
  // that advances with the tile (scalar ALU) plus a per-thread 32-bit offset that never changes: no vector
  // arithmetic per tile except on the last, ragged one, whose key rows are clamped.
  constexpr int SJ = 512 / (FW * 64);
  uint4v kreg[SJ], vreg[SJ];
  uint32_t koff[SJ], voff[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int u = tid + FW * 64 * j;
    const int row = u >> 3, cu = u & 7;
    koff[j] = (uint32_t)((row * k_ld + cu * 8) * 2);
    voff[j] = (uint32_t)((row * vt_ld + cu * 8) * 2);
  }
  auto gload = [&](int kt) {
    const char* kb = (const char*)kp + (int64_t)kt * 64 * k_ld * 2;
    const char* vb = (const char*)vp + (int64_t)kt * 128;
    if (kt * 64 + 64 <= T) {
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        kreg[j] = *(const uint4v*)(kb + koff[j]);
        vreg[j] = *(const uint4v*)(vb + voff[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int u = tid + FW * 64 * j;
        const int row = u >> 3, cu = u & 7;
        int key = kt * 64 + row; if (key > T - 1) key = T - 1;
        kreg[j] = *(const uint4v*)(kp + (int64_t)key * k_ld + cu * 8);
        vreg[j] = *(const uint4v*)(vb + voff[j]);
      }
    }
  };
  auto lstore = [&](int buf) {
    char* sK = smem + buf * 16384;
    char* sV = sK + 8192;
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const int u = tid + FW * 64 * j;
      const int row = u >> 3, cu = u & 7;
      *(uint4v*)(sK + swz_byte(row, cu)) = kreg[j];
      if constexpr (VTP) {
        *(uint2v*)(sV + swz_byte(row, cu & ~1) + (cu & 1) * 8) = uint2v{vreg[j][0], vreg[j][1]};
        *(uint2v*)(sV + swz_byte(row, cu | 1) + (cu & 1) * 8) = uint2v{vreg[j][2], vreg[j][3]};
      } else {
        *(uint4v*)(sV + swz_byte(row, cu)) = vreg[j];
      }
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const char* sK = smem + cur * 16384;
    const char* sV = sK + 8192;

    // ---- S^T tiles: keys kb*32.. x 32 queries
    float16v sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float s0 = PRESCALED ? -m_run : 0.f;   // (a separate C operand holding -m_run measured slower than these v_mov)
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[kb][i] = s0;
      const int krow = kb * 32 + (lane & 31);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // (requesting all eight K fragments / all eight V^T fragments up front costs 17 VGPRs, i.e. the third wave per
        // SIMD, and measured 10 % slower than this read-wait-multiply order)
        const half8v kf = *(const half8v*)(sK + swz_byte(krow, 2 * s + hi));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
      }
    }
    // ---- mask tail keys (only the last tile can be ragged)
    if (kt == nkt - 1 && (T & 63) != 0) {
      asm volatile("" ::: "memory");                 // keeps this a branch (if-converted it is 31 v_cndmask on EVERY tile)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= T) sacc[kb][r] = WH_NEG_INF;
        }
    }
    // ---- online softmax (lane owns query lane&31; partner lane^32 holds the other 16 keys per block).
    // The loop is bound by the vector ALU, not by the matrix cores, so the non-exp half is trimmed: 3-input maxima,
    // and the accumulator rescale (32 multiplies + the bookkeeping around it) is DEFERRED — the running reference
    // maximum only moves when some query of the wave has outgrown it by more than 2^DEFER (exp2 domain); until then
    // p = exp2(s - m_ref) <= 2^DEFER stays well inside fp16 and fp32 ranges and the result is mathematically
    // unchanged (softmax is invariant to the reference point).
    constexpr float DEFER = 6.0f;
    float mx = __builtin_fmaxf(__builtin_fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
#pragma unroll
    for (int r = 3; r + 1 < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sacc[0][r]), sacc[0][r + 1]);   // v_max3_f32
    mx = fmaxf(mx, sacc[0][15]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sacc[1][r]), sacc[1][r + 1]);
    { float pa, pb; lane_swap32(mx, pa, pb); mx = fmaxf(pa, pb); }   // partner half (lane ^ 32) without LDS
    half8v pf[2][2];
    if constexpr (PRESCALED) {
      // mx is already relative to the reference (the accumulators started at -m_run); key 0 of every tile is valid,
      // so it is finite.  The first tile always sets the reference.
      if (kt == 0 || __any(mx > DEFER)) {            // wave-uniform: rare after the first tiles
        const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; sacc[0][i] -= delta; sacc[1][i] -= delta; }
        l_run *= alpha;
        m_run += delta;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[kb][r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(sacc[kb][r]);   // v_exp_f32
    } else {
      // m_run is kept in the exp2 domain (already multiplied by SCALE_LOG2E); -inf before the first tile
      const float mxs = mx * SCALE_LOG2E;
      if (__any(mxs > m_run + DEFER)) {              // wave-uniform: rare after the first tiles
        const float m_new = fmaxf(m_run, mxs);       // finite: key 0 of every tile is valid
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 0 on the first tile (m_run = -inf)
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        m_run = m_new;
      }
      const float mc = m_run;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][r], SCALE_LOG2E, -mc));   // v_exp_f32
          psum += p;
          pf[kb][r >> 3][r & 7] = (half_t)p;
        }
      l_run += psum;
    }

    // ---- O^T += V^T · P^T
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int drow = dt * 32 + (lane & 31);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int unit = kb * 4 + 2 * s2;
          half8v vf;
          if constexpr (VTP) {
            vf = *(const half8v*)(sV + swz_byte(drow, unit + hi));
          } else {
            const half4v v0 = *(const half4v*)(sV + swz_byte(drow, unit) + hi * 8);
            const half4v v1 = *(const half4v*)(sV + swz_byte(drow, unit + 1) + hi * 8);
            vf = half8v{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s2], oacc[dt], 0, 0, 0);
        }
    }
    if constexpr (PRESCALED) {
      // row sums of the fp16 P by v_dot2_f32_f16, two scores per instruction (a third "V^T" tile of ones on the
      // matrix cores does the same for 4 MFMAs and 16 more VGPRs: measured 1-3 % slower)
      const half_t one = (half_t)1.0f;
      const half2v one2 = {one, one};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            l_run = __builtin_amdgcn_fdot2(half2v{pf[kb][s2][e], pf[kb][s2][e + 1]}, one2, l_run, false);
    }

    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }

  float l_tot;
  {
    float l_a, l_b; lane_swap32(l_run, l_a, l_b);
    l_tot = l_a + l_b;
  }
  const float inv = 1.0f / l_tot;
  if (qrow < Tq) {
    half_t* op = out + (int64_t)b * o_bs + (int64_t)qrow * o_ld + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4v o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (half_t)(oacc[dt][g * 4 + e] * inv);
        *(half4v*)(op + dt * 32 + 8 * g + 4 * hi) = o4;
      }
  }
}

// =============================================================================================
// decode attention: one query per row.  The whole K and V slice of a workgroup is requested up front
// (2 x NL 16-byte loads per lane, non-temporal) and held in registers: one HBM round trip per launch,
// scores never touch LDS, softmax statistics travel by DPP / permlane + two small LDS exchanges.
// A wave issues one VALU instruction per ~8.6 cycles, so the per-wave instruction chain is kept short:
// WAVES = 8 waves per (split, head, row), v_dot2_f32_f16 for q.k, 32-bit offsets with a clamped stride walk.
// =============================================================================================

__device__ __forceinline__ float qk_unit(half8v q, half8v k) {
  float d = __builtin_amdgcn_fdot2(half2v{q[0], q[1]}, half2v{k[0], k[1]}, 0.f, false);
  d = __builtin_amdgcn_fdot2(half2v{q[2], q[3]}, half2v{k[2], k[3]}, d, false);
  d = __builtin_amdgcn_fdot2(half2v{q[4], q[5]}, half2v{k[4], k[5]}, d, false);
  d = __builtin_amdgcn_fdot2(half2v{q[6], q[7]}, half2v{k[6], k[7]}, d, false);
  return d;
}
__device__ __forceinline__ float qk_unit(float4v q, float4v k) {
  float d = q[0] * k[0];
  d = __builtin_fmaf(q[1], k[1], d); d = __builtin_fmaf(q[2], k[2], d); d = __builtin_fmaf(q[3], k[3], d);
  return d;
}
// q * 0.125 is exact in fp16 (power of two) unless the product is subnormal — irrelevant at |q| ~ 1
__device__ __forceinline__ half8v scale_q(half8v q) {
  half8v r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (half_t)((float)q[e] * SCALE);
  return r;
}
__device__ __forceinline__ float4v scale_q(float4v q) { return q * SCALE; }

// SKIP: rounds beyond the live key count are skipped by wave-uniform branches (self attention, where the
// cached length is only known on the device); without it every round is issued unconditionally.
template <typename T, int NL, int WAVES, bool SKIP>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_kernel(whk::DecAttnArgs a) {
  pin_kernargs(a);
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  constexpr int LPK = 64 / UNIT;     // lanes per key (8 fp16 / 16 fp32)
  constexpr int KPW = 64 / LPK;      // keys per wave instruction
  constexpr int KPR = WAVES * KPW;   // keys per round of the workgroup
  __shared__ float red[WAVES][64];
  __shared__ float redm[WAVES], reds[WAVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, h = blockIdx.y, r = blockIdx.z;
  const int S = a.splits;
  const int kvb0 = r / a.kv_group;
  const int cu0 = lane % LPK, ks0 = lane / LPK;
  const int64_t hs0 = a.kv_hs ? a.kv_hs : 64;
  // Self attention (SKIP): the cached length lives on the device, and everything below — split bounds, the clamp of the
  // load offsets — waits for that L2 round trip (~1 us of a 5.5 us launch).  The first EAG rounds of KEYS do not need it:
  // with one split their addresses are fixed (keys 0 .. 2 KPR - 1 of the row; the cache row exists for all n_ctx
  // positions), so they are requested right behind the position loads, unclamped.  Slots at or beyond the cached length
  // may then hold anything (stale positions of an earlier clip) — their score is replaced by -inf below, and only keys are
  // fetched this way (a masked VALUE would still enter the sum as 0 x NaN).
  constexpr int EAG = SKIP ? 2 : 0;
  // (only when the cache row really holds EAG * KPR positions: k_bs / k_ld = n_text_ctx of the model — tiny test dims may not)
  const bool eager = SKIP && S == 1 && a.d_len != nullptr && (int64_t)(EAG * KPR) * a.k_ld <= a.k_bs;
  int Tk = a.Tk;
  int vn = 0, vl = 0;
  if (a.d_len) {                          // cached length and this row's lag: two independent agent-scope loads
    // (measured alternative: `s_load_dword ... glc` past the scalar cache — 1280 waves polling one word that way take
    // 38 us per launch instead of 5.6; the plain s_load is stale between launches, see load_agent_int)
    vn = load_agent_int(a.d_len);
    vl = load_agent_int(a.lag ? a.lag + r : a.d_len);
  }
  // q first (L2 hit, needed first): loads return in issue order
  const unit_t qraw = *(const unit_t*)((const T*)a.q + (int64_t)r * a.q_ld + h * 64 + cu0 * UNIT);
  asm volatile("" ::: "memory");
  typename ET<T>::unit_t ku_e[EAG > 0 ? EAG : 1];
  if (eager) {
    const T* kpe = (const T*)a.k + (int64_t)kvb0 * a.k_bs + h * hs0 + cu0 * UNIT;
#pragma unroll
    for (int i = 0; i < EAG; ++i)
      ku_e[i] = __builtin_nontemporal_load((const unit_t*)(kpe + (uint32_t)((i * WAVES + wave) * KPW + ks0) * (uint32_t)a.k_ld));
  }
  if (a.d_len) Tk = uniform(vn) + a.len_plus - (a.lag ? uniform(vl) : 0);
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + KPR - 1) / KPR * KPR;
  const int k0 = s * chunk;
  int k1 = k0 + chunk; if (k1 > Tk) k1 = Tk;
  const int nkeys = k1 > k0 ? k1 - k0 : 0;
  const int nround = (nkeys + KPR - 1) / KPR;          // uniform: rounds that hold at least one key

  const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  (void)wgid;
  WH_PROBE_AT(a, wgid, 0);
  const int kvb = r / a.kv_group;
  const int cu = lane % LPK;          // unit within the head row
  const int ks = lane / LPK;          // key slot within the wave instruction
  const int64_t hs = a.kv_hs ? a.kv_hs : 64;
  const T* kp = (const T*)a.k + (int64_t)kvb * a.k_bs + h * hs + cu * UNIT;   // wave-uniform part folded by the compiler
  const T* vp = (const T*)a.v + (int64_t)kvb * a.v_bs + h * hs + cu * UNIT;

  // key of round i for this lane: kk_i = (i * WAVES + wave) * KPW + ks; element offset = (k0 + kk_i) * ld,
  // walked with a constant stride and clamped to the last valid key of the split (branch-free loads: a
  // zero-fill else-arm would make the compiler drain vmcnt at every join).  Rounds >= nround are skipped by a
  // wave-uniform branch; slots past the split inside a live round are neutralised by score = -inf.
  const int kk0 = wave * KPW + ks;
  const int klast = nkeys > 0 ? nkeys - 1 : 0;
  const uint32_t ldk = (uint32_t)a.k_ld, ldv = (uint32_t)a.v_ld;
  const uint32_t ok0 = (uint32_t)(k0 + kk0) * ldk, okl = (uint32_t)(k0 + klast) * ldk;
  const uint32_t ov0 = (uint32_t)(k0 + kk0) * ldv, ovl = (uint32_t)(k0 + klast) * ldv;
  unit_t ku[NL], vu[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (i < EAG && eager) {
      ku[i] = ku_e[i];
    } else if (!SKIP || i < nround) {
      uint32_t o = ok0 + (uint32_t)(i * KPR) * ldk; if (o > okl) o = okl;
      ku[i] = __builtin_nontemporal_load((const unit_t*)(kp + o));
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (!SKIP || i < nround) {
      uint32_t o = ov0 + (uint32_t)(i * KPR) * ldv; if (o > ovl) o = ovl;
      vu[i] = __builtin_nontemporal_load((const unit_t*)(vp + o));
    }
  }
  WH_PROBE_AT(a, wgid, 1);

  const unit_t qs = scale_q(qraw);

  // ---- scores (registers), split max
  float sc[NL];
  float mx = WH_NEG_INF;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    sc[i] = WH_NEG_INF;
    if (!SKIP || i < nround) {
      float d = qk_unit(qs, ku[i]);
      d = LPK == 8 ? group8_sum(d) : group16_sum(d);
      sc[i] = (kk0 + i * KPR < nkeys) ? d : WH_NEG_INF;
      mx = fmaxf(mx, sc[i]);
    }
  }
  mx = LPK == 8 ? across_groups8_max(mx) : across_groups16_max(mx);
  if (lane == 0) redm[wave] = mx;
  __syncthreads();
  mx = redm[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) mx = fmaxf(mx, redm[w]);
  WH_PROBE_AT(a, wgid, 2);

  // ---- p = exp(s - max), o = sum_k p[k] V[k]
  float acc[UNIT];
#pragma unroll
  for (int e = 0; e < UNIT; ++e) acc[e] = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    if (!SKIP || i < nround) {
      const float p = (sc[i] == WH_NEG_INF) ? 0.f : __expf(sc[i] - mx);
      sum += p;
#pragma unroll
      for (int e = 0; e < UNIT; ++e) acc[e] = __builtin_fmaf(p, to_f32(vu[i][e]), acc[e]);
    }
  }
  // reduce over key slots (lane bits above log2(LPK)); every lane of a key group holds the same p
  sum = LPK == 8 ? across_groups8_sum(sum) : across_groups16_sum(sum);
#pragma unroll
  for (int e = 0; e < UNIT; ++e) acc[e] = LPK == 8 ? across_groups8_sum(acc[e]) : across_groups16_sum(acc[e]);
  if (ks == 0) {
#pragma unroll
    for (int e = 0; e < UNIT; ++e) red[wave][cu * UNIT + e] = acc[e];
  }
  if (lane == 0) reds[wave] = sum;
  __syncthreads();
  WH_PROBE_AT(a, wgid, 3);
  if constexpr (sizeof(T) == 2) {
    if (S > 1 && a.merge_cnt) {
      // ---- the LAST of the S workgroups of (row, head) to get here merges the S partials itself — no merge launch (4.8 us
      // per layer at 24 rows) and no PRO_COMBINE prologue in the projection that follows.  No workgroup waits for another:
      // each publishes its partial with write-through (agent-scope) stores, drains them, and takes a ticket; the one that
      // draws S - 1 knows all S partials are in memory, reads them with agent-scope loads and does merge_partials_kernel's
      // arithmetic on the same fp16 partials (16 lanes x 4 head dims, same order of operations: bit-identical output).
      // MI355X_MICROARCH.md "handoff": sc1 payload -> vmcnt(0) -> ticket; sc1 loads after the ticket.  The ticket counter is
      // back at 0 when the launch ends.
      typedef unsigned long long u64;
      if (wave == 0) {
        const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
        if (lane < 16) {
          float l = reds[0];
#pragma unroll
          for (int w = 1; w < WAVES; ++w) l += reds[w];
          half4v hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o = red[0][4 * lane + e];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) o += red[w][4 * lane + e];
            hv[e] = (half_t)(nkeys > 0 ? o / l : 0.f);
          }
          __hip_atomic_store((u64*)((half_t*)a.part_o + pi * 64 + 4 * lane), __builtin_bit_cast(u64, hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lane == 0) {
            const float2v ml = {nkeys > 0 ? mx : WH_NEG_INF, nkeys > 0 ? l : 0.f};
            __hip_atomic_store((u64*)(a.part_ml + pi * 2), __builtin_bit_cast(u64, ml), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the partial is in memory, THEN the ticket
        int* cnt = a.merge_cnt + r * a.H + h;
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket == S - 1) {
          if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lane < 16) {
            constexpr int MS = 4;                                     // callers pass merge_cnt only with splits <= 4
            float2v ml[MS];
            float4v o[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) {
              const int sc = j < S ? j : S - 1;                       // branch-free loads, clamped (as merge_partials_kernel)
              const int64_t pb = ((int64_t)sc * a.R + r) * a.H + h;
              ml[j] = __builtin_bit_cast(float2v, __hip_atomic_load((u64*)(a.part_ml + pb * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
              const half4v hp = __builtin_bit_cast(half4v, __hip_atomic_load((u64*)((half_t*)a.part_o + pb * 64 + 4 * lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
              o[j] = float4v{(float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3]};
            }
            float M = WH_NEG_INF;
#pragma unroll
            for (int j = 0; j < MS; ++j) { if (j >= S) ml[j] = float2v{WH_NEG_INF, 0.f}; M = fmaxf(M, ml[j][0]); }
            float wgt[MS], den = 0.f;
#pragma unroll
            for (int j = 0; j < MS; ++j) {
              wgt[j] = (ml[j][0] == WH_NEG_INF) ? 0.f : __expf(ml[j][0] - M);
              den = __builtin_fmaf(wgt[j], ml[j][1], den);
            }
            const float inv = 1.0f / den;
            float4v num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MS; ++j) {
              const float f = wgt[j] * ml[j][1] * inv;
              num[0] = __builtin_fmaf(f, o[j][0], num[0]); num[1] = __builtin_fmaf(f, o[j][1], num[1]);
              num[2] = __builtin_fmaf(f, o[j][2], num[2]); num[3] = __builtin_fmaf(f, o[j][3], num[3]);
            }
            const half4v res = {(half_t)num[0], (half_t)num[1], (half_t)num[2], (half_t)num[3]};
            *(half4v*)((half_t*)a.out + (a.o_frag ? frag_elem(r, h * 64 + 4 * lane, a.H * 64) : (int64_t)r * a.o_ld + h * 64 + 4 * lane)) = res;
          }
        }
      }
      WH_PROBE_AT(a, wgid, 4);
      return;
    }
  }
  if (tid < 64) {
    float o = red[0][tid], l = reds[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red[w][tid]; l += reds[w]; }
    if (S == 1) {
      ((T*)a.out)[a.o_frag ? frag_elem(r, h * 64 + tid, a.H * 64) : (int64_t)r * a.o_ld + h * 64 + tid] = from_f32<T>(o / l);
    } else {
      // split-major partials [split][row][head]: the consumer projection reads, per split, one contiguous
      // [row][head][64] block with the same lane map; o is stored NORMALISED (o / l) in the element type
      const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
      ((T*)a.part_o)[pi * 64 + tid] = from_f32<T>(nkeys > 0 ? o / l : 0.f);
      if (tid == 0) {
        a.part_ml[pi * 2 + 0] = nkeys > 0 ? mx : WH_NEG_INF;
        a.part_ml[pi * 2 + 1] = nkeys > 0 ? l : 0.f;
      }
    }
  }
  WH_PROBE_AT(a, wgid, 4);
}

// =============================================================================================
// decode cross attention for beam groups: the G rows of one audio (beams) attend to the SAME K/V, so one
// workgroup per (split, head, audio) loads the tile once and scores all G queries against it — the per-row
// kernel re-read each audio's K/V once per beam (5x the bytes at beam 5).
// =============================================================================================
template <typename T, int NL, int WAVES, int GQ>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_group_kernel(whk::DecAttnArgs a) {
  pin_kernargs(a);
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  constexpr int LPK = 64 / UNIT, KPW = 64 / LPK, KPR = WAVES * KPW;
  __shared__ float red[WAVES][GQ][64];
  __shared__ float redm[WAVES][GQ], reds[WAVES][GQ];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int S = a.splits, G = a.kv_group, Tk = a.Tk;
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + KPR - 1) / KPR * KPR;
  const int k0 = s * chunk;
  int k1 = k0 + chunk; if (k1 > Tk) k1 = Tk;
  const int nkeys = k1 > k0 ? k1 - k0 : 0;
  const int cu = lane % LPK, ks = lane / LPK;
  const int64_t hs = a.kv_hs ? a.kv_hs : 64;
  const T* kp = (const T*)a.k + (int64_t)b * a.k_bs + h * hs + cu * UNIT;
  const T* vp = (const T*)a.v + (int64_t)b * a.v_bs + h * hs + cu * UNIT;

  unit_t qraw[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    const int r = b * G + (g < G ? g : G - 1);
    qraw[g] = *(const unit_t*)((const T*)a.q + (int64_t)r * a.q_ld + h * 64 + cu * UNIT);
  }
  asm volatile("" ::: "memory");
  const int kk0 = wave * KPW + ks;
  const int klast = nkeys > 0 ? nkeys - 1 : 0;
  const uint32_t ldk = (uint32_t)a.k_ld, ldv = (uint32_t)a.v_ld;
  const uint32_t ok0 = (uint32_t)(k0 + kk0) * ldk, okl = (uint32_t)(k0 + klast) * ldk;
  const uint32_t ov0 = (uint32_t)(k0 + kk0) * ldv, ovl = (uint32_t)(k0 + klast) * ldv;
  unit_t ku[NL], vu[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t o = ok0 + (uint32_t)(i * KPR) * ldk; if (o > okl) o = okl;
    ku[i] = __builtin_nontemporal_load((const unit_t*)(kp + o));
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t o = ov0 + (uint32_t)(i * KPR) * ldv; if (o > ovl) o = ovl;
    vu[i] = __builtin_nontemporal_load((const unit_t*)(vp + o));
  }

  float sc[GQ][NL], mx[GQ];
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    mx[g] = WH_NEG_INF;
    if (g >= G) continue;                  // wave-uniform: beam slots beyond the group size cost nothing
    const unit_t qs = scale_q(qraw[g]);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float d = qk_unit(qs, ku[i]);
      d = LPK == 8 ? group8_sum(d) : group16_sum(d);
      sc[g][i] = (kk0 + i * KPR < nkeys) ? d : WH_NEG_INF;
      mx[g] = fmaxf(mx[g], sc[g][i]);
    }
    mx[g] = LPK == 8 ? across_groups8_max(mx[g]) : across_groups16_max(mx[g]);
    if (lane == 0) redm[wave][g] = mx[g];
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    float m = redm[0][g];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) m = fmaxf(m, redm[w][g]);
    mx[g] = m;
  }

#pragma unroll
  for (int g = 0; g < GQ; ++g) {
    if (g >= G) continue;
    float acc[UNIT];
#pragma unroll
    for (int e = 0; e < UNIT; ++e) acc[e] = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float p = (sc[g][i] == WH_NEG_INF) ? 0.f : __expf(sc[g][i] - mx[g]);
      sum += p;
#pragma unroll
      for (int e = 0; e < UNIT; ++e) acc[e] = __builtin_fmaf(p, to_f32(vu[i][e]), acc[e]);
    }
    sum = LPK == 8 ? across_groups8_sum(sum) : across_groups16_sum(sum);
#pragma unroll
    for (int e = 0; e < UNIT; ++e) acc[e] = LPK == 8 ? across_groups8_sum(acc[e]) : across_groups16_sum(acc[e]);
    if (ks == 0) {
#pragma unroll
      for (int e = 0; e < UNIT; ++e) red[wave][g][cu * UNIT + e] = acc[e];
    }
    if (lane == 0) reds[wave][g] = sum;
  }
  __syncthreads();
  for (int t = tid; t < G * 64; t += WAVES * 64) {
    const int g = t >> 6, d = t & 63;
    float o = red[0][g][d], l = reds[0][g];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red[w][g][d]; l += reds[w][g]; }
    const int r = b * G + g;
    if (S == 1) {
      ((T*)a.out)[a.o_frag ? frag_elem(r, h * 64 + d, a.H * 64) : (int64_t)r * a.o_ld + h * 64 + d] = from_f32<T>(o / l);
    } else {
      const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
      ((T*)a.part_o)[pi * 64 + d] = from_f32<T>(nkeys > 0 ? o / l : 0.f);
      if (d == 0) {
        a.part_ml[pi * 2 + 0] = nkeys > 0 ? mx[g] : WH_NEG_INF;
        a.part_ml[pi * 2 + 1] = nkeys > 0 ? l : 0.f;
      }
    }
  }
}

// =============================================================================================
// The same on the matrix cores (fp16).  With G beams against one K/V tile the vector-ALU form above does G dot
// products, G softmax passes and G p.v accumulations per key — 24 us per launch at 40 rows against 11.7 us for the
// 8-row greedy kernel that streams the same bytes.  Here the beams are the N dimension of v_mfma_f32_16x16x32_f16
// (G <= 8 of 16 columns used):
//   S[key][beam]   = K[key][:] . q[beam][:]     A = K rows as loaded (16 B per lane = 8 head dims of one key),
//                                               B = q (8 head dims of one beam)
//   O^T[d][beam]   = V^T[d][keys] . P^T[keys][beam]   A = V^T rows (16 B per lane = 8 KEYS of one head dim),
//                                               B = P straight from the S accumulators: no LDS, no permute
// The second product needs 8 consecutive keys per lane, the first hands a lane 4 consecutive ROWS of each 16-row
// tile, so the two tiles of a 32-key block take the keys {0-3, 8-11, 16-19, 24-27} and {4-7, 12-15, 20-23, 28-31}:
// lane (beam, g) then owns keys 8g..8g+7 of the block, which is exactly the B fragment of the second product (a
// key row is its own 128-byte line either way, so the permutation costs nothing in the loads).  V^T comes from a
// GEMM with swapped operands at set_audio (as in the encoder).  Everything a workgroup needs is requested up front:
// one HBM round trip, as in the kernels above.  4 waves x NBLK blocks of 32 keys per (split, head, audio).
// =============================================================================================
template <int NBLK, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_group_mfma_kernel(whk::DecAttnArgs a) {
  pin_kernargs(a);
  __shared__ float red_o[WAVES][16][64];
  __shared__ float red_m[WAVES][16], red_l[WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int S = a.splits, G = a.kv_group, Tk = a.Tk;
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + 127) / 128 * 128;                                  // (<= NBLK * WAVES * 32 keys: launcher)
  const int k0 = s * chunk;
  int k1 = k0 + chunk; if (k1 > Tk) k1 = Tk;
  const int nkeys = k1 > k0 ? k1 - k0 : 0;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int64_t hs = a.kv_hs ? a.kv_hs : 64;

  // q: B operand of the first product, lane (beam, g) holds head dims 32 ks + 8 g .. + 7, pre-scaled (exact in fp16)
  half8v qf[2];
  {
    const half_t* qp = (const half_t*)a.q + (int64_t)(b * G + (i16 < G ? i16 : G - 1)) * a.q_ld + h * 64 + g4 * 8;
    qf[0] = scale_q(*(const half8v*)qp);
    qf[1] = scale_q(*(const half8v*)(qp + 32));
  }
  asm volatile("" ::: "memory");
  const half_t* kp = (const half_t*)a.k + (int64_t)b * a.k_bs + h * hs + g4 * 8;
  const half_t* vp = (const half_t*)a.vt + (int64_t)b * a.vt_bs + (int64_t)(h * 64 + i16) * a.vt_ld + g4 * 8;
  const int klast = nkeys > 0 ? k1 - 1 : 0;
  const uint32_t ldk = (uint32_t)a.k_ld;
  half8v kf[NBLK][2][2], vf[NBLK][4];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    const int kb = k0 + (wave * NBLK + j) * 32;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int key = kb + 8 * (i16 >> 2) + 4 * t + (i16 & 3);
      if (key > klast) key = klast;                                   // masked below; the load stays in bounds
      const half_t* kr = kp + (uint32_t)key * ldk;
      kf[j][t][0] = __builtin_nontemporal_load((const half8v*)kr);
      kf[j][t][1] = __builtin_nontemporal_load((const half8v*)(kr + 32));
    }
  }
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    // the rows are padded to S * chunk keys; a block that lies wholly beyond this split's keys (chunk < WAVES * NBLK * 32)
    // has P = 0 everywhere and is pointed at the last 8 columns of the row, so that what it multiplies by 0 is finite
    // and inside the row
    int col = k0 + (wave * NBLK + j) * 32 + g4 * 8;
    if (col > (int)a.vt_ld - 8) col = (int)a.vt_ld - 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vf[j][dt] = __builtin_nontemporal_load((const half8v*)(vp - g4 * 8 + (int64_t)(dt * 16) * a.vt_ld + col));
  }
  // every request goes out before the first use: left alone, the scheduler hoists the first MFMAs (and the waits for
  // their operands) in between the loads to save registers — several dependent round trips instead of one
  __builtin_amdgcn_sched_barrier(0);

  // ---- scores: lane (beam, g) ends with keys kb + 8 g + {0..3} (tile 0) and + {4..7} (tile 1)
  float4v sc[NBLK][2];
  float mx = WH_NEG_INF;
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    const int kb = k0 + (wave * NBLK + j) * 32;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float4v c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[j][t][0], qf[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[j][t][1], qf[1], c, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kb + 8 * g4 + 4 * t + e >= k1) c[e] = WH_NEG_INF;
        mx = fmaxf(mx, c[e]);
      }
      sc[j][t] = c;
    }
  }
  mx = across_groups16_max(mx);                                       // over g: this wave's keys, per beam
  if (g4 == 0) red_m[wave][i16] = mx;
  __syncthreads();
  float M = red_m[0][i16];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) M = fmaxf(M, red_m[w][i16]);

  // ---- p = exp(s - M), row sums, and the second product with P taken from the accumulators as they are
  float4v oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = float4v{0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    half8v pf;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = (sc[j][t][e] == WH_NEG_INF) ? 0.f : __expf(sc[j][t][e] - M);
        lsum += p;
        pf[4 * t + e] = (half_t)p;
      }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[j][dt], pf, oacc[dt], 0, 0, 0);
  }
  lsum = across_groups16_sum(lsum);
  if (g4 == 0) red_l[wave][i16] = lsum;
  // O^T[d][beam]: lane (beam, g) holds d = 16 dt + 4 g + e
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) *(float4v*)&red_o[wave][i16][dt * 16 + g4 * 4] = oacc[dt];
  __syncthreads();
  for (int t = tid; t < G * 64; t += WAVES * 64) {
    const int g = t >> 6, d = t & 63;
    float o = red_o[0][g][d], l = red_l[0][g], m = red_m[0][g];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red_o[w][g][d]; l += red_l[w][g]; m = fmaxf(m, red_m[w][g]); }
    const int r = b * G + g;
    if (S == 1) {
      ((half_t*)a.out)[a.o_frag ? frag_elem(r, h * 64 + d, a.H * 64) : (int64_t)r * a.o_ld + h * 64 + d] = (half_t)(o / l);
    } else {
      const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
      ((half_t*)a.part_o)[pi * 64 + d] = (half_t)(nkeys > 0 ? o / l : 0.f);
      if (d == 0) {
        a.part_ml[pi * 2 + 0] = nkeys > 0 ? m : WH_NEG_INF;
        a.part_ml[pi * 2 + 1] = nkeys > 0 ? l : 0.f;
      }
    }
  }
}

// =============================================================================================
// The same beam-group cross attention with WHOLE cache lines per request (round 6).  In the kernel above an MFMA operand row is
// shared by 4 lanes, so a wave instruction fetches 16 half lines (64 B of a key's 128-byte head row, or of a V^T row): 256 half-line
// requests per 64 keys, and the launch streams at 3.5 TB/s where the per-row kernel (8 lanes per key, whole lines) reaches 5.
// Here both products use the "diagonal" tile of gemv8_kernel: the 16 A rows are 8 keys x the 2 halves of the head (first product) /
// 8 head dims x the 2 halves of a 64-key block (second product), the 16 B columns the 8 beams x the same 2 halves; lane
// l = 16 c + 8 half + i requests 16 bytes at offset 64 half + 16 c of row i — 8 lanes per row, 8 whole lines per wave instruction,
// 128 line requests per 64 keys.  Of each 16 x 16 product only the two diagonal 8 x 8 blocks (same half on both sides) mean
// anything: S[key][beam] = C[key][beam] + C[8 + key][8 + beam] (one DPP row_ror:8 + one v_permlane32_swap per register); half of the
// MFMA work is discarded, irrelevant next to the stream.  P goes from the S accumulators (4 consecutive keys per lane, 16 lanes) to
// the B operand of the second product (8 consecutive keys per lane) through 1 KB of wave-private LDS per 64-key block.
// One 64-key block per wave, everything requested up front (one HBM round trip), WAVES x 64 keys per (split, head, audio).
// =============================================================================================
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_group_diag_kernel(whk::DecAttnArgs a) {
  pin_kernargs(a);
  __shared__ float red_o[WAVES][8][64];
  __shared__ float red_m[WAVES][8], red_l[WAVES][8];
  __shared__ __attribute__((aligned(16))) half_t p_lds[WAVES][8][64];          // [wave][beam][key of the block]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int S = a.splits, G = a.kv_group, Tk = a.Tk;
  int chunk = (Tk + S - 1) / S;
  chunk = (chunk + 127) / 128 * 128;                                  // (<= WAVES * 64 keys: launcher)
  const int k0 = s * chunk;
  int k1 = k0 + chunk; if (k1 > Tk) k1 = Tk;
  const int nkeys = k1 > k0 ? k1 - k0 : 0;
  const int i8 = lane & 7, half = (lane >> 3) & 1, c4 = lane >> 4;    // lane = 16 c4 + 8 half + i8
  const int eoff = 32 * half + 8 * c4;                                // element offset inside a 64-element row
  const int64_t hs = a.kv_hs ? a.kv_hs : 64;
  const int kb = k0 + wave * 64;                                      // this wave's block of 64 keys

  // q: B operand of the first product — lane (c4, half, beam i8) holds head dims [32 half + 8 c4, +8), pre-scaled (exact in fp16)
  const half8v qf = scale_q(*(const half8v*)((const half_t*)a.q + (int64_t)(b * G + (i8 < G ? i8 : G - 1)) * a.q_ld + h * 64 + eoff));
  asm volatile("" ::: "memory");
  // K: 8 wave-loads, load j = keys kb + 8 j + i8 (whole 128-byte head rows); V^T: 8 wave-loads, load g = head dims 8 g + i8, keys
  // kb .. kb + 63 (whole lines of the transposed rows)
  const half_t* kp = (const half_t*)a.k + (int64_t)b * a.k_bs + h * hs + eoff;
  const int klast = nkeys > 0 ? k1 - 1 : 0;
  const uint32_t ldk = (uint32_t)a.k_ld;
  half8v kf[8], vf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int key = kb + 8 * j + i8;
    if (key > klast) key = klast;                                     // masked below; the load stays in bounds
    kf[j] = __builtin_nontemporal_load((const half8v*)(kp + (uint32_t)key * ldk));
  }
  {
    // the rows are padded to S * chunk keys (zeros beyond Tk: task_reset_impl); a block that lies wholly beyond this split's keys has
    // P = 0 everywhere and is pointed at the last 64 columns of the row, so that what it multiplies by 0 is finite and inside the row
    int col = kb; if (col > (int)a.vt_ld - 64) col = (int)a.vt_ld - 64;
    const half_t* vp = (const half_t*)a.vt + (int64_t)b * a.vt_bs + (int64_t)(h * 64 + i8) * a.vt_ld + col + eoff;
#pragma unroll
    for (int g = 0; g < 8; ++g) vf[g] = __builtin_nontemporal_load((const half8v*)(vp + (int64_t)(8 * g) * a.vt_ld));
  }
  // every request goes out before the first use (see the kernel above)
  __builtin_amdgcn_sched_barrier(0);

  // ---- scores: after the diagonal sum lane (c4 < 2, half = 0, beam i8) — and its three mirror lanes — holds keys kb + 8 j + 4 c4 + e
  const int ka = 4 * (c4 & 1);
  float4v sc[8];
  float mx = WH_NEG_INF;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[j], qf, c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // C[key][beam] + C[8 + key][8 + beam]: the partner is lane ^ 40 (ror 8 inside the row of 16, then the other half of the wave).
      // Exact in the lanes that are used below (c4 < 2, half = 0) and their partners; the other lanes hold sums nobody reads.
      float p_, q_; lane_swap32(lane_xor8(c[e]), p_, q_);
      float z = c[e] + (lane < 32 ? q_ : p_);
      if (kb + 8 * j + ka + e >= k1) z = WH_NEG_INF;
      c[e] = z;
      mx = fmaxf(mx, z);
    }
    sc[j] = c;
  }
  // lanes (c4, beam) and (c4 ^ 1, beam) hold the two halves of the block's keys for that beam: xor 16
  { float a_, b_; lane_swap16(mx, a_, b_); mx = fmaxf(a_, b_); }
  if (lane < 8) red_m[wave][lane] = mx;
  __syncthreads();
  float M = red_m[0][i8];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) M = fmaxf(M, red_m[w][i8]);

  // ---- p = exp(s - M): row sums, and P into LDS as [beam][key] (the 16 lanes c4 < 2, half = 0 write; every lane computes)
  float lsum = 0.f;
  half_t* pw = &p_lds[wave][i8][0];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    half4v p4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = (sc[j][e] == WH_NEG_INF) ? 0.f : __expf(sc[j][e] - M);
      lsum += p;
      p4[e] = (half_t)p;
    }
    if (lane < 32 && half == 0) *(half4v*)(pw + 8 * j + ka) = p4;
  }
  { float a_, b_; lane_swap16(lsum, a_, b_); lsum = a_ + b_; }          // both key halves of the block
  if (lane < 8) red_l[wave][lane] = lsum;
  // B operand of the second product: lane (c4, half, beam i8) = P[keys 32 half + 8 c4 .. + 7][beam].  Wave-private LDS: the LDS
  // queue of a wave is in order, so no workgroup barrier — only the compiler must not move the read above the writes
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  const half8v pf = *(const half8v*)(&p_lds[wave][i8][eoff]);

  // ---- O^T[dim][beam] = V^T . P on the diagonal tile, 8 head dims per product
  float4v oacc[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    float4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[g], pf, c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p_, q_; lane_swap32(lane_xor8(c[e]), p_, q_);
      c[e] += lane < 32 ? q_ : p_;
    }
    oacc[g] = c;
  }
  // lane (c4 < 2, half = 0, beam i8) holds dims 8 g + 4 c4 + e
  if (lane < 32 && half == 0) {
#pragma unroll
    for (int g = 0; g < 8; ++g) *(float4v*)&red_o[wave][i8][8 * g + ka] = oacc[g];
  }
  __syncthreads();
  for (int t = tid; t < G * 64; t += WAVES * 64) {
    const int g = t >> 6, d = t & 63;
    float o = red_o[0][g][d], l = red_l[0][g], m = red_m[0][g];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) { o += red_o[w][g][d]; l += red_l[w][g]; m = fmaxf(m, red_m[w][g]); }
    const int r = b * G + g;
    if (S == 1) {
      ((half_t*)a.out)[a.o_frag ? frag_elem(r, h * 64 + d, a.H * 64) : (int64_t)r * a.o_ld + h * 64 + d] = (half_t)(o / l);
    } else {
      const int64_t pi = ((int64_t)s * a.R + r) * a.H + h;
      ((half_t*)a.part_o)[pi * 64 + d] = (half_t)(nkeys > 0 ? o / l : 0.f);
      if (d == 0) {
        a.part_ml[pi * 2 + 0] = nkeys > 0 ? m : WH_NEG_INF;
        a.part_ml[pi * 2 + 1] = nkeys > 0 ? l : 0.f;
      }
    }
  }
}

// =============================================================================================
// cross QK capture
// =============================================================================================
template <typename T>
__global__ void cross_qk_kernel(const T* __restrict__ q, int64_t q_ld, const T* __restrict__ k,
                                int64_t k_ld, int head, int Tk, float* __restrict__ out) {
  __shared__ float qs[64];
  const int t = blockIdx.y;
  if (threadIdx.x < 64) qs[threadIdx.x] = to_f32(q[(int64_t)t * q_ld + head * 64 + threadIdx.x]);
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Tk) return;
  const T* kr = k + (int64_t)j * k_ld + head * 64;
  float s = 0.f;
#pragma unroll 16
  for (int d = 0; d < 64; ++d) s = __builtin_fmaf(qs[d], to_f32(kr[d]), s);
  out[(int64_t)t * Tk + j] = s * SCALE;
}

// all (row, pair) score planes of a batch in one launch (word timestamps for a batch of clips)
template <typename T>
__global__ void cross_qk_batch_kernel(const T* __restrict__ qcap, int64_t q_layer_stride, int64_t q_row_stride, int D,
                                      const T* __restrict__ cross_kv, int64_t kv_layer_stride, int64_t kv_audio_stride,
                                      int kv_group, const int* __restrict__ layers, const int* __restrict__ heads,
                                      int n_pairs, const int* __restrict__ ntok, int Tmax, int Tk, float* __restrict__ out) {
  __shared__ float qs[64];
  const int t = blockIdx.y;
  const int r = blockIdx.z / n_pairs, p = blockIdx.z - r * n_pairs;
  if (t >= ntok[r]) return;                        // workgroup-uniform
  const int l = layers[p], head = heads[p];
  const T* q = qcap + l * q_layer_stride + r * q_row_stride + (int64_t)t * D + head * 64;
  if (threadIdx.x < 64) qs[threadIdx.x] = to_f32(q[threadIdx.x]);
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Tk) return;
  const T* kr = cross_kv + l * kv_layer_stride + (r / kv_group) * kv_audio_stride + (int64_t)j * 2 * D + head * 64;
  float s = 0.f;
#pragma unroll 16
  for (int d = 0; d < 64; ++d) s = __builtin_fmaf(qs[d], to_f32(kr[d]), s);
  out[(((int64_t)r * n_pairs + p) * Tmax + t) * Tk + j] = s * SCALE;
}

// The same score planes on the matrix cores (fp16).  The kernel above re-reads a clip's 1500 keys of a head for every
// one of the ~200 tokens (100 GB of L2 reads for 8 clips x 320 pairs: 6.4-7.9 ms per launch in
// profiles/r03_kernel_stats_extras.csv); here a workgroup owns 64 tokens x 128 frames of one (row, pair) plane: each wave
// keeps its 16 tokens' q as the A operand (2 x 8 halves per lane) and walks 8 sub-tiles of 16 frames, K rows straight from
// global memory in B-fragment layout (lane = frame, 16 bytes of the head row), two v_mfma_f32_16x16x32_f16 per sub-tile.
// out[token 4 (lane >> 4) + e][frame lane & 15] = acc[e] * d_head^-0.5: 16 lanes write 64 contiguous bytes of a token row.
__global__ __launch_bounds__(256) void cross_qk_batch_mfma_kernel(const half_t* __restrict__ qcap, int64_t q_layer_stride,
                                                                  int64_t q_row_stride, int D, const half_t* __restrict__ cross_kv,
                                                                  int64_t kv_layer_stride, int64_t kv_audio_stride, int kv_group,
                                                                  const int* __restrict__ layers, const int* __restrict__ heads,
                                                                  int n_pairs, const int* __restrict__ ntok, int Tmax, int Tk,
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.z / n_pairs, p = blockIdx.z - r * n_pairs;
  const int nt = ntok[r];
  const int t0 = blockIdx.y * 64 + wave * 16;
  if (blockIdx.y * 64 >= nt) return;               // workgroup-uniform: token tile beyond this clip
  const int l = layers[p], head = heads[p];
  const int tq = t0 + (lane & 15);
  const half_t* q = qcap + l * q_layer_stride + r * q_row_stride + (int64_t)(tq < nt ? tq : nt - 1) * D + head * 64 + 8 * (lane >> 4);
  const half8v qa0 = *(const half8v*)q, qa1 = *(const half8v*)(q + 32);
  const half_t* kbase = cross_kv + l * kv_layer_stride + (r / kv_group) * kv_audio_stride + head * 64 + 8 * (lane >> 4);
  float* orow = out + (((int64_t)r * n_pairs + p) * Tmax) * Tk;
  const int j0 = blockIdx.x * 128;
#pragma unroll 2
  for (int sub = 0; sub < 8; ++sub) {
    const int j = j0 + sub * 16 + (lane & 15);
    if (j0 + sub * 16 >= Tk) break;                // wave-uniform
    const half_t* kr = kbase + (int64_t)(j < Tk ? j : Tk - 1) * 2 * D;
    const half8v kb0 = *(const half8v*)kr, kb1 = *(const half8v*)(kr + 32);
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa0, kb0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa1, kb1, acc, 0, 0, 0);
    if (j < Tk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = t0 + 4 * (lane >> 4) + e;
        if (t < nt) orow[(int64_t)t * Tk + j] = acc[e] * SCALE;
      }
    }
  }
}

}  // namespace

namespace whk {

hipError_t launch_cross_qk_batch(const void* qcap, int64_t q_layer_stride, int64_t q_row_stride, int D, const void* cross_kv,
                                 int64_t kv_layer_stride, int64_t kv_audio_stride, int kv_group, const int* d_layers,
                                 const int* d_heads, int n_pairs, const int* d_ntok, int R, int Tmax, int Tk, float* out,
                                 int dtype, hipStream_t stream) {
  if ((int64_t)R * n_pairs > 65535) return hipErrorInvalidValue;
  const bool valu_qk = WH_DEV_FLAG("WH_QK_VALU");   // developer A/B switch
  if (dtype == 1 && !valu_qk && D % 8 == 0) {
    dim3 mgrid((Tk + 127) / 128, (Tmax + 63) / 64, R * n_pairs);
    hipLaunchKernelGGL(cross_qk_batch_mfma_kernel, mgrid, dim3(256), 0, stream, (const half_t*)qcap, q_layer_stride, q_row_stride, D,
                       (const half_t*)cross_kv, kv_layer_stride, kv_audio_stride, kv_group, d_layers, d_heads, n_pairs, d_ntok, Tmax, Tk, out);
    return hipGetLastError();
  }
  dim3 grid((Tk + 255) / 256, Tmax, R * n_pairs), block(256);
  if (dtype == 1)
    hipLaunchKernelGGL((cross_qk_batch_kernel<half_t>), grid, block, 0, stream, (const half_t*)qcap, q_layer_stride, q_row_stride, D,
                       (const half_t*)cross_kv, kv_layer_stride, kv_audio_stride, kv_group, d_layers, d_heads, n_pairs, d_ntok, Tmax, Tk, out);
  else
    hipLaunchKernelGGL((cross_qk_batch_kernel<float>), grid, block, 0, stream, (const float*)qcap, q_layer_stride, q_row_stride, D,
                       (const float*)cross_kv, kv_layer_stride, kv_audio_stride, kv_group, d_layers, d_heads, n_pairs, d_ntok, Tmax, Tk, out);
  return hipGetLastError();
}

hipError_t launch_attn_generic(const AttnArgs& a, int batch, int dtype, hipStream_t stream) {
  dim3 grid((a.Tq + GQ - 1) / GQ, a.H, batch), block(256);
  if (dtype == 1) hipLaunchKernelGGL((attn_generic_kernel<half_t>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_generic_kernel<float>), grid, block, 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_attn_flash_f16(const void* q, int64_t q_ld, int64_t q_bs, const void* k, int64_t k_ld,
                                 int64_t k_bs, const void* vt, int64_t vt_ld, int64_t vt_bs, void* out,
                                 int64_t o_ld, int64_t o_bs, int B, int H, int T, int prescaled, hipStream_t stream,
                                 int Tq) {
  if (Tq <= 0) Tq = T;                               // encoder self attention: as many queries as keys
  dim3 grid((Tq + FQ - 1) / FQ, H, B), block(FW * 64);
#define WH_FLASH_LAUNCH(P, V)                                                                                          \
  hipLaunchKernelGGL((attn_flash_f16_kernel<P, V>), grid, block, 0, stream, (const half_t*)q, q_ld, q_bs,              \
                     (const half_t*)k, k_ld, k_bs, (const half_t*)vt, vt_ld, vt_bs, (half_t*)out, o_ld, o_bs, T, Tq)
  // prescaled: bit 0 = q, k carry the softmax scale; bit 1 (tools/probe_gemm only) = plain V^T tile layout in LDS
  switch (prescaled & 3) {
    case 0: WH_FLASH_LAUNCH(false, true); break;
    case 1: WH_FLASH_LAUNCH(true, true); break;
    case 2: WH_FLASH_LAUNCH(false, false); break;
    default: WH_FLASH_LAUNCH(true, false); break;
  }
#undef WH_FLASH_LAUNCH
  return hipGetLastError();
}

int attn_decode_capacity(int dtype) { return 512 / (dtype == 1 ? 1 : 2); }   // 4 x 16 or 8 x 8 rounds of 8 (4) keys

// Cross attention (fixed 1500 keys, HBM-bound): 4 waves x up to 16 rounds measured fastest (11.7 us vs 12.9 us for
// 8 x 8 at large-v3 B = 8).  Self attention (cached length read on the device): 8 waves x 8 rounds — short
// chains, and rounds beyond the cached length are skipped by a wave-uniform branch.
template <typename T>
static hipError_t launch_attn_decode_t(const DecAttnArgs& a, hipStream_t stream) {
  constexpr int KPW = ET<T>::UNIT;                 // keys per wave-load: 64 lanes / (64 / UNIT lanes per key)
  if (a.d_len) {
    dim3 grid(a.splits, a.H, a.R), block(8 * 64);
    hipLaunchKernelGGL((attn_decode_kernel<T, 8, 8, true>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  const int chunk = (a.Tk + a.splits - 1) / a.splits;
  if constexpr (sizeof(T) == 2) {
    const bool no_mfma = WH_DEV_FLAG("WH_GROUP_ATTN_VALU");   // developer A/B switch
    if (a.vt && !no_mfma && a.kv_group > 1 && a.kv_group <= 8 && a.R % a.kv_group == 0 && chunk <= 512 &&
        (int64_t)a.splits * ((chunk + 127) / 128 * 128) <= a.vt_ld) {
      dim3 ggrid(a.splits, a.H, a.R / a.kv_group);
      // round 6: whole cache lines per request (the diagonal tile); A/B against the kernels below: WH_GROUP_ATTN_HALF_LINES=1
      if (!WH_DEV_FLAG("WH_GROUP_ATTN_HALF_LINES") && a.vt_ld >= 64) {
        hipLaunchKernelGGL((attn_decode_group_diag_kernel<8>), ggrid, dim3(512), 0, stream, a);
        return hipGetLastError();
      }
      // 8 waves x 2 blocks and 4 waves x 4 blocks of 32 keys measured the same (232.8 vs 233.0 ms per 64-step beam pass)
      if ((chunk + 127) / 128 <= 1) hipLaunchKernelGGL((attn_decode_group_mfma_kernel<1, 8>), ggrid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((attn_decode_group_mfma_kernel<2, 8>), ggrid, dim3(512), 0, stream, a);
      return hipGetLastError();
    }
  }
  if (a.kv_group > 1 && a.kv_group <= 8 && a.R % a.kv_group == 0 && chunk <= 8 * 8 * KPW) {
    // beam groups: one workgroup per (split, head, audio) scores all beams against one K/V tile
    dim3 ggrid(a.splits, a.H, a.R / a.kv_group), gblock(8 * 64);
    hipLaunchKernelGGL((attn_decode_group_kernel<T, 8, 8, 8>), ggrid, gblock, 0, stream, a);
    return hipGetLastError();
  }
  dim3 grid(a.splits, a.H, a.R), block(4 * 64);
  const int rounds = (chunk + 4 * KPW - 1) / (4 * KPW);
  if (rounds > 16) return hipErrorInvalidValue;
  if (rounds <= 8) hipLaunchKernelGGL((attn_decode_kernel<T, 8, 4, false>), grid, block, 0, stream, a);
  else if (rounds <= 12) hipLaunchKernelGGL((attn_decode_kernel<T, 12, 4, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_decode_kernel<T, 16, 4, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_attn_decode(const DecAttnArgs& a, int dtype, hipStream_t stream) {
  // a split must fit the register-resident K/V tile: callers size `splits` with attn_decode_capacity()
  if (a.splits < 1 || a.splits > DEC_ATTN_MAX_SPLITS) return hipErrorInvalidValue;
  // the in-launch merge exists in attn_decode_kernel<half> only, for up to 4 splits: anything else must not be asked for silently
  if (a.merge_cnt && (dtype != 1 || a.kv_group != 1 || a.splits < 2 || a.splits > 4 || !a.out)) return hipErrorInvalidValue;
  if (a.o_frag && dtype != 1) return hipErrorInvalidValue;
  if ((int64_t)a.k_ld * 2048 > 0x7fffffff || (int64_t)a.v_ld * 2048 > 0x7fffffff) return hipErrorInvalidValue;
  return dtype == 1 ? launch_attn_decode_t<half_t>(a, stream) : launch_attn_decode_t<float>(a, stream);
}

hipError_t launch_cross_qk(const void* q, int64_t q_ld, const void* k, int64_t k_ld, int head, int n_tok,
                           int Tk, float* out, int dtype, hipStream_t stream) {
  dim3 grid((Tk + 255) / 256, n_tok), block(256);
  if (dtype == 1)
    hipLaunchKernelGGL((cross_qk_kernel<half_t>), grid, block, 0, stream, (const half_t*)q, q_ld, (const half_t*)k, k_ld, head, Tk, out);
  else
    hipLaunchKernelGGL((cross_qk_kernel<float>), grid, block, 0, stream, (const float*)q, q_ld, (const float*)k, k_ld, head, Tk, out);
  return hipGetLastError();
}

}  // namespace whk
