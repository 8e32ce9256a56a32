// gemm.hip — MFMA "NT" GEMM for gfx950:  C[m][n] = act( sum_k A[m][k] * W[n][k] + bias ) + residual
//
// This is `Linear.forward` (whisper/model.py:44-50: x @ W.T + b with W stored [out][in]) for the
// token-parallel parts of the path: every encoder projection / MLP (1500 frames x batch rows), the two
// convolutions expressed as GEMMs over overlapping rows (model.py:193-194), the cross-attention K/V
// projection (model.py:101-105) and the decoder prefill.  The width-1 decode step does NOT come here
// (see gemv.hip).
//
// Structure (CDNA4): every wave owns a 64x64 output block (4x4 MFMA 16x16 tiles); a workgroup is 4x4 waves on a
// 256x256 tile (encoder-sized problems) or 2x2 waves on 128x128.  K is consumed 128 bytes per step (64 fp16 /
// 32 fp32).  Both operands are K-contiguous, so a tile row is exactly 8 units of 16 bytes; tiles are brought in with
// global_load_lds (16 B/lane, no VGPR round trip) into a two-buffer ring, one barrier per K step.  The LDS image is
// lane-linear per wave instruction, so the bank-conflict swizzle (unit ^ ((row>>1)&7)) is applied on the per-lane
// *source* address and again on the ds_read_b128 fragment reads.  The MFMA is issued with the weight fragment as
// the A operand, so each lane ends up holding 4 consecutive n of one m -> vectorised bias/residual loads and stores.
// fp16: v_mfma_f32_16x16x32_f16; fp32 (strict-parity mode): v_mfma_f32_16x16x4_f32, an exact fp32 FMA chain.
// Workgroup ids are remapped so each XCD's L2 sees a contiguous band of tiles.
//
// Two kernels: gemm_nt_kernel (any dtype / residual / alignment, one tile per workgroup) and gemm_f16_rows_kernel,
// the fp16-in fp16-out 256x256 case that is most of the encoder (QKV, V^T, fc1, cross K/V), see below.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __forceinline__ void mma16(half8v a, half8v b, float4v& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(float4v a, float4v b, float4v& c) {
  // lane (i, g) holds k = 16q + 4g + e in element e: step e sums the four k of the four lane groups
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
}

__device__ __forceinline__ void glds16(const void* g, char* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

template <typename OutT> struct Out4;
template <> struct Out4<float> {
  static __device__ __forceinline__ void store(float* p, float4v v) { *(float4v*)p = v; }
};
template <> struct Out4<half_t> {
  static __device__ __forceinline__ void store(half_t* p, float4v v) {
    half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *(half4v*)p = h;
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void order_memory_ops() { asm volatile("" ::: "memory"); }

constexpr int FM = 4, FN = 4;            // MFMA 16x16 tiles per wave: a 64x64 block


// The general kernel: any dtype, residual, alignment; one tile per workgroup.
constexpr int NS = 2;                    // LDS ring depth (tiles in flight = NS - 1)
template <typename T, typename OutT, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_nt_kernel(whk::GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int NW = WGM * WGN;
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int IA = BM / (NW * 8), IW = BN / (NW * 8);      // wave-loads (8 rows each) per wave per K step
  static_assert(IA * NW * 8 == BM && IW * NW * 8 == BN, "tile rows must split evenly over the waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware, bijective remap of the linear workgroup id (block b runs on XCD b % 8)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int64_t bz = blockIdx.z;

  const T* A = (const T*)p.A + bz * p.a_bs;
  const T* W = (const T*)p.W + bz * p.w_bs;

  // staging: wave issues IA A + IW W instructions per K step; instruction i covers tile rows
  // [(wave*I+i)*8, +8): lane -> row (lane>>3), LDS unit slot (lane&7)
  const T* ga[IA];
  const T* gw[IW];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int r = (wave * IA + i) * 8 + (lane >> 3);
    const int u = (lane & 7) ^ ((r >> 1) & 7);
    int am = m0 + r; if (am > p.M - 1) am = p.M - 1;
    ga[i] = A + (int64_t)am * p.lda + u * UNIT;
  }
#pragma unroll
  for (int i = 0; i < IW; ++i) {
    const int r = (wave * IW + i) * 8 + (lane >> 3);
    const int u = (lane & 7) ^ ((r >> 1) & 7);
    int wr = n0 + r; if (wr > p.N - 1) wr = p.N - 1;
    gw[i] = W + (int64_t)wr * p.ldw + u * UNIT;
  }

  float4v acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  // bias of this lane's outputs, requested before the K loop (fetched after it, the epilogue starts with a dependent
  // round trip: 5 us per 256x256 tile in the phase probe): 4 consecutive n per tn, or one value per tm
  float4v bpre[FN];
  float bmpre[FM];
#pragma unroll
  for (int tn = 0; tn < FN; ++tn) {
    bpre[tn] = float4v{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + wn * (FN * 16) + tn * 16 + (lane >> 4) * 4;
    if (p.bias && !p.bias_on_m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bpre[tn][e] = p.bias[n + e < p.N ? n + e : p.N - 1];
    }
  }
#pragma unroll
  for (int tm = 0; tm < FM; ++tm) {
    const int m = m0 + wm * (FM * 16) + tm * 16 + (lane & 15);
    bmpre[tm] = (p.bias && p.bias_on_m) ? p.bias[m < p.M ? m : p.M - 1] : 0.f;
  }

  const int nk = p.K / BKE;
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE_BYTES + (wave * IA) * 1024;
    char* sW = smem + buf * STAGE_BYTES + A_BYTES + (wave * IW) * 1024;
    const int64_t ko = (int64_t)kt * BKE;
#pragma unroll
    for (int i = 0; i < IA; ++i) glds16(ga[i] + ko, sA + i * 1024);
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(gw[i] + ko, sW + i * 1024);
  };

  // Ring of NS tile buffers, ONE barrier per K step: wait for my loads of tile kt (tiles kt+1 .. kt+NS-2 stay in
  // flight), barrier (=> every wave sees tile kt AND has finished reading tile kt-1), refill tile kt-1's buffer
  // with tile kt+NS-1, then the MFMAs of tile kt run while those loads travel.
  constexpr int LPS = IA + IW;
  static_assert(NS >= 2 && NS <= 4 && LPS * (NS - 2) <= 63, "vmcnt immediates");
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nk) stage(i, i);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;                 // tiles issued after tile kt that may still be in flight
    if (NS >= 4 && ahead >= 2) wait_vmcnt<LPS * 2>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<(NS >= 3 ? LPS : 0)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nk) stage((kt + NS - 1) % NS, kt + NS - 1);

    const char* sA = smem + (kt % NS) * STAGE_BYTES;
    const char* sW = sA + A_BYTES;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      unit_t af[FM], wf[FN];
      const int u = 4 * q + (lane >> 4);
#pragma unroll
      for (int t = 0; t < FM; ++t) af[t] = *(const unit_t*)(sA + swz_byte(wm * (FM * 16) + t * 16 + (lane & 15), u));
#pragma unroll
      for (int t = 0; t < FN; ++t) wf[t] = *(const unit_t*)(sW + swz_byte(wn * (FN * 16) + t * 16 + (lane & 15), u));
#pragma unroll
      for (int tn = 0; tn < FN; ++tn)
#pragma unroll
        for (int tm = 0; tm < FM; ++tm) mma16(wf[tn], af[tm], acc[tn][tm]);
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3] for m = m0+wm*FM*16+tm*16+(lane&15), n = n0+wn*FN*16+tn*16+(lane>>4)*4
  OutT* C = (OutT*)p.C + bz * p.c_bs;
  const float* R = p.res ? p.res + bz * p.r_bs : nullptr;
  const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
  if (vec_ok && R && p.act == 0 && n0 + BN <= p.N && !p.serial_epilogue) {
    // Residual GEMMs whose tile lies inside N (`attn.out`, `mlp.2`: N = 1280): branch-free, the four float4 of a 16-row band
    // requested together before any is added (two bands at once spill at the 128 VGPRs of a 16-wave workgroup).  The general loop below asks for one
    // float4, waits, stores, 16 times over — 16 (here: 4) dependent HBM round trips per wave at the moment every CU of a single-
    // round GEMM is in its epilogue: 16 of the 55 us of `attn.out`.  Rows beyond M re-read row M - 1 and store nothing.
#pragma unroll
    for (int tm = 0; tm < FM; ++tm) {
      const int m = m0 + wm * (FM * 16) + tm * 16 + (lane & 15);
      const int mc = m < p.M ? m : p.M - 1;
      const int rm = (p.res_mod > 0) ? (mc % p.res_mod) : mc;
      const float* rrow = R + (int64_t)rm * p.ldr + n0 + wn * (FN * 16) + (lane >> 4) * 4;
      float4v r4[FN];
#pragma unroll
      for (int tn = 0; tn < FN; ++tn) r4[tn] = *(const float4v*)(rrow + tn * 16);
      const float bm = bmpre[tm];
      OutT* crow = C + (int64_t)m * p.ldc + n0 + wn * (FN * 16) + (lane >> 4) * 4;
#pragma unroll
      for (int tn = 0; tn < FN; ++tn) {
        float4v v = acc[tn][tm];
        v += bpre[tn];
        v[0] += bm; v[1] += bm; v[2] += bm; v[3] += bm;
        v += r4[tn];
        if (m < p.M) Out4<OutT>::store(crow + tn * 16, v);
      }
    }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < FM; ++tm) {
    const int m = m0 + wm * (FM * 16) + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const int rm = (p.res_mod > 0) ? (m % p.res_mod) : m;
    const float bm = bmpre[tm];
#pragma unroll
    for (int tn = 0; tn < FN; ++tn) {
      const int n = n0 + wn * (FN * 16) + tn * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      float4v v = acc[tn][tm];
      v += bpre[tn];
      v[0] += bm; v[1] += bm; v[2] += bm; v[3] += bm;
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      if (vec_ok && n + 3 < p.N) {
        if (R) {
          const float4v r4 = *(const float4v*)(R + (int64_t)rm * p.ldr + n);
          v += r4;
        }
        Out4<OutT>::store(C + (int64_t)m * p.ldc + n, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < p.N) {
            float x = v[e];
            if (R) x += R[(int64_t)rm * p.ldr + n + e];
            C[(int64_t)m * p.ldc + n + e] = from_f32<OutT>(x);
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// fp16 in, fp16 out, no residual, 256x256 tiles: QKV / V^T / fc1 of the encoder and the cross-attention K/V projection.
//
// What the round-2 phase probe (tools/probe_gemm -DWH_PROBE, s_memtime per workgroup) showed for gemm_nt_kernel on
// these shapes (M = 12000, K = 1280): the K loop of a tile takes 28 us and the 13 us around it were the problem —
// 5.4 us waiting for bias loads issued after the last MFMA, 3.4 us issuing 32-byte-per-row stores straight from the
// accumulator layout (16 rows per instruction), 3.9 us for the next workgroup's first loads.  Here:
//   * persistent workgroups (one per CU) walk the tiles when there are more tiles than CUs;
//   * the next tile's first K tile is requested right after the last MFMA, BEFORE the epilogue's stores.  vmcnt
//     retires in issue order across loads and stores, so the wait for it is counted (vmcnt(stores)) and does not
//     wait for the stores;
//   * the output goes through a 2 KB per-wave LDS patch (beside the ring, never part of it) that turns the
//     accumulator layout into whole 128-byte rows: 8 global_store_dwordx4 per wave instead of 16 dwordx2;
//   * the 64 bias values of a wave's block are fetched into one VGPR before the K loop and handed round through the
//     same patch, so no global load sits between the last MFMA and the first store;
//   * GELU uses gelu_erf_f16out (common.h);
//   * LDS is laid out [A buf 0][A buf 1][W buf 0][W buf 1][patches]: the two buffers of an operand are 32 KB apart,
//     inside the offset field of ds_read_b128, so with the K loop written out for buffer 0 and buffer 1 a lane keeps
//     four fragment addresses and nothing per buffer.  A 16-wave workgroup has 128 VGPRs per lane; 64 are
//     accumulators and 32 fragments, and anything the compiler hoists out of the tile walk spills — hence opaque().
// Tried and dropped (same probe, same box, A/B): requesting all 16 fragments of a K step before its MFMAs (+-0),
// s_setprio(1) around the MFMAs (-35 %), non-temporal stores (+-0), half of the waves issuing every load and the
// other half every store (-25 %: eight 1 KB LDS-DMA pieces per wave and step delay that wave's own MFMAs).
// ------------------------------------------------------------------------------------------------------------------
constexpr int ROWS_PATCH = 2048;                              // per-wave epilogue patch: 16 rows x 128 B
constexpr int ROWS_LDS = 2 * (256 + 256) * 128 + 16 * ROWS_PATCH;
static_assert(ROWS_LDS <= 160 * 1024, "LDS ring + epilogue patches do not fit");

// BIAS: 0 none, 1 per output column n, 2 per output row m
template <int ACT, int BIAS>
__global__ __launch_bounds__(1024) void gemm_f16_rows_kernel(whk::GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, WGN = 4;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, W_BASE = 2 * A_BYTES, RING_BYTES = 2 * (A_BYTES + W_BYTES);
  constexpr int IA = 2, IW = 2, LPS = IA + IW;                // wave-loads (8 rows each) per wave per K step
  constexpr int NST = FM * 2;                                 // global stores per wave and tile
  static_assert(A_BYTES + 3 * 2048 < 65536 && W_BYTES + 3 * 2048 < 65536, "ds_read offset field");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // Tile walk.  One-shot launches (grid = tiles): XCD-aware, bijective remap of the linear workgroup id (block b runs
  // on XCD b % 8), one tile per workgroup.  Persistent launches (p.persistent, grid a multiple of 8 workgroups): the
  // workgroups of XCD x share that XCD's contiguous band of tiles and walk it with stride grid/8.
  const int nwg = gridDim.x, orig = blockIdx.x;
  int tile, tile_end, tile_step;
  {
    const int xcd = orig & 7;
    const int total = p.persistent ? p.tiles_m * p.tiles_n : nwg;
    const int qq = total >> 3, rr = total & 7;
    const int band = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
    tile = band + (orig >> 3);
    tile_end = p.persistent ? band + qq + (xcd < rr ? 1 : 0) : tile + 1;
    tile_step = p.persistent ? (nwg >> 3) : 1;
  }
  if (tile >= tile_end) return;
  const int64_t bz = blockIdx.z;
  const char* A = (const char*)((const half_t*)p.A + bz * p.a_bs);
  const char* W = (const char*)((const half_t*)p.W + bz * p.w_bs);
  half_t* C = (half_t*)p.C + bz * p.c_bs;
  const int nk = p.K / 64;

  // A 16-wave workgroup has 128 VGPRs per lane; 64 are accumulators, 32 fragments.  Lane-only address arithmetic is
  // invariant across the tile walk, and whatever the compiler hoists out of it is held through the K loop (and
  // spills), so the lane id is produced afresh where it is needed and the staging offsets are kept opaque (otherwise
  // base + offset is hoisted as 64-bit pairs).
  auto fresh_lane = []() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto opaque = [](uint32_t v) { asm volatile("" : "+v"(v)); return v; };

  // staging: a wave issues IA A + IW W instructions per K step; instruction i covers tile rows
  // [(wave*I+i)*8, +8): lane -> row (lane>>3), LDS unit slot (lane&7).  32-bit byte offsets from the uniform bases:
  // the launcher refuses operands of 4 GB or more.
  int m0, n0;
  uint32_t ga[IA], gw[IW];
  auto set_tile = [&](int t) {
    const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
    m0 = tile_m * BM; n0 = tile_n * BN;
    const int ln = fresh_lane();
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int r = (wave * IA + i) * 8 + (ln >> 3);
      const int u = (ln & 7) ^ ((r >> 1) & 7);
      int am = m0 + r; if (am > p.M - 1) am = p.M - 1;
      ga[i] = (uint32_t)(((int64_t)am * p.lda + u * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < IW; ++i) {
      const int r = (wave * IW + i) * 8 + (ln >> 3);
      const int u = (ln & 7) ^ ((r >> 1) & 7);
      int wr = n0 + r; if (wr > p.N - 1) wr = p.N - 1;
      gw[i] = (uint32_t)(((int64_t)wr * p.ldw + u * 8) * 2);
    }
  };
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * A_BYTES + (wave * IA) * 1024;
    char* sW = smem + W_BASE + buf * W_BYTES + (wave * IW) * 1024;
    const char* Ak = A + (int64_t)kt * 128;                   // one K step = 128 bytes of a row
    const char* Wk = W + (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < IA; ++i) glds16(Ak + opaque(ga[i]), sA + i * 1024);
#pragma unroll
    for (int i = 0; i < IW; ++i) glds16(Wk + opaque(gw[i]), sW + i * 1024);
  };

  // fragment reads: row wm*64 + t*16 + (lane&15), 16-byte unit 4q + (lane>>4), swizzled by ((row>>1)&7) = (lane>>1)&7
  uint32_t fa[2], fw[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int su = ((4 * q + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4;
    fa[q] = (wm * 64 + (lane & 15)) * 128 + su;
    fw[q] = W_BASE + (wn * 64 + (lane & 15)) * 128 + su;
  }

  WH_PROBE_AT(p, orig, 0);
  set_tile(tile);
  stage(0, 0);
  // counted: this wave issued exactly NST stores after the stage() call above (a block that lies fully inside C)
  bool counted = false;
  char* patch = smem + RING_BYTES + wave * ROWS_PATCH;

  for (int visit = 0;; ++visit) {
    // the 64 bias values this wave's block needs, one per lane, requested now and used after the K loop
    float bias_lane = 0.f;
    if (BIAS != 0) {
      int bi = (BIAS == 2 ? m0 + wm * 64 : n0 + wn * 64) + fresh_lane();
      const int lim = (BIAS == 2 ? p.M : p.N) - 1;
      if (bi > lim) bi = lim;
      bias_lane = p.bias[bi];
    }

    float4v acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    // K loop.  K tile 0 is already on its way, requested BEFORE this wave's NST epilogue stores of the previous tile:
    // when those are known to have been issued the wait for it is vmcnt(NST) and does not wait for the stores.
    // After that every trip requests the next K tile into the buffer all waves left at the last barrier, runs the
    // MFMAs of the current one, and waits for the request (the newest one: vmcnt(0)) + one barrier.
    if (counted) wait_vmcnt<NST>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    WH_PROBE_AT(p, orig, visit == 0 ? 1 : 7);
    auto mfmas = [&](auto buf_tag) {
      constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        half8v af[FM], wf[FN];
#pragma unroll
        for (int t = 0; t < FM; ++t) af[t] = *(const half8v*)(smem + fa[q] + (BUF * A_BYTES + t * 2048));
#pragma unroll
        for (int t = 0; t < FN; ++t) wf[t] = *(const half8v*)(smem + fw[q] + (BUF * W_BYTES + t * 2048));
#pragma unroll
        for (int tn = 0; tn < FN; ++tn)
#pragma unroll
          for (int tm = 0; tm < FM; ++tm) mma16(wf[tn], af[tm], acc[tn][tm]);
      }
    };
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; kt += 2) {             // two K steps per trip (nk is even): buffer 0, then buffer 1
      stage(1, kt + 1);
      mfmas(std::integral_constant<int, 0>());
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 2 < nk;
      if (more) stage(0, kt + 2);
      mfmas(std::integral_constant<int, 1>());
      if (more) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (visit == 0) WH_PROBE_AT(p, orig, 2);

    // ---- between two K loops -------------------------------------------------------------------------------------
    const int cm0 = m0, cn0 = n0;                    // the finished tile
    const int next = tile + tile_step;
    const bool has_next = next < tile_end;
    __builtin_amdgcn_s_barrier();                    // every wave has read the last K tile: the ring is free
    if (visit == 0) WH_PROBE_AT(p, orig, 3);
    if (has_next) {
      set_tile(next);
      stage(0, 0);
      order_memory_ops();                            // the epilogue's stores stay behind this request
    }
    tile = next;

#ifdef WH_GEMM_PROBE_NOEPI                           // tools/probe_gemm: the K-loop floor
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) asm volatile("" ::"v"(acc[i][j]));
    counted = false;
#else
    // ---- epilogue: lane holds C[m][n..n+3] for m = cm0+wm*64+tm*16+(lane&15), n = cn0+wn*64+tn*16+(lane>>4)*4 ----
    const int le = fresh_lane();
    const int i16 = le & 15, g4 = le >> 4;
    if (BIAS != 0) {     // the lane-per-value register goes round through the patch and is added in place
      *(float*)(patch + le * 4) = bias_lane;
      if (BIAS == 2) {
#pragma unroll
        for (int tm = 0; tm < FM; ++tm) {
          const float b = *(const float*)(patch + (tm * 16 + i16) * 4);
#pragma unroll
          for (int tn = 0; tn < FN; ++tn) { acc[tn][tm][0] += b; acc[tn][tm][1] += b; acc[tn][tm][2] += b; acc[tn][tm][3] += b; }
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < FN; ++tn) {
          const float4v b = *(const float4v*)(patch + (tn * 16 + g4 * 4) * 4);
#pragma unroll
          for (int tm = 0; tm < FM; ++tm) acc[tn][tm] += b;
        }
      }
    }
    // whole rows through the patch, one 16-row band (tm) at a time
    const bool full = cm0 + (wm + 1) * 64 <= p.M && cn0 + (wn + 1) * 64 <= p.N;
#pragma unroll
    for (int tm = 0; tm < FM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < FN; ++tn) {
        float4v v = acc[tn][tm];
        if (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f16out(v[e]);
        }
        const half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        const int slot = (tn * 4 + g4) ^ ((i16 & 7) << 1);               // 8-byte slots, pairs stay adjacent
        *(half4v*)(patch + i16 * 128 + slot * 8) = h;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 8 + (le >> 3), c = le & 7;
        const float4v q = *(const float4v*)(patch + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = cm0 + wm * 64 + tm * 16 + row, n = cn0 + wn * 64 + c * 8;
        half_t* dst = C + (int64_t)m * p.ldc + n;
        if (m < p.M) {
          if (n + 8 <= p.N) *(float4v*)dst = q;
          else if (n + 4 <= p.N) *(float2v*)dst = float2v{q[0], q[1]};
        }
      }
    }
    counted = full;
    if (visit == 0) WH_PROBE_AT(p, orig, 5);
#endif
    if (!has_next) break;
  }
}

template <int ACT, int BIAS>
hipError_t launch_rows(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  whk::GemmArgs p = a;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  static whk::LdsAttr attr;
  { hipError_t e = whk::raise_dynamic_lds(attr, (const void*)gemm_f16_rows_kernel<ACT, BIAS>, ROWS_LDS); if (e != hipSuccess) return e; }
  // more tiles than CUs: 256 persistent workgroups (one per CU: the LDS ring allows no second one) walk the tiles
  const int dev = WH_DEV_INT("WH_GEMM_DEV");   // developer switch; 4: never persistent
  const int ntiles = p.tiles_m * p.tiles_n;
  p.persistent = (batch == 1 && ntiles > 256 && !(dev & 4)) ? 1 : 0;
  dim3 grid(p.persistent ? 256 : ntiles, 1, batch);
  hipLaunchKernelGGL((gemm_f16_rows_kernel<ACT, BIAS>), grid, dim3(1024), ROWS_LDS, stream, p);
  return hipGetLastError();
}

// fp16 in / fp16 out, no residual, rows of C 16-byte aligned, N a multiple of 4, operands below 4 GB
bool rows_kernel_applies(const whk::GemmArgs& a) {
  const int dev = WH_DEV_INT("WH_GEMM_DEV");   // developer switch; 1: never
  if (dev & 1) return false;
  if (a.res || (a.ldc & 7) || (a.N & 3) || (a.c_bs & 7) || (((uintptr_t)a.C) & 15)) return false;
  if (((int64_t)a.M * a.lda + a.K) * 2 >= (1LL << 32) || ((int64_t)a.N * a.ldw + a.K) * 2 >= (1LL << 32)) return false;
  if (!(a.M >= 1024 && a.N >= 1024 && a.K % 128 == 0)) return false;      // the K loop takes two 64-wide steps per trip
  // small models (D <= 768: N, K <= 3072) with few 256 x 256 tiles — one 30 s window is 36 - 72 of them on 256 CUs — go to
  // the general kernel's smaller tiles instead (launch_t)
  if (a.N <= 3072 && a.K <= 3072 && (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) < 160) return false;
  return true;
}

hipError_t launch_rows_any(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  const int bias = !a.bias ? 0 : (a.bias_on_m ? 2 : 1);
  if (a.act == 1) {
    if (bias == 0) return launch_rows<1, 0>(a, batch, stream);
    return bias == 1 ? launch_rows<1, 1>(a, batch, stream) : launch_rows<1, 2>(a, batch, stream);
  }
  if (bias == 0) return launch_rows<0, 0>(a, batch, stream);
  return bias == 1 ? launch_rows<0, 1>(a, batch, stream) : launch_rows<0, 2>(a, batch, stream);
}

template <typename T, typename OutT, int WGM, int WGN>
hipError_t launch_shape(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16;
  constexpr int LDS = NS * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "LDS ring does not fit");
  whk::GemmArgs p = a;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  static whk::LdsAttr attr;
  { hipError_t e = whk::raise_dynamic_lds(attr, (const void*)gemm_nt_kernel<T, OutT, WGM, WGN>, LDS); if (e != hipSuccess) return e; }
  p.serial_epilogue = WH_DEV_FLAG("WH_GEMM_SERIAL_EPILOGUE") ? 1 : 0;   // developer A/B switch
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, WGM, WGN>), grid, dim3(WGM * WGN * 64), LDS, stream, p);
  return hipGetLastError();
}

template <typename T, typename OutT>
hipError_t launch_t(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  const int force = WH_DEV_INT("WH_GEMM_TILE");   // 128 / 256: developer override
  // tools/probe_gemm on MI355X (M = 12000): 256x256 is 12-17 % faster than 128x128 on every encoder shape (e.g.
  // N=2560 K=1280: 648 vs 548 TFLOP/s; N=1280 K=5120: 949 vs 809), and among 256x256 layouts 16 waves of 64x64 beat
  // 8 waves of 128x64 (760 vs 650 on the QKV shape); deeper LDS rings at lower occupancy were slower.
  // Few tiles on SMALL models (D <= 768, i.e. N, K <= 3072; BASELINE configs[1] is base x 1 clip: M = 1500, N = 512 ... 2048,
  // 12 - 48 tiles of 256 x 256 on 256 CUs): the tile shrinks until ~200 workgroups exist — 128 x 128 (4 waves), 64 x 128
  // (2 waves), 64 x 64 (1 wave); base x 1 encoder 0.87 -> 0.78 ms.  At larger widths it buys nothing: the 185-token teacher-forced
  // pass of the word-timestamp leg (M = 1480, large-v3) takes 78.7 ms per batch with 128 x 128 tiles everywhere against 79.3 ms
  // with the product's 256 x 256 choice (development build, WH_GEMM_DEV=1 WH_GEMM_TILE=128, gpurun call 11 of round 4), and the
  // 8-clip encoder loses 2 % — so there the round-2 choice stands.  The K order of every output element is the same in
  // all shapes (bit-identical results).
  auto tiles = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * batch; };
  const bool small_model = a.N <= 3072 && a.K <= 3072;
  const bool big = force ? force == 256 : (a.M >= 1024 && a.N >= 1024 && !(small_model && tiles(256, 256) < 160));
  if (big) return launch_shape<T, OutT, 4, 4>(a, batch, stream);
  if (force == 128 || !small_model || tiles(128, 128) >= 192 || (int64_t)a.M * a.N <= 128 * 128)
    return launch_shape<T, OutT, 2, 2>(a, batch, stream);
  if (tiles(64, 128) >= 192) return launch_shape<T, OutT, 1, 2>(a, batch, stream);
  return launch_shape<T, OutT, 1, 1>(a, batch, stream);
}

}  // namespace

namespace whk {

hipError_t launch_gemm(const GemmArgs& a, int dtype, int out_f32, int batch, hipStream_t stream) {
  const int bke = dtype == 1 ? 64 : 32;
  if (a.K % bke != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (dtype == 1) {
    if (!out_f32 && rows_kernel_applies(a)) return launch_rows_any(a, batch, stream);
    return out_f32 ? launch_t<half_t, float>(a, batch, stream) : launch_t<half_t, half_t>(a, batch, stream);
  }
  return launch_t<float, float>(a, batch, stream);
}

}  // namespace whk
