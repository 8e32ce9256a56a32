// gemm.hip — MFMA "NT" GEMM for gfx950:  C[m][n] = act( sum_k A[m][k] * W[n][k] + bias ) + residual
//
// This is `Linear.forward` (whisper/model.py:44-50: x @ W.T + b with W stored [out][in]) for the
// token-parallel parts of the path: every encoder projection / MLP (1500 frames x batch rows), the two
// convolutions expressed as GEMMs over overlapping rows (model.py:193-194), the cross-attention K/V
// projection (model.py:101-105) and the decoder prefill.  The width-1 decode step does NOT come here
// (see gemv.hip).
//
// Structure (CDNA4): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave =
// 4x4 MFMA 16x16 tiles), K consumed 128 bytes per step (64 fp16 / 32 fp32).  Both operands are
// K-contiguous, so a tile row is exactly 8 units of 16 bytes; tiles are brought in with
// global_load_lds (16 B/lane, no VGPR round trip), double-buffered, the next tile left in flight
// across the barrier with a counted vmcnt.  LDS image is lane-linear per wave instruction, so the
// bank-conflict swizzle (unit ^ ((row>>1)&7)) is applied on the per-lane *source* address and again on
// the ds_read_b128 fragment reads.  The MFMA is issued with the weight fragment as the A operand, so
// each lane ends up holding 4 consecutive n of one m -> vectorised bias/residual loads and stores.
// fp16: v_mfma_f32_16x16x32_f16; fp32 (strict-parity mode): v_mfma_f32_16x16x4_f32, an exact fp32
// FMA chain.  Workgroup ids are remapped so each XCD's L2 sees a contiguous band of tiles.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

// Two tile shapes share the code: 128x128 (4 waves, 2x2 of 64x64) for small / skinny problems and 256x256
// (16 waves, 4x4 of 64x64) for the encoder-sized ones.  Measured with tools/probe_gemm (M = 12000): the 256x256
// tile needs half the L2->LDS bytes per FLOP, but what moved the needle was WAVES — the same tile with 8 waves of
// 128x64 gave 650 TFLOP/s on the QKV shape, with 16 waves of 64x64 it gives 760 (fc2, K = 5120: 1040); deeper LDS
// rings (3-4 tiles in flight) at lower occupancy were slower.  One barrier per K step.

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

// WGM x WGN waves, each computing FM x FN MFMA 16x16 tiles
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NS: LDS ring depth (tiles in flight = NS - 1)
template <typename T, typename OutT, int WGM, int WGN, int FM, int FN, int NS>
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
#pragma unroll
  for (int tm = 0; tm < FM; ++tm) {
    const int m = m0 + wm * (FM * 16) + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const int rm = (p.res_mod > 0) ? (m % p.res_mod) : m;
    const float bm = (p.bias && p.bias_on_m) ? p.bias[m] : 0.f;
#pragma unroll
    for (int tn = 0; tn < FN; ++tn) {
      const int n = n0 + wn * (FN * 16) + tn * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      float4v v = acc[tn][tm];
      if (p.bias) {
        if (p.bias_on_m) { v[0] += bm; v[1] += bm; v[2] += bm; v[3] += bm; }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += p.bias[n + e];
        }
      }
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

template <typename T, typename OutT, int WGM, int WGN, int FM, int FN, int NS>
hipError_t launch_shape(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16;
  constexpr int LDS = NS * (BM + BN) * 128;
  static_assert(LDS <= 160 * 1024, "LDS ring does not fit");
  whk::GemmArgs p = a;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<T, OutT, WGM, WGN, FM, FN, NS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch);
  hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, WGM, WGN, FM, FN, NS>), grid, dim3(WGM * WGN * 64), LDS, stream, p);
  return hipGetLastError();
}

template <typename T, typename OutT>
hipError_t launch_t(const whk::GemmArgs& a, int batch, hipStream_t stream) {
  static const int force = [] { const char* e = getenv("WH_GEMM_TILE"); return e ? atoi(e) : 0; }();   // 128 / 256: developer override
  // tools/probe_gemm on MI355X (M = 12000): 256x256 is 12-17 % faster on every encoder shape (e.g. N=2560 K=1280:
  // 648 vs 548 TFLOP/s; N=1280 K=5120: 949 vs 809).  The K loop itself runs at ~1.2 PFLOP/s; at K = 1280 half of a
  // launch is fixed cost (first-tile latency, fp32 residual read-modify-write epilogue, exact-erf GELU +13 %).
  const bool big = force ? force == 256 : (a.M >= 1024 && a.N >= 1024);
  if (force == 2563) return launch_shape<T, OutT, 4, 2, 4, 4, 3>(a, batch, stream);   // 256x128, 3-deep ring (experiment)
  if (force == 1284) return launch_shape<T, OutT, 2, 2, 4, 4, 4>(a, batch, stream);   // 128x128, 4-deep ring (experiment)
  if (force == 1283) return launch_shape<T, OutT, 2, 2, 4, 4, 3>(a, batch, stream);
  if (force == 25616) return launch_shape<T, OutT, 4, 4, 4, 4, 2>(a, batch, stream);  // 256x256, 16 waves of 64x64
  if (force == 25612) return launch_shape<T, OutT, 2, 4, 8, 2, 2>(a, batch, stream);  // 256x128, 8 waves of 128x32
  if (force == 1288) return launch_shape<T, OutT, 4, 2, 2, 4, 2>(a, batch, stream);   // 128x128, 8 waves of 32x64
  if (force == 12816) return launch_shape<T, OutT, 4, 4, 2, 2, 2>(a, batch, stream);  // 128x128, 16 waves of 32x32
  if (force == 2568) return launch_shape<T, OutT, 2, 4, 8, 4, 2>(a, batch, stream);   // 256x256, 8 waves of 128x64
  if (big) return launch_shape<T, OutT, 4, 4, 4, 4, 2>(a, batch, stream);
  return launch_shape<T, OutT, 2, 2, 4, 4, 2>(a, batch, stream);
}

}  // namespace

namespace whk {

hipError_t launch_gemm(const GemmArgs& a, int dtype, int out_f32, int batch, hipStream_t stream) {
  const int bke = dtype == 1 ? 64 : 32;
  if (a.K % bke != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (dtype == 1) {
    return out_f32 ? launch_t<half_t, float>(a, batch, stream) : launch_t<half_t, half_t>(a, batch, stream);
  }
  return launch_t<float, float>(a, batch, stream);
}

}  // namespace whk
