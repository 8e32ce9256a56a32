// common.h — shared device helpers for the gfx950 kernels (wave64, MFMA, LDS swizzle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
// The decode step's weight stream: every byte is read once per step by one CU -> non-temporal.  -DWH_PLAIN_WEIGHT_LOADS builds
// the A/B variant with the default cache policy (tools/README.md; profiles/r03_probe_weight_policy.txt).
#ifdef WH_PLAIN_WEIGHT_LOADS
#define WH_WEIGHT_LOAD(p) (*(p))
#else
#define WH_WEIGHT_LOAD(p) __builtin_nontemporal_load(p)
#endif
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint2v __attribute__((ext_vector_type(2)));

#define WH_WAVE 64

// ---------------------------------------------------------------------------------------------
// element traits: T in {float, half_t}; a "unit" is 16 bytes of T (4 floats / 8 halves)
// ---------------------------------------------------------------------------------------------
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int UNIT = 4;   // elements per 16-byte unit
  typedef float4v unit_t;
};
template <> struct ET<half_t> {
  static constexpr int UNIT = 8;
  typedef half8v unit_t;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(half_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ half_t from_f32<half_t>(float x) { return (half_t)x; }

// dot of one 16-byte unit of T with another, accumulated in fp32
__device__ __forceinline__ float dot_unit(float4v a, float4v b, float acc) {
  acc = __builtin_fmaf(a[0], b[0], acc);
  acc = __builtin_fmaf(a[1], b[1], acc);
  acc = __builtin_fmaf(a[2], b[2], acc);
  acc = __builtin_fmaf(a[3], b[3], acc);
  return acc;
}
__device__ __forceinline__ float dot_unit(half8v a, half8v b, float acc) {
  // v_dot2_f32_f16: two fp16 products accumulated in fp32
  acc = __builtin_amdgcn_fdot2(half2v{a[0], a[1]}, half2v{b[0], b[1]}, acc, false);
  acc = __builtin_amdgcn_fdot2(half2v{a[2], a[3]}, half2v{b[2], b[3]}, acc, false);
  acc = __builtin_amdgcn_fdot2(half2v{a[4], a[5]}, half2v{b[4], b[5]}, acc, false);
  acc = __builtin_amdgcn_fdot2(half2v{a[6], a[7]}, half2v{b[6], b[7]}, acc, false);
  return acc;
}

// exact-erf GELU (nn.GELU() default, whisper/model.py:156,193)
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// The same function for values that are rounded to fp16 right afterwards: x * Phi(x) with erfc from the five-term
// Abramowitz-Stegun 7.1.26 form (|error| < 1.5e-7 in erf, 4.2e-7 in the result over [-12, 12]; an fp16 ulp at 1 is
// 9.8e-4).  Branch-free: one v_rcp, one v_exp and a dozen FMAs instead of libm's erff (two polynomial branches, both
// executed by a divergent wave) — the fc1 epilogue of the encoder went from 13 us to 4 us per launch.
__device__ __forceinline__ float gelu_erf_f16out(float x) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  q = __builtin_fmaf(t, q, 1.421413741f);
  q = __builtin_fmaf(t, q, -0.284496736f);
  q = __builtin_fmaf(t, q, 0.254829592f);
  const float half_erfc = 0.5f * t * q * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);   // erfc(|x|/sqrt2) / 2
  return x * (x >= 0.f ? 1.0f - half_erfc : half_erfc);
}
// element index of x[r][k] in a FRAGMENT-ORDER activation of K columns (kernels.h, GemvArgs::x_frag): unit (row tile, K block) = 1 KB,
// lane 16 ((k & 31) >> 3) + 8 ((k >> 5) & 1) + r % 8 holds the 8 elements [k & ~7, +8)
__device__ __forceinline__ int64_t frag_elem(int64_t r, int k, int K) {
  return ((((r >> 3) * (K >> 6) + (k >> 6)) * 64 + 16 * ((k & 31) >> 3) + 8 * ((k >> 5) & 1) + (r & 7)) << 3) + (k & 7);
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<half_t>(float x) { return gelu_erf_f16out(x); }

// ---------------------------------------------------------------------------------------------
// cross-lane exchange without LDS.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip of
// ~100 cycles per step); a latency-bound decode kernel cannot afford 6 of those per reduction.  On
// gfx950 every xor distance has a VALU form:
//   xor 1, 2   DPP quad_perm            xor 8   DPP row_ror:8 (rotation by half a 16-lane row)
//   xor 16     v_permlane16_swap        xor 32  v_permlane32_swap
//   xor 4      has no exact DPP form on gfx9-class DPP; row_half_mirror (lane i <-> 7-i of each 8) is
//              used instead, which is a valid butterfly step only when the lanes of each quad already
//              agree (i.e. right after the xor 1, xor 2 steps of a reduction) — group8_sum / wave_sum.
// The xor 8 / 16 / 32 steps are exact exchanges: across_groups*_sum may be applied to per-lane-distinct
// data (the decode attention reduces 8 different head dims per key group this way).
// ---------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }     // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }     // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_mirror8(float v) { return dpp_mov<0x141>(v); }  // row_half_mirror
__device__ __forceinline__ float lane_xor8(float v) { return dpp_mov<0x128>(v); }     // row_ror:8 == xor 8 exactly
// value held by the partner row (xor 16) / partner half (xor 32), given both copies start equal
__device__ __forceinline__ void lane_swap16(float v, float& a, float& b) {
  typedef unsigned uint2s __attribute__((ext_vector_type(2)));
  const uint2s r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap32(float v, float& a, float& b) {
  typedef unsigned uint2s __attribute__((ext_vector_type(2)));
  const uint2s r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

// sum / max over groups of 8, 16 or 64 lanes; every lane of the group ends with the result
__device__ __forceinline__ float group8_sum(float v) {
  v += lane_xor1(v); v += lane_xor2(v); v += lane_mirror8(v);
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v = group8_sum(v); v += lane_xor8(v);
  return v;
}
// reduce across the 8 (or 4) groups of a wave once each group already agrees internally
__device__ __forceinline__ float across_groups8_sum(float v) {   // groups of 8 lanes -> wave
  float a, b;
  v += lane_xor8(v);
  lane_swap16(v, a, b); v = a + b;
  lane_swap32(v, a, b); v = a + b;
  return v;
}
__device__ __forceinline__ float across_groups16_sum(float v) {  // groups of 16 lanes -> wave
  float a, b;
  lane_swap16(v, a, b); v = a + b;
  lane_swap32(v, a, b); v = a + b;
  return v;
}
__device__ __forceinline__ float across_groups8_max(float v) {
  float a, b;
  v = fmaxf(v, lane_xor8(v));
  lane_swap16(v, a, b); v = fmaxf(a, b);
  lane_swap32(v, a, b); v = fmaxf(a, b);
  return v;
}
__device__ __forceinline__ float across_groups16_max(float v) {
  float a, b;
  lane_swap16(v, a, b); v = fmaxf(a, b);
  lane_swap32(v, a, b); v = fmaxf(a, b);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return across_groups8_sum(group8_sum(v)); }
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, lane_xor1(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, lane_mirror8(v));
  return across_groups8_max(v);
}

// ---------------------------------------------------------------------------------------------
// LDS tile swizzle for row-major tiles whose rows are 128 bytes = 8 units of 16 bytes:
// unit u of row r lives at unit slot u ^ ((r >> 1) & 7).  Conflict-free for ds_read_b128 when a
// wave reads one unit column of 16 or 32 consecutive rows (MFMA A/B fragments).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_unit(int row, int unit) { return unit ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_byte(int row, int unit) { return row * 128 + (swz_unit(row, unit) << 4); }

// ordered-int encoding of floats so that atomicMax on int32 orders like float
__device__ __forceinline__ int float_to_ordered(float f) {
  int i = __float_as_int(f);
  return (i >= 0) ? i : (i ^ 0x7fffffff);
}
__device__ __forceinline__ float ordered_to_float(int i) {
  return __int_as_float((i >= 0) ? i : (i ^ 0x7fffffff));
}

#define WH_NEG_INF (-__builtin_huge_valf())

// Request the whole kernel-argument struct at kernel entry.  The compiler otherwise sinks the s_load of an argument
// into the block that first uses it, and every such block costs one more serialized round trip to the (cold)
// kernarg segment — ~0.2-0.5 us each on a launch chain of 5-12 us kernels.  Naming every dword as an SGPR input of
// an empty asm in the entry block keeps all the loads there, behind a single s_waitcnt.
// A workgroup-uniform int written by an EARLIER kernel (the position counter, a row's lag).  Read at agent scope, i.e.
// from L2: a scalar load through the constant address space is faster, but the scalar cache is not invalidated between
// back-to-back launches of one stream — a kernel that s_loaded the counter before it was bumped leaves a stale line
// behind (seen on the GPU: the first beam-search update read position 0 after the prefill's add).  Independent loads
// still issue back to back; the value lands in an SGPR.
__device__ __forceinline__ int load_agent_int(const int* p) {       // per-lane value; issue several, then uniform()
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int load_uniform_int(const int* p) { return uniform(load_agent_int(p)); }

template <typename A>
__device__ __forceinline__ void pin_kernargs(const A& a) {
  constexpr int N = sizeof(A) / 4;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&a);
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" ::"s"(w[i]));
}

