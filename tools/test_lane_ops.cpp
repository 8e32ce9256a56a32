// developer check of the DPP / permlane reductions in common.h against __shfl_xor butterflies
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../whisper_amd/csrc/common.h"

__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  const float v = in[l];
  float r8 = v, r16 = v, r64 = v, m64 = v;
  for (int o = 1; o < 8; o <<= 1) r8 += __shfl_xor(r8, o, 64);
  for (int o = 1; o < 16; o <<= 1) r16 += __shfl_xor(r16, o, 64);
  for (int o = 1; o < 64; o <<= 1) r64 += __shfl_xor(r64, o, 64);
  for (int o = 1; o < 64; o <<= 1) m64 = fmaxf(m64, __shfl_xor(m64, o, 64));
  out[l] = group8_sum(v);            out[64 + l] = r8;
  out[128 + l] = group16_sum(v);     out[192 + l] = r16;
  out[256 + l] = wave_sum(v);        out[320 + l] = r64;
  out[384 + l] = wave_max(v);        out[448 + l] = m64;
  out[512 + l] = across_groups8_sum(group8_sum(v));   out[576 + l] = r64;
  out[640 + l] = across_groups16_sum(group16_sum(v)); out[704 + l] = r64;
  float g8m = v; for (int o = 1; o < 8; o <<= 1) g8m = fmaxf(g8m, __shfl_xor(g8m, o, 64));
  out[768 + l] = across_groups8_max(g8m);             out[832 + l] = m64;
  float g16m = v; for (int o = 1; o < 16; o <<= 1) g16m = fmaxf(g16m, __shfl_xor(g16m, o, 64));
  out[896 + l] = across_groups16_max(g16m);           out[960 + l] = m64;
}

int main() {
  float h[64], o[1024];
  for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 3.f + i * 0.01f;
  float *d, *e; hipMalloc(&d, 256); hipMalloc(&e, 4096);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, 4096, hipMemcpyDeviceToHost);
  const char* names[] = {"group8_sum", "group16_sum", "wave_sum", "wave_max", "across8_sum", "across16_sum", "across8_max", "across16_max"};
  int bad = 0;
  for (int t = 0; t < 8; ++t) {
    float md = 0;
    for (int l = 0; l < 64; ++l) md = fmaxf(md, fabsf(o[t * 128 + l] - o[t * 128 + 64 + l]));
    printf("%-14s max|diff| = %g %s\n", names[t], md, md < 1e-4 ? "ok" : "MISMATCH");
    if (md >= 1e-4) { bad++; for (int l = 0; l < 64; l += 7) printf("   lane %d: got %g want %g\n", l, o[t * 128 + l], o[t * 128 + 64 + l]); }
  }
  return bad;
}
