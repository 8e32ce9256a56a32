// developer microbenchmark: issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU forms a
// fp16-weights x fp16/fp32-activations dot product can be built from on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, long long* cyc) {
  unsigned a = in[threadIdx.x], b = in[threadIdx.x + 256];
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (float)i;
  float2v pacc[8];
  for (int i = 0; i < 8; ++i) pacc[i] = float2v{(float)i, 1.f};
  float2v fa = {__uint_as_float(a), 1.f}, fb = {__uint_as_float(b), 2.f};
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < REP; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(fa), "v"(fb));
      if (MODE == 5) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 6) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(acc[i]) : "v"(a));
      if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i][0] + pacc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, const unsigned* in, float* out, long long* cyc, int blocks) {
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc);
  hipDeviceSynchronize();
  long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s %6.2f cycles per wave-instruction (1 wave/SIMD, %d blocks)\n", name, (double)h[0] / (REP * 8), blocks);
}

int main() {
  unsigned* in; float* out; long long* cyc;
  hipMalloc(&in, 4096); hipMemset(in, 0x3c, 4096); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  for (int blocks : {1, 256}) {
    run<0>("v_dot2c_f32_f16 (VOP2)", in, out, cyc, blocks);
    run<1>("v_dot2_f32_f16 (VOP3P)", in, out, cyc, blocks);
    run<2>("v_fma_mix_f32 (f16 x f16 + f32)", in, out, cyc, blocks);
    run<3>("v_fma_f32", in, out, cyc, blocks);
    run<4>("v_pk_fma_f32", in, out, cyc, blocks);
    run<5>("v_pk_fma_f16", in, out, cyc, blocks);
    run<6>("v_cvt_f32_f16", in, out, cyc, blocks);
    run<7>("v_fmac_f32", in, out, cyc, blocks);
  }
  return 0;
}
