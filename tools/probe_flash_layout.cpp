// tools/probe_flash_layout.cpp — the encoder flash-attention kernel with the V^T tile in P order (product) against the
// plain tile layout (launch mode bit 1), bit for bit, on small and ragged key counts and both scale modes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/probe_flash_layout.cpp -o tools/probe_flash_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../whisper_amd/csrc/attention.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  const int H = 3, B = 2, D = H * 64;
  const int Ts[] = {1, 20, 63, 64, 65, 100, 128, 333, 1500};
  hipStream_t st; CK(hipStreamCreate(&st));
  int bad = 0;
  for (int T : Ts) {
    const int VL = (T + 63) / 64 * 64;
    std::vector<half_t> hq((size_t)B * T * 2 * D), hv((size_t)B * D * VL);
    uint32_t s = 12345u + T;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& x : hq) x = (half_t)(rnd() * 2.0f);
    for (auto& x : hv) x = (half_t)rnd();
    half_t *q, *v, *o;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&v, hv.size() * 2)); CK(hipMalloc(&o, (size_t)B * T * D * 2));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    std::vector<half_t> out[4];
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipMemset(o, 0xff, (size_t)B * T * D * 2));
      CK(whk::launch_attn_flash_f16(q, 2 * D, (int64_t)T * 2 * D, q + D, 2 * D, (int64_t)T * 2 * D, v, VL, (int64_t)D * VL, o, D,
                                    (int64_t)T * D, B, H, T, mode, st));
      CK(hipStreamSynchronize(st));
      out[mode].resize((size_t)B * T * D);
      CK(hipMemcpy(out[mode].data(), o, out[mode].size() * 2, hipMemcpyDeviceToHost));
    }
    for (int pre = 0; pre < 2; ++pre) {
      size_t diff = 0, nan = 0;
      for (size_t i = 0; i < out[pre].size(); ++i) {
        const float a = (float)out[pre][i];
        if (a != a) ++nan;
        if (memcmp(&out[pre][i], &out[pre + 2][i], 2) != 0) ++diff;
      }
      printf("T=%4d %s: %zu of %zu outputs differ between the two tile layouts, %zu NaN\n", T, pre ? "prescaled" : "unscaled ", diff,
             out[pre].size(), nan);
      bad += diff != 0 || nan != 0;
    }
    CK(hipFree(q)); CK(hipFree(v)); CK(hipFree(o));
  }
  printf(bad ? "MISMATCH\n" : "all equal\n");
  return bad ? 2 : 0;
}
