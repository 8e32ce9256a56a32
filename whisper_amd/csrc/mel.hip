// mel.hip — fused log-mel front end for gfx950.
//
// Replaces whisper/audio.py:110-157 (log_mel_spectrogram): reflect-pad(200) -> periodic Hann(400) ->
// 400-point real DFT at hop 160 (torch.stft, center=True, last frame dropped audio.py:149) -> power
// -> mel filterbank (audio.py:151-152) -> clamp(1e-10).log10() -> max(x, global_max - 8) -> (x+4)/4.
//
// One workgroup produces FT consecutive frames of one clip: the windowed frames live in LDS, each
// thread owns one DFT bin and walks the 400 taps with an incrementally-reduced twiddle index
// (table of cos/sin(2*pi*i/400) in LDS), the power spectrum is staged through LDS and the banded
// (<= 14 taps) mel filter is applied from there.  HBM traffic is the algorithmic minimum: each
// audio sample is read ~1.3x (frame overlap inside the block), each output written once, plus one
// 4-byte atomic per block for the global max.  A second tiny kernel applies the global clamp.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NBIN = 201;
constexpr int FT = 8;            // frames per workgroup
constexpr int MEL_THREADS = 256;

// scratch layout (ints): [0] ordered-int global max, [16 .. 16+128) band lo, [144 .. 272) band hi
constexpr int SCR_LO = 16;
constexpr int SCR_HI = 16 + 128;

// first / last non-zero tap of every mel filter: one wave per filter (64 lanes x 4 taps cover the 201 bins, min / max
// across the wave by DPP) — the round-2 version scanned the 201 taps serially in one thread per filter, 32.7 us per call
__global__ __launch_bounds__(64) void mel_bands_kernel(const float* __restrict__ filters, int n_mels, int* __restrict__ scratch) {
  const int m = blockIdx.x, lane = threadIdx.x;
  if (m == 0 && lane == 0) scratch[0] = float_to_ordered(WH_NEG_INF);
  int lo = NBIN, hi = -1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = lane + 64 * e;
    if (k < NBIN && filters[m * NBIN + k] != 0.0f) {
      if (k < lo) lo = k;
      if (k > hi) hi = k;
    }
  }
  const float flo = -wave_max(-(float)lo), fhi = wave_max((float)hi);     // exact: |values| <= 201
  if (lane == 0) {
    scratch[SCR_LO + m] = (int)flo;
    scratch[SCR_HI + m] = (int)fhi;
  }
}

__global__ __launch_bounds__(MEL_THREADS) void mel_kernel(
    const float* __restrict__ audio, int64_t n_samples, int n_frames, int n_mels,
    const float* __restrict__ filters, const float* __restrict__ tables,  // cos[400] sin[400] win[400]
    float* __restrict__ out, int* __restrict__ scratch) {
  __shared__ float tw_cos[NFFT];
  __shared__ float tw_sin[NFFT];
  __shared__ float xw[FT][NFFT];
  __shared__ float pw[FT][NBIN + 3];
  __shared__ float red[MEL_THREADS / 64];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  const float* a = audio + (int64_t)b * n_samples;

  for (int i = tid; i < NFFT; i += MEL_THREADS) {
    tw_cos[i] = tables[i];
    tw_sin[i] = tables[NFFT + i];
  }
  for (int i = tid; i < FT * NFFT; i += MEL_THREADS) {
    int f = i / NFFT, j = i - f * NFFT;
    float v = 0.f;
    if (f0 + f < n_frames) {
      int64_t s = (int64_t)(f0 + f) * HOP + j - NFFT / 2;
      if (s < 0) s = -s;                                  // reflect padding of torch.stft(center=True)
      if (s >= n_samples) s = 2 * (n_samples - 1) - s;
      v = a[s] * tables[2 * NFFT + j];
    }
    xw[f][j] = v;
  }
  __syncthreads();

  if (tid < NBIN) {
    float re[FT], im[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) { re[f] = 0.f; im[f] = 0.f; }
    int idx = 0;
    for (int j = 0; j < NFFT; ++j) {
      const float c = tw_cos[idx], s = tw_sin[idx];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const float x = xw[f][j];
        re[f] = __builtin_fmaf(c, x, re[f]);
        im[f] = __builtin_fmaf(s, x, im[f]);
      }
      idx += tid;
      if (idx >= NFFT) idx -= NFFT;
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) pw[f][tid] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();

  float lmax = WH_NEG_INF;
  for (int i = tid; i < n_mels * FT; i += MEL_THREADS) {
    const int m = i / FT, f = i - m * FT;
    if (f0 + f >= n_frames) continue;
    const int lo = scratch[SCR_LO + m], hi = scratch[SCR_HI + m];
    float acc = 0.f;
    for (int k = lo; k <= hi; ++k) acc = __builtin_fmaf(filters[m * NBIN + k], pw[f][k], acc);
    const float v = log10f(fmaxf(acc, 1e-10f));
    out[((int64_t)b * n_mels + m) * n_frames + f0 + f] = v;
    lmax = fmaxf(lmax, v);
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int w = 1; w < MEL_THREADS / 64; ++w) m = fmaxf(m, red[w]);
    atomicMax(&scratch[0], float_to_ordered(m));
  }
}

__global__ void mel_finish_kernel(float* __restrict__ out, int64_t n, const int* __restrict__ scratch) {
  const float floor_v = ordered_to_float(scratch[0]) - 8.0f;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (fmaxf(out[i], floor_v) + 4.0f) / 4.0f;
}

}  // namespace

namespace whk {

hipError_t launch_log_mel(const float* audio, int64_t n_samples, int batch, int n_mels,
                          const float* filters, const float* tables, float* out, void* scratch,
                          hipStream_t stream) {
  const int n_frames = (int)(n_samples / HOP);
  int* scr = (int*)scratch;
  hipLaunchKernelGGL(mel_bands_kernel, dim3(n_mels), dim3(64), 0, stream, filters, n_mels, scr);
  dim3 grid((n_frames + FT - 1) / FT, batch);
  hipLaunchKernelGGL(mel_kernel, grid, dim3(MEL_THREADS), 0, stream, audio, n_samples, n_frames, n_mels,
                     filters, tables, out, scr);
  const int64_t n = (int64_t)batch * n_mels * n_frames;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mel_finish_kernel, dim3(blocks), dim3(256), 0, stream, out, n, scr);
  return hipGetLastError();
}

}  // namespace whk
