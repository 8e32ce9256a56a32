// mel.hip — fused log-mel front end for gfx950.
//
// Replaces whisper/audio.py:110-157 (log_mel_spectrogram): reflect-pad(200) -> periodic Hann(400) ->
// 400-point real DFT at hop 160 (torch.stft, center=True, last frame dropped audio.py:149) -> power
// -> mel filterbank (audio.py:151-152) -> clamp(1e-10).log10() -> max(x, global_max - 8) -> (x+4)/4.
//
// One workgroup produces FT consecutive frames of one clip: the windowed, folded frames live in LDS
// (see mel_kernel), each lane owns DFT bin pairs (k, 200-k) and walks the 201 folded taps with a
// twiddle rotated in registers (table of cos/sin(2*pi*i/400) in LDS for the exact re-reads), the
// power spectrum is staged through LDS and the banded (<= 14 taps) mel filter is applied from there.  HBM traffic is the algorithmic minimum: each
// audio sample is read ~1.3x (frame overlap inside the block), each output written once, plus one
// 4-byte atomic per block for the global max.  A second tiny kernel applies the global clamp.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NBIN = 201;
constexpr int FT = 16;           // frames per workgroup (eight per wave)
constexpr int MEL_THREADS = 128;  // two waves, eight frames each

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

// Round 3.  The DFT of a REAL 400-sample frame is folded twice before any multiply:
//   taps j and 400-j share cos and have opposite sin            -> e[j] = x[j] + x[400-j],  d[j] = x[j] - x[400-j], j <= 200
//   bins k and 200-k share |cos| and |sin|, sign (-1)^j         -> even and odd taps are accumulated separately and
//                                                                  X[k] = E + O,  X[200-k] = E - O
// so one lane produces bins k and 200-k of eight frames from 201 folded taps: a quarter of the multiply-adds of the plain sum
// (round 2: one bin per thread over 400 taps, ten LDS reads per tap, 273 us per 8 clips).  The folded frames are stored
// tap-major (`ev[tap][frame]`: the 8 frames of a tap are two broadcast ds_read_b128), and because those broadcast reads are
// what the loop is bound by, ONE wave covers all 101 bin pairs of its eight frames (lane t: pairs t and t + 51) so every
// read feeds 32 packed multiply-adds.  The twiddles advance by a rotation in registers, re-read exactly from the table
// every RESYNC taps (fp32 drift over 16 rotations ~1e-6 relative, far inside the 1e-4 log-mel bar).
__global__ __launch_bounds__(MEL_THREADS) void mel_kernel(
    const float* __restrict__ audio, int64_t n_samples, int n_frames, int n_mels,
    const float* __restrict__ filters, const float* __restrict__ tables,  // cos[400] sin[400] win[400]
    float* __restrict__ out, int* __restrict__ scratch) {
  constexpr int RESYNC = 16;
  constexpr int HALF = NFFT / 2;                                   // 200
  constexpr int NTAP = (HALF + 1 + RESYNC - 1) / RESYNC * RESYNC;  // 208: folded taps 0..200, zero padded
  constexpr int FW = 8;                                            // frames per wave
  constexpr int NP2 = 51;                                          // lane t owns bin pairs t and t + 51 (101 pairs)
  static_assert(FT == FW * (MEL_THREADS / 64), "one wave per eight frames");
  static_assert(2 * NTAP >= NBIN, "power spectrum reuses the folded-frame buffer");
  __shared__ float tw_cos[NFFT];
  __shared__ float tw_sin[NFFT];
  __shared__ __attribute__((aligned(16))) float fold[2 * NTAP * FT];
  __shared__ float red[MEL_THREADS / 64];
  float (*ev)[FT] = reinterpret_cast<float (*)[FT]>(fold);
  float (*od)[FT] = reinterpret_cast<float (*)[FT]>(fold + NTAP * FT);
  float (*pw)[FT] = reinterpret_cast<float (*)[FT]>(fold);               // [bin][frame], after the DFT

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  const float* a = audio + (int64_t)b * n_samples;

  for (int i = tid; i < NFFT; i += MEL_THREADS) {
    tw_cos[i] = tables[i];
    tw_sin[i] = tables[NFFT + i];
  }
  // Folded taps: thread -> tap j (and j + 128), all FT frames of it in flight at once (2 x FT independent loads per tap)
  for (int j = tid; j < NTAP; j += MEL_THREADS) {
    float e[FT], d[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) { e[f] = 0.f; d[f] = 0.f; }
    if (j <= HALF) {
      const bool single = j == 0 || j == HALF;
      const int jm = single ? j : NFFT - j;
      const float w1 = tables[2 * NFFT + j], w2 = single ? 0.f : tables[2 * NFFT + jm];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        if (f0 + f < n_frames) {
          int64_t s1 = (int64_t)(f0 + f) * HOP + j - HALF, s2 = (int64_t)(f0 + f) * HOP + jm - HALF;
          if (s1 < 0) s1 = -s1;                                      // torch.stft(center=True, pad_mode="reflect")
          if (s1 >= n_samples) s1 = 2 * (n_samples - 1) - s1;
          if (s2 < 0) s2 = -s2;
          if (s2 >= n_samples) s2 = 2 * (n_samples - 1) - s2;
          const float x1 = a[s1] * w1, x2 = a[s2] * w2;
          e[f] = x1 + x2;
          d[f] = single ? 0.f : x1 - x2;
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FT; f += 4) {
      *(float4v*)&ev[j][f] = float4v{e[f], e[f + 1], e[f + 2], e[f + 3]};
      *(float4v*)&od[j][f] = float4v{d[f], d[f + 1], d[f + 2], d[f + 3]};
    }
  }
  // Mel band of this thread's filter (thread -> mel in the last phase): taps lo..hi, at most MAXBAND of them, requested now
  constexpr int MAXBAND = 16;
  float wband[MAXBAND];
  int band_lo = 0;
  {
    const int m = tid < n_mels ? tid : 0;
    band_lo = scratch[SCR_LO + m];
    const int hi = scratch[SCR_HI + m];
#pragma unroll
    for (int t = 0; t < MAXBAND; ++t) wband[t] = (tid < n_mels && band_lo + t <= hi) ? filters[m * NBIN + band_lo + t] : 0.f;
  }
  __syncthreads();

  const int lane = tid & 63, fw = (tid >> 6) * FW;                 // this wave's first frame inside the workgroup
  const int k1 = lane < NP2 ? lane : 0, k2 = lane + NP2 <= HALF / 2 ? lane + NP2 : 0;
  float2v re1[2][FW / 2], im1[2][FW / 2], re2[2][FW / 2], im2[2][FW / 2];   // [tap parity][frame pair]
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int f = 0; f < FW / 2; ++f) {
      re1[p][f] = float2v{0.f, 0.f}; im1[p][f] = float2v{0.f, 0.f};
      re2[p][f] = float2v{0.f, 0.f}; im2[p][f] = float2v{0.f, 0.f};
    }
  {
    const float ct1 = tw_cos[k1], st1 = tw_sin[k1], ct2 = tw_cos[k2], st2 = tw_sin[k2];  // rotation by one tap
    int idx1 = 0, idx2 = 0;                                        // (k * j0) mod 400
    for (int j0 = 0; j0 < NTAP; j0 += RESYNC) {
      float c1 = tw_cos[idx1], s1 = tw_sin[idx1], c2 = tw_cos[idx2], s2 = tw_sin[idx2];
#pragma unroll
      for (int jj = 0; jj < RESYNC; ++jj) {
        const float4v e0 = *(const float4v*)&ev[j0 + jj][fw];
        const float4v e1 = *(const float4v*)&ev[j0 + jj][fw + 4];
        const float4v d0 = *(const float4v*)&od[j0 + jj][fw];
        const float4v d1 = *(const float4v*)&od[j0 + jj][fw + 4];
        const float2v ea = {e0[0], e0[1]}, eb = {e0[2], e0[3]}, ec = {e1[0], e1[1]}, ed = {e1[2], e1[3]};
        const float2v da = {d0[0], d0[1]}, db = {d0[2], d0[3]}, dc = {d1[0], d1[1]}, dd = {d1[2], d1[3]};
        {
          const float2v cc = {c1, c1}, ss = {s1, s1};
          float2v* r = re1[jj & 1];                                // j0 is even: tap parity = jj & 1, a compile-time constant
          float2v* m = im1[jj & 1];
          r[0] = __builtin_elementwise_fma(cc, ea, r[0]); r[1] = __builtin_elementwise_fma(cc, eb, r[1]);
          r[2] = __builtin_elementwise_fma(cc, ec, r[2]); r[3] = __builtin_elementwise_fma(cc, ed, r[3]);
          m[0] = __builtin_elementwise_fma(ss, da, m[0]); m[1] = __builtin_elementwise_fma(ss, db, m[1]);
          m[2] = __builtin_elementwise_fma(ss, dc, m[2]); m[3] = __builtin_elementwise_fma(ss, dd, m[3]);
        }
        {
          const float2v cc = {c2, c2}, ss = {s2, s2};
          float2v* r = re2[jj & 1];
          float2v* m = im2[jj & 1];
          r[0] = __builtin_elementwise_fma(cc, ea, r[0]); r[1] = __builtin_elementwise_fma(cc, eb, r[1]);
          r[2] = __builtin_elementwise_fma(cc, ec, r[2]); r[3] = __builtin_elementwise_fma(cc, ed, r[3]);
          m[0] = __builtin_elementwise_fma(ss, da, m[0]); m[1] = __builtin_elementwise_fma(ss, db, m[1]);
          m[2] = __builtin_elementwise_fma(ss, dc, m[2]); m[3] = __builtin_elementwise_fma(ss, dd, m[3]);
        }
        const float cn1 = __builtin_fmaf(c1, ct1, -(s1 * st1));
        s1 = __builtin_fmaf(s1, ct1, c1 * st1);
        c1 = cn1;
        const float cn2 = __builtin_fmaf(c2, ct2, -(s2 * st2));
        s2 = __builtin_fmaf(s2, ct2, c2 * st2);
        c2 = cn2;
      }
      idx1 += (k1 * RESYNC) % NFFT;
      if (idx1 >= NFFT) idx1 -= NFFT;
      idx2 += (k2 * RESYNC) % NFFT;
      if (idx2 >= NFFT) idx2 -= NFFT;
    }
  }
  __syncthreads();                                                 // every wave is done reading ev/od: pw may overwrite them
  auto spill = [&](float2v (*re)[FW / 2], float2v (*im)[FW / 2], int k) {
#pragma unroll
    for (int f = 0; f < FW / 2; ++f) {
      const float2v rk = re[0][f] + re[1][f], ik = im[0][f] + im[1][f];    // bin k
      const float2v rm = re[0][f] - re[1][f], imm = im[0][f] - im[1][f];   // bin 200 - k
      const float2v pk = rk * rk + ik * ik, pm = rm * rm + imm * imm;
      *(float2v*)&pw[k][fw + 2 * f] = pk;
      *(float2v*)&pw[HALF - k][fw + 2 * f] = pm;
    }
  };
  if (lane < NP2) spill(re1, im1, lane);
  if (lane + NP2 <= HALF / 2) spill(re2, im2, lane + NP2);
  __syncthreads();

  float lmax = WH_NEG_INF;
  for (int m = tid; m < n_mels; m += MEL_THREADS) {                // n_mels <= 128: one pass, weights already in registers
    float acc[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) acc[f] = 0.f;
    if (m == tid) {
#pragma unroll
      for (int t = 0; t < MAXBAND; ++t) {
        const int k = band_lo + t < NBIN ? band_lo + t : NBIN - 1;   // weight 0 beyond the band
#pragma unroll
        for (int f = 0; f < FT; f += 4) {
          const float4v p = *(const float4v*)&pw[k][f];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[f + q] = __builtin_fmaf(wband[t], p[q], acc[f + q]);
        }
      }
      for (int k = band_lo + MAXBAND; k <= scratch[SCR_HI + m]; ++k) {   // wider bands than MAXBAND (not the case for 80/128 mels)
        const float w = filters[m * NBIN + k];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = __builtin_fmaf(w, pw[k][f], acc[f]);
      }
    } else {
      for (int k = scratch[SCR_LO + m]; k <= scratch[SCR_HI + m]; ++k) {
        const float w = filters[m * NBIN + k];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = __builtin_fmaf(w, pw[k][f], acc[f]);
      }
    }
    float* o = out + ((int64_t)b * n_mels + m) * n_frames + f0;
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      if (f0 + f < n_frames) {
        const float v = log10f(fmaxf(acc[f], 1e-10f));
        o[f] = v;
        lmax = fmaxf(lmax, v);
      }
    }
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
