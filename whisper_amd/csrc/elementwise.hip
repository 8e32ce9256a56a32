// elementwise.hip — small HBM-bound helpers around the GEMM / attention kernels.
//   mel_transpose : (B, n_mels, F) -> (B, F+2, n_mels) so that conv1/conv2 (whisper/model.py:193-194)
//                   become GEMMs over overlapping, contiguous rows
//   layernorm     : LayerNorm.forward, whisper/model.py:39-41 (fp32 statistics, eps 1e-5)
//   embed         : token_embedding(x) + positional_embedding[offset:offset+T], model.py:235-238
//   scatter_kv    : KV-cache append for the prefill (the torch.cat of model.py:332, done in place)
//   gather_cache  : PyTorchInference.rearrange_kv_cache, whisper/decoding.py:172-176
#include "common.h"
#include "kernels.h"

namespace {

// ---- mel transpose -------------------------------------------------------------------------
template <typename TIn, typename T>
__global__ void mel_transpose_kernel(const TIn* __restrict__ mel, int n_mels, int F, T* __restrict__ out) {
  __shared__ float tile[32][129];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * 32;
  const TIn* src = mel + (int64_t)b * n_mels * F;
  T* dst = out + (int64_t)b * (F + 2) * n_mels;
  // load: threads along frames (contiguous in the input)
  for (int i = threadIdx.x; i < n_mels * 32; i += blockDim.x) {
    const int m = i >> 5, f = i & 31;
    float v = 0.f;
    if (f0 + f < F) v = to_f32(src[(int64_t)m * F + f0 + f]);
    tile[f][m] = v;
  }
  __syncthreads();
  // store: threads along mels (contiguous in the output); frame f lands in padded row f+1
  for (int i = threadIdx.x; i < n_mels * 32; i += blockDim.x) {
    const int f = i / n_mels, m = i - f * n_mels;
    if (f0 + f < F) dst[(int64_t)(f0 + f + 1) * n_mels + m] = from_f32<T>(tile[f][m]);
  }
  if (blockIdx.x == 0) {
    for (int m = threadIdx.x; m < n_mels; m += blockDim.x) {
      dst[m] = from_f32<T>(0.f);
      dst[(int64_t)(F + 1) * n_mels + m] = from_f32<T>(0.f);
    }
  }
}

// ---- layernorm -----------------------------------------------------------------------------
// one wave per row; NC = float4 chunks per lane (compile-time so the row stays in registers)
template <typename T, int NC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ out,
                                                        int64_t ldo, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const int nv = D >> 2;
  float4v v[NC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < nv) ? *(const float4v*)(xr + c * 4) : float4v{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  s = wave_sum(s);
  const float mean = s / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; ss = __builtin_fmaf(d, d, ss); }
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)D + 1e-5f);
  T* orow = out + row * ldo;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4v wv = *(const float4v*)(w + c * 4);
      const float4v bv = *(const float4v*)(b + c * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) orow[c * 4 + e] = from_f32<T>((v[i][e] - mean) * rstd * wv[e] + bv[e]);
    }
  }
}

template <typename T>
void layernorm_dispatch(const float* x, int64_t ldx, const float* w, const float* b, T* out, int64_t ldo,
                        int64_t rows, int D, hipStream_t stream) {
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nc = (D / 4 + 63) / 64;
  switch (nc) {
#define LN_CASE(N) case N: hipLaunchKernelGGL((layernorm_kernel<T, N>), grid, block, 0, stream, x, ldx, w, b, out, ldo, rows, D); break;
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
#undef LN_CASE
    default: break;
  }
}

// ---- embedding -----------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(const int64_t* __restrict__ tokens, int64_t stride, int T0,
                             const T* __restrict__ emb, const float* __restrict__ pos,
                             const int* __restrict__ d_offset, const int* __restrict__ lag, int D, int n_vocab,
                             float* __restrict__ x) {
  const int row = blockIdx.x;           // r*T0 + t
  const int r = row / T0, t = row - r * T0;
  int64_t tok = tokens[(int64_t)r * stride + t];
  if (tok < 0) tok = 0;
  if (tok >= n_vocab) tok = n_vocab - 1;
  const int vo = load_agent_int(d_offset), vl = load_agent_int(lag ? lag + r : d_offset);
  const int p = uniform(vo) - (lag ? uniform(vl) : 0) + t;
  const T* e = emb + tok * D;
  const float* pp = pos + (int64_t)p * D;
  float* xr = x + (int64_t)row * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) xr[d] = to_f32(e[d]) + pp[d];
}

// ---- prefill KV scatter ----------------------------------------------------------------------
template <typename T>
__global__ void scatter_kv_kernel(const T* __restrict__ qkv, int T0, int D, const int* __restrict__ d_offset,
                                  int n_ctx, T* __restrict__ kc, T* __restrict__ vc) {
  typedef typename ET<T>::unit_t unit_t;
  constexpr int UNIT = ET<T>::UNIT;
  const int row = blockIdx.x;
  const int r = row / T0, t = row - r * T0;
  const int p = *d_offset + t;
  const T* src = qkv + (int64_t)row * 3 * D;
  T* kd = kc + ((int64_t)r * n_ctx + p) * D;
  T* vd = vc + ((int64_t)r * n_ctx + p) * D;
  for (int u = threadIdx.x; u < D / UNIT; u += blockDim.x) {
    *(unit_t*)(kd + u * UNIT) = *(const unit_t*)(src + D + u * UNIT);
    *(unit_t*)(vd + u * UNIT) = *(const unit_t*)(src + 2 * D + u * UNIT);
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ sel, int D,
                                   float* __restrict__ out) {
  const int i = blockIdx.x;
  const float* s = x + (int64_t)sel[i] * D;
  float* d = out + (int64_t)i * D;
  for (int k = threadIdx.x; k < D; k += blockDim.x) d[k] = s[k];
}

// dst row i (row_bytes apart) <- first used_bytes of src row src_idx[i]; 16-byte units
__global__ void gather_cache_kernel(const uint4v* __restrict__ src, uint4v* __restrict__ dst,
                                    const int* __restrict__ src_idx, int64_t row_units, int64_t used_units) {
  const int i = blockIdx.y;
  const uint4v* s = src + (int64_t)src_idx[i] * row_units;
  uint4v* d = dst + (int64_t)i * row_units;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < used_units;
       u += (int64_t)gridDim.x * blockDim.x)
    d[u] = s[u];
}

// Beam reorder in place, ONE launch for every layer and both caches: new row i of an audio's beam group takes the
// cache of old row src_idx[i] of the SAME group (BeamSearchDecoder never mixes audios, decoding.py:323-382).
// A thread owns one 16-byte column of all G rows of its (cache, audio): it reads the G units, then writes them back
// permuted — no staging buffer and no copy-back, because nobody else touches that column.
template <int GMAX>
__global__ __launch_bounds__(256) void permute_group_kernel(uint4v* __restrict__ k_base, uint4v* __restrict__ v_base,
                                                            int64_t layer_units, int64_t row_units, int64_t used_units,
                                                            const int* __restrict__ src_idx, int G,
                                                            const int* __restrict__ copy_from, int pos_units) {
  const int audio = blockIdx.y, lz = blockIdx.z;
  uint4v* base = ((lz & 1) ? v_base : k_base) + (int64_t)(lz >> 1) * layer_units + (int64_t)audio * G * row_units;
  int src[GMAX];
  int64_t first[GMAX];                      // first 16-byte unit row g has to take from its source
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    src[g] = g < G ? src_idx[audio * G + g] - audio * G : 0;
    first[g] = (g < G && copy_from) ? (int64_t)copy_from[audio * G + g] * pos_units : 0;
  }
  int64_t begin = used_units;
#pragma unroll
  for (int g = 0; g < GMAX; ++g)
    if (g < G && src[g] != g && first[g] < begin) begin = first[g];
  if (begin >= used_units) return;          // nothing moves in this segment (uniform: decided from the index arrays alone)
  begin = begin / 256 * 256;
  for (int64_t u = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; u < used_units; u += (int64_t)gridDim.x * 256) {
    uint4v val[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
      if (g < G) val[g] = base[(int64_t)g * row_units + u];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < G) {
        uint4v v = val[0];
#pragma unroll
        for (int j = 1; j < GMAX; ++j) v = (src[g] == j) ? val[j] : v;
        if (src[g] != g && u >= first[g]) base[(int64_t)g * row_units + u] = v;
      }
    }
  }
}

__global__ void add_int_kernel(int* p, int v) { *p += v; }

// the first `used_words` 4-byte words of row `src_row` -> rows [dst_row0, dst_row0 + G) of every layer slab
// (grid.y = layer): prompt positions / prefill logits of a segment's leader row to the rows of its beam group
__global__ __launch_bounds__(256) void replicate_row_kernel(uint32_t* __restrict__ base, int64_t layer_words, int64_t row_words,
                                                            int src_row, int dst_row0, int G, int64_t used_words) {
  uint32_t* slab = base + (int64_t)blockIdx.y * layer_words;
  const uint32_t* src = slab + (int64_t)src_row * row_words;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < used_words; u += (int64_t)gridDim.x * 256) {
    const uint32_t v = src[u];
    for (int g = 0; g < G; ++g)
      if (dst_row0 + g != src_row) slab[(int64_t)(dst_row0 + g) * row_words + u] = v;
  }
}

}  // namespace

namespace whk {

hipError_t launch_mel_transpose(const void* mel, int mel_is_f16, int B, int n_mels, int F, void* out,
                                int dtype, hipStream_t stream) {
  if (n_mels > 128) return hipErrorInvalidValue;
  dim3 grid((F + 31) / 32, B), block(256);
  if (dtype == 1) {
    if (mel_is_f16)
      hipLaunchKernelGGL((mel_transpose_kernel<half_t, half_t>), grid, block, 0, stream, (const half_t*)mel, n_mels, F, (half_t*)out);
    else
      hipLaunchKernelGGL((mel_transpose_kernel<float, half_t>), grid, block, 0, stream, (const float*)mel, n_mels, F, (half_t*)out);
  } else {
    if (mel_is_f16)
      hipLaunchKernelGGL((mel_transpose_kernel<half_t, float>), grid, block, 0, stream, (const half_t*)mel, n_mels, F, (float*)out);
    else
      hipLaunchKernelGGL((mel_transpose_kernel<float, float>), grid, block, 0, stream, (const float*)mel, n_mels, F, (float*)out);
  }
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, int64_t ldx, const float* w, const float* b, void* out,
                            int64_t ldo, int64_t rows, int D, int dtype, hipStream_t stream) {
  if (D % 4 != 0 || D > 2048) return hipErrorInvalidValue;
  if (dtype == 1) layernorm_dispatch<half_t>(x, ldx, w, b, (half_t*)out, ldo, rows, D, stream);
  else layernorm_dispatch<float>(x, ldx, w, b, (float*)out, ldo, rows, D, stream);
  return hipGetLastError();
}

hipError_t launch_embed(const int64_t* tokens, int64_t stride, int R, int T0, const void* tok_emb,
                        const float* pos, const int* d_offset, const int* lag, int D, int n_vocab, float* x,
                        int dtype, hipStream_t stream) {
  dim3 grid(R * T0), block(256);
  if (dtype == 1)
    hipLaunchKernelGGL((embed_kernel<half_t>), grid, block, 0, stream, tokens, stride, T0, (const half_t*)tok_emb, pos, d_offset, lag, D, n_vocab, x);
  else
    hipLaunchKernelGGL((embed_kernel<float>), grid, block, 0, stream, tokens, stride, T0, (const float*)tok_emb, pos, d_offset, lag, D, n_vocab, x);
  return hipGetLastError();
}

hipError_t launch_scatter_kv(const void* qkv, int R, int T0, int D, const int* d_offset, int n_ctx,
                             void* kcache, void* vcache, int dtype, hipStream_t stream) {
  dim3 grid(R * T0), block(128);
  if (dtype == 1)
    hipLaunchKernelGGL((scatter_kv_kernel<half_t>), grid, block, 0, stream, (const half_t*)qkv, T0, D, d_offset, n_ctx, (half_t*)kcache, (half_t*)vcache);
  else
    hipLaunchKernelGGL((scatter_kv_kernel<float>), grid, block, 0, stream, (const float*)qkv, T0, D, d_offset, n_ctx, (float*)kcache, (float*)vcache);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const float* x, const int* sel, int n_sel, int D, float* out,
                              hipStream_t stream) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_sel), dim3(256), 0, stream, x, sel, D, out);
  return hipGetLastError();
}

hipError_t launch_gather_cache(const void* src, void* dst, const int* src_idx, int R, int64_t row_bytes,
                               int64_t used_bytes, hipStream_t stream) {
  if (used_bytes <= 0) return hipSuccess;
  const int64_t used_units = used_bytes / 16, row_units = row_bytes / 16;
  int bx = (int)((used_units + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(gather_cache_kernel, dim3(bx, R), dim3(256), 0, stream, (const uint4v*)src, (uint4v*)dst,
                     src_idx, row_units, used_units);
  return hipGetLastError();
}

hipError_t launch_permute_groups(void* k_base, void* v_base, int n_layers, int64_t layer_bytes, int n_audio, int G,
                                 int64_t row_bytes, int64_t used_bytes, const int* src_idx, const int* copy_from,
                                 int64_t pos_bytes, hipStream_t stream) {
  if (used_bytes <= 0) return hipSuccess;
  if (G > 8 || (copy_from && (pos_bytes <= 0 || pos_bytes % 16))) return hipErrorInvalidValue;
  const int64_t used_units = used_bytes / 16;
  int bx = (int)((used_units + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(permute_group_kernel<8>, dim3(bx, n_audio, n_layers * 2), dim3(256), 0, stream, (uint4v*)k_base,
                     (uint4v*)v_base, layer_bytes / 16, row_bytes / 16, used_units, src_idx, G, copy_from,
                     (int)(pos_bytes / 16));
  return hipGetLastError();
}

hipError_t launch_replicate_row(void* base, int64_t layer_bytes, int n_layers, int64_t row_bytes, int src_row, int dst_row0,
                                int G, int64_t used_bytes, hipStream_t stream) {
  if (used_bytes <= 0 || G <= 0) return hipSuccess;
  if ((layer_bytes | row_bytes | used_bytes) & 3) return hipErrorInvalidValue;
  const int64_t words = used_bytes / 4;
  int bx = (int)((words + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(replicate_row_kernel, dim3(bx, n_layers), dim3(256), 0, stream, (uint32_t*)base, layer_bytes / 4,
                     row_bytes / 4, src_row, dst_row0, G, words);
  return hipGetLastError();
}

hipError_t launch_add_int(int* p, int v, hipStream_t stream) {
  hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, stream, p, v);
  return hipGetLastError();
}

}  // namespace whk
