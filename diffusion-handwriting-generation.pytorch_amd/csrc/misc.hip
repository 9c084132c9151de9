// misc.hip — the small kernels around the GEMM/attention core: sigma MLP and the
// all-layer FiLM table, token embedding + LayerNorm, FiLM apply, input/output
// heads, and the fused scheduler step with counter-based N(0,1) noise.
#include "dhw_common.h"
#include "dhw_kernels.h"
#include "heads_core.h"

namespace {

// ---- sigma_ffn (model.py:83,134; utils/nn.py:165-175): one workgroup per sigma value
__global__ __launch_bounds__(256) void sigma_ffn_kernel(const float* sigma, const float* w1, const float* b1,
                                                         const float* w2, const float* b2, float* sig32) {
  __shared__ float hid[2048];
  __shared__ float part[8][32];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float s = silu_f(sigma[n]);
  for (int j = tid; j < 2048; j += 256) hid[j] = silu_f(w1[j] * s + b1[j]);
  __syncthreads();
  // 32 outputs x 8 partial sums over 256-wide slices of the hidden vector
  const int o = tid & 31, sl = tid >> 5;
  float acc = 0.f;
  const float* w = w2 + (size_t)o * 2048 + sl * 256;
  for (int j = 0; j < 256; ++j) acc += w[j] * hid[sl * 256 + j];
  part[sl][o] = acc;
  __syncthreads();
  if (tid < 32) {
    float t = b2[tid];
    for (int k = 0; k < 8; ++k) t += part[k][tid];
    sig32[n * 32 + tid] = t;
  }
}

// ---- FiLM table: all gamma/beta Linears(32 -> C) of the model at once (conditioning.py:16-18)
__global__ __launch_bounds__(256) void film_kernel(const float* sig32, const float* wcat, const float* bcat,
                                                    int cols, float* film) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= cols) return;
  const float* s = sig32 + n * 32;
  const float* w = wcat + (size_t)c * 32;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) acc += w[k] * s[k];
  film[(size_t)n * cols + c] = acc + bcat[c];
}

// ---- LN(emb[text]) (text_style.py:96-97): one wave per token
template <typename T>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* text, int rows, const float* emb, int dim, int n_true,
                                                        int vocab, T* out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  long id = text[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* e = emb + id * dim;
  float s = 0.f;
  for (int c = lane; c < n_true; c += 64) s += e[c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / n_true;
  float v = 0.f;
  for (int c = lane; c < n_true; c += 64) { const float d = e[c] - mean; v += d * d; }
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = rsqrtf(v / n_true + 1e-6f);
  for (int c = lane; c < dim; c += 64) out[(size_t)row * dim + c] = from_f<T>(c < n_true ? (e[c] - mean) * rstd : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void film_apply_kernel(const T* in, int in_B, int rows, int dim, const float* gam,
                                                          const float* bet, long bs, int div, T* out, long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long e = i * 4;
  const long r = e / dim;
  const int c = (int)(e - r * dim);
  const long b = r / rows;
  const f32x4 x = load4(in + ((b % in_B) * rows + (r - b * rows)) * dim + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + (b / div) * bs + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(bet + (b / div) * bs + c);
  store4(out + e, x * ga + be);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* in, long n4, T* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  store4(out + i * 4, *reinterpret_cast<const f32x4*>(in + i * 4));
}

// ---- input_dense: Linear(2 -> C) (model.py:139)
template <typename T>
__global__ __launch_bounds__(256) void input_dense_kernel(const float* strokes, long rows, const float* w,
                                                           const float* b, int C, T* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 channels of one row
  const int per_row = C / 4;
  const long r = i / per_row;
  if (r >= rows) return;
  const int c = (int)(i - r * per_row) * 4;
  const float s0 = strokes[r * 2], s1 = strokes[r * 2 + 1];
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = w[(c + k) * 2] * s0 + w[(c + k) * 2 + 1] * s1 + b[c + k];
  store4(out + r * C + c, v);
}

// ---- heads (model.py:179-182) + fused scheduler step (utils/nn.py:84-87,110-112): one wave per 4 rows
__global__ __launch_bounds__(256) void heads_kernel(const HeadsParams p) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int l15 = lane & 15;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  if (row < p.rows) {
    const float* x = p.x + row * p.C;
    for (int c = l15 * 4; c < p.C; c += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.w_out + c);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.w_out + p.C + c);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(p.w_pen + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) { a0 += v[k] * w0[k]; a1 += v[k] * w1[k]; a2 += v[k] * w2[k]; }
    }
  }
#pragma unroll
  for (int o = 8; o; o >>= 1) {
    a0 += __shfl_xor(a0, o);
    a1 += __shfl_xor(a1, o);
    a2 += __shfl_xor(a2, o);
  }
  if (row >= p.rows || l15) return;
  heads_finish(p, row, a0, a1, a2);
}

__global__ void set_seed_kernel(uint64_t* p, uint64_t seed, int64_t first) {
  p[0] = seed;
  p[1] = (uint64_t)first;
}

// iter = -1: x_T; iter = k >= 0: the draw the k-th sampler step adds (the same call heads_finish makes)
__global__ __launch_bounds__(256) void randn_init_kernel(float* xt, long rows, int L, const uint64_t* seed_ptr, int sample_off, int iter) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float z0, z1;
  normal2(seed_ptr[0], (int64_t)seed_ptr[1] + sample_off + row / L, (int)(row % L), iter, z0, z1);
  xt[row * 2] = z0;
  xt[row * 2 + 1] = z1;
}

inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

hipError_t launch_sigma_ffn(const float* sigma, int n, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* sig32, hipStream_t st) {
  hipLaunchKernelGGL(sigma_ffn_kernel, dim3(n), dim3(256), 0, st, sigma, w1, b1, w2, b2, sig32);
  return hipGetLastError();
}
hipError_t launch_film(const float* sig32, int n, const float* wcat, const float* bcat, int cols, float* film,
                       hipStream_t st) {
  hipLaunchKernelGGL(film_kernel, dim3(nblk(cols, 256), n), dim3(256), 0, st, sig32, wcat, bcat, cols, film);
  return hipGetLastError();
}
hipError_t launch_embed_ln(int prec, const int64_t* text, int rows, const float* emb, int dim, int n_true, int vocab, void* out,
                           hipStream_t st) {
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(embed_ln_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, text, rows, emb, dim, n_true, vocab, (bf16_t*)out);
  else
    hipLaunchKernelGGL(embed_ln_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, text, rows, emb, dim, n_true, vocab, (float*)out);
  return hipGetLastError();
}
hipError_t launch_film_apply(int prec, const void* in, int in_B, int B, int rows, int dim, const float* gam,
                             const float* bet, long bs, int div, void* out, hipStream_t st) {
  const long total4 = (long)B * rows * dim / 4;
  if (div < 1 || in_B < 1) return hipErrorInvalidValue;
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(film_apply_kernel<bf16_t>, dim3(nblk(total4, 256)), dim3(256), 0, st, (const bf16_t*)in, in_B, rows, dim, gam, bet, bs, div, (bf16_t*)out, total4);
  else
    hipLaunchKernelGGL(film_apply_kernel<float>, dim3(nblk(total4, 256)), dim3(256), 0, st, (const float*)in, in_B, rows, dim, gam, bet, bs, div, (float*)out, total4);
  return hipGetLastError();
}
hipError_t launch_cast(int prec, const float* in, long n, void* out, hipStream_t st) {
  const long n4 = n / 4;
  if (prec == PREC_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(nblk(n4, 256)), dim3(256), 0, st, in, n4, (bf16_t*)out);
  else hipLaunchKernelGGL(cast_kernel<float>, dim3(nblk(n4, 256)), dim3(256), 0, st, in, n4, (float*)out);
  return hipGetLastError();
}
hipError_t launch_input_dense(int prec, const float* strokes, long rows, const float* w, const float* b, int C,
                              void* out, hipStream_t st) {
  const long n = rows * (C / 4);
  if (prec == PREC_BF16)
    hipLaunchKernelGGL(input_dense_kernel<bf16_t>, dim3(nblk(n, 256)), dim3(256), 0, st, strokes, rows, w, b, C, (bf16_t*)out);
  else
    hipLaunchKernelGGL(input_dense_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, st, strokes, rows, w, b, C, (float*)out);
  return hipGetLastError();
}
hipError_t launch_heads(const HeadsParams& p, hipStream_t st) {
  hipLaunchKernelGGL(heads_kernel, dim3(nblk(p.rows, 16)), dim3(256), 0, st, p);
  return hipGetLastError();
}
hipError_t launch_randn_init(float* xt, long rows, int L, const uint64_t* seed_ptr, int sample_off, hipStream_t st, int iter) {
  hipLaunchKernelGGL(randn_init_kernel, dim3(nblk(rows, 256)), dim3(256), 0, st, xt, rows, L, seed_ptr, sample_off, iter);
  return hipGetLastError();
}
hipError_t launch_set_seed(uint64_t* seed_ptr, uint64_t seed, int64_t first_sample, hipStream_t st) {
  hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, st, seed_ptr, seed, first_sample);
  return hipGetLastError();
}
