// train.hip — first kernels of the training step (reference train.py:26-67): the forward-diffusion
// perturbation, the loss and its gradient, gradient-norm clipping + Adam over flat parameter buffers, and the backward
// pieces of a ConvBlock (cnn.py:64-87) that are not plain GEMMs — FiLM / SiLU backward with its per-sample reductions,
// the weight-gradient contraction over stroke rows on the exact-f32 MFMA, bias gradients, the FiLM Linear backward.
// fp32 throughout (the gradients are checked against the reference's autograd at fp32 tolerances); the data-gradient
// convolutions run on the generic GEMM kernel with transposed / tap-flipped packed weights (dhw_train_api.cpp).
#include <cstdlib>

#include "dhw_common.h"
#include <type_traits>
#include "dhw_kernels.h"
#include "heads_core.h"

namespace {

DHW_DEV float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
DHW_DEV float dsilu_f(float x) { const float s = sigmoid_f(x); return s * (1.0f + x * (1.0f - s)); }

// x_perturbed = sqrt(abar) x + sqrt(1 - abar) eps (train.py:41-43); alphas [B], x / eps [B, L, 2]
__global__ __launch_bounds__(256) void perturb_kernel(const float* x, const float* eps, const float* alphas, long n, int per_sample, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = alphas[i / per_sample];
  out[i] = sqrtf(a) * x[i] + sqrtf(1.0f - a) * eps[i];
}

// loss.py:5-37: score = mean_{b,l} sum_c (eps - pred)^2;  pen = mean_b( mean_l BCE(pen_pred, clamp(pen, 1e-7, 1-1e-7)) * abar_b ).
// One block per sample accumulates its two partial sums into out[1], out[2] (fp32 atomics, B adds each) and writes the
// gradients d loss / d pred [B,L,2], d loss / d pen_pred [B,L].  out[0] = out[1] + out[2] is formed by loss_finish.
__global__ __launch_bounds__(256) void loss_kernel(const float* eps, const float* pred, const float* pen, const float* pen_pred,
                                                    const float* alphas, int B, int L, float* out, float* d_pred, float* d_pen) {
  const int b = blockIdx.x;
  const float a = alphas[b];
  const float inv_bl = 1.0f / ((float)B * (float)L);
  float s = 0.f, q = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const long r = (long)b * L + l;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float d = eps[r * 2 + c] - pred[r * 2 + c];
      s += d * d;
      if (d_pred) d_pred[r * 2 + c] = -2.0f * d * inv_bl;
    }
    const float y = fminf(fmaxf(pen[r], 1e-7f), 1.0f - 1e-7f), pp = pen_pred[r];
    // torch's binary_cross_entropy clamps the logs at -100
    const float lp = fmaxf(logf(pp), -100.0f), lq = fmaxf(logf(1.0f - pp), -100.0f);
    q += -(y * lp + (1.0f - y) * lq);
    if (d_pen) d_pen[r] = (pp - y) / fmaxf(pp * (1.0f - pp), 1e-12f) * a * inv_bl;
  }
  __shared__ float rs[8], rq[8];
  for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rq[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out + 1, (rs[0] + rs[1] + rs[2] + rs[3]) * inv_bl);
    atomicAdd(out + 2, (rq[0] + rq[1] + rq[2] + rq[3]) * a * inv_bl);
  }
}
__global__ void loss_finish_kernel(float* out) { out[0] = out[1] + out[2]; }

// sum of squares of a flat buffer -> *out (atomic), for the global gradient norm (clip_grad.py:42-43, torch clip_grad_norm_)
// (16-byte loads, four independent partial sums per lane: 40 MB in ~10 us; the one-float-per-lane loop took 28)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, long n, float* out) {
  float s = 0.f;
  const long stride = (long)gridDim.x * 256, t = (long)blockIdx.x * 256 + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    const long n4 = n / 4;
    f32x4 a = (f32x4){0, 0, 0, 0}, b = a;
    long i = t;
    // (round 5: eight requests in flight per pass and 512 workgroups — 2048 workgroups x two requests took 32 us for the 40 MB gradient buffer:
    // five dependent round trips per thread and 2048 atomics on one address)
    for (; i + 7 * stride < n4; i += 8 * stride) {
      f32x4 u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = g4[i + k * stride];
#pragma unroll
      for (int k = 0; k < 8; k += 2) { a += u[k] * u[k]; b += u[k + 1] * u[k + 1]; }
    }
    for (; i + stride < n4; i += 2 * stride) {
      const f32x4 u = g4[i], v = g4[i + stride];
      a += u * u;
      b += v * v;
    }
    if (i < n4) { const f32x4 u = g4[i]; a += u * u; }
    a += b;
    s = (a[0] + a[1]) + (a[2] + a[3]);
    for (long k = n4 * 4 + t; k < n; k += stride) s += g[k] * g[k];
  } else {
    for (long i = t; i < n; i += stride) s += g[i] * g[i];
  }
  __shared__ float r[4];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, r[0] + r[1] + r[2] + r[3]);
}

// torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected moments) on a flat buffer, with the
// clip_grad_norm_ factor min(1, max_norm / (||g|| + 1e-6)) read from the device (sqnorm holds ||g||^2).
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, const float* sqnorm, float max_norm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float clip = 1.0f;
  if (sqnorm) clip = fminf(1.0f, max_norm / (sqrtf(*sqnorm) + 1e-6f));
  const float gi = g[i] * clip + wd * p[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr / bc1 * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}

// The same update with its scalars read from device memory — hyper = {lr, beta1, beta2, eps, weight_decay, 1 - beta1^t,
// 1 - beta2^t, max_norm} — so that a captured hipGraph of the whole training step can be replayed with the step's own
// learning rate and bias corrections (the host rewrites the 8 floats before each replay).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, long n, const float* hyper, const float* sqnorm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6], max_norm = hyper[7], gs = hyper[8];
  float clip = gs;   // (gs = 1 / world size behind a SUM all-reduce: the averaging costs no pass of its own; 1 otherwise)
  if (max_norm > 0.f) clip = gs * fminf(1.0f, max_norm / (sqrtf(*sqnorm) * gs + 1e-6f));
  const float gi = g[i] * clip + wd * p[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr / bc1 * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}

// The step's random draws on the device (train.py:39 eps = randn_like(x); text_style.py:97 Dropout(0.3) on the style
// features), from the sampler's counter-based Philox generator: rng = {seed, draw index} in DEVICE memory, so a captured
// graph draws fresh numbers at every replay once the host has bumped the index.  eps[b][l][:] = normal2(seed,
// sample = index * B + b, pos = l, iter = 0); the keep-mask uses iter = 1 and one Philox block per 4 elements.
__global__ __launch_bounds__(256) void train_draw_kernel(const uint64_t* rng, int B, int L, float* eps, long n_keep, int keep_per_sample, float p, float* keep) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const uint64_t seed = rng[0];
  const int64_t base = (int64_t)rng[1] * B;
  if (i < (long)B * L) {
    float z0, z1;
    normal2(seed, base + i / L, (int)(i % L), 0, z0, z1);
    eps[i * 2] = z0;
    eps[i * 2 + 1] = z1;
  }
  if (i * 4 < n_keep) {
    const long e = i * 4;
    const int64_t sample = base + e / keep_per_sample;
    uint32_t c0 = (uint32_t)sample, c1 = (uint32_t)((uint64_t)sample >> 32), c2 = (uint32_t)((e % keep_per_sample) / 4), c3 = 2u;   // iter + 1 = 2
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e + k < n_keep) keep[e + k] = ((float)(c[k] >> 8) + 0.5f) * (1.0f / 16777216.0f) >= p ? 1.0f : 0.0f;
  }
}

// A Dropout(p) keep-mask (1 = kept) over n elements, per_sample of them per batch sample, from the same generator:
// Philox block (sample = index * B + b, element / 4, site) — `site` >= 3 numbers the model's dropout sites (1 = eps, 2 = the
// style mask of train_draw_kernel), so every site of every update draws an independent mask.
__global__ __launch_bounds__(256) void keep_mask_kernel(const uint64_t* rng, uint32_t site, long n, int per_sample, float p, float* keep) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n) return;
  const uint64_t seed = rng[0];
  const int64_t sample = (int64_t)rng[1] * (n / per_sample) + e / per_sample;
  uint32_t c0 = (uint32_t)sample, c1 = (uint32_t)((uint64_t)sample >> 32), c2 = (uint32_t)((e % per_sample) / 4), c3 = site;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (e + k < n) keep[e + k] = ((float)(c[k] >> 8) + 0.5f) * (1.0f / 16777216.0f) >= p ? 1.0f : 0.0f;
}

// a = u * gamma[b] + beta[b];  h = SiLU(a)   (rows C-last [B*L, C]; gamma/beta [B][cols] at column offset)
__global__ __launch_bounds__(256) void film_silu_fwd_kernel(const float* u, const float* film, long film_bs, int goff, int boff, int L, int C,
                                                             long n4, float* a_out, float* h_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long e = i * 4, r = e / C;
  const int c = (int)(e - r * C), b = (int)(r / L);
  const f32x4 x = *reinterpret_cast<const f32x4*>(u + e);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(film + b * film_bs + goff + c), be = *reinterpret_cast<const f32x4*>(film + b * film_bs + boff + c);
  f32x4 a = x * ga + be, h;
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = silu_f(a[k]);
  *reinterpret_cast<f32x4*>(a_out + e) = a;
  if (h_out) *reinterpret_cast<f32x4*>(h_out + e) = h;
}

// Backward of a = u gamma + beta (and, with act, of h = SiLU(a)): given d = dL/dh (or dL/da), writes dL/du = d' gamma and
// accumulates dgamma[b][c] = sum_l d' u, dbeta[b][c] = sum_l d'  (d' = d * SiLU'(a) with act).  One thread per (b, c):
// lanes run along the contiguous channel axis, the loop over the sample's L rows is sequential (deterministic).
__global__ __launch_bounds__(64) void film_bwd_kernel(const float* d, const float* a, const float* u, const float* film, long film_bs, int goff,
                                                       int L, int C, int act, float* du, float* dfilm, long dfilm_bs, int dgoff, int dboff) {
  const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const float ga = film[b * film_bs + goff + c];
  float sg = 0.f, sb = 0.f;
  for (int l = 0; l < L; ++l) {
    const long e = ((long)b * L + l) * C + c;
    float dd = d[e];
    if (act) dd *= dsilu_f(a[e]);
    sg += dd * u[e];
    sb += dd;
    du[e] = dd * ga;
  }
  dfilm[b * dfilm_bs + dgoff + c] = sg;
  dfilm[b * dfilm_bs + dboff + c] = sb;
}

// dx += d_sx * SiLU'(x)   (the conv1 branch reads SiLU(x); the skip branch's dx is already in dx)
__global__ __launch_bounds__(256) void silu_bwd_add_kernel(const float* d_sx, const float* x, long n, float* dx) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] += d_sx[i] * dsilu_f(x[i]);
}
__global__ __launch_bounds__(256) void silu_fwd_kernel(const float* x, long n, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = silu_f(x[i]);
}
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, long n, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// db[c] += sum over all rows of dy[r][c]: block = 64 channels x 4 row groups over a chunk of rows, one atomic per (block, channel)
__global__ __launch_bounds__(256) void colsum_kernel(const float* dy, long rows, int C, int rows_per_block, float* db) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rg; r < r1; r += 4) s += dy[r * C + c];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && c < C) atomicAdd(db + c, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// Weight gradient of Conv1d(k = taps, 'same' zero padding inside each sample) / Linear (taps = 1), torch layout:
//   dW[co][ci][tap] += sum_{b,l} dY[b,l,co] * X[b, l + tap - taps/2, ci]
// a [Cout x Cin] contraction over the B*L stroke rows per tap, on the exact-f32 MFMA (16x16x4 f32: bitwise an fmaf chain).
// One wave = one 16 x 16 (co, ci) tile of one tap over a 256-row chunk; lane (i = lane & 15, g = lane >> 4) feeds
// dY[row 8g + j][co0 + i] as the A operand and X[row 8g + j + shift][ci0 + i] as the B operand (16 lanes read 64
// contiguous bytes of a row).  Chunks accumulate with fp32 atomics (<= rows / 256 adds per element).
__global__ __launch_bounds__(64) void wgrad_kernel(const float* dy, const float* x, int B, int L, int Cout, int Cin, int taps, float* dw) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int ci_tiles = Cin / 16;
  int t = blockIdx.x;
  const int tap = t % taps; t /= taps;
  const int ci0 = (t % ci_tiles) * 16, co0 = (t / ci_tiles) * 16;
  const long rows = (long)B * L, r_begin = (long)blockIdx.y * 256;
  const int shift = tap - taps / 2;
  f32x4 acc = (f32x4){0, 0, 0, 0};
  for (int s = 0; s < 8; ++s) {
    Frag<float> fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long r = r_begin + s * 32 + 8 * g + j;
      float va = 0.f, vb = 0.f;
      if (r < rows) {
        va = dy[r * Cout + co0 + i];
        const int l = (int)(r % L) + shift;
        if (l >= 0 && l < L) vb = x[(r + shift) * Cin + ci0 + i];
      }
      if (j < 4) { fa.lo[j] = va; fb.lo[j] = vb; } else { fa.hi[j - 4] = va; fb.hi[j - 4] = vb; }
    }
    mma32(acc, fa, fb);
  }
  // acc[r] = dW tile[co = 4g + r][ci = i]
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(dw + ((size_t)(co0 + 4 * g + r) * Cin + ci0 + i) * taps + tap, acc[r]);
}

// FiLM Linears (conditioning.py:16-18): film[b][c] = Wcat[c][:] . sigma[b] + bcat[c].  Backward, all of a block's 6 Linears:
// dW[c][k] = sum_b dfilm[b][c] sigma[b][k], db[c] = sum_b dfilm[b][c], dsigma[b][k] = sum_c dfilm[b][c] W[c][k].
__global__ __launch_bounds__(256) void film_linear_bwd_kernel(const float* dfilm, const float* sigma, const float* wcat, int B, int cols,
                                                               float* dw, float* db, float* dsigma) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < cols * 32) {
    const int c = idx / 32, k = idx % 32;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) { s += dfilm[(long)b * cols + c] * sigma[b * 32 + k]; sb += dfilm[(long)b * cols + c]; }
    dw[idx] = s;
    if (k == 0) db[c] = sb;
  }
  if (idx < B * 32) {
    const int b = idx / 32, k = idx % 32;
    float s = 0.f;
    for (int c = 0; c < cols; ++c) s += dfilm[(long)b * cols + c] * wcat[c * 32 + k];
    dsigma[idx] = s;
  }
}

inline unsigned nb(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

hipError_t launch_perturb(const float* x, const float* eps, const float* alphas, int B, int L, float* out, hipStream_t st) {
  const long n = (long)B * L * 2;
  hipLaunchKernelGGL(perturb_kernel, dim3(nb(n)), dim3(256), 0, st, x, eps, alphas, n, L * 2, out);
  return hipGetLastError();
}
hipError_t launch_loss(const float* eps, const float* pred, const float* pen, const float* pen_pred, const float* alphas, int B, int L,
                       float* out3, float* d_pred, float* d_pen, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out3, 0, 3 * sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(loss_kernel, dim3(B), dim3(256), 0, st, eps, pred, pen, pen_pred, alphas, B, L, out3, d_pred, d_pen);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, out3);
  return hipGetLastError();
}
hipError_t launch_sqnorm(const float* g, long n, float* out, hipStream_t st) {   // out must be zeroed by the caller (several buffers add up)
  hipLaunchKernelGGL(sqnorm_kernel, dim3(std::min<unsigned>(nb(n), 512u)), dim3(256), 0, st, g, n, out);
  return hipGetLastError();
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                       const float* sqnorm, float max_norm, hipStream_t st) {
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nb(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2, sqnorm, max_norm);
  return hipGetLastError();
}
hipError_t launch_train_draw(const uint64_t* rng, int B, int L, float* eps, long n_keep, int keep_per_sample, float p, float* keep, hipStream_t st) {
  if (keep_per_sample % 4) return hipErrorInvalidValue;
  const long n = std::max((long)B * L, (n_keep + 3) / 4);
  hipLaunchKernelGGL(train_draw_kernel, dim3(nb(n)), dim3(256), 0, st, rng, B, L, eps, n_keep, keep_per_sample, p, keep);
  return hipGetLastError();
}
hipError_t launch_keep_mask(const uint64_t* rng, int site, long n, int per_sample, float p, float* keep, hipStream_t st) {
  if (per_sample % 4 || n % per_sample || site < 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(keep_mask_kernel, dim3(nb((n + 3) / 4)), dim3(256), 0, st, rng, (uint32_t)site, n, per_sample, p, keep);
  return hipGetLastError();
}
hipError_t launch_adam_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, const float* sqnorm, hipStream_t st) {
  hipLaunchKernelGGL(adam_dev_kernel, dim3(nb(n)), dim3(256), 0, st, p, g, m, v, n, hyper, sqnorm);
  return hipGetLastError();
}
hipError_t launch_film_silu_fwd(const float* u, const float* film, long film_bs, int goff, int boff, int B, int L, int C, float* a, float* h, hipStream_t st) {
  const long n4 = (long)B * L * C / 4;
  hipLaunchKernelGGL(film_silu_fwd_kernel, dim3(nb(n4)), dim3(256), 0, st, u, film, film_bs, goff, boff, L, C, n4, a, h);
  return hipGetLastError();
}
hipError_t launch_film_bwd(const float* d, const float* a, const float* u, const float* film, long film_bs, int goff, int B, int L, int C, int act,
                           float* du, float* dfilm, long dfilm_bs, int dgoff, int dboff, hipStream_t st) {
  hipLaunchKernelGGL(film_bwd_kernel, dim3(nb(C, 64), B), dim3(64), 0, st, d, a, u, film, film_bs, goff, L, C, act, du, dfilm, dfilm_bs, dgoff, dboff);
  return hipGetLastError();
}
hipError_t launch_silu_bwd_add(const float* d_sx, const float* x, long n, float* dx, hipStream_t st) {
  hipLaunchKernelGGL(silu_bwd_add_kernel, dim3(nb(n)), dim3(256), 0, st, d_sx, x, n, dx);
  return hipGetLastError();
}
hipError_t launch_silu_fwd(const float* x, long n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(nb(n)), dim3(256), 0, st, x, n, out);
  return hipGetLastError();
}
hipError_t launch_add(const float* a, const float* b, long n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(add_kernel, dim3(nb(n)), dim3(256), 0, st, a, b, n, out);
  return hipGetLastError();
}
hipError_t launch_colsum(const float* dy, long rows, int C, float* db, hipStream_t st) {   // db zeroed by the caller
  const int rpb = 64;
  hipLaunchKernelGGL(colsum_kernel, dim3(nb(C, 64), nb(rows, rpb)), dim3(256), 0, st, dy, rows, C, rpb, db);
  return hipGetLastError();
}
hipError_t launch_wgrad(const float* dy, const float* x, int B, int L, int Cout, int Cin, int taps, float* dw, hipStream_t st) {   // dw zeroed by the caller
  if (Cout % 16 || Cin % 16 || (taps != 1 && taps != 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wgrad_kernel, dim3((Cout / 16) * (Cin / 16) * taps, nb((long)B * L, 256)), dim3(64), 0, st, dy, x, B, L, Cout, Cin, taps, dw);
  return hipGetLastError();
}
hipError_t launch_film_linear_bwd(const float* dfilm, const float* sigma, const float* wcat, int B, int cols, float* dw, float* db, float* dsigma, hipStream_t st) {
  hipLaunchKernelGGL(film_linear_bwd_kernel, dim3(nb(std::max((long)cols * 32, (long)B * 32))), dim3(256), 0, st, dfilm, sigma, wcat, B, cols, dw, db, dsigma);
  return hipGetLastError();
}

// =====================================================================================================================
// Generic fp32 building blocks of the training step (dhw_train.h "dhw_op_*"): everything the denoiser's forward and
// backward need beyond the fused inference kernels, each a plain device-pointer operation so the host side
// (train_model.py) can chain them the way autograd chains the reference's modules.  Correctness first: the GEMM reads its
// operands straight from global memory with caller-given strides (one description covers Linear / Conv1d forward,
// data gradient, weight gradient and the per-head attention products), on the exact-f32 MFMA.
namespace {

// C[z][m][n] (+)= alpha * sum_k A(z, m, k) * B(z, k, n) (+ bias[n]);  z = zo * nzi + zi (two batch levels, e.g. sample x head)
// with K = taps * Kt and k = tap * Kt + kk (dhw_gemm_desc in include/dhw_train.h):
//   A(z,m,k) = A[zo*sazo + zi*sazi + (m + sa)*sam + kk*sak],  sa = a_shift + tap*a_tap_shift, zero unless (m mod lr) + sa in [0, lr)
//   B(z,k,n) = B[zo*sbzo + zi*sbzi + tap*sbt + (kk + sb)*sbk + n*sbn],  sb = b_shift + zi*b_z_shift, zero unless (kk mod lr) + sb in [0, lr)
// LDS-tiled: one workgroup (4 waves as 2 x 2) = one 64 x 64 tile of C over one K slice; each wave
// owns 32 x 32 (2 x 2 MFMA tiles).  Per 32-wide K step the 64 x 32 A tile and the 32 x 64 B tile go global -> registers
// (issued one step ahead, so their latency hides behind the 32 MFMAs of the current step) -> LDS as As[m][k] / Bs[n][k]
// (row stride 36 floats: the lanes' 16-byte fragment reads fall on disjoint banks) -> two ds_read_b128 per fragment.
// AM / BK pick which index runs along the lanes of a load so that the unit (or smaller) stride is the coalesced one.
// ksplit > 1 (only with accumulate): the K range is cut into slices across workgroups and C is updated with fp32 atomics
// — weight gradients contract over all B*L stroke rows into a few small tiles, and would otherwise run on a few CUs.
constexpr int GT = 64, GK = 32, GS = 36;

// TS = float: exact-f32 MFMA (the default: gradients match the reference's autograd to 1e-5).  TS = bf16_t: the operand tiles
// are rounded to bf16 on their way into LDS and contracted with v_mfma_f32_16x16x32_bf16 (fp32 accumulation, fp32 operands
// in memory, fp32 master weights) — mixed-precision training, 8x fewer MFMA instructions and half the LDS traffic per step.
template <typename TS> constexpr int tile_row = sizeof(TS) == 4 ? GS : 48;   // elements; bf16: 96-byte rows ((stride / 16) mod 4 = 2, gemm_core.h)
DHW_DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
DHW_DEV void st4(bf16_t* p, f32x4 v) {
  bf16_t h[4] = {from_f<bf16_t>(v[0]), from_f<bf16_t>(v[1]), from_f<bf16_t>(v[2]), from_f<bf16_t>(v[3])};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(h);
}
DHW_DEV Frag<float> ld_frag(const float* p) { Frag<float> f; f.lo = *reinterpret_cast<const f32x4*>(p); f.hi = *reinterpret_cast<const f32x4*>(p + 4); return f; }
DHW_DEV Frag<bf16_t> ld_frag(const bf16_t* p) { return frag_load(p); }
// the same 8 k-values out of a k-major fp32 tile: p = &tile[first k][row], rows `stride` floats apart
template <typename TS> DHW_DEV Frag<TS> ld_frag_k(const float* p, int stride);
template <> DHW_DEV Frag<float> ld_frag_k<float>(const float* p, int stride) {
  Frag<float> f;
  f.lo = (f32x4){p[0], p[stride], p[2 * stride], p[3 * stride]};
  f.hi = (f32x4){p[4 * stride], p[5 * stride], p[6 * stride], p[7 * stride]};
  return f;
}
template <> DHW_DEV Frag<bf16_t> ld_frag_k<bf16_t>(const float*, int) { return Frag<bf16_t>{}; }   // (never selected: AKM / BKM are fp32-only)

DHW_DEV float frag_sum(const Frag<float>& f) { return ((f.lo[0] + f.lo[1]) + (f.lo[2] + f.lo[3])) + ((f.hi[0] + f.hi[1]) + (f.hi[2] + f.hi[3])); }
DHW_DEV float frag_sum(const Frag<bf16_t>& f) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)f.v[e];
  return s;
}

// CV: the Conv1d features of the description are in use (taps, row shifts, lr).  The plain variant (every nn.Linear and the
// attention products) compiles without their integer divisions and per-element range tests — the prologue of the general
// form was ~1400 instructions with 19 divisions, as long as the whole K loop of a K = 128 GEMM.
#ifdef DHW_STAMPS
#define SG_STAMP(slot) do { if (g.stamps && bx == 0 && by == gy / 2 && bz == 0 && threadIdx.x == 0) g.stamps[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SG_STAMP(slot) do { } while (0)
#endif
// GM: rows of the output tile, 64 or 32 (columns: always 64).  32-row tiles are for GEMMs whose 64-row tiling would leave CUs
// idle (1 600 - 1 920 rows x 384 columns = 150 - 180 workgroups at the attention level): twice the workgroups, half the K-loop
// work each.
// The body is a device function of the workgroup's tile coordinates (bx = column tile, by = row tile, bz = batch x K slice; gy =
// row tiles, for the diagnostics) and of its LDS block, so that one launch can run two independent GEMMs (sgemm_pair_kernel).
constexpr int SG_BUF = 2 * GT * GS;   // floats per operand buffer (sized for TS = float); a workgroup has two
template <bool AM, bool BK, bool AV, bool BV, typename TS, bool CV, int GM = GT, typename GD = OpGemm>
DHW_DEV void sgemm_body(const GD& g, int ksplit, int kslice, int bx, int by, int bz, int gy, float* smem) {
  static_assert(GM == 64 || GM == 32, "row tile");
  constexpr int MA = GM / 32;   // 16-row MFMA tiles per wave along M
  SG_STAMP(0);
  constexpr int TR = tile_row<TS>;
  // two buffers of operand tiles (TS) — step s is contracted out of one while step s + 1 is staged into the other — then the
  // fp32 output tile
  // fp32 tiles of an operand whose lanes run along m / n (A^T: AM, B [K][N]: !BK) stay k-major in LDS, [k][m] with a row of 66
  // floats: the 16-byte loads go in as two 8-byte stores, conflict-free, and a fragment is eight 4-byte reads (k = 8 q + e: the
  // four lane groups sit 8 rows = 16 banks apart) — instead of transposing with sixteen 4-way-conflicted ds_write_b32 per thread
  // and step (the weight-gradient GEMMs' K step took 1.04 us against 0.74 us for the forms that need no transpose).
  constexpr bool AKM = AM && sizeof(TS) == 4, BKM = !BK && sizeof(TS) == 4;
  constexpr int TRK = 66;
  static_assert(GK * TRK <= GT * GS, "a k-major tile fits the operand's half of a buffer");
  constexpr int BUF = SG_BUF;
  TS* As = reinterpret_cast<TS*>(smem);
  TS* Bs = As + GT * TR;
  constexpr int BUFE = BUF * (int)(sizeof(float) / sizeof(TS));          // the same in elements of TS
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, q = lane >> 4;
  // grid: x = column tile, y = row tile, z = batch index * ksplit + K slice (each division only where its divisor is not 1)
  const int n0 = bx * GT, m0 = by * GM;
  int ks = 0, z = bz, zo = z, zi = 0;
  if (ksplit > 1) { ks = z % ksplit; z /= ksplit; zo = z; }
  if (g.nzi > 1) { zo = z / g.nzi; zi = z - zo * g.nzi; }
  const float* A = g.A + zo * g.sazo + zi * g.sazi;
  const float* B = g.B + zo * g.sbzo + zi * g.sbzi;
  float* C = g.C + zo * g.sczo + zi * g.sczi;
  const int k_begin = ks * kslice, k_end = min(g.K, k_begin + kslice);

  // Staging coordinates of this thread's 8 + 8 elements per K step.  Scalar form (one dword per load):
  //   A: AM (m along lanes): m = t & 63, k = (t >> 6) + 4 j;   else (k along lanes): k = t & 31, m = (t >> 5) + 8 j
  //   B: BK (k along lanes): k = t & 31, n = (t >> 5) + 8 j;   else (n along lanes): n = t & 63, k = (t >> 6) + 4 j
  // Vector form (AV / BV: the lane index has stride exactly 1 and everything is 16-byte aligned), two 16-byte loads:
  //   A: AM: m = 4 (t & 15) .. +3 at k = (t >> 4) + 16 jj;     else: k = 4 (t & 7) .. +3 of row m = (t >> 3) + 32 jj
  //   B: BK: k = 4 (t & 7) .. +3 of column n = (t >> 3) + 32 jj;   else: n = 4 (t & 15) .. +3 at k = (t >> 4) + 16 jj
  // PD K steps of operands are kept in flight in registers (kstep below).
  // Address arithmetic is kept out of the K loop (it was as long as the MFMA work): each element's offset inside its operand
  // is a per-thread 32-bit constant, everything that changes from step to step (k position, tap, row shift) is uniform and
  // goes into the scalar base pointer; the per-step vector work is the validity compares.
  constexpr int PD = 4;
  constexpr int NA = (AV ? 2 : 8) * GM / GT, NB = BV ? 2 : 8;      // loads per thread and step (A: half of them for a 32-row tile)
  float rar[PD][8], rbr[PD][8];
  const int Kt = CV ? g.K / g.taps : g.K;       // taps > 1: Kt is a multiple of GK, so a K step lies inside one tap
  const int b_sh = CV ? g.b_shift + zi * g.b_z_shift : 0;
  const unsigned lr_a = CV && g.lr > 0 ? (unsigned)g.lr : 0x7fffffffu;     // no row shift: every row "in range"
  const unsigned lr_b = CV && b_sh != 0 ? (unsigned)g.lr : 0x7fffffffu;
  // local (tile) coordinates of load j: (am, ak) / (bn, bk); for a vector load the first of its 4 elements
  // (GM = 32 with m along the lanes: 32 m per k row, so 8 / 32 lanes per row and 32 / 8 k rows per pass)
  auto a_m = [&](int j) { return AV ? (AM ? 4 * (t & (GM / 4 - 1)) : (t >> 3) + 32 * j) : (AM ? (t & (GM - 1)) : (t >> 5) + 8 * j); };
  auto a_k = [&](int j) { return AV ? (AM ? (GM == 64 ? (t >> 4) + 16 * j : (t >> 3)) : 4 * (t & 7)) : (AM ? (GM == 64 ? (t >> 6) + 4 * j : (t >> 5) + 8 * j) : (t & 31)); };
  auto b_n = [&](int j) { return BV ? (BK ? (t >> 3) + 32 * j : 4 * (t & 15)) : (BK ? (t >> 5) + 8 * j : (t & 63)); };
  auto b_k = [&](int j) { return BV ? (BK ? 4 * (t & 7) : (t >> 4) + 16 * j) : (BK ? (t & 31) : (t >> 6) + 4 * j); };
  unsigned voa[NA], vob[NB];
  int mla[NA], klb[NB];
  bool mva[NA], nvb[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + a_m(j);
    mva[j] = m < g.M;                            // (a vector load's 4 rows / 4 k are valid together: M, K multiples of 4)
    mla[j] = CV && g.lr > 0 ? m % g.lr : 0;
    voa[j] = (unsigned)(m * (int)g.sam + a_k(j) * (int)g.sak);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + b_n(j);
    nvb[j] = n < g.N;
    vob[j] = (unsigned)(b_k(j) * (int)g.sbk + n * (int)g.sbn);
    klb[j] = CV && b_sh != 0 ? (k_begin + b_k(j)) % g.lr : 0;   // row of the contraction index inside its sample (weight gradients)
  }
  int k_next = k_begin;                          // load() is called for consecutive K steps
  // Every load is issued unconditionally at a clamped (always valid) address and its validity bit is kept with the ring slot;
  // stage() zeroes the invalid elements.  Written as `ok ? *p : 0` each load became a branch with s_waitcnt vmcnt(0) behind
  // it, i.e. every K step waited for the loads it had just issued for three steps ahead: the prefetch ring hid nothing and a
  // step cost one full L2 round trip (17-30 us per GEMM of 0.5 GFLOP; r3 ISA).
  unsigned okm[PD];                              // bits 0..7: the A loads of the slot, bits 8..15: the B loads
  // FAST (interior tile of a plain GEMM whose K slice is whole steps — uniform per workgroup): nothing per element at all, the
  // step's base pointers are scalar (past the end of the slice: the first step again — valid memory, never contracted).  The
  // per-element selects, compares and 64-bit address adds of the general form are ~100 VALU instructions per step, and VALU
  // issue stalls the same SIMD's MFMA pipe: a step took 0.85-0.95 us against 0.43 us of MFMA work (tools/bench_sgemm stamps).
  // MODE 2 / 3: the same for a Conv1d GEMM whose tile (2: forward / data gradient, row-shifted A) or K slice (3: weight gradient,
  // row-shifted B) stays inside the operand: unmasked loads at the shifted addresses, one range test per load for the rows that
  // cross a sample edge (zeroed at staging), nothing for the other operand.
  auto load = [&](float (&ra)[8], float (&rb)[8], unsigned& okbits, auto modec) {
    constexpr int MODE = decltype(modec)::value;
    constexpr bool FAST = MODE != 0;
    if constexpr (FAST) {
      const bool past = k_next >= k_end;      // (uniform) a request past the end of the slice: never contracted
      const int k0 = past ? k_begin : k_next;
      k_next += GK;
      const int tap = MODE >= 2 && g.taps > 1 ? k0 / Kt : 0, kb = k0 - tap * Kt;
      const int a_sh = MODE >= 2 ? g.a_shift + tap * g.a_tap_shift : 0;
      const float* Ab = A + (long)a_sh * g.sam + (long)kb * g.sak;
      const float* Bb = B + (MODE >= 2 ? tap * g.sbt : 0) + (long)(kb + b_sh) * g.sbk;
      // (a row that falls outside the operand is always a row that crosses a sample edge — M and K are whole samples — so the one
      // range test also keeps the load inside the buffer: it is issued at the operand's base instead)
      unsigned bits = 0xffffu;
      bool ea[NA], eb[NB];
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        ea[j] = MODE == 2 && (unsigned)(mla[j] + a_sh) >= lr_a;
        if constexpr (MODE == 2) bits &= ~((ea[j] ? 1u : 0u) << j);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        eb[j] = MODE == 3 && (unsigned)(klb[j] + b_sh) >= lr_b;
        if constexpr (MODE == 3) {
          bits &= ~((eb[j] ? 1u : 0u) << (8 + j));
          klb[j] += GK;
          if (g.lr >= GK) klb[j] -= klb[j] >= g.lr ? g.lr : 0;
          else klb[j] %= g.lr;
        }
      }
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const float* src = MODE == 2 && ea[j] ? A : Ab + voa[j];
        if constexpr (AV) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
        } else {
          ra[j] = *src;
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        // (MODE 3 past the end: klb has moved on while the address went back to the first step — everything from the base)
        const float* src = MODE == 3 && (eb[j] || past) ? B : Bb + vob[j];
        if constexpr (BV) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
        } else {
          rb[j] = *src;
        }
      }
      okbits = bits;
      return;
    }
    const int k0 = k_next;
    k_next += GK;
    const int tap = CV && g.taps > 1 ? k0 / Kt : 0, kb = k0 - tap * Kt;
    const int a_sh = CV ? g.a_shift + tap * g.a_tap_shift : 0;
    const int krem = k_end - k0;                 // <= 0 past the end of the slice: every element invalid
    const float* Ab = A + (long)a_sh * g.sam + (long)kb * g.sak;
    const float* Bb = B + (CV ? tap * g.sbt : 0) + (long)(kb + b_sh) * g.sbk;
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = mva[j] && a_k(j) < krem && (!CV || (unsigned)(mla[j] + a_sh) < lr_a);
      bits |= (ok ? 1u : 0u) << j;
      const float* src = ok ? Ab + voa[j] : A;
      if constexpr (AV) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
      } else {
        ra[j] = *src;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool okb = nvb[j] && b_k(j) < krem && (!CV || (unsigned)(klb[j] + b_sh) < lr_b);
      bits |= (okb ? 1u : 0u) << (8 + j);
      const float* src = okb ? Bb + vob[j] : B;
      if constexpr (BV) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3];
      } else {
        rb[j] = *src;
      }
      if (CV && b_sh != 0) {                     // uniform
        klb[j] += GK;
        if (g.lr >= GK) klb[j] -= klb[j] >= g.lr ? g.lr : 0;   // (uniform; samples shorter than a K step: the general form)
        else klb[j] %= g.lr;
      }
    }
    okbits = bits;
  };
  auto stage = [&](const float (&ra)[8], const float (&rb)[8], unsigned bits, int buf, auto modec) {
    constexpr int MODE = decltype(modec)::value;
    TS* As = reinterpret_cast<TS*>(smem) + buf * BUFE;
    TS* Bs = As + GT * TR;
    auto z = [&](int bit, float v) {   // (bit is a constant after unrolling: A loads 0..7, B loads 8..15)
      if constexpr (MODE == 1) return v;
      else if ((MODE == 2 && bit >= 8) || (MODE == 3 && bit < 8)) return v;
      else return (bits >> bit) & 1u ? v : 0.f;
    };
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if constexpr (AKM) {
        float* Ak = reinterpret_cast<float*>(As);
        if constexpr (AV) {
          float2* d = reinterpret_cast<float2*>(Ak + a_k(j) * TRK + a_m(j));
          d[0] = make_float2(z(j, ra[4 * j]), z(j, ra[4 * j + 1]));
          d[1] = make_float2(z(j, ra[4 * j + 2]), z(j, ra[4 * j + 3]));
        } else Ak[a_k(j) * TRK + a_m(j)] = z(j, ra[j]);
      } else if constexpr (AV && !AM) st4(As + a_m(j) * TR + a_k(j), (f32x4){z(j, ra[4 * j]), z(j, ra[4 * j + 1]), z(j, ra[4 * j + 2]), z(j, ra[4 * j + 3])});
      else if constexpr (AV) {
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(a_m(j) + e) * TR + a_k(j)] = from_f<TS>(z(j, ra[4 * j + e]));
      } else As[a_m(j) * TR + a_k(j)] = from_f<TS>(z(j, ra[j]));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (BKM) {
        float* Bk = reinterpret_cast<float*>(Bs);
        if constexpr (BV) {
          float2* d = reinterpret_cast<float2*>(Bk + b_k(j) * TRK + b_n(j));
          d[0] = make_float2(z(8 + j, rb[4 * j]), z(8 + j, rb[4 * j + 1]));
          d[1] = make_float2(z(8 + j, rb[4 * j + 2]), z(8 + j, rb[4 * j + 3]));
        } else Bk[b_k(j) * TRK + b_n(j)] = z(8 + j, rb[j]);
      } else if constexpr (BV && BK) st4(Bs + b_n(j) * TR + b_k(j), (f32x4){z(8 + j, rb[4 * j]), z(8 + j, rb[4 * j + 1]), z(8 + j, rb[4 * j + 2]), z(8 + j, rb[4 * j + 3])});
      else if constexpr (BV) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(b_n(j) + e) * TR + b_k(j)] = from_f<TS>(z(8 + j, rb[4 * j + e]));
      } else Bs[b_n(j) * TR + b_k(j)] = from_f<TS>(z(8 + j, rb[j]));
    }
  };

  f32x4 acc[MA][2];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  const int wm = (wave >> 1) * (GM / 2), wn = (wave & 1) * 32;

  // bias gradient riding on the weight-gradient GEMM (g.rowsum): the waves that hold the A fragments of the first column tile
  // of batch 0 also add them up — 16 additions per lane and step instead of a second pass over dy (colsum_kernel: one launch
  // per Linear / Conv1d, 8.7 % of the update)
  const bool rs_on = g.rowsum != nullptr && n0 == 0 && z == 0 && wn == 0;   // (wave-uniform)
  float rs[MA] = {};
  // Step s (ring slot p = s mod PD, LDS buffer p & 1): request step s + PD, read this step's fragments, and stage step
  // s + 1 into the other buffer between the two halves of the MFMA work — the matrix pipe runs while the wave does the
  // staging's selects and LDS writes; ONE barrier per step (everybody's reads of this buffer and writes of the next are done).
  // (Fragments of step s + 1 read ahead into a second register set, tile s + 2 staged meanwhile: no faster per step and
  // 8.1 vs 7.6 ms per update for the registers it costs.)
  auto run = [&](auto fastc) {
    SG_STAMP(1);
#pragma unroll
    for (int p = 0; p < PD; ++p) load(rar[p], rbr[p], okm[p], fastc);
    SG_STAMP(2);
    stage(rar[0], rbr[0], okm[0], 0, fastc);
    __syncthreads();
    SG_STAMP(3);
    auto kstep = [&](auto pc) {
      constexpr int p = decltype(pc)::value, pn = (p + 1) % PD;
      const TS* Ac = reinterpret_cast<const TS*>(smem) + (p & 1) * BUFE;
      const TS* Bc = Ac + GT * TR;
      load(rar[p], rbr[p], okm[p], fastc);   // slot p was staged one step ago: it takes step s + PD (past k_end: clamped, all-zero)
      Frag<TS> fa[MA], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (a < MA) {
          if constexpr (AKM) fa[a % MA] = ld_frag_k<TS>(reinterpret_cast<const float*>(Ac) + 8 * q * TRK + wm + 16 * a + i, TRK);
          else fa[a % MA] = ld_frag(Ac + (wm + 16 * a + i) * TR + 8 * q);
        }
        if constexpr (BKM) fb[a] = ld_frag_k<TS>(reinterpret_cast<const float*>(Bc) + 8 * q * TRK + wn + 16 * a + i, TRK);
        else fb[a] = ld_frag(Bc + (wn + 16 * a + i) * TR + 8 * q);
      }
      mma32(acc[0][0], fa[0], fb[0]);
      if constexpr (MA == 1) stage(rar[pn], rbr[pn], okm[pn], pn & 1, fastc);
      mma32(acc[0][1], fa[0], fb[1]);
      if constexpr (MA == 2) {
        stage(rar[pn], rbr[pn], okm[pn], pn & 1, fastc);
        mma32(acc[1][0], fa[1], fb[0]);
        mma32(acc[1][1], fa[1], fb[1]);
      }
      if (rs_on) {
#pragma unroll
        for (int a = 0; a < MA; ++a) rs[a] += frag_sum(fa[a]);
      }
      __syncthreads();
    };
    // Steady state: PD steps per iteration with NO branch inside — hipcc's s_waitcnt insertion loses track of which loads have
    // landed at every control-flow merge and then waits for (nearly) all of them before it reuses a ring register, which
    // serialises the ring just like the vmcnt(0) above.  The remaining steps (up to PD, the last one partial) follow with their tests.
    int kb = k_begin;
    for (; kb + PD * GK <= k_end; kb += PD * GK) {
      kstep(std::integral_constant<int, 0>{});
      kstep(std::integral_constant<int, 1>{});
      kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{});
    }
    static_assert(PD == 4, "the unrolled ring above");
    if (kb < k_end) kstep(std::integral_constant<int, 0>{});
    if (kb + GK < k_end) kstep(std::integral_constant<int, 1>{});
    if (kb + 2 * GK < k_end) kstep(std::integral_constant<int, 2>{});
    if (kb + 3 * GK < k_end) kstep(std::integral_constant<int, 3>{});   // (fewer than PD * GK elements left can still be PD steps, the last one partial)
  };
  // (uniform over the workgroup)
  const bool interior = m0 + GM <= g.M && n0 + GT <= g.N && (k_end - k_begin) % GK == 0;
  if constexpr (!CV) {
    if (interior) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
  } else {
    int mode = 0;
    const bool ashift = g.a_shift != 0 || g.a_tap_shift != 0, bshift = g.b_shift != 0 || g.b_z_shift != 0;
    if (interior && g.lr > 0) {
      if (ashift && !bshift && g.M % g.lr == 0) mode = 2;
      else if (bshift && !ashift && g.taps == 1 && g.K % g.lr == 0) mode = 3;
    }
    if (mode == 2) run(std::integral_constant<int, 2>{});
    else if (mode == 3) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 0>{});
  }

  // acc[a][b][r] = C[m0 + wm + 16 a + 4 q + r][n0 + wn + 16 b + i].  The tile goes through LDS so that a wave-instruction
  // writes 64 consecutive columns of one row (256 contiguous bytes when scn = 1) instead of 16 columns of 4 rows: fp32
  // atomics run at their full rate only for whole 256-byte wave-instructions (MI355X_MICROARCH.md, atomics), and the split-K
  // weight gradients are made of them.
  SG_STAMP(4);
  if (rs_on) {   // lanes i, i + 16, i + 32, i + 48 hold the four k-quarters of row wm + 16 a + i
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      float v = rs[a];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int m = m0 + wm + 16 * a + i;
      if (q == 0 && m < g.M) atomicAdd(g.rowsum + m, v);
    }
  }
  constexpr int CS = GT + 4;   // (16-byte rows for the vector path below; conflict-free for the accumulator writes either way)
  static_assert(GT * CS <= 2 * BUF, "the output tile reuses the operand tiles");
  // (the last K step ended with a barrier: every fragment read and staging write of the operand buffers is done)
  float* Cs = smem;
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(wm + 16 * a + 4 * q + r) * CS + wn + 16 * b + i] = acc[a][b][r];
  __syncthreads();
  // interior tile of a row-major output without split-K: 16 bytes per lane, 4 store instructions per thread instead of 16
  const float* Dd = g.addend ? g.addend + zo * g.sczo + zi * g.sczi : nullptr;   // (ksplit == 1 with an addend / act_out: launch_sgemm)
  float* Ao = g.act_out ? g.act_out + zo * g.sczo + zi * g.sczi : nullptr;
  const float* Du = g.dsilu_of ? g.dsilu_of + zo * g.sczo + zi * g.sczi : nullptr;
  // FiLM (+ SiLU) (+ addend) of the written value as a further output (unbatched GEMMs: the ConvBlock's convolutions and fc)
  float* Fo = g.film_out;
  const float* Fa = g.film_add;
  const bool vec_out = ksplit == 1 && g.scn == 1 && (g.scm & 3) == 0 && m0 + GM <= g.M && n0 + GT <= g.N &&
                       ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(Dd) | reinterpret_cast<uintptr_t>(Ao) | reinterpret_cast<uintptr_t>(Du) | (g.bias ? reinterpret_cast<uintptr_t>(g.bias) : 0) |
                         reinterpret_cast<uintptr_t>(Fo) | reinterpret_cast<uintptr_t>(Fa) | (Fo ? (reinterpret_cast<uintptr_t>(g.film_g) | reinterpret_cast<uintptr_t>(g.film_b) | (uintptr_t)(g.film_ps * 4)) : 0)) & 15) == 0;   // uniform
  if (vec_out) {
    const int c4 = 4 * (t & 15);
    const f32x4 bias = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n0 + c4) : (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < GM / 16; ++it) {
      const int rr = (t >> 4) + 16 * it;
      f32x4* c = reinterpret_cast<f32x4*>(C + (long)(m0 + rr) * g.scm + n0 + c4);
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * CS + c4) * g.alpha + bias;
      if (Dd) v += *reinterpret_cast<const f32x4*>(Dd + (long)(m0 + rr) * g.scm + n0 + c4);
      if (Du) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(Du + (long)(m0 + rr) * g.scm + n0 + c4);
        v = v * (f32x4){dsilu_f(u[0]), dsilu_f(u[1]), dsilu_f(u[2]), dsilu_f(u[3])};
      }
      *c = g.accumulate ? *c + v : v;
      if (Ao) *reinterpret_cast<f32x4*>(Ao + (long)(m0 + rr) * g.scm + n0 + c4) = (f32x4){silu_f(v[0]), silu_f(v[1]), silu_f(v[2]), silu_f(v[3])};
      if (Fo) {
        const long fb = (long)((m0 + rr) / g.film_rows) * g.film_ps + n0 + c4;
        f32x4 f = v * *reinterpret_cast<const f32x4*>(g.film_g + fb) + *reinterpret_cast<const f32x4*>(g.film_b + fb);
        if (g.film_act) f = (f32x4){silu_f(f[0]), silu_f(f[1]), silu_f(f[2]), silu_f(f[3])};
        if (Fa) f += *reinterpret_cast<const f32x4*>(Fa + (long)(m0 + rr) * g.scm + n0 + c4);
        *reinterpret_cast<f32x4*>(Fo + (long)(m0 + rr) * g.scm + n0 + c4) = f;
      }
    }
  } else {
    const int n = n0 + lane;
    if (n < g.N) {
      const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.f;
      float* cn = C + (long)n * g.scn;
      const int rot = ks * 20;   // K slices of one tile start at different rows: their atomics meet on different cache lines
      for (int r0 = wave; r0 < GM; r0 += 4) {
        const int rr = (r0 + rot) & (GM - 1);
        const int m = m0 + rr;
        if (m >= g.M) continue;
        float* c = cn + (long)m * g.scm;
        float v = g.alpha * Cs[rr * CS + lane] + bias;
        if (Dd) v += Dd[(long)n * g.scn + (long)m * g.scm];
        if (Du) v *= dsilu_f(Du[(long)n * g.scn + (long)m * g.scm]);
        if (ksplit > 1) atomicAdd(c, v);
        else *c = g.accumulate ? *c + v : v;
        if (Ao) Ao[(long)n * g.scn + (long)m * g.scm] = silu_f(v);
        if (Fo) {
          const long fb = (long)(m / g.film_rows) * g.film_ps + n;
          float f = v * g.film_g[fb] + g.film_b[fb];
          if (g.film_act) f = silu_f(f);
          if (Fa) f += Fa[(long)n * g.scn + (long)m * g.scm];
          Fo[(long)n * g.scn + (long)m * g.scm] = f;
        }
      }
    }
  }
  SG_STAMP(5);
}

template <bool AM, bool BK, bool AV, bool BV, typename TS, bool CV, int GM = GT>
__global__ __launch_bounds__(256) void sgemm_tiled_kernel(const OpGemm g, int ksplit, int kslice) {
  __shared__ __attribute__((aligned(16))) float smem[2 * SG_BUF];
  sgemm_body<AM, BK, AV, BV, TS, CV, GM>(g, ksplit, kslice, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, smem);
}

// TWO independent GEMMs in one launch (round 4): workgroups [0, n0) run the first, the rest the second.  A layer's weight-gradient
// and data-gradient GEMMs (both read dy, neither reads the other's output) were two of the update's 306 GEMM launches each, and
// every launch of this graph lives >= 4.5 us whatever it computes (the smallest kernels of the trace); as one launch the
// second GEMM's workgroups also fill the CUs that the first one's split-K tail leaves idle.  VA / VB: SgV<...> below (fp32,
// 16-byte-load forms).  The first GEMM's workgroups are dispatched first: the longer one (the weight gradient) goes there.
template <bool AM, bool BK, bool CV, int GM>
struct SgV {
  static DHW_DEV void run(const OpGemm& g, int ksplit, int kslice, int bx, int by, int bz, int gy, float* smem) {
    sgemm_body<AM, BK, true, true, float, CV, GM>(g, ksplit, kslice, bx, by, bz, gy, smem);
  }
};
struct SgGrid { unsigned gx, gy, gz; };
template <typename VA, typename VB>
__global__ __launch_bounds__(256) void sgemm_pair_kernel(const OpGemm g0, int ks0, int kl0, SgGrid r0, const OpGemm g1, int ks1, int kl1, SgGrid r1) {
  __shared__ __attribute__((aligned(16))) float smem[2 * SG_BUF];
  unsigned id = blockIdx.x;
  const unsigned n0 = r0.gx * r0.gy * r0.gz;
  if (id < n0) {
    const unsigned t = id / r0.gx;
    VA::run(g0, ks0, kl0, (int)(id - t * r0.gx), (int)(t % r0.gy), (int)(t / r0.gy), (int)r0.gy, smem);
  } else {
    id -= n0;
    const unsigned t = id / r1.gx;
    VB::run(g1, ks1, kl1, (int)(id - t * r1.gx), (int)(t % r1.gy), (int)(t / r1.gy), (int)r1.gy, smem);
  }
}

// UP TO SIX independent GEMMs in one launch, any mix of the fp32 16-byte-load forms (the q / k / v projections of an attention
// and, backward, their three weight- and three data-gradient GEMMs; dV with dP, dQ with dK).  The descriptors travel by value in
// the kernel-argument segment and are read from there through a constant-address-space pointer (scalar loads, no private copy of
// the one a workgroup picks); a workgroup finds its GEMM from the cumulative workgroup counts.
constexpr int SG_MAXG = 6;
struct SgGroupArgs {
  OpGemm g[SG_MAXG];
  int ksplit[SG_MAXG], kslice[SG_MAXG];
  SgGrid r[SG_MAXG];
  unsigned end[SG_MAXG];   // cumulative workgroup counts (entries past the last GEMM: the total)
  int var[SG_MAXG];        // form: ((A^T ? 2 : B^T ? 1 : 0) * 2 + conv) * 2 + (32-row tiles)
};
typedef const __attribute__((address_space(4))) SgGroupArgs* SgGroupPtr;
typedef const __attribute__((address_space(4))) OpGemm SgDescC;
template <typename TS>   // float: exact-f32 MFMA; bf16_t: operands rounded to bf16 at staging (every member of a group has the same mode)
__global__ __launch_bounds__(256) void sgemm_group_kernel(const SgGroupArgs by_value) {
  __shared__ __attribute__((aligned(16))) float smem[2 * SG_BUF];
  SgGroupPtr a = (SgGroupPtr)__builtin_amdgcn_kernarg_segment_ptr();   // = &by_value
  unsigned id = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 0; k < SG_MAXG - 1; ++k) i += id >= a->end[k] ? 1 : 0;
  if (i) id -= a->end[i - 1];
  const unsigned gx = a->r[i].gx, gy = a->r[i].gy, t = id / gx;
  const int bx = (int)(id - t * gx), by = (int)(t % gy), bz = (int)(t / gy), ks = a->ksplit[i], kl = a->kslice[i];
  SgDescC& g = a->g[i];
#define DHW_SGG(V_, AM_, BK_, CV_, GM_) case V_: sgemm_body<AM_, BK_, true, true, TS, CV_, GM_, SgDescC>(g, ks, kl, bx, by, bz, (int)gy, smem); break
  switch (a->var[i]) {
    DHW_SGG(0, false, false, false, 64); DHW_SGG(1, false, false, false, 32); DHW_SGG(2, false, false, true, 64); DHW_SGG(3, false, false, true, 32);
    DHW_SGG(4, false, true, false, 64);  DHW_SGG(5, false, true, false, 32);  DHW_SGG(6, false, true, true, 64);  DHW_SGG(7, false, true, true, 32);
    DHW_SGG(8, true, false, false, 64);  DHW_SGG(9, true, false, false, 32);  DHW_SGG(10, true, false, true, 64); DHW_SGG(11, true, false, true, 32);
    default: break;
  }
#undef DHW_SGG
}

__global__ __launch_bounds__(256) void unary_kernel(int kind, const float* x, long n, float* y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = kind == 0 ? silu_f(v) : sigmoid_f(v);
}
// kind 0: dx (+)= dy * SiLU'(x);  kind 1: dx (+)= dy * y (1 - y) with y = sigmoid output passed as x
__global__ __launch_bounds__(256) void unary_bwd_kernel(int kind, const float* dy, const float* x, long n, float* dx, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = kind == 0 ? dy[i] * dsilu_f(x[i]) : dy[i] * x[i] * (1.0f - x[i]);
  dx[i] = accumulate ? dx[i] + v : v;
}
// out = a + b (b may be null: copy);  accumulate: out += a (+ b)
__global__ __launch_bounds__(256) void add_kernel2(const float* a, const float* b, long n, float* out, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = a[i] + (b ? b[i] : 0.f);
  out[i] = accumulate ? out[i] + v : v;
}
// out[b][l][c] = x[b][l][c] + table[l][c]   (positional encodings: a constant, no gradient)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* x, const float* table, long n, long per_sample, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] + table[i % per_sample];
}
// y = x * gamma[b] + beta[b] (per-sample [B][C] rows at given strides)
__global__ __launch_bounds__(256) void film_fwd_kernel(const float* x, const float* gam, const float* bet, long pstride, int L, int C, long n, float* y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long r = i / C;
  const int c = (int)(i - r * C), b = (int)(r / L);
  y[i] = x[i] * gam[b * pstride + c] + bet[b * pstride + c];
}
// LayerNorm(eps 1e-6, no affine) over the C channels of each row; one wave per row; keeps mean / rstd for the backward
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, long rows, int C, float* y, float* mean_out, float* rstd_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = rsqrtf(v / C + 1e-6f);
  for (int c = lane; c < C; c += 64) y[row * C + c] = (xr[c] - mean) * rstd;
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}
// dx (+)= rstd * (dy - mean(dy) - y * mean(dy * y)),  y = the normalised output
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* y, const float* rstd, long rows, int C, float* dx, int accumulate) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* d = dy + row * C;
  const float* yr = y + row * C;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) { s1 += d[c]; s2 += d[c] * yr[c]; }
  for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  s1 /= C; s2 /= C;
  const float r = rstd[row];
  for (int c = lane; c < C; c += 64) {
    const float v = r * (d[c] - s1 - yr[c] * s2);
    dx[row * C + c] = accumulate ? dx[row * C + c] + v : v;
  }
}
// P = softmax(S * scale + mask[b][key] * (-1e9)) over the `cols` keys of each row; rows are [B][H][Lq], mask [B][cols] or null
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* s, long rows, int cols, long rows_per_sample, const float* mask, float scale, float* p) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + row * cols;
  const float* mr = mask ? mask + (row / rows_per_sample) * cols : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c] * scale + (mr ? mr[c] * -1e9f : 0.f));
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += expf(sr[c] * scale + (mr ? mr[c] * -1e9f : 0.f) - mx);
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
  for (int c = lane; c < cols; c += 64) p[row * cols + c] = expf(sr[c] * scale + (mr ? mr[c] * -1e9f : 0.f) - mx) / sum;
}
// dS = scale * P * (dP - sum_key(dP * P))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* dp, const float* p, long rows, int cols, float scale, float* ds) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += dp[row * cols + c] * p[row * cols + c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane; c < cols; c += 64) ds[row * cols + c] = scale * p[row * cols + c] * (dp[row * cols + c] - s);
}
// AvgPool1d(2) over rows / its backward;  nearest x2 upsampling / its backward  (rows C-last, L even)
__global__ __launch_bounds__(256) void pool_kernel(int mode, const float* x, long n_out, int C, float* y, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  float v;
  if (mode == 0) v = 0.5f * (x[(2 * r) * C + c] + x[(2 * r + 1) * C + c]);      // pool fwd: out row r <- rows 2r, 2r+1
  else if (mode == 1) v = 0.5f * x[(r / 2) * C + c];                              // pool bwd: dx row r <- 0.5 dy[r/2]
  else if (mode == 2) v = x[(r / 2) * C + c];                                     // upsample fwd: out row r <- row r/2
  else v = x[(2 * r) * C + c] + x[(2 * r + 1) * C + c];                           // upsample bwd: dx row r <- dy[2r] + dy[2r+1]
  y[i] = accumulate ? y[i] + v : v;
}
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* ids, const float* table, long n, int C, float* y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = table[ids[i / C] * C + i % C];
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* ids, const float* dy, long n, int C, float* dtable) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(dtable + ids[i / C] * C + i % C, dy[i]);
}
// y = x * mask * scale (dropout with a supplied keep-mask; also its own backward)
__global__ __launch_bounds__(256) void mask_mul_kernel(const float* x, const float* mask, float scale, long n, float* y, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i] * mask[i] * scale;
  y[i] = accumulate ? y[i] + v : v;
}
// FiLM backward without activation for per-sample [B][C] parameter rows: du (+)= d * gamma, dgamma[b][c] += sum_l d u, dbeta += sum_l d
__global__ __launch_bounds__(256) void film_bwd2_kernel(const float* d, const float* u, const float* gam, long pstride, int L, int C, float* du, int accumulate,
                                                         float* dgam, float* dbet) {
  // block = 64 channels x 4 row groups over a 64-row chunk of one sample; per-(block, channel) partial sums go out as atomics
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, b = blockIdx.y;
  const int l0 = blockIdx.z * 64, l1 = min(L, l0 + 64);
  float sg = 0.f, sb = 0.f;
  if (c < C) {
    const float ga = gam[b * pstride + c];
    for (int l = l0 + rg; l < l1; l += 4) {
      const long e = ((long)b * L + l) * C + c;
      const float dd = d[e];
      sg += dd * u[e];
      sb += dd;
      du[e] = accumulate ? du[e] + dd * ga : dd * ga;
    }
  }
  __shared__ float rs[256], rb[256];
  rs[threadIdx.x] = sg;
  rb[threadIdx.x] = sb;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int x = threadIdx.x;
    atomicAdd(dgam + b * pstride + c, rs[x] + rs[x + 64] + rs[x + 128] + rs[x + 192]);
    atomicAdd(dbet + b * pstride + c, rb[x] + rb[x + 64] + rb[x + 128] + rb[x + 192]);
  }
}

// ---- fused element-wise chains (round 3): the element-wise launches of the op-by-op tape are HBM-bound passes over
// [rows, C] fp32 activations (5-10 us each, ~25 % of an update); FiLM -> SiLU and LayerNorm -> FiLM are evaluated in ONE pass
// each way, the intermediate (FiLM output / normalised rows) is recomputed in the backward instead of stored and re-read.
// y = act ? SiLU(x gamma[b] + beta[b]) : x gamma[b] + beta[b];   4 channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void film_act_fwd_kernel(const float* x, const float* gam, const float* bet, long pstride, int L, int C, long n4,
                                                            int act, const float* addend, float* y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long e = i * 4, r = e / C;
  const int c = (int)(e - r * C), b = (int)(r / L);
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + b * pstride + c), be = *reinterpret_cast<const f32x4*>(bet + b * pstride + c);
  f32x4 a = v * ga + be;
  if (act) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = silu_f(a[k]);
  }
  if (addend) a += *reinterpret_cast<const f32x4*>(addend + e);   // (a residual add riding on the pass)
  *reinterpret_cast<f32x4*>(y + e) = a;
}
// backward of the above: d' = act ? dy * SiLU'(x gamma + beta) : dy;  dx (+)= d' gamma;  dgamma[b][c] += sum_l d' x;  dbeta[b][c] += sum_l d'
// (block = 64 channels x 4 row groups over a 64-row chunk of one sample, as film_bwd2_kernel)
__global__ __launch_bounds__(256) void film_act_bwd_kernel(const float* d, const float* u, const float* gam, const float* bet, long pstride, int L, int C,
                                                            int act, float* du, int accumulate, float* dgam, float* dbet) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6, b = blockIdx.y;
  const int l0 = blockIdx.z * 64, l1 = min(L, l0 + 64);
  float sg = 0.f, sb = 0.f;
  if (c < C) {
    const float ga = gam[b * pstride + c], be = bet[b * pstride + c];
    for (int l = l0 + rg; l < l1; l += 4) {
      const long e = ((long)b * L + l) * C + c;
      const float x = u[e];
      float dd = d[e];
      if (act) dd *= dsilu_f(x * ga + be);
      sg += dd * x;
      sb += dd;
      du[e] = accumulate ? du[e] + dd * ga : dd * ga;
    }
  }
  __shared__ float rs[256], rb[256];
  rs[threadIdx.x] = sg;
  rb[threadIdx.x] = sb;
  __syncthreads();
  if (rg == 0 && c < C) {
    const int x = threadIdx.x;
    atomicAdd(dgam + b * pstride + c, rs[x] + rs[x + 64] + rs[x + 128] + rs[x + 192]);
    atomicAdd(dbet + b * pstride + c, rb[x] + rb[x + 64] + rb[x + 128] + rb[x + 192]);
  }
}
// y = LayerNorm(x) gamma[b] + beta[b]  (eps 1e-6, no LN affine; one wave per row; mean / rstd kept for the backward)
__global__ __launch_bounds__(256) void ln_film_fwd_kernel(const float* x, long rows, int C, const float* gam, const float* bet, long pstride, int L,
                                                           const float* addend, float* y, float* act_out, const float* pe, float* pe_out, float* mean_out, float* rstd_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * C;
  const long pb = (row / L) * pstride;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = rsqrtf(v / C + 1e-6f);
  for (int c = lane; c < C; c += 64) {
    const float v = (xr[c] - mean) * rstd * gam[pb + c] + bet[pb + c] + (addend ? addend[row * C + c] : 0.f);
    y[row * C + c] = v;
    if (act_out) act_out[row * C + c] = silu_f(v);   // (the SiLU an ff_network opens with, utils/nn.py:145)
    if (pe_out) pe_out[row * C + c] = v + pe[(row % L) * C + c];   // (x + PE: what the q / k projections of the next attention take, model.py:41-48)
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}
// backward: xn = (x - mean) rstd;  dn = dy gamma;  dx (+)= rstd (dn - mean(dn) - xn mean(dn xn));  dgamma[b][c] += sum_l dy xn;  dbeta += sum_l dy.
// grid (ceil(L / 8), B): a block's 4 waves take 2 rows each of one sample (8-row chunks keep >= 1 000 blocks in flight at the
// stroke levels; 64-row chunks ran at 43 us per launch), per-lane channel partial sums in registers (C <= 64 * 8), then one LDS
// reduction and one atomic per channel and block.
__global__ __launch_bounds__(256) void ln_film_bwd_kernel(const float* dy, const float* x, const float* mean, const float* rstd, const float* gam, long pstride,
                                                           int L, int C, float* dx, int accumulate, float* dgam, float* dbet) {
  constexpr int KMAX = 8;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
  const int l0 = blockIdx.x * 8, l1 = min(L, l0 + 8);
  float sg[KMAX], sb[KMAX], ga[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { sg[k] = 0.f; sb[k] = 0.f; ga[k] = lane + 64 * k < C ? gam[b * pstride + lane + 64 * k] : 0.f; }
  for (int l = l0 + w; l < l1; l += 4) {
    const long row = (long)b * L + l;
    const float mu = mean[row], rs = rstd[row];
    float xn[KMAX], dn[KMAX], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = lane + 64 * k;
      const bool in = c < C;
      const float d = in ? dy[row * C + c] : 0.f;
      xn[k] = in ? (x[row * C + c] - mu) * rs : 0.f;
      dn[k] = d * ga[k];
      sg[k] += d * xn[k];
      sb[k] += d;
      s1 += dn[k];
      s2 += dn[k] * xn[k];
    }
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= C; s2 /= C;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float v = rs * (dn[k] - s1 - xn[k] * s2);
        dx[row * C + c] = accumulate ? dx[row * C + c] + v : v;
      }
    }
  }
  __shared__ float red[2][4][64 * KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { red[0][w][lane + 64 * k] = sg[k]; red[1][w][lane + 64 * k] = sb[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgam + b * pstride + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbet + b * pstride + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// ---- 16-byte forms of the three passes above (round 4): C % 4 == 0, C <= 512, every pointer and row 16-byte aligned (the launchers
// check and fall back).  A lane holds channels 4 * lane + 256 * k .. + 3 (k < 2), a row is read once into registers; the scalar forms
// moved 4 bytes per lane and instruction and re-read the row for each of LayerNorm's passes (8-10 us per launch against 5-6 us of
// memory time, rocprofv3 trace of tools/bench_train.py).  Same element-wise arithmetic; the reductions associate differently.
DHW_DEV float sum4(const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }
DHW_DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__global__ __launch_bounds__(256) void ln_film_fwd4_kernel(const float* x, long rows, int C, const float* gam, const float* bet, long pstride, int L,
                                                            const float* addend, float* y, float* act_out, const float* pe, float* pe_out, float* mean_out, float* rstd_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const long pb = (row / L) * pstride;
  f32x4 v[2];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = 4 * lane + 256 * k;
    v[k] = c < C ? ld4(x + row * C + c) : (f32x4){0, 0, 0, 0};
    s += sum4(v[k]);
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (4 * lane + 256 * k < C) { const f32x4 d = v[k] - mean; q += sum4(d * d); }
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / C + 1e-6f);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = 4 * lane + 256 * k;
    if (c < C) {
      f32x4 o = (v[k] - mean) * rstd * ld4(gam + pb + c) + ld4(bet + pb + c);
      if (addend) o += ld4(addend + row * C + c);
      *reinterpret_cast<f32x4*>(y + row * C + c) = o;
      if (act_out) *reinterpret_cast<f32x4*>(act_out + row * C + c) = (f32x4){silu_f(o[0]), silu_f(o[1]), silu_f(o[2]), silu_f(o[3])};
      if (pe_out) *reinterpret_cast<f32x4*>(pe_out + row * C + c) = o + ld4(pe + (row % L) * C + c);
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

__global__ __launch_bounds__(256) void ln_film_bwd4_kernel(const float* dy, const float* x, const float* mean, const float* rstd, const float* gam, long pstride,
                                                            int L, int C, float* dx, int accumulate, float* dgam, float* dbet) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
  const int l0 = blockIdx.x * 8, l1 = min(L, l0 + 8);
  const f32x4 z4 = (f32x4){0, 0, 0, 0};
  f32x4 sg[2] = {z4, z4}, sb[2] = {z4, z4}, ga[2];
  bool in[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    in[k] = 4 * lane + 256 * k < C;
    ga[k] = in[k] ? ld4(gam + b * pstride + 4 * lane + 256 * k) : z4;
  }
  for (int l = l0 + w; l < l1; l += 4) {
    const long row = (long)b * L + l;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xn[2], dn[2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = 4 * lane + 256 * k;
      const f32x4 d = in[k] ? ld4(dy + row * C + c) : z4;
      xn[k] = in[k] ? (ld4(x + row * C + c) - mu) * rs : z4;
      dn[k] = d * ga[k];
      sg[k] += d * xn[k];
      sb[k] += d;
      s1 += sum4(dn[k]);
      s2 += sum4(dn[k] * xn[k]);
    }
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    s1 /= C; s2 /= C;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = 4 * lane + 256 * k;
      if (in[k]) {
        f32x4 v = rs * (dn[k] - s1 - xn[k] * s2);
        if (accumulate) v += ld4(dx + row * C + c);
        *reinterpret_cast<f32x4*>(dx + row * C + c) = v;
      }
    }
  }
  __shared__ __attribute__((aligned(16))) float red[2][4][512];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    *reinterpret_cast<f32x4*>(&red[0][w][4 * lane + 256 * k]) = sg[k];
    *reinterpret_cast<f32x4*>(&red[1][w][4 * lane + 256 * k]) = sb[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgam + b * pstride + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbet + b * pstride + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// (block = 16 lanes x 4 channels = 64 channels, 16 row groups over a 64-row chunk of one sample)
__global__ __launch_bounds__(256) void film_act_bwd4_kernel(const float* d, const float* u, const float* gam, const float* bet, long pstride, int L, int C,
                                                             int act, float* du, int accumulate, float* dgam, float* dbet) {
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 64 + 4 * cl, b = blockIdx.y;
  const int l0 = blockIdx.z * 64, l1 = min(L, l0 + 64);
  f32x4 sg = (f32x4){0, 0, 0, 0}, sb = sg;
  if (c < C) {
    const f32x4 ga = ld4(gam + b * pstride + c), be = ld4(bet + b * pstride + c);
    for (int l = l0 + rg; l < l1; l += 16) {
      const long e = ((long)b * L + l) * C + c;
      const f32x4 x = ld4(u + e);
      f32x4 dd = ld4(d + e);
      if (act) {
        const f32x4 a = x * ga + be;
        dd *= (f32x4){dsilu_f(a[0]), dsilu_f(a[1]), dsilu_f(a[2]), dsilu_f(a[3])};
      }
      sg += dd * x;
      sb += dd;
      f32x4 o = dd * ga;
      if (accumulate) o += ld4(du + e);
      *reinterpret_cast<f32x4*>(du + e) = o;
    }
  }
  __shared__ __attribute__((aligned(16))) float rs[16][64], rb[16][64];
  *reinterpret_cast<f32x4*>(&rs[rg][4 * cl]) = sg;
  *reinterpret_cast<f32x4*>(&rb[rg][4 * cl]) = sb;
  __syncthreads();
  const int x = threadIdx.x;
  if (x < 64 && blockIdx.x * 64 + x < C) {
    float a = 0.f, bsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a += rs[r][x]; bsum += rb[r][x]; }
    atomicAdd(dgam + b * pstride + blockIdx.x * 64 + x, a);
    atomicAdd(dbet + b * pstride + blockIdx.x * 64 + x, bsum);
  }
}

// ---- 16-byte forms of the one-float-per-thread passes (same precondition: n, C multiples of 4, aligned bases)
__global__ __launch_bounds__(256) void unary4_kernel(int kind, const float* x, long n4, float* y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = ld4(x + 4 * i);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = kind == 0 ? silu_f(v[k]) : sigmoid_f(v[k]);
  *reinterpret_cast<f32x4*>(y + 4 * i) = o;
}
__global__ __launch_bounds__(256) void add4_kernel(const float* a, const float* b, long n4, float* out, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = ld4(a + 4 * i);
  if (b) v += ld4(b + 4 * i);
  if (accumulate) v += ld4(out + 4 * i);
  *reinterpret_cast<f32x4*>(out + 4 * i) = v;
}
__global__ __launch_bounds__(256) void add_rows4_kernel(const float* x, const float* table, long n4, long per_sample4, float* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) *reinterpret_cast<f32x4*>(out + 4 * i) = ld4(x + 4 * i) + ld4(table + 4 * (i % per_sample4));
}
__global__ __launch_bounds__(256) void mask_mul4_kernel(const float* x, const float* mask, float scale, long n4, float* y, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = ld4(x + 4 * i) * ld4(mask + 4 * i) * scale;
  if (accumulate) v += ld4(y + 4 * i);
  *reinterpret_cast<f32x4*>(y + 4 * i) = v;
}
// (a block = 256 / (C / 4) whole output rows, a thread = 4 channels of one of them: one 32-bit division per thread)
__global__ __launch_bounds__(256) void pool4_kernel(int mode, const float* x, int C, long rows, float* y, int accumulate) {
  const int c4 = C / 4, rpb = 256 / c4, rr = threadIdx.x / c4;
  const long r = (long)blockIdx.x * rpb + rr;
  const int c = 4 * (threadIdx.x - rr * c4);
  if (rr >= rpb || r >= rows) return;
  f32x4 v;
  if (mode == 0) v = 0.5f * (ld4(x + (2 * r) * C + c) + ld4(x + (2 * r + 1) * C + c));
  else if (mode == 1) v = 0.5f * ld4(x + (r / 2) * C + c);
  else if (mode == 2) v = ld4(x + (r / 2) * C + c);
  else v = ld4(x + (2 * r) * C + c) + ld4(x + (2 * r + 1) * C + c);
  if (accumulate) v += ld4(y + r * C + c);
  *reinterpret_cast<f32x4*>(y + r * C + c) = v;
}
// softmax over rows of <= 256 columns held in registers (one exponential per element instead of three evaluations, one read of s)
__global__ __launch_bounds__(256) void softmax_fwd_r_kernel(const float* s, long rows, int cols, long rows_per_sample, const float* mask, float scale, float* p) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + row * cols;
  const float* mr = mask ? mask + (row / rows_per_sample) * cols : nullptr;
  float v[4], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < cols ? sr[c] * scale + (mr ? mr[c] * -1e9f : 0.f) : -INFINITY;
    mx = fmaxf(mx, v[k]);
  }
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = lane + 64 * k < cols ? expf(v[k] - mx) : 0.f;
    sum += v[k];
  }
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < cols) p[row * cols + lane + 64 * k] = v[k] / sum;
}
__global__ __launch_bounds__(256) void softmax_bwd_r_kernel(const float* dp, const float* p, long rows, int cols, float scale, float* ds) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float a[4], b[4], s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    a[k] = c < cols ? dp[row * cols + c] : 0.f;
    b[k] = c < cols ? p[row * cols + c] : 0.f;
    s += a[k] * b[k];
  }
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < cols) ds[row * cols + lane + 64 * k] = scale * b[k] * (a[k] - s);
}

// All AffineTransformLayers' gamma / beta Linears (conditioning.py:16-18; 76 Linears of 32 inputs for num_layers = 2) as
// ONE launch each way.  Column j of the table film[B][TOT] belongs to output channel woff[j] / 32 of some Linear: its weight
// row starts at flat[woff[j]] (32 floats), its bias is flat[boff[j]] — the parameters stay where the state_dict puts them.
// (round 4: a thread owns one column for a chunk of samples — its 32 weights and its bias stay in registers, sigma is broadcast from
// LDS — instead of one block per sample re-reading every weight row: 12 -> ~5 us.  Same sum order per element as before.)
constexpr int FTB = 16;   // samples per block
__global__ __launch_bounds__(64) void film_table_fwd_kernel(const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int TOT,
                                                             float* film) {
  const int j = blockIdx.x * 64 + threadIdx.x, b0 = blockIdx.y * FTB, nb_ = min(FTB, B - b0);
  __shared__ float sg[FTB][32];
  for (int t = threadIdx.x; t < nb_ * 32; t += 64) sg[t >> 5][t & 31] = sigma[(b0 + (t >> 5)) * 32 + (t & 31)];
  __syncthreads();
  if (j >= TOT) return;
  const float* w = flat + woff[j];
  f32x4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(w + 4 * k);
  const float bias = flat[boff[j]];
  for (int b = 0; b < nb_; ++b) {
    float a = bias;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += v[k][0] * sg[b][4 * k] + v[k][1] * sg[b][4 * k + 1] + v[k][2] * sg[b][4 * k + 2] + v[k][3] * sg[b][4 * k + 3];
    film[(long)(b0 + b) * TOT + j] = a;
  }
}
// dW[j][k] += sum_b dfilm[b][j] sigma[b][k];  db[j] += sum_b dfilm[b][j].  Block = 64 columns: the dfilm tile [32 samples][64] and
// sigma [32][32] go through LDS in coalesced passes, a thread = (column, 8 of the 32 k).  (One thread per (column, k) reading dfilm
// straight from memory fetched 8 useful bytes per wave-instruction: 30 us for 38 MFLOP.)
__global__ __launch_bounds__(256) void film_table_wgrad_kernel(const float* dfilm, const float* sigma, const int64_t* woff, const int64_t* boff, int B,
                                                                int TOT, float* gflat) {
  __shared__ float df[32][64], sg[32][32];
  const int t = threadIdx.x, jl = t & 63, kq = t >> 6, j0 = blockIdx.x * 64, j = j0 + jl;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.f;
  for (int b0 = 0; b0 < B; b0 += 32) {
    const int nb_ = min(32, B - b0);
    __syncthreads();
    // (unconditional clamped requests, all in flight before the first LDS store: see film_table_dgrad_kernel)
    float dv[8], sv[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = t + u * 256, b = e >> 6, c = e & 63;
      dv[u] = dfilm[(long)(b0 + min(b, nb_ - 1)) * TOT + min(j0 + c, TOT - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + u * 256;
      sv[u] = sigma[(b0 + min(e >> 5, nb_ - 1)) * 32 + (e & 31)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = t + u * 256, b = e >> 6, c = e & 63;
      df[b][c] = b < nb_ && j0 + c < TOT ? dv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + u * 256;
      sg[e >> 5][e & 31] = (e >> 5) < nb_ ? sv[u] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int b = 0; b < 32; ++b) {
      const float d = df[b][jl];
      sb += d;
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += d * sg[b][8 * kq + k];
    }
  }
  if (j >= TOT) return;
  float* gw = gflat + woff[j] + 8 * kq;
#pragma unroll
  for (int k = 0; k < 8; ++k) gw[k] += s[k];
  if (kq == 0) gflat[boff[j]] += sb;
}
// dsigma[b][k] += sum_j dfilm[b][j] W[j][k]: block = FTD columns for 32 samples; the weight rows of the chunk ([FTD][32]) and the
// dfilm tile ([32][FTD]) through LDS once, a thread = (sample, 4 of the 32 k), four atomics per thread.  (One block per (sample,
// chunk) re-read the chunk's weight rows for every sample: 76 MB of L2 reads for a 2.4 MB matrix, 19 us.)
constexpr int FTD = 64;   // (290 column chunks for the 18 560 columns of num_layers = 2, walked by 64 workgroups)
__global__ __launch_bounds__(256) void film_table_dgrad_kernel(const float* dfilm, const float* flat, const int64_t* woff, int B, int TOT, float* dsigma) {
  __shared__ __attribute__((aligned(16))) float W[FTD][36];
  __shared__ float df[32][FTD + 1];
  const int t = threadIdx.x, b0 = blockIdx.y * 32, nb_ = min(32, B - b0);
  const int b = t >> 3, kq = t & 7;
  f32x4 s = (f32x4){0, 0, 0, 0};
  // (round 5: a workgroup walks over several column chunks and adds its 1 024 partial sums to dsigma ONCE — 290 workgroups x 1 024 atomics on the same
  // 1 024 addresses were 25 of the kernel's 31 us; every request unconditional at a clamped address, all of a pass in flight before its first LDS store)
  for (int j0 = blockIdx.x * FTD; j0 < TOT; j0 += gridDim.x * FTD) {
    const int nj = min(FTD, TOT - j0);
    int64_t wo[FTD * 8 / 256];
#pragma unroll
    for (int u = 0; u < FTD * 8 / 256; ++u) wo[u] = woff[j0 + min((t + u * 256) >> 3, nj - 1)];
    float dv[32 * FTD / 256];
#pragma unroll
    for (int u = 0; u < 32 * FTD / 256; ++u) {
      const int e = t + u * 256, bb = e / FTD, c = e - bb * FTD;
      dv[u] = dfilm[(long)(b0 + min(bb, nb_ - 1)) * TOT + j0 + min(c, nj - 1)];
    }
    f32x4 wv[FTD * 8 / 256];
#pragma unroll
    for (int u = 0; u < FTD * 8 / 256; ++u) wv[u] = *reinterpret_cast<const f32x4*>(flat + wo[u] + 4 * ((t + u * 256) & 7));
    __syncthreads();   // (the previous chunk's tiles have been read)
#pragma unroll
    for (int u = 0; u < 32 * FTD / 256; ++u) {
      const int e = t + u * 256, bb = e / FTD, c = e - bb * FTD;
      df[bb][c] = bb < nb_ && c < nj ? dv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < FTD * 8 / 256; ++u) {
      const int e = t + u * 256, c = e >> 3, k4 = e & 7;
      *reinterpret_cast<f32x4*>(&W[c][4 * k4]) = c < nj ? wv[u] : (f32x4){0, 0, 0, 0};
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < FTD; ++c) s += df[b][c] * *reinterpret_cast<const f32x4*>(&W[c][4 * kq]);
  }
  if (b < nb_) {
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(dsigma + (b0 + b) * 32 + 4 * kq + k, s[k]);
  }
}

}  // namespace

hipError_t launch_film_table(int dir, const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int TOT, float* film,
                             float* gflat, float* dsigma, hipStream_t st) {
  if (dir == 0) {
    hipLaunchKernelGGL(film_table_fwd_kernel, dim3(nb(TOT, 64), nb(B, FTB)), dim3(64), 0, st, sigma, flat, woff, boff, B, TOT, film);
  } else {
    hipLaunchKernelGGL(film_table_wgrad_kernel, dim3(nb(TOT, 64)), dim3(256), 0, st, film, sigma, woff, boff, B, TOT, gflat);
    hipLaunchKernelGGL(film_table_dgrad_kernel, dim3(std::min<unsigned>(nb(TOT, FTD), 64u), nb(B, 32)), dim3(256), 0, st, film, flat, woff, B, TOT, dsigma);
  }
  return hipGetLastError();
}
// tile / split / load-form choice of one GEMM (shared by the single and the paired launch)
struct SgPlan { int ksplit, kslice; bool am, bk, av, bv, cv, gm32; dim3 grid; };
static hipError_t plan_sgemm(const OpGemm& g, SgPlan& pl) {
  if (g.M < 1 || g.N < 1 || g.K < 1 || g.nzo < 1 || g.nzi < 1 || g.taps < 1) return hipErrorInvalidValue;
  if (g.taps > 1 && (g.K % g.taps || (g.K / g.taps) % GK)) return hipErrorInvalidValue;
  const int tiles_m = (g.M + GT - 1) / GT, tiles_n = (g.N + GT - 1) / GT;
  const long wgs = (long)tiles_m * tiles_n * g.nzo * g.nzi;
  // split K across workgroups while the tile count leaves most of the 256 CUs idle (accumulating outputs only: atomics)
  int ksplit = 1;
  static const long sk_target = getenv("DHW_SGEMM_SPLIT_WGS") ? atol(getenv("DHW_SGEMM_SPLIT_WGS")) : 512;   // (two workgroups per CU: 7.6 vs 7.8 ms per update against 256)
  static const long sk_steps = getenv("DHW_SGEMM_SPLIT_STEPS") ? atol(getenv("DHW_SGEMM_SPLIT_STEPS")) : 8;
  if ((g.act_out || g.film_out) && g.accumulate) return hipErrorInvalidValue;
  if (g.dsilu_of && (g.bias || g.addend)) return hipErrorInvalidValue;   // (a factor on the product alone)
  if (g.accumulate && !g.addend && !g.dsilu_of && wgs < sk_target && g.K >= 2 * sk_steps * GK) ksplit = (int)std::min<long>((sk_target + wgs - 1) / wgs, g.K / (sk_steps * GK));
  if (ksplit < 1) ksplit = 1;
  const int kslice = ((g.K + ksplit - 1) / ksplit + GK - 1) / GK * GK;
  ksplit = (g.K + kslice - 1) / kslice;
  if (tiles_m > 32767 || (long)g.nzo * g.nzi * ksplit > 65535) return hipErrorInvalidValue;
  // lanes run along the index whose stride is the smaller one; 16-byte loads where that stride is 1 and everything is aligned
  const bool am = std::llabs(g.sam) < std::llabs(g.sak), bk = std::llabs(g.sbk) < std::llabs(g.sbn);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool av = am ? (g.sam == 1 && g.sak % 4 == 0 && g.M % 4 == 0 && g.a_shift == 0 && g.a_tap_shift == 0)
               : (g.sak == 1 && g.sam % 4 == 0 && g.K % 4 == 0);
  av = av && al16(g.A) && g.sazo % 4 == 0 && g.sazi % 4 == 0;
  bool bv = bk ? (g.sbk == 1 && g.sbn % 4 == 0 && g.K % 4 == 0 && g.b_shift == 0 && g.b_z_shift == 0)
               : (g.sbn == 1 && g.sbk % 4 == 0 && g.N % 4 == 0);
  bv = bv && al16(g.B) && g.sbzo % 4 == 0 && g.sbzi % 4 == 0 && g.sbt % 4 == 0;
  static const bool novec = [] { const char* e = getenv("DHW_SGEMM_SCALAR"); return e && *e == '1'; }();
  if (novec) av = bv = false;
  const bool cv = g.lr > 0 || g.taps > 1 || g.a_shift || g.a_tap_shift || g.b_shift || g.b_z_shift;
  // 32-row tiles where the 64-row tiling would leave CUs idle (fp32, 16-byte-load forms, no split-K): DHW_SGEMM_GM32=0 to compare
  static const bool gm32_on = !(getenv("DHW_SGEMM_GM32") && atoi(getenv("DHW_SGEMM_GM32")) == 0);
  const bool gm32 = gm32_on && av && bv && !g.bf16 && ksplit == 1 && wgs < 224 && g.M > 32;
  pl = SgPlan{ksplit, kslice, am, bk, av, bv, cv, gm32, dim3((unsigned)tiles_n, (unsigned)(gm32 ? (g.M + 31) / 32 : tiles_m), (unsigned)(g.nzo * g.nzi * ksplit))};
  return hipSuccess;
}
static hipError_t launch_planned(const OpGemm& g, const SgPlan& pl, hipStream_t st) {
  using KFn = void (*)(const OpGemm, int, int);
#define DHW_SG4(AM_, BK_, TS_, CV_) sgemm_tiled_kernel<AM_, BK_, false, false, TS_, CV_>, sgemm_tiled_kernel<AM_, BK_, false, true, TS_, CV_>, \
                                    sgemm_tiled_kernel<AM_, BK_, true, false, TS_, CV_>, sgemm_tiled_kernel<AM_, BK_, true, true, TS_, CV_>
#define DHW_SG16(TS_, CV_) DHW_SG4(false, false, TS_, CV_), DHW_SG4(false, true, TS_, CV_), DHW_SG4(true, false, TS_, CV_), DHW_SG4(true, true, TS_, CV_)
  static const KFn variants[64] = {DHW_SG16(float, false), DHW_SG16(bf16_t, false), DHW_SG16(float, true), DHW_SG16(bf16_t, true)};
#undef DHW_SG16
#undef DHW_SG4
  const dim3 block(256);
  if (pl.gm32) {
#define DHW_SG32(AM_, BK_) sgemm_tiled_kernel<AM_, BK_, true, true, float, false, 32>, sgemm_tiled_kernel<AM_, BK_, true, true, float, true, 32>
    static const KFn v32[8] = {DHW_SG32(false, false), DHW_SG32(false, true), DHW_SG32(true, false), DHW_SG32(true, true)};
#undef DHW_SG32
    hipLaunchKernelGGL(v32[pl.am * 4 + pl.bk * 2 + (pl.cv ? 1 : 0)], pl.grid, block, 0, st, g, pl.ksplit, pl.kslice);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(variants[(pl.cv ? 32 : 0) + (g.bf16 ? 16 : 0) + pl.am * 8 + pl.bk * 4 + pl.av * 2 + pl.bv], pl.grid, block, 0, st, g, pl.ksplit, pl.kslice);
  return hipGetLastError();
}
hipError_t launch_sgemm(const OpGemm& g, hipStream_t st) {
  SgPlan pl;
  const hipError_t e = plan_sgemm(g, pl);
  return e != hipSuccess ? e : launch_planned(g, pl, st);
}
hipError_t launch_sgemm_group(const OpGemm* g, int n, hipStream_t st, int* launches);
// a: a weight gradient (A^T B: m along the lanes of A, n along the lanes of B, 64-row tiles), b: a data gradient (A B with B [K][N]);
// both fp32 with 16-byte loads.  Anything else, or DHW_SGEMM_PAIR=0: two launches.
hipError_t launch_sgemm_pair(const OpGemm& a, const OpGemm& b, hipStream_t st, int* launches) {
  if (launches) *launches = 2;
  SgPlan pa, pb;
  hipError_t e;
  if ((e = plan_sgemm(a, pa)) != hipSuccess || (e = plan_sgemm(b, pb)) != hipSuccess) return e;
  static const bool off = [] { const char* v = getenv("DHW_SGEMM_PAIR"); return v && atoi(v) == 0; }();
  if (!off && a.bf16 && b.bf16) {   // (the mixed-precision mode: through the general grouped kernel)
    const OpGemm two[2] = {a, b};
    return launch_sgemm_group(two, 2, st, launches);
  }
  const bool ok = !off && !a.bf16 && !b.bf16 && !a.stamps && !b.stamps && pa.av && pa.bv && pb.av && pb.bv && pa.am && !pa.bk && !pa.gm32 && !pb.am && !pb.bk;
  if (!ok) {
    if ((e = launch_planned(a, pa, st)) != hipSuccess) return e;
    return launch_planned(b, pb, st);
  }
  using PFn = void (*)(const OpGemm, int, int, SgGrid, const OpGemm, int, int, SgGrid);
#define DHW_SGP(CVA_, CVB_, GMB_) sgemm_pair_kernel<SgV<true, false, CVA_, 64>, SgV<false, false, CVB_, GMB_>>
  static const PFn pairs[8] = {DHW_SGP(false, false, 64), DHW_SGP(false, false, 32), DHW_SGP(false, true, 64), DHW_SGP(false, true, 32),
                               DHW_SGP(true, false, 64),  DHW_SGP(true, false, 32),  DHW_SGP(true, true, 64),  DHW_SGP(true, true, 32)};
#undef DHW_SGP
  const SgGrid ra{pa.grid.x, pa.grid.y, pa.grid.z}, rb{pb.grid.x, pb.grid.y, pb.grid.z};
  const unsigned long n = (unsigned long)ra.gx * ra.gy * ra.gz + (unsigned long)rb.gx * rb.gy * rb.gz;
  if (n > 0x7fffffffUL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pairs[(pa.cv ? 4 : 0) + (pb.cv ? 2 : 0) + (pb.gm32 ? 1 : 0)], dim3((unsigned)n), dim3(256), 0, st, a, pa.ksplit, pa.kslice, ra, b, pb.ksplit, pb.kslice, rb);
  if (launches) *launches = 1;
  return hipGetLastError();
}
// n <= SG_MAXG independent GEMMs: one launch when every one of them is an fp32 16-byte-load form (and not A^T B^T), else one by one.
// DHW_SGEMM_GROUP=0: one by one.
hipError_t launch_sgemm_group(const OpGemm* g, int n, hipStream_t st, int* launches) {
  if (n < 1 || n > SG_MAXG) return hipErrorInvalidValue;
  if (launches) *launches = n;
  SgPlan pl[SG_MAXG];
  hipError_t e;
  static const bool off = [] { const char* v = getenv("DHW_SGEMM_GROUP"); return v && atoi(v) == 0; }();
  bool ok = !off && n > 1;
  for (int i = 0; i < n; ++i) {
    if ((e = plan_sgemm(g[i], pl[i])) != hipSuccess) return e;
    ok = ok && g[i].bf16 == g[0].bf16 && !g[i].stamps && pl[i].av && pl[i].bv && !(pl[i].am && pl[i].bk) && !(g[i].bf16 && pl[i].gm32);
  }
  if (!ok) {
    for (int i = 0; i < n; ++i)
      if ((e = launch_planned(g[i], pl[i], st)) != hipSuccess) return e;
    return hipSuccess;
  }
  SgGroupArgs a{};
  unsigned long tot = 0;
  for (int i = 0; i < SG_MAXG; ++i) {
    if (i < n) {
      a.g[i] = g[i];
      a.ksplit[i] = pl[i].ksplit; a.kslice[i] = pl[i].kslice;
      a.r[i] = SgGrid{pl[i].grid.x, pl[i].grid.y, pl[i].grid.z};
      a.var[i] = ((pl[i].am ? 2 : pl[i].bk ? 1 : 0) * 2 + (pl[i].cv ? 1 : 0)) * 2 + (pl[i].gm32 ? 1 : 0);
      tot += (unsigned long)pl[i].grid.x * pl[i].grid.y * pl[i].grid.z;
    } else {
      a.r[i] = SgGrid{1, 1, 1};
      a.var[i] = -1;
    }
    a.end[i] = (unsigned)tot;
  }
  if (tot > 0x7fffffffUL) return hipErrorInvalidValue;
  if (g[0].bf16) hipLaunchKernelGGL(sgemm_group_kernel<bf16_t>, dim3((unsigned)tot), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(sgemm_group_kernel<float>, dim3((unsigned)tot), dim3(256), 0, st, a);
  if (launches) *launches = 1;
  return hipGetLastError();
}
// the 16-byte kernels' precondition: whole f32x4 per lane (C, the table's row stride) and 16-byte aligned bases (null = absent).
// DHW_TRAIN_VEC4=0: the scalar forms everywhere (A/B)
static bool vec4_ok(int C, long pstride, std::initializer_list<const void*> ptrs) {
  static const bool off = [] { const char* e = getenv("DHW_TRAIN_VEC4"); return e && atoi(e) == 0; }();
  if (off || C % 4 || pstride % 4) return false;
  for (const void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) & 15) return false;
  return true;
}
hipError_t launch_unary(int kind, const float* x, long n, float* y, hipStream_t st) {
  if (n % 4 == 0 && vec4_ok(4, 0, {x, y})) hipLaunchKernelGGL(unary4_kernel, dim3(nb(n / 4)), dim3(256), 0, st, kind, x, n / 4, y);
  else hipLaunchKernelGGL(unary_kernel, dim3(nb(n)), dim3(256), 0, st, kind, x, n, y);
  return hipGetLastError();
}
hipError_t launch_unary_bwd(int kind, const float* dy, const float* x, long n, float* dx, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(unary_bwd_kernel, dim3(nb(n)), dim3(256), 0, st, kind, dy, x, n, dx, accumulate);
  return hipGetLastError();
}
hipError_t launch_add2(const float* a, const float* b, long n, float* out, int accumulate, hipStream_t st) {
  if (n % 4 == 0 && vec4_ok(4, 0, {a, b, out})) hipLaunchKernelGGL(add4_kernel, dim3(nb(n / 4)), dim3(256), 0, st, a, b, n / 4, out, accumulate);
  else hipLaunchKernelGGL(add_kernel2, dim3(nb(n)), dim3(256), 0, st, a, b, n, out, accumulate);
  return hipGetLastError();
}
hipError_t launch_add_rows(const float* x, const float* table, long n, long per_sample, float* out, hipStream_t st) {
  if (n % 4 == 0 && per_sample % 4 == 0 && vec4_ok(4, 0, {x, table, out}))
    hipLaunchKernelGGL(add_rows4_kernel, dim3(nb(n / 4)), dim3(256), 0, st, x, table, n / 4, per_sample / 4, out);
  else hipLaunchKernelGGL(add_rows_kernel, dim3(nb(n)), dim3(256), 0, st, x, table, n, per_sample, out);
  return hipGetLastError();
}
hipError_t launch_film_fwd(const float* x, const float* gam, const float* bet, long pstride, int B, int L, int C, float* y, hipStream_t st) {
  const long n = (long)B * L * C;
  hipLaunchKernelGGL(film_fwd_kernel, dim3(nb(n)), dim3(256), 0, st, x, gam, bet, pstride, L, C, n, y);
  return hipGetLastError();
}
hipError_t launch_film_bwd2(const float* d, const float* u, const float* gam, long pstride, int B, int L, int C, float* du, int accumulate, float* dgam,
                            float* dbet, hipStream_t st) {
  hipLaunchKernelGGL(film_bwd2_kernel, dim3(nb(C, 64), B, nb(L, 64)), dim3(256), 0, st, d, u, gam, pstride, L, C, du, accumulate, dgam, dbet);
  return hipGetLastError();
}
hipError_t launch_film_act_fwd(const float* x, const float* gam, const float* bet, long pstride, int B, int L, int C, int act, const float* addend, float* y,
                               hipStream_t st) {
  const long n4 = (long)B * L * C / 4;
  hipLaunchKernelGGL(film_act_fwd_kernel, dim3(nb(n4)), dim3(256), 0, st, x, gam, bet, pstride, L, C, n4, act, addend, y);
  return hipGetLastError();
}
hipError_t launch_film_act_bwd(const float* d, const float* u, const float* gam, const float* bet, long pstride, int B, int L, int C, int act, float* du,
                               int accumulate, float* dgam, float* dbet, hipStream_t st) {
  if (vec4_ok(C, pstride, {d, u, gam, bet, du}))
    hipLaunchKernelGGL(film_act_bwd4_kernel, dim3(nb(C, 64), B, nb(L, 64)), dim3(256), 0, st, d, u, gam, bet, pstride, L, C, act, du, accumulate, dgam, dbet);
  else
    hipLaunchKernelGGL(film_act_bwd_kernel, dim3(nb(C, 64), B, nb(L, 64)), dim3(256), 0, st, d, u, gam, bet, pstride, L, C, act, du, accumulate, dgam, dbet);
  return hipGetLastError();
}
hipError_t launch_ln_film_fwd(const float* x, long rows, int C, const float* gam, const float* bet, long pstride, int L, const float* addend, float* y,
                              float* act_out, const float* pe, float* pe_out, float* mean, float* rstd, hipStream_t st) {
  if (C <= 512 && vec4_ok(C, pstride, {x, gam, bet, addend, y, act_out, pe, pe_out}))
    hipLaunchKernelGGL(ln_film_fwd4_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, x, rows, C, gam, bet, pstride, L, addend, y, act_out, pe, pe_out, mean, rstd);
  else
    hipLaunchKernelGGL(ln_film_fwd_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, x, rows, C, gam, bet, pstride, L, addend, y, act_out, pe, pe_out, mean, rstd);
  return hipGetLastError();
}
hipError_t launch_ln_film_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gam, long pstride, int B, int L, int C,
                              float* dx, int accumulate, float* dgam, float* dbet, hipStream_t st) {
  if (C <= 512 && vec4_ok(C, pstride, {dy, x, gam, dx}))
    hipLaunchKernelGGL(ln_film_bwd4_kernel, dim3(nb(L, 8), B), dim3(256), 0, st, dy, x, mean, rstd, gam, pstride, L, C, dx, accumulate, dgam, dbet);
  else
    hipLaunchKernelGGL(ln_film_bwd_kernel, dim3(nb(L, 8), B), dim3(256), 0, st, dy, x, mean, rstd, gam, pstride, L, C, dx, accumulate, dgam, dbet);
  return hipGetLastError();
}
hipError_t launch_ln_fwd(const float* x, long rows, int C, float* y, float* mean, float* rstd, hipStream_t st) {
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, x, rows, C, y, mean, rstd);
  return hipGetLastError();
}
hipError_t launch_ln_bwd(const float* dy, const float* y, const float* rstd, long rows, int C, float* dx, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, dy, y, rstd, rows, C, dx, accumulate);
  return hipGetLastError();
}
hipError_t launch_softmax_fwd(const float* s, long rows, int cols, long rows_per_sample, const float* mask, float scale, float* p, hipStream_t st) {
  if (cols <= 256) hipLaunchKernelGGL(softmax_fwd_r_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, s, rows, cols, rows_per_sample, mask, scale, p);
  else hipLaunchKernelGGL(softmax_fwd_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, s, rows, cols, rows_per_sample, mask, scale, p);
  return hipGetLastError();
}
hipError_t launch_softmax_bwd(const float* dp, const float* p, long rows, int cols, float scale, float* ds, hipStream_t st) {
  if (cols <= 256) hipLaunchKernelGGL(softmax_bwd_r_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, dp, p, rows, cols, scale, ds);
  else hipLaunchKernelGGL(softmax_bwd_kernel, dim3(nb(rows, 4)), dim3(256), 0, st, dp, p, rows, cols, scale, ds);
  return hipGetLastError();
}
hipError_t launch_pool(int mode, const float* x, long n_out, int C, float* y, int accumulate, hipStream_t st) {
  if (n_out % C == 0 && C <= 1024 && vec4_ok(C, 0, {x, y}))
    hipLaunchKernelGGL(pool4_kernel, dim3(nb(n_out / C, 256 / (C / 4))), dim3(256), 0, st, mode, x, C, n_out / C, y, accumulate);
  else hipLaunchKernelGGL(pool_kernel, dim3(nb(n_out)), dim3(256), 0, st, mode, x, n_out, C, y, accumulate);
  return hipGetLastError();
}
hipError_t launch_embed(int bwd, const int64_t* ids, const float* src, long n, int C, float* dst, hipStream_t st) {
  if (bwd) hipLaunchKernelGGL(embed_bwd_kernel, dim3(nb(n)), dim3(256), 0, st, ids, src, n, C, dst);
  else hipLaunchKernelGGL(embed_fwd_kernel, dim3(nb(n)), dim3(256), 0, st, ids, src, n, C, dst);
  return hipGetLastError();
}
hipError_t launch_mask_mul(const float* x, const float* mask, float scale, long n, float* y, int accumulate, hipStream_t st) {
  if (n % 4 == 0 && vec4_ok(4, 0, {x, mask, y})) hipLaunchKernelGGL(mask_mul4_kernel, dim3(nb(n / 4)), dim3(256), 0, st, x, mask, scale, n / 4, y, accumulate);
  else hipLaunchKernelGGL(mask_mul_kernel, dim3(nb(n)), dim3(256), 0, st, x, mask, scale, n, y, accumulate);
  return hipGetLastError();
}
