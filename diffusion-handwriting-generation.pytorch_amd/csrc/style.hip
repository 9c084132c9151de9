// style.hip — the spatial kernels of the StyleExtractor front end (reference text_style.py:11-59: torchvision
// MobileNetV2 `features` + AvgPool2d(3, 3) + AdaptiveAvgPool2d((1, 14))).  Activations are NHWC [B, H, W, Cp] with the
// channel count padded to a multiple of 64 (padding channels carry zero weights and stay exactly 0), BatchNorm is
// folded into the convolution weights at finalize.  The 1x1 convolutions are rows x channels GEMMs and run on the generic
// MFMA GEMM kernel (gemm.hip); here: the 3x3 stem, the depthwise 3x3 convolutions and the two pools — pure streaming
// kernels (each output element reads 9 inputs), one thread per 4 channels of an output pixel, 8/16-byte coalesced
// accesses along the channel axis.
#include "dhw_common.h"
#include "dhw_kernels.h"

namespace {

DHW_DEV float relu6(float x) { return fminf(fmaxf(x, 0.f), 6.f); }

// features.0: Conv2d(3 -> 32, k=3, s=2, p=1) on x = img / 127.5 - 1 replicated over the 3 input channels
// (text_style.py:51-52) => a 1-channel convolution with the kernel summed over its input channels (w: [9][Cp], fp32).
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, int B, int H, int W, int Ho, int Wo,
                                                    const float* __restrict__ w, const float* __restrict__ bias, int Cp,
                                                    T* __restrict__ out) {
  const int c4 = Cp / 4;
  const long total = (long)B * Ho * Wo * c4;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int c = (int)(id % c4) * 4;
  const long pix = id / c4;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
  f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int y = 2 * yo - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int x = 2 * xo - 1 + kx;
      if (y < 0 || y >= H || x < 0 || x >= W) continue;   // zero padding of the NORMALISED image
      const float v = img[((long)b * H + y) * W + x] / 127.5f - 1.0f;
      acc += v * *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * Cp + c);
    }
  }
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = relu6(acc[k]);
  store4(out + pix * Cp + c, r);
}

// depthwise Conv2d(C -> C, k=3, stride s, p=1, groups=C) + folded BN + ReLU6; w: [9][Cp] fp32
template <typename T>
__global__ __launch_bounds__(256) void dw_kernel(const T* __restrict__ in, int B, int H, int W, int Ho, int Wo, int stride,
                                                  const float* __restrict__ w, const float* __restrict__ bias, int Cp,
                                                  T* __restrict__ out) {
  const int c4 = Cp / 4;
  const long total = (long)B * Ho * Wo * c4;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int c = (int)(id % c4) * 4;
  const long pix = id / c4;
  const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
  f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int y = stride * yo - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int x = stride * xo - 1 + kx;
      if (y < 0 || y >= H || x < 0 || x >= W) continue;
      acc += load4(in + (((long)b * H + y) * W + x) * Cp + c) * *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * Cp + c);
    }
  }
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = relu6(acc[k]);
  store4(out + pix * Cp + c, r);
}

// AvgPool2d(kernel 3, stride 3) then AdaptiveAvgPool2d((1, NB)) then squeeze / permute (text_style.py:55-58):
// out[b][j][c] = mean over hp < Hp, wp in [floor(j Wp / NB), ceil((j+1) Wp / NB)) of mean_{3x3}(in[b][3hp.., 3wp..][c]), fp32.
template <typename T>
__global__ __launch_bounds__(256) void style_pool_kernel(const T* __restrict__ in, int B, int H, int W, int C, int NB,
                                                          float* __restrict__ out) {
  const int c4 = C / 4;
  const long total = (long)B * NB * c4;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int c = (int)(id % c4) * 4;
  const int j = (int)((id / c4) % NB), b = (int)(id / ((long)c4 * NB));
  const int Hp = H / 3, Wp = W / 3;
  const int w0 = (j * Wp) / NB, w1 = ((j + 1) * Wp + NB - 1) / NB;
  f32x4 sum = (f32x4){0, 0, 0, 0};
  for (int hp = 0; hp < Hp; ++hp)
    for (int wp = w0; wp < w1; ++wp) {
      f32x4 s = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) s += load4(in + (((long)b * H + 3 * hp + dy) * W + 3 * wp + dx) * C + c);
      sum += s * (1.0f / 9.0f);
    }
  const float inv = 1.0f / (float)(Hp * (w1 - w0));
  *reinterpret_cast<f32x4*>(out + ((long)b * NB + j) * C + c) = sum * inv;
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t launch_style_stem(int prec, const float* img, int B, int H, int W, const float* w, const float* bias, int Cp,
                             void* out, hipStream_t st) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * Ho * Wo * (Cp / 4);
  if (prec == PREC_BF16) hipLaunchKernelGGL(stem_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, st, img, B, H, W, Ho, Wo, w, bias, Cp, (bf16_t*)out);
  else hipLaunchKernelGGL(stem_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, img, B, H, W, Ho, Wo, w, bias, Cp, (float*)out);
  return hipGetLastError();
}

hipError_t launch_style_dw(int prec, const void* in, int B, int H, int W, int stride, const float* w, const float* bias,
                           int Cp, void* out, hipStream_t st) {
  if (stride != 1 && stride != 2) return hipErrorInvalidValue;
  const int Ho = stride == 2 ? (H + 1) / 2 : H, Wo = stride == 2 ? (W + 1) / 2 : W;
  const long total = (long)B * Ho * Wo * (Cp / 4);
  if (prec == PREC_BF16) hipLaunchKernelGGL(dw_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, st, (const bf16_t*)in, B, H, W, Ho, Wo, stride, w, bias, Cp, (bf16_t*)out);
  else hipLaunchKernelGGL(dw_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, (const float*)in, B, H, W, Ho, Wo, stride, w, bias, Cp, (float*)out);
  return hipGetLastError();
}

hipError_t launch_style_pool(int prec, const void* in, int B, int H, int W, int C, int NB, float* out, hipStream_t st) {
  if (H < 3 || W < 3) return hipErrorInvalidValue;
  const long total = (long)B * NB * (C / 4);
  if (prec == PREC_BF16) hipLaunchKernelGGL(style_pool_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, st, (const bf16_t*)in, B, H, W, C, NB, out);
  else hipLaunchKernelGGL(style_pool_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, (const float*)in, B, H, W, C, NB, out);
  return hipGetLastError();
}
