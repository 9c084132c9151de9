// gemm_core.h — building blocks shared by the fused block kernels (convblock.hip, enclayer.hip):
// the activation-stationary MFMA main loop with a register prefetch ring for the weight stream,
// LDS tile geometry, and the cross-wave LayerNorm.
#pragma once
#include "dhw_common.h"

// An activation tile in LDS: `rows` rows of C elements, row stride padded by 16 bytes so the
// 16-lane fragment reads (row = lane&15, 16-byte column slot = lane>>4) spread over the banks.
template <typename T> __host__ __device__ inline int tile_stride(int C) { return C * (int)sizeof(T) + 16; }

// acc[i][j] += W(16 x 32*KT, channel tile i) * Act^T(32*KT x 16, row tile j)
//   wbase : packed weights of this wave's first channel tile, already offset by lane*8 elements;
//           fragment (i, kt) sits at wbase + (i*KT + kt)*512
//   abase : LDS address of (this wave's first row + lane&15, element (lane>>4)*8) of the activation
//           tile, including any row offset; tap t reads `stride` bytes further per tap
//   KC    : k-chunks (of 32 channels) per tap;  taps: 1 or 3
//   KTS   : fragment stride between consecutive channel tiles of the packed matrix (0 = KC*taps); lets a
//           caller contract over a K-slice [kt0, kt0+KC) of a wider matrix (wbase advanced by kt0*512)
// Weight fragments stream L2 -> VGPRs through a D-deep register ring; the body is branch-free and
// statically indexed so hipcc emits counted s_waitcnt vmcnt((D-1)*NT) instead of draining the queue.
template <typename T, int MT, int NT, int RING = (sizeof(T) == 2 ? 24 : 12)>
DHW_DEV void mainloop(f32x4 (&acc)[NT][MT], const T* __restrict__ wbase, const char* abase, int stride, int KC, int taps,
                      int KTS = 0) {
  constexpr int ES = sizeof(T);
  constexpr int D0 = RING / NT;                     // ring depth: ~RING weight fragments (1 KiB each) in flight per wave
  constexpr int D = D0 < 2 ? 2 : (D0 > 8 ? 8 : D0);
  const int KT = KC * taps;
  if (KTS == 0) KTS = KT;
  Frag<T> wq[D][NT];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int k = d < KT ? d : KT - 1;
#pragma unroll
    for (int i = 0; i < NT; ++i) wq[d][i] = frag_load(wbase + ((size_t)i * KTS + k) * 512);
  }
  int aoff = 0, kc = 0;
  const int tap_step = stride - (KC - 1) * 32 * ES;
  auto step = [&](int d, int knext) {
    Frag<T> a[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) a[j] = frag_load(reinterpret_cast<const T*>(abase + j * 16 * stride + aoff));
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) mma32(acc[i][j], wq[d][i], a[j]);
    const int kn = knext < KT ? knext : KT - 1;
#pragma unroll
    for (int i = 0; i < NT; ++i) wq[d][i] = frag_load(wbase + ((size_t)i * KTS + kn) * 512);
    const bool wrap = ++kc == KC;
    aoff += wrap ? tap_step : 32 * ES;
    kc = wrap ? 0 : kc;
  };
  int kt = 0;
  for (; kt + D <= KT; kt += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) step(d, kt + d + D);
  }
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (kt + d < KT) step(d, KT - 1);
}

template <int NT, int MT>
DHW_DEV void acc_zero(f32x4 (&acc)[NT][MT]) {
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
}

// LayerNorm (eps 1e-6, no affine; reference model.py:25) over the N channels of each row of a tile whose
// channels are split over WN waves (each holding NT tiles of 16) — two-pass, fp32.
// red: LDS scratch of 2*WN*ROWS floats.  Row of (j, lane): row0 + j*16 + (lane&15).  Contains barriers.
template <int MT, int NT, int WN, int ROWS>
DHW_DEV void layernorm_rows(f32x4 (&acc)[NT][MT], float* red, int wn, int row0, int lane, int N) {
  const int l15 = lane & 15, g = lane >> 4;
  float mean[MT], rstd[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0) red[wn * ROWS + row0 + j * 16 + l15] = s;
  }
  __syncthreads();
  const float invn = 1.0f / (float)N;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) s += red[w * ROWS + row0 + j * 16 + l15];
    mean[j] = s * invn;
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[j]; s += d * d; }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0) red[(WN + wn) * ROWS + row0 + j * 16 + l15] = s;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) s += red[(WN + w) * ROWS + row0 + j * 16 + l15];
    rstd[j] = rsqrtf(s * invn + 1e-6f);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (acc[i][j] - mean[j]) * rstd[j];
}
