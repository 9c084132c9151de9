// convblock.hip — one whole ConvBlock (reference cnn.py:64-87) as a single kernel:
//
//     out = FiLM3(fc(SiLU(FiLM2(conv2(SiLU(FiLM1(conv1(SiLU(x))))))))) + conv_skip(x)
//
// A workgroup owns BM-2 consecutive stroke rows of one sample.  x (raw and SiLU'd, +2 halo rows each
// side) is staged into LDS once; conv1 -> h1 and conv2 -> h2 never leave LDS; the k=3 taps of every
// conv read row-shifted views of the staged tiles; fc and conv_skip accumulate into the same MFMA
// accumulators with the FiLM3 affine applied in between.  Weights stream from L2 in MFMA-fragment
// order (gemm_core.h).  Replaces three launches and two HBM/L2 round trips of h1/h2.
#include "gemm_core.h"
#include "dhw_kernels.h"

namespace {

template <typename T, int BM, int CO>
__global__ __launch_bounds__(256) void convblock_kernel(const ConvBlockParams p) {
  constexpr int ES = sizeof(T);
  constexpr int BMO = BM - 2;            // output rows per workgroup
  constexpr int RX = BM + 2;             // staged x rows: sample rows [m0-2, m0+BM)
  constexpr int C1 = CO / 2;             // conv1 output channels
  constexpr int MT1 = BM / 2 / 16, NT1 = C1 / 2 / 16;   // stage 1: waves 2 (rows) x 2 (channels)
  constexpr int MT2 = BM / 16, NT2 = CO / 4 / 16;       // stages 2,3: waves 1 x 4
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int tiles = (p.L + BMO - 1) / BMO;
  const int b = blockIdx.x / tiles;
  const int m0 = (blockIdx.x % tiles) * BMO;
  const int Cin = p.Cin;

  const int SX = tile_stride<T>(Cin), SH1 = tile_stride<T>(C1), SH2 = tile_stride<T>(CO);
  char* XS = smem;                       // SiLU(x)   [RX][Cin]
  char* XR = XS + RX * SX;               // x         [RX][Cin]
  char* H1 = XR + RX * SX;               // h1        [BM+2][C1]  (index i <-> sample row m0-1+i)
  char* H2 = H1 + (BM + 2) * SH1;        // h2        [BM][CO]    (index i <-> sample row m0+i)

  // ---- stage 0: x tile -> LDS (raw + SiLU), zero outside the sample ('same' padding)
  {
    const int cpr = Cin * ES / 16;
    const int total = RX * cpr;
    const char* src = reinterpret_cast<const char*>(p.x);
    constexpr int U = 4;
    for (int base = tid; base < total; base += 256 * U) {
      uint4 v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * 256;
        const int r = id / cpr, cc = id - r * cpr;
        const int lrow = m0 - 2 + r;
        v[u] = make_uint4(0, 0, 0, 0);
        dst[u] = id < total ? r * SX + cc * 16 : -1;
        if (id < total && lrow >= 0 && lrow < p.L)
          v[u] = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.L + lrow) * Cin) * ES + (size_t)cc * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (dst[u] < 0) continue;
        *reinterpret_cast<uint4*>(XR + dst[u]) = v[u];
        T* e = reinterpret_cast<T*>(&v[u]);
#pragma unroll
        for (int i = 0; i < 16 / ES; ++i) e[i] = from_f<T>(silu_f(to_f(e[i])));
        *reinterpret_cast<uint4*>(XS + dst[u]) = v[u];
      }
    }
    // the two h1 rows past the computed BM (read only by the discarded output rows) must be finite
    for (int id = tid; id < 2 * SH1 / 16; id += 256)
      *reinterpret_cast<uint4*>(H1 + BM * SH1 + id * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const float* gam = p.film + (size_t)b * p.film_bs;
  const float* bet = gam + p.film_tot;

  // ---- stage 1: h1 = SiLU(FiLM1(conv1(SiLU(x)))) for sample rows [m0-1, m0-1+BM)
  {
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = wm * (BM / 2), ntile0 = wn * NT1;
    f32x4 acc[NT1][MT1];
    acc_zero(acc);
    const int KC = Cin / 32;
    const T* wbase = reinterpret_cast<const T*>(p.w_c1) + ((size_t)ntile0 * KC * 3 * 64 + lane) * 8;
    mainloop<T, MT1, NT1>(acc, wbase, XS + (row0 + l15) * SX + g * 8 * ES, SX, KC, 3);
#pragma unroll
    for (int i = 0; i < NT1; ++i) {
      const int n = (ntile0 + i) * 16 + 4 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b_c1 + n);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + p.f1 + n);
      const f32x4 be = *reinterpret_cast<const f32x4*>(bet + p.f1 + n);
#pragma unroll
      for (int j = 0; j < MT1; ++j) {
        const int r = row0 + j * 16 + l15;
        const int srow = m0 - 1 + r;
        f32x4 v = (acc[i][j] + bi) * ga + be;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (srow >= 0 && srow < p.L) ? silu_f(v[k]) : 0.f;   // conv2 pads h1 with zeros
        store4(reinterpret_cast<T*>(H1 + r * SH1) + n, v);
      }
    }
  }
  __syncthreads();

  const int ntile0 = wave * NT2;
  // ---- stage 2: h2 = SiLU(FiLM2(conv2(h1))) for sample rows [m0, m0+BM) (the last 2 are discarded)
  {
    f32x4 acc[NT2][MT2];
    acc_zero(acc);
    const int KC = C1 / 32;
    const T* wbase = reinterpret_cast<const T*>(p.w_c2) + ((size_t)ntile0 * KC * 3 * 64 + lane) * 8;
    mainloop<T, MT2, NT2>(acc, wbase, H1 + l15 * SH1 + g * 8 * ES, SH1, KC, 3);
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
      const int n = (ntile0 + i) * 16 + 4 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b_c2 + n);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + p.f2 + n);
      const f32x4 be = *reinterpret_cast<const f32x4*>(bet + p.f2 + n);
#pragma unroll
      for (int j = 0; j < MT2; ++j) {
        f32x4 v = (acc[i][j] + bi) * ga + be;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu_f(v[k]);
        store4(reinterpret_cast<T*>(H2 + (j * 16 + l15) * SH2) + n, v);
      }
    }
  }
  __syncthreads();

  // ---- stage 3: out = FiLM3(fc(h2)) + conv_skip(x)
  {
    f32x4 acc[NT2][MT2];
    acc_zero(acc);
    {
      const int KC = CO / 32;
      const T* wbase = reinterpret_cast<const T*>(p.w_fc) + ((size_t)ntile0 * KC * 64 + lane) * 8;
      mainloop<T, MT2, NT2>(acc, wbase, H2 + l15 * SH2 + g * 8 * ES, SH2, KC, 1);
    }
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
      const int n = (ntile0 + i) * 16 + 4 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b_fc + n);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + p.f3 + n);
      const f32x4 be = *reinterpret_cast<const f32x4*>(bet + p.f3 + n);
#pragma unroll
      for (int j = 0; j < MT2; ++j) acc[i][j] = (acc[i][j] + bi) * ga + be;
    }
    {
      const int KC = Cin / 32;
      const T* wbase = reinterpret_cast<const T*>(p.w_skip) + ((size_t)ntile0 * KC * 3 * 64 + lane) * 8;
      mainloop<T, MT2, NT2>(acc, wbase, XR + (l15 + 1) * SX + g * 8 * ES, SX, KC, 3);   // out row i <- x rows i+1+tap
    }
#pragma unroll
    for (int i = 0; i < NT2; ++i) {
      const int n = (ntile0 + i) * 16 + 4 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b_skip + n);
#pragma unroll
      for (int j = 0; j < MT2; ++j) {
        const int r = j * 16 + l15;
        const int srow = m0 + r;
        const bool valid = r < BMO && srow < p.L;
        const f32x4 v = acc[i][j] + bi;
        if (valid) {
          const size_t o = (size_t)(b * p.L + srow) * CO + n;
          if (p.out_f32) store4(reinterpret_cast<float*>(p.out) + o, v);
          else store4(reinterpret_cast<T*>(p.out) + o, v);
        }
        if (p.pool) {   // AvgPool1d(2) side output (model.py:93): rows 2i, 2i+1 are lanes l, l^1
          f32x4 q;
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = 0.5f * (v[k] + __shfl_xor(v[k], 1));
          if (valid && !(lane & 1))
            store4(reinterpret_cast<T*>(p.pool) + ((size_t)b * (p.L / 2) + (srow >> 1)) * CO + n, q);
        }
      }
    }
  }
}

template <typename T, int BM, int CO>
size_t lds_bytes(int Cin) {
  return (size_t)2 * (BM + 2) * tile_stride<T>(Cin) + (size_t)(BM + 2) * tile_stride<T>(CO / 2) + (size_t)BM * tile_stride<T>(CO);
}

template <typename T, int BM, int CO>
hipError_t launch_t(const ConvBlockParams& p, hipStream_t st) {
  const size_t lds = lds_bytes<T, BM, CO>(p.Cin);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int tiles = (p.L + BM - 3) / (BM - 2);
  hipLaunchKernelGGL((convblock_kernel<T, BM, CO>), dim3(p.B * tiles), dim3(256), lds, st, p);
  return hipGetLastError();
}

template <typename T, int BM, int CO>
hipError_t attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(convblock_kernel<T, BM, CO>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

hipError_t convblock_init() {
  hipError_t e;
  if ((e = attr<bf16_t, 64, 128>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 64, 192>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 64, 256>()) != hipSuccess) return e;
  if ((e = attr<float, 32, 128>()) != hipSuccess) return e;
  if ((e = attr<float, 32, 192>()) != hipSuccess) return e;
  return attr<float, 32, 256>();
}

hipError_t launch_convblock(int prec, const ConvBlockParams& p, hipStream_t st) {
  if (p.Cin % 32 || (p.L & 1)) return hipErrorInvalidValue;
  if (prec == PREC_BF16) {
    switch (p.Cout) {
      case 128: return launch_t<bf16_t, 64, 128>(p, st);
      case 192: return launch_t<bf16_t, 64, 192>(p, st);
      case 256: return launch_t<bf16_t, 64, 256>(p, st);
    }
  } else {
    switch (p.Cout) {
      case 128: return launch_t<float, 32, 128>(p, st);
      case 192: return launch_t<float, 32, 192>(p, st);
      case 256: return launch_t<float, 32, 256>(p, st);
    }
  }
  return hipErrorInvalidValue;
}
