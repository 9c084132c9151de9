// rowgemm.hip — persistent, software-pipelined form of the fused GEMM for LONG row lists: the all-steps text plane
// (reference text_style.py:91-104 and the text half of model.py:37-58 evaluated for all T steps at once) is a dozen
// Linear layers over B*T*Lt ~ 10^5 independent rows.  One launch of the tile-per-workgroup kernel (gemm.hip) spends
// most of a workgroup's life in latency chains — activation tile from HBM (3-8 us under load), weight stream start,
// residual loads, output drain — with nothing else to run on the CU.  Here a workgroup (8 waves, one per CU) walks its
// row tiles in a loop: the NEXT tile's activations are already in flight (registers -> second LDS buffer) while the
// current tile is multiplied, the residual operand is requested before the main loop, and the weights (L2-resident,
// 0.1-0.6 MB) stream through the same register ring as everywhere else (gemm_core.h).
//
// Same epilogue vocabulary as gemm.hip (bias, PE.W position bias, residual, LayerNorm, sigma-FiLM, SiLU, transposed-V
// side output); rows are flat (tile boundaries ignore sample boundaries), the sample of a row is row / L.
#include <algorithm>
#include <cstdlib>
#include "gemm_core.h"
#include "dhw_kernels.h"

namespace {

#define RSTAMP(slot)                                                                                                   \
  do {                                                                                                                 \
    if (p.stamps && blockIdx.x == gridDim.x / 2 && t == (int)blockIdx.x + 2 * (int)gridDim.x && threadIdx.x == 0)    \
      p.stamps[slot] = __builtin_amdgcn_s_memrealtime();                                                               \
  } while (0)

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void rowgemm_kernel(const GemmParams p, int tiles) {
  static_assert(WM * WN == 8, "8 waves");
  constexpr int ES = sizeof(T), NTHR = 512;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
  constexpr int SOT = BN * ES + 16, SVT = BM * ES + 16;
  constexpr int APT = 6;   // 16-byte activation pieces per thread and tile (BM * K <= 24576 elements, checked by the launcher)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = p.seg[0].C, KC = K / 32, SA = K * ES + 16;
  const int M = p.B * p.L;
  char* A0 = smem;
  char* A1 = smem + BM * SA;
  char* OT = smem + 2 * BM * SA;                       // output tile: [BM][BN], or [BN][BM] for the transposed-V columns
  // (LayerNorm / FiLM scratch sits behind the row-major tile: launches with transposed-V columns have neither)
  float* red = reinterpret_cast<float*>(OT + BM * SOT);   // LayerNorm partial sums [2][WN][BM]
  float* FC = red + 2 * WN * BM;                       // sigma-FiLM rows of the current tiles: [2 candidates][gamma | beta][BN]
  const int row0 = wm * (BM / WM);
  const int cpr = K * ES / 16, atotal = BM * cpr;
  const char* asrc = reinterpret_cast<const char*>(p.seg[0].A);
  const bool silu_in = p.seg[0].silu != 0;

  uint4 apre[APT];
  auto a_issue = [&](int t) {
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      const int id = tid + u * NTHR;
      const int r = id / cpr, cc = id - r * cpr;
      const long grow = (long)t * BM + r;
      apre[u] = make_uint4(0, 0, 0, 0);
      if (id < atotal && grow < M) apre[u] = *reinterpret_cast<const uint4*>(asrc + (size_t)grow * K * ES + (size_t)cc * 16);
    }
  };
  auto a_commit = [&](char* dst) {
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      const int id = tid + u * NTHR;
      const int r = id / cpr, cc = id - r * cpr;
      uint4 v = apre[u];
      if (silu_in) {
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int i = 0; i < 16 / ES; ++i) e[i] = from_f<T>(silu_t<T>(to_f(e[i])));
      }
      if (id < atotal) *reinterpret_cast<uint4*>(dst + r * SA + cc * 16) = v;
    }
  };

  int t = blockIdx.x;
  int cur = 0;
  int fc_lo = -1, fc_hi = -1;   // FiLM rows held in FC
  if (t < tiles) {
    a_issue(t);
    a_commit(A0);
  }
  lds_barrier();
  for (; t < tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < tiles) a_issue(tn);   // lands while this tile is multiplied (measured better than issuing it behind the main loop)
    const char* Acur = cur ? A1 : A0;
    const int r0 = t * BM;
    // sigma-FiLM (LayerNorm blocks, BN == N): a tile spans at most two FiLM rows (the launcher checks film_div * L >= BM);
    // they are kept in LDS and re-read from global only when they change (every film_div * L rows).
    const int f_lo = (r0 / p.L) / p.film_div, f_hi = ((min(r0 + BM, M) - 1) / p.L) / p.film_div;
    if (p.film_mode == 1 && (f_lo != fc_lo || f_hi != fc_hi)) {
      for (int id = tid; id < 4 * BN; id += NTHR) {
        const int which = id / (2 * BN), rem = id - which * 2 * BN, gb = rem / BN, n = rem - gb * BN;
        FC[id] = (gb ? p.bet : p.gam)[(long)(which ? f_hi : f_lo) * p.film_bs + n];
      }
      fc_lo = f_lo; fc_hi = f_hi;
      lds_barrier();
    }
    int sb[MT], lr[MT];            // sample and row-in-sample of this lane's rows
    bool valid[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int r = r0 + row0 + j * 16 + l15;
      valid[j] = r < M;
      const int rr = valid[j] ? r : M - 1;
      sb[j] = rr / p.L;
      lr[j] = rr - sb[j] * p.L;
    }
    RSTAMP(0);
    for (int nb = 0; nb < p.N; nb += BN) {
      const int ntile0 = (nb + wn * (BN / WN)) / 16;
      WRing<T, NT, (MT * NT >= 12 ? 9 : 12)> ring;   // (fewer fragments in flight for the widest tile: VGPR budget)
      ring.fill(reinterpret_cast<const T*>(p.seg[0].W) + ((size_t)ntile0 * KC * 64 + lane) * 8, KC);
      f32x4 bias[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) bias[i] = *reinterpret_cast<const f32x4*>(p.bias0 + (ntile0 + i) * 16 + 4 * g);
      uint2 res[NT][MT];   // 4 packed bf16 each
      if (p.res1) {   // requested before the main loop, consumed after it
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const long r = (long)r0 + row0 + j * 16 + l15;
            res[i][j] = valid[j] ? *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(p.res1) + (size_t)r * p.N + (ntile0 + i) * 16 + 4 * g)
                                 : make_uint2(0, 0);
          }
      }
      f32x4 acc[NT][MT];
      acc_zero(acc);
      ring.template run<MT>(acc, Acur + (row0 + l15) * SA + g * 8 * ES, SA, KC);
      // the next tile's activations are requested HERE: behind every weight fragment this tile waits for (a wave's loads
      // complete in order), ahead of an epilogue that loads nothing else
      if (nb == 0) RSTAMP(1);
      if (nb + BN >= p.N && tn < tiles) a_issue(tn);

#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = (ntile0 + i) * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          f32x4 v = acc[i][j] + bias[i];
          if (p.posb && n < p.posb_cols) v += *reinterpret_cast<const f32x4*>(p.posb + (size_t)lr[j] * p.posb_cols + n);
          if (p.res1) {
            const uint2 q = res[i][j];
            v += (f32x4){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
          }
          acc[i][j] = v;
        }
      }
      if (p.ln) layernorm_rows<MT, NT, WN, BM>(acc, red, wn, row0, lane, BN);   // (the launcher guarantees BN == N)

      const bool vblock = nb >= p.n_store;
      if (nb == 0) RSTAMP(2);
      lds_barrier();   // the previous block's / tile's output tile has been copied out by every wave
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = (ntile0 + i) * 16 + 4 * g, nl = n - nb;
        f32x4 ga0, be0, ga1, be1;
        if (p.film_mode == 1) {
          ga0 = *reinterpret_cast<const f32x4*>(FC + nl); be0 = *reinterpret_cast<const f32x4*>(FC + BN + nl);
          ga1 = *reinterpret_cast<const f32x4*>(FC + 2 * BN + nl); be1 = *reinterpret_cast<const f32x4*>(FC + 3 * BN + nl);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          f32x4 v = acc[i][j];
          if (p.film_mode == 1) {
            const bool lo = sb[j] / p.film_div == f_lo;
            v = v * (lo ? ga0 : ga1) + (lo ? be0 : be1);
          }
          if (p.silu_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
          }
          const int rl = row0 + j * 16 + l15;
          if (vblock) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<T*>(OT + (nl + r) * SVT + rl * ES) = from_f<T>(valid[j] ? v[r] : 0.f);
          } else {
            store4(reinterpret_cast<T*>(OT + rl * SOT) + nl, v);
          }
        }
      }
      lds_barrier();
      if (nb == 0) RSTAMP(3);
      const int rows_valid = min(BM, M - r0);
      if (!vblock) {
        tile_copy_out<T>(OT, SOT, reinterpret_cast<T*>(p.out) + (size_t)r0 * p.n_store + nb, p.n_store, rows_valid, BN, tid, NTHR);
      } else {
        // transposed-V columns: [sample][channel][key], written per sample segment in 2-key pieces (L and r0 are even);
        // the padding keys [L, lpad) of every row stay as allocated (zero)
        const int NV = p.N - p.n_store, chb = nb - p.n_store;
        constexpr int PPR = BM / 2;
        for (int id = tid; id < BN * PPR; id += NTHR) {
          const int ch = id / PPR, pp = id - ch * PPR;
          const int r = r0 + 2 * pp;
          if (r < M) {
            const int b = r / p.L, key = r - b * p.L;
            *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.vt) + ((size_t)b * NV + chb + ch) * p.vt_lpad + key) =
                *reinterpret_cast<const uint32_t*>(OT + ch * SVT + pp * 4);
          }
        }
      }
    }
    RSTAMP(4);
    if (tn < tiles) a_commit(cur ? A0 : A1);
    lds_barrier();
    RSTAMP(5);
    cur ^= 1;
  }
}

template <typename T, int BM, int BN, int WM, int WN>
size_t lds_need(int K) {
  constexpr size_t SOT = BN * sizeof(T) + 16, SVT = BM * sizeof(T) + 16;
  return (size_t)2 * BM * (K * sizeof(T) + 16) + std::max((size_t)BM * SOT + 2 * WN * BM * sizeof(float) + 4 * BN * sizeof(float), (size_t)BN * SVT);
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t launch_t(const GemmParams& p, hipStream_t st) {
  const int K = p.seg[0].C;
  const size_t lds = lds_need<T, BM, BN, WM, WN>(K);
  if (lds > 160 * 1024 || (long)BM * K > 24576) return hipErrorInvalidValue;
  const long M = (long)p.B * p.L;
  const int tiles = (int)((M + BM - 1) / BM);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidValue;
    cus = prop.multiProcessorCount;
  }
  hipLaunchKernelGGL((rowgemm_kernel<T, BM, BN, WM, WN>), dim3(std::min(tiles, cus)), dim3(512), lds, st, p, tiles);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_kernel<T, BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int pick_bn(const GemmParams& p) {
  for (int bn : {384, 256, 192})
    if (p.N % bn == 0 && p.n_store % bn == 0 && (!p.ln || p.N == bn)) return bn;
  return 0;
}

}  // namespace

hipError_t rowgemm_init() {
  hipError_t e;
  if ((e = attr<bf16_t, 64, 384, 1, 8>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 64, 256, 1, 8>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 64, 192, 2, 4>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 32, 384, 1, 8>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 32, 256, 1, 8>()) != hipSuccess) return e;
  return attr<bf16_t, 32, 192, 2, 4>();
}

// Long row lists only: one Linear segment, bf16, no k=3 taps / upsampled residual / pool / fp32 output.
bool rowgemm_supported(int prec, const GemmParams& p) {
  if (prec != PREC_BF16 || p.nseg != 1 || p.seg[0].taps != 1 || p.res2 || p.pool || p.out_f32 || p.film_mode == 2) return false;
  const int K = p.seg[0].C;
  if (K % 32 || K > 768 || !pick_bn(p) || p.film_div < 1) return false;
  if (p.n_store < p.N && ((p.L & 1) || !p.vt || p.ln || p.film_mode)) return false;
  if (p.film_mode == 1 && ((long)p.film_div * p.L < 64 || !p.ln)) return false;   // FiLM: LayerNorm blocks only; a tile spans <= two FiLM rows
  // measured on the all-steps text plane (tools/bench_text.cpp): ahead of the tile-per-workgroup kernel for the plain and
  // transposed-V projections (ts.kv 550 -> 356 us), level or behind for the LayerNorm blocks -> those stay on gemm.hip
  if (p.ln || K < 256) return false;
  return (long)p.B * p.L >= 16384;
}

hipError_t launch_rowgemm(int prec, const GemmParams& p, hipStream_t st) {
  if (!rowgemm_supported(prec, p)) return hipErrorInvalidValue;
  const int bn = pick_bn(p);
  if (p.seg[0].C <= 384) {
    switch (bn) {
      case 384: return launch_t<bf16_t, 64, 384, 1, 8>(p, st);
      case 256: return launch_t<bf16_t, 64, 256, 1, 8>(p, st);
      case 192: return launch_t<bf16_t, 64, 192, 2, 4>(p, st);
    }
  } else {
    switch (bn) {
      case 384: return launch_t<bf16_t, 32, 384, 1, 8>(p, st);
      case 256: return launch_t<bf16_t, 32, 256, 1, 8>(p, st);
      case 192: return launch_t<bf16_t, 32, 192, 2, 4>(p, st);
    }
  }
  return hipErrorInvalidValue;
}
