// rowgemm.hip — the fused GEMM for LONG flat row lists: the all-steps text plane (reference text_style.py:91-104 and the
// text half of model.py:37-58, evaluated for all T steps at once) is a dozen Linear layers over B*T*Lt ~ 10^5
// independent rows.  The tile-per-sample kernel (gemm.hip) gives a workgroup 30 rows of one (sample, step): every
// 30 rows re-stream the whole weight matrix through the CU's L1 path and the main loop sits at 40 % of the MFMA rate
// with nothing to hide the rest behind.  Here a workgroup (8 waves) takes 128 rows of the flat list (64 when K > 384)
// against all N columns: 24 MFMAs per wave and k-chunk for 3 KB of weights, so the stream needs only a 4-deep ring and
// the main loop runs MFMA-bound; the output goes out in two 64-row halves through one LDS staging tile.
//
// Same epilogue vocabulary as gemm.hip (bias, PE.W position bias, residual, LayerNorm, sigma-FiLM, SiLU, transposed-V
// side output) and the SAME per-element arithmetic as its 8-wave variants: wave w owns channels [w*BN/8, (w+1)*BN/8) of
// a block, the k-walk and the LayerNorm partial sums are grouped identically, so which of the two kernels a launch
// gets (it depends on the row count) never changes a sample bit — tests/test_gpu_parity.py checks that.
#include <algorithm>
#include <cstdlib>
#include "gemm_core.h"
#include "dhw_kernels.h"

namespace {

#define RSTAMP(slot)                                                                                                 \
  do {                                                                                                               \
    if (p.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

template <typename T, int BM, int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void rowgemm_kernel(const GemmParams p) {
  constexpr int ES = sizeof(T), NTHR = 512, WN = 8;
  constexpr int MT = BM / 16, NT = BN / WN / 16, HALF = BM / 2, MH = MT / 2;
  constexpr int SOT = BN * ES + 16, SVT = HALF * ES + 16;
  static_assert(NT * WN * 16 == BN && MH * 2 == MT, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = p.seg[0].C, KC = K / 32, SA = K * ES + 16;
  const long M = (long)p.B * p.L;
  char* AT = smem;                                  // activation tile [BM][K]
  char* OT = smem + BM * SA;                        // half an output tile: [HALF][BN], or [BN][HALF] for transposed-V columns
  float* red = reinterpret_cast<float*>(OT);        // LayerNorm partial sums [2][WN][BM]: dead before the output tile is written
  constexpr int FCO = HALF * SOT > 2 * WN * BM * 4 ? HALF * SOT : 2 * WN * BM * 4;
  float* FC = reinterpret_cast<float*>(OT + FCO);   // FiLM rows [2 candidates][gamma | beta][BN]
  const long r0 = (long)blockIdx.x * BM;

  RSTAMP(0);
  // ---- stage the activation tile (SiLU prologue optional); zero past the last row
  {
    const int cpr = K * ES / 16;
    const char* asrc = reinterpret_cast<const char*>(p.seg[0].A);
    const bool silu_in = p.seg[0].silu != 0;
    constexpr int U = 6;
    for (int base = tid; base < BM * cpr; base += NTHR * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * NTHR;
        const int r = id / cpr, cc = id - r * cpr;
        v[u] = make_uint4(0, 0, 0, 0);
        if (id < BM * cpr && r0 + r < M) v[u] = *reinterpret_cast<const uint4*>(asrc + (size_t)(r0 + r) * K * ES + (size_t)cc * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * NTHR;
        const int r = id / cpr, cc = id - r * cpr;
        if (silu_in) {
          T* e = reinterpret_cast<T*>(&v[u]);
#pragma unroll
          for (int i = 0; i < 16 / ES; ++i) e[i] = from_f<T>(silu_t<T>(to_f(e[i])));
        }
        if (id < BM * cpr) *reinterpret_cast<uint4*>(AT + r * SA + cc * 16) = v[u];
      }
    }
  }
  // sigma-FiLM (LayerNorm launches, one column block): a tile spans at most two FiLM rows (launcher: film_div * L >= BM)
  const long rlast = (r0 + BM < M ? r0 + BM : M) - 1;
  const int f_lo = (int)((r0 / p.L) / p.film_div), f_hi = (int)((rlast / p.L) / p.film_div);
  if (p.film_mode == 1) {
    for (int id = tid; id < 4 * BN; id += NTHR) {
      const int which = id / (2 * BN), rem = id - which * 2 * BN, gb = rem / BN, n = rem - gb * BN;
      FC[id] = (gb ? p.bet : p.gam)[(long)(which ? f_hi : f_lo) * p.film_bs + n];
    }
  }
  lds_barrier();
  RSTAMP(1);

  int sb[MT], lr[MT];   // sample and row-in-sample of this lane's rows
  bool valid[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const long r = r0 + j * 16 + l15;
    valid[j] = r < M;
    const long rr = valid[j] ? r : M - 1;
    sb[j] = (int)(rr / p.L);
    lr[j] = (int)(rr - (long)sb[j] * p.L);
  }

  for (int nb = 0; nb < p.N; nb += BN) {
    const int ntile0 = (nb + wn * (BN / WN)) / 16;
    f32x4 acc[NT][MT];
    acc_zero(acc);
    mainloop<T, MT, NT, (MT * NT >= 24 ? 9 : 12)>(acc, reinterpret_cast<const T*>(p.seg[0].W) + ((size_t)ntile0 * KC * 64 + lane) * 8,
                            AT + l15 * SA + g * 8 * ES, SA, KC, 1);
    if (nb == 0) RSTAMP(2);

#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = (ntile0 + i) * 16 + 4 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(p.bias0 + n);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        f32x4 v = acc[i][j] + bi;
        if (valid[j]) {
          if (p.posb && n < p.posb_cols) v += *reinterpret_cast<const f32x4*>(p.posb + (size_t)lr[j] * p.posb_cols + n);
          if (p.res1) v += load4(reinterpret_cast<const T*>(p.res1) + (size_t)(r0 + j * 16 + l15) * p.N + n);
        }
        acc[i][j] = v;
      }
    }
    if (p.ln) layernorm_rows<MT, NT, WN, BM>(acc, red, wn, 0, lane, BN);   // (launcher: BN == N)

    const bool vblock = nb >= p.n_store;
    if (nb == 0) RSTAMP(3);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      lds_barrier();   // LayerNorm scratch / the previous half's staging tile have been consumed by every wave
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = (ntile0 + i) * 16 + 4 * g, nl = n - nb;
        f32x4 ga0, be0, ga1, be1;
        if (p.film_mode == 1) {
          ga0 = *reinterpret_cast<const f32x4*>(FC + nl); be0 = *reinterpret_cast<const f32x4*>(FC + BN + nl);
          ga1 = *reinterpret_cast<const f32x4*>(FC + 2 * BN + nl); be1 = *reinterpret_cast<const f32x4*>(FC + 3 * BN + nl);
        }
#pragma unroll
        for (int jj = 0; jj < MH; ++jj) {
          const int j = hf * MH + jj;
          f32x4 v = acc[i][j];
          if (p.film_mode == 1) {
            const bool lo = sb[j] / p.film_div == f_lo;
            v = v * (lo ? ga0 : ga1) + (lo ? be0 : be1);
          }
          if (p.silu_out) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
          }
          const int rl = jj * 16 + l15;   // row inside the half
          if (vblock) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<T*>(OT + (nl + r) * SVT + rl * ES) = from_f<T>(valid[j] ? v[r] : 0.f);
          } else {
            store4(reinterpret_cast<T*>(OT + rl * SOT) + nl, v);
          }
        }
      }
      lds_barrier();
      if (nb == 0 && hf == 0) RSTAMP(4);
      const long h0 = r0 + hf * HALF;
      const long left = M - h0;
      const int rows_valid = left <= 0 ? 0 : (left < HALF ? (int)left : HALF);
      if (!vblock) {
        tile_copy_out<T>(OT, SOT, reinterpret_cast<T*>(p.out) + (size_t)h0 * p.n_store + nb, p.n_store, rows_valid, BN, tid, NTHR);
      } else {
        // transposed-V columns: [sample][channel][key], written per sample segment in 2-key pieces (L, BM even); the
        // padding keys [L, lpad) of every row stay as allocated (zero)
        const int NV = p.N - p.n_store, chb = nb - p.n_store;
        constexpr int PPR = HALF / 2;
        for (int id = tid; id < BN * PPR; id += NTHR) {
          const int ch = id / PPR, pp = id - ch * PPR;
          const long r = h0 + 2 * pp;
          if (r < M) {
            const long b = r / p.L;
            const int key = (int)(r - b * p.L);
            *reinterpret_cast<uint32_t*>(reinterpret_cast<T*>(p.vt) + ((size_t)b * NV + chb + ch) * p.vt_lpad + key) =
                *reinterpret_cast<const uint32_t*>(OT + ch * SVT + pp * 4);
          }
        }
      }
    }
    if (nb == 0) RSTAMP(5);
  }
  RSTAMP(6);
}

template <typename T, int BM, int BN>
size_t lds_need(int K) {
  constexpr size_t HALF = BM / 2, SOT = BN * sizeof(T) + 16, SVT = HALF * sizeof(T) + 16;
  const size_t stage = std::max(std::max(HALF * SOT, (size_t)2 * 8 * BM * 4) + (size_t)4 * BN * sizeof(float), (size_t)BN * SVT);
  return (size_t)BM * (K * sizeof(T) + 16) + stage;
}

template <typename T, int BM, int BN>
hipError_t launch_t(const GemmParams& p, hipStream_t st) {
  const size_t lds = lds_need<T, BM, BN>(p.seg[0].C);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const long M = (long)p.B * p.L;
  hipLaunchKernelGGL((rowgemm_kernel<T, BM, BN>), dim3((unsigned)((M + BM - 1) / BM)), dim3(512), lds, st, p);
  return hipGetLastError();
}

template <typename T, int BM, int BN>
hipError_t attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_kernel<T, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int pick_bn(const GemmParams& p) {
  for (int bn : {384, 256})
    if (p.N % bn == 0 && p.n_store % bn == 0 && (!p.ln || p.N == bn)) return bn;
  return 0;
}
int pick_bm(const GemmParams& p) { return p.seg[0].C <= 384 ? 128 : 64; }

}  // namespace

hipError_t rowgemm_init() {
  hipError_t e;
  if ((e = attr<bf16_t, 128, 384>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 128, 256>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 64, 384>()) != hipSuccess) return e;
  return attr<bf16_t, 64, 256>();
}

// Long row lists only: one Linear segment, bf16, no k=3 taps / upsampled residual / pool / fp32 output, column blocks of
// 384 or 256 (the widths whose 8-wave gemm.hip variant groups channels the same way).
bool rowgemm_supported(int prec, const GemmParams& p) {
  if (getenv("DHW_ROWGEMM") && atoi(getenv("DHW_ROWGEMM")) == 0) return false;
  if (prec != PREC_BF16 || p.nseg != 1 || p.seg[0].taps != 1 || p.res2 || p.pool || p.out_f32 || p.film_mode == 2 || p.film_div < 1) return false;
  const int K = p.seg[0].C;
  if (K % 32 || K > 768 || !pick_bn(p)) return false;
  if (p.n_store < p.N && ((p.L & 1) || !p.vt || p.ln || p.film_mode)) return false;
  if (p.film_mode == 1 && (!p.ln || (long)p.film_div * p.L < pick_bm(p))) return false;   // a tile spans <= two FiLM rows
  // measured on the all-steps text plane (tools/bench_text.cpp): ts.kv 552 -> 408 us, ts.ffn1 189 -> 147 us, level on the
  // LayerNorm blocks (which keep gemm.hip); below ~8k rows the per-sample tiling fills the chip better
  if (p.ln) return false;
  return (long)p.B * p.L >= 8192;
}

hipError_t launch_rowgemm(int prec, const GemmParams& p, hipStream_t st) {
  if (!rowgemm_supported(prec, p)) return hipErrorInvalidValue;
  const int bn = pick_bn(p);
  if (pick_bm(p) == 128) return bn == 384 ? launch_t<bf16_t, 128, 384>(p, st) : launch_t<bf16_t, 128, 256>(p, st);
  return bn == 384 ? launch_t<bf16_t, 64, 384>(p, st) : launch_t<bf16_t, 64, 256>(p, st);
}
