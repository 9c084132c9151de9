// gemm.hip — the fused "activation-stationary" MFMA GEMM that carries every
// Linear / Conv1d(k=3) of the denoiser (reference cnn.py:64-87, model.py:37-58,
// text_style.py:91-104, attention.py:78-85), with the surrounding elementwise
// work fused into its prologue/epilogue:
//
//   prologue : SiLU on the activations while they are staged into LDS;
//              k=3 taps read three row-shifted views of ONE staged tile (+halo)
//   mainloop : weights are pre-packed in MFMA-fragment order, so each wave
//              streams its own 1-KiB-contiguous weight fragments straight from
//              L2 into VGPRs (no LDS, no barrier in the loop); activations are
//              staged once per workgroup and read as ds_read_b128 fragments
//   epilogue : bias, PE·W position bias, residual, LayerNorm (cross-wave),
//              sigma-FiLM, residual / nearest-upsampled residual, SiLU,
//              AvgPool1d(2) side output, transposed V side output.
//
// Orientation: the MFMA "A" operand is the WEIGHT fragment (rows = output
// channels) and the "B" operand the ACTIVATION fragment (cols = stroke rows),
// so each lane ends up with 4 consecutive output channels of one stroke row:
// C-last stores are 8/16-byte vectors and LayerNorm needs 2 shuffles.
//
// A workgroup = 4 waves computes BM rows (of one sample) x BN channels; waves
// are arranged WM x WN, each owning (BM/WM) x (BN/WN).
#include <algorithm>
#include <cstdlib>
#include "gemm_core.h"
#include "dhw_kernels.h"

namespace {

#define GSTAMP(slot) DHW_STAMP_IF(p.stamps && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0, slot, __builtin_amdgcn_s_memrealtime())

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const GemmParams p) {
  constexpr int NTHR = WM * WN * 64;
  constexpr int MT = BM / WM / 16;   // activation (column) tiles per wave
  constexpr int NT = BN / WN / 16;   // channel (row) tiles per wave
  constexpr int ES = sizeof(T);
  constexpr int D0 = (ES == 2 ? 12 : 6) / NT;       // prefetch depth: ~12 (bf16) / 6 (fp32) fragments in flight per lane
  constexpr int D = D0 < 2 ? 2 : (D0 > 8 ? 8 : D0);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  const int tiles = (p.L + BM - 1) / BM;
  const int b = blockIdx.x / tiles;
  const int m0 = (blockIdx.x % tiles) * BM;
  const int nb0 = blockIdx.y * BN;

  // ---- LDS carve: one activation tile per segment, then LN scratch
  int lds_off[2], lds_stride[2];
  int off = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    lds_off[s] = off;
    lds_stride[s] = 0;
    if (s < p.nseg) {
      const int rows = BM + (p.seg[s].taps == 3 ? 2 : 0);
      lds_stride[s] = tile_stride<T>(p.seg[s].C);
      off += rows * lds_stride[s];
    }
  }
  float* red = reinterpret_cast<float*>(smem + off);   // [2][WN][BM]

  GSTAMP(0);
  // ---- stage activations (zero outside the sample: 'same' padding and row tail)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s >= p.nseg) break;
    const GemmSeg& sg = p.seg[s];
    const int halo = sg.taps == 3 ? 1 : 0;
    const int rows = BM + 2 * halo;
    const int cpr = sg.C * ES / 16;              // 16-byte chunks per row
    const int total = rows * cpr;
    const char* src = reinterpret_cast<const char*>(sg.A);
    constexpr int U = 4;   // independent 16-byte loads in flight per thread
    for (int base = tid; base < total; base += NTHR * U) {
      uint4 v[U];
      int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * NTHR;
        const int r = id / cpr, cc = id - r * cpr;
        const int lrow = m0 - halo + r;
        v[u] = make_uint4(0, 0, 0, 0);
        dst[u] = id < total ? lds_off[s] + r * lds_stride[s] + cc * 16 : -1;
        if (id < total && lrow >= 0 && lrow < p.L)
          v[u] = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.L + lrow) * sg.C) * ES + (size_t)cc * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (sg.silu) {
          T* e = reinterpret_cast<T*>(&v[u]);
#pragma unroll
          for (int i = 0; i < 16 / ES; ++i) e[i] = from_f<T>(silu_t<T>(to_f(e[i])));
        }
        if (dst[u] >= 0) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
      }
    }
  }
  __syncthreads();
  GSTAMP(1);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

  const int ntile0 = (nb0 + wn * (BN / WN)) / 16;   // first global channel tile of this wave
  const int row0 = wm * (BM / WM);                  // first tile-local row of this wave

#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s >= p.nseg) break;
    const GemmSeg& sg = p.seg[s];
    const int KC = sg.C / 32;
    const int KT = KC * sg.taps;                    // k-chunks of this segment
    const T* wbase = reinterpret_cast<const T*>(sg.W) + ((size_t)ntile0 * KT * 64 + lane) * 8;
    const char* abase = smem + lds_off[s] + (row0 + l15) * lds_stride[s] + g * 8 * ES;

    // Weight fragments stream L2 -> VGPRs through a D-deep register ring.  The loop body is branch-free and
    // statically indexed so hipcc emits counted s_waitcnt vmcnt((D-1)*NT) instead of draining the queue.
    Frag<T> wq[D][NT];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int k = d < KT ? d : KT - 1;
#pragma unroll
      for (int i = 0; i < NT; ++i) wq[d][i] = frag_load(wbase + ((size_t)i * KT + k) * 512);
    }
    int aoff = 0, kc = 0;                       // LDS byte offset of the current k-chunk: tap*stride + kc*32*ES
    const int tap_step = lds_stride[s] - (KC - 1) * 32 * ES;
    auto step = [&](int d, int knext) {
      Frag<T> a[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) a[j] = frag_load(reinterpret_cast<const T*>(abase + j * 16 * lds_stride[s] + aoff));
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) mma32(acc[i][j], wq[d][i], a[j]);
      const int kn = knext < KT ? knext : KT - 1;   // clamped: the last D re-loads are harmless
#pragma unroll
      for (int i = 0; i < NT; ++i) wq[d][i] = frag_load(wbase + ((size_t)i * KT + kn) * 512);
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

    if (s == 0 && p.nseg == 2) {
      // between the segments: acc = FiLM(acc + bias0) (ConvBlock: affine3(fc(.)), cnn.py:81-82)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = (ntile0 + i) * 16 + 4 * g;
        const f32x4 bi = *reinterpret_cast<const f32x4*>(p.bias0 + n);
        f32x4 ga = (f32x4){1, 1, 1, 1}, be = (f32x4){0, 0, 0, 0};
        if (p.film_mode == 2) {
          ga = *reinterpret_cast<const f32x4*>(p.gam + (long)(b / p.film_div) * p.film_bs + n);
          be = *reinterpret_cast<const f32x4*>(p.bet + (long)(b / p.film_div) * p.film_bs + n);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (acc[i][j] + bi) * ga + be;
      }
    }
  }

  GSTAMP(2);
  // ---------------------------------------------------------------- epilogue
  const float* bias_last = p.nseg == 2 ? p.bias1 : p.bias0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = (ntile0 + i) * 16 + 4 * g;
    const f32x4 bi = bias_last ? *reinterpret_cast<const f32x4*>(bias_last + n) : (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int lrow = m0 + row0 + j * 16 + l15;
      f32x4 v = acc[i][j] + bi;
      if (lrow < p.L) {
        if (p.posb && n < p.posb_cols) v += *reinterpret_cast<const f32x4*>(p.posb + (size_t)lrow * p.posb_cols + n);
        if (p.res1) v += load4(reinterpret_cast<const T*>(p.res1) + (size_t)(b * p.L + lrow) * p.N + n);
      }
      acc[i][j] = v;
    }
  }

  if (p.ln) {
    // LayerNorm over the N = BN channels of each row (model.py:25: eps 1e-6, no affine); two-pass in fp32.
    float mean[MT], rstd[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) red[wn * BM + row0 + j * 16 + l15] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) s += red[w * BM + row0 + j * 16 + l15];
      mean[j] = s * p.ln_inv;
    }
    const bool padded = p.ln_n < BN;   // (uniform) statistics over the first ln_n channels; the rest is zero padding
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d = acc[i][j][r] - mean[j];
          if (padded && wn * (BN / WN) + i * 16 + 4 * g + r >= p.ln_n) d = 0.f;
          s += d * d;
        }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) red[(WN + wn) * BM + row0 + j * 16 + l15] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) s += red[(WN + w) * BM + row0 + j * 16 + l15];
      rstd[j] = rsqrtf(s * p.ln_inv + 1e-6f);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (acc[i][j] - mean[j]) * rstd[j];
    if (padded) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (wn * (BN / WN) + i * 16 + 4 * g + r >= p.ln_n) {
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j][r] = 0.f;
          }
    }
  }

  // ---- final epilogue -> output tile in LDS -> whole coalesced rows to global (per-lane 8-byte stores straight from
  // the accumulators touch 16 rows per instruction and are store-issue bound).  A workgroup's BN columns are either all
  // regular output columns or all transposed-V columns (the launcher picks BN | n_store).
  GSTAMP(3);
  const bool vblock = nb0 >= p.n_store;
  constexpr int SOT = BN * ES + 16, SOF = BN * 4 + 16, SVT = BM * ES + 16;
  __syncthreads();   // every wave is done with the operand tiles: reuse LDS
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nl = (wn * (BN / WN)) + i * 16 + 4 * g;   // column inside the block
    const int n = nb0 + nl;
    f32x4 ga = (f32x4){1, 1, 1, 1}, be = (f32x4){0, 0, 0, 0};
    if (p.film_mode == 1) {
      ga = *reinterpret_cast<const f32x4*>(p.gam + (long)(b / p.film_div) * p.film_bs + n);
      be = *reinterpret_cast<const f32x4*>(p.bet + (long)(b / p.film_div) * p.film_bs + n);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int rl = row0 + j * 16 + l15;
      const int lrow = m0 + rl;
      const bool valid = lrow < p.L;
      f32x4 v = acc[i][j];
      if (p.film_mode == 1) v = v * ga + be;
      if (p.res2 && valid) {
        const size_t rr = p.res2_half ? (size_t)b * (p.L / 2) + (lrow >> 1) : (size_t)b * p.L + lrow;
        v += load4(reinterpret_cast<const T*>(p.res2) + rr * p.N + n);
      }
      if (p.silu_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
      }
      if (p.relu6_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], 0.f), 6.f);
      }
      if (vblock) {   // [channel][row], rows past L zero
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<T*>(smem + (nl + r) * SVT + rl * ES) = from_f<T>(valid ? v[r] : 0.f);
      } else if (p.out_f32) {
        store4(reinterpret_cast<float*>(smem + rl * SOF) + nl, v);
      } else {
        store4(reinterpret_cast<T*>(smem + rl * SOT) + nl, v);
      }
    }
  }
  __syncthreads();
  GSTAMP(4);
  const int rows_valid = min(BM, p.L - m0);
  if (vblock) {
    constexpr int EPV = 16 / ES, PPR = BM / EPV;
    const int NV = p.N - p.n_store;
    T* vt = reinterpret_cast<T*>(p.vt) + ((size_t)b * NV + (nb0 - p.n_store)) * p.vt_lpad + m0;
    for (int id = tid; id < BN * PPR; id += NTHR) {
      const int ch = id / PPR, part = id - ch * PPR;
      if (m0 + (part + 1) * EPV <= p.vt_lpad)
        *reinterpret_cast<uint4*>(vt + (size_t)ch * p.vt_lpad + part * EPV) = *reinterpret_cast<const uint4*>(smem + ch * SVT + part * 16);
    }
  } else if (p.out_f32) {
    tile_copy_out<float>(smem, SOF, reinterpret_cast<float*>(p.out) + (size_t)(b * p.L + m0) * p.n_store + nb0, p.n_store, rows_valid, BN, tid, NTHR);
  } else {
    tile_copy_out<T>(smem, SOT, reinterpret_cast<T*>(p.out) + (size_t)(b * p.L + m0) * p.n_store + nb0, p.n_store, rows_valid, BN, tid, NTHR);
    if (p.pool)
      tile_copy_out_pool<T>(smem, SOT, reinterpret_cast<T*>(p.pool) + ((size_t)b * (p.L / 2) + m0 / 2) * p.N + nb0, p.N, rows_valid, BN, tid, NTHR);
  }
  GSTAMP(5);
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t launch_one(const GemmParams& p, hipStream_t st) {
  const int tiles = (p.L + BM - 1) / BM;
  dim3 grid(p.B * tiles, p.N / BN);
  size_t lds = 0;
  for (int s = 0; s < p.nseg; ++s)
    lds += (size_t)(BM + (p.seg[s].taps == 3 ? 2 : 0)) * tile_stride<T>(p.seg[s].C);
  lds += 2 * WN * BM * sizeof(float);
  const size_t out_tile = std::max((size_t)BM * (BN * (p.out_f32 ? 4 : sizeof(T)) + 16), p.n_store < p.N ? (size_t)BN * (BM * sizeof(T) + 16) : 0);
  lds = std::max(lds, out_tile);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, st, p);
  return hipGetLastError();
}

// The widest blocks run as 8 narrower waves (NT = 3 / 2 channel tiles each instead of 6 / 4): a workgroup's life is a
// latency chain (stage A -> weight stream -> epilogue), and the deep grids of the all-steps text plane want waves in
// flight more than registers per wave (ts.dense 138 -> 121 us).  The choice must not depend on the batch: a prompt's
// samples are bit-identical whatever batch it is sampled in (LayerNorm partial sums are grouped per wave).
static bool wide_w8() {
  static const bool on = !(getenv("DHW_GEMM_W8") && atoi(getenv("DHW_GEMM_W8")) == 0);
  return on;
}

template <typename T, int BM>
hipError_t launch_bn(int BN, const GemmParams& p, hipStream_t st) {
  if (sizeof(T) == 2 && wide_w8()) {
    if (BN == 384) return launch_one<T, BM, 384, 1, 8>(p, st);
    if (BN == 256) return launch_one<T, BM, 256, 1, 8>(p, st);
  }
  switch (BN) {
    case 64: return launch_one<T, BM, 64, 1, 4>(p, st);
    case 96: return launch_one<T, BM, 96, 2, 2>(p, st);
    case 128: return launch_one<T, BM, 128, 1, 4>(p, st);
    case 192: return launch_one<T, BM, 192, 1, 4>(p, st);
    case 256: return launch_one<T, BM, 256, 1, 4>(p, st);
    case 384: return launch_one<T, BM, 384, 1, 4>(p, st);
  }
  return hipErrorInvalidValue;
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, BM, BN, WM, WN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename T, int BM>
hipError_t set_attr_bm() {
  hipError_t e;
  if ((e = set_attr<T, BM, 64, 1, 4>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 96, 2, 2>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 128, 1, 4>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 192, 1, 4>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 256, 1, 4>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 256, 1, 8>()) != hipSuccess) return e;
  if ((e = set_attr<T, BM, 384, 1, 8>()) != hipSuccess) return e;
  return set_attr<T, BM, 384, 1, 4>();
}

}  // namespace

hipError_t gemm_init() {
  hipError_t e;
  if ((e = set_attr_bm<bf16_t, 64>()) != hipSuccess) return e;
  if ((e = set_attr_bm<bf16_t, 32>()) != hipSuccess) return e;
  return set_attr_bm<float, 32>();
}

void gemm_tile_for(int prec, const GemmParams& p, int* BM, int* BN) {
  int bm = 32;
  // 64-row tiles unless 32-row tiles waste markedly fewer padded rows (tiles never span samples), e.g. L = 70
  if (prec == PREC_BF16 && p.L >= 48 && (((p.L + 63) / 64) * 64) * 4 <= (((p.L + 31) / 32) * 32) * 5) bm = 64;
  const int cands[6] = {384, 256, 192, 128, 96, 64};
  int bn = 0;
  if (p.ln) {
    bn = p.N;
  } else {
    const long tiles = (long)p.B * ((p.L + bm - 1) / bm);
    // largest BN dividing N that still yields >= 512 workgroups; else the smallest divisor >= 128 (or the only one)
    // a workgroup's columns must be all regular or all transposed-V: BN divides n_store (and N - n_store)
    auto ok = [&](int c) { return p.N % c == 0 && p.n_store % c == 0; };
    for (int c : cands)
      if (ok(c) && tiles * (p.N / c) >= 512) { bn = c; break; }
    if (!bn) {
      for (int k = 5; k >= 0; --k)
        if (ok(cands[k]) && (cands[k] >= 128 || p.N == cands[k])) { bn = cands[k]; break; }
    }
    if (!bn)
      for (int k = 5; k >= 0; --k)
        if (ok(cands[k])) { bn = cands[k]; break; }
    if (const char* e = getenv("DHW_GEMM_BN")) {   // experiments only
      const int f = atoi(e);
      if (f > 0 && ok(f)) bn = f;
    }
  }
  *BM = bm;
  *BN = bn;
}

hipError_t launch_gemm(int prec, const GemmParams& p_in, hipStream_t st) {
  GemmParams p = p_in;
  if (p.ln) {
    if (p.ln_n <= 0 || p.ln_n > p.N) p.ln_n = p.N;
    p.ln_inv = 1.0f / (float)p.ln_n;
  }
  if (p.film_div < 1 || p.nseg < 1 || p.nseg > 2 || p.N % 16 || p.n_store % 16 || (p.pool && (p.L & 1))) return hipErrorInvalidValue;
  for (int s = 0; s < p.nseg; ++s)
    if (p.seg[s].C % 32 || (p.seg[s].taps != 1 && p.seg[s].taps != 3)) return hipErrorInvalidValue;
  int bm, bn;
  gemm_tile_for(prec, p, &bm, &bn);
  if (!bn) return hipErrorInvalidValue;
  if (prec == PREC_BF16) {
    return bm == 64 ? launch_bn<bf16_t, 64>(bn, p, st) : launch_bn<bf16_t, 32>(bn, p, st);
  }
  return launch_bn<float, 32>(bn, p, st);
}
