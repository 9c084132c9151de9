// enclayer.hip — the stroke side of one EncoderLayer (reference model.py:37-58) in TWO launches instead of
// eight.  A workgroup (8 waves) owns 64 / 32 / 16 stroke rows of one sample (pick_bm: enough tiles for every CU);
// every intermediate of its rows lives in LDS.  The only hand-off through global memory is the one the algorithm
// forces: self-attention needs the K/V of ALL rows of the sample, so the layer is cut between the q/k/v projection
// and the attention — and since everything between two self-attentions is row-local, enc_bc<NEXT> goes on with the
// next layer's enc_a (enc5: via AvgPool1d + att_dense) on the tile it holds, so a chain of layers costs one launch each.
//
//   enc_a : q1 = Wq(x+PE) -> cross-attention over the text keys -> dense -> LN -> FiLM1 -> +x = x2
//           -> [q2|k2|v2] = W(x2 (+PE))                       (writes x2, qk2, vt2)
//   enc_bc: self-attention(q2,k2,v2) -> dense -> +x2 -> LN -> FiLM2 = x3 -> ffn1 (SiLU) -> ffn2 -> +x3
//           -> LN -> FiLM3 = out (+ AvgPool1d(2) side output)
//
// GEMM stages: every wave covers all rows and 1/WN of the channels (no two waves stream the same weights), weights
// streamed from L2 in fragment order (gemm_core.h).  Attention stages: K/V blocks staged in LDS, wave = 16 rows x
// one or two heads (attn_core.h).  Workgroup ids are XCD-aware (xcd_swizzle.h): a sample's tiles share one L2.
#include <algorithm>
#include <cstdlib>
#include "enc_bc_core.h"

namespace {

template <typename T, int DM, int BM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void enc_a_kernel(const EncLayerParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles = (p.Lk + BM - 1) / BM;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);   // the row tiles of one sample run on one XCD (shared K/V in its L2)
  enc_a_tile<T, DM, BM>(p, bid / tiles, (bid % tiles) * BM, smem);
}

template <typename T, int DM, int BM, int NEXT = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void enc_bc_kernel(const EncLayerParams p, const EncChain nx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles = (p.Lk + BM - 1) / BM;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);   // the row tiles of one sample run on one XCD (shared K/V in its L2)
  enc_bc_body<T, DM, BM, NEXT>(p, nx, bid / tiles, (bid % tiles) * BM, smem);
}

template <typename T, int DM, int BM, int NEXT = 0>
hipError_t launch_bc(const EncLayerParams& p, const EncChain& nx, hipStream_t st) {
  const int tiles = (p.Lk + BM - 1) / BM;
  constexpr size_t lds = lds_bc_chain_bytes<T, DM, BM, NEXT>();
  static_assert(lds <= 160 * 1024, "enc_bc tile does not fit LDS");
  hipLaunchKernelGGL((enc_bc_kernel<T, DM, BM, NEXT>), dim3(p.B * tiles), dim3(512), lds, st, p, nx);
  return hipGetLastError();
}

// variants with a compiled chain: mode 1 for the attention layers (DM = 384, tiles up to 32 rows: LDS), mode 2 for enc5
template <typename T, int DM, int BM>
constexpr bool has_chain(int mode) {   // (LDS: the chained layer's text K/V block sits behind the three stage tiles; bf16 only)
  return sizeof(T) == 2 && ((mode == 1 && DM == 384 && BM <= 32) || (mode == 2 && DM == 256 && BM == 32));
}

// does the (element type, width, row tile) combination fit the 160 KiB of LDS in both halves of the layer?
template <typename T, int DM, int BM>
constexpr bool fits() { return lds_a_bytes<T, DM, BM>() <= 160 * 1024 && lds_bc_bytes<T, DM, BM>() <= 160 * 1024 && (DM % 128 == 0 || BM >= 32); }

template <typename T, int DM, int BM>
hipError_t launch_pair(const EncLayerParams& p, int which, hipStream_t st, const EncChain* chain) {
  if constexpr (!fits<T, DM, BM>()) {
    return hipErrorInvalidValue;
  } else {
  const int tiles = (p.Lk + BM - 1) / BM;
  if (which == 0) {
    constexpr size_t lds = lds_a_bytes<T, DM, BM>();
    hipLaunchKernelGGL((enc_a_kernel<T, DM, BM>), dim3(p.B * tiles), dim3(512), lds, st, p);
    return hipGetLastError();
  }
  {
    if (chain && chain->mode) {
      if (chain->a.x) return hipErrorInvalidValue;   // the chained layer reads its x tile from LDS
      if constexpr (has_chain<T, DM, BM>(1)) { if (chain->mode == 1 && chain->a.d == DM) return launch_bc<T, DM, BM, 1>(p, *chain, st); }
      if constexpr (has_chain<T, DM, BM>(2)) { if (chain->mode == 2 && chain->a.d == 384 && !(p.Lk & 1)) return launch_bc<T, DM, BM, 2>(p, *chain, st); }
      return hipErrorInvalidValue;
    }
    return launch_bc<T, DM, BM, 0>(p, EncChain{}, st);
  }
  }
}

template <typename T, int DM, int BM, int NEXT>
hipError_t attr_bc() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(enc_bc_kernel<T, DM, BM, NEXT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <typename T, int DM, int BM>
hipError_t attr() {
  if constexpr (fits<T, DM, BM>()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_a_kernel<T, DM, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if ((e = attr_bc<T, DM, BM, 0>()) != hipSuccess) return e;
    if constexpr (has_chain<T, DM, BM>(1)) { if ((e = attr_bc<T, DM, BM, 1>()) != hipSuccess) return e; }
    if constexpr (has_chain<T, DM, BM>(2)) { if ((e = attr_bc<T, DM, BM, 2>()) != hipSuccess) return e; }
  }
  return hipSuccess;
}

// row tile: 64 rows when that still gives every CU a workgroup, else 32, else 16 (twice / four times the workgroups)
template <typename T, int DM>
int pick_bm(int B, int Lk, int bm_min = 0) {
  const char* e = getenv(DM == 384 ? "DHW_ENC_BM384" : DM == 256 ? "DHW_ENC_BM256" : "DHW_ENC_BM192");   // experiments only
  if (!e) e = getenv("DHW_ENC_BM");
  const int force = e ? atoi(e) : 0;
  const char* t = getenv("DHW_ENC_WGS");
  const long target = t ? atol(t) : 256;   // smallest tile count that still gives every CU a workgroup
  int bm = 64;
  if ((long)B * ((Lk + 63) / 64) < target) bm = 32;
  if (DM % 128 == 0 && (long)B * ((Lk + 31) / 32) < target) bm = 16;   // (the 4x2 wave layout of DM=192 needs >= 32 rows)
  if (force) bm = force;
  // LDS budget (160 KiB) of both halves: shrink the row tile until the combination fits (fp32 tiles are twice as wide)
  auto ok = [](int m) { return m == 64 ? fits<T, DM, 64>() : m == 32 ? fits<T, DM, 32>() : fits<T, DM, 16>(); };
  while (bm > 16 && !ok(bm)) bm /= 2;
  if (DM % 128 != 0 && bm < 32) bm = 32;
  if (bm < bm_min) bm = bm_min;
  return bm == 16 || bm == 32 ? bm : 64;
}

template <typename T, int DM>
hipError_t launch_bm(const EncLayerParams& p, int which, hipStream_t st, const EncChain* chain) {
  const int bm = pick_bm<T, DM>(p.B, p.Lk, p.bm_min);
  if (bm == 16) return launch_pair<T, DM, (DM % 128 == 0 ? 16 : 32)>(p, which, st, chain);
  return bm == 32 ? launch_pair<T, DM, 32>(p, which, st, chain) : launch_pair<T, DM, 64>(p, which, st, chain);
}

}  // namespace

hipError_t enclayer_init() {
  hipError_t e;
  if ((e = attr<bf16_t, 192, 64>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 256, 64>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 384, 64>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 192, 32>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 256, 32>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 384, 32>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 256, 16>()) != hipSuccess) return e;
  if ((e = attr<bf16_t, 384, 16>()) != hipSuccess) return e;
  // fp32 parity mode: the same kernels (exact-f32 MFMA), the row tiles that fit LDS
  if ((e = attr<float, 192, 64>()) != hipSuccess) return e;
  if ((e = attr<float, 192, 32>()) != hipSuccess) return e;
  if ((e = attr<float, 256, 32>()) != hipSuccess) return e;
  if ((e = attr<float, 256, 16>()) != hipSuccess) return e;
  return attr<float, 384, 16>();
}

bool enclayer_supported(int prec, int d, int heads) {
  const bool f32_fused = !(getenv("DHW_FUSE_F32") && atoi(getenv("DHW_FUSE_F32")) == 0);   // A/B: one launch per GEMM in fp32
  return (prec == PREC_BF16 || (prec == PREC_F32 && f32_fused)) && (d == 192 || d == 256 || d == 384) && heads * 64 == d;
}

// which: 0 = enc_a (cross attention half + q/k/v projection), 1 = enc_bc (self attention + FFN half)
bool enclayer_chain_supported(int prec, int d, int B, int Lk, int mode, int d_next) {
  if (prec != PREC_BF16) return false;
  if (mode == 1) return d == 384 && d_next == 384 && pick_bm<bf16_t, 384>(B, Lk) <= 32;
  if (mode == 2) return d == 256 && d_next == 384 && (Lk & 1) == 0 && pick_bm<bf16_t, 256>(B, Lk, 32) == 32;   // (with bm_min = 32)
  return false;
}

hipError_t launch_enclayer(int prec, const EncLayerParams& p, int which, hipStream_t st, const EncChain* chain) {
  if (!enclayer_supported(prec, p.d, p.heads) || (p.pool && (p.Lk & 1)) || (chain && chain->mode && which != 1)) return hipErrorInvalidValue;
  if (prec == PREC_F32) {
    if (chain && chain->mode) return hipErrorInvalidValue;
    switch (p.d) {
      case 192: return launch_bm<float, 192>(p, which, st, nullptr);
      case 256: return launch_bm<float, 256>(p, which, st, nullptr);
      case 384: return launch_bm<float, 384>(p, which, st, nullptr);
    }
    return hipErrorInvalidValue;
  }
  switch (p.d) {
    case 192: return launch_bm<bf16_t, 192>(p, which, st, chain);
    case 256: return launch_bm<bf16_t, 256>(p, which, st, chain);
    case 384: return launch_bm<bf16_t, 384>(p, which, st, chain);
  }
  return hipErrorInvalidValue;
}
