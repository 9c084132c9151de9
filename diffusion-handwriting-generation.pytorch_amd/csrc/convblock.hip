// convblock.hip — one whole ConvBlock (reference cnn.py:64-87) as a single kernel:
//
//     out = FiLM3(fc(SiLU(FiLM2(conv2(SiLU(FiLM1(conv1(SiLU(x))))))))) + conv_skip(x)
//
// A workgroup owns BM-2 consecutive stroke rows of one sample.  x (raw and SiLU'd, +2 halo rows each side) is
// staged into LDS once — for enc1 it is computed on the fly from the (dx,dy) strokes (input Linear(2->C),
// model.py:139), so the sampler state never round-trips through an activation buffer; conv1 -> h1 and
// conv2 -> h2 never leave LDS; the k=3 taps of every conv read row-shifted views of the staged tiles; fc and
// conv_skip accumulate into the same MFMA accumulators with the FiLM3 affine applied in between.  Weights
// stream from L2 in MFMA-fragment order through a register ring that is re-filled for the NEXT stage before
// the current stage's epilogue (gemm_core.h).  The output tile goes through LDS so global stores are whole
// coalesced rows (+ the AvgPool1d side output).  Replaces three launches and the HBM/L2 round trips of h1/h2.
//
// bf16: 8 waves (2 per SIMD), 62-row tiles (126 at full resolution, 46 at the L/4 level: launch_convblock).  fp32 (parity
// mode): 4 waves, 30-row tiles (LDS budget).  Decoder blocks (UPC) evaluate Upsample(low) + skip_conv(h) (model.py:169-175)
// as one more 3-tap GEMM stage in front; dec1 evaluates the eps / pen heads and the scheduler step from its fp32 tile;
// CH = 1 continues into the next EncoderLayer's first half on the output tile (enc_a_core.h; off by default).
#include <algorithm>
#include <cstdlib>
#include "convblock_core.h"

namespace {

template <typename T, int BM, int CO, int NW, int OCC = 1, int UPC = 0, int CH = 0, int CIN = 0, int TIGHT = 0, int PP = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(OCC * NW / 4, (OCC * NW / 4) < 2 ? 2 : OCC * NW / 4)))
void convblock_kernel(const ConvBlockParams p, const EncChain nx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BMO = TIGHT == 2 ? BM : BM - 2 - 2 * TIGHT;
  const int tiles = (p.L + BMO - 1) / BMO;
  // XCD-aware workgroup id, the same sample -> XCD assignment in every fused kernel of the model: a sample's activations
  // are then handed from kernel to kernel inside one XCD's L2 (measured: a 98 KB tile written by the previous kernel on
  // the same XCD is read in 1.6 us, from another XCD in 3.8 us — tools/bench_handoff.cpp)
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  convblock_body<T, BM, CO, NW, OCC, UPC, CH, CIN, TIGHT, PP>(p, nx, bid / tiles, (bid % tiles) * BMO, smem);
}

template <typename T, int BM, int CO, int NW, int OCC = 1, int UPC = 0, int CH = 0, int CIN = 0, int TIGHT = 0, int PP = 0>
hipError_t launch_t(const ConvBlockParams& p, hipStream_t st, const EncChain* nx = nullptr) {
  size_t lds = lds_bytes<T, BM, CO>(p.Cin, UPC ? p.up_cin : 0, TIGHT == 2 ? BM + 16 : BM);
  if (CH) lds = std::max(lds, (size_t)2 * BM * tile_stride<T>(CO) + 2 * 8 * BM * sizeof(float) + enc_a_text_kv_bytes<T, CO, BM>() + enc_a_param_bytes<T, CO>());
  if (lds > 160 * 1024 || (UPC && (p.Cin != UPC || p.up_cin % 32)) || (CH != 0) != (nx != nullptr)) return hipErrorInvalidValue;
  if (CIN && (p.Cin != CIN || (UPC && p.up_cin != up_skip_width<UPC>()))) return hipErrorInvalidValue;
  constexpr int BMO = TIGHT == 2 ? BM : BM - 2 - 2 * TIGHT;
  const int tiles = (p.L + BMO - 1) / BMO;
  hipLaunchKernelGGL((convblock_kernel<T, BM, CO, NW, OCC, UPC, CH, CIN, TIGHT, PP>), dim3(p.B * tiles), dim3(NW * 64), lds, st, p, nx ? *nx : EncChain{});
  return hipGetLastError();
}

template <typename T, int BM, int CO, int NW, int OCC = 1, int UPC = 0, int CH = 0, int CIN = 0, int TIGHT = 0, int PP = 0>
hipError_t attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(convblock_kernel<T, BM, CO, NW, OCC, UPC, CH, CIN, TIGHT, PP>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// The reference's encoder blocks (model.py:85-89): Cout 128 <- 128, 192 <- 128, 256 <- 192; these input widths are compiled in.
constexpr int enc_cin(int CO) { return CO == 256 ? 192 : 128; }

}  // namespace

hipError_t convblock_init() {
  hipError_t e;
#define A(...) if ((e = attr<__VA_ARGS__>()) != hipSuccess) return e
  // run-time input width (decoder blocks without the fused input stage, experiments)
  A(bf16_t, 64, 128, 8); A(bf16_t, 128, 128, 8); A(bf16_t, 64, 128, 8, 2, 0, 0, 128); A(bf16_t, 64, 192, 8, 2, 0, 0, 128); A(bf16_t, 64, 192, 8);
  A(bf16_t, 64, 256, 8); A(bf16_t, 32, 256, 8); A(bf16_t, 48, 256, 8);
  // encoder blocks, input width compiled in
  A(bf16_t, 64, 128, 8, 1, 0, 0, 128); A(bf16_t, 128, 128, 8, 1, 0, 0, 128); A(bf16_t, 64, 192, 8, 1, 0, 0, 128);
  A(bf16_t, 64, 256, 8, 1, 0, 0, 192); A(bf16_t, 48, 256, 8, 1, 0, 0, 192);
  // decoder blocks with the fused Upsample + skip_conv input stage
  A(bf16_t, 64, 128, 8, 1, 192, 0, 192); A(bf16_t, 128, 128, 8, 1, 192, 0, 192); A(bf16_t, 64, 192, 8, 1, 256, 0, 256);
  A(bf16_t, 64, 256, 8, 1, 384, 0, 384); A(bf16_t, 48, 256, 8, 1, 384, 0, 384);
  A(bf16_t, 128, 128, 8, 1, 192, 0, 192, 1); A(bf16_t, 48, 256, 8, 1, 384, 0, 384, 1);   // ... writing BM - 4 rows per tile (convblock_core.h, TIGHT)
  A(bf16_t, 128, 128, 8, 1, 0, 0, 128, 0, 1); A(bf16_t, 128, 128, 8, 1, 192, 0, 192, 1, 1); A(bf16_t, 128, 128, 8, 1, 192, 0, 192, 0, 1);   // the tall tiles with the row halves one phase apart (PP)
  A(bf16_t, 32, 256, 8, 1, 0, 0, 192, 2); A(bf16_t, 32, 256, 8, 1, 384, 0, 384, 2);       // 32-row tiles of the widest blocks, conv1 over 48 rows (TIGHT = 2)
  // encoder blocks that continue into the next EncoderLayer's first half
  A(bf16_t, 64, 192, 8, 1, 0, 1, 128); A(bf16_t, 48, 256, 8, 1, 0, 1, 192); A(bf16_t, 64, 256, 8, 1, 0, 1, 192);
  A(bf16_t, 32, 256, 8, 1, 0, 1, 192, 2);   // ... on the asymmetric 32-row tiles (the layer's own 32-row tiling: enc4 -> enc5.a)
  A(float, 32, 128, 4); A(float, 32, 192, 4); A(float, 32, 256, 4);
#undef A
  return hipSuccess;
}

// workgroups that fill the device in one round (experiments: DHW_CONV_WGS, e.g. 128 for a half batch that shares the CUs with the other half's kernels)
static long conv_wgs() {
  const char* e = getenv("DHW_CONV_WGS");
  return e ? atol(e) : 256;
}

// 46-row tiles for the widest blocks (L/4 level) when the 62-row tiling leaves CUs idle and the finer one still fits one round
static bool use_bm48(const ConvBlockParams& p) {
  if (const char* e = getenv("DHW_CONV_BM")) return atoi(e) == 48;
  const long t64 = (long)p.B * ((p.L + 61) / 62), t48 = (long)p.B * ((p.L + 45) / 46);
  return t64 < conv_wgs() && t48 <= conv_wgs() && t48 > t64;
}

// 32-row tiles with conv1 over 48 rows (TIGHT = 2) for the widest blocks when they fill the CUs in one round where the 46-row
// tiling leaves some idle (B = 64 at L / 4 = 122: 256 instead of 192 workgroups); DHW_CONV_ASYM=0 switches it off (A/B)
static bool use_asym32(const ConvBlockParams& p) {
  static const bool on = !(getenv("DHW_CONV_ASYM") && atoi(getenv("DHW_CONV_ASYM")) == 0);
  if (!on || getenv("DHW_CONV_BM")) return false;
  const long t32 = (long)p.B * ((p.L + 31) / 32), t46 = (long)p.B * ((p.L + 45) / 46);
  return t32 <= conv_wgs() && t32 > t46;
}

// DHW_CONV_PP: bit 0 = enc1-type tall tiles (126 rows x 128 channels), bit 1 = dec1-type (fused input stage), run with their row
// halves one phase apart (convblock_core.h, PP).  Default: see the measurement in DESIGN 14.
static int conv_pp() {   // (read per launch — capture time under graph replay — so that a test can flip it between two handles)
  const char* e = getenv("DHW_CONV_PP");
  return e ? atoi(e) : 0;
}

hipError_t launch_convblock(int prec, const ConvBlockParams& p_in, hipStream_t st) {
  ConvBlockParams p = p_in;
  if (const char* e = getenv("DHW_CONV_STAGGER")) p.stagger = atoi(e);
  if (p.Cin % 32 || (p.L & 1) || (p.pool && p.out_f32) || (p.fuse_heads && !p.out_f32) || (!p.out && !p.fuse_heads)) return hipErrorInvalidValue;
  static const bool rt = getenv("DHW_CONV_RT") && atoi(getenv("DHW_CONV_RT"));   // A/B: force the run-time-width variants
  if (p.up_h) {   // decoder block with the fused Upsample + skip_conv input stage (bf16 only)
    if (prec != PREC_BF16 || p.strokes || !p.up_w || !p.up_b || !p.up_low) return hipErrorInvalidValue;
    // (BM - 4 output rows per tile where that needs no more tiles than BM - 2: the input stage then has no partial row tile)
    static const bool tight_ok = !(getenv("DHW_CONV_TIGHT") && atoi(getenv("DHW_CONV_TIGHT")) == 0);
    auto tight = [&](int bm) { return tight_ok && (p.L + bm - 5) / (bm - 4) == (p.L + bm - 3) / (bm - 2); };
    if (p.Cout == 128 && p.Cin == 192 && p.up_cin == 128) {
      const bool big = (long)p.B * ((p.L + 61) / 62) > conv_wgs() && lds_bytes<bf16_t, 128, 128>(p.Cin, p.up_cin) <= 160 * 1024;
      if (big && (conv_pp() & 2)) return tight(128) ? launch_t<bf16_t, 128, 128, 8, 1, 192, 0, 192, 1, 1>(p, st) : launch_t<bf16_t, 128, 128, 8, 1, 192, 0, 192, 0, 1>(p, st);
      if (big && tight(128)) return launch_t<bf16_t, 128, 128, 8, 1, 192, 0, 192, 1>(p, st);
      return big ? launch_t<bf16_t, 128, 128, 8, 1, 192, 0, 192>(p, st) : launch_t<bf16_t, 64, 128, 8, 1, 192, 0, 192>(p, st);
    }
    if (p.Cout == 192 && p.Cin == 256 && p.up_cin == 192) return launch_t<bf16_t, 64, 192, 8, 1, 256, 0, 256>(p, st);
    if (p.Cout == 256 && p.Cin == 384 && p.up_cin == 256) {
      if (use_asym32(p)) return launch_t<bf16_t, 32, 256, 8, 1, 384, 0, 384, 2>(p, st);
      if (use_bm48(p) && tight(48)) return launch_t<bf16_t, 48, 256, 8, 1, 384, 0, 384, 1>(p, st);
      return use_bm48(p) ? launch_t<bf16_t, 48, 256, 8, 1, 384, 0, 384>(p, st) : launch_t<bf16_t, 64, 256, 8, 1, 384, 0, 384>(p, st);
    }
    return hipErrorInvalidValue;
  }
  if (prec == PREC_BF16) {
    const bool sk = !rt && p.Cin == enc_cin(p.Cout);   // an encoder block: static contraction lengths
    switch (p.Cout) {
      case 128: {   // full-resolution blocks: 126-row tiles keep the grid within one round of workgroups (one 8-wave WG per CU)
        const bool big = (long)p.B * ((p.L + 61) / 62) > conv_wgs() && lds_bytes<bf16_t, 128, 128>(p.Cin) <= 160 * 1024 &&
                         !(getenv("DHW_CONV_BM") && atoi(getenv("DHW_CONV_BM")) == 64);
        if (getenv("DHW_CONV_OCC") && atoi(getenv("DHW_CONV_OCC")) == 2 && sk && lds_bytes<bf16_t, 64, 128>(p.Cin) <= 80 * 1024)
          return launch_t<bf16_t, 64, 128, 8, 2, 0, 0, 128>(p, st);
        if (sk && big && (conv_pp() & 1)) return launch_t<bf16_t, 128, 128, 8, 1, 0, 0, 128, 0, 1>(p, st);
        if (sk) return big ? launch_t<bf16_t, 128, 128, 8, 1, 0, 0, 128>(p, st) : launch_t<bf16_t, 64, 128, 8, 1, 0, 0, 128>(p, st);
        return big ? launch_t<bf16_t, 128, 128, 8>(p, st) : launch_t<bf16_t, 64, 128, 8>(p, st);
      }
      case 192:
        if (getenv("DHW_CONV_OCC") && atoi(getenv("DHW_CONV_OCC")) == 2 && sk && lds_bytes<bf16_t, 64, 192>(p.Cin) <= 80 * 1024)
          return launch_t<bf16_t, 64, 192, 8, 2, 0, 0, 128>(p, st);
        return sk ? launch_t<bf16_t, 64, 192, 8, 1, 0, 0, 128>(p, st) : launch_t<bf16_t, 64, 192, 8>(p, st);
      case 256:   // (30-row tiles = 2.5x the workgroups at the L/4 level measured slower: 34.1 vs 30.8 us; env DHW_CONV_BM=32 to retry)
        if (sk && use_asym32(p)) return launch_t<bf16_t, 32, 256, 8, 1, 0, 0, 192, 2>(p, st);
        if (use_bm48(p)) return sk ? launch_t<bf16_t, 48, 256, 8, 1, 0, 0, 192>(p, st) : launch_t<bf16_t, 48, 256, 8>(p, st);
        if (getenv("DHW_CONV_BM") && atoi(getenv("DHW_CONV_BM")) == 32) return launch_t<bf16_t, 32, 256, 8>(p, st);
        return sk ? launch_t<bf16_t, 64, 256, 8, 1, 0, 0, 192>(p, st) : launch_t<bf16_t, 64, 256, 8>(p, st);
    }
  } else {
    switch (p.Cout) {
      case 128: return launch_t<float, 32, 128, 4>(p, st);
      case 192: return launch_t<float, 32, 192, 4>(p, st);
      case 256: return launch_t<float, 32, 256, 4>(p, st);
    }
  }
  return hipErrorInvalidValue;
}

// The encoder-side blocks in front of an EncoderLayer (enc2 -> enc3, enc4 -> enc5): bf16, no fused input stage, no fp32
// output, the block's width is the layer's width.
bool convblock_chain_supported(int prec, const ConvBlockParams& p, const EncChain& chain) {
  if (prec != PREC_BF16 || chain.mode != 1 || chain.a.x || p.up_h || p.strokes || p.out_f32 || p.fuse_heads) return false;
  if (chain.a.d != p.Cout || chain.a.Lk != p.L || chain.a.B != p.B || (p.L & 1)) return false;
  if (getenv("DHW_CONV_BM") || getenv("DHW_CONV_OCC")) return false;
  return (p.Cout == 192 || p.Cout == 256) && p.Cin == enc_cin(p.Cout) && enclayer_supported(prec, chain.a.d, chain.a.heads);
}

// Whether continuing this block into the next EncoderLayer's first half is a measured WIN for its launch geometry (the default of
// DHW_CHAIN_CONV for enc4 -> enc5.a): only on the asymmetric 32-row tiles, which are the layer's own tiling (r5: 18.02 -> 17.89 ms
// same-box, profiles/r05_enc4_chain_ab.log; on the 46-row tiles the chain measured slower in rounds 1 and 3).
bool convblock_chain_auto(const ConvBlockParams& p) { return p.Cout == 256 && use_asym32(p); }

hipError_t launch_convblock_chain(int prec, const ConvBlockParams& p, const EncChain& chain, hipStream_t st) {
  if (!convblock_chain_supported(prec, p, chain) || p.Cin % 32 || (!p.out)) return hipErrorInvalidValue;
  if (p.Cout == 192) return launch_t<bf16_t, 64, 192, 8, 1, 0, 1, 128>(p, st, &chain);
  if (use_asym32(p)) return launch_t<bf16_t, 32, 256, 8, 1, 0, 1, 192, 2>(p, st, &chain);
  return use_bm48(p) ? launch_t<bf16_t, 48, 256, 8, 1, 0, 1, 192>(p, st, &chain) : launch_t<bf16_t, 64, 256, 8, 1, 0, 1, 192>(p, st, &chain);
}
