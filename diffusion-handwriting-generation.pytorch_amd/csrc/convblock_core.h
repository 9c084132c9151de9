// convblock_core.h — one whole ConvBlock (reference cnn.py:64-87) for one row tile of one sample, as a device function:
//
//     out = FiLM3(fc(SiLU(FiLM2(conv2(SiLU(FiLM1(conv1(SiLU(x))))))))) + conv_skip(x)
//
// (the description of the stages is in convblock.hip, which launches it as one kernel per block; persist.hip runs it as
// one phase of the per-step kernel).
#pragma once
#include <algorithm>
#include "enc_a_core.h"
#include "heads_core.h"

namespace {

// diagnostic builds only (-DDHW_ABL=n, tools/build_tools.sh): bit3 = SiLU -> identity, bit4 = no output / pool copy-out,
// bit5 = x tile not loaded (zeros), bit6 = workgroup barriers removed (results are wrong; only the timing is read)
#define CB_SILU(x) ((DHW_ABL & 8) ? (x) : silu_t<T>(x))
#define CB_SILU_TILES(NT_, MT_, v) do { if constexpr (!(DHW_ABL & 8)) silu_tiles2<T, NT_, MT_>(v); } while (0)
#define CB_BARRIER() do { if constexpr (!(DHW_ABL & 64)) lds_barrier(); } while (0)

#define STAMP(slot) DHW_STAMP_IF(p.stamps && (int)blockIdx.x == p.stagger && (threadIdx.x & 63) == 0, (threadIdx.x >> 6) * 16 + slot, __builtin_amdgcn_s_memrealtime())

// Row stride of the h2 / output staging tile: the conflict-free operand padding, except for the 126-row tiles, where the
// decoder block with the fused input stage would overflow LDS by 2 KB (128 channels with the 16-byte padding: one
// 2-way conflict per 16 lanes).
template <typename T, int BM> __host__ __device__ constexpr int h2_stride(int CO) { return BM >= 128 ? CO * (int)sizeof(T) + 16 : tile_stride<T>(CO); }

template <int NT>
struct Epi {   // this lane's bias / FiLM gamma / beta for its NT channel tiles, requested before the main loop
  f32x4 bias[NT], gam[NT], bet[NT];
  // Two forms, chosen at the call site: a run-time `g ? load : 1` turns into a branch whose merge copies the loaded value
  // at once — hipcc then waits s_waitcnt vmcnt(0) right behind the weight prefetch the caller has just issued, i.e. it
  // drains the whole queue once per stage (found in the .s of every fused kernel, r2).
  DHW_DEV void load(const float* b, const float* g, const float* be, int n0) {   // bias + FiLM gamma / beta (all non-null)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      bias[i] = *reinterpret_cast<const f32x4*>(b + n0 + 16 * i);
      gam[i] = *reinterpret_cast<const f32x4*>(g + n0 + 16 * i);
      bet[i] = *reinterpret_cast<const f32x4*>(be + n0 + 16 * i);
    }
  }
  DHW_DEV void load_bias(const float* b, int n0) {   // bias only (gamma = 1, beta = 0)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      bias[i] = *reinterpret_cast<const f32x4*>(b + n0 + 16 * i);
      gam[i] = (f32x4){1, 1, 1, 1};
      bet[i] = (f32x4){0, 0, 0, 0};
    }
  }
};

// OCC = workgroups meant to be co-resident per CU (VGPR budget 512 / (OCC * NW / 4) per lane): with 2, one workgroup's
// VALU-heavy epilogue / staging overlaps the other's MFMA phases.
// UPC = 0, or the block's input width Cin when the input itself is produced here (decoder blocks):
// x = Upsample(low) + skip_conv(h) (model.py:169-175), one more 3-tap GEMM stage in front of the block.
// CH = 1: the workgroup continues with enc_a of the EncoderLayer that follows the block (nx.a) on its output tile.
// CIN = the block's input width when it is compiled in (0 = run-time p.Cin): the weight rings of the stages that contract
// over Cin then request exactly their fragments (gemm_core.h, fill_s / run_s) instead of clamped look-ahead re-loads.
template <int UPC> constexpr int up_skip_width() { return UPC == 384 ? 256 : UPC == 256 ? 192 : UPC == 192 ? 128 : 0; }   // model.py:169-175
// The block for the BM-2-row tile at m0 of sample b, as a device function over the workgroup's LDS: the per-launch kernel
// (convblock.hip) and the persistent per-step kernel (persist.hip) both run it.
// TIGHT = 1: the workgroup writes BM - 4 rows instead of BM - 2, so that the x rows it needs — its output rows plus the two
// halo rows on each side of the two stacked 3-tap convolutions — are exactly BM: whole 16-row MFMA tiles.  With BM - 2 output
// rows the fused decoder input stage (the block's largest GEMM: Upsample + skip_conv for BM + 2 rows) computes one more row tile
// of which 2 rows are used: 4 tiles for 50 rows at BM = 48, 9 for 130 at BM = 128.  Chosen at launch when the row count of the
// level needs the same number of tiles either way (L / 4 = 122: 3 tiles of 44 or of 46).  The h1 / h2 / output rows past the
// BM - 4 valid ones are computed from whatever finite values follow the staged tiles in LDS and are never stored.
// TIGHT = 2 ("asymmetric"): the workgroup writes ALL BM rows of its tile; only conv1 — which must produce BM + 2 rows of h1 for the
// 3-tap conv2 — runs over one more 16-row tile (BM + 16 rows, 2 of the extra 16 used), conv2 / fc / conv_skip over exactly BM.
// For the L / 4 level of the bench shape (122 rows per sample): 4 tiles of 32 rows = 256 workgroups with 2 row tiles in three of
// the four GEMM stages, instead of 3 tiles of 46 (44) rows = 192 workgroups with 3 row tiles in every stage and 64 idle CUs.
// PP = 1 ("ping-pong", round 5): the two row halves of the tile run ONE PHASE APART.  The 2 row-group x 4 channel-group layouts
// of the tall tiles put waves w and w + 4 — same channels, lower / upper row half — on one SIMD, and in lockstep both are in
// their MFMA main loop, then both in their VALU epilogue: the matrix pipe idles through every epilogue and the vector issue
// through half of every main loop (SQ: MFMA busy 25 % of the kernel, VALU / MFMA co-execution 2-11 %).  With PP the upper half
// (waves 4-7) leads by one phase: each stage is cut in two phases, main loop | epilogue, with a workgroup barrier after each, and
// the lower half executes one extra barrier in front of its first phase (the upper half one behind its last) — the SAME
// instruction stream, no role-specific code.  In every slot one wave of a SIMD issues MFMAs while its partner runs the previous
// phase's bias + FiLM + SiLU + LDS stores.  Dependencies: conv2 of the lower half reads the upper half's first two h1 rows
// (written two slots earlier: the upper half leads); nothing of the upper half reads the lower half's tiles; the bf16 output
// tile overlays SiLU(x) only, which is dead once both halves have run conv1 (the fp32 tile of dec1 also overlays x: there the
// upper half waits one slot in front of its output phase instead of behind it).
template <typename T, int BM, int CO, int NW, int OCC = 1, int UPC = 0, int CH = 0, int CIN = 0, int TIGHT = 0, int PP = 0, typename P, typename X>
DHW_DEV void convblock_body(const P& p, const X& nx, const int b, const int m0, char* smem) {
  constexpr int ES = sizeof(T), NTHR = NW * 64;
  constexpr bool SK = CIN != 0;          // static contraction lengths
  constexpr int KT1 = 3 * CIN / 32;      // conv1 / conv_skip k-chunks when SK
  static_assert(UPC == 0 || CIN == 0 || CIN == UPC, "a fused input stage produces the block's own input width");
  constexpr bool ASYM = TIGHT == 2;
  constexpr int BMO = ASYM ? BM : BM - 2 - 2 * TIGHT;   // output rows per workgroup
  constexpr int RX = BMO + 4;                           // staged x rows: sample rows [m0 - 2, m0 + BMO + 2)
  constexpr int BM1 = ASYM ? BM + 16 : BM;              // h1 rows conv1 computes (index i <-> sample row m0 - 1 + i; BMO + 2 are needed)
  constexpr int C1 = CO / 2;             // conv1 output channels
  // Wave layouts (row groups x channel groups): stage 1 (conv1, C1 channels) and stages 2,3 (CO channels).
  // Every wave streams its OWN weight fragments from L2, so waves that differ only in their row group fetch the same
  // bytes again: with 4 row groups the conv1 stage of dec3 pulled 4 x 295 KB through the CU's 64 B/clk L1 path and ran
  // 9 us where 2 us of MFMA work was issued.  With 8 waves the channels are therefore split as finely as the 16-channel
  // MFMA tile allows (one row group whenever there are >= 6 channel tiles; waves beyond the tile count idle), and each
  // wave covers all rows.  (fp32 parity mode, 4 waves, keeps the 2x2 / 1x4 layouts.)
  constexpr int T1 = C1 / 16, T2 = CO / 16;                                   // 16-channel tiles per stage
  constexpr int WN1 = NW == 8 ? (T1 >= 8 ? 8 : (T1 == 6 ? 6 : 4)) : 2;
  constexpr int WM1 = NW == 8 ? (T1 >= 6 ? 1 : 2) : 2;
  // Per k-chunk a workgroup issues 4 (BM/16)(N/16) MFMA-cycles, reads WN (BM/16) KB of activation fragments from LDS
  // (128 B/clk) and WM (N/16) KB of weight fragments through L1 (64 B/clk).  126-row tiles of the 128-channel blocks with
  // 1 x 8 waves are LDS-bound (512 vs 256 cycles); 2 row groups x 4 channel groups balance all three at 256.
  constexpr bool TALL = NW == 8 && BM >= 128 && T2 == 8;
  constexpr int WN2 = NW == 8 ? (TALL ? 4 : (T2 % 8 == 0 ? 8 : 6)) : 4;
  constexpr int WM2 = TALL ? 2 : 1;
  constexpr int MT1 = BM1 / WM1 / 16, NT1 = T1 / WN1;
  constexpr int MT2 = BM / WM2 / 16, NT2 = T2 / WN2;
  static_assert(NT1 * WN1 == T1 && NT2 * WN2 == T2 && MT1 * 16 * WM1 == BM1 && WM1 * WN1 <= NW && WM2 * WN2 <= NW, "unsupported tile / wave layout");
  static_assert(!ASYM || WM1 == 1, "the asymmetric tiling: blocks with one row group in conv1");
#ifndef DHW_HEADS_PRE
#define DHW_HEADS_PRE 0   // measured neutral (18.120 vs 18.086 ms on a noisy box, profiles/r05_spread_ab.log r5av): off
#endif
#ifndef DHW_CONV_STROKES_ONCE
#define DHW_CONV_STROKES_ONCE 1
#endif
#ifndef DHW_CONV_FCX
#define DHW_CONV_FCX 1
#endif
#ifndef DHW_CONV_SPREAD
#define DHW_CONV_SPREAD 0   // measured neutral (18.763 vs 18.764 ms, profiles/r05_spread_ab.log): off
#endif
  constexpr bool CSPREAD = DHW_CONV_SPREAD != 0 && sizeof(T) == 2;
  constexpr bool PPX = PP != 0;
  static_assert(!PPX || (TALL && WM1 == 2 && WN1 == 4 && CH == 0 && !ASYM && OCC == 1), "ping-pong: the 2 x 4 wave layouts of the tall tiles");
  constexpr int RING = (ES == 2 ? 24 : 12) * (CO == 256 ? 2 : 3) / 3 / (OCC * NW > 8 ? OCC : 1);   // fewer fragments in flight for the widest block / at 2 WGs per CU (VGPR budget)

  // (DHW_UNIFORM_WAVE: the wave index as a scalar — `wave < 6` of the 192-channel blocks, whose layouts leave two waves without channels, is then a
  // scalar branch instead of an exec mask)
  const int tid = body_tid(), lane = tid & 63, wave = DHW_UNIFORM_WAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Cin = SK ? CIN : p.Cin;
  // ping-pong: the upper row half (waves NW/2 ..) leads by one phase; a scalar, so the skew barriers sit in scalar branches
  const bool pp_lead = PPX && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
#define PP_BARRIER() do { if constexpr (PPX) CB_BARRIER(); } while (0)

  const int SX = tile_stride<T>(Cin), SH1 = tile_stride<T>(C1), SH2 = h2_stride<T, BM>(CO);
  char* XS = smem;                       // SiLU(x)   [RX][Cin]
  char* XR = XS + RX * SX;               // x         [RX][Cin]
  char* H1 = XR + RX * SX;               // h1        [BM1+2][C1]  (index i <-> sample row m0-1+i)
  char* H2 = H1 + (BM1 + 2) * SH1;       // h2        [BM][CO]    (index i <-> sample row m0+i)
  const float* gam = p.film + (size_t)b * p.film_bs;
  const float* bet = gam + p.film_tot;

  // wave coordinates of the two layouts
  // idle waves only join the barriers / copies.  When a layout uses all NW waves the flag must FOLD to true: a run-time
  // `if (act)` around a stage whose loads are consumed inside it leaves, on the (never taken) skip path, loads that were
  // never waited for, and hipcc's s_waitcnt merge at the join then drains the next stage's weight prefetch (r2, .s).
  // DHW_CONV_DUP (round 5; gemm_core.h): in the 192-channel blocks (6 channel groups on 8 waves) the two spare waves REPEAT waves 0, 1 — the same tiles,
  // the same values to the same LDS addresses — so that the flags fold here too: behind a run-time `if (act)` hipcc drained the weight ring in front of
  // every main loop of these blocks (s_waitcnt vmcnt(2) / (1) / (0) behind their barriers, vmcnt(11 .. 21) in the other blocks).
  constexpr bool DUP1 = DHW_CONV_DUP != 0 && ES == 2 && NW == 8 && WM1 == 1 && WN1 < NW, DUP2 = DHW_CONV_DUP != 0 && ES == 2 && NW == 8 && WM2 == 1 && WN2 < NW;
  const bool act1 = DUP1 ? true : (WM1 * WN1 == NW || wave < WM1 * WN1), act2 = DUP2 ? true : (WM2 * WN2 == NW || wave < WM2 * WN2);
  const int wm1 = DUP1 ? 0 : (act1 ? wave / WN1 : 0), wn1 = DUP1 ? wave % WN1 : (act1 ? wave % WN1 : 0), row01 = wm1 * (BM1 / WM1), nt01 = wn1 * NT1;
  const int wm2 = DUP2 ? 0 : (act2 ? wave / WN2 : 0), wn2 = DUP2 ? wave % WN2 : (act2 ? wave % WN2 : 0), row02 = wm2 * (BM / WM2), nt02 = wn2 * NT2;
  const int n1 = nt01 * 16 + 4 * g, n2 = nt02 * 16 + 4 * g;   // this lane's first channel in each layout
  const int KCin = Cin / 32;

  if constexpr (OCC == 2) {
    // co-resident workgroups that run the same program in lockstep reach their MFMA phases, VALU epilogues and barriers
    // together; starting the later-dispatched half of the grid a fraction of a stage late lets one's epilogue overlap
    // the other's matrix work (MI355X_MICROARCH.md, Two waves per SIMD, item 9)
    if (p.stagger && blockIdx.x >= gridDim.x / 2)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  STAMP(0);
  WRing<T, NT1, RING> ring1;
  Epi<NT1> ep1;
  auto fill1 = [&]() {
    const T* w = reinterpret_cast<const T*>(p.w_c1) + ((size_t)nt01 * KCin * 3 * 64 + lane) * 8;
    if constexpr (SK) ring1.template fill_s<KT1>(w);
    else ring1.fill(w, KCin * 3);
  };

  // ---- stage 0: x tile -> LDS (raw + SiLU), zero outside the sample ('same' padding)
  if constexpr (UPC != 0) {
    // decoder input: x[r] = low[r/2] + b + sum_tap Wsk[tap] h[r-1+tap], rows [m0-2, m0-2+RX) rounded up to 16-row tiles.
    // h rows [m0-3, m0-3+RH) are staged where h1/h2 will live later; the up-sampled low rows go straight into the XR
    // tile, and the GEMM epilogue adds the convolution in place (same lane reads and writes an element).
    constexpr int RXP = (RX + 15) / 16 * 16, RH = RXP + 2, MTU = RXP / 16;
    // Wave layout: channels over WNU waves; 12 channel tiles (dec1) would leave 4 of 8 waves idle through the whole stage,
    // so there the rows are split over WMU = 2 row groups of MTG tiles (2 x 4 waves; the second group's surplus tile reads
    // LDS rows past the staged ones and is discarded by the r < RX test below).
    constexpr int TU = UPC / 16, WNU = TU % 8 == 0 ? 8 : 4, WMU = NW / WNU, NTU = TU / WNU;
    constexpr int MTG = (MTU + WMU - 1) / WMU;
    static_assert(NTU * WNU == TU && NW == 8, "unsupported input width");
    const int wmu = wave / WNU, wnu = wave % WNU;
    const int ntu0 = wnu * NTU, nu = ntu0 * 16 + 4 * g, rowu0 = wmu * MTG * 16;
    constexpr int UCH = up_skip_width<UPC>();
    const int Ch = SK ? UCH : p.up_cin, KCh = Ch / 32, SHh = tile_stride<T>(Ch);
    char* HS = H1;
    WRing<T, NTU, (NTU * MTG >= 24 ? 12 : RING)> ringu;
    Epi<NTU> epu;
    {
      // both input tiles in ONE memory round trip: every load of the h rows and of the low rows is requested before the
      // first LDS store, at clamped (always valid) addresses; rows outside the sample are zeroed by the store's select.
      // (Two back-to-back staged copies with per-lane conditional loads cost two dependent round trips plus the
      // s_waitcnt vmcnt(0) that hipcc puts at the join of every conditional load: 2.2-2.4 us of a 9-12 us stage, r2.)
      constexpr int CPRH = UCH * ES / 16, CPX = UPC * ES / 16;
      constexpr int UH = (RH * CPRH + NTHR - 1) / NTHR, UL = (RX * CPX + NTHR - 1) / NTHR;
      static_assert(SK, "the fused input stage is compiled for static widths");
      const char* src = reinterpret_cast<const char*>(p.up_h);
      const char* low = reinterpret_cast<const char*>(p.up_low);
      CopyRegs<UH> ch;
      CopyRegs<UL> cl;
      ch.load(RH * CPRH, tid, NTHR, [&](int id) { const int r = id / CPRH, cc = id - r * CPRH, lrow = min(max(m0 - 3 + r, 0), p.L - 1);
                                                  return reinterpret_cast<const uint4*>(src + ((size_t)(b * p.L + lrow) * UCH) * ES + (size_t)cc * 16); });
      cl.load(RX * CPX, tid, NTHR, [&](int id) { const int r = id / CPX, cc = id - r * CPX, lrow = min(max(m0 - 2 + r, 0), p.L - 1);
                                                 return reinterpret_cast<const uint4*>(low + ((size_t)(b * (p.L / 2) + (lrow >> 1)) * UPC) * ES + (size_t)cc * 16); });
      ch.store(RH * CPRH, tid, NTHR, [&](int id) { const int r = id / CPRH, cc = id - r * CPRH; return reinterpret_cast<uint4*>(HS + r * SHh + cc * 16); },
               [&](int id) { const int lrow = m0 - 3 + id / CPRH; return lrow >= 0 && lrow < p.L; });
      cl.store(RX * CPX, tid, NTHR, [&](int id) { const int r = id / CPX, cc = id - r * CPX; return reinterpret_cast<uint4*>(XR + r * SX + cc * 16); },
               [&](int id) { const int lrow = m0 - 2 + id / CPX; return lrow >= 0 && lrow < p.L; });
    }
    // (requested behind the staging loads: see below)
    ringu.template fill_s<3 * UCH / 32>(reinterpret_cast<const T*>(p.up_w) + ((size_t)ntu0 * KCh * 3 * 64 + lane) * 8);
    epu.load_bias(p.up_b, nu);
    CB_BARRIER();
    STAMP(10);
    f32x4 acc[NTU][MTG];
    acc_zero(acc);
    ringu.template run_s<MTG, 3 * UCH / 32>(acc, HS + (rowu0 + l15) * SHh + g * 8 * ES, SHh, KCh);
    STAMP(11);
    // conv1's first weight fragments fly during this stage's epilogue.  DHW_CONV_SPREAD (round 5): requested in quarters BETWEEN its pieces — 758 vector
    // instructions per wave, the block's longest epilogue — instead of as one burst in front of it (a wave sits in instruction issue until the CU's
    // L1 path has accepted its whole request: enc_bc_core.h, DHW_ENC_SPREAD)
    constexpr int FC1 = decltype(ring1)::template fill_chunks<KT1>(), FQ1 = (FC1 + 3) / 4;
    const T* w1p = reinterpret_cast<const T*>(p.w_c1) + ((size_t)nt01 * KCin * 3 * 64 + lane) * 8;
    if (act1) {
      if constexpr (CSPREAD) { ring1.template fill_begin<KT1>(w1p); ring1.template fill_range<KT1, 0, FQ1>(); }
      else fill1();   // flies during the epilogue
      ep1.load(p.b_c1, gam + p.f1, bet + p.f1, n1);
    }
    {
      // x = round(conv + bias + low) (the block's own 'same' padding: zero outside the sample) -> XR; SiLU(x) -> XS
      auto keep = [&](int j) { const int lrow = m0 - 2 + rowu0 + j * 16 + l15; return lrow >= 0 && lrow < p.L; };
      auto valid = [&](int j) { return rowu0 + j * 16 + l15 < RX; };
#pragma unroll
      for (int i = 0; i < NTU; ++i)
#pragma unroll
        for (int j = 0; j < MTG; ++j) {
          const int r = min(rowu0 + j * 16 + l15, RX - 1);   // (clamped: rows past RX are computed but never stored)
          acc[i][j] = round_to<T>(acc[i][j] + epu.bias[i] + load4(reinterpret_cast<const T*>(XR + r * SX) + nu + 16 * i));
        }
      if constexpr (CSPREAD) { if (act1) ring1.template fill_range<KT1, FQ1, 2 * FQ1>(); }
      // (a 16-byte store covers the partner lane's `low` values too: its data depends, through the lane swap, on both lanes'
      // loads of the pair, so no store can be issued ahead of them)
      store_tiles<T, NTU, MTG>(lane, XR, SX, rowu0, nu, acc, keep, valid);
      if constexpr (CSPREAD) { if (act1) ring1.template fill_range<KT1, 2 * FQ1, 3 * FQ1>(); }
      CB_SILU_TILES(NTU, MTG, acc);
      if constexpr (CSPREAD) { if (act1) ring1.template fill_range<KT1, 3 * FQ1, FC1>(); }
      store_tiles<T, NTU, MTG>(lane, XS, SX, rowu0, nu, acc, keep, valid);
    }
    STAMP(12);
  } else if (p.strokes && SK && DHW_CONV_STROKES_ONCE && NTHR % (CIN / 4 > 0 ? CIN / 4 : 1) == 0) {
    // enc1: x = input_dense(strokes) = W[:,0]*dx + W[:,1]*dy + b, evaluated in place of a load.  A thread's 4 channels are the same in every pass over
    // the rows (the thread count is a multiple of the items per row), so its 12 weights / biases are requested ONCE and the stroke points of all its
    // rows up front: the loop below then holds no load at all.  (As one loop of load - compute - store per item — the form below — the stage was
    // 8-9 dependent memory round trips: 2.0 of enc1's 14.3 us, tools/bench_conv stage0.)
    constexpr int CPR = CIN / 4 > 0 ? CIN / 4 : 1, IT = (RX * CPR + NTHR - 1) / NTHR, RSTEP = NTHR / CPR;
    const int c = (tid % CPR) * 4, r0 = tid / CPR;
    const f32x4 wa = *reinterpret_cast<const f32x4*>(p.in_w + c * 2), wb = *reinterpret_cast<const f32x4*>(p.in_w + c * 2 + 4);   // (w[c][0], w[c][1], w[c+1][0], ..)
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.in_b + c);
    float2 sp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int lrow = min(max(m0 - 2 + r0 + it * RSTEP, 0), p.L - 1);
      sp[it] = *reinterpret_cast<const float2*>(p.strokes + (size_t)(b * p.L + lrow) * 2);
    }
    const float w0[4] = {wa[0], wa[2], wb[0], wb[2]}, w1[4] = {wa[1], wa[3], wb[1], wb[3]};
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = r0 + it * RSTEP, lrow = m0 - 2 + r;
      if (r < RX) {
        const bool in = lrow >= 0 && lrow < p.L;
        const float s0 = sp[it].x, s1 = sp[it].y;
        f32x4 v, sv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = to_f(from_f<T>(w0[k] * s0 + w1[k] * s1 + bb[k]));
          v[k] = in ? t : 0.f;
          sv[k] = in ? CB_SILU(t) : 0.f;
        }
        store4(reinterpret_cast<T*>(XR + r * SX) + c, v);
        store4(reinterpret_cast<T*>(XS + r * SX) + c, sv);
      }
    }
  } else if (p.strokes) {
    // (run-time widths: one item at a time)
    const int cpr = Cin / 4;   // 4 channels per item
    for (int id = tid; id < RX * cpr; id += NTHR) {
      const int r = id / cpr, c = (id - r * cpr) * 4;
      const int lrow = m0 - 2 + r;
      f32x4 v = (f32x4){0, 0, 0, 0}, sv = v;
      if (lrow >= 0 && lrow < p.L) {
        const float s0 = p.strokes[(size_t)(b * p.L + lrow) * 2], s1 = p.strokes[(size_t)(b * p.L + lrow) * 2 + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = to_f(from_f<T>(p.in_w[(c + k) * 2] * s0 + p.in_w[(c + k) * 2 + 1] * s1 + p.in_b[c + k]));
          sv[k] = CB_SILU(v[k]);
        }
      }
      store4(reinterpret_cast<T*>(XR + r * SX) + c, v);
      store4(reinterpret_cast<T*>(XS + r * SX) + c, sv);
    }
  } else {
    // x tile: every load is requested (clamped, always valid address) before the first use; rows outside the sample are
    // zeroed by a select.  (Per-lane conditional loads compile to branches with s_waitcnt vmcnt(0) at their joins.)
    const int cpr = Cin * ES / 16;
    const int total = RX * cpr;
    const char* src = reinterpret_cast<const char*>(p.x);
    constexpr int U = 4;
    for (int base = tid; base < total; base += NTHR * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = min(base + u * NTHR, total - 1);
        const int r = id / cpr, cc = id - r * cpr;
        const int lrow = min(max(m0 - 2 + r, 0), p.L - 1);
        if constexpr (DHW_ABL & 32) v[u] = make_uint4(0, 0, 0, 0);
        else v[u] = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.L + lrow) * Cin) * ES + (size_t)cc * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * NTHR;
        if (id < total) {
          const int r = id / cpr, cc = id - r * cpr;
          const int lrow = m0 - 2 + r;
          const bool k = lrow >= 0 && lrow < p.L;
          uint4 w = make_uint4(k ? v[u].x : 0u, k ? v[u].y : 0u, k ? v[u].z : 0u, k ? v[u].w : 0u);
          *reinterpret_cast<uint4*>(XR + r * SX + cc * 16) = w;
          *reinterpret_cast<uint4*>(XS + r * SX + cc * 16) = (DHW_ABL & 8) ? w : silu_piece<T>(w);
        }
      }
    }
  }
  // the first stage's weights are requested BEHIND the staging loads (a wave's loads complete in order and the L1 miss
  // queue is shared: a 24 KB-per-wave prefetch in front of them delays the tile everything waits for)
  if (UPC == 0 && act1) {
    fill1();
    ep1.load(p.b_c1, gam + p.f1, bet + p.f1, n1);
  }
  STAMP(15);
  CB_BARRIER();   // x tiles complete (and the staged skip rows, which overlay h1, consumed)
  if constexpr (PPX) {
    if (!pp_lead) {
      // the trailing half idles through the leading half's first phase; it uses the slot to zero the two h1 rows past the computed
      // ones (in lockstep every thread does that behind conv1's epilogue — here that would race with the leading half's conv2)
      for (int id = tid; id < 2 * SH1 / 16; id += NTHR / 2)
        *reinterpret_cast<uint4*>(H1 + BM1 * SH1 + id * 16) = make_uint4(0, 0, 0, 0);
      CB_BARRIER();
    }
  }
  STAMP(1);

  WRing<T, NT2, RING> ring2;
  Epi<NT2> ep2;

  // ---- stage 1: h1 = SiLU(FiLM1(conv1(SiLU(x)))) for sample rows [m0-1, m0-1+BM)
  {
    f32x4 acc[NT1][MT1];
    acc_zero(acc);
    if (act1) {
      if constexpr (SK) ring1.template run_s<MT1, KT1>(acc, XS + (row01 + l15) * SX + g * 8 * ES, SX, KCin);
      else ring1.template run<MT1>(acc, XS + (row01 + l15) * SX + g * 8 * ES, SX, KCin);
    }
    STAMP(2);
    PP_BARRIER();   // (ping-pong: main loop | epilogue are two slots)
    // epilogue of conv1, tile pair by tile pair, with the conv2 weight prefetch requested between the pairs
    constexpr int KT2 = (C1 / 32) * 3;
    const T* w2 = reinterpret_cast<const T*>(p.w_c2) + ((size_t)nt02 * KT2 * 64 + lane) * 8;
    if (act2) { ring2.template fill_begin<KT2>(w2); ep2.load(p.b_c2, gam + p.f2, bet + p.f2, n2); }
    constexpr int ST1 = epilogue_steps<NT1, MT1>(), CH2 = decltype(ring2)::template fill_chunks<KT2>(), PER1 = (CH2 + ST1 - 1) / ST1;
    if (act1) {
      epilogue_pairs<T, NT1, MT1, !(DHW_ABL & 8)>(lane, H1, SH1, row01, n1, acc,
          [&](int i, const f32x4& a) { return (a + ep1.bias[i]) * ep1.gam[i] + ep1.bet[i]; },
          [&](int j) { const int srow = m0 - 1 + row01 + j * 16 + l15; return srow >= 0 && srow < p.L; },   // conv2 pads h1 with zeros
          [](int) { return true; },
          [&](int step) {
            if (act2) {
#pragma unroll
              for (int c = 0; c < PER1; ++c) ring2.template fill_chunk<KT2>(step * PER1 + c);
            }
          });
    } else if (act2) {
#pragma unroll
      for (int c = 0; c < CH2; ++c) ring2.template fill_chunk<KT2>(c);
    }
    // the two h1 rows past the computed BM (read only by the discarded output rows) must be finite
    if constexpr (!PPX)
    for (int id = tid; id < 2 * SH1 / 16; id += NTHR)
      *reinterpret_cast<uint4*>(H1 + BM1 * SH1 + id * 16) = make_uint4(0, 0, 0, 0);
  }
  STAMP(13);
  CB_BARRIER();
  STAMP(3);

  // ---- stage 2: h2 = SiLU(FiLM2(conv2(h1))) for sample rows [m0, m0+BM) (the last 2 are discarded)
  if (act2) {
    f32x4 acc[NT2][MT2];
    acc_zero(acc);
    ring2.template run_s<MT2, (C1 / 32) * 3>(acc, H1 + (row02 + l15) * SH1 + g * 8 * ES, SH1, C1 / 32);
    STAMP(4);
    PP_BARRIER();
    {
      constexpr int KTF = CO / 32;
      ring2.template fill_begin<KTF>(reinterpret_cast<const T*>(p.w_fc) + ((size_t)nt02 * KTF * 64 + lane) * 8);
      constexpr int ST2 = epilogue_steps<NT2, MT2>(), CHF = decltype(ring2)::template fill_chunks<KTF>(), PER2 = (CHF + ST2 - 1) / ST2;
      epilogue_pairs<T, NT2, MT2, !(DHW_ABL & 8)>(lane, H2, SH2, row02, n2, acc,
          [&](int i, const f32x4& a) { return (a + ep2.bias[i]) * ep2.gam[i] + ep2.bet[i]; },
          [](int) { return true; }, [](int) { return true; },
          [&](int step) {
#pragma unroll
            for (int c = 0; c < PER2; ++c) ring2.template fill_chunk<KTF>(step * PER2 + c);
          });
    }
    ep2.load(p.b_fc, gam + p.f3, bet + p.f3, n2);
  }
  STAMP(14);
  CB_BARRIER();
  STAMP(5);

  // ---- stage 3: out = FiLM3(fc(h2)) + conv_skip(x)
  f32x4 acc[NT2][MT2];
  acc_zero(acc);
  if (act2) {
    const T* wsk = reinterpret_cast<const T*>(p.w_skip) + ((size_t)nt02 * KCin * 3 * 64 + lane) * 8;
    // (DHW_CONV_FCX: conv_skip's first fragments requested under fc's MFMAs — run_n — instead of as one burst behind them)
    constexpr bool FCX = SK && (DHW_CONV_FCX != 0) && CO / 32 <= decltype(ring2)::D && DHW_ABL == 0;
    if constexpr (FCX) ring2.template run_n<MT2, CO / 32, KT1>(acc, H2 + (row02 + l15) * SH2 + g * 8 * ES, SH2, CO / 32, wsk);
    else ring2.template run_s<MT2, CO / 32>(acc, H2 + (row02 + l15) * SH2 + g * 8 * ES, SH2, CO / 32);
    STAMP(6);
    if constexpr (FCX) {}
    else if constexpr (SK) ring2.template fill_s<KT1>(wsk);
    else ring2.fill(wsk, KCin * 3);
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
      for (int j = 0; j < MT2; ++j) acc[i][j] = (acc[i][j] + ep2.bias[i]) * ep2.gam[i] + ep2.bet[i];
    ep2.load_bias(p.b_skip, n2);
    if constexpr (SK) ring2.template run_s<MT2, KT1>(acc, XR + (row02 + l15 + 1) * SX + g * 8 * ES, SX, KCin);   // out row i <- x rows i+1+tap
    else ring2.template run<MT2>(acc, XR + (row02 + l15 + 1) * SX + g * 8 * ES, SX, KCin);
  }
  STAMP(7);
  CB_BARRIER();   // every wave is done with the operand tiles: reuse LDS for the output tile
  // (ping-pong: "every wave" = this half; the bf16 tile only overlays SiLU(x), dead since both halves' conv1.  The fp32 tile also
  // overlays x, which the trailing half's conv_skip is reading in this slot: the leading half spends its extra slot HERE.)
  if constexpr (PPX) { if (pp_lead && p.out_f32) CB_BARRIER(); }
  STAMP(8);

  const int rows_valid = min(BMO, p.L - m0);
  if (p.out_f32) {
    constexpr int SO = CO * 4 + 16;
    if (act2) {
#pragma unroll
      for (int i = 0; i < NT2; ++i)
#pragma unroll
        for (int j = 0; j < MT2; ++j)
          store4(reinterpret_cast<float*>(smem + (row02 + j * 16 + l15) * SO) + n2 + 16 * i, acc[i][j] + ep2.bias[i]);
    }
    CB_BARRIER();
    // (ping-pong, fp32 tile: both halves stored in the same slot — the leading half waited in front of it)
    if (p.out)
      tile_copy_out<float>(smem, SO, reinterpret_cast<float*>(p.out) + (size_t)(b * p.L + m0) * CO, CO, rows_valid, CO, tid, NTHR);
    if (p.fuse_heads) {
      // eps / pen heads (model.py:179-182) + scheduler step straight from the fp32 tile: 4 lanes per stroke row
      const int r = tid >> 2, q = tid & 3;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      // (DHW_HEADS_PRE: what the scheduler step reads — biases, seed, sampler state, noise — requested, and the Philox draw computed, in front of the
      // dot products; rows past the tile read row m0's)
      const long hrow = (long)b * p.L + m0 + (r < rows_valid ? r : 0);
      HeadsPre hpre;
      if constexpr (DHW_HEADS_PRE != 0) heads_prefetch(p.hp, hrow, hpre);
      if (r < rows_valid) {
        const float* xr = reinterpret_cast<const float*>(smem + r * SO);
        // (unrolled: the 3 x CO / 16 weight pieces are requested together — rolled, every pass waited for its own three loads: CO / 16 dependent
        // round trips in the tail of the step's last kernel)
#pragma unroll
        for (int c = q * 4; c < CO; c += 16) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.hp.w_out + c);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.hp.w_out + CO + c);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(p.hp.w_pen + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) { a0 += v[k] * w0[k]; a1 += v[k] * w1[k]; a2 += v[k] * w2[k]; }
        }
      }
#pragma unroll
      for (int o = 2; o; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
      }
      if (r < rows_valid && q == 0) {
        if constexpr (DHW_HEADS_PRE != 0) heads_finish_pre(p.hp, hrow, a0, a1, a2, hpre);
        else heads_finish(p.hp, (long)b * p.L + m0 + r, a0, a1, a2);
      }
    }
  } else {
    if (act2) {
#pragma unroll
      for (int i = 0; i < NT2; ++i)
#pragma unroll
        for (int j = 0; j < MT2; ++j) acc[i][j] += ep2.bias[i];
      store_tiles<T, NT2, MT2>(lane, smem, SH2, row02, n2, acc);
    }
    CB_BARRIER();
    if constexpr (PPX) { if (pp_lead) CB_BARRIER(); }   // the trailing half stores its rows one slot later
    if constexpr (DHW_COPY_UNROLL != 0 && !(DHW_ABL & 16)) tile_copy_out_u<T, BM, CO, NTHR>(smem, SH2, reinterpret_cast<T*>(p.out) + (size_t)(b * p.L + m0) * CO, CO, rows_valid, tid);
    else if constexpr (!(DHW_ABL & 16))
    tile_copy_out<T>(smem, SH2, reinterpret_cast<T*>(p.out) + (size_t)(b * p.L + m0) * CO, CO, rows_valid, CO, tid, NTHR);
    if (p.pool && !(DHW_ABL & 16))   // AvgPool1d(2) side output (model.py:93); m0 and rows_valid are even
      tile_copy_out_pool<T>(smem, SH2, reinterpret_cast<T*>(p.pool) + ((size_t)b * (p.L / 2) + m0 / 2) * CO, CO, rows_valid, CO, tid, NTHR);
    if constexpr (CH == 1) {
      // the output tile (rows [m0, m0 + rows_valid), row stride SH2 = tile_stride(CO)) is the next layer's x tile
      static_assert(NW == 8 && sizeof(T) == 2, "enc_a_body runs on 8 waves");
      EncALds m;
      m.XR = smem;
      m.QR = smem + BM * SH2;
      m.red = reinterpret_cast<float*>(m.QR + BM * SH2);
      m.KT = reinterpret_cast<char*>(m.red) + 2 * 8 * BM * sizeof(float);
      m.VT = m.KT + 32 * tile_stride<T>(CO);
      m.VS = smem;
      m.PL = reinterpret_cast<float*>(m.KT + enc_a_text_kv_bytes<T, CO, BM>());
      enc_a_body<T, CO, BM, (ASYM ? 16 : 4)>(nx.a, m, b, m0, rows_valid);   // (tile starts are multiples of BM - 2: even, not 8-aligned; of BM for the asymmetric tiling)
    }
  }
  STAMP(9);
}

// (bm1: the h1 rows conv1 computes = BM, or BM + 16 for the asymmetric tiling, whose x tiles are BM + 4 rows)
template <typename T, int BM, int CO>
inline size_t lds_bytes(int Cin, int up_cin = 0, int bm1 = BM) {
  const int rx = bm1 > BM ? BM + 4 : BM + 2;
  // (conv1's discarded rows read up to 2 x rows past its last computed row: they must lie inside the allocation)
  const size_t xt = (size_t)2 * rx * tile_stride<T>(Cin);
  const size_t ops = xt + (size_t)(bm1 + 2) * tile_stride<T>(CO / 2) + (size_t)BM * h2_stride<T, BM>(CO);
  const size_t outf = (size_t)BM * (CO * 4 + 16);
  // x tiles + staged h rows (with 12 channel tiles the input GEMM runs as 2 row groups of ceil(MTU / 2) tiles: the second
  // group's surplus tile reads rows past the staged ones, which must still lie inside the allocation)
  const int mtu = (rx + 15) / 16, wmu = (Cin / 16) % 8 == 0 ? 1 : 2, rows_read = wmu * ((mtu + wmu - 1) / wmu) * 16 + 2;
  const size_t up = up_cin ? xt + (size_t)rows_read * tile_stride<T>(up_cin) : 0;
  return std::max(ops, std::max(outf, up));
}

}  // namespace
