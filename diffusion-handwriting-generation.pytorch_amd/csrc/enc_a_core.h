// enc_a_core.h — the first half of an EncoderLayer's stroke side (reference model.py:37-58) as a device function over an
// x tile that lives in LDS: q-projection, cross-attention over the text keys, dense + LayerNorm + FiLM + residual, and
// the q/k/v projection of the self-attention.  Every stage is row-local, so besides the stand-alone enc_a kernel
// (enclayer.hip) it is appended to whatever kernel produced the x tile: the previous layer's enc_bc, or the ConvBlock
// in front of the layer (convblock.hip) — no launch boundary, no round trip of x through memory.
#pragma once
#include "attn_core.h"
#include "gemm_core.h"
#include "dhw_kernels.h"

// diagnostic stage stamps (100 MHz s_memrealtime), only when the caller passes a buffer
// (p.dbg bit 2, tools/bench_encw a: the same slots per wave, in shader cycles, where enc_bc keeps its per-wave timeline)
#define ENC_STAMP(slot) do { DHW_STAMP_IF(p.stamps && blockIdx.x == 0 && threadIdx.x == 0, slot, __builtin_amdgcn_s_memrealtime()); \
                             DHW_STAMP_IF(p.stamps && (p.dbg & 4) && blockIdx.x == 0 && (threadIdx.x & 63) == 0, 64 + (threadIdx.x >> 6) * 32 + (slot), __builtin_amdgcn_s_memtime()); } while (0)

template <typename T, int BM>
DHW_DEV void stage_rows(char* dst, int S, const T* src, int C, int b, int L, int m0, int tid, int nthreads) {
  constexpr int ES = sizeof(T);
  const int cpr = C * ES / 16;
  const int total = BM * cpr;
  constexpr int U = 4;
  for (int base = tid; base < total; base += nthreads * U) {
    uint4 v[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = base + u * nthreads;
      const int r = id / cpr, cc = id - r * cpr;
      v[u] = make_uint4(0, 0, 0, 0);
      off[u] = id < total ? r * S + cc * 16 : -1;
      if (id < total && m0 + r < L)
        v[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + ((size_t)(b * L + m0 + r) * C) * ES + (size_t)cc * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) *reinterpret_cast<uint4*>(dst + off[u]) = v[u];
  }
}

// epilogue parameters of one stage (this lane's 4 channels of each of its NT channel tiles), requested BEFORE the
// stage's main loop so their L2 latency is hidden behind it
template <int NT>
struct EpiParams {
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
  // The same from a parameter block in LDS (ParamStage below): b / g / be point at the vectors' first channel.  The 16
  // lanes of a lane group read the same 16 bytes (broadcast), the 4 groups consecutive 16-byte slots: one conflict-free
  // ds_read_b128 per vector and channel tile instead of a 1-KiB-per-wave trip through the CU's L1 return path.
  DHW_DEV void lds(const float* b, const float* g, const float* be, int n0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      bias[i] = *reinterpret_cast<const f32x4*>(b + n0 + 16 * i);
      gam[i] = *reinterpret_cast<const f32x4*>(g + n0 + 16 * i);
      bet[i] = *reinterpret_cast<const f32x4*>(be + n0 + 16 * i);
    }
  }
  DHW_DEV void lds_bias(const float* b, int n0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) bias[i] = *reinterpret_cast<const f32x4*>(b + n0 + 16 * i);
  }
};

// Epilogue parameter vectors of a fused kernel, staged ONCE per workgroup into LDS.  Loaded per lane from global memory
// (16 identical copies per wave: EpiParams::load) every vector costs each wave a full 1-KiB wave-instruction on the CU's
// L1 return path (64 B/clk, shared by the 8 waves), the path the weight stream of the attention-level layers is bound
// by: 9 of a stage's 45 vector-memory instructions per wave at d = 384 (profiles/r03_inst_mix.json).  Here
// NS <= 8 vectors of W floats are fetched by two waves each (128 threads, one 16-byte piece per lane, W / 4 <= 128 pieces),
// four vectors per pass; a wave's vector index is uniform, so the source pointer is chosen with scalar selects.
// load() only requests (unconditional, clamped: no branch, no s_waitcnt vmcnt(0) at a join), store() writes the block
// pl[slot * W + c]; the caller's next workgroup barrier publishes it.  512-thread workgroups.
template <int NS>
struct ParamStage {
  static_assert(NS <= 8, "two passes of four vectors");
  uint4 v0, v1;   // (named members and pointer arguments, no arrays: hipcc left small arrays of either kind in scratch memory)
  template <int W>
  DHW_DEV void load(int tid, const float* s0, const float* s1, const float* s2, const float* s3, const float* s4 = nullptr,
                    const float* s5 = nullptr, const float* s6 = nullptr, const float* s7 = nullptr) {
    static_assert(W % 4 == 0 && W / 4 <= 128, "vector width");
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 7), c = tid & 127;
    const int cc = c < W / 4 ? c : W / 4 - 1;
    const float* a = grp == 0 ? s0 : grp == 1 ? s1 : grp == 2 ? s2 : s3;
    v0 = *reinterpret_cast<const uint4*>(a + 4 * cc);
    if constexpr (NS > 4) {
      // (vectors past NS: the last real one is fetched again and not stored)
      const float* l = NS == 5 ? s4 : NS == 6 ? s5 : NS == 7 ? s6 : s7;
      const float* b = grp == 0 ? s4 : grp == 1 ? (NS > 5 ? s5 : l) : grp == 2 ? (NS > 6 ? s6 : l) : (NS > 7 ? s7 : l);
      v1 = *reinterpret_cast<const uint4*>(b + 4 * cc);
    }
  }
  template <int W>
  DHW_DEV void store(float* pl, int tid) const {
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 7), c = tid & 127;
    // two unconditional-shaped stores of by-value copies (written as `if (..) *p0 = v0; if (..) *p1 = v1;` on the members hipcc
    // turned the pair into ONE store of a run-time-indexed member, which pinned the struct to scratch memory)
    const uint4 a = v0;
    float* pa = pl + grp * W + 4 * c;
    if (c < W / 4 && grp < NS) *reinterpret_cast<uint4*>(pa) = a;
    if constexpr (NS > 4) {
      const uint4 b = v1;
      asm volatile("" ::: "memory");
      if (c < W / 4 && grp + 4 < NS) *reinterpret_cast<uint4*>(pa + 4 * W) = b;
    }
  }
};
// enc_a's parameter block: [b_q1 | b_d1 | gamma1 | beta1 | b_qkv2 (3 vectors)], bf16 kernels only (the fp32 parity mode's
// tiles leave no room at d = 384 and keep the per-lane loads)
template <typename T> constexpr bool enc_plds() { return sizeof(T) == 2; }
template <typename T, int DM> constexpr size_t enc_a_param_bytes() { return enc_plds<T>() ? (size_t)7 * DM * sizeof(float) : 0; }

// LDS regions of the enc_a stages.  XR and QR are [BM][DM] tiles (row stride tile_stride(DM)); KT / VT hold one 32-key block of
// text keys / values (bf16: both [keys][DM]; fp32 parity mode: V^T [DM][keys]); VS (fp32 only) is the staging area of the
// transposed v2 tile (DM rows of BM keys), free to overlay XR / QR.
struct EncALds {
  char* XR;     // x, later x2
  char* QR;     // q1, later a1
  float* red;   // LayerNorm partial sums [2][8][BM]
  char* KT;
  char* VT;
  char* VS;
  float* PL;    // parameter block (enc_a_param_bytes), bf16 kernels
};
template <typename T, int DM, int BM>
constexpr size_t enc_a_text_kv_bytes() {
  return sizeof(T) == 2 ? (size_t)2 * 32 * tile_stride<T>(DM) : (size_t)32 * tile_stride<T>(DM) + (size_t)DM * (32 * sizeof(T) + OPAD<T>);
}
// row stride (elements) of the layer's [q2 | k2 (| v2)] buffer: the bf16 kernels keep v2 beside q2 / k2, row-major (attn_core.h)
template <typename T, int DM> constexpr int qkv_stride() { return sizeof(T) == 2 ? 3 * DM : 2 * DM; }

// enc_a for the BM-row tile [m0, m0+BM) of sample b, of which the first rows_valid rows are this workgroup's to write.
// p.x == null: the x tile is already in m.XR (written by the caller's previous stage, behind a barrier) — this is how a
// ConvBlock or the previous layer's enc_bc continues into the next layer without a launch boundary.  VPIECE = bytes per
// store of the transposed v2 tile (16, or 4 when m0 is only even).
// P: EncLayerParams — a kernel argument, or the same struct in the constant address space (persist.hip: parameters in a plan in memory).
template <typename T, int DM, int BM, int VPIECE = 16, typename P>
DHW_DEV void enc_a_body(const P& p, const EncALds& m, int b, int m0, int rows_valid) {
  constexpr int ES = sizeof(T);
  // GEMM stages: every wave covers all BM rows and 1/WN of the channels, so no two waves stream the same weight
  // fragments (row groups would re-fetch them: the L2 -> CU weight stream is what bounds these kernels).  DM = 192 has
  // 12 channel tiles: 6 waves take 2 each, the other 2 waves only join the barriers (and the attention stage).
  constexpr int WN = (DM % 128 == 0) ? 8 : (sizeof(T) == 4 ? 6 : DHW_WN192), WM = (DM % 128 == 0 || sizeof(T) == 4) ? 1 : DHW_WM192;
  constexpr int MT = BM / WM / 16, NT = DM / WN / 16, H = DM / 64, KC = DM / 32;
  static_assert(NT * WN * 16 == DM, "channel tiles must divide over the waves");
  const int tid = body_tid(), lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // DHW_ENC_DUP (round 5): where the layout leaves waves without channels (DM = 192: 6 x 2 tiles on 8 waves), the spare waves REPEAT waves 0, 1 — the
  // same tiles, the same values, written to the same LDS addresses — instead of skipping the stages.  With a run-time `if (act)` around every stage hipcc's
  // s_waitcnt bookkeeping loses the weight ring at each join and drains it in front of every main loop (vmcnt(2) / (1) / (0) behind the barriers of the
  // d = 192 kernels where the others wait vmcnt(15 .. 26)); with `act` folded away the code is the straight line of the 8-wave layouts.
  constexpr bool DUP = DHW_ENC_DUP != 0 && sizeof(T) == 2 && WM == 1 && WN < 8;
  const bool act = DUP ? true : (WN * WM == 8 || wave < WN * WM);   // (DM = 192, 6 x 1 without DUP: waves 6, 7 own no channels in the GEMM stages)
  const int wm = DUP ? 0 : (act ? wave / WN : 0), wn = DUP ? wave % WN : (act ? wave % WN : 0);
  const int S = tile_stride<T>(DM);
  char* XR = m.XR;
  char* QR = m.QR;
  float* red = m.red;
  const float* gam = p.film + (size_t)b * p.film_bs;
  const float* bet = gam + p.film_tot;
  const int row0 = wm * (BM / WM), ntile0 = wn * NT;
  const int n0 = ntile0 * 16 + 4 * g;                          // this lane's first channel
  const size_t wlane = ((size_t)ntile0 * KC * 64 + lane) * 8;  // this wave/lane's offset into a packed [DM x DM] block
  const char* xop = XR + (row0 + l15) * S + g * 8 * ES;
  const char* qop = QR + (row0 + l15) * S + g * 8 * ES;

#ifndef DHW_ENC_XSTREAM
#define DHW_ENC_XSTREAM 0   // measured: 19.05 vs 18.98 ms per 60-step batch (profiles/r04_xstream_ab.log) -> off
#endif
#ifndef DHW_ENC_EARLYFILL
#define DHW_ENC_EARLYFILL 0   // measured 19.13 vs 19.08 ms (profiles/r04_earlyfill_ab.log): the prefetch queues in front of the K / V block the attention waits for
#endif
  // DHW_ENC_EARLYA (round 5): the dense stage's first weight fragments requested in the q1 epilogue, IN FRONT of the cross attention — here, unlike in
  // enc_bc (DHW_ENC_EARLYFILL, r4: slower), no K / V loads follow them: the text K / V tiles were staged at the top of the kernel, so the request
  // rides under the attention's LDS / MFMA / softmax work instead of standing as one burst between the attention and its barrier.
  // MEASURED: 17.466 vs 17.454 ms without (profiles/r05_spread_ab.log, r5am): no gain — the attention's own LDS traffic and the request share the wave's issue.  Off.
#ifndef DHW_ENC_EARLYA
#define DHW_ENC_EARLYA 0
#endif
  constexpr bool EARLYA = sizeof(T) == 2 && (DHW_ENC_EARLYFILL || (DHW_ENC_EARLYA && !(DM == 384 && BM >= 32)));   // (d = 384 on 32-row tiles: the ring's registers across the attention spill)
  constexpr bool XS = sizeof(T) == 2 && DHW_ENC_XSTREAM && !(DM == 384 && BM >= 32);   // (d = 384 with 32-row tiles: two accumulator rows + the ring spill)   // cross-stage weight stream with whole-stage rings (gemm_core.h, run_x)
  constexpr int XDE = KC <= 8 ? KC : 8;   // ring depth (chunks) of the cross-stage stream: a whole stage at d = 192 / 256, 8 of 12 chunks at d = 384
#ifndef DHW_RINGA384
#define DHW_RINGA384 24   // fragments in flight per wave in enc_a's d = 384 stages on 16-row tiles (experiments: 30 / 36 = a whole stage)
#endif
  constexpr int RINGA = sizeof(T) == 4 ? 12 : (DM == 384 && BM == 16 ? DHW_RINGA384 : 24);
  typedef WRing<T, NT, (XS ? XDE * NT : RINGA), (XS ? XDE : (RINGA + NT - 1) / NT)> RingT;
  RingT ring;
#ifndef DHW_ENC_QALT
#define DHW_ENC_QALT 1
#endif
#ifndef DHW_ENC_SPREAD
#define DHW_ENC_SPREAD 11  // bit 0: enc_bc's stages (enc_bc_core.h), bit 1: enc_a's
#endif
  constexpr bool SPREADA = sizeof(T) == 2 && (DHW_ENC_SPREAD & 2) != 0 && !XS;
  constexpr bool SPREADQ = sizeof(T) == 2 && (DHW_ENC_SPREAD & 8) != 0 && !XS && MT > 1;   // bit 3: the q / k / v chunks of the multi-row-tile layouts
  constexpr int FCHA = RingT::template fill_chunks<KC>(), FQA = (FCHA + 3) / 4;
  EpiParams<NT> ep;
  ENC_STAMP(0);
  // x tile, first block of text keys and of text values (usually all of them): every load is requested before the first
  // LDS store, so the three tiles cost one memory round trip together; its latency hides behind nothing (the q1 GEMM
  // needs x), the K / V tiles are only needed after q1.
  constexpr bool VROW = sizeof(T) == 2;   // V tiles [keys][channels] (bf16) or V^T [channels][keys] (fp32), attn_core.h
  constexpr int KBC = 32, SKC = tile_stride<T>(DM), SVC = VROW ? SKC : KBC * ES + OPAD<T>;
  constexpr int EPV = 16 / ES, CPR = DM / EPV, PPR = KBC / EPV;
  constexpr int UX = (BM * CPR + 511) / 512, UK = (KBC * CPR + 511) / 512, UV = VROW ? UK : (DM * PPR + 511) / 512;
  constexpr int QKS = qkv_stride<T, DM>();
  char* KT = m.KT;
  char* VT = m.VT;
  const T* k1s = reinterpret_cast<const T*>(p.k1) + (size_t)b * p.Lt * DM;
  const T* v1s = reinterpret_cast<const T*>(p.vt1) + (VROW ? (size_t)b * p.Lt * DM : (size_t)b * DM * p.lpadT);   // v1 [Lt][DM] / V^T [DM][lpadT]
  constexpr bool PLDS = enc_plds<T>();
  const float* PL = m.PL;   // [b_q1 | b_d1 | gamma1 | beta1 | b_qkv2 x 3], DM floats each
  // DHW_ENC_KVLATE (round 5, bf16): the text K / V tiles are needed only behind q1, so they are requested BEHIND the q1 weights and written to LDS
  // behind the q1 stage: the kernel (or, chained behind a ConvBlock / an enc_bc, this half) starts its first GEMM after ONE memory round trip —
  // parameters + x tile, with the weights right behind them — instead of two (tiles, then weights requested once the tiles had been stored).
  // MEASURED: 19.284 ms against 19.239 without and 19.270 for the build before (profiles/r05_spread_ab.log, r5an): no gain — the bytes through the CU's L1 path
  // are the same and that path, not the number of round trips, sets the start-up time.  Off.
#ifndef DHW_ENC_KVLATE
#define DHW_ENC_KVLATE 0
#endif
  constexpr bool KVLATE = sizeof(T) == 2 && DHW_ENC_KVLATE != 0;
  const int64_t* trow = p.text ? p.text + (size_t)b * p.Lt : nullptr;
  PadMask<KBC> pad;   // key-padding mask of the first block: requested here, used after q1
  struct KVRegs { CopyRegs<UK> ck; CopyRegs<UV> cv; };
  KVRegs kv_late;   // (KVLATE: alive across the q1 stage)
  auto kv_store = [&](KVRegs& kv) {
    CopyRegs<UK>& ck = kv.ck;
    CopyRegs<UV>& cv = kv.cv;
    ck.store(KBC * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR; return reinterpret_cast<uint4*>(KT + r * SKC + cc * 16); },
             [&](int id) { return id / CPR < p.Lt; });
    if constexpr (VROW)
      cv.store(KBC * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR; return reinterpret_cast<uint4*>(VT + r * SVC + cc * 16); },
               [&](int id) { return id / CPR < p.Lt; });
    else
      cv.store_to(DM * PPR, tid, 512, [&](int id, const uint4& v) { const int ch = id / PPR, part = id - ch * PPR; vt_store_piece<T>(VT + ch * SVC, part, v); },
                  [&](int id) { return (id % PPR + 1) * EPV <= p.lpadT; });
  };
  {
    KVRegs kv_now;
    KVRegs& kv = [&]() -> KVRegs& { if constexpr (KVLATE) return kv_late; else return kv_now; }();
    CopyRegs<UK>& ck = kv.ck;
    CopyRegs<UV>& cv = kv.cv;
    CopyRegs<UX> cx;
    ParamStage<7> cp;
    if constexpr (PLDS) {
      cp.template load<DM>(tid, p.b_q1, p.b_d1, gam + p.f1, bet + p.f1, p.b_qkv2, p.b_qkv2 + DM, p.b_qkv2 + 2 * DM);
    }
    const T* xs = reinterpret_cast<const T*>(p.x);
    if (p.x)
      cx.load(BM * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR;
                                                return reinterpret_cast<const uint4*>(xs + (size_t)(b * p.Lk + (m0 + r < p.Lk ? m0 + r : p.Lk - 1)) * DM + cc * EPV); });
    if constexpr (KVLATE) { if (act) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_q1) + wlane); }
    ck.load(KBC * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR;
                                               return reinterpret_cast<const uint4*>(k1s + (size_t)(r < p.Lt ? r : p.Lt - 1) * DM + cc * EPV); });
    if constexpr (VROW)
      cv.load(KBC * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR;
                                                 return reinterpret_cast<const uint4*>(v1s + (size_t)(r < p.Lt ? r : p.Lt - 1) * DM + cc * EPV); });
    else
      cv.load(DM * PPR, tid, 512, [&](int id) { const int ch = id / PPR, part = id - ch * PPR;
                                                return reinterpret_cast<const uint4*>(v1s + (size_t)ch * p.lpadT + ((part + 1) * EPV <= p.lpadT ? part * EPV : 0)); });
    if (p.x)
      cx.store(BM * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR; return reinterpret_cast<uint4*>(XR + r * S + cc * 16); },
               [&](int id) { return m0 + id / CPR < p.Lk; });
    if constexpr (!KVLATE) kv_store(kv);
    if constexpr (PLDS) cp.template store<DM>(m.PL, tid);
  }
  pad.load(lane, trow, 0, p.Lt);
  // the q1 weights are requested BEHIND the staging loads: a wave's loads complete in order and the L1 miss queue is
  // shared, so a 24 KB-per-wave prefetch in front of them delays the tiles everything waits for
  if (act) {
    if constexpr (!KVLATE) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_q1) + wlane);
    if constexpr (!PLDS) ep.load_bias(p.b_q1, n0);
  }
  lds_barrier();
  ENC_STAMP(1);

  if (act) {  // ---- q1 = Wq x + b + PE·Wq[row]
    f32x4 acc[NT][MT];
    acc_zero(acc);
    f32x4 pb[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        pb[i][j] = *reinterpret_cast<const f32x4*>(p.pb_q1 + (unsigned)((m0 + row0 + j * 16 + l15) * DM + n0 + 16 * i));
    ring.template run_s<MT, KC>(acc, xop, S, KC);
    ENC_STAMP(8);
    if constexpr (PLDS) ep.lds_bias(PL, n0);
    // the dense stage's first weight fragments: requested in front of the cross attention (the vector-memory path is idle during
    // it), not behind it — in halves around the q1 tile's stores
    if constexpr (EARLYA) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_d1) + wlane); ring.template fill_range<KC, 0, 2 * FQA>(); }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i] + pb[i][j];
    enc_store_tiles<T, NT, MT>(lane, QR, S, row0, n0, acc);
    ENC_STAMP(9);
    if constexpr (EARLYA) ring.template fill_range<KC, 2 * FQA, FCHA>();
  }
  if constexpr (KVLATE) kv_store(kv_late);
  lds_barrier();
  ENC_STAMP(2);

  {  // ---- cross attention over the Lt text keys (K/V staged in LDS, 32 keys per block); a1 overwrites q1 in place
    constexpr int RG = BM / 16, HS = 8 / RG, UMAX = (H + HS - 1) / HS;
    constexpr int SK = SKC, SV = SVC;
    const int rg = wave % RG, hs = wave / RG;
    Frag<T> qf[UMAX][2];
    float mr[UMAX], lr[UMAX];
    f32x4 o[UMAX][4];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = hs + u * HS;
      const T* qrow = reinterpret_cast<const T*>(QR + (rg * 16 + l15) * S) + (h < H ? h : 0) * 64 + 8 * g;
      qf[u][0] = frag_load(qrow);
      qf[u][1] = frag_load(qrow + 32);
      mr[u] = -INFINITY;
      lr[u] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[u][t] = (f32x4){0, 0, 0, 0};
    }
    for (int kb = 0; kb < p.Lt; kb += KBC) {
      if (kb) pad.load(lane, trow, kb, p.Lt);
      const unsigned padbits = pad.template bits<sizeof(T) == 2>();
      if (kb) {
        attn_stage_kv<T, KBC>(KT, SK, VT, SV, k1s, DM, v1s, VROW ? DM : p.lpadT, DM, kb, p.Lt, tid, 512);
        lds_barrier();
      }
      attn_units<T, KBC, true, UMAX>(lane, qf, KT, SK, VT, SV, hs, HS, H, kb, padbits, p.Lt, mr, lr, o);
      if (kb + KBC < p.Lt) lds_barrier();   // the staging tiles are rewritten by the next block (after the last one the
                                             // barrier behind the a1 store below does)
    }
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = hs + u * HS;
      float l = lr[u];
      l = xg_sum(l);
      const float inv = 1.0f / l;
      if (h < H) {   // (wave-uniform)
        T* dst = reinterpret_cast<T*>(QR + (rg * 16 + l15) * S) + h * 64 + 4 * g;
        store_pair(lane, dst, dst + 16, o[u][0] * inv, o[u][1] * inv);
        store_pair(lane, dst + 32, dst + 48, o[u][2] * inv, o[u][3] * inv);
      }
    }
  }
  if (act) {
    if constexpr (!EARLYA) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d1) + wlane);   // in flight across the barrier
    if constexpr (!PLDS) ep.load(p.b_d1, gam + p.f1, bet + p.f1, n0);
  }
  lds_barrier();
  ENC_STAMP(3);

  {  // ---- x2 = FiLM1(LN(Wd a1 + b)) + x
    f32x4 acc[NT][MT];
    acc_zero(acc);
    if (act) {
      if constexpr (XS) ring.template run_x<MT, KC, 0, KC>(acc, qop, S, KC, reinterpret_cast<const T*>(p.w_qkv2) + wlane);   // + the q2 chunk's weights
      else ring.template run_s<MT, KC>(acc, qop, S, KC);
      ENC_STAMP(10);
      // the q2 chunk's first fragments: spread through the LayerNorm epilogue (SPREADA, as enc_bc_core.h DHW_ENC_SPREAD) or one burst in front of it
      if constexpr (SPREADA) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_qkv2) + wlane); ring.template fill_range<KC, 0, FQA>(); }
      else if constexpr (!XS) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_qkv2) + wlane);   // q2 chunk: flies during the LayerNorm epilogue
      if constexpr (PLDS) ep.lds(PL + DM, PL + 2 * DM, PL + 3 * DM, n0);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i];
      if constexpr (SPREADA) ring.template fill_range<KC, FQA, 2 * FQA>();
    }
    if constexpr (SPREADA) ln_rows<T, MT, NT, WN, BM>(acc, red, wn, row0, lane, DM, act, [&]() { if (act) ring.template fill_range<KC, 2 * FQA, 3 * FQA>(); });
    else ln_rows<T, MT, NT, WN, BM>(acc, red, wn, row0, lane, DM, act);
    if constexpr (SPREADA) { if (act) ring.template fill_range<KC, 3 * FQA, FCHA>(); }
    ENC_STAMP(11);
    if (act) {
      // x2 replaces x in LDS (x is no longer an operand: q1 finished two barriers ago).  A 16-byte paired store covers the
      // partner lane's 4 channels too; its data depends, through the lane swap, on both lanes' reads of x, so no store of a
      // pair is issued ahead of them.
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = acc[i][j] * ep.gam[i] + ep.bet[i] + load4(reinterpret_cast<const T*>(XR + (row0 + j * 16 + l15) * S) + n0 + 16 * i);
      // (DUP: the spare waves repeat waves 0, 1 — this is the one stage that updates a tile IN PLACE, so every wave's reads of x come before any
      // wave's stores of x2)
      if constexpr (DUP) lds_barrier();
      enc_store_tiles<T, NT, MT>(lane, XR, S, row0, n0, acc);
    }
  }
  lds_barrier();
  ENC_STAMP(4);
  if constexpr (DHW_COPY_UNROLL != 0) tile_copy_out_u<T, BM, DM, 512>(XR, S, reinterpret_cast<T*>(p.x2) + (size_t)(b * p.Lk + m0) * DM, DM, rows_valid, tid);
  else tile_copy_out<T>(XR, S, reinterpret_cast<T*>(p.x2) + (size_t)(b * p.Lk + m0) * DM, DM, rows_valid, DM, tid, 512);

  // ---- [q2 | k2 | v2] = W x2 + b (+ PE·W for q, k), one DM-wide chunk at a time.  The opaque zero keeps hipcc
  // from treating the x2 fragment reads / store addresses as chunk-invariant and hoisting (then spilling) them.
  // bf16: all three chunks are rows of the [.., 3 DM] buffer; fp32: v2 goes out transposed (V^T [DM][lpadX]).
  // (three calls of one body with compile-time chunk index and ring rotation: at d = 384 the rotation alternates, gemm_core.h)
  auto qkv_chunk = [&](auto CHUNK_, auto ROT_) __attribute__((always_inline)) {
    constexpr int chunk = decltype(CHUNK_)::value, ROTC = decltype(ROT_)::value;
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    f32x4 acc[NT][MT];
    acc_zero(acc);
    f32x4 pb[NT][MT];
    const bool rows_out = chunk < 2 || VROW;   // this chunk is written as rows
    if (act) {
      if constexpr (!PLDS) ep.load_bias(p.b_qkv2 + chunk * DM, n0 + opaque);
      if (chunk < 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            pb[i][j] = *reinterpret_cast<const f32x4*>(p.pb_qk2 + (unsigned)((m0 + row0 + j * 16 + l15) * 2 * DM + chunk * DM + n0 + 16 * i + opaque));
      } else {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) pb[i][j] = (f32x4){0, 0, 0, 0};
      }
      if constexpr (XS) {
        if constexpr (chunk < 2) ring.template run_x<MT, KC, ROTC, KC>(acc, xop + opaque, S, KC, reinterpret_cast<const T*>(p.w_qkv2) + (size_t)(chunk + 1) * DM * DM + wlane);
        else ring.template run_x<MT, KC, ROTC, 0>(acc, xop + opaque, S, KC);
      } else {
        ring.template run_s<MT, KC>(acc, xop + opaque, S, KC);
      }
      ENC_STAMP(12 + chunk);
      if constexpr (PLDS) ep.lds_bias(PL + (4 + chunk) * DM, n0 + opaque);
      if constexpr (SPREADA && MT == 1) {
        // (the single-row-tile form stores straight from the accumulators below: its few VALU + store instructions go between the request's halves)
        if (chunk < 2) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_qkv2) + (size_t)(chunk + 1) * DM * DM + wlane); ring.template fill_range<KC, 0, 2 * FQA>(); }
      } else if constexpr (SPREADQ) {
        // (several row tiles per wave — the 64- / 32-row tiles of d = 192 / 256: the chunk goes out through the LDS staging tile; the next chunk's
        // request in quarters around the pieces of that path, DHW_ENC_SPREAD bit 3)
        if (chunk < 2) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_qkv2) + (size_t)(chunk + 1) * DM * DM + wlane); ring.template fill_range<KC, 0, FQA>(); }
      } else if constexpr (!XS) { if (chunk < 2) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_qkv2) + (size_t)(chunk + 1) * DM * DM + wlane); }
    }
    if (rows_out && MT == 1) {
      // one row tile per wave (3 store instructions per chunk): straight from the accumulators, no LDS round trip
      const int r = row0 + l15;
      if (act && r < rows_valid) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          store4(reinterpret_cast<T*>(p.qk2) + (unsigned)((b * p.Lk + m0 + r) * QKS + chunk * DM + n0 + 16 * i + opaque), acc[i][0] + ep.bias[i] + pb[i][0]);
      }
      if constexpr (SPREADA && MT == 1) { if (act && chunk < 2) ring.template fill_range<KC, 2 * FQA, FCHA>(); }
      ENC_STAMP(5 + chunk);
      return;
    }
    // DHW_ENC_QALT (round 5, bf16): the chunks alternate between TWO staging tiles — q and v through the q1 / a1 tile, k through the text K tile
    // (dead since the cross-attention) — so the only barrier a chunk needs is the one between its LDS stores and its copy-out: the tile a chunk
    // writes was last read two chunks (one barrier) earlier; q's tile by dense1's main loop, in front of the LayerNorm's barriers.  3 barriers
    // instead of 6 for the three chunks.
    constexpr bool QALT = sizeof(T) == 2 && DHW_ENC_QALT != 0 && VROW;
    char* const QT = QALT && chunk == 1 ? m.KT : QR;
    static_assert(!QALT || (size_t)BM * tile_stride<T>(DM) <= enc_a_text_kv_bytes<T, DM, BM>(), "a staging tile inside the text K / V tiles");
    if constexpr (!QALT) lds_barrier();   // the staging tile (q1/a1 region, or x2+q1 regions for V^T) is free: every wave is past its readers
    if (rows_out) {
      // q2 / k2 (/ v2) chunk -> LDS tile [row][DM] -> coalesced rows of the [.., QKS] buffer
      if (act) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i] + pb[i][j];
        if constexpr (SPREADQ) { if (chunk < 2) ring.template fill_range<KC, FQA, 2 * FQA>(); }
        enc_store_tiles<T, NT, MT>(lane, QT, S, row0, n0 + opaque, acc);
        if constexpr (SPREADQ) { if (chunk < 2) ring.template fill_range<KC, 2 * FQA, 3 * FQA>(); }
      }
      lds_barrier();
      if constexpr (DHW_COPY_UNROLL != 0) tile_copy_out_u<T, BM, DM, 512>(QT, S, reinterpret_cast<T*>(p.qk2) + (size_t)(b * p.Lk + m0) * QKS + chunk * DM, QKS, rows_valid, tid);
      else tile_copy_out<T>(QT, S, reinterpret_cast<T*>(p.qk2) + (size_t)(b * p.Lk + m0) * QKS + chunk * DM, QKS, rows_valid, DM, tid, 512);
      if constexpr (SPREADQ) { if (act && chunk < 2) ring.template fill_range<KC, 3 * FQA, FCHA>(); }
    } else if constexpr (!VROW) {
      // v2 chunk -> LDS tile [channel][key] (key-contiguous, zero past the valid rows) -> coalesced rows of vt2
      constexpr int SV = BM * ES + 16;
      char* VS = m.VS;
      if (act) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const int rl = row0 + j * 16 + l15;
            const f32x4 v = acc[i][j] + ep.bias[i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              *reinterpret_cast<T*>(VS + (n0 + 16 * i + k) * SV + rl * ES) = from_f<T>(rl < rows_valid ? v[k] : 0.f);
          }
      }
      lds_barrier();
      // keys this tile owns: its valid rows; the sample's last tile also zero-fills the padding up to lpadX
      constexpr int KPP = VPIECE / ES, PPR = BM / KPP;   // keys per piece, pieces per channel row
      const int klimit = m0 + rows_valid >= p.Lk ? min(BM, p.lpadX - m0) : rows_valid;
      for (int id = tid; id < DM * PPR; id += 512) {
        const int ch = id / PPR, part = id - ch * PPR;
        if ((part + 1) * KPP <= klimit) {
          T* dst = reinterpret_cast<T*>(p.vt2) + ((size_t)b * DM + ch) * p.lpadX + m0 + part * KPP;
          const char* src = VS + ch * SV + part * VPIECE;
          if constexpr (VPIECE == 16) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
          else *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
        }
      }
    }
    ENC_STAMP(5 + chunk);
  };
  constexpr int RC0 = XS ? RingT::template next_rot<KC, 0>() : 0, RC1 = XS ? RingT::template next_rot<KC, RC0>() : 0, RC2 = XS ? RingT::template next_rot<KC, RC1>() : 0;
  qkv_chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, RC0>{});
  qkv_chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, RC1>{});
  qkv_chunk(std::integral_constant<int, 2>{}, std::integral_constant<int, RC2>{});
}
