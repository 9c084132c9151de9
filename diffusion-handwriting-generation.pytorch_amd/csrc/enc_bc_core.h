// enc_bc_core.h — the second half of an EncoderLayer's stroke side (reference model.py:49-58) for one row tile of one
// sample as a device function, the stand-alone layout of the first half, and their LDS budgets (see enclayer.hip).
#pragma once
#include "enc_a_core.h"

namespace {

// -DDHW_ENC_XSTREAM=1: the weight stream of the bf16 EncoderLayer stages refilled ACROSS stage boundaries (WRing::run_x: a stage's
// first chunks are requested chunk by chunk during the previous stage's main loop, rings of a whole stage at d = 192 / 256 and
// of 8 of the 12 chunks at d = 384) instead of as one burst behind the main loop.  Bit-identical; built on the per-wave
// timelines' "fill" segments (0.9-1.9 kcycles per stage) and measured no faster: the burst is not additive.
#ifndef DHW_RING384
#define DHW_RING384 24   // weight fragments in flight per wave in enc_bc's d = 384 stages on 16-row tiles (15 in round 3; 30 / 36 = a whole stage: experiments)
#endif
#ifndef DHW_ENC_EARLYFILL
#define DHW_ENC_EARLYFILL 0   // measured 19.13 vs 19.08 ms (profiles/r04_earlyfill_ab.log): the prefetch queues in front of the K / V block the attention waits for
#endif
#ifndef DHW_ENC_XSTREAM
#define DHW_ENC_XSTREAM 0   // measured: 19.05 vs 18.98 ms per 60-step batch (profiles/r04_xstream_ab.log) -> off
#endif



#define STAMP(slot) ENC_STAMP(slot)
// per-wave stamps of enc_bc (diagnostic builds, tools/bench_encw.cpp): shader-clock time of every wave of workgroup 0 at the
// phase boundaries inside the stages -> p.stamps[64 + wave * 32 + slot]
#ifdef DHW_STAMPS
#define WST(slot) do { if (p.stamps && blockIdx.x == 0 && lane == 0) p.stamps[64 + wave * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WST(slot) do { } while (0)
#endif

// enc_a for the BM-row tile at m0 of sample b with the stand-alone LDS layout (x tile | q1 tile | LayerNorm scratch | text K/V | parameters)
template <typename T, int DM, int BM, typename P>
DHW_DEV void enc_a_tile(const P& p, const int b, const int m0, char* smem) {
  const int S = tile_stride<T>(DM);
  EncALds m;
  m.XR = smem;
  m.QR = m.XR + BM * S;
  m.red = reinterpret_cast<float*>(m.QR + BM * S);
  m.KT = reinterpret_cast<char*>(m.red) + 2 * 8 * BM * sizeof(float);
  m.VT = m.KT + 32 * tile_stride<T>(DM);
  m.VS = smem;   // spans the x2 and q1 tiles: DM * (BM * ES + 16) <= 2 * BM * S
  m.PL = reinterpret_cast<float*>(m.KT + enc_a_text_kv_bytes<T, DM, BM>());
  enc_a_body<T, DM, BM>(p, m, b, m0, min(BM, p.Lk - m0));
}

// Self-attention K / V staging.  bf16: 64-key blocks, DOUBLE buffered when two blocks of K [keys][DM] + V^T [DM][keys] fit
// beside the a2 tile (they overlay the x3 / FFN tiles, which are written only after the attention): the next block's
// global loads are in flight during the current block's softmax / MFMA work and only their LDS stores remain afterwards.
// (Single-buffered 128-key blocks spent as long in the two dependent copy round trips per block as in the math:
// profiles/r03_head_enclayer_stage_stamps.log, d = 192: staged 2.7 / 2.0 us, computed 3.3 / 3.3 us.)
// fp32 parity mode: single-buffered 32-key blocks (its tiles are twice as wide).
template <typename T, int DM, int BM>
constexpr size_t self_att_bytes(int kbs, int nbuf, int vpad = 16) {   // a2 tile + nbuf x (K tile + V tile: bf16 [keys][DM] like K, fp32 V^T [DM][keys])
  return (size_t)BM * tile_stride<T>(DM) + nbuf * ((size_t)kbs * tile_stride<T>(DM) + (sizeof(T) == 2 ? (size_t)kbs * tile_stride<T>(DM) : (size_t)DM * (kbs * sizeof(T) + vpad)));
}
template <typename T, int DM, int BM>
constexpr int self_kbs() { return sizeof(T) == 2 ? 64 : 32; }
template <typename T, int DM, int BM>
constexpr bool self_db() { return sizeof(T) == 2 && self_att_bytes<T, DM, BM>(64, 2) <= 160 * 1024; }
// V^T row pad: the conflict-free 32 bytes when the staging buffers still fit, else 16 (attn_core.h)
template <typename T, int DM, int BM>
constexpr int self_vpad() { return sizeof(T) == 2 && self_att_bytes<T, DM, BM>(64, self_db<T, DM, BM>() ? 2 : 1, 32) <= 160 * 1024 ? 32 : 16; }
template <typename T, int DM, int BM>
constexpr size_t lds_bc_tiles() {
  constexpr size_t S = tile_stride<T>(DM);
  constexpr size_t stages = 3 * BM * S + 2 * 8 * BM * sizeof(float), att = self_att_bytes<T, DM, BM>(self_kbs<T, DM, BM>(), self_db<T, DM, BM>() ? 2 : 1, self_vpad<T, DM, BM>());
  return stages > att ? stages : att;
}
// enc_bc's parameter block (bf16 kernels): [b_d2 | gamma2 | beta2 | b_f1 (2 vectors) | b_f2 | gamma3 | beta3].  It sits behind
// the tiles when that fits the 160 KiB; the double-buffered d = 256 variants have 3 KB to spare, there the block is written
// after the last key block into the staging buffer that block does NOT use (free since the previous iteration's barrier).
template <typename T, int DM> constexpr size_t enc_bc_param_bytes() { return enc_plds<T>() ? (size_t)8 * DM * sizeof(float) : 0; }
template <typename T, int DM, int BM>
constexpr bool bc_params_fixed() { return lds_bc_tiles<T, DM, BM>() + enc_bc_param_bytes<T, DM>() <= 160 * 1024; }
template <typename T, int DM, int BM>
constexpr size_t lds_bc_bytes() { return lds_bc_tiles<T, DM, BM>() + (bc_params_fixed<T, DM, BM>() ? enc_bc_param_bytes<T, DM>() : 0); }

#ifndef DHW_RING192
#define DHW_RING192 12   // weight-ring fragments of the d = 192, 64-row variant (4 waves x 3 tiles: 12 accumulators twice in the FFN stages)
#endif
// NEXT: 0, or the EncChain mode compiled into this variant (DN = width of the chained layer)
template <typename T, int DM, int BM, int NEXT>
constexpr size_t lds_bc_chain_bytes() {
  constexpr size_t S = tile_stride<T>(DM), base = 3 * BM * S + 2 * 8 * BM * sizeof(float);
  constexpr size_t chain = NEXT == 1 ? base + enc_a_text_kv_bytes<T, DM, BM>() + enc_a_param_bytes<T, DM>()
                         : NEXT == 2 ? base + enc_a_text_kv_bytes<T, 384, BM / 2>() + enc_a_param_bytes<T, 384>() : 0;
  return chain > lds_bc_bytes<T, DM, BM>() ? chain : lds_bc_bytes<T, DM, BM>();
}

template <typename T, int DM, int BM>
constexpr size_t lds_a_bytes() { return (size_t)2 * BM * tile_stride<T>(DM) + 2 * 8 * BM * sizeof(float) + enc_a_text_kv_bytes<T, DM, BM>() + enc_a_param_bytes<T, DM>(); }

// The second half of the layer for the BM-row tile at m0 of sample b (+ what NEXT chains behind it), as a device function over
// the workgroup's LDS: the per-launch kernel (enclayer.hip) and the persistent per-step kernel (persist.hip) both run it.
template <typename T, int DM, int BM, int NEXT = 0, typename P, typename X>
DHW_DEV void enc_bc_body(const P& p, const X& nx, const int b, const int m0, char* smem) {
  constexpr int ES = sizeof(T);
  constexpr int WN = (DM % 128 == 0) ? 8 : (sizeof(T) == 4 ? 6 : DHW_WN192), WM = (DM % 128 == 0 || sizeof(T) == 4) ? 1 : DHW_WM192;   // as in enc_a_body: all rows per wave, channels split over WN waves
  constexpr int MT = BM / WM / 16, NT = DM / WN / 16, H = DM / 64, KC = DM / 32;
  constexpr bool XS = sizeof(T) == 2 && DHW_ENC_XSTREAM && !(DM == 384 && BM >= 32);   // (d = 384 with 32-row tiles: two accumulator rows + the ring spill)   // cross-stage weight stream (gemm_core.h, run_x); the fp32 parity mode keeps run_s + fill_s
  constexpr int XDE = KC <= 8 ? KC : 8;   // ring depth (chunks) of the cross-stage stream
  constexpr int RING = sizeof(T) == 4 ? 12 : (XS ? XDE * NT : (DM == 384 ? (BM >= 32 ? 15 : DHW_RING384) : (DM == 192 && NT == 3 && BM == 64 ? DHW_RING192 : 24))), RDMAX = XS ? XDE : (RING + NT - 1) / NT;   // (d = 384, 32 rows: two accumulator rows, 24 fragments spill; d = 192 as 4 waves x 3 tiles on 64 rows: the whole stage, 6 chunks — 8 slots spill)
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
  char* R1 = smem;               // a2, later SiLU(x3)
  char* R2 = R1 + BM * S;        // x3
  char* R3 = R2 + BM * S;        // one DM-wide half of the FFN hidden layer
  float* red = reinterpret_cast<float*>(R3 + BM * S);
  const float* gam = p.film + (size_t)b * p.film_bs;
  const float* bet = gam + p.film_tot;
  const int row0 = wm * (BM / WM), ntile0 = wn * NT;
  const int n0 = ntile0 * 16 + 4 * g;
  const size_t wlane = ((size_t)ntile0 * KC * 64 + lane) * 8;
  const char* op1 = R1 + (row0 + l15) * S + g * 8 * ES;
  const char* op3 = R3 + (row0 + l15) * S + g * 8 * ES;

  WRing<T, NT, RING, RDMAX> ring;
#ifndef DHW_ENC_SPREAD
#define DHW_ENC_SPREAD 11  // bit 0: this kernel's stages, bit 1: enc_a's (enc_a_core.h)
#endif
  constexpr bool SPREAD = sizeof(T) == 2 && (DHW_ENC_SPREAD & 1) != 0;
  constexpr bool SPREAD_ATT = sizeof(T) == 2 && (DHW_ENC_SPREAD & 4) != 0 && !DHW_ENC_EARLYFILL && !DHW_ENC_XSTREAM;   // bit 2: the dense stage's request around the last key block
  constexpr int FCH = WRing<T, NT, RING, RDMAX>::template fill_chunks<KC>(), FQ = (FCH + 3) / 4;   // ring slots a stage's first request fills, and a quarter of them
  // slot rotation of the stages (gemm_core.h): dense 0, FFN first halves ROT1, second halves ROT2 — and back to ROT1, so the loop over the
  // two halves of the hidden layer stays rolled
  typedef WRing<T, NT, RING, RDMAX> RingT;
  constexpr int ROT1 = XS ? RingT::template next_rot<KC, 0>() : 0, ROT2 = XS ? RingT::template next_rot<KC, ROT1>() : 0;
  static_assert(!XS || RingT::template next_rot<KC, ROT2>() == ROT1, "FFN loop rotation");
  EpiParams<NT> ep;
  constexpr bool PLDS = enc_plds<T>(), PLFIX = bc_params_fixed<T, DM, BM>();
  constexpr bool EARLY = sizeof(T) == 2 && DHW_ENC_EARLYFILL;   // the post-attention stage's weight prefetch issued in front of the attention
  float* PL = reinterpret_cast<float*>(smem + lds_bc_tiles<T, DM, BM>());   // (PLFIX; else chosen behind the attention loop)
  ParamStage<8> cp;
  // (a macro, not a lambda: with `cp` captured by a closure hipcc kept it in scratch memory)
#define BC_PARAMS_REQUEST() cp.template load<DM>(tid, p.b_d2, gam + p.f2, bet + p.f2, p.b_f1, p.b_f1 + DM, p.b_f2, gam + p.f3, bet + p.f3)
  STAMP(16);
  WST(0);
  DHW_STAMP_IF(p.stamps && blockIdx.x == 0 && threadIdx.x == 0, 40, __builtin_amdgcn_s_memtime());
  if (!(p.dbg & 1)) {  // ---- self attention over all Lk rows of the sample (K/V staged in LDS, 64 keys per block) -> a2 in LDS
    constexpr int RG = BM / 16, HS = 8 / RG, UMAX = (H + HS - 1) / HS, KBS = self_kbs<T, DM, BM>();
    constexpr bool DB = self_db<T, DM, BM>();
    constexpr bool VROW = sizeof(T) == 2;   // V tile [keys][DM] read with the transposing LDS read (bf16) or V^T [DM][keys] (fp32): attn_core.h
    constexpr int SK = tile_stride<T>(DM), SV = VROW ? SK : KBS * ES + self_vpad<T, DM, BM>();
    constexpr int BUFB = KBS * SK + (VROW ? KBS * SV : DM * SV);   // one staged block: K tile, then V tile
    constexpr int QKS = qkv_stride<T, DM>();
    const int rg = wave % RG, hs = wave / RG;
    // DHW_ATT_KSPLIT (round 5).  d = 192 with 64-row tiles: 4 row groups x 3 heads = 12 units on 8 waves — waves 0-3 ran heads 0 and 2, waves 4-7
    // head 1 alone, and every key block lasted two units.  Here the two waves of a row group SHARE the third head's keys: wave (rg, 0) takes
    // head 0 and keys [0, 32) of head 2 of every block, wave (rg, 1) head 1 and keys [32, 64) of head 2 — 1.5 units each — and the two
    // partial (max, sum, output) states of head 2 are merged once behind the last block (the split-key combination of the running softmax).
    // The 32-row tiles (2 row groups x 4 head slots, one of them idle) compute head 2 the same way — slot 2 the first halves, slot 3 the
    // second — so that a row's arithmetic stays independent of the tile the launcher picks (a shard of a batch == the same samples inside
    // it, bit for bit).  Against DHW_ATT_KSPLIT=0 the summation order of head 2's softmax differs: equal to fp32 rounding, not bit for bit.
    // MEASURED NEUTRAL (17.959 vs 17.964 ms same-box, profiles/r05_spread_ab.log r5ad_ks; parity suite green with it on): the stage is bound by
    // the SIMD's total softmax VALU work — 3 units per SIMD and block either way, ~1 kcycle of vector issue each — not by the longest wave.  Off.
#ifndef DHW_ATT_KSPLIT
#define DHW_ATT_KSPLIT 0
#endif
    constexpr bool KSPLIT = sizeof(T) == 2 && DHW_ATT_KSPLIT != 0 && H == 3 && (HS == 2 || HS == 4) && KBS == 64 && DB && PLFIX;
    constexpr int SU = HS == 2 ? 1 : 0;                       // the state slot of a wave's share of head 2
    const bool ks_full = HS == 2 || hs < 2;                   // this wave runs a whole head (slot 0: head hs)
    const bool ks_part = HS == 2 || hs >= 2;                  // this wave runs one half of head 2's keys ..
    const int ks_half = HS == 2 ? hs : hs - 2;                // .. this one
    auto unit_head = [&](int u) { return !KSPLIT ? hs + u * HS : (HS == 2 ? (u == 1 ? 2 : hs) : (hs >= 2 ? 2 : hs)); };
    const T* qk = reinterpret_cast<const T*>(p.qk2);
    const T* ksrc = qk + (size_t)b * p.Lk * QKS + DM;
    const T* vsrc = VROW ? ksrc + DM : reinterpret_cast<const T*>(p.vt2) + (size_t)b * DM * p.lpadX;
    constexpr int EPV = 16 / ES, CPR = DM / EPV, PPR = KBS / EPV;
    constexpr int UK = (KBS * CPR + 511) / 512, UV = VROW ? UK : (DM * PPR + 511) / 512;
    CopyRegs<UK> ck;
    CopyRegs<UV> cv;
    // K rows [kb, kb + KBS) x DM channels and V^T rows [0, DM) x keys [kb, kb + KBS): requested at clamped (valid) addresses;
    // K rows at or past Lk and V^T pieces past lpadX are zero-filled by the store
    auto request = [&](int kb) {
      ck.load(KBS * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR;
                                                 return reinterpret_cast<const uint4*>(ksrc + (size_t)min(kb + r, p.Lk - 1) * QKS + cc * EPV); });
      if constexpr (VROW)
        cv.load(KBS * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR;
                                                   return reinterpret_cast<const uint4*>(vsrc + (size_t)min(kb + r, p.Lk - 1) * QKS + cc * EPV); });
      else
        cv.load(DM * PPR, tid, 512, [&](int id) { const int ch = id / PPR, part = id - ch * PPR;
                                                  return reinterpret_cast<const uint4*>(vsrc + (size_t)ch * p.lpadX + (kb + (part + 1) * EPV <= p.lpadX ? kb + part * EPV : 0)); });
    };
    auto commit = [&](int kb, char* KT, char* VT) {
      ck.store(KBS * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR; return reinterpret_cast<uint4*>(KT + r * SK + cc * 16); },
               [&](int id) { return kb + id / CPR < p.Lk; });
      if constexpr (VROW)   // (V rows at or past Lk are zero: their softmax weights are exactly 0, the products must stay finite)
        cv.store(KBS * CPR, tid, 512, [&](int id) { const int r = id / CPR, cc = id - r * CPR; return reinterpret_cast<uint4*>(VT + r * SV + cc * 16); },
                 [&](int id) { return kb + id / CPR < p.Lk; });
      else
        cv.store_to(DM * PPR, tid, 512, [&](int id, const uint4& v) { const int ch = id / PPR, part = id - ch * PPR; vt_store_piece<T>(VT + ch * SV, part, v); },
                    [&](int id) { return kb + (id % PPR + 1) * EPV <= p.lpadX; });
    };
    Frag<T> qf[UMAX][2];
    float mr[UMAX], lr[UMAX];
    f32x4 o[UMAX][4];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = unit_head(u);
      const T* qrow = qk + (size_t)(b * p.Lk + m0 + rg * 16 + l15) * QKS + (h < H ? h : 0) * 64 + 8 * g;
      qf[u][0] = frag_load(qrow);
      qf[u][1] = frag_load(qrow + 32);
      mr[u] = -INFINITY;
      lr[u] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[u][t] = (f32x4){0, 0, 0, 0};
    }
    if constexpr (PLDS) BC_PARAMS_REQUEST();   // (!PLFIX: held in registers across the key blocks, 8 VGPRs)
    request(0);   // (one round trip together with the q fragments)
    // The dense stage's first weight fragments are requested HERE, behind the first K / V block, not behind the attention: the
    // wave waits for that block anyway, the vector-memory path is idle during the attention's LDS / MFMA work, and the stage
    // then starts on a full ring instead of paying the request burst between the attention and its barrier (per-wave
    // timelines: 1.6-2.3 kcycles from "att.end" to the barrier, profiles/r04_encbc_wave_timeline_*.log).  DHW_ENC_EARLYFILL=0: as before.
    if constexpr (EARLY) { if (act) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d2) + wlane); }
    if constexpr (PLDS && PLFIX) cp.template store<DM>(PL, tid);
    commit(0, R2, R2 + KBS * SK);
    lds_barrier();
    int ib = 0;
    auto key_block = [&](int kb) {
      const bool more = kb + KBS < p.Lk;
      char* KT = R2 + (DB && (ib & 1) ? BUFB : 0);
      char* VT = KT + KBS * SK;
      char* KN = R2 + (DB && !(ib & 1) ? BUFB : 0);   // where the next block goes
      if (DB && more) request(kb + KBS);
      if constexpr (SPREAD_ATT) {
        // the post-attention stage's first weight fragments: half of them in front of the LAST key block's math (no K / V request follows them,
        // so nothing the attention waits for queues behind them — the r4 EARLYFILL form put all of them in front of the FIRST block), the
        // other half behind the a2 stores below
        if (!more && act) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_d2) + wlane); ring.template fill_range<KC, 0, 2 * FQ>(); }
      }
      if (ib < 3) STAMP(26 + 2 * ib);
      if constexpr (KSPLIT) {
        if (ks_full) attn_unit_bf16<KBS>(lane, qf[0], KT, SK, VT, SV, hs, 0, kb, p.Lk, mr[0], lr[0], o[0]);
        const int kh = (KBS / 2) * ks_half;   // (wave-uniform; a half past the end of the sequence is skipped: its state stays empty)
        if (ks_part && kb + kh < p.Lk) attn_unit_bf16<KBS / 2>(lane, qf[SU], KT, SK, VT, SV, 2, kh, kb + kh, p.Lk, mr[SU], lr[SU], o[SU]);
      } else
      attn_units<T, KBS, false, UMAX>(lane, qf, KT, SK, VT, SV, hs, HS, H, kb, 0u, p.Lk, mr, lr, o);
      if (ib < 3) STAMP(27 + 2 * ib);
      if (more) {
        if (!DB) {          // single buffer: every wave must be past its reads before the tiles are rewritten
          lds_barrier();
          request(kb + KBS);
        }
        commit(kb + KBS, KN, KN + KBS * SK);
        lds_barrier();      // next block complete (double buffer: its previous contents were read one iteration ago)
      }
      // (after the last block the barrier behind the a2 store below separates the staging tiles from their next use)
    };
    if constexpr (EARLY) {
      // the first block straight-line, outside the loop: inside it hipcc cannot count the loads in flight (the weight prefetch
      // above among them) and waits for all of them — vmcnt(0) in front of the first block's math, i.e. the prefetch hid nothing
      key_block(0);
      ib = 1;
      for (int kb = KBS; kb < p.Lk; kb += KBS, ++ib) key_block(kb);
    } else {
      for (int kb = 0; kb < p.Lk; kb += KBS, ++ib) key_block(kb);
    }
    if constexpr (PLDS && !PLFIX) {
      // the last block (index ib - 1) was read from buffer (ib - 1) & 1; the other one is free: its readers finished an
      // iteration ago, behind a barrier.  In buffer 0 the block goes behind the stage tiles and the LayerNorm scratch.
      static_assert(DB, "a single staging buffer leaves room for a fixed parameter block");
      static_assert((size_t)2 * BM * tile_stride<T>(DM) + 2 * 8 * BM * sizeof(float) + enc_bc_param_bytes<T, DM>() <= (size_t)BUFB, "parameter block inside buffer 0");
      PL = reinterpret_cast<float*>(((ib - 1) & 1) ? R2 + 2 * BM * S + 2 * 8 * BM * sizeof(float) : R2 + BUFB);
      cp.template store<DM>(PL, tid);
    }
    if constexpr (KSPLIT) {
      // head 2: the wave with the second halves hands its partial state to the one with the first halves through the staging buffer the last
      // block did not use (free since the previous iteration's barrier; the parameter block sits behind both buffers: PLFIX)
      float* MS = reinterpret_cast<float*>(R2 + (((ib - 1) & 1) ? 0 : BUFB)) + (rg * 64 + lane) * 20;
      static_assert((size_t)RG * 64 * 20 * sizeof(float) <= (size_t)BUFB, "merge scratch inside a staging buffer");
      if (ks_part && ks_half == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(MS + 4 * t) = o[SU][t];
        MS[16] = mr[SU];
        MS[17] = lr[SU];
      }
      lds_barrier();
      if (ks_part && ks_half == 0) attn_merge_state(mr[SU], lr[SU], o[SU], MS);
    }
    WST(1);
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = !KSPLIT ? hs + u * HS : (u == SU && ks_part) ? (ks_half == 0 ? 2 : H) : hs;   // (H: nothing to store)
      float l = lr[u];
      l = xg_sum(l);
      const float inv = 1.0f / l;
      if (h < H) {   // (wave-uniform)
        T* dst = reinterpret_cast<T*>(R1 + (rg * 16 + l15) * S) + h * 64 + 4 * g;
        store_pair(lane, dst, dst + 16, o[u][0] * inv, o[u][1] * inv);
        store_pair(lane, dst + 32, dst + 48, o[u][2] * inv, o[u][3] * inv);
      }
    }
  }
  else if constexpr (PLDS) {   // (diagnostic path without the attention stage)
    BC_PARAMS_REQUEST();
    if constexpr (!PLFIX) PL = reinterpret_cast<float*>(R2 + 2 * BM * S + 2 * 8 * BM * sizeof(float));
    cp.template store<DM>(PL, tid);
  }
  if (act) {
    if constexpr (SPREAD_ATT) {
      if (p.dbg & 1) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d2) + wlane);
      else ring.template fill_range<KC, 2 * FQ, FCH>();
    } else if (!EARLY || (p.dbg & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d2) + wlane);   // in flight across the barrier
    if constexpr (!PLDS) ep.load(p.b_d2, gam + p.f2, bet + p.f2, n0);
  }
  WST(2);
  lds_barrier();
  WST(3);
  STAMP(17);

  {  // ---- x3 = FiLM2(LN(x2 + Wd a2 + b))
    f32x4 acc[NT][MT];
    acc_zero(acc);
    if (act) {
      f32x4 res[NT][MT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          // unconditional (rows past the sample read the next sample / the buffer's slack rows and are never written
          // back): a per-lane `r < Lk ? load : 0` compiles to a branch with s_waitcnt vmcnt(0) inside, which drains the
          // weight prefetch issued in front of the barrier — once per row tile
          const int r = m0 + row0 + j * 16 + l15;
          res[i][j] = load4(reinterpret_cast<const T*>(p.x2) + (unsigned)((b * p.Lk + r) * DM + n0 + 16 * i));
        }
      if constexpr (XS) {
        ring.template run_x<MT, KC, 0, KC>(acc, op1, S, KC, reinterpret_cast<const T*>(p.w_f1) + wlane);   // + FFN half 0's first chunks
        WST(4);
      } else {
        ring.template run_s<MT, KC>(acc, op1, S, KC);
        WST(4);
        // FFN half 0's first fragments.  DHW_ENC_SPREAD (round 5): requested a few chunks at a time BETWEEN the pieces of the epilogue below
        // instead of as one burst in front of it — a wave that requests its whole ring at once sits in instruction issue until the CU's L1
        // path has accepted all of it (per-wave timelines at HEAD, profiles/r05_encbc_wave_timeline_d384.log: "f1.fill" 1.2-2.9 kcycles
        // per stage, during which no wave of the workgroup executes the LayerNorm / FiLM / SiLU work that follows)
        if constexpr (SPREAD) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_f1) + wlane); ring.template fill_range<KC, 0, FQ>(); }
        else ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f1) + wlane);   // FFN half 0: flies during the LayerNorm epilogue
      }
      WST(5);
      if constexpr (PLDS) ep.lds(PL, PL + DM, PL + 2 * DM, n0);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i] + res[i][j];
      if constexpr (SPREAD && !XS) ring.template fill_range<KC, FQ, 2 * FQ>();
    }
    STAMP(18);
    WST(6);
    if constexpr (SPREAD && !XS) ln_rows<T, MT, NT, WN, BM>(acc, red, wn, row0, lane, DM, act, [&]() { if (act) ring.template fill_range<KC, 2 * FQ, 3 * FQ>(); });
    else ln_rows<T, MT, NT, WN, BM>(acc, red, wn, row0, lane, DM, act);   // its barriers also fence the a2 reads above
    WST(7);
    if (act) {
      // x3 -> R2, SiLU(x3) -> R1: whole-tile-group SiLU and 16-byte paired stores (epilogue.h)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = acc[i][j] * ep.gam[i] + ep.bet[i];
      enc_store_tiles<T, NT, MT>(lane, R2, S, row0, n0, acc);
      if constexpr (SPREAD && !XS) ring.template fill_range<KC, 3 * FQ, FCH>();
      silu_tiles2<T, NT, MT>(acc);
      enc_store_tiles<T, NT, MT>(lane, R1, S, row0, n0, acc);
    }
  }
  WST(8);
  lds_barrier();
  WST(9);
  STAMP(19);

  // ---- out = FiLM3(LN(W2 SiLU(W1 SiLU(x3) + b1) + b2 + x3)); the 2*DM hidden layer is processed in two halves
  f32x4 acc2[NT][MT];
  acc_zero(acc2);
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    if (act) {
      f32x4 acc[NT][MT];
      acc_zero(acc);
      if constexpr (!PLDS) ep.load_bias(p.b_f1 + hh * DM, n0);
      // K-slice [hh*DM, (hh+1)*DM) of W2 [DM][2*DM]: the next stage's weights
      const T* w2 = reinterpret_cast<const T*>(p.w_f2) + (((size_t)ntile0 * 2 * KC + hh * KC) * 64 + lane) * 8;
      if constexpr (XS) ring.template run_x<MT, KC, ROT1, KC>(acc, op1, S, KC, w2, 2 * KC);
      else ring.template run_s<MT, KC>(acc, op1, S, KC);
      WST(10 + 6 * hh);
      if constexpr (PLDS) ep.lds_bias(PL + (3 + hh) * DM, n0);
      if constexpr (SPREAD && !XS) { ring.template fill_begin<KC>(w2, 2 * KC); ring.template fill_range<KC, 0, FQ>(); }
      else if constexpr (!XS) ring.template fill_s<KC>(w2, 2 * KC);   // flies during the SiLU epilogue and the barrier
      WST(11 + 6 * hh);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i];
      if constexpr (SPREAD && !XS) ring.template fill_range<KC, FQ, 2 * FQ>();
      silu_tiles2<T, NT, MT>(acc);
      if constexpr (SPREAD && !XS) ring.template fill_range<KC, 2 * FQ, 3 * FQ>();
      enc_store_tiles<T, NT, MT>(lane, R3, S, row0, n0, acc);
      if constexpr (SPREAD && !XS) ring.template fill_range<KC, 3 * FQ, FCH>();
    }
    WST(12 + 6 * hh);
    lds_barrier();
    WST(13 + 6 * hh);
    STAMP(20 + 2 * hh);
    if (act) {
      if constexpr (XS) {
        if (hh == 0) ring.template run_x<MT, KC, ROT2, KC>(acc2, op3, S, KC, reinterpret_cast<const T*>(p.w_f1) + (size_t)DM * DM + wlane);   // + FFN half 1
        else ring.template run_x<MT, KC, ROT2, 0>(acc2, op3, S, KC);
        WST(14 + 6 * hh);
      } else {
        ring.template run_s<MT, KC>(acc2, op3, S, KC);
        WST(14 + 6 * hh);
        if (hh == 0) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f1) + (size_t)DM * DM + wlane);   // FFN half 1
      }
    }
    if (hh == 0) lds_barrier();   // R3 is rewritten by the next half (after the last one the LayerNorm barrier below does)
    WST(15 + 6 * hh);
    STAMP(21 + 2 * hh);
  }
  if (act) {
    if constexpr (PLDS) ep.lds(PL + 5 * DM, PL + 6 * DM, PL + 7 * DM, n0);
    else ep.load(p.b_f2, gam + p.f3, bet + p.f3, n0);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
        acc2[i][j] += ep.bias[i] + load4(reinterpret_cast<const T*>(R2 + (row0 + j * 16 + l15) * S) + n0 + 16 * i);
  }
  WST(22);
  ln_rows<T, MT, NT, WN, BM>(acc2, red, wn, row0, lane, DM, act);
  WST(23);
  // out tile -> LDS (R3 is free: the last FFN half was consumed two barriers ago) -> coalesced rows
  if (act) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc2[i][j] = acc2[i][j] * ep.gam[i] + ep.bet[i];
    enc_store_tiles<T, NT, MT>(lane, R3, S, row0, n0, acc2);
  }
  lds_barrier();
  const int rows_valid = min(BM, p.Lk - m0);
  if constexpr (DHW_COPY_UNROLL != 0) tile_copy_out_u<T, BM, DM, 512>(R3, S, reinterpret_cast<T*>(p.out) + (size_t)(b * p.Lk + m0) * DM, DM, rows_valid, tid);
  else tile_copy_out<T>(R3, S, reinterpret_cast<T*>(p.out) + (size_t)(b * p.Lk + m0) * DM, DM, rows_valid, DM, tid, 512);
  if (p.pool)
    tile_copy_out_pool<T>(R3, S, reinterpret_cast<T*>(p.pool) + ((size_t)b * (p.Lk / 2) + m0 / 2) * DM, DM, rows_valid, DM, tid, 512);
  STAMP(24);
  WST(24);
  DHW_STAMP_IF(p.stamps && blockIdx.x == 0 && threadIdx.x == 0, 41, __builtin_amdgcn_s_memtime());

  if constexpr (NEXT == 1) {
    // the next layer's enc_a on the out tile (R3): q1 / a1 in R1, the v2 staging area over R1..R2 (both dead by now)
    EncALds m;
    m.XR = R3; m.QR = R1; m.red = red;
    m.KT = reinterpret_cast<char*>(red) + 2 * 8 * BM * sizeof(float);
    m.VT = m.KT + 32 * tile_stride<T>(DM);
    m.VS = R1;
    m.PL = reinterpret_cast<float*>(m.KT + enc_a_text_kv_bytes<T, DM, BM>());
    enc_a_body<T, DM, BM>(nx.a, m, b, m0, rows_valid);
  } else if constexpr (NEXT == 2) {
    // AvgPool1d(2) of the out tile -> R1; Linear DM -> DN (att_dense) -> x tile of the first attention layer; its enc_a
    constexpr int DN = 384, BN2 = BM / 2, SN = tile_stride<T>(DN), NTN = DN / 8 / 16, MTN = BN2 / 16;
    {
      constexpr int EPV = 16 / ES;
      const int cpr = DM / EPV;
      for (int id = tid; id < BN2 * cpr; id += 512) {
        const int r = id / cpr, cc = id - r * cpr;
        uint4 a = *reinterpret_cast<const uint4*>(R3 + (2 * r) * S + cc * 16);
        const uint4 bq = *reinterpret_cast<const uint4*>(R3 + (2 * r + 1) * S + cc * 16);
        T* ea = reinterpret_cast<T*>(&a);
        const T* eb = reinterpret_cast<const T*>(&bq);
#pragma unroll
        for (int k = 0; k < EPV; ++k) ea[k] = from_f<T>(0.5f * (to_f(ea[k]) + to_f(eb[k])));
        *reinterpret_cast<uint4*>(R1 + r * S + cc * 16) = a;
      }
    }
    WRing<T, NTN> rd;
    EpiParams<NTN> epd;
    const int nt0 = wave * NTN, nn0 = nt0 * 16 + 4 * g;
    rd.template fill_s<KC>(reinterpret_cast<const T*>(nx.w_dense) + ((size_t)nt0 * KC * 64 + lane) * 8);
    epd.load_bias(nx.b_dense, nn0);
    lds_barrier();   // pooled tile complete; every read of the out tile (copy-out, pooling) is done
    char* XN = R2;   // x tile of the chained layer, then its q1 tile: together BM * SN <= 2 * BM * S bytes over R2..R3
    {
      f32x4 acc[NTN][MTN];
      acc_zero(acc);
      rd.template run_s<MTN, KC>(acc, R1 + l15 * S + g * 8 * ES, S, KC);
#pragma unroll
      for (int i = 0; i < NTN; ++i)
#pragma unroll
        for (int j = 0; j < MTN; ++j) acc[i][j] += epd.bias[i];
      enc_store_tiles<T, NTN, MTN>(lane, XN, SN, 0, nn0, acc);
    }
    lds_barrier();
    const int m02 = m0 / 2, rows2 = rows_valid / 2;
    tile_copy_out<T>(XN, SN, reinterpret_cast<T*>(nx.dense_out) + (size_t)(b * (p.Lk / 2) + m02) * DN, DN, rows2, DN, tid, 512);
    EncALds m;
    m.XR = XN; m.QR = XN + BN2 * SN; m.red = red;
    m.KT = reinterpret_cast<char*>(red) + 2 * 8 * BM * sizeof(float);
    m.VT = m.KT + 32 * tile_stride<T>(DN);
    m.VS = XN;
    m.PL = reinterpret_cast<float*>(m.KT + enc_a_text_kv_bytes<T, DN, BN2>());
    enc_a_body<T, DN, BN2>(nx.a, m, b, m02, rows2);
  }
}


}  // namespace
