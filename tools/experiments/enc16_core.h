// enc16_core.h — the attention-level EncoderLayer halves (d = 384, 16-row tiles, bf16) with SPECIALISED waves.
//
// Why (tools/experiments/bench_roles.cpp): at 16 rows per workgroup these layers are bound by the weight stream
// (1.47 MB per half-layer through the CU's 64 B/clk L1 path = 11.8 us), and a wave that issues vector-memory instructions is
// blocked until the path accepts them — so in the symmetric kernels (every wave streams its weights AND does its share of the
// epilogue / LayerNorm / attention work) the two add up: 4.5 us per GEMM stage where the stream alone takes 2.4.  Measured
// in isolation: symmetric 2.4 + 0.45 W us per stage (W = units of vector work), specialised 3.05 us flat.
//
// 12 waves (768 threads, 3 per SIMD, 168 VGPRs):
//   waves 0-7  GEMM waves  : channel tiles 3 gw .. 3 gw + 2; hold a WHOLE stage of weights (36 fragments) in registers; per stage:
//                            MFMA over the operand tile in LDS -> raw fp32 accumulators into an LDS tile (ACC) -> barrier ->
//                            request the next stage's 36 fragments (they stream while the vector waves work);
//   waves 8-11 vector waves: rows 4 vw .. 4 vw + 3, all 384 channels (24 consecutive channels per lane): bias / residual /
//                            LayerNorm (a 16-lane reduction, no cross-wave step) / FiLM / SiLU straight from ACC, the operand
//                            tile of the next stage back into LDS, global outputs straight from registers; the attentions
//                            (wave vw = heads vw and vw + 4) and all staging copies.
// Hand-offs are workgroup barriers that ALL 12 waves execute in the same order: "operands ready" (the GEMM waves arrive when
// their prefetch has been issued) and "accumulators ready" (they arrive straight after the MFMAs).
//
//   enc_bc: self-attention -> dense -> +x2 -> LN -> FiLM2 = x3 -> ffn1 (two halves, SiLU) -> ffn2 -> +x3 -> LN -> FiLM3 = out
//   enc_a (NEXT = 1, on the out tile): q1 -> cross-attention over the text keys -> dense -> LN -> FiLM1 -> +x = x2 -> q2 | k2 | v2
// Same arithmetic per element as enclayer.hip / enc_a_core.h (reference model.py:37-58, attention.py:63-87).
#pragma once
#include "enc_a_core.h"

#ifndef ENC16_ABL
#define ENC16_ABL 0   // diagnostic builds only: bit0 = the GEMM waves request only their first stage of weights; bit1 = the vector waves do not read the accumulator tiles; bit2 = nor write their tiles
#endif

namespace enc16 {

constexpr int DM = 384, BM = 16, H = 6, KC = DM / 32, NTG = 3;
constexpr int S = DM * 2 + 32;          // bf16 operand tile row stride (tile_stride<bf16_t>(384))
constexpr int ACCS = DM * 4 + 32;       // fp32 accumulator tile row stride
constexpr int NV = 256;                 // vector-wave threads

// LDS map (bytes).  TA..TD: operand tiles; ACC0/1: accumulator tiles; KV: the attention staging area.  The self-attention
// K / V^T block (64 keys) overlays TB..ACC1 (dead until the attention is over); the chained layer's text K / V^T block sits
// behind the stage tiles (160 KiB to the byte).
constexpr int O_TA = 0, O_TB = O_TA + BM * S, O_TC = O_TB + BM * S, O_TD = O_TC + BM * S, O_ACC0 = O_TD + BM * S, O_ACC1 = O_ACC0 + BM * ACCS;
constexpr int O_END = O_ACC1 + BM * ACCS;
constexpr int KBS = 64, SKS = S, SVS = KBS * 2 + 32;                       // self-attention block: K [64][S], V^T [384][SVS]
constexpr int O_KVS = O_TB, KVS_BYTES = KBS * SKS + DM * SVS;
constexpr int KBC = 32, SKC = S, SVC = KBC * 2 + 32;                       // text block: K [32][S], V^T [384][SVC]
constexpr int KVC_BYTES = KBC * SKC + DM * SVC;
constexpr int O_SELF_END = O_KVS + KVS_BYTES;
constexpr int O_KVC = O_END;             // (staged after the self-attention: only the stage tiles must stay clear of it)
constexpr int O_STAMPS = O_SELF_END;      // (diagnostic builds, NEXT = 0: 64 stamp slots behind everything)
template <int NEXT> constexpr int lds_bytes() {
  constexpr int stages = NEXT ? O_KVC + KVC_BYTES : O_END;
  return NEXT ? (stages > O_SELF_END ? stages : O_SELF_END) : O_SELF_END + 512;
}

// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad permutes), then the two mirrors
DHW_DEV float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
  return x;
}

// 24 fp32 / 24 bf16 parameters of a lane, requested early (see vector_waves) and consumed behind a barrier
struct Vec24 {
  f32x4 v[6];
  DHW_DEV void load(const float* p) {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const f32x4*>(p + 4 * k);
  }
};
struct Half24 {
  uint4 w[3];
  DHW_DEV void load(const bf16_t* p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = *reinterpret_cast<const uint4*>(p + 8 * k);
  }
};

// ---- one vector-wave lane's 24 consecutive channels of one row
struct Row24 {
  f32x4 v[6];
  DHW_DEV void load_f32(const char* base) {   // 96 bytes of fp32 (LDS accumulator tile or global table)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if constexpr (ENC16_ABL & 2) { const float c = (float)(size_t)base; v[k] = (f32x4){c, c + k, c, c - k}; }
      else v[k] = *reinterpret_cast<const f32x4*>(base + 16 * k);
    }
  }
  DHW_DEV void add_f32(const float* p) {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += *reinterpret_cast<const f32x4*>(p + 4 * k);
  }
  DHW_DEV void add(const Vec24& a) {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += a.v[k];
  }
  DHW_DEV void film(const Vec24& gam, const Vec24& bet) {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = v[k] * gam.v[k] + bet.v[k];
  }
  DHW_DEV void add(const Half24& h) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint4 w = h.w[k];
      v[2 * k] += (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
      v[2 * k + 1] += (f32x4){__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
    }
  }
  DHW_DEV void add_bf16(const bf16_t* p) {   // 48 bytes of bf16
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint4 w = *reinterpret_cast<const uint4*>(p + 8 * k);
      v[2 * k] += (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
      v[2 * k + 1] += (f32x4){__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
    }
  }
  DHW_DEV void store_bf16(bf16_t* p) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint2 a = pack4_bf16(v[2 * k]), b = pack4_bf16(v[2 * k + 1]);
      if constexpr (ENC16_ABL & 4) asm volatile("" ::"v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
      else *reinterpret_cast<uint4*>(p + 8 * k) = make_uint4(a.x, a.y, b.x, b.y);
    }
  }
  // LayerNorm over the row's 384 channels = this lane's 24 values x the 16 lanes of its row group (eps 1e-6, model.py:25).
  // Sum and sum of squares in one pass (as layernorm_rows_1pass), reduced with DPP lane permutations — pure VALU: an LDS
  // round trip per step (ds_bpermute) queues behind the weight stream's returns like any other LDS read.
  DHW_DEV void layernorm() {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      q += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
    }
    s = row16_sum(s);
    q = row16_sum(q);
    const float mean = s * (1.0f / DM);
    const float var = fmaxf(q * (1.0f / DM) - mean * mean, 0.f);
    const float rstd = rsqrtf(var + 1e-6f);
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = (v[k] - mean) * rstd;
  }
  DHW_DEV void film(const float* gam, const float* bet) {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = v[k] * *reinterpret_cast<const f32x4*>(gam + 4 * k) + *reinterpret_cast<const f32x4*>(bet + 4 * k);
  }
  DHW_DEV void silu() { silu_tiles<bf16_t, 6>(v); }
};

// GEMM waves: one stage = 12 k-chunks x 3 channel tiles over the operand tile `tile`; accumulators (optionally added to) -> ACC
DHW_DEV void store_acc(char* ACC, const f32x4 (&acc)[NTG][1], int gw, int lane) {
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < NTG; ++i) *reinterpret_cast<f32x4*>(ACC + l15 * ACCS + (48 * gw + 16 * i + 4 * g) * 4) = acc[i][0];
}

typedef WRing<bf16_t, NTG, KC * NTG, KC> Ring;   // a whole stage: 36 fragments
typedef bf16_t T;
// stage stamps (diagnostic builds, NEXT = 0 only): kept in LDS and written out by the last wave at the end — a global store from
// inside the pipeline would itself queue behind the weight stream (and the next s_waitcnt vmcnt behind it)
#ifdef DHW_STAMPS
#define LDS16_STAMP(cond, slot) do { if (!NEXT && p.stamps && blockIdx.x == 0 && (cond) && lane == 0) reinterpret_cast<volatile unsigned long long*>(smem + O_STAMPS)[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LDS16_STAMP(cond, slot) do { } while (0)
#endif
#define G16_STAMP(slot) LDS16_STAMP(gw == 0, slot)
#define V16_STAMP(slot) LDS16_STAMP(vw == 0, slot)

// One block of KB keys for the 256 vector-wave threads: K rows [kb, kb + KB) x 384 channels and V^T rows [0, 384) x keys
// [kb, kb + KB), requested at clamped (valid) addresses in one round trip, then stored (K rows at or past Lk and V^T pieces
// past lpad as zeros).  Thread = (K row, 16-byte column phase) and (V^T channel phase, 8-key piece): every address is one
// base plus compile-time offsets, one validity predicate per thread.
template <int KB>
struct StageKV {
  static constexpr int TPR = NV / KB, UK = DM / 8 / TPR;          // threads per K row (4 / 8), pieces per thread (12 / 6)
  static constexpr int PPR = KB / 8, CH0 = NV / PPR, UV = DM / CH0;   // V^T pieces per row (8 / 4), channels per pass, passes
  uint4 k[UK], v[UV];
  DHW_DEV void load(const T* ksrc, int ldk, int Lk, const T* vsrc, int lpad, int kb, int tv) {
    const int r = tv / TPR, q = tv % TPR;
    const T* kp = ksrc + (size_t)min(kb + r, Lk - 1) * ldk + q * 8;
#pragma unroll
    for (int u = 0; u < UK; ++u) k[u] = *reinterpret_cast<const uint4*>(kp + u * TPR * 8);
    const int ch0 = tv / PPR, part = tv % PPR;
    const T* vp = vsrc + (size_t)ch0 * lpad + (kb + (part + 1) * 8 <= lpad ? kb + part * 8 : 0);
    const size_t step = (size_t)CH0 * lpad;
#pragma unroll
    for (int u = 0; u < UV; ++u) v[u] = *reinterpret_cast<const uint4*>(vp + u * step);
  }
  DHW_DEV void store(char* KT, int SK, char* VT, int SV, int Lk, int lpad, int kb, int tv) const {
    const int r = tv / TPR, q = tv % TPR;
    const bool kok = kb + r < Lk;
    char* kd = KT + r * SK + q * 16;
#pragma unroll
    for (int u = 0; u < UK; ++u) *reinterpret_cast<uint4*>(kd + u * TPR * 16) = make_uint4(kok ? k[u].x : 0u, kok ? k[u].y : 0u, kok ? k[u].z : 0u, kok ? k[u].w : 0u);
    const int ch0 = tv / PPR, part = tv % PPR;
    const bool vok = kb + (part + 1) * 8 <= lpad;
#pragma unroll
    for (int u = 0; u < UV; ++u) vt_store_piece<T>(VT + (ch0 + u * CH0) * SV, part, make_uint4(vok ? v[u].x : 0u, vok ? v[u].y : 0u, vok ? v[u].z : 0u, vok ? v[u].w : 0u));
  }
};

// The two roles are two separate straight-line programs (one `if` at the top of the kernel on a wave-uniform value): written
// as branches of one program, the GEMM waves' 144 ring registers would be live through the vector waves' code as well.  Their
// barrier sequences must match one to one — the labels [B..] below pair them up.
//
// ---- GEMM waves
template <int NEXT>
DHW_DEV void gemm_waves(const EncLayerParams& p, const EncChain& nx, char* smem, int gw, int lane) {
  const int l15 = lane & 15, g = lane >> 4;
  char* TA = smem + O_TA;
  char* TB = smem + O_TB;
  char* TC = smem + O_TC;
  char* TD = smem + O_TD;
  char* ACC0 = smem + O_ACC0;
  char* ACC1 = smem + O_ACC1;
  const size_t wlane = ((size_t)(3 * gw) * KC * 64 + lane) * 8;   // offset into a packed [384 x 384] block
  const int aoff = l15 * S + g * 16;
  Ring ring;
  f32x4 acc[NTG][1];
  G16_STAMP(16);
  lds_barrier();                      // [B.l0] the vector waves' staging loads go first
  ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d2) + wlane);
  G16_STAMP(17);
  for (int kb = 0; kb < p.Lk; kb += KBS) {
    if (kb) lds_barrier();            // [B.s0]
    lds_barrier();                    // [B.s1]
  }
  lds_barrier();                      // [B.a2]
  G16_STAMP(18);
  acc_zero(acc);
  ring.template run_s<1, KC>(acc, TA + aoff, S, KC);
  store_acc(ACC0, acc, gw, lane);
  G16_STAMP(19);
  lds_barrier();                      // [B.d2]
  if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f1) + wlane);                       // ffn1 half 0
  G16_STAMP(20);
  lds_barrier();                      // [B.x3]
  G16_STAMP(21);
  acc_zero(acc);
  ring.template run_s<1, KC>(acc, TA + aoff, S, KC);
  store_acc(ACC0, acc, gw, lane);
  lds_barrier();                      // [B.f1a]
  G16_STAMP(22);
  if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f1) + (size_t)DM * DM + wlane);     // ffn1 half 1
  lds_barrier();                      // [B.tc]
  acc_zero(acc);
  ring.template run_s<1, KC>(acc, TA + aoff, S, KC);
  store_acc(ACC1, acc, gw, lane);
  lds_barrier();                      // [B.f1b]
  if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f2) + (((size_t)(3 * gw) * 2 * KC) * 64 + lane) * 8, 2 * KC);   // K-slice 0 of W2 [384][768]
  lds_barrier();                      // [B.td]
  G16_STAMP(23);
  acc_zero(acc);
  ring.template run_s<1, KC>(acc, TC + aoff, S, KC);
  // (the second slice streams here with nothing to hide behind: a stage's fragments fill the register file)
  if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_f2) + (((size_t)(3 * gw) * 2 * KC + KC) * 64 + lane) * 8, 2 * KC);
  ring.template run_s<1, KC>(acc, TD + aoff, S, KC);
  store_acc(ACC0, acc, gw, lane);
  lds_barrier();                      // [B.f2]
  G16_STAMP(24);
  if constexpr (NEXT) {
    const EncLayerParams& a = nx.a;
    if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(a.w_q1) + wlane);
    lds_barrier();                    // [B.x]
    acc_zero(acc);
    ring.template run_s<1, KC>(acc, TC + aoff, S, KC);
    store_acc(ACC0, acc, gw, lane);
    lds_barrier();                    // [B.q1]
    if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(a.w_d1) + wlane);
    for (int kb = KBC; kb < a.Lt; kb += KBC) {
      lds_barrier();                  // [B.t0]
      lds_barrier();                  // [B.t1]
    }
    lds_barrier();                    // [B.a1]
    acc_zero(acc);
    ring.template run_s<1, KC>(acc, TA + aoff, S, KC);
    store_acc(ACC1, acc, gw, lane);
    lds_barrier();                    // [B.d1]
    if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(a.w_qkv2) + wlane);                   // q2 chunk
    lds_barrier();                    // [B.x2]
#pragma unroll 1
    for (int chunk = 0; chunk < 3; ++chunk) {
      acc_zero(acc);
      ring.template run_s<1, KC>(acc, TB + aoff, S, KC);
      store_acc((chunk & 1) ? ACC1 : ACC0, acc, gw, lane);
      lds_barrier();                  // [B.c]
      if (chunk < 2) if (!(ENC16_ABL & 1)) ring.template fill_s<KC>(reinterpret_cast<const T*>(a.w_qkv2) + (size_t)(chunk + 1) * DM * DM + wlane);
    }
  }
}

// ---- vector waves.  Every global load of a stage (parameters, residual rows, staging copies) is requested BEFORE the barrier
// behind which the GEMM waves start their next weight request: the CU's vector-memory path serves requests in order, and a
// load queued behind 288 KB of weights waits 2-3 us (first version of this kernel: 3-4 us per epilogue).
template <int NEXT>
DHW_DEV void vector_waves(const EncLayerParams& p, const EncChain& nx, char* smem, int b, int m0, int rows_valid, int vw, int lane) {
  const int l15 = lane & 15, g = lane >> 4, tv = vw * 64 + lane;
  char* TA = smem + O_TA;
  char* TB = smem + O_TB;
  char* TC = smem + O_TC;
  char* TD = smem + O_TD;
  char* ACC0 = smem + O_ACC0;
  char* ACC1 = smem + O_ACC1;
  // this lane in the row-wise stages: row rv of the tile, channels cb .. cb + 23
  const int rv = 4 * vw + g, cb = l15 * 24;
  const int grow = min(m0 + rv, p.Lk - 1);                 // (clamped: rows past the sample are computed, never stored)
  const bool rvalid = rv < rows_valid;
  const float* gam = p.film + (size_t)b * p.film_bs;
  const float* bet = gam + p.film_tot;
  constexpr int HS = 4, UMAX = 2;                          // attention units of wave vw: heads vw and vw + 4 (the latter for vw < 2)
  const EncLayerParams& a = nx.a;

  // ================= enc_bc =================
  {  // ---- self-attention over all Lk rows of the sample, 64 keys per staged block -> a2 in TA
    char* KT = smem + O_KVS;
    char* VT = KT + KBS * SKS;
    const T* qk = reinterpret_cast<const T*>(p.qk2);
    const T* ksrc = qk + (size_t)b * p.Lk * 2 * DM + DM;
    const T* vsrc = reinterpret_cast<const T*>(p.vt2) + (size_t)b * DM * p.lpadX;
    Frag<T> qf[UMAX][2];
    float mr[UMAX], lr[UMAX];
    f32x4 o[UMAX][4];
    {
      StageKV<KBS> st;
      st.load(ksrc, 2 * DM, p.Lk, vsrc, p.lpadX, 0, tv);
#pragma unroll
      for (int u = 0; u < UMAX; ++u) {
        const int h = vw + u * HS;
        const T* qrow = qk + (size_t)(b * p.Lk + min(m0 + l15, p.Lk - 1)) * 2 * DM + (h < H ? h : 0) * 64 + 8 * g;
        qf[u][0] = frag_load(qrow);
        qf[u][1] = frag_load(qrow + 32);
      }
      lds_barrier();                  // [B.l0] requested: the GEMM waves may start theirs
      st.store(KT, SKS, VT, SVS, p.Lk, p.lpadX, 0, tv);
    }
    V16_STAMP(26);
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      mr[u] = -INFINITY;
      lr[u] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[u][t] = (f32x4){0, 0, 0, 0};
    }
    for (int kb = 0; kb < p.Lk; kb += KBS) {
      if (kb) {
        lds_barrier();                // [B.s0] every vector wave is past its reads of the previous block
        StageKV<KBS> st;
        st.load(ksrc, 2 * DM, p.Lk, vsrc, p.lpadX, kb, tv);
        st.store(KT, SKS, VT, SVS, p.Lk, p.lpadX, kb, tv);
      }
      lds_barrier();                  // [B.s1] block staged
      if (kb == 0) V16_STAMP(27);
      attn_units<T, KBS, false, UMAX>(qf, KT + l15 * SKS, SKS, VT + l15 * SVS, SVS, vw, HS, H, kb, 0u, p.Lk, mr, lr, o);
    }
    // a2 -> TA (outside the staging area, which becomes TB .. ACC1 again behind the barrier)
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = vw + u * HS;
      if (h < H) {
        const float inv = 1.0f / xg_sum(lr[u]);
        T* dst = reinterpret_cast<T*>(TA + l15 * S) + h * 64 + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t) store4(dst + 16 * t, o[u][t] * inv);
      }
    }
  }
  V16_STAMP(28);
  {  // ---- x3 = FiLM2(LN(x2 + Wd a2 + b))
    Vec24 bias, fg, fb;
    Half24 res;
    bias.load(p.b_d2 + cb);
    fg.load(gam + p.f2 + cb);
    fb.load(bet + p.f2 + cb);
    res.load(reinterpret_cast<const T*>(p.x2) + (size_t)(b * p.Lk + grow) * DM + cb);
    lds_barrier();                    // [B.a2]
    lds_barrier();                    // [B.d2]
    V16_STAMP(29);
    Row24 r;
    r.load_f32(ACC0 + rv * ACCS + cb * 4);
    r.add(bias);
    r.add(res);
    r.layernorm();
    r.film(fg, fb);
    r.store_bf16(reinterpret_cast<T*>(TB + rv * S) + cb);                                  // x3
    r.silu();
    r.store_bf16(reinterpret_cast<T*>(TA + rv * S) + cb);                                  // SiLU(x3): operand of both ffn1 halves
  }
  V16_STAMP(30);
  {  // ---- hidden half 0 = SiLU(W1[0:384] SiLU(x3) + b1[0:384]) -> TC, half 1 -> TD
    Vec24 b0, b1;
    b0.load(p.b_f1 + cb);
    b1.load(p.b_f1 + DM + cb);
    lds_barrier();                    // [B.x3]
    lds_barrier();                    // [B.f1a]
    Row24 r;
    r.load_f32(ACC0 + rv * ACCS + cb * 4);
    r.add(b0);
    r.silu();
    r.store_bf16(reinterpret_cast<T*>(TC + rv * S) + cb);
    lds_barrier();                    // [B.tc]
    lds_barrier();                    // [B.f1b]
    r.load_f32(ACC1 + rv * ACCS + cb * 4);
    r.add(b1);
    r.silu();
    r.store_bf16(reinterpret_cast<T*>(TD + rv * S) + cb);
  }
  {  // ---- out = FiLM3(LN(W2 hidden + b2 + x3)); the chained layer's first text block is requested with the parameters and
     // stored behind the epilogue
    Vec24 bias, fg, fb;
    bias.load(p.b_f2 + cb);
    fg.load(gam + p.f3 + cb);
    fb.load(bet + p.f3 + cb);
    StageKV<KBC> st;
    if constexpr (NEXT)
      st.load(reinterpret_cast<const T*>(a.k1) + (size_t)b * a.Lt * DM, DM, a.Lt, reinterpret_cast<const T*>(a.vt1) + (size_t)b * DM * a.lpadT, a.lpadT, 0, tv);
    lds_barrier();                    // [B.td]
    lds_barrier();                    // [B.f2]
    Row24 r;
    r.load_f32(ACC0 + rv * ACCS + cb * 4);
    r.add(bias);
    r.add_bf16(reinterpret_cast<const T*>(TB + rv * S) + cb);                             // + x3
    r.layernorm();
    r.film(fg, fb);
    if constexpr (NEXT) {
      r.store_bf16(reinterpret_cast<T*>(TC + rv * S) + cb);                               // x tile of the chained layer
      char* KT = smem + O_KVC;
      st.store(KT, SKC, KT + KBC * SKC, SVC, a.Lt, a.lpadT, 0, tv);
    }
    if (rvalid) r.store_bf16(reinterpret_cast<T*>(p.out) + (size_t)(b * p.Lk + m0 + rv) * DM + cb);
  }
  V16_STAMP(31);
#ifdef DHW_STAMPS
  if (!NEXT && p.stamps && blockIdx.x == 0 && vw == 3) {   // (the GEMM waves' last stamp is in front of [B.f2])
    __builtin_amdgcn_s_sleep(64);
    if (lane >= 16 && lane < 32) p.stamps[lane] = reinterpret_cast<volatile unsigned long long*>(smem + O_STAMPS)[lane];
  }
#endif

  // ================= enc_a of the next layer (x tile in TC) =================
  if constexpr (NEXT) {
    const float* gam_a = a.film + (size_t)b * a.film_bs;
    const float* bet_a = gam_a + a.film_tot;
    {  // ---- cross-attention over the Lt text keys; the query fragments q1 = Wq x + b + PE Wq come straight from ACC0
      char* KT = smem + O_KVC;
      char* VT = KT + KBC * SKC;
      const T* k1s = reinterpret_cast<const T*>(a.k1) + (size_t)b * a.Lt * DM;
      const T* v1s = reinterpret_cast<const T*>(a.vt1) + (size_t)b * DM * a.lpadT;
      const int64_t* trow = a.text ? a.text + (size_t)b * a.Lt : nullptr;
      PadMask<KBC> pad;
      pad.load(trow, 0, a.Lt);
      f32x4 qadd[UMAX][2][2];         // bias + PE Wq of this lane's query elements
#pragma unroll
      for (int u = 0; u < UMAX; ++u) {
        const int h = vw + u * HS, hh = h < H ? h : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ch = hh * 64 + 32 * c + 8 * g;
          const float* pb = a.pb_q1 + (size_t)(m0 + l15) * DM + ch;         // (rows past Lk: the table has slack rows)
          qadd[u][c][0] = *reinterpret_cast<const f32x4*>(a.b_q1 + ch) + *reinterpret_cast<const f32x4*>(pb);
          qadd[u][c][1] = *reinterpret_cast<const f32x4*>(a.b_q1 + ch + 4) + *reinterpret_cast<const f32x4*>(pb + 4);
        }
      }
      lds_barrier();                  // [B.x]
      lds_barrier();                  // [B.q1]
      Frag<T> qf[UMAX][2];
      float mr[UMAX], lr[UMAX];
      f32x4 o[UMAX][4];
#pragma unroll
      for (int u = 0; u < UMAX; ++u) {
        const int h = vw + u * HS, hh = h < H ? h : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const char* ap = ACC0 + l15 * ACCS + (hh * 64 + 32 * c + 8 * g) * 4;
          frag_from_f32(qf[u][c], *reinterpret_cast<const f32x4*>(ap) + qadd[u][c][0], *reinterpret_cast<const f32x4*>(ap + 16) + qadd[u][c][1]);
        }
        mr[u] = -INFINITY;
        lr[u] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[u][t] = (f32x4){0, 0, 0, 0};
      }
      for (int kb = 0; kb < a.Lt; kb += KBC) {
        if (kb) {
          lds_barrier();              // [B.t0]
          {
            StageKV<KBC> st;
            st.load(k1s, DM, a.Lt, v1s, a.lpadT, kb, tv);
            st.store(KT, SKC, VT, SVC, a.Lt, a.lpadT, kb, tv);
          }
          pad.load(trow, kb, a.Lt);
          lds_barrier();              // [B.t1]
        }
        attn_units<T, KBC, true, UMAX>(qf, KT + l15 * SKC, SKC, VT + l15 * SVC, SVC, vw, HS, H, kb, pad.bits(), a.Lt, mr, lr, o);
      }
#pragma unroll
      for (int u = 0; u < UMAX; ++u) {
        const int h = vw + u * HS;
        if (h < H) {
          const float inv = 1.0f / xg_sum(lr[u]);
          T* dst = reinterpret_cast<T*>(TA + l15 * S) + h * 64 + 4 * g;
#pragma unroll
          for (int t = 0; t < 4; ++t) store4(dst + 16 * t, o[u][t] * inv);
        }
      }
    }
    {  // ---- x2 = FiLM1(LN(Wd a1 + b)) + x
      Vec24 bias, fg, fb;
      bias.load(a.b_d1 + cb);
      fg.load(gam_a + a.f1 + cb);
      fb.load(bet_a + a.f1 + cb);
      lds_barrier();                  // [B.a1]
      lds_barrier();                  // [B.d1]
      Row24 r;
      r.load_f32(ACC1 + rv * ACCS + cb * 4);
      r.add(bias);
      r.layernorm();
      r.film(fg, fb);
      r.add_bf16(reinterpret_cast<const T*>(TC + rv * S) + cb);                            // + x
      r.store_bf16(reinterpret_cast<T*>(TB + rv * S) + cb);                                // operand of the q / k / v chunks
      if (rvalid) r.store_bf16(reinterpret_cast<T*>(a.x2) + (size_t)(b * a.Lk + m0 + rv) * DM + cb);
    }
    // ---- [q2 | k2 | v2] = W x2 + b (+ PE W for q, k): chunk c is written out while the GEMM waves stream chunk c + 1
    {
      Vec24 add[2];                   // bias + PE W of the q chunk, of the k chunk
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        Vec24 pb;
        add[c].load(a.b_qkv2 + c * DM + cb);
        pb.load(a.pb_qk2 + (size_t)(m0 + rv) * 2 * DM + c * DM + cb);
#pragma unroll
        for (int k = 0; k < 6; ++k) add[c].v[k] += pb.v[k];
      }
      // v2 -> vt2 [b][channel][lpadX], key-contiguous: thread = 3 x (channel, 8 keys); zero past the valid rows; the sample's
      // last tile also zero-fills the padding up to lpadX
      float bv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) bv[i] = a.b_qkv2[2 * DM + ((tv + i * NV) >> 1)];
      lds_barrier();                  // [B.x2]
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        lds_barrier();                // [B.c]
        Row24 r;
        r.load_f32((c ? ACC1 : ACC0) + rv * ACCS + cb * 4);
        r.add(add[c]);
        if (rvalid) r.store_bf16(reinterpret_cast<T*>(a.qk2) + (size_t)(b * a.Lk + m0 + rv) * 2 * DM + c * DM + cb);
      }
      lds_barrier();                  // [B.c]
      const int klimit = m0 + rows_valid >= a.Lk ? min(BM, a.lpadX - m0) : rows_valid;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int id = tv + i * NV, ch = id >> 1, part = id & 1;
        if ((part + 1) * 8 <= klimit) {
          f32x4 lo, hi;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r0 = part * 8 + k, r1 = r0 + 4;
            lo[k] = r0 < rows_valid ? *reinterpret_cast<const float*>(ACC0 + r0 * ACCS + ch * 4) + bv[i] : 0.f;
            hi[k] = r1 < rows_valid ? *reinterpret_cast<const float*>(ACC0 + r1 * ACCS + ch * 4) + bv[i] : 0.f;
          }
          const uint2 pa = pack4_bf16(lo), pb = pack4_bf16(hi);
          *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.vt2) + ((size_t)b * DM + ch) * a.lpadX + m0 + part * 8) = make_uint4(pa.x, pa.y, pb.x, pb.y);
        }
      }
    }
  }
}

// NEXT = 1: continue with the next layer's enc_a (nx.a) on the out tile.
template <int NEXT>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void enc16_kernel(const EncLayerParams p, const EncChain nx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the role test below is a real branch)
  const int tiles = (p.Lk + BM - 1) / BM;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = bid / tiles, m0 = (bid % tiles) * BM;
#ifndef ENC16_NO_G
  if (wave < 8) gemm_waves<NEXT>(p, nx, smem, wave, lane);
#endif
#ifndef ENC16_NO_V
  if (wave >= 8) __builtin_amdgcn_s_setprio(3);
  if (wave >= 8) vector_waves<NEXT>(p, nx, smem, b, m0, min(BM, p.Lk - m0), wave - 8, lane);
#endif
}

}  // namespace enc16
