// textside.hip — the sigma-dependent text side of the denoiser as fused kernels, one workgroup per (sampler step, prompt)
// pair, every intermediate in LDS:
//
//   text_style_kernel : TextStyleEncoder.forward behind its sigma-independent prefix (reference text_style.py:94-104):
//                       s = FiLM1(LN(style_ffn(style)));  t = FiLM2(LN(emb(text)));  m = MHA8(q = t, k = s, v = s);
//                       t = FiLM3(LN(t + m));  out = FiLM4(LN(text_ffn(t)))
//   text_layer_kernel : the text half of one EncoderLayer (model.py:38-42):
//                       tl = FiLM0(LN(text_dense(SiLU(text))));  k1 = Wk(tl + PE);  v1 = Wv(tl)
//
// In dhw_sample the schedule is known, so these run once for all T steps (n = T*B pairs, FiLM row per step: the
// "all-steps text plane").  The one-launch-per-GEMM form of the same math (dhw_api.cpp, generic gemm.hip +
// attn.hip) moved ~5 GB per sample call through HBM — the FiLM'd style copy, its K / V^T (412 MB written and read back by
// the attention), q, the attention output, the 768-wide FFN hidden layer, tl — and ran at 40-430 TFLOP/s; here only
// text_out, k1 and vt1 are written.  bf16, Lt <= 32 tokens, <= 80 style rows (LDS: the FiLM'd style tile is overwritten
// in place by its own K projection, t / q / attention output share one tile); other shapes and the fp32 parity mode keep
// the generic path.
#include <algorithm>
#include "enc_a_core.h"

namespace {

// softmax(q K^T / sqrt(D)) V for 16 queries of one head against <= 16*NK keys held in LDS, one shot (no running max):
// the single-block form of attn_block_lds (attn_core.h) for key counts that are not a multiple of 32.
//   kt : LDS address of K tile row (lane&15), this head's first channel;   SK: K row stride in bytes
//   vt : LDS address of V^T tile row (head channel lane&15), key 4*(lane>>4);  SV: V^T row stride in bytes
template <typename T, int D, int NK>
DHW_DEV void attn_once(  // (bf16 only)
    const Frag<T> (&qf)[2], const char* kt, int SK, const char* vt, int SV, int Lk, f32x4 (&o)[D / 16]) {
  constexpr int DT = D / 16, NPF = (NK + 1) / 2;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const float scale = rsqrtf((float)D);
  f32x4 s[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    s[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int d = 32 * c + 8 * g;
      const Frag<T> kf = d < D ? frag_load(reinterpret_cast<const T*>(kt + t * 16 * SK) + d) : frag_zero<T>();
      mma32(s[t], kf, qf[c]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = 16 * t + 4 * g + r < Lk ? s[t][r] * scale : -INFINITY;
      s[t][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(s[t][r] - mx);
      s[t][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  const f32x4 zero = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    o[t] = zero;
#pragma unroll
    for (int pp = 0; pp < NPF; ++pp) {
      Frag<T> pf;
      frag_from_f32(pf, s[2 * pp], 2 * pp + 1 < NK ? s[2 * pp + 1] : zero);
      const T* vp = reinterpret_cast<const T*>(vt + (16 * t) * SV) + 32 * pp;
      // (the odd last key tile has no partner: that half of the V^T fragment is zero and is never read from LDS)
      const bf16x4 va = *reinterpret_cast<const bf16x4*>(vp);
      bf16x4 vb = (bf16x4){(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
      if (2 * pp + 1 < NK) vb = *reinterpret_cast<const bf16x4*>(vp + 16);
      Frag<T> vf;
      vf.v = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
      mma32(o[t], vf, pf);
    }
    o[t] = o[t] * inv;
  }
}

// per-lane FiLM parameters of NT channel tiles (as EpiParams, without a bias)
template <int NT>
struct FilmRow {
  f32x4 gam[NT], bet[NT];
  DHW_DEV void load(const float* g, const float* be, int n0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      gam[i] = *reinterpret_cast<const f32x4*>(g + n0 + 16 * i);
      bet[i] = *reinterpret_cast<const f32x4*>(be + n0 + 16 * i);
    }
  }
};

// Stage rows [0, rows_valid) x 384 channels of `src` into an LDS tile through FiLM (x * gamma + beta, rounded to T as the
// stand-alone film_apply kernel does), zero rows up to `rows_tile`.  Thread t < 480 owns the 16-byte column piece t % 48 of
// rows t / 48 + 10 k, so its 8 gamma / beta values are loaded once.
template <typename T, int KMAX>
DHW_DEV void stage_film_384(char* dst, int S, const T* src, int rows_valid, int rows_tile, const float* gam, const float* bet, int tid) {
  static_assert(sizeof(T) == 2, "bf16 only");
  if (tid >= 480) return;
  const int cc = tid % 48, r0 = tid / 48;
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + cc * 8), g1 = *reinterpret_cast<const f32x4*>(gam + cc * 8 + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bet + cc * 8), b1 = *reinterpret_cast<const f32x4*>(bet + cc * 8 + 4);
  uint4 v[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int r = r0 + 10 * k;
    v[k] = make_uint4(0, 0, 0, 0);
    if (r < rows_valid) v[k] = *reinterpret_cast<const uint4*>(src + (size_t)r * 384 + cc * 8);
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int r = r0 + 10 * k;
    if (r >= rows_tile) continue;
    if (r < rows_valid) {
      T* e = reinterpret_cast<T*>(&v[k]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = from_f<T>(to_f(e[i]) * g0[i] + b0[i]);
        e[4 + i] = from_f<T>(to_f(e[4 + i]) * g1[i] + b1[i]);
      }
    }
    *reinterpret_cast<uint4*>(dst + r * S + cc * 16) = v[k];
  }
}

template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void text_style_kernel(const TextStyleParams p) {
  constexpr int ES = sizeof(T), DM = 384, BM = 32, SM = 80, D = 48, KC = DM / 32;
  constexpr int NT = DM / 8 / 16, MT = BM / 16, MTS = SM / 16;
  constexpr int S = tile_stride<T>(DM), SVT = SM * ES + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* RS = smem;                       // FiLM'd style rows [SM][DM]; later the keys K [SM][DM]; later t2 / SiLU(t2) / FFN hidden tiles
  char* RV = RS + SM * S;                // V^T [DM][SM]
  char* RT = RV + DM * SVT;              // t1, then q, then the attention output [BM][DM]
  float* red = reinterpret_cast<float*>(RT + BM * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int pair = blockIdx.x;
  const int bs = pair % p.in_B;                                     // prompt of this (step, prompt) pair
  const float* gam = p.film + (long)(pair / p.film_div) * p.film_bs;   // this step's FiLM row
  const float* bet = gam + p.film_tot;
  const int ntile0 = wave * NT, n0 = ntile0 * 16 + 4 * g;
  const size_t wlane = ((size_t)ntile0 * KC * 64 + lane) * 8;       // into a packed [384 x 384] block
  const T* sty = reinterpret_cast<const T*>(p.sty_n) + (size_t)bs * p.S5 * DM;
  const T* tn = reinterpret_cast<const T*>(p.t_n) + (size_t)bs * p.Lt * DM;
  const char* sop = RS + l15 * S + g * 8 * ES;
  const char* top = RT + l15 * S + g * 8 * ES;

  WRing<T, NT> ring;
#ifndef DHW_TEXT_SPREAD
#define DHW_TEXT_SPREAD 1
#endif
  constexpr bool TSPREAD = DHW_TEXT_SPREAD != 0;
  constexpr int FC = WRing<T, NT>::template fill_chunks<KC>(), FQ = (FC + 3) / 4;
  EpiParams<NT> ep;

  // ---- s = FiLM1(style rows), t1 = FiLM2(token rows) -> LDS
  stage_film_384<T, 8>(RS, S, sty, p.S5, SM, gam + p.f1, bet + p.f1, tid);
  stage_film_384<T, 4>(RT, S, tn, p.Lt, BM, gam + p.f2, bet + p.f2, tid);
  ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_kv8) + (size_t)DM * DM + wlane);   // V half of the stacked K|V projection
  ep.load_bias(p.b_kv8 + DM, n0);
  lds_barrier();

  {  // ---- V^T = (s Wv + bv)^T, keys contiguous, zero past the style rows
    f32x4 acc[NT][MTS];
    acc_zero(acc);
    ring.template run_s<MTS, KC>(acc, sop, S, KC);
    // (DHW_TEXT_SPREAD: the K projection's first weight fragments in quarters between the tiles of the V^T scatter, not one burst in front of it)
    if constexpr (TSPREAD) ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_kv8) + wlane);
    else ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_kv8) + wlane);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if constexpr (TSPREAD) {
        if (i == 0) ring.template fill_range<KC, 0, FQ>();
        else if (i == 1) ring.template fill_range<KC, FQ, 2 * FQ>();
        else if (i == 2) ring.template fill_range<KC, 2 * FQ, 3 * FQ>();
      }
#pragma unroll
      for (int j = 0; j < MTS; ++j) {
        const int key = j * 16 + l15;
        const f32x4 v = acc[i][j] + ep.bias[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<T*>(RV + (n0 + 16 * i + k) * SVT + key * ES) = from_f<T>(key < p.S5 ? v[k] : 0.f);
      }
    }
    if constexpr (TSPREAD) { static_assert(NT == 3, "three channel tiles per wave"); ring.template fill_range<KC, 3 * FQ, FC>(); }
    ep.load_bias(p.b_kv8, n0);
  }
  {  // ---- K = s Wk + bk, written over s once every wave has finished reading it
    f32x4 acc[NT][MTS];
    acc_zero(acc);
    ring.template run_s<MTS, KC>(acc, sop, S, KC);
    ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_q8) + wlane);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MTS; ++j) store4(reinterpret_cast<T*>(RS + (j * 16 + l15) * S) + n0 + 16 * i, acc[i][j] + ep.bias[i]);
    ep.load_bias(p.b_q8, n0);
  }
  {  // ---- q = t1 Wq + bq, in place
    f32x4 acc[NT][MT];
    acc_zero(acc);
    ring.template run_s<MT, KC>(acc, top, S, KC);
    ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_d8) + wlane);
    lds_barrier();   // every wave is past its t1 reads (and the K / V^T tiles are complete)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) store4(reinterpret_cast<T*>(RT + (j * 16 + l15) * S) + n0 + 16 * i, acc[i][j] + ep.bias[i]);
  }
  // FiLM rows and residual operands of the next stage: requested here, used after the attention
  FilmRow<NT> f2r, f3r;
  f2r.load(gam + p.f2, bet + p.f2, n0);
  f3r.load(gam + p.f3, bet + p.f3, n0);
  ep.load_bias(p.b_d8, n0);
  lds_barrier();

  // ---- m = MHA8(q, K, V): wave = one head, both 16-row query groups; the output replaces q in place (a wave reads and
  // writes only its own head's columns)
#pragma unroll
  for (int rg = 0; rg < MT; ++rg) {
    Frag<T> qf[2];
    const T* qrow = reinterpret_cast<const T*>(RT + (rg * 16 + l15) * S) + wave * D + 8 * g;
    qf[0] = frag_load(qrow);
    qf[1] = g < 2 ? frag_load(qrow + 32) : frag_zero<T>();
    f32x4 o[D / 16];
    attn_once<T, D, MTS>(qf, RS + l15 * S + wave * D * ES, S, RV + (wave * D + l15) * SVT + 4 * g * ES, SVT, p.S5, o);
    T* dst = reinterpret_cast<T*>(RT + (rg * 16 + l15) * S) + wave * D + 4 * g;
#pragma unroll
    for (int t = 0; t < D / 16; ++t) store4(dst + 16 * t, o[t]);
  }
  lds_barrier();

  char* T2 = RS;                  // t2            [BM][DM]   (K / V^T are dead)
  char* ST2 = T2 + BM * S;        // SiLU(t2)
  char* HID = ST2 + BM * S;       // one 384-wide half of the FFN hidden layer
  {  // ---- t2 = FiLM3(LN(t1 + m Wd + bd))
    f32x4 acc[NT][MT];
    acc_zero(acc);
    ring.template run_s<MT, KC>(acc, top, S, KC);
    if constexpr (TSPREAD) { ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_tf1) + wlane); ring.template fill_range<KC, 0, FQ>(); }
    else ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_tf1) + wlane);   // FFN half 0: flies during the LayerNorm
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int r = j * 16 + l15;
        // t1 again (it was overwritten by q), with the rounding of the staged copy.  Unconditional load (rows past Lt read
        // the next prompt's / the slack rows; those output rows are never written): a per-lane `if (r < Lt) load` becomes
        // a branch with s_waitcnt vmcnt(0) inside and drains the weight prefetch
        f32x4 res;
        {
          const f32x4 x = load4(tn + (size_t)r * DM + n0 + 16 * i);
#pragma unroll
          for (int k = 0; k < 4; ++k) res[k] = r < p.Lt ? to_f(from_f<T>(x[k] * f2r.gam[i][k] + f2r.bet[i][k])) : 0.f;
        }
        acc[i][j] += ep.bias[i] + res;
      }
    if constexpr (TSPREAD) {
      ring.template fill_range<KC, FQ, 2 * FQ>();
      ln_rows<T, MT, NT, 8, BM>(acc, red, wave, 0, lane, DM, true, [&]() { ring.template fill_range<KC, 2 * FQ, 3 * FQ>(); });
      ring.template fill_range<KC, 3 * FQ, FC>();
    } else ln_rows<T, MT, NT, 8, BM>(acc, red, wave, 0, lane, DM);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int r = j * 16 + l15;
        f32x4 v = acc[i][j] * f3r.gam[i] + f3r.bet[i];
        store4(reinterpret_cast<T*>(T2 + r * S) + n0 + 16 * i, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu_t<T>(to_f(from_f<T>(v[k])));   // SiLU of the ROUNDED t2, as a staged operand would be
        store4(reinterpret_cast<T*>(ST2 + r * S) + n0 + 16 * i, v);
      }
  }
  lds_barrier();

  // ---- out = FiLM4(LN(W2 SiLU(W1 SiLU(t2) + b1) + b2)), the 768-wide hidden layer in two halves (no residual: text_style.py:103)
  f32x4 acc2[NT][MT];
  acc_zero(acc2);
  const char* s2op = ST2 + l15 * S + g * 8 * ES;
  const char* hop = HID + l15 * S + g * 8 * ES;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    f32x4 acc[NT][MT];
    acc_zero(acc);
    ep.load_bias(p.b_tf1 + hh * DM, n0);
    ring.template run_s<MT, KC>(acc, s2op, S, KC);
    if constexpr (TSPREAD) ring.template fill_begin<KC>(reinterpret_cast<const T*>(p.w_tf3) + (((size_t)ntile0 * 2 * KC + hh * KC) * 64 + lane) * 8, 2 * KC);
    else ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_tf3) + (((size_t)ntile0 * 2 * KC + hh * KC) * 64 + lane) * 8, 2 * KC);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if constexpr (TSPREAD) {
        if (i == 0) ring.template fill_range<KC, 0, FQ>();
        else if (i == 1) ring.template fill_range<KC, FQ, 2 * FQ>();
        else if (i == 2) ring.template fill_range<KC, 2 * FQ, 3 * FQ>();
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        f32x4 v = acc[i][j] + ep.bias[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu_t<T>(v[k]);
        store4(reinterpret_cast<T*>(HID + (j * 16 + l15) * S) + n0 + 16 * i, v);
      }
    }
    if constexpr (TSPREAD) ring.template fill_range<KC, 3 * FQ, FC>();
    lds_barrier();
    ring.template run_s<MT, KC>(acc2, hop, S, KC);
    if (hh == 0) {
      ring.template fill_s<KC>(reinterpret_cast<const T*>(p.w_tf1) + (size_t)DM * DM + wlane);
      lds_barrier();   // HID is rewritten by the next half
    }
  }
  ep.load(p.b_tf3, gam + p.f4, bet + p.f4, n0);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc2[i][j] += ep.bias[i];
  ln_rows<T, MT, NT, 8, BM>(acc2, red, wave, 0, lane, DM);   // (its barriers also fence the HID reads above)
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) store4(reinterpret_cast<T*>(T2 + (j * 16 + l15) * S) + n0 + 16 * i, acc2[i][j] * ep.gam[i] + ep.bet[i]);
  lds_barrier();
  tile_copy_out<T>(T2, S, reinterpret_cast<T*>(p.text_out) + (size_t)pair * p.Lt * DM, DM, p.Lt, DM, tid, 512);
}

// ---- the text half of one EncoderLayer for PAIRS (step, prompt) pairs
// PAIRS = 1: two workgroups per CU (VGPR budget 128 per lane, 58 KB LDS): there are T*B >> 256 independent pairs, so co-resident
// workgroups in different stages hide each other's stage boundaries.  PAIRS = 2 (round 4): ONE workgroup per CU holds two
// consecutive pairs that share a FiLM row (the same step: film_div even) as rows [0, 32) and [32, 64) of one 64-row tile, so the
// layer's three weight matrices (0.3-0.9 MB) are streamed through the CU's L1 once per two pairs instead of once per pair — the
// stream was half of the kernel — with a full-depth ring; per row the arithmetic is the same sequence of MFMAs, so
// the outputs are bit-identical to PAIRS = 1.  Measured: no difference (what the shared stream saves, the lost co-residency costs):
// DHW_TEXT_PAIRS=2 only.
template <typename T, int DMO, int PAIRS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4 / PAIRS, 4 / PAIRS))) void text_layer_kernel(const TextLayerParams p) {
  constexpr int OCC = 2 / PAIRS;
  constexpr int ES = sizeof(T), DI = 384, BM = 32 * PAIRS, KCI = DI / 32, KCO = DMO / 32;
  constexpr int WN = (DMO % 128 == 0) ? 8 : 6, NT = DMO / WN / 16, MT = BM / 16;
  constexpr int SI = tile_stride<T>(DI), SO = tile_stride<T>(DMO);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* XS = smem;                       // SiLU(text_out) [BM][384]; later the k1 / v1 staging tile
  char* TL = XS + BM * SI;               // tl [BM][DMO]
  float* red = reinterpret_cast<float*>(TL + BM * SO);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
#ifndef DHW_TEXT_SPREAD
#define DHW_TEXT_SPREAD 1
#endif
  constexpr bool TSPREAD = DHW_TEXT_SPREAD != 0;
  const int pair = blockIdx.x * PAIRS;
  const int npair = min(PAIRS, p.n - pair);          // (the last workgroup of an odd n holds one pair: its second half stays zero)
  const float* gam = p.film + (long)(pair / p.film_div) * p.film_bs;
  const float* bet = gam + p.film_tot;
  const bool act = WN == 8 || wave < WN;
  const int wn = act ? wave : 0, ntile0 = wn * NT, n0 = ntile0 * 16 + 4 * g;

  WRing<T, NT, 24 / OCC> ring;
  constexpr int FCO = WRing<T, NT, 24 / OCC>::template fill_chunks<KCO>(), FQO = (FCO + 3) / 4;   // ring slots a stage's first request fills, a quarter of them
  EpiParams<NT> ep;
  {  // text_out rows -> LDS through SiLU (text_dense's input activation, nn.py:165-175)
    const T* src = reinterpret_cast<const T*>(p.text_out) + (size_t)pair * p.Lt * DI;
    constexpr int cpr = DI * ES / 16, NU = BM * cpr / 512;
    static_assert(BM * cpr % 512 == 0, "staging chunks per thread");
    // every piece is requested at a clamped, always valid address and zeroed by a select: written `if (row exists) v = *p` each load was a branch
    // with s_waitcnt vmcnt(0) at its join — SIX dependent memory round trips at the top of every workgroup (round 5, .s: vmcnt(0) after each of 6 loads)
    uint4 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int id = tid + u * 512, r = id / cpr, cc = id - r * cpr, pr = r >> 5, rr = r & 31;
      const bool ok = pr < npair && rr < p.Lt;
      v[u] = *reinterpret_cast<const uint4*>(src + (ok ? (size_t)pr * p.Lt + rr : (size_t)0) * DI + cc * (16 / ES));
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int id = tid + u * 512, r = id / cpr, cc = id - r * cpr, pr = r >> 5, rr = r & 31;
      const bool ok = pr < npair && rr < p.Lt;
      T* e = reinterpret_cast<T*>(&v[u]);
#pragma unroll
      for (int i = 0; i < 16 / ES; ++i) e[i] = from_f<T>(silu_t<T>(to_f(e[i])));
      *reinterpret_cast<uint4*>(XS + r * SI + cc * 16) = ok ? v[u] : make_uint4(0, 0, 0, 0);
    }
  }
  if (act) {
    ring.template fill_s<KCI>(reinterpret_cast<const T*>(p.w_td) + ((size_t)ntile0 * KCI * 64 + lane) * 8);
    ep.load(p.b_td, gam + p.f0, bet + p.f0, n0);
  }
  lds_barrier();

  {  // ---- tl = FiLM0(LN(W SiLU(text) + b))
    f32x4 acc[NT][MT];
    acc_zero(acc);
    if (act) {
      ring.template run_s<MT, KCI>(acc, XS + l15 * SI + g * 8 * ES, SI, KCI);
      // (DHW_TEXT_SPREAD, round 5: the next stage's first weight fragments in quarters between the pieces of the epilogue, as enc_bc_core.h DHW_ENC_SPREAD)
      if constexpr (TSPREAD) { ring.template fill_begin<KCO>(reinterpret_cast<const T*>(p.w_kv) + ((size_t)ntile0 * KCO * 64 + lane) * 8); ring.template fill_range<KCO, 0, FQO>(); }
      else ring.template fill_s<KCO>(reinterpret_cast<const T*>(p.w_kv) + ((size_t)ntile0 * KCO * 64 + lane) * 8);   // K half
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] += ep.bias[i];
      if constexpr (TSPREAD) ring.template fill_range<KCO, FQO, 2 * FQO>();
    }
    if constexpr (TSPREAD) ln_rows<T, MT, NT, WN, BM>(acc, red, wn, 0, lane, DMO, act, [&]() { if (act) ring.template fill_range<KCO, 2 * FQO, 3 * FQO>(); });
    else ln_rows<T, MT, NT, WN, BM>(acc, red, wn, 0, lane, DMO, act);
    if (act) {
      if constexpr (TSPREAD) ring.template fill_range<KCO, 3 * FQO, FCO>();
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) store4(reinterpret_cast<T*>(TL + (j * 16 + l15) * SO) + n0 + 16 * i, acc[i][j] * ep.gam[i] + ep.bet[i]);
    }
  }
  lds_barrier();

  // (both pairs' rows, each pair's Lt rows to its own [Lt][DMO] block of the output)
  auto copy_out = [&](void* base) {
#pragma unroll
    for (int pr = 0; pr < PAIRS; ++pr)
      if (pr < npair) tile_copy_out<T>(XS + pr * 32 * SO, SO, reinterpret_cast<T*>(base) + (size_t)(pair + pr) * p.Lt * DMO, DMO, p.Lt, DMO, tid, 512);
  };
  const char* lop = TL + l15 * SO + g * 8 * ES;
  {  // ---- k1 = Wk tl + bk + PE·Wk[row] -> coalesced rows
    f32x4 acc[NT][MT];
    acc_zero(acc);
    if (act) {
      ep.load_bias(p.b_kv, n0);
      ring.template run_s<MT, KCO>(acc, lop, SO, KCO);
      if constexpr (TSPREAD) ring.template fill_begin<KCO>(reinterpret_cast<const T*>(p.w_kv) + ((size_t)(DMO / 16 + ntile0) * KCO * 64 + lane) * 8);
      else ring.template fill_s<KCO>(reinterpret_cast<const T*>(p.w_kv) + ((size_t)(DMO / 16 + ntile0) * KCO * 64 + lane) * 8);   // V half
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if constexpr (TSPREAD) { if (i == 0) ring.template fill_range<KCO, 0, 2 * FQO>(); else if (i == NT - 1) ring.template fill_range<KCO, 2 * FQO, FCO>(); }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int r = j * 16 + l15, rr = r & 31;                       // rr: the token's position inside its pair
          f32x4 v = acc[i][j] + ep.bias[i];
          v += *reinterpret_cast<const f32x4*>(p.pb_k1 + (size_t)(rr < p.Lt ? rr : p.Lt - 1) * DMO + n0 + 16 * i);   // (clamped, not branched)
          store4(reinterpret_cast<T*>(XS + r * SO) + n0 + 16 * i, v);   // (the SiLU(text) tile is dead: two barriers ago)
        }
      }
      if constexpr (TSPREAD && NT == 1) ring.template fill_range<KCO, 2 * FQO, FCO>();
    }
    lds_barrier();
    copy_out(p.k1);
  }
  {  // ---- v1 = Wv tl + bv (no PE: model.py:46) -> coalesced rows, like k1 (the cross-attention reads V^T with the transposing LDS
     // read: attn_core.h)
    f32x4 acc[NT][MT];
    acc_zero(acc);
    if (act) {
      ep.load_bias(p.b_kv + DMO, n0);
      ring.template run_s<MT, KCO>(acc, lop, SO, KCO);
    }
    lds_barrier();   // the k1 copy-out has read the staging tile
    if (act) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) store4(reinterpret_cast<T*>(XS + (j * 16 + l15) * SO) + n0 + 16 * i, acc[i][j] + ep.bias[i]);
    }
    lds_barrier();
    copy_out(p.vt1);
  }
}

constexpr size_t text_style_lds() { return (size_t)80 * tile_stride<bf16_t>(384) + (size_t)384 * (80 * 2 + 16) + (size_t)32 * tile_stride<bf16_t>(384) + 2 * 8 * 32 * sizeof(float); }
template <int DMO, int PAIRS>
constexpr size_t text_layer_lds() {
  return (size_t)32 * PAIRS * tile_stride<bf16_t>(384) + (size_t)32 * PAIRS * tile_stride<bf16_t>(DMO) + 2 * 8 * 32 * PAIRS * sizeof(float);
}

}  // namespace

hipError_t textside_init() {
  static_assert(text_style_lds() <= 160 * 1024, "text_style tile does not fit LDS");
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(text_style_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#define DHW_TL_ATTR(D_, P_) if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(text_layer_kernel<bf16_t, D_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e
  DHW_TL_ATTR(192, 1); DHW_TL_ATTR(256, 1); DHW_TL_ATTR(384, 1);
  DHW_TL_ATTR(192, 2); DHW_TL_ATTR(256, 2); DHW_TL_ATTR(384, 2);
#undef DHW_TL_ATTR
  return hipSuccess;
}

bool textside_supported(int prec, int Lt, int S5, int dt) { return prec == PREC_BF16 && Lt >= 1 && Lt <= 32 && S5 >= 1 && S5 <= 80 && dt == 384; }

hipError_t launch_text_style(int prec, const TextStyleParams& p, hipStream_t st) {
  if (!textside_supported(prec, p.Lt, p.S5, 384) || p.n < 1 || p.in_B < 1 || p.film_div < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(text_style_kernel<bf16_t>, dim3(p.n), dim3(512), text_style_lds(), st, p);
  return hipGetLastError();
}

hipError_t launch_text_layer(int prec, const TextLayerParams& p, hipStream_t st) {
  if (prec != PREC_BF16 || p.Lt < 1 || p.Lt > 32 || p.n < 1 || p.film_div < 1 || p.lpadT < 32 || p.lpadT % 8) return hipErrorInvalidValue;
  // two pairs per workgroup where consecutive pairs share their FiLM row (the all-steps text plane with an even batch) (p.pairs: dhw_create's DHW_TEXT_PAIRS = 1 / 2 forces either form; A/B, equivalence test)
  // r4 measured the two forms equal (18.72 vs 18.73 ms per 60-step batch, profiles/r04_text_pairs_ab.log); with the activation fragments requested
  // ahead of their MFMAs (r5, gemm_core.h run_p) the 64-row form is ahead — 18.52 -> 18.44 / 18.48 ms same-box, profiles/r05_switches_ab.log — and is
  // the default where the plane is large enough to fill the CUs with one workgroup each (>= 1024 pairs); DHW_TEXT_PAIRS=1 forces one pair per workgroup
  const bool two = p.film_div % 2 == 0 && (p.pairs == 2 || (p.pairs == 0 && p.n >= 1024));
#define DHW_TL_LAUNCH(D_) \
  if (two) hipLaunchKernelGGL((text_layer_kernel<bf16_t, D_, 2>), dim3((p.n + 1) / 2), dim3(512), (text_layer_lds<D_, 2>()), st, p); \
  else hipLaunchKernelGGL((text_layer_kernel<bf16_t, D_, 1>), dim3(p.n), dim3(512), (text_layer_lds<D_, 1>()), st, p)
  switch (p.d) {
    case 192: DHW_TL_LAUNCH(192); break;
    case 256: DHW_TL_LAUNCH(256); break;
    case 384: DHW_TL_LAUNCH(384); break;
    default: return hipErrorInvalidValue;
  }
#undef DHW_TL_LAUNCH
  return hipGetLastError();
}
