// attn_core.h — one wave, 16 queries of one (sample, head): softmax(QK^T/sqrt(D) + mask·(-1e9)) V
// (reference attention.py:26-46).  Shared by the stand-alone attention kernel (attn.hip) and the fused
// EncoderLayer kernels (enclayer.hip).  No LDS, no barrier:
//
//   S^T = K · Q^T      MFMA A-operand = K rows  (16 keys x D, straight from L2)
//                      MFMA B-operand = Q rows  (16 queries x D, fragments supplied by the caller)
//        -> lane (query = lane&15) holds keys 4g..4g+3 of each 16-key tile: softmax statistics are
//           per lane + two cross-group shuffles.
//   O^T = V^T · P^T    B-operand = P^T taken from the S^T accumulators IN PLACE (k-slot order: tile0 keys
//                      4g..4g+3, tile1 keys 16+4g..); A-operand = V^T read with the SAME slot order
//                      from the key-contiguous Vt buffer the projection GEMM wrote.
//   Online softmax over 32-key blocks, fp32 statistics, exact for any Lk.
#pragma once
#include "dhw_common.h"

// krow : K + (first key row)*ldk + head column offset, already advanced by (lane&15)*ldk
// vrow : Vt + (head row offset + lane&15)*lpad + 4*(lane>>4)
// trow : int64 token ids of this sample (key k masked iff trow[k] == 0) or nullptr
// o[t] : on return O^T tile t (d = 16t + 4g + r, query = lane&15), already divided by the softmax sum
// KB = keys per block (32, 64 or 128).  All K and V^T operands of a block are requested up front, so a block
// costs ONE L2/MALL round trip and its MFMAs / exps are independent; the serial part per block (max/sum
// shuffles, rescale) is amortised over KB keys.  Keys past Lk are masked to -inf (P = 0 exactly); the V^T rows
// are zero/finite-padded by the producer, the K rows past Lk are never used unmasked.
template <typename T, int D, int KB>
DHW_DEV void attn_wave16(const Frag<T> (&qf)[(D + 31) / 32], const T* krow, int ldk, const T* vrow, int lpad,
                         const int64_t* trow, int Lk, f32x4 (&o)[D / 16]) {
  constexpr int DT = D / 16, KCH = (D + 31) / 32, NTILE = KB / 16, NPF = KB / 32;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const float scale = rsqrtf((float)D);
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = (f32x4){0, 0, 0, 0};
  for (int kb = 0; kb < Lk; kb += KB) {
    Frag<T> kf[NTILE][KCH];
    Frag<T> vf[DT][NPF];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        const int d = 32 * c + 8 * g;
        kf[t][c] = d < D ? frag_load(krow + (size_t)(kb + 16 * t) * ldk + d) : frag_zero<T>();
      }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int pp = 0; pp < NPF; ++pp) {
        const T* vp = vrow + (size_t)(16 * t) * lpad + kb + 32 * pp;
        vf[t][pp] = frag_load_halves(vp, vp + 16);
      }
    __builtin_amdgcn_sched_barrier(0);   // keep every operand request of the block ahead of the math
    f32x4 s[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      s[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < KCH; ++c) mma32(s[t], kf[t][c], qf[c]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + 16 * t + 4 * g + r;
        float v = s[t][r] * scale;
        if (key < Lk) {
          if (trow && trow[key] == 0) v += -1e9f;   // attention.py:44
        } else {
          v = -INFINITY;
        }
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = xg_max(mx);
    const float m_new = fmaxf(m_run, mx);       // finite: every block has >= 1 real key
    const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first block
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - m_new);
        s[t][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    Frag<T> pf[NPF];
#pragma unroll
    for (int pp = 0; pp < NPF; ++pp) frag_from_f32(pf[pp], s[2 * pp], s[2 * pp + 1]);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      o[t] = o[t] * alpha;
#pragma unroll
      for (int pp = 0; pp < NPF; ++pp) mma32(o[t], vf[t][pp], pf[pp]);
    }
  }
  float l = l_run;
  l = xg_sum(l);
  const float inv = 1.0f / l;
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = o[t] * inv;
}

// block size by key count: one block whenever the keys fit
template <typename T, int D>
DHW_DEV void attn_wave16_auto(const Frag<T> (&qf)[(D + 31) / 32], const T* krow, int ldk, const T* vrow, int lpad,
                              const int64_t* trow, int Lk, f32x4 (&o)[D / 16]) {
  if (Lk <= 32) attn_wave16<T, D, 32>(qf, krow, ldk, vrow, lpad, trow, Lk, o);
  else if (Lk <= 64) attn_wave16<T, D, 64>(qf, krow, ldk, vrow, lpad, trow, Lk, o);
  else attn_wave16<T, D, 128>(qf, krow, ldk, vrow, lpad, trow, Lk, o);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-staged form for the fused EncoderLayer kernels: the workgroup first copies one KB-key block of K
// ([KB keys][all channels], row stride SK bytes) and of V^T ([all channels][KB keys], row stride SV bytes) into
// LDS with fully coalesced 16-byte loads; every wave then takes its fragments with ds_read.  (Per-wave global
// fragment loads touch 16 rows per instruction and re-fetch the same K/V once per row group: measured 34 us for
// 244 keys x 192 channels inside enc_bc.)  One call = one block; the online-softmax state (m, l, o) of the
// (16 rows x 1 head) unit lives in the caller's registers across blocks.
//   kt : LDS address of K tile row (lane&15), this head's first channel
//   vt : LDS address of V^T tile row (head channel lane&15), key 4*(lane>>4)
//   padbits: bit 4t+r set = this lane's key kb + 16t + 4*(lane>>4) + r is a padded text token (score += -1e9); the caller
//            reads the mask ahead of time (attn_pad_bits) so no global load sits in the softmax
template <typename T, int D, int KB>
DHW_DEV void attn_block_lds(int lane, const Frag<T> (&qf)[(D + 31) / 32], const char* kt, int SK, const char* vt, int SV, int kb,
                            unsigned padbits, int Lk, float& m_run, float& l_run, f32x4 (&o)[D / 16]) {
  constexpr int ES = sizeof(T), DT = D / 16, KCH = (D + 31) / 32, NTILE = KB / 16, NPF = KB / 32;
  const int g = lane >> 4;
  const float scale = rsqrtf((float)D);
  f32x4 s[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    s[t] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const int d = 32 * c + 8 * g;
      const Frag<T> kf = d < D ? frag_load(reinterpret_cast<const T*>(kt + t * 16 * SK) + d) : frag_zero<T>();
      mma32(s[t], kf, qf[c]);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * t + 4 * g + r;
      float v = s[t][r] * scale;
      if (key < Lk) {
        if ((padbits >> (4 * t + r)) & 1u) v += -1e9f;   // attention.py:44
      } else {
        v = -INFINITY;
      }
      s[t][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = xg_max(mx);
  const float m_new = fmaxf(m_run, mx);
  const float alpha = __expf(m_run - m_new);
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(s[t][r] - m_new);
      s[t][r] = e;
      psum += e;
    }
  l_run = l_run * alpha + psum;
  m_run = m_new;
  Frag<T> pf[NPF];
#pragma unroll
  for (int pp = 0; pp < NPF; ++pp) frag_from_f32(pf[pp], s[2 * pp], s[2 * pp + 1]);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    o[t] = o[t] * alpha;
#pragma unroll
    for (int pp = 0; pp < NPF; ++pp) {
      const T* vp = reinterpret_cast<const T*>(vt + (16 * t) * SV) + 32 * pp;
      mma32(o[t], frag_load_halves(vp, vp + 16), pf[pp]);
    }
  }
}

// bf16 form of attn_block_lds with the softmax arithmetic trimmed, for NU independent (16 rows x 1 head) units at once:
//   * scores are scaled once by c = log2(e) / sqrt(D) (packed multiplies) and the running maximum is kept in those units, so
//     the exponential is exp2(u - m): one packed subtract + v_exp per score.  (Folding the scale into an fma in front of the
//     exponential is one instruction shorter but NOT robust: fma(s, c, -round(m c)) of the maximum itself is the rounding
//     residue of m c, which exceeds 1 once |m c| > 2^24 — the long-schedule stress test reaches that — and alpha computed from
//     a re-multiplied m_run changed with hipcc's CSE / contraction choices per instantiation, which broke shard invariance.)
//   * keys past Lk are masked only in the block that contains Lk (wave-uniform test), the padding mask only when MASKED
//     (cross-attention): -1e9 in score units = -1e9 log2(e) in scaled units;
//   * row maxima with v_max3_f32, lane-group reductions with v_permlane swaps (xg_max) instead of ds_bpermute;
//   * the O rescale is skipped in the first block (O = 0);
//   * the units' phases are written unit-interleaved (all QK^T MFMAs, all maxima, all exponentials, all PV MFMAs): a wave
//     that owns two heads overlaps one unit's MFMAs with the other's VALU work instead of running two serial chains.
// The fp32 parity mode keeps attn_block_lds (exact reference operation order).
// ---- V tiles in LDS.  bf16 kernels: [keys][channels] exactly like the K tile (the projection writes V rows beside the K rows,
// the staging copy is the same 16-byte-per-lane row copy) and the PV operand — V^T, 16 channels x 32 keys in the k-slot order
// the P^T accumulators supply (keys 4g .. 4g+3 of the group's first 16-key tile, then of its second) — is read with gfx950's
// transposing LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block whose row q
// is addressed by lanes 4q .. 4q+3 (cdna_hip_programming.md, T10).  Two such reads (8 bytes each) per fragment; with the K
// tile's row padding (stride = 32 mod 128 bytes) the 8 rows a 32-lane half touches tile the 64 banks: conflict-free.
// This replaced a V^T buffer in memory (written through a 2-byte LDS scatter in the projection epilogue, re-staged key-permuted
// with 8-byte stores): LDS bank-conflict fraction 0.31 / 0.32 of enc_a / enc_bc, 4-6 MB more write traffic per launch (r3).
// fp32 parity mode: V^T tiles [channel][keys] as before (the transposing read is a 16-bit instruction).
typedef __bf16 bf16x4_lds __attribute__((ext_vector_type(4)));
DHW_DEV Frag<bf16_t> frag_load_tr(const char* p0, const char* p1) {   // rows q = 0..3 at p0 (lane 4q+p: its row, columns 4p..4p+3), rows 4..7 of the fragment at p1
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_lds*)(uintptr_t)p0);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_lds*)(uintptr_t)p1);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// (fp32 parity mode) one 16-byte piece of a V^T tile row
template <typename T> DHW_DEV void vt_store_piece(char* row, int part, const uint4& v) {
  static_assert(sizeof(T) == 4, "bf16 kernels keep V row-major");
  *reinterpret_cast<uint4*>(row + part * 16) = v;
}

#ifndef DHW_ATT_ABL
#define DHW_ATT_ABL 0   // diagnostic builds only: bit0 = no MFMAs, bit1 = no exponentials, bit2 = no LDS operand reads, bit3 = no max / rescale
#endif
template <int D, int KB, bool MASKED, int NU, bool TAIL>
DHW_DEV void attn_block_bf16(int lane, const Frag<bf16_t> (*qf)[(D + 31) / 32], const char* const (&kt)[NU], int SK, const char* const (&vt)[NU], int SV, int kb,
                             unsigned padbits, int Lk, float (&m_run)[NU], float (&l_run)[NU], f32x4 (*o)[D / 16]) {
  // No implicit fma contraction in here: with nothing between them (TAIL = false) hipcc fused `u = s * c` and `u - m` into
  // fma(s, c, -m) — the non-robust form described above (NaN in the long-schedule stress test) — while the TAIL = true copy,
  // where a select separates the two, kept them apart: two instantiations of one function with different results.
#pragma clang fp contract(off)
  typedef bf16_t T;
  constexpr int DT = D / 16, KCH = (D + 31) / 32, NTILE = KB / 16, NPF = KB / 32;
  static_assert(D % 32 == 0, "head width");
  const int g = lane >> 4;
  const float c = rsqrtf((float)D) * 1.4426950408889634f;
  f32x4 s[NU][NTILE];
  // DHW_ATT_KPF (round 5): a unit's NTILE x KCH key fragments are all requested BEFORE its QK^T MFMAs (hipcc otherwise places each
  // ds_read_b128 directly in front of the MFMA that consumes it: one LDS round trip per MFMA, as in the GEMM main loops — gemm_core.h, run_p);
  // 2 = both units' fragments before the first MFMA.  Same reads, same MFMAs: bit-identical.
#ifndef DHW_ATT_KPF
#define DHW_ATT_KPF 1
#endif
  if constexpr (DHW_ATT_KPF != 0 && DHW_ATT_ABL == 0) {
    Frag<T> kf[NU][NTILE][KCH];
    auto request = [&](int u) {
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) kf[u][t][ch] = frag_load(reinterpret_cast<const T*>(kt[u] + t * 16 * SK) + 32 * ch + 8 * g);
    };
    auto multiply = [&](int u) {
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        s[u][t] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) mma32(s[u][t], kf[u][t][ch], qf[u][ch]);
      }
    };
    if constexpr (DHW_ATT_KPF == 2) {
#pragma unroll
      for (int u = 0; u < NU; ++u) request(u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NU; ++u) multiply(u);
    } else {
      request(0);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) request(u + 1);     // the next unit's fragments fly under this unit's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        multiply(u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      s[u][t] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int ch = 0; ch < KCH; ++ch)
      {
        const Frag<T> kf = (DHW_ATT_ABL & 4) ? qf[u][ch] : frag_load(reinterpret_cast<const T*>(kt[u] + t * 16 * SK) + 32 * ch + 8 * g);
        if constexpr (DHW_ATT_ABL & 1) { asm volatile("" ::"v"(kf.v)); s[u][t][ch] += (float)kf.v[0]; }
        else mma32(s[u][t], kf, qf[u][ch]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < NTILE; ++t) s[u][t] = s[u][t] * c;
  if constexpr (TAIL) {   // the block that holds the end of the sequence (the caller tests kb + KB > Lk: written as a run-time
                          // test here, hipcc if-converted it into a compare + v_cndmask per score in EVERY block)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[u][t][r] = kb + 16 * t + 4 * g + r < Lk ? s[u][t][r] : -INFINITY;
  }
  if constexpr (MASKED) {
    const float neg = -1e9f * 1.4426950408889634f;   // attention.py:44, in scaled units
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[u][t][r] += ((padbits >> (4 * t + r)) & 1u) ? neg : 0.f;
  }
  // the PV operand fragments are requested HERE, in front of the softmax arithmetic: their LDS latency (and the queueing
  // behind the other waves' reads: the LDS operand reads bound this stage) then runs under the max / exp work
  Frag<T> vf[NU][DT][NPF];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int pp = 0; pp < NPF; ++pp)
        vf[u][t][pp] = (DHW_ATT_ABL & 4) ? qf[u][0] : frag_load_tr(vt[u] + 32 * t + (32 * pp) * SV, vt[u] + 32 * t + (32 * pp + 16) * SV);
  __builtin_amdgcn_sched_barrier(0);
  float m_new[NU], alpha[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float mx = fmaxf(fmaxf(s[u][0][0], s[u][0][1]), fmaxf(s[u][0][2], s[u][0][3]));
#pragma unroll
    for (int t = 1; t < NTILE; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[u][t][0], s[u][t][1])), fmaxf(s[u][t][2], s[u][t][3]));
    m_new[u] = (DHW_ATT_ABL & 8) ? 0.f : fmaxf(m_run[u], xg_max(mx));   // finite: every block has >= 1 real key
    alpha[u] = __builtin_amdgcn_exp2f(m_run[u] - m_new[u]);   // exp2(-inf) = 0 in the first block
    m_run[u] = m_new[u];
  }
  Frag<T> pf[NU][NPF];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const f32x4 negm = (f32x4){-m_new[u], -m_new[u], -m_new[u], -m_new[u]};   // (v_pk_add_f32)
    f32x4 psum = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      s[u][t] = s[u][t] - m_new[u];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[u][t][r] = (DHW_ATT_ABL & 2) ? s[u][t][r] : __builtin_amdgcn_exp2f(s[u][t][r]);
      psum += s[u][t];
    }
    l_run[u] = __builtin_fmaf(l_run[u], alpha[u], (psum[0] + psum[1]) + (psum[2] + psum[3]));
#pragma unroll
    for (int pp = 0; pp < NPF; ++pp) frag_from_f32(pf[u][pp], s[u][2 * pp], s[u][2 * pp + 1]);
  }
  if (kb != 0) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < DT; ++t) o[u][t] = o[u][t] * alpha[u];
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int pp = 0; pp < NPF; ++pp) {
        if constexpr (DHW_ATT_ABL & 1) { asm volatile("" ::"v"(vf[u][t][pp].v)); o[u][t][pp] += (float)vf[u][t][pp].v[0]; }
        else mma32(o[u][t], vf[u][t][pp], pf[u][pp]);
      }
}
// One unit — 16 query rows of head h — against KB keys of the staged tiles starting at tile row `krow` (sample key index `kb`): the
// building block of the key-split wave layout of enc_bc_core.h (DHW_ATT_KSPLIT), where the two waves of a row group share the
// third head's keys, 32 each.  kt / vt: the staged K / V tiles ([keys][channels], row strides SK / SV bytes).
template <int KB>
DHW_DEV void attn_unit_bf16(int lane, const Frag<bf16_t> (&qf)[2], const char* kt, int SK, const char* vt, int SV, int h, int krow, int kb, int Lk,
                            float& m, float& l, f32x4 (&o)[4]) {
  const int l15 = lane & 15, g = lane >> 4;
  const char* const ktu[1] = {kt + (krow + l15) * SK + h * 128};
  const char* const vtu[1] = {vt + (krow + 4 * g + (l15 >> 2)) * SV + 8 * (l15 & 3) + h * 128};
  float m1[1] = {m}, l1[1] = {l};
  if (kb + KB > Lk) attn_block_bf16<64, KB, false, 1, true>(lane, &qf, ktu, SK, vtu, SV, kb, 0u, Lk, m1, l1, &o);
  else attn_block_bf16<64, KB, false, 1, false>(lane, &qf, ktu, SK, vtu, SV, kb, 0u, Lk, m1, l1, &o);
  m = m1[0];
  l = l1[0];
}

// (m, l, o) <- the running-softmax state over the union of this wave's keys and another wave's (its state at ms: o as 4 x f32x4, then
// max, then this lane's partial sum; maxima in log2 units as attn_block_bf16 keeps them).  One function, no implicit contraction: every
// kernel that merges gets the same bits.  exp2(-inf) = 0: an empty partner state leaves this one unchanged (this one is never empty).
DHW_DEV void attn_merge_state(float& m, float& l, f32x4 (&o)[4], const float* ms) {
#pragma clang fp contract(off)
  const float m1 = ms[16], l1 = ms[17];
  const float mm = fmaxf(m, m1);
  const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
  l = l * a0 + l1 * a1;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = o[t] * a0 + *reinterpret_cast<const f32x4*>(ms + 4 * t) * a1;
  m = mm;
}

// One KB-key block for the UMAX (1 or 2) units of a wave: unit u = head hs + u * HS, active when that head exists (the
// second unit of a wave may not).  kt / vt: the staged tiles (row stride SK / SV bytes); head h lies h * 64 channels further:
// columns of the K tile and of the bf16 V tile ([keys][channels]), rows of the fp32 V^T tile ([channels][keys]).
template <typename T, int KB, bool MASKED, int UMAX>
DHW_DEV void attn_units(int lane, const Frag<T> (&qf)[UMAX][2], const char* kt, int SK, const char* vt, int SV, int hs, int HS, int H, int kb,
                        unsigned padbits, int Lk, float (&mr)[UMAX], float (&lr)[UMAX], f32x4 (&o)[UMAX][4]) {
  constexpr int ES = sizeof(T);
  const int l15 = lane & 15, g = lane >> 4;
  const char* kt0 = kt + l15 * SK;   // this lane's K row (key l15 of a 16-key tile)
  if constexpr (sizeof(T) == 2) {
    // this lane's address in a 4-key x 16-channel block of the transposing read: key 4g + (l15 >> 2), channels 4 (l15 & 3) ..
    const char* vt0 = vt + (4 * g + (l15 >> 2)) * SV + 4 * (l15 & 3) * ES;
    // (both units of a two-head wave in ONE straight-line block — NU = 2 — measured slower: 27.4 vs 26.4 us for the d = 192
    // layer; the LDS operand reads, not the dependent chains, bound this stage: profiles/r03_attention_ablation.log)
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = hs + u * HS;
      if (h < H) {
        const char* const ktu[1] = {kt0 + h * 64 * ES};
        const char* const vtu[1] = {vt0 + h * 64 * ES};
        float m1[1] = {mr[u]}, l1[1] = {lr[u]};
        if (kb + KB > Lk) attn_block_bf16<64, KB, MASKED, 1, true>(lane, qf + u, ktu, SK, vtu, SV, kb, padbits, Lk, m1, l1, o + u);
        else attn_block_bf16<64, KB, MASKED, 1, false>(lane, qf + u, ktu, SK, vtu, SV, kb, padbits, Lk, m1, l1, o + u);
        mr[u] = m1[0]; lr[u] = l1[0];
      }
    }
  } else {
    const char* vt0 = vt + l15 * SV;   // V^T row (head channel l15 of a 16-channel tile)
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int h = hs + u * HS;
      if (h < H) attn_block_lds<T, 64, KB>(lane, qf[u], kt0 + h * 64 * ES, SK, vt0 + h * 64 * SV + 4 * g * ES, SV, kb, padbits, Lk, mr[u], lr[u], o[u]);
    }
  }
}

// This lane's slice of the key-padding mask of the KB-key block at kb (trow: the sample's token ids, 0 = pad; null = no
// mask): load() only requests the token ids, bits() turns them into the padbits word where it is first needed, so the
// loads are in flight across whatever the caller does in between.
template <int KB>
struct PadMask {
  int64_t tok[KB / 4];
  // Unconditional loads at a clamped index: a per-lane `key < Lk ? trow[key] : 1` compiles to one branch per token with
  // s_waitcnt vmcnt(0) inside — eight dependent memory round trips at the top of every enc_a (r2, .s) — while keys at or
  // past Lk are masked by attn_block_lds itself, so their bits may hold anything.
  DHW_DEV void load(int lane, const int64_t* trow, int kb, int Lk) {
    const int g = lane >> 4;
    if (!trow) {   // (wave-uniform)
#pragma unroll
      for (int i = 0; i < KB / 4; ++i) tok[i] = 1;
      return;
    }
#pragma unroll
    for (int t = 0; t < KB / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + 16 * t + 4 * g + r;
        tok[4 * t + r] = trow[key < Lk ? key : Lk - 1];
      }
  }
  // LAZY: the compares stay HERE.  hipcc otherwise evaluates them right behind the loads — s_waitcnt vmcnt(7) .. vmcnt(0) in front of the caller's
  // first barrier, i.e. a wait for everything requested before them, the first stage's weights included.  Costs the 16 registers of the ids
  // until then (the fp32 parity kernels, at their register limit, keep the eager form).
  template <bool LAZY = false>
  DHW_DEV unsigned bits() const {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < KB / 4; ++i) {
      int64_t t = tok[i];
      if constexpr (LAZY) asm volatile("" : "+v"(t));
      m |= (t == 0 ? 1u : 0u) << i;
    }
    return m;
  }
};

// cooperative copy of one key block into LDS: K rows [kb, kb+KB) x C channels from `ksrc` (row stride ldk elements,
// already offset to the sample's first key row and first K channel) and V^T rows [0,C) x keys [kb, kb+KB) from `vsrc`
// ([C][lpad], offset to the sample).  K rows at or past `kmax` (the sample's key count) and V^T keys past lpad are
// zero-filled, never read.
template <typename T, int KB>
DHW_DEV void attn_stage_kv(char* kt, int SK, char* vt, int SV, const T* ksrc, int ldk, const T* vsrc, int lpad, int C,
                           int kb, int kmax, int tid, int nthreads) {
  constexpr int ES = sizeof(T), EPV = 16 / ES, PPR = KB / EPV;
  const int cpr = C / EPV;            // 16-byte pieces per K row; PPR = pieces per V^T row
  staged_copy_clamped<6>(KB * cpr, tid, nthreads,
      [&](int id) { const int r = id / cpr, cc = id - r * cpr;
                    return reinterpret_cast<const uint4*>(ksrc + (size_t)min(kb + r, kmax - 1) * ldk + cc * EPV); },
      [&](int id) { return kb + id / cpr < kmax; },
      [&](int id) { const int r = id / cpr, cc = id - r * cpr; return reinterpret_cast<uint4*>(kt + r * SK + cc * 16); });
  if constexpr (sizeof(T) == 2) {
    // V rows [kb, kb + KB) x C channels, row stride `lpad` ELEMENTS here (the caller's V row stride), zero past kmax
    staged_copy_clamped<6>(KB * cpr, tid, nthreads,
        [&](int id) { const int r = id / cpr, cc = id - r * cpr;
                      return reinterpret_cast<const uint4*>(vsrc + (size_t)min(kb + r, kmax - 1) * lpad + cc * EPV); },
        [&](int id) { return kb + id / cpr < kmax; },
        [&](int id) { const int r = id / cpr, cc = id - r * cpr; return reinterpret_cast<uint4*>(vt + r * SV + cc * 16); });
  } else {
    constexpr int U = 6;
    const int total = C * PPR;
    for (int base = tid; base < total; base += nthreads * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = min(base + u * nthreads, total - 1);
        const int ch = id / PPR, part = id - ch * PPR;
        v[u] = *reinterpret_cast<const uint4*>(vsrc + (size_t)ch * lpad + (kb + (part + 1) * EPV <= lpad ? kb + part * EPV : 0));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = base + u * nthreads;
        if (id < total) {
          const int ch = id / PPR, part = id - ch * PPR;
          const bool k = kb + (part + 1) * EPV <= lpad;
          vt_store_piece<T>(vt + ch * SV, part, make_uint4(k ? v[u].x : 0u, k ? v[u].y : 0u, k ? v[u].z : 0u, k ? v[u].w : 0u));
        }
      }
    }
  }
}
