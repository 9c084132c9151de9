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
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
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
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
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
DHW_DEV void attn_block_lds(const Frag<T> (&qf)[(D + 31) / 32], const char* kt, int SK, const char* vt, int SV, int kb,
                            unsigned padbits, int Lk, float& m_run, float& l_run, f32x4 (&o)[D / 16]) {
  constexpr int ES = sizeof(T), DT = D / 16, KCH = (D + 31) / 32, NTILE = KB / 16, NPF = KB / 32;
  const int lane = threadIdx.x & 63, g = lane >> 4;
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
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
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

// This lane's slice of the key-padding mask of the KB-key block at kb (trow: the sample's token ids, 0 = pad; null = no
// mask): load() only requests the token ids, bits() turns them into the padbits word where it is first needed, so the
// loads are in flight across whatever the caller does in between.
template <int KB>
struct PadMask {
  int64_t tok[KB / 4];
  // Unconditional loads at a clamped index: a per-lane `key < Lk ? trow[key] : 1` compiles to one branch per token with
  // s_waitcnt vmcnt(0) inside — eight dependent memory round trips at the top of every enc_a (r2, .s) — while keys at or
  // past Lk are masked by attn_block_lds itself, so their bits may hold anything.
  DHW_DEV void load(const int64_t* trow, int kb, int Lk) {
    const int g = (threadIdx.x & 63) >> 4;
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
  DHW_DEV unsigned bits() const {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < KB / 4; ++i) m |= (tok[i] == 0 ? 1u : 0u) << i;
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
  staged_copy<6>(KB * cpr, tid, nthreads,
      [&](int id) { const int r = id / cpr, cc = id - r * cpr;
                    return kb + r < kmax ? reinterpret_cast<const uint4*>(ksrc + (size_t)(kb + r) * ldk + cc * EPV) : nullptr; },
      [&](int id) { const int r = id / cpr, cc = id - r * cpr; return reinterpret_cast<uint4*>(kt + r * SK + cc * 16); });
  staged_copy<6>(C * PPR, tid, nthreads,
      [&](int id) { const int ch = id / PPR, part = id - ch * PPR;
                    return kb + (part + 1) * EPV <= lpad ? reinterpret_cast<const uint4*>(vsrc + (size_t)ch * lpad + kb + part * EPV) : nullptr; },
      [&](int id) { const int ch = id / PPR, part = id - ch * PPR; return reinterpret_cast<uint4*>(vt + ch * SV + part * 16); });
}
