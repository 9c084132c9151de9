// epilogue.h — what a wave does with its MFMA accumulator tiles after a GEMM stage of the fused kernels: SiLU and the
// write of the tiles into an LDS activation tile.  Both were the binding phase of every stage boundary (per-wave stamps,
// profiles/r03_convblock_wave_stamps.log: an epilogue of 16-32 values per lane took 0.7-1.5 us, longer than the stage's
// MFMA loop), for two reasons visible in the ISA:
//   * SiLU per element as written (x * rcp(1 + exp(-x)) inside the store loop) became ~9 dependent VALU instructions per
//     value with the two quarter-rate transcendentals in a serial chain, single conversions (v_cvt_pk_bf16_f32 v, v, 0),
//     per-element v_cndmask row masks and v_perm re-packing;
//   * the stores were one ds_write_b64 per tile: 16 lanes of a store group write the same 8-byte column of 16 rows whose
//     stride is 32 (mod 128) bytes — the padding the ds_read_b128 operand reads need (gemm_core.h) — i.e. a 4-way bank
//     conflict on every store (SQ_LDS_BANK_CONFLICT 33-35 % of the LDS cycles, profiles/r02_sq_counters.json).
// Here: SiLU runs over a whole group of tiles phase by phase (packed multiplies, all exponentials back to back, packed adds,
// all reciprocals, packed multiplies: ~3.9 VALU instructions per value), conversions and row masks act on packed pairs, and
// two tiles are exchanged between lane groups with v_permlane16_swap_b32 so that each lane owns 8 consecutive channels of
// one row and stores 16 bytes (ds_write_b128: 2-way instead of 4-way conflicts, half the store instructions).
#pragma once
#include <type_traits>
#include "dhw_common.h"

// ---- SiLU over N accumulator tiles, in place.  bf16 kernels: exp2 / rcp hardware approximations (the result is rounded
// to bf16), identical arithmetic to silu_t<bf16_t>; fp32 parity mode: the exact form, element by element.
template <typename T, int N>
DHW_DEV void silu_tiles(f32x4 (&v)[N]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[n][k] = silu_f(v[n][k]);
  } else {
    f32x4 t[N];
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = v[n] * -1.4426950408889634f;   // __expf(-x) = exp2(-x * log2(e))
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[n][k] = __builtin_amdgcn_exp2f(t[n][k]);
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = t[n] + 1.0f;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[n][k] = __builtin_amdgcn_rcpf(t[n][k]);
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] * t[n];
  }
}
// the same over a wave's NT x MT tiles, in groups of at most 8 (the temporaries of a group are 4 VGPRs per tile)
template <typename T, int N, int BASE = 0>
DHW_DEV void silu_flat(f32x4* flat) {
  constexpr int G = N - BASE >= 8 ? 8 : N - BASE;
  if constexpr (G > 0) {
    f32x4 grp[G];
#pragma unroll
    for (int n = 0; n < G; ++n) grp[n] = flat[BASE + n];
    silu_tiles<T, G>(grp);
#pragma unroll
    for (int n = 0; n < G; ++n) flat[BASE + n] = grp[n];
    silu_flat<T, N, BASE + G>(flat);
  }
}
template <typename T, int NT, int MT>
DHW_DEV void silu_tiles2(f32x4 (&v)[NT][MT]) { silu_flat<T, NT * MT>(&v[0][0]); }

// round an fp32 tile to the element type and back (what a store + reload of the tile would do)
template <typename T>
DHW_DEV f32x4 round_to(const f32x4& v) {
  if constexpr (sizeof(T) == 4) return v;
  else {
    const bf16x4 o = __builtin_convertvector(v, bf16x4);
    return (f32x4){(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
  }
}

// 4 fp32 values -> 4 packed bf16 (two v_cvt_pk_bf16_f32)
DHW_DEV uint2 pack4_bf16(const f32x4& v) {
  const bf16x4 o = __builtin_convertvector(v, bf16x4);
  return __builtin_bit_cast(uint2, o);
}

// One pair of accumulator tiles -> LDS.  p0 / p1: where THIS lane's 4 channels of tile 0 / tile 1 would go with a plain
// 8-byte store (row of the lane, channel n0 + 4g).  keep0 / keep1: false = the lane's row of that tile is written as
// zeros; do0 / do1: false = the row of that tile is not written at all.  After the swap lane (l15, g) holds channels
// 8 (g >> 1) .. + 7 of row l15 of tile (g & 1), i.e. the 16 bytes at (g & 1 ? p1 - 8 bytes : p0).
// Must be called by all 64 lanes (v_permlane16_swap reads the partner lanes' registers).
// lane: the caller's lane index (a body derives it from body_tid(), dhw_common.h: never from threadIdx.x directly)
DHW_DEV void store_pair(int lane, bf16_t* p0, bf16_t* p1, const f32x4& v0, const f32x4& v1, bool keep0 = true, bool keep1 = true,
                        bool do0 = true, bool do1 = true) {
  uint2 a = pack4_bf16(v0), b = pack4_bf16(v1);
  a.x = keep0 ? a.x : 0u; a.y = keep0 ? a.y : 0u;
  b.x = keep1 ? b.x : 0u; b.y = keep1 ? b.y : 0u;
  const auto lo = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  const bool odd = (lane >> 4) & 1;
  char* dst = odd ? reinterpret_cast<char*>(p1) - 8 : reinterpret_cast<char*>(p0);
  if (odd ? do1 : do0) *reinterpret_cast<uint4*>(dst) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
}
DHW_DEV void store_pair(int lane, float* p0, float* p1, const f32x4& v0, const f32x4& v1, bool keep0 = true, bool keep1 = true,
                        bool do0 = true, bool do1 = true) {
  if (do0) store4(p0, keep0 ? v0 : (f32x4){0, 0, 0, 0});
  if (do1) store4(p1, keep1 ? v1 : (f32x4){0, 0, 0, 0});
}
template <typename T>
DHW_DEV void store_one(T* p, const f32x4& v, bool keep = true, bool doit = true) {
  if (doit) store4(p, keep ? v : (f32x4){0, 0, 0, 0});
}

// The NT x MT accumulator tiles of a wave -> an LDS tile: tile (i, j) = rows row0 + 16 j + (lane & 15), channels
// n0 + 16 i + 0..3 (n0 includes this lane's 4 (lane >> 4)).  keep(j): the lane's row of row tile j is real (else zeros);
// valid(j): it is written at all.  Tiles are paired in (i, j) order; an odd tile count leaves one 8-byte store.
template <typename T, int NT, int MT, typename KeepF, typename ValidF>
DHW_DEV void store_tiles(int lane, char* tile, int S, int row0, int n0, const f32x4 (&v)[NT][MT], KeepF keep, ValidF valid) {
  const int l15 = lane & 15;
  constexpr int N = NT * MT;
  auto ptr = [&](int t) { const int i = t / MT, j = t - i * MT; return reinterpret_cast<T*>(tile + (row0 + j * 16 + l15) * S) + n0 + 16 * i; };
#pragma unroll
  for (int t = 0; t + 1 < N; t += 2) {
    const int j0 = t % MT, j1 = (t + 1) % MT;
    store_pair(lane, ptr(t), ptr(t + 1), v[t / MT][j0], v[(t + 1) / MT][j1], keep(j0), keep(j1), valid(j0), valid(j1));
  }
  if constexpr (N & 1) store_one<T>(ptr(N - 1), v[NT - 1][MT - 1], keep(MT - 1), valid(MT - 1));
}
#ifndef DHW_ENC_PAIRST
#define DHW_ENC_PAIRST 0   // 1 = paired 16-byte stores (lane exchange) in the EncoderLayer epilogues too: measured 19.08 vs 19.01 ms (profiles/r04_enc_pairstore_ab.log) -> off
#endif
// the EncoderLayer kernels' form: paired 16-byte stores, or (DHW_ENC_PAIRST=0) one 8-byte store per tile
template <typename T, int NT, int MT>
DHW_DEV void enc_store_tiles(int lane, char* tile, int S, int row0, int n0, const f32x4 (&v)[NT][MT]) {
  if constexpr (DHW_ENC_PAIRST) {
    store_tiles<T, NT, MT>(lane, tile, S, row0, n0, v, [](int) { return true; }, [](int) { return true; });
  } else {
    const int l15 = lane & 15;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) store4(reinterpret_cast<T*>(tile + (row0 + j * 16 + l15) * S) + n0 + 16 * i, v[i][j]);
  }
}
template <typename T, int NT, int MT>
DHW_DEV void store_tiles(int lane, char* tile, int S, int row0, int n0, const f32x4 (&v)[NT][MT]) {
  store_tiles<T, NT, MT>(lane, tile, S, row0, n0, v, [](int) { return true; }, [](int) { return true; });
}

// 8 packed bf16 (one 16-byte piece of a staged tile) -> SiLU of each, packed again; fp32: 4 values
template <typename T>
DHW_DEV uint4 silu_piece(const uint4& w) {
  if constexpr (sizeof(T) == 4) {
    uint4 o = w;
    float* e = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = silu_f(e[i]);
    return o;
  } else {
    f32x4 v[2];
    v[0] = (f32x4){__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
    v[1] = (f32x4){__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
    silu_tiles<T, 2>(v);
    const uint2 a = pack4_bf16(v[0]), b = pack4_bf16(v[1]);
    return make_uint4(a.x, a.y, b.x, b.y);
  }
}

// A whole stage epilogue, tile pair by tile pair: v = aff(i, acc[i][j]); optional SiLU; store into the LDS tile; then
// between(pair index) — the caller's hook for requesting the next stage's weight fragments (WRing::fill_chunk) so that
// those requests are spread between the pairs' VALU work instead of blocking in front of it.  The scheduling barrier keeps
// hipcc from clustering the loads again.  Returns the number of pairs processed (= calls of between()).
template <typename T, int NT, int MT, bool SILU, typename AffF, typename KeepF, typename ValidF, typename BetweenF>
DHW_DEV void epilogue_pairs(int lane, char* tile, int S, int row0, int n0, const f32x4 (&acc)[NT][MT], AffF aff, KeepF keep, ValidF valid, BetweenF between) {
  const int l15 = lane & 15;
  constexpr int N = NT * MT;
  auto ptr = [&](int t) { const int i = t / MT, j = t - i * MT; return reinterpret_cast<T*>(tile + (row0 + j * 16 + l15) * S) + n0 + 16 * i; };
#pragma unroll
  for (int t = 0; t + 1 < N; t += 2) {
    const int i0 = t / MT, j0 = t % MT, i1 = (t + 1) / MT, j1 = (t + 1) % MT;
    f32x4 v[2] = {aff(i0, acc[i0][j0]), aff(i1, acc[i1][j1])};
    if constexpr (SILU) silu_tiles<T, 2>(v);
    store_pair(lane, ptr(t), ptr(t + 1), v[0], v[1], keep(j0), keep(j1), valid(j0), valid(j1));
    between(t / 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (N & 1) {
    f32x4 v[1] = {aff(NT - 1, acc[NT - 1][MT - 1])};
    if constexpr (SILU) silu_tiles<T, 1>(v);
    store_one<T>(ptr(N - 1), v[0], keep(MT - 1), valid(MT - 1));
    between(N / 2);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NT, int MT> constexpr int epilogue_steps() { return (NT * MT + 1) / 2; }
