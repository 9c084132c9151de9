// heads_core.h — what follows the eps / pen dot products of one stroke row (reference model.py:179-182):
// bias, sigmoid, and the fused scheduler step (utils/nn.py:84-87,110-112, inference.py:85-94) with either the
// caller's noise stream or the counter-based device generator.  Shared by the stand-alone heads kernel (misc.hip)
// and the dec1 ConvBlock kernel, which evaluates the heads straight from its fp32 output tile in LDS.
#pragma once
#include "dhw_common.h"
#include "dhw_kernels.h"

// ---- Philox4x32-10 -> two N(0,1) via Box-Muller
DHW_DEV void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// key = seed; counter = (global sample, position in sample, iteration+1, 0): identical for any batch sharding
DHW_DEV void normal2(uint64_t seed, int64_t sample, int pos, int iter, float& z0, float& z1) {
  uint32_t c0 = (uint32_t)sample, c1 = (uint32_t)((uint64_t)sample >> 32), c2 = (uint32_t)pos, c3 = (uint32_t)(iter + 1);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-2.0f * logf(u0));
  float sn, cs;
  sincosf(6.28318530717958647692f * u1, &sn, &cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

// What heads_finish reads from memory for one stroke row, requested AHEAD of the dot products (the dec1 ConvBlock kernel: its tail was a chain of
// dependent round trips — biases, then the seed, then the sampler state — at the end of every step's last launch).  Every load is unconditional, at a
// valid address: a field the configuration does not have (no scheduler step, external noise) is read from the bias vector instead and never used.
struct HeadsPre {
  float b0, b1, bp, x0, x1, z0, z1;
};
template <typename HP>
DHW_DEV void heads_prefetch(const HP& p, long row, HeadsPre& h) {
  const float* bo = p.b_out;
  const float* xs = p.xt ? p.xt + row * 2 : bo;
  const float* zs = p.z ? p.z + row * 2 : bo;
  const uint64_t* sp = p.seed_ptr ? p.seed_ptr : reinterpret_cast<const uint64_t*>(p.w_out);   // (16 valid bytes either way)
  const uint64_t s0 = sp[0], s1 = sp[1];
  h.b0 = bo[0]; h.b1 = bo[1]; h.bp = p.b_pen[0];
  h.x0 = xs[0]; h.x1 = xs[1];
  h.z0 = zs[0]; h.z1 = zs[1];
  if (p.xt && p.add_noise && !p.z)   // (uniform; arithmetic only: the draw runs while the heads' weight pieces are in flight)
    normal2(s0, (int64_t)s1 + p.sample_off + row / p.L, (int)(row % p.L), p.iter, h.z0, h.z1);
}
// heads_finish on prefetched inputs: the same operations in the same order
template <typename HP>
DHW_DEV void heads_finish_pre(const HP& p, long row, float a0, float a1, float a2, const HeadsPre& h) {
  const float e0 = a0 + h.b0, e1 = a1 + h.b1;
  const float pen = 1.0f / (1.0f + expf(-(a2 + h.bp)));
  if (p.eps) { p.eps[row * 2] = e0; p.eps[row * 2 + 1] = e1; }
  if (p.pen) p.pen[row] = pen;
  if (p.xt) {
    const float z0 = p.add_noise ? h.z0 : 0.f, z1 = p.add_noise ? h.z1 : 0.f;
    float x0 = h.x0, x1 = h.x1;
    if (p.mode == 0) {
      x0 = __fdiv_rn(__fsub_rn(x0, __fmul_rn(p.k0, e0)), p.k1);
      x1 = __fdiv_rn(__fsub_rn(x1, __fmul_rn(p.k0, e1)), p.k1);
      if (p.add_noise) { x0 = __fadd_rn(x0, __fmul_rn(z0, p.k2)); x1 = __fadd_rn(x1, __fmul_rn(z1, p.k2)); }
    } else {
      x0 = __fmul_rn(p.k1, __fsub_rn(x0, __fdiv_rn(__fmul_rn(p.k3, e0), p.k0)));
      x1 = __fmul_rn(p.k1, __fsub_rn(x1, __fdiv_rn(__fmul_rn(p.k3, e1), p.k0)));
      if (p.add_noise) { x0 = __fadd_rn(x0, __fmul_rn(p.k2, z0)); x1 = __fadd_rn(x1, __fmul_rn(p.k2, z1)); }
    }
    p.xt[row * 2] = x0;
    p.xt[row * 2 + 1] = x1;
    if (p.out3) { p.out3[row * 3] = x0; p.out3[row * 3 + 1] = x1; p.out3[row * 3 + 2] = pen; }
  }
}

// a0, a1: eps dot products (without bias); a2: pen-lift logit (without bias); row: global stroke row
// (HP: HeadsParams, or the same struct in the constant address space when the caller's parameters live in a plan in memory)
template <typename HP>
DHW_DEV void heads_finish(const HP& p, long row, float a0, float a1, float a2) {
  const float e0 = a0 + p.b_out[0], e1 = a1 + p.b_out[1];
  const float pen = 1.0f / (1.0f + expf(-(a2 + p.b_pen[0])));
  if (p.eps) { p.eps[row * 2] = e0; p.eps[row * 2 + 1] = e1; }
  if (p.pen) p.pen[row] = pen;
  if (p.xt) {
    float z0 = 0.f, z1 = 0.f;
    if (p.add_noise) {
      if (p.z) { z0 = p.z[row * 2]; z1 = p.z[row * 2 + 1]; }
      else normal2(p.seed_ptr[0], (int64_t)p.seed_ptr[1] + p.sample_off + row / p.L, (int)(row % p.L), p.iter, z0, z1);
    }
    float x0 = p.xt[row * 2], x1 = p.xt[row * 2 + 1];
    // same operation order as the reference, no FMA contraction
    if (p.mode == 0) {   // new: (xt - sqrt(1-abar)*eps)/sqrt(1-beta) + z*sqrt(1-abar_next)
      x0 = __fdiv_rn(__fsub_rn(x0, __fmul_rn(p.k0, e0)), p.k1);
      x1 = __fdiv_rn(__fsub_rn(x1, __fmul_rn(p.k0, e1)), p.k1);
      if (p.add_noise) { x0 = __fadd_rn(x0, __fmul_rn(z0, p.k2)); x1 = __fadd_rn(x1, __fmul_rn(z1, p.k2)); }
    } else {             // standard: (1/sqrt(1-beta)) * (xt - beta*eps/sqrt(1-abar)) [+ sqrt(beta)*z]
      x0 = __fmul_rn(p.k1, __fsub_rn(x0, __fdiv_rn(__fmul_rn(p.k3, e0), p.k0)));
      x1 = __fmul_rn(p.k1, __fsub_rn(x1, __fdiv_rn(__fmul_rn(p.k3, e1), p.k0)));
      if (p.add_noise) { x0 = __fadd_rn(x0, __fmul_rn(p.k2, z0)); x1 = __fadd_rn(x1, __fmul_rn(p.k2, z1)); }
    }
    p.xt[row * 2] = x0;
    p.xt[row * 2 + 1] = x1;
    if (p.out3) { p.out3[row * 3] = x0; p.out3[row * 3 + 1] = x1; p.out3[row * 3 + 2] = pen; }
  }
}
