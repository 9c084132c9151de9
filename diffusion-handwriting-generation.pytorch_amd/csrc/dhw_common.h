// dhw_common.h — shared device helpers for the gfx950 kernels.
//
// MFMA conventions used everywhere (cdna_hip_programming.md §3):
//   * one "k-chunk" = 32 contraction elements; a fragment holds, per lane,
//     the 8 consecutive k-elements  k = 8*(lane>>4) + j  of row/col (lane&15).
//   * bf16: one v_mfma_f32_16x16x32_bf16 per k-chunk.
//   * f32 : eight v_mfma_f32_16x16x4_f32 per k-chunk (MFMA i consumes element
//     j = i of both fragments: the 4 lane groups then supply k = 8g+i — any
//     k-permutation is fine as long as A and B agree).  Exact f32.
//   * accumulator tile 16x16: acc[r] = C[row = 4*(lane>>4) + r][col = lane&15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DHW_DEV __device__ __forceinline__

// The thread index a fused block body works with.  In the persistent per-step kernel (persist.hip, -> DHW_OPAQUE_TID) the
// bodies run inside a loop over phases: everything they derive from the thread index (tile addresses, lane offsets of the
// weight stream, ...) is loop-invariant there, and hipcc hoisted all of it in front of the loop — hundreds of values live
// across every body, 836 spilled VGPRs.  An opaque copy per body invocation keeps those computations where they are used.
#ifdef DHW_OPAQUE_TID
__device__ __forceinline__ int body_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#else
__device__ __forceinline__ int body_tid() { return threadIdx.x; }
#endif
#ifndef DHW_UNIFORM_WAVE
#define DHW_UNIFORM_WAVE 1
#endif

// Diagnostic per-stage time stamps (s_memrealtime -> p.stamps[slot]) exist only in builds with -DDHW_STAMPS (the
// micro-benchmarks under tools/).  Even behind a run-time `if (p.stamps ...)` the stamp blocks changed the product's code:
// each is a conditional region with a global store, and hipcc's s_waitcnt bookkeeping at the region's join point waited
// for the weight prefetch issued just before (s_waitcnt vmcnt(0..2) instead of a counted vmcnt(17)).
#ifdef DHW_STAMPS
#define DHW_STAMP_IF(cond, slot_expr, value) do { if (cond) p.stamps[slot_expr] = (value); } while (0)
#else
#define DHW_STAMP_IF(cond, slot_expr, value) do { } while (0)
#endif

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 lo, hi; };

// acc += A(16 x 32) * B(32 x 16); a = A-operand fragment (rows), b = B-operand fragment (cols)
DHW_DEV void mma32(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
DHW_DEV void mma32(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[0], b.lo[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[1], b.lo[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[2], b.lo[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[3], b.lo[3], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[0], b.hi[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[1], b.hi[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[2], b.hi[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[3], b.hi[3], acc, 0, 0, 0);
}

// load a fragment (8 consecutive elements) from a 16-byte aligned address
DHW_DEV Frag<bf16_t> frag_load(const bf16_t* p) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(p);
  return f;
}
DHW_DEV Frag<float> frag_load(const float* p) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const f32x4*>(p);
  f.hi = *reinterpret_cast<const f32x4*>(p + 4);
  return f;
}
template <typename T> DHW_DEV Frag<T> frag_zero();
template <> DHW_DEV Frag<bf16_t> frag_zero<bf16_t>() {
  Frag<bf16_t> f;
  for (int i = 0; i < 8; ++i) f.v[i] = (bf16_t)0.0f;
  return f;
}
template <> DHW_DEV Frag<float> frag_zero<float>() {
  Frag<float> f;
  f.lo = (f32x4){0, 0, 0, 0};
  f.hi = (f32x4){0, 0, 0, 0};
  return f;
}
// fragment from two groups of 4 fp32 values (element j<4 from a, j>=4 from b)
DHW_DEV void frag_from_f32(Frag<bf16_t>& f, const f32x4& a, const f32x4& b) {
  for (int i = 0; i < 4; ++i) { f.v[i] = (bf16_t)a[i]; f.v[4 + i] = (bf16_t)b[i]; }
}
DHW_DEV void frag_from_f32(Frag<float>& f, const f32x4& a, const f32x4& b) { f.lo = a; f.hi = b; }

// fragment from two 4-element halves in memory (elements 0..3 at p0, 4..7 at p1), no conversion
DHW_DEV Frag<bf16_t> frag_load_halves(const bf16_t* p0, const bf16_t* p1) {
  Frag<bf16_t> f;
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p0), b = *reinterpret_cast<const bf16x4*>(p1);
  f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
DHW_DEV Frag<float> frag_load_halves(const float* p0, const float* p1) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const f32x4*>(p0);
  f.hi = *reinterpret_cast<const f32x4*>(p1);
  return f;
}

// 4 consecutive elements <-> f32x4
DHW_DEV f32x4 load4(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
DHW_DEV f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DHW_DEV void store4(bf16_t* p, const f32x4& v) {
  bf16x4 o;
  for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x4*>(p) = o;
}
DHW_DEV void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

#include "xcd_swizzle.h"

// Cooperative global -> LDS copy of `total` 16-byte pieces: piece id -> source pointer (null = zero fill) and LDS
// destination.  U loads are issued back to back before the first store, so a thread pays the L2 / Infinity-Cache latency
// once per U pieces, not once per piece (a plain load-store loop measured 4 us per 98 KB K/V block in enc_bc).
template <int U, typename SrcF, typename DstF>
DHW_DEV void staged_copy(int total, int tid, int nthreads, SrcF src, DstF dst) {
  for (int base = tid; base < total; base += nthreads * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = base + u * nthreads;
      v[u] = make_uint4(0, 0, 0, 0);
      if (id < total) {
        const uint4* sp = src(id);
        if (sp) v[u] = *sp;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = base + u * nthreads;
      if (id < total) *dst(id) = v[u];
    }
  }
}

// The same with src(id) ALWAYS a valid address (the caller clamps the row) and keep(id) saying whether the piece is real or reads as zero: no branch
// around a load, so the U requests of a pass really are in flight together (the null-pointer form above compiles to one branch per piece with
// s_waitcnt vmcnt(0) at its join — U dependent round trips per pass: round 5, .s of the text-key blocks behind the first).
template <int U, typename SrcF, typename KeepF, typename DstF>
DHW_DEV void staged_copy_clamped(int total, int tid, int nthreads, SrcF src, KeepF keep, DstF dst) {
  for (int base = tid; base < total; base += nthreads * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = min(base + u * nthreads, total - 1);
      v[u] = *src(id);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = base + u * nthreads;
      if (id < total) *dst(id) = keep(id) ? v[u] : make_uint4(0, 0, 0, 0);
    }
  }
}

// The same copy split in two: load() requests this thread's <= U pieces, store() writes them to LDS.  Several tiles staged
// by one workgroup issue ALL their loads before the first store, so they cost one memory round trip together instead
// of one each (enc_a staged x, then the text keys, then the text values: three dependent L2 / HBM latencies).
// Requires total <= nthreads * U.
template <int U>
struct CopyRegs {
  uint4 v[U];
  // src(id) must return a VALID address for every id < total (clamp the row instead of returning null): a conditional
  // per-lane load compiles to a branch with s_waitcnt vmcnt(0) at its join, which serialises the tiles again.  Pieces that
  // must read as zero are cleared by store() (keep(id) == false), with a select instead of a branch.
  template <typename SrcF>
  DHW_DEV void load(int total, int tid, int nthreads, SrcF src) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = tid + u * nthreads;
      v[u] = *src(id < total ? id : total - 1);
    }
  }
  // sink(id, value): the caller places the (already zero-selected) piece itself
  template <typename SinkF, typename KeepF>
  DHW_DEV void store_to(int total, int tid, int nthreads, SinkF sink, KeepF keep) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = tid + u * nthreads;
      if (id < total) {
        const bool k = keep(id);
        sink(id, make_uint4(k ? v[u].x : 0u, k ? v[u].y : 0u, k ? v[u].z : 0u, k ? v[u].w : 0u));
      }
    }
  }
  template <typename DstF, typename KeepF>
  DHW_DEV void store(int total, int tid, int nthreads, DstF dst, KeepF keep) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = tid + u * nthreads;
      if (id < total) {
        const bool k = keep(id);
        *dst(id) = make_uint4(k ? v[u].x : 0u, k ? v[u].y : 0u, k ? v[u].z : 0u, k ? v[u].w : 0u);
      }
    }
  }
};

// Reductions over the 4 lane groups g = lane >> 4 of a wave (lanes with equal lane & 15): every lane ends with the result.
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two registers inside the VALU; __shfl_xor
// is a ds_bpermute_b32, i.e. an LDS-pipe round trip (~100 cycles) on the serial chain of every softmax block and LayerNorm.
DHW_DEV float xg_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
DHW_DEV float xg_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue
// (s_waitcnt vmcnt(0)), which would stall every wave until the NEXT stage's prefetched weight fragments have
// landed; the fused kernels exchange data between waves through LDS only, so lgkmcnt(0) + s_barrier suffices
// (cdna_hip_programming.md §5 "Pipelining across barriers").
DHW_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

DHW_DEV float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// SiLU for element type T: the bf16 path uses the 1-ulp hardware reciprocal (an IEEE fp32 divide is ~10 VALU
// instructions and the result is rounded to bf16 anyway); the fp32 parity path keeps the exact divide.
template <typename T> DHW_DEV float silu_t(float x);
template <> DHW_DEV float silu_t<bf16_t>(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <> DHW_DEV float silu_t<float>(float x) { return silu_f(x); }
DHW_DEV float to_f(bf16_t x) { return (float)x; }
DHW_DEV float to_f(float x) { return x; }
template <typename T> DHW_DEV T from_f(float x) { return (T)x; }
