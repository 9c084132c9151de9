// persist.hip — the persistent per-step kernel (see persist.h): the fused block bodies of convblock_core.h / enc_bc_core.h run
// as phases of one launch, tiles handed out by per-XCD tickets, hand-offs through per-sample counters.
#define DHW_OPAQUE_TID 1
#include "persist.h"
#include "convblock_core.h"
#include "enc_bc_core.h"

#define AS4 __attribute__((address_space(4)))

namespace {

DHW_DEV unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x & (STEP_XCDS - 1);
}
typedef __attribute__((address_space(1))) unsigned gu32;
DHW_DEV unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DHW_DEV unsigned add_agent(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// diagnostics: s_memrealtime (100 MHz) of workgroup w in phase ph at {ticket drawn, inputs ready, body done} -> plan->trace, when the
// host gave the plan a trace buffer (DHW_PERSIST_TRACE=1 at dhw_create; tools/persist_trace.py).  Outside the block bodies.
#define PTRACE(slot) do { if (trace && tid == 0) trace[((size_t)blockIdx.x * STEP_MAX_PHASES + ph) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// One phase body = one real function (not inlined): inlined into one kernel the eleven bodies shared a register allocation
// that spilled 836 VGPRs.  Arguments of a device function travel in vector registers; the callee makes the (uniform) ones
// scalar again, so the plan is read with scalar loads exactly as kernel arguments are.
#ifdef DHW_PERSIST_CALLS
#define PHASE_FN __device__ __attribute__((noinline))
#else
#define PHASE_FN __device__ __forceinline__
#endif
DHW_DEV const AS4 StepPhase& phase_ref(unsigned long long pp) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pp), hi = __builtin_amdgcn_readfirstlane((unsigned)(pp >> 32));
  return *(const AS4 StepPhase*)(((unsigned long long)hi << 32) | lo);
}
template <int BM, int CO, int UPC, int CH, int CIN>
PHASE_FN void phase_conv(unsigned long long pp, int b, int m0, char* smem) {
  const AS4 StepPhase& P = phase_ref(pp);
  convblock_body<bf16_t, BM, CO, 8, 1, UPC, CH, CIN>(P.cb, P.nx, __builtin_amdgcn_readfirstlane(b), __builtin_amdgcn_readfirstlane(m0), smem);
}
template <int DM, int BM>
PHASE_FN void phase_a(unsigned long long pp, int b, int m0, char* smem) {
  const AS4 StepPhase& P = phase_ref(pp);
  enc_a_tile<bf16_t, DM, BM>(P.el, __builtin_amdgcn_readfirstlane(b), __builtin_amdgcn_readfirstlane(m0), smem);
}
template <int DM, int BM, int NEXT>
PHASE_FN void phase_bc(unsigned long long pp, int b, int m0, char* smem) {
  const AS4 StepPhase& P = phase_ref(pp);
  enc_bc_body<bf16_t, DM, BM, NEXT>(P.el, P.nx, __builtin_amdgcn_readfirstlane(b), __builtin_amdgcn_readfirstlane(m0), smem);
}

constexpr unsigned long long SPIN_LIMIT_TICKS = 300000000ull;   // 3 s of the 100 MHz s_memrealtime clock

// What a workgroup needs to know between two tiles, re-derived from the plan every time (LAUNDER makes the plan pointer opaque per
// iteration): nothing but that pointer and three LDS words stays live across a block body.  With the loop state kept in registers
// the bodies that need all 256 VGPRs / ~100 SGPRs as kernels spilled inside their stages, and a scratch reload waits — vector
// memory returns in order — for the whole weight prefetch issued in front of it.
struct StepCtx {
  int xcc, B, ns, s0, total;
  unsigned *head, *done, *exitc;
  unsigned long long* trace;
};
DHW_DEV const AS4 StepPlan* launder(const StepPlan* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (const AS4 StepPlan*)v;
}
DHW_DEV StepCtx step_ctx(const AS4 StepPlan* plan) {
  StepCtx c;
  c.xcc = (int)xcc_id();
  c.B = plan->B;
  c.s0 = c.xcc * plan->spx;
  c.ns = max(0, min(c.B - c.s0, plan->spx));   // samples this XCD owns
  c.total = c.ns * plan->cum_tps[plan->nphase];
  unsigned* const sync = plan->sync;
  c.head = sync + c.xcc * 16;                   // this XCD's ticket counter (a 64-byte line of its own)
  c.done = sync + STEP_XCDS * 16;
  c.exitc = c.done + (size_t)STEP_MAX_PHASES * c.B;
  c.trace = plan->trace;
  return c;
}
#undef PTRACE
#define PTRACE(slot) do { if (c.trace && threadIdx.x == 0) c.trace[((size_t)blockIdx.x * STEP_MAX_PHASES + ph) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_kernel(const StepPlan* plan_g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ __attribute__((aligned(16))) int ctl[4];   // [0] current ticket, [1] next ticket, [2] exit order
  {
    const AS4 StepPlan* plan = launder(plan_g);
    const StepCtx c = step_ctx(plan);
    if (c.trace && threadIdx.x == 0) c.trace[((size_t)blockIdx.x * STEP_MAX_PHASES + 0) * 4 + 3] = __builtin_amdgcn_s_memrealtime();   // kernel entry
    if (threadIdx.x == 0) ctl[1] = (int)add_agent(c.head, 1u);
  }
  for (;;) {
    int g, ph, b, m0;
    unsigned long long pp;
    int kind;
    {
      const AS4 StepPlan* plan = launder(plan_g);
      const StepCtx c = step_ctx(plan);
      if (threadIdx.x == 0) {
        const int gt = ctl[1];
        if (gt < c.total) {
          ph = 0;
          while (gt >= c.ns * plan->cum_tps[ph + 1]) ++ph;
          PTRACE(0);
          if (ph > 0) {
            // L1 first: from here to the tile's loads this workgroup touches no activation (persist.h)
#ifndef DHW_PERSIST_NOFENCE   // (timing experiment only: without the L1 invalidate the tile's loads may return stale rows)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            const int t = gt - c.ns * plan->cum_tps[ph];
            const unsigned need = (unsigned)plan->ph[ph - 1].tps;
            const unsigned* cnt = c.done + (size_t)(ph - 1) * c.B + c.s0 + t / plan->ph[ph].tps;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (ld_agent(cnt) < need) {
              __builtin_amdgcn_s_sleep(1);
              if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
                __hip_atomic_store(plan->err, 1u + (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
              }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed
          }
        }
        ctl[0] = gt;
      }
      __syncthreads();
      g = __builtin_amdgcn_readfirstlane(ctl[0]);
      if (g >= c.total) break;
      ph = 0;
      while (g >= c.ns * plan->cum_tps[ph + 1]) ++ph;
      const AS4 StepPhase& P = plan->ph[ph];
      const int tps = P.tps, t = g - c.ns * plan->cum_tps[ph];
      b = c.s0 + t / tps;
      m0 = (t - (t / tps) * tps) * P.rows;
      kind = P.kind;
      pp = (unsigned long long)(const AS4 void*)&P;
      PTRACE(1);
    }
#ifdef DHW_PERSIST_ONLY   // diagnostics: a kernel that holds ONLY the bodies in this bit mask (the other phases complete at once):
                          // what a body costs inside the launch when it is not compiled together with the ten others
#define PK_ON(k) (((DHW_PERSIST_ONLY) >> (k)) & 1)
#else
#define PK_ON(k) 1
#endif
    switch (kind) {
      case PK_CONV_ENC1: if constexpr (PK_ON(PK_CONV_ENC1)) phase_conv<128, 128, 0, 0, 128>(pp, b, m0, smem); break;
      case PK_CONV_ENC2A: if constexpr (PK_ON(PK_CONV_ENC2A)) phase_conv<64, 192, 0, 1, 128>(pp, b, m0, smem); break;
      case PK_BC192: if constexpr (PK_ON(PK_BC192)) phase_bc<192, 64, 0>(pp, b, m0, smem); break;
      case PK_CONV_ENC4: if constexpr (PK_ON(PK_CONV_ENC4)) phase_conv<48, 256, 0, 0, 192>(pp, b, m0, smem); break;
      case PK_A256: if constexpr (PK_ON(PK_A256)) phase_a<256, 32>(pp, b, m0, smem); break;
      case PK_BC256_N2: if constexpr (PK_ON(PK_BC256_N2)) phase_bc<256, 32, 2>(pp, b, m0, smem); break;
      case PK_BC384_N1: if constexpr (PK_ON(PK_BC384_N1)) phase_bc<384, 16, 1>(pp, b, m0, smem); break;
      case PK_BC384: if constexpr (PK_ON(PK_BC384)) phase_bc<384, 16, 0>(pp, b, m0, smem); break;
      case PK_CONV_DEC3: if constexpr (PK_ON(PK_CONV_DEC3)) phase_conv<48, 256, 384, 0, 384>(pp, b, m0, smem); break;
      case PK_CONV_DEC2: if constexpr (PK_ON(PK_CONV_DEC2)) phase_conv<64, 192, 256, 0, 256>(pp, b, m0, smem); break;
      case PK_CONV_DEC1: if constexpr (PK_ON(PK_CONV_DEC1)) phase_conv<128, 128, 192, 0, 192>(pp, b, m0, smem); break;
      default: break;
    }
    {
      const AS4 StepPlan* plan = launder(plan_g);
      const StepCtx c = step_ctx(plan);
      // which tile this was: from the ticket again (nothing but LDS words lives across the body)
      g = __builtin_amdgcn_readfirstlane(ctl[0]);
      ph = 0;
      while (g >= c.ns * plan->cum_tps[ph + 1]) ++ph;
      const int tps = plan->ph[ph].tps, t = g - c.ns * plan->cum_tps[ph];
      b = c.s0 + t / tps;
      // the next ticket is drawn while this tile's stores drain (the draw's round trip is off the critical path)
      if (threadIdx.x == 0) ctl[1] = (int)add_agent(c.head, 1u);
      // publish: every wave's stores have reached the XCD's L2, then ONE counter add for the sample
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PTRACE(2);
      if (threadIdx.x == 0) (void)add_agent(c.done + (size_t)ph * c.B + b, 1u);
    }
  }
  {
    const AS4 StepPlan* plan = launder(plan_g);
    const StepCtx c = step_ctx(plan);
    if (c.trace && threadIdx.x == 0) c.trace[((size_t)blockIdx.x * STEP_MAX_PHASES + 1) * 4 + 3] = ((unsigned long long)c.xcc << 56) | __builtin_amdgcn_s_memrealtime();   // exit (+ XCC id)
    // the last workgroup to leave zeroes the tickets and counters for the next launch
    if (threadIdx.x == 0) ctl[2] = (int)add_agent(c.exitc, 1u);
    __syncthreads();
    if ((unsigned)ctl[2] == gridDim.x - 1) {
      unsigned* const sync = plan->sync;
      // Every XCD that owns samples must have run workgroups of this launch: its samples' tiles are dequeued only there.  An XCD
      // without one (partitioned / CU-masked device, fewer than eight XCDs) leaves its ticket counter below its tile count — nobody
      // waits on those tiles, so no bounded spin trips; the last workgroup to leave reports it (error word 0x100 + XCD, read by
      // the next dhw_sample, which falls back to the per-kernel launches).  A served XCD ends at tiles + its resident workgroups.
      if (threadIdx.x < STEP_XCDS) {
        const int x = threadIdx.x, ns_x = max(0, min(c.B - x * plan->spx, plan->spx));
        const unsigned tot = (unsigned)(ns_x * plan->cum_tps[plan->nphase]);
        if (tot && ld_agent(sync + x * 16) < tot) __hip_atomic_store(plan->err, 0x100u + (unsigned)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      const int n = STEP_XCDS * 16 + STEP_MAX_PHASES * c.B + 16;
      for (int i = threadIdx.x; i < n; i += 512) sync[i] = 0u;
    }
  }
}

size_t step_lds_bytes() {
  size_t m = 0;
  auto up = [&](size_t v) { m = v > m ? v : m; };
  up(lds_bytes<bf16_t, 128, 128>(128));
  up(std::max(lds_bytes<bf16_t, 64, 192>(128), (size_t)2 * 64 * tile_stride<bf16_t>(192) + 2 * 8 * 64 * sizeof(float) + enc_a_text_kv_bytes<bf16_t, 192, 64>() + enc_a_param_bytes<bf16_t, 192>()));
  up(lds_bc_chain_bytes<bf16_t, 192, 64, 0>());
  up(lds_bytes<bf16_t, 48, 256>(192));
  up(lds_a_bytes<bf16_t, 256, 32>());
  up(lds_bc_chain_bytes<bf16_t, 256, 32, 2>());
  up(lds_bc_chain_bytes<bf16_t, 384, 16, 1>());
  up(lds_bc_chain_bytes<bf16_t, 384, 16, 0>());
  up(lds_bytes<bf16_t, 48, 256>(384, 256));
  up(lds_bytes<bf16_t, 64, 192>(256, 192));
  up(lds_bytes<bf16_t, 128, 128>(192, 128));
  return m;
}

}  // namespace

bool step_kind_geometry(int kind, int L, int* rows_per_tile, int* level_rows) {
  int rows = 0, lv = 0;
  switch (kind) {
    case PK_CONV_ENC1: case PK_CONV_DEC1: rows = 126; lv = L; break;
    case PK_CONV_ENC2A: case PK_CONV_DEC2: rows = 62; lv = L / 2; break;
    case PK_BC192: rows = 64; lv = L / 2; break;
    case PK_CONV_ENC4: case PK_CONV_DEC3: rows = 46; lv = L / 4; break;
    case PK_A256: case PK_BC256_N2: rows = 32; lv = L / 4; break;
    case PK_BC384_N1: case PK_BC384: rows = 16; lv = L / 8; break;
    default: return false;
  }
  *rows_per_tile = rows;
  *level_rows = lv;
  return true;
}

hipError_t persist_init() {
  if (step_lds_bytes() + 16 > 160 * 1024) return hipErrorInvalidValue;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16);
}

hipError_t launch_step(const StepPlan* d_plan, int grid, hipStream_t st) {
  static const size_t lds = step_lds_bytes();
  hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(512), lds, st, d_plan);
  return hipGetLastError();
}
