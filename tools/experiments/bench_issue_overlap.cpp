// bench_issue_overlap.cpp — does a wave's VALU work overlap with its own / its SIMD partner's vector-memory instruction issue?
// One 512-thread workgroup per CU.  Each wave requests NL x 1 KiB (global_load_dwordx4, L2-resident buffer shared by all
// workgroups, like a weight stream) and executes NV dependent-free v_fma (8 independent chains).  Orders:
//   0 loads then VALU        1 VALU then loads        2 interleaved (1 load per NV/NL VALU)
//   3 waves 0-3 loads first, waves 4-7 VALU first (partners on a SIMD in opposite phases)
//   4 loads only             5 VALU only
//   6 as 3, VALU phases at s_setprio 3      7 as 0 (same order in every wave), VALU phases at s_setprio 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NL> __device__ __forceinline__ void loads(const uint4* p, uint4 (&v)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = p[i * 64];
}
template <int NV> __device__ __forceinline__ void valu(float (&a)[8]) {
#pragma unroll
  for (int i = 0; i < NV / 8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
}

template <int NL, int NV, int ORDER>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ buf, float* sink, int reps, unsigned long long* t) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  uint4 acc = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    const uint4* p = buf + ((size_t)(r * 8 + wave) * NL * 64) % (1 << 16) + lane;   // 1 MiB window, fragment-order like
    uint4 v[NL];
    if constexpr (ORDER == 0) { loads<NL>(p, v); __builtin_amdgcn_sched_barrier(0); valu<NV>(a); }
    else if constexpr (ORDER == 1) { valu<NV>(a); __builtin_amdgcn_sched_barrier(0); loads<NL>(p, v); }
    else if constexpr (ORDER == 2) {
#pragma unroll
      for (int i = 0; i < NL; ++i) { v[i] = p[i * 64]; __builtin_amdgcn_sched_barrier(0); valu<NV / NL / 8 * 8>(a); __builtin_amdgcn_sched_barrier(0); }
    } else if constexpr (ORDER == 3) {
      if (wave < 4) { loads<NL>(p, v); __builtin_amdgcn_sched_barrier(0); valu<NV>(a); }
      else { valu<NV>(a); __builtin_amdgcn_sched_barrier(0); loads<NL>(p, v); }
    } else if constexpr (ORDER == 6) {
      if (wave < 4) { loads<NL>(p, v); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); valu<NV>(a); __builtin_amdgcn_s_setprio(0); }
      else { __builtin_amdgcn_s_setprio(3); valu<NV>(a); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); loads<NL>(p, v); }
    } else if constexpr (ORDER == 7) {
      loads<NL>(p, v); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); valu<NV>(a); __builtin_amdgcn_s_setprio(0);
    } else if constexpr (ORDER == 4) { loads<NL>(p, v); }
    else { valu<NV>(a); for (int i = 0; i < NL; ++i) v[i] = make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int i = 0; i < NL; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f || acc.x == 0x12345u) sink[threadIdx.x] = s + acc.y;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int NL, int NV, int ORDER>
void run(const uint4* buf, float* sink, unsigned long long* t, const char* name) {
  const int reps = 200, wgs = 256;
  hipLaunchKernelGGL((k<NL, NV, ORDER>), dim3(wgs), dim3(512), 0, 0, buf, sink, reps, t);
  hipLaunchKernelGGL((k<NL, NV, ORDER>), dim3(wgs), dim3(512), 0, 0, buf, sink, reps, t);
  CK(hipDeviceSynchronize());
  unsigned long long h[256]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < wgs; ++i) s += h[i];
  printf("NL=%2d NV=%4d %-28s: %.3f us per phase\n", NL, NV, name, s / wgs / 100.0 / reps);
}

int main() {
  uint4* buf; float* sink; unsigned long long* t;
  CK(hipMalloc(&buf, 4 << 20)); CK(hipMemset(buf, 1, 4 << 20)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&t, 256 * 8));
#define ALL(NL, NV) run<NL, NV, 4>(buf, sink, t, "loads only"); run<NL, NV, 5>(buf, sink, t, "VALU only"); run<NL, NV, 0>(buf, sink, t, "loads then VALU"); \
  run<NL, NV, 1>(buf, sink, t, "VALU then loads"); run<NL, NV, 2>(buf, sink, t, "interleaved"); run<NL, NV, 3>(buf, sink, t, "partners in opposite order"); \
  run<NL, NV, 6>(buf, sink, t, "opposite order + VALU prio 3"); run<NL, NV, 7>(buf, sink, t, "loads then VALU at prio 3");
  ALL(12, 192) ALL(12, 384) ALL(24, 384) ALL(24, 768) ALL(6, 192)
  return 0;
}
