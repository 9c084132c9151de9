// bench_pipe_overlap.cpp — which per-CU pipes overlap ACROSS waves?  One 512-thread workgroup per CU (2 waves per SIMD, as the
// fused kernels run).  Waves 0-3 run work A, waves 4-7 run work B (each SIMD hosts one wave of each kind), for every pair of
//   V = VMEM  (24 x global_load_dwordx4 of an L2-resident buffer per wave, results consumed)
//   A = VALU  (768 independent-chain v_fma_f32)
//   M = MFMA  (96 x v_mfma_f32_16x16x32_bf16, 4 accumulators)
//   L = LDS   (96 x ds_read_b128, conflict-free)
// Reported: time of A alone (other half idle), B alone, and both together; "overlap" = (A + B - both) / min(A, B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ float work(const uint4* gbuf, const char* lds, int lane, int wave, int r) {
  float out = 0.f;
  if constexpr (KIND == 0) {   // VMEM
    const uint4* p = gbuf + ((size_t)(r * 8 + wave) * 24 * 64) % (1 << 16) + lane;
    uint4 v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = p[i * 64];
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) x ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    out = (float)x;
  } else if constexpr (KIND == 1) {   // VALU
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = lane * 0.001f + k + r;
#pragma unroll
    for (int i = 0; i < 96; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
#pragma unroll
    for (int k = 0; k < 8; ++k) out += a[k];
  } else if constexpr (KIND == 2) {   // MFMA
    bf16x8 x, y;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(lane * 0.01f + k); y[k] = (__bf16)(r * 0.5f + k); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 24; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[k], 0, 0, 0);
    out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else {   // LDS
    const char* p = lds + (lane & 15) * 416 + (lane >> 4) * 16 + wave * 64;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 96; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + (i & 7) * 16 * 416 + (i >> 3) * 512 % 256);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    out = (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
  }
  return out;
}

// MODE 0: waves 0-3 run KA, waves 4-7 idle; 1: waves 4-7 run KB, 0-3 idle; 2: both
template <int KA, int KB, int MODE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ buf, float* sink, int reps, unsigned long long* t) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    if (wave < 4) { if (MODE != 1) s += work<KA>(buf, lds, lane, wave, r); }
    else { if (MODE != 0) s += work<KB>(buf, lds, lane, wave, r); }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int KA, int KB, int MODE>
double run(const uint4* buf, float* sink, unsigned long long* t) {
  const int reps = 200, wgs = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KA, KB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<KA, KB, MODE>), dim3(wgs), dim3(512), 64 * 1024, 0, buf, sink, reps, t);
  CK(hipDeviceSynchronize());
  unsigned long long h[256]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < wgs; ++i) s += h[i];
  return s / wgs / 100.0 / reps;
}
template <int KA, int KB>
void pair(const uint4* buf, float* sink, unsigned long long* t) {
  const char* names[] = {"VMEM", "VALU", "MFMA", "LDS "};
  const double a = run<KA, KB, 0>(buf, sink, t), b = run<KA, KB, 1>(buf, sink, t), ab = run<KA, KB, 2>(buf, sink, t);
  printf("%s (waves 0-3) | %s (waves 4-7): alone %.3f / %.3f us, together %.3f us, overlap %.0f %% of the shorter\n", names[KA], names[KB], a, b, ab,
         100.0 * (a + b - ab) / (a < b ? a : b));
}
int main() {
  uint4* buf; float* sink; unsigned long long* t;
  CK(hipMalloc(&buf, 4 << 20)); CK(hipMemset(buf, 1, 4 << 20)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&t, 256 * 8));
  pair<0, 0>(buf, sink, t); pair<0, 1>(buf, sink, t); pair<0, 2>(buf, sink, t); pair<0, 3>(buf, sink, t);
  pair<1, 1>(buf, sink, t); pair<1, 2>(buf, sink, t); pair<1, 3>(buf, sink, t);
  pair<2, 2>(buf, sink, t); pair<2, 3>(buf, sink, t); pair<3, 3>(buf, sink, t);
  return 0;
}
