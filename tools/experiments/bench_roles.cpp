// bench_roles.cpp — would wave SPECIALISATION pay for the 16-row attention-level tiles?  A chain of NS "stages", each = a
// [16 x 384] x [384 x 384] bf16 GEMM (weights streamed from an L2-resident buffer in fragment order, 36 KB per wave and stage with
// 8 waves) followed by W units of vector work (the epilogue / LayerNorm / attention share of a stage), one workgroup per CU:
//   S  symmetric (production form): 8 waves, each streams its own 36 fragments (ring of 24 in registers, the rest re-loaded inside
//      the main loop), then does its share of the vector work, barrier.
//   R  roles: 12 waves (3 per SIMD, 168 VGPRs): waves 0-7 only stream weights (whole stage, 36 fragments, in registers) and issue the
//      MFMAs; waves 8-11 do ALL the vector work of the stage while the GEMM waves' loads for the next stage are in flight.
// Reports us per stage for W = 0, 1, 2, 3 (units of ~0.55 us of single-SIMD VALU time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float valu_work(float seed, int units) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = seed + k;
  for (int u = 0; u < units; ++u)
#pragma unroll
    for (int i = 0; i < 36; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);   // 288 VALU per unit and wave
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  return s;
}

// symmetric: RING fragments prefetched before the vector work, the remaining 36 - RING streamed inside the main loop
template <int RING>
__global__ __launch_bounds__(512) void sym_kernel(const uint4* __restrict__ w, int ns, int units, float* sink, unsigned long long* t) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 800];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 800 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float vs = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint4 q[RING];
  const uint4* base = w + (size_t)wave * 36 * 64 + lane;
#pragma unroll
  for (int f = 0; f < RING; ++f) q[f] = base[f * 64];
  for (int s = 0; s < ns; ++s) {
    const uint4* cur = w + ((size_t)(s % 10) * 8 + wave) * 36 * 64 + lane, *nxt = w + ((size_t)((s + 1) % 10) * 8 + wave) * 36 * 64 + lane;
    vs += valu_work(vs + lane, units);          // the stage's vector work (epilogue of the previous GEMM, LN, attention ...)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < 12; ++k) {               // 12 k-chunks x 3 channel tiles
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + (lane & 15) * 800 + k * 64 + (lane >> 4) * 16);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int f = k * 3 + i;
        const uint4 b = q[f % RING];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), a, acc[i], 0, 0, 0);
        if (f + RING < 36) q[f % RING] = cur[(f + RING) * 64];          // re-load inside the main loop
        else q[f % RING] = nxt[(f + RING - 36) * 64];                    // next stage's first RING fragments
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float r = vs;
  for (int i = 0; i < 3; ++i) r += acc[i][0] + acc[i][3];
  for (int f = 0; f < RING; ++f) r += (float)q[f].x;
  if (r == 1.2345f) sink[threadIdx.x] = r;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// roles: waves 0-7 GEMM (whole-stage ring), waves 8-11 vector work
__global__ __launch_bounds__(768) void role_kernel(const uint4* __restrict__ w, int ns, int units, float* sink, unsigned long long* t) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 800];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 800 / 4; i += 768) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float vs = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (wave < 8) {
    uint4 q[36];
    const uint4* base = w + (size_t)wave * 36 * 64 + lane;
#pragma unroll
    for (int f = 0; f < 36; ++f) q[f] = base[f * 64];
    for (int s = 0; s < ns; ++s) {
      const uint4* nxt = w + ((size_t)((s + 1) % 10) * 8 + wave) * 36 * 64 + lane;
      __builtin_amdgcn_s_barrier();              // operands of this stage ready (vector waves done)
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + (lane & 15) * 800 + k * 64 + (lane >> 4) * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q[k * 3 + i]), a, acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();              // accumulators handed over
#pragma unroll
      for (int f = 0; f < 36; ++f) q[f] = nxt[f * 64];   // the NEXT stage's whole weight slice: streams while the vector waves work
    }
    float r = 0.f;
    for (int f = 0; f < 36; ++f) r += (float)q[f].x;
    vs = r;
  } else {
    for (int s = 0; s < ns; ++s) {
      vs += valu_work(vs + lane, 2 * units);     // 4 waves do the vector work of 8
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float r = vs;
  for (int i = 0; i < 3; ++i) r += acc[i][0] + acc[i][3];
  if (r == 1.2345f) sink[threadIdx.x] = r;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// symmetric with 12 waves: 2 channel tiles per wave, the WHOLE stage (24 fragments per wave) prefetched before the vector work
__global__ __launch_bounds__(768) void sym12_kernel(const uint4* __restrict__ w, int ns, int units, float* sink, unsigned long long* t) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 800];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 800 / 4; i += 768) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float vs = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint4 q[24];
  const uint4* base = w + (size_t)wave * 24 * 64 + lane;
#pragma unroll
  for (int f = 0; f < 24; ++f) q[f] = base[f * 64];
  for (int s = 0; s < ns; ++s) {
    const uint4* nxt = w + ((size_t)((s + 1) % 10) * 12 + wave) * 24 * 64 + lane;
    vs += valu_work(vs + lane, units) * (8.0f / 12.0f);     // (the same total vector work spread over 12 waves: 2/3 of a unit each)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + (lane & 15) * 800 + k * 64 + (lane >> 4) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q[k * 2 + i]), a, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 24; ++f) q[f] = nxt[f * 64];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float r = vs;
  for (int i = 0; i < 2; ++i) r += acc[i][0] + acc[i][3];
  for (int f = 0; f < 24; ++f) r += (float)q[f].x;
  if (r == 1.2345f) sink[threadIdx.x] = r;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

int main() {
  uint4* w; float* sink; unsigned long long* t;
  const size_t wbytes = (size_t)10 * 8 * 36 * 1024;   // 10 stages x 288 KB
  CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0x3c, wbytes)); CK(hipMalloc(&sink, 8192)); CK(hipMalloc(&t, 256 * 8));
  const int ns = 200, wgs = 256;
  auto report = [&](const char* name, int units) {
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < wgs; ++i) s += h[i];
    printf("%-44s W=%d: %.3f us per stage\n", name, units, s / wgs / 100.0 / ns);
  };
  for (int units = 0; units <= 3; ++units) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(sym_kernel<24>, dim3(wgs), dim3(512), 0, 0, w, ns, units, sink, t);
    report("symmetric, 8 waves, ring 24 (production)", units);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(role_kernel, dim3(wgs), dim3(768), 0, 0, w, ns, units, sink, t);
    report("roles: 8 GEMM waves + 4 vector waves", units);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(sym12_kernel, dim3(wgs), dim3(768), 0, 0, w, ns, units * 12 / 12, sink, t);
    report("symmetric, 12 waves, whole-stage ring", units);
  }
  return 0;
}
