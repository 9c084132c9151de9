// bench_stage.cpp — one 384x384 GEMM stage of the fused EncoderLayer kernels in isolation (64 rows per workgroup, 8 waves,
// weights streamed from L2), with ablations, to see what bounds a stage.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/gemm_core.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int DM = 384, BM = 64;
template <int WN, int RING, int ABL, int REPS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void stage_kernel(const bf16_t* x, const bf16_t* w, bf16_t* out, unsigned long long* stamps) {
  constexpr int WM = 8 / WN, MT = BM / WM / 16, NT = DM / WN / 16, KC = DM / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int S = tile_stride<bf16_t>(DM);
  for (int id = tid; id < BM * DM * 2 / 16; id += 512) {
    const int r = id / (DM * 2 / 16), cc = id % (DM * 2 / 16);
    *reinterpret_cast<uint4*>(smem + r * S + cc * 16) = *reinterpret_cast<const uint4*>((const char*)x + ((size_t)(blockIdx.x * BM + r) * DM) * 2 + cc * 16);
  }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc[NT][MT];
  acc_zero(acc);
  const int row0 = wm * (BM / WM), ntile0 = wn * NT;
  for (int rep = 0; rep < REPS; ++rep) {
    mainloop<bf16_t, MT, NT, RING, ABL>(acc, w + ((size_t)(rep % 4) * DM * DM) + ((size_t)ntile0 * KC * 64 + lane) * 8,
                                        smem + (row0 + l15) * S + g * 8 * 2, S, KC, 1);
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && tid == 0) { stamps[0] = t1 - t0; }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
      store4(out + (size_t)(blockIdx.x * BM + row0 + j * 16 + l15) * DM + (ntile0 + i) * 16 + 4 * g, acc[i][j]);
}

template <int WN, int RING, int ABL>
void run(const char* name, const bf16_t* x, const bf16_t* w, bf16_t* out, unsigned long long* stamps, int wgs) {
  constexpr int REPS = 8;
  auto k = stage_kernel<WN, RING, ABL, REPS>;
  const int lds = BM * tile_stride<bf16_t>(DM);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, x, w, out, stamps);
  CK(hipDeviceSynchronize());
  unsigned long long h = 0;
  CK(hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost));
  printf("%-44s WGs=%3d: %.2f us per 384x384 stage (WG0)\n", name, wgs, h / 100.0 / REPS);
}

int main() {
  bf16_t *x, *w, *out; unsigned long long* stamps;
  const size_t nx = (size_t)256 * BM * DM, nw = (size_t)4 * DM * DM;
  CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&out, nx * 2)); CK(hipMalloc(&stamps, 64));
  std::vector<unsigned short> h(nx > nw ? nx : nw);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3d00 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0));
  CK(hipMemcpy(x, h.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice));
  for (int wgs : {1, 64, 256}) {
    run<8, 24, 0>("WN=8 ring24 full", x, w, out, stamps, wgs);
    run<8, 24, 1>("WN=8 ring24 no-MFMA", x, w, out, stamps, wgs);
    run<8, 24, 2>("WN=8 ring24 no-weight-reload", x, w, out, stamps, wgs);
    run<8, 24, 4>("WN=8 ring24 no-LDS", x, w, out, stamps, wgs);
    run<8, 24, 6>("WN=8 ring24 MFMA only", x, w, out, stamps, wgs);
    run<8, 12, 0>("WN=8 ring12 full", x, w, out, stamps, wgs);
    run<8, 36, 0>("WN=8 ring36 full", x, w, out, stamps, wgs);
    run<4, 24, 0>("WN=4 ring24 full", x, w, out, stamps, wgs);
  }
  return 0;
}
