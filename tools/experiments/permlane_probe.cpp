// permlane_probe.cpp — prints what v_permlane16_swap_b32 does to two registers holding (reg id, lane id)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = 0x100 + threadIdx.x, b = 0x200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, 1, 64, 0, 0, d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) { printf("r[%d]:", r); for (int i = 0; i < 64; i += 4) printf(" %03x", h[r * 64 + i]); printf("\n"); }
  return 0;
}
