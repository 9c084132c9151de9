// bench_copy.cpp — data-movement floor of a text-plane GEMM: read an [M x 384] bf16 matrix in row tiles, write one back
// (through registers only), with the tile sizes / grids the GEMM kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int U>
__global__ __launch_bounds__(512) void copy_tiles(const uint4* __restrict__ in, uint4* __restrict__ out, long pieces_per_tile, long total_pieces, int write) {
  const long base = (long)blockIdx.x * pieces_per_tile;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (long i = threadIdx.x; i < pieces_per_tile; i += 512 * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long id = base + i + u * 512; v[u] = (i + u * 512 < pieces_per_tile && id < total_pieces) ? in[id] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long id = base + i + u * 512;
      if (i + u * 512 < pieces_per_tile && id < total_pieces) { if (write) out[id] = v[u]; else { acc.x ^= v[u].x; acc.y ^= v[u].y; } }
    }
  }
  if (!write && acc.x == 0x1234567u) out[threadIdx.x] = acc;
}
int main() {
  const long M = 115200, K = 384;
  const long bytes = M * K * 2, pieces = bytes / 16;
  uint4 *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rows : {32, 64, 128, 512})
    for (int write = 0; write < 2; ++write) {
      const long ppt = (long)rows * K * 2 / 16;
      const int grid = (int)((pieces + ppt - 1) / ppt);
      auto go = [&]() { hipLaunchKernelGGL(copy_tiles<6>, dim3(grid), dim3(512), 0, 0, in, out, ppt, pieces, write); };
      for (int i = 0; i < 3; ++i) go();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / 20;
      printf("%4d-row tiles (%5d WGs), %s: %7.1f us  %6.2f TB/s\n", rows, grid, write ? "read + write" : "read only   ", us, (write ? 2 : 1) * bytes / us / 1e6);
    }
  return 0;
}
