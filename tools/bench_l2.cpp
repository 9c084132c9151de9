// bench_l2.cpp — what the L2 -> CU path delivers when every CU streams the SAME small buffer (the weight stream of the
// fused kernels) at once: workgroups of 512 threads, 16 bytes per lane, `depth` independent loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int D>
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ buf, int pieces, int passes, int per_wg_offset, uint4* sink) {
  const uint4* base = buf + (size_t)blockIdx.x * per_wg_offset;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int p = 0; p < passes; ++p)
    for (int i = threadIdx.x; i + (D - 1) * 512 < pieces; i += D * 512) {
      uint4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = base[i + d * 512];
#pragma unroll
      for (int d = 0; d < D; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
  if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
  const size_t total = 256u << 20;
  uint4 *buf, *sink;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 1 << 16));
  CK(hipMemset(buf, 1, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int kb_sizes[] = {288, 1152, 2880};
  for (int kb : kb_sizes)
    for (int wgs : {1, 32, 64, 128, 256})
      for (int distinct = 0; distinct < 2; ++distinct) {
        const int pieces = kb * 1024 / 16, passes = 64;
        if (distinct && (size_t)wgs * kb * 1024 > total) continue;
        auto go = [&]() { hipLaunchKernelGGL(stream_kernel<8>, dim3(wgs), dim3(512), 0, 0, buf, pieces, passes, distinct ? pieces : 0, sink); };
        go(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)wgs * passes * kb * 1024;
        printf("%5d KB per WG, %3d WGs, %s: %8.1f GB/s per WG, %7.2f TB/s aggregate\n", kb, wgs, distinct ? "distinct buffers" : "same buffer    ",
               bytes / wgs / ms / 1e6, bytes / ms / 1e9);
      }
  return 0;
}
