// bench_stream_depth.cpp — per-CU L2 -> VGPR streaming rate of a shared 288 KB / 1.5 MB buffer as a function of the loads
// in flight per lane (D) and the waves per CU: is the ~150 GB/s per CU that the fused kernels' weight rings reach a
// bandwidth limit (64 B/clk) or a concurrency limit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int D, int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const uint4* __restrict__ buf, int pieces, int passes, uint4* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int p = 0; p < passes; ++p)
    for (int i = threadIdx.x; i + (D - 1) * NT < pieces; i += D * NT) {
      uint4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = buf[i + d * NT];
#pragma unroll
      for (int d = 0; d < D; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
  if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int D, int NT>
void run(const uint4* buf, uint4* sink, int kb, int wgs, hipEvent_t e0, hipEvent_t e1) {
  const int pieces = kb * 1024 / 16, passes = 32;
  auto go = [&]() { hipLaunchKernelGGL((stream_kernel<D, NT>), dim3(wgs), dim3(NT), 0, 0, buf, pieces, passes, sink); };
  go(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int per_pass = pieces / (D * NT) * (D * NT);
  const double bytes = (double)passes * per_pass * 16;
  printf("%5d KB, %3d WGs x %4d thr, D=%2d (%3d KB in flight/CU): %7.1f GB/s per WG\n", kb, wgs, NT, D, D * NT * 16 / 1024, bytes / ms / 1e6);
}

int main() {
  uint4 *buf, *sink;
  CK(hipMalloc(&buf, 8 << 20)); CK(hipMalloc(&sink, 1 << 16));
  CK(hipMemset(buf, 1, 8 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kb : {288, 1536})
    for (int wgs : {1, 256}) {
#define R(D, NT) run<D, NT>(buf, sink, kb, wgs, e0, e1)
      R(4, 256); R(8, 256); R(16, 256); R(32, 256);
      R(4, 512); R(8, 512); R(16, 512); R(24, 512); R(32, 512);
      R(4, 1024); R(8, 1024); R(16, 1024);
    }
  return 0;
}
