// bench_handoff.cpp — does a tile written by workgroup i of one kernel come from L2 when workgroup i of the NEXT kernel
// reads it (same XCD under the round-robin dispatch), or is L2 invalidated at the kernel boundary?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(512) void produce(uint4* buf, int pieces, unsigned tag) {
  uint4* t = buf + (size_t)blockIdx.x * pieces;
  for (int i = threadIdx.x; i < pieces; i += 512) t[i] = make_uint4(tag, i, blockIdx.x, 7);
}
__global__ __launch_bounds__(512) void consume(const uint4* buf, int pieces, int shift, uint4* sink, unsigned long long* stamps) {
  const int src = (blockIdx.x + shift) % gridDim.x;
  const uint4* t = buf + (size_t)src * pieces;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = threadIdx.x; i + 5 * 512 < pieces + 5 * 512; i += 6 * 512) {
    uint4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = i + u * 512 < pieces ? t[i + u * 512] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 6; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
  if (acc.x == 0x7654321u) sink[threadIdx.x] = acc;
}
int main() {
  const int wgs = 256, pieces = 98 * 1024 / 16;
  uint4 *buf, *sink; unsigned long long* stamps;
  CK(hipMalloc(&buf, (size_t)wgs * pieces * 16)); CK(hipMalloc(&sink, 1 << 16)); CK(hipMalloc(&stamps, wgs * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int shift : {0, 8, 1, 3, 0, 1}) {
    double tot = 0, inner = 0;
    for (int rep = 0; rep < 20; ++rep) {
      hipLaunchKernelGGL(produce, dim3(wgs), dim3(512), 0, 0, buf, pieces, (unsigned)rep);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(consume, dim3(wgs), dim3(512), 0, 0, buf, pieces, shift, sink, stamps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
      unsigned long long h[256]; CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
      double s = 0; for (int i = 0; i < wgs; ++i) s += h[i]; inner += s / wgs / 100.0;
    }
    printf("consumer reads tile (i + %d) %% 256: kernel %.2f us, mean in-kernel read time per workgroup %.2f us (98 KB each)\n", shift, tot * 1e3 / 20, inner / 20);
  }
  return 0;
}
