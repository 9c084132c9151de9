// bench_lat.cpp — calibrate: shader clock and dependent-load latency (L2 / MALL / HBM), idle and with 64 busy CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void chase(const unsigned* buf, int steps, unsigned long long* out, unsigned n) {
  unsigned idx = (unsigned)(((unsigned long long)(blockIdx.x * 7919u + threadIdx.x) * 32ull) % n);   // always in bounds
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; ++i) idx = buf[idx];
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = c1 - c0; out[blockIdx.x * 3 + 2] = idx; }
}
// 16-byte-per-lane streaming loads of a shared region, N loads in flight, repeated: measures per-CU streaming rate
__global__ void stream(const uint4* buf, int iters, int n_uint4, unsigned long long* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned off = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint4 v = buf[(off + u * blockDim.x) % n_uint4];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    off += 8 * blockDim.x;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 2] = acc.x ^ acc.y ^ acc.z ^ acc.w; }
}

int main() {
  unsigned long long* out;
  CK(hipMalloc(&out, 4096 * 24));
  std::vector<unsigned long long> h(4096 * 3);
  for (size_t bytes : {(size_t)256 << 10, (size_t)16 << 20, (size_t)1 << 30}) {
    const size_t n = bytes / 4;
    std::vector<unsigned> hb(n);
    // random cyclic permutation with large strides (cache-line granular)
    const size_t lines = n / 32;
    std::vector<unsigned> perm(lines);
    for (size_t i = 0; i < lines; ++i) perm[i] = (unsigned)i;
    unsigned long long s = 88172645463325252ull;
    for (size_t i = lines - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (size_t i = 0; i < lines; ++i) for (int k = 0; k < 32; ++k) hb[(size_t)perm[i] * 32 + k] = perm[(i + 1) % lines] * 32 + k;
    unsigned* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemcpy(buf, hb.data(), bytes, hipMemcpyHostToDevice));
    for (int blocks : {1, 64, 256}) {
      const int steps = 2000;
      hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, steps, out, (unsigned)n);   // warm
      hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, steps, out, (unsigned)n);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, blocks * 24, hipMemcpyDeviceToHost));
      printf("chase %8zu KB, %3d waves: %.1f ns/load, %.0f shader cycles/load, clock %.2f GHz\n", bytes >> 10, blocks,
             h[0] * 10.0 / steps, (double)h[1] / steps, (double)h[1] / (h[0] * 10.0));
    }
    CK(hipFree(buf));
  }
  // streaming a 300 KB region (like one GEMM stage's weights) by every workgroup simultaneously
  const int n_uint4 = 300 * 1024 / 16;
  uint4* sb;
  CK(hipMalloc(&sb, n_uint4 * 16));
  CK(hipMemset(sb, 1, n_uint4 * 16));
  for (int blocks : {1, 64, 256}) for (int threads : {256, 512, 1024}) {
    const int iters = n_uint4 / (8 * threads) * 4;   // 4 passes
    hipLaunchKernelGGL(stream, dim3(blocks), dim3(threads), 0, 0, sb, iters, n_uint4, out);
    hipLaunchKernelGGL(stream, dim3(blocks), dim3(threads), 0, 0, sb, iters, n_uint4, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, blocks * 24, hipMemcpyDeviceToHost));
    const double bytes = (double)iters * 8 * threads * 16;
    printf("stream 300KB x4, %3d WGs x %4d thr: %.2f us, %.1f GB/s per WG\n", blocks, threads, h[0] / 100.0, bytes / (h[0] * 10.0));
  }
  return 0;
}
