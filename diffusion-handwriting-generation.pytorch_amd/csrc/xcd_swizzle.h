// xcd_swizzle.h — shared by the kernels (dhw_common.h) and the host library (test hook dhw_debug_xcd_swizzle)
#pragma once
#include <hip/hip_runtime.h>

// XCD-aware workgroup id: the dispatcher deals consecutive blockIdx round-robin over the 8 XCDs (private L2 each), so
// the workgroups that share data (the row tiles of one sample read the same K/V) would each miss in a different L2.
// This bijection hands every XCD a contiguous range of logical ids instead (any grid size).
__host__ __device__ inline int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
