// attn.hip — multi-head attention core softmax(QK^T/sqrt(D) + mask·(-1e9))V
// (reference attention.py:26-46, called from attention.py:78-85) for the
// stroke<->text cross attention, the stroke self attention and the text->style
// attention.  One wave = 16 queries of one (sample, head); no LDS, no barrier:
//
//   S^T = K · Q^T      MFMA A-operand = K rows  (16 keys  x D, straight from L2)
//                      MFMA B-operand = Q rows  (16 queries x D, held in VGPRs)
//        -> lane (query = lane&15) holds keys 4g..4g+3 of each 16-key tile:
//           softmax statistics are per lane + two cross-group shuffles.
//   O^T = V^T · P^T    B-operand = P^T taken from the S^T accumulators IN PLACE
//                      (cdna_hip_programming.md §3 "accumulator tile as the next
//                      MFMA's operand": the k-slot order is (tile0 keys 4g..4g+3,
//                      tile1 keys 16+4g..)); A-operand = V^T read with the SAME
//                      slot order from the key-contiguous Vt buffer the
//                      projection GEMM wrote.
//   Online softmax over 32-key blocks (fp32 statistics), exact for any Lk.
#include "attn_core.h"
#include "dhw_kernels.h"

namespace {

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
  constexpr int DT = D / 16, KCH = (D + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 64 + wave * 16;
  if (q0 >= p.Lq) return;   // wave-uniform

  const T* Q = reinterpret_cast<const T*>(p.Q);
  // Q fragments (B operand): query q0+l15, d = 32c + 8g .. +8; d >= D is zero (D = 48)
  Frag<T> qf[KCH];
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int d = 32 * c + 8 * g;
    qf[c] = d < D ? frag_load(Q + (size_t)(b * p.Lq + q0 + l15) * p.ldq + h * D + d) : frag_zero<T>();
  }
  f32x4 o[DT];
  attn_wave16_auto<T, D>(qf, reinterpret_cast<const T*>(p.K) + (size_t)(b * p.Lk + l15) * p.ldk + p.koff + h * D, p.ldk,
                    reinterpret_cast<const T*>(p.Vt) + ((size_t)(b * p.H + h) * D + l15) * p.lpad + 4 * g, p.lpad,
                    p.text ? p.text + (size_t)b * p.ldt : nullptr, p.Lk, o);
  if (q0 + l15 < p.Lq) {
    T* out = reinterpret_cast<T*>(p.out) + (size_t)(b * p.Lq + q0 + l15) * p.ldo + h * D + 4 * g;
#pragma unroll
    for (int t = 0; t < DT; ++t) store4(out + 16 * t, o[t]);
  }
}

template <typename T>
hipError_t launch_t(const AttnParams& p, hipStream_t st) {
  dim3 grid((p.Lq + 63) / 64, p.H, p.B);
  if (p.D == 64) hipLaunchKernelGGL((attn_kernel<T, 64>), grid, dim3(256), 0, st, p);
  else if (p.D == 48) hipLaunchKernelGGL((attn_kernel<T, 48>), grid, dim3(256), 0, st, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace

hipError_t launch_attn(int prec, const AttnParams& p, hipStream_t st) {
  if (p.lpad % 32 || p.lpad < ((p.Lk + 31) / 32) * 32) return hipErrorInvalidValue;
  return prec == PREC_BF16 ? launch_t<bf16_t>(p, st) : launch_t<float>(p, st);
}
