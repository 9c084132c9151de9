// bench_enc.cpp — isolated timing + per-stage stamps of the fused EncoderLayer kernels on synthetic data.
// Build: hipcc -c tools/bench_enc.cpp -o tools/bin/bench_enc.o; hipcc tools/bin/bench_enc.o <pkg>/build/enclayer.o -o tools/bin/bench_enc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/dhw_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static void* dev_rand(size_t bytes, bool f32 = false) {
  void* p;
  CK(hipMalloc(&p, bytes));
  if (f32) {
    std::vector<float> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((float)((i * 2654435761u) >> 20 & 0xfff) / 4096.0f - 0.5f) * 0.1f;
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
  } else {
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i)   // bf16 in +-[0.03, 0.06)
      h[i] = (unsigned short)(0x3d00 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0));
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
  }
  return p;
}

int main(int argc, char** argv) {
  const int B = 64, Lt = 30;
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  CK(enclayer_init());
  struct Cfg { int d, heads, Lk; };
  const Cfg cfgs[] = {{384, 6, 61}, {256, 4, 122}, {192, 3, 244}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 64 * 8));
  for (const Cfg& c : cfgs) {
    const int d = c.d, Lk = c.Lk;
    const size_t rows = (size_t)B * Lk + 128;
    const int lpadT = 32, lpadX = ((Lk + 31) / 32) * 32;
    EncLayerParams p{};
    p.B = B; p.Lk = Lk; p.Lt = Lt; p.d = d; p.heads = c.heads;
    p.x = dev_rand(rows * d * 2);
    p.w_q1 = dev_rand((size_t)d * d * 2); p.w_d1 = dev_rand((size_t)d * d * 2); p.w_qkv2 = dev_rand((size_t)3 * d * d * 2);
    p.w_d2 = dev_rand((size_t)d * d * 2); p.w_f1 = dev_rand((size_t)2 * d * d * 2); p.w_f2 = dev_rand((size_t)2 * d * d * 2);
    p.b_q1 = (float*)dev_rand(4 * d * 4, true); p.b_d1 = p.b_q1; p.b_qkv2 = p.b_q1; p.b_d2 = p.b_q1; p.b_f1 = p.b_q1; p.b_f2 = p.b_q1;
    p.pb_q1 = (float*)dev_rand((size_t)(Lk + 128) * d * 4, true);
    p.pb_qk2 = (float*)dev_rand((size_t)(Lk + 128) * 2 * d * 4, true);
    p.film = (float*)dev_rand(1 << 20, true); p.film_bs = 0; p.film_tot = 9280; p.f1 = 0; p.f2 = 384; p.f3 = 768;
    p.k1 = dev_rand((size_t)(B * Lt + 128) * d * 2); p.vt1 = dev_rand((size_t)(B * d + 128) * lpadT * 2); p.lpadT = lpadT;
    p.text = nullptr;
    p.x2 = dev_rand(rows * d * 2); p.qk2 = dev_rand(rows * 3 * d * 2); p.vt2 = dev_rand((size_t)(B * d + 128) * lpadX * 2); p.lpadX = lpadX;
    p.out = dev_rand(rows * d * 2); p.pool = nullptr;
    p.stamps = stamps;
    {  // realistic order: enc_a writes qk2 / vt2 / x2, enc_bc reads them straight after (timed per launch with events)
      p.dbg = 0; p.stamps = stamps;
      CK(hipMemset(stamps, 0, 64 * 8));
      hipEvent_t ev[3];
      for (auto& evt : ev) CK(hipEventCreate(&evt));
      double ta = 0, tb = 0;
      for (int i = 0; i < reps + 3; ++i) {
        CK(hipEventRecord(ev[0], st));
        CK(launch_enclayer(PREC_BF16, p, 0, st));
        CK(hipEventRecord(ev[1], st));
        CK(launch_enclayer(PREC_BF16, p, 1, st));
        CK(hipEventRecord(ev[2], st));
        CK(hipEventSynchronize(ev[2]));
        float a, bms;
        CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&bms, ev[1], ev[2]));
        if (i >= 3) { ta += a; tb += bms; }
      }
      printf("pair  d=%d Lk=%d: enc_a %.2f us, enc_bc %.2f us (alternating launches, event-timed)\n", d, Lk, ta * 1e3 / reps, tb * 1e3 / reps);
      unsigned long long hs[64];
      CK(hipMemcpy(hs, stamps, sizeof hs, hipMemcpyDeviceToHost));
      printf("      enc_bc WG0 in the pair [us]: att blocks (staged, computed):");
      for (int k = 26; k < 32; ++k) printf(" %.2f", hs[k] ? (double)(hs[k] - hs[16]) / 100.0 : -1.0);
      printf(" | stages:");
      for (int k = 17; k <= 24; ++k) printf(" %.2f", (double)(hs[k] - hs[16]) / 100.0);
      printf("\n      enc_a WG0 in the pair [us]:");
      for (int k = 1; k <= 14; ++k) printf(" %.2f", (double)(hs[k] - hs[0]) / 100.0);
      printf("\n");
    }
    for (int dbg = 0; dbg < 2; ++dbg)
    for (int which = dbg; which < 2; ++which) {
      p.dbg = dbg;
      CK(hipMemset(stamps, 0, 64 * 8));
      for (int i = 0; i < 3; ++i) CK(launch_enclayer(PREC_BF16, p, which, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CK(launch_enclayer(PREC_BF16, p, which, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[64];
      CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
      printf("dbg=%d d=%d Lk=%d %s: %.2f us/launch (%d WGs); stage stamps of WG0 [us since start]:", dbg, d, Lk, which ? "enc_bc" : "enc_a ",
             ms * 1e3 / reps, B * ((Lk + 63) / 64));
      const int s0 = which ? 16 : 0, s1 = which ? 24 : 14;
      for (int k = s0; k <= s1; ++k) printf(" %.2f", h[k] ? (double)(h[k] - h[s0]) / 100.0 : -1.0);
      if (which) printf("  [shader clock %.2f GHz]", (double)(h[41] - h[40]) / ((double)(h[24] - h[16]) * 10.0));
      printf("\n");
    }
  }
  return 0;
}
