// bench_encw.cpp — per-wave timeline of one enc_bc workgroup (diagnostic build of enclayer.hip with -DDHW_STAMPS): shader-clock
// stamps of all 8 waves of workgroup 0 at the phase boundaries inside the stages (csrc/enclayer.hip, WST slots).
// usage: bench_encw [d=384] [Lk=61] [a]   (a: the first half, enc_a, instead of enc_bc)
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
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3d00 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0));
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
  }
  return p;
}
int main(int argc, char** argv) {
  const int B = 64, Lt = 30;
  const int d = argc > 1 ? atoi(argv[1]) : 384, Lk = argc > 2 ? atoi(argv[2]) : 61, heads = d / 64;
  CK(enclayer_init());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int NS = 64 + 8 * 32;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, NS * 8));
  const size_t rows = (size_t)B * Lk + 128;
  const int lpadT = 32, lpadX = ((Lk + 31) / 32) * 32;
  EncLayerParams p{};
  p.B = B; p.Lk = Lk; p.Lt = Lt; p.d = d; p.heads = heads;
  p.x = dev_rand(rows * d * 2);
  p.w_q1 = dev_rand((size_t)d * d * 2); p.w_d1 = dev_rand((size_t)d * d * 2); p.w_qkv2 = dev_rand((size_t)3 * d * d * 2);
  p.w_d2 = dev_rand((size_t)d * d * 2); p.w_f1 = dev_rand((size_t)2 * d * d * 2); p.w_f2 = dev_rand((size_t)2 * d * d * 2);
  p.b_q1 = (float*)dev_rand(4 * d * 4, true); p.b_d1 = p.b_q1; p.b_qkv2 = p.b_q1; p.b_d2 = p.b_q1; p.b_f1 = p.b_q1; p.b_f2 = p.b_q1;
  p.pb_q1 = (float*)dev_rand((size_t)(Lk + 128) * d * 4, true);
  p.pb_qk2 = (float*)dev_rand((size_t)(Lk + 128) * 2 * d * 4, true);
  p.film = (float*)dev_rand(1 << 20, true); p.film_bs = 0; p.film_tot = 9280; p.f1 = 0; p.f2 = 384; p.f3 = 768;
  p.k1 = dev_rand((size_t)(B * Lt + 128) * d * 2); p.vt1 = dev_rand((size_t)(B * d + 128) * lpadT * 2); p.lpadT = lpadT;
  p.x2 = dev_rand(rows * d * 2); p.qk2 = dev_rand(rows * 3 * d * 2); p.vt2 = dev_rand((size_t)(B * d + 128) * lpadX * 2); p.lpadX = lpadX;
  p.out = dev_rand(rows * d * 2);
  p.stamps = stamps;
  for (int it = 0; it < 5; ++it) {
    CK(hipMemset(stamps, 0, NS * 8));
    CK(launch_enclayer(PREC_BF16, p, 0, st));
    CK(launch_enclayer(PREC_BF16, p, 1, st));
    CK(hipStreamSynchronize(st));
  }
  std::vector<unsigned long long> h(NS);
  if (argc > 3 && argv[3][0] == 'a') {
    // the first half alone (enc_a_core.h, ENC_STAMP slots per wave: p.dbg bit 2)
    CK(hipMemset(stamps, 0, NS * 8));
    p.dbg = 4;
    CK(launch_enclayer(PREC_BF16, p, 0, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), stamps, NS * 8, hipMemcpyDeviceToHost));
    const int order[15] = {0, 1, 8, 9, 2, 3, 10, 11, 4, 12, 5, 13, 6, 14, 7};
    const char* an[15] = {"start", "staged", "q1.run", "q1.epi", "bar", "xatt+bar", "d1.run", "ln.epi", "x2st+bar", "q.run", "q.end", "k.run", "k.end", "v.run", "v.end"};
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 8; ++w) if (h[64 + w * 32] && h[64 + w * 32] < t0) t0 = h[64 + w * 32];
    printf("enc_a d=%d Lk=%d: per-wave stamps of workgroup 0 [cycles since the first wave's start / 1000]\n%-10s", d, Lk, "slot");
    for (int w = 0; w < 8; ++w) printf("   w%d  ", w);
    printf("  max-min  d(max)\n");
    double prevmax = 0;
    for (int k = 0; k < 15; ++k) {
      printf("%-10s", an[k]);
      double mn = 1e30, mx = 0;
      for (int w = 0; w < 8; ++w) {
        const unsigned long long v = h[64 + w * 32 + order[k]];
        const double t = v ? (double)(v - t0) / 1000.0 : -1;
        if (v) { mn = t < mn ? t : mn; mx = t > mx ? t : mx; }
        printf(" %6.2f", t);
      }
      printf("   %6.2f  %6.2f\n", mx - mn, mx - prevmax);
      prevmax = mx;
    }
    return 0;
  }
  CK(hipMemcpy(h.data(), stamps, NS * 8, hipMemcpyDeviceToHost));
  const char* names[25] = {"start", "att.end", "a2+fill", "bar17", "dense.run", "f1.fill", "bias+res", "ln", "film.st", "bar19",
                           "ffn1a.run", "f2a.fill", "silu.st", "bar", "ffn2a.run", "fill+bar", "ffn1b.run", "f2b.fill", "silu.st", "bar", "ffn2b.run", "(bar)", "res", "ln", "end"};
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < 8; ++w) if (h[64 + w * 32] && h[64 + w * 32] < t0) t0 = h[64 + w * 32];
  printf("enc_bc d=%d Lk=%d: per-wave stamps of workgroup 0 [cycles since the first wave's start / 1000]\n%-10s", d, Lk, "slot");
  for (int w = 0; w < 8; ++w) printf("   w%d  ", w);
  printf("  max-min  d(max)\n");
  double prevmax = 0;
  for (int sl = 0; sl < 25; ++sl) {
    printf("%-10s", names[sl]);
    double mn = 1e30, mx = 0;
    for (int w = 0; w < 8; ++w) {
      const unsigned long long v = h[64 + w * 32 + sl];
      const double t = v ? (double)(v - t0) / 1000.0 : -1;
      if (v) { mn = t < mn ? t : mn; mx = t > mx ? t : mx; }
      printf(" %6.2f", t);
    }
    printf("   %6.2f  %6.2f\n", mx - mn, mx - prevmax);
    prevmax = mx;
  }
  return 0;
}
