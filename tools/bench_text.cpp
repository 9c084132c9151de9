// bench_text.cpp — timing of the fused GEMM kernel on the all-steps text-plane shapes (B*T = 3840 "samples" of Lt = 30
// text rows / S5 = 70 style rows), per-sample tiling vs one flat row list.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/dhw_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Shape { const char* name; int L, C, N, n_store, ln, silu_in, silu_out, res1; };
int main(int argc, char** argv) {
  const int BT = 3840, reps = argc > 1 ? atoi(argv[1]) : 20;
  setvbuf(stdout, nullptr, _IONBF, 0);
  CK(gemm_init());
  const Shape shapes[] = {
      {"ts.q        384->384", 30, 384, 384, 384, 0, 0, 0, 0},
      {"ts.kv       384->768 (V^T)", 70, 384, 768, 384, 0, 0, 0, 0},
      {"ts.dense    384->384 res+ln+film", 30, 384, 384, 384, 1, 0, 0, 1},
      {"ts.ffn1     384->768 silu", 30, 384, 768, 768, 0, 1, 1, 0},
      {"ts.ffn2     768->384 ln+film", 30, 768, 384, 384, 1, 0, 0, 0},
      {"text_dense  384->384 ln+film", 30, 384, 384, 384, 1, 1, 0, 0},
      {"kv_text     384->768 (V^T)", 30, 384, 768, 384, 0, 0, 0, 0},
      {"kv_text     192->384 (V^T)", 30, 192, 384, 192, 0, 0, 0, 0},
  };
  const size_t big = (size_t)BT * 72 * 768 * 2 + (1 << 20);
  void *A, *W, *O, *V, *R;
  float *bias, *film;
  CK(hipMalloc(&A, big)); CK(hipMalloc(&O, big)); CK(hipMalloc(&V, big)); CK(hipMalloc(&R, big));
  CK(hipMalloc(&W, 8 << 20)); CK(hipMalloc(&bias, 1 << 16)); CK(hipMalloc(&film, 8 << 20));
  CK(hipMemset(bias, 0, 1 << 16)); CK(hipMemset(film, 0, 8 << 20));
  {
    std::vector<unsigned short> h(big / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3d00 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0));
    CK(hipMemcpy(A, h.data(), big, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, h.data(), big, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 8 << 20, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 64 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes)
    for (int flat = 0; flat < 2; ++flat) {   // 0 per-sample tiles, 1 flat tiles (gemm.hip), 2 persistent pipelined kernel (rowgemm.hip)
      if (flat == 1 && s.n_store != s.N) continue;   // the transposed-V output is per sample
      GemmParams p{};
      p.nseg = 1;
      p.seg[0] = GemmSeg{A, W, s.C, 1, s.silu_in};
      p.B = flat == 1 ? 1 : BT; p.L = flat == 1 ? BT * s.L : s.L; p.N = s.N; p.n_store = s.n_store;
      p.bias0 = bias; p.gam = film; p.bet = film + 4096; p.film_bs = 18560; p.film_div = 64; p.film_mode = s.ln ? 1 : 0;
      p.ln = s.ln; p.silu_out = s.silu_out; p.res1 = s.res1 ? R : nullptr;
      p.out = O; p.vt = V; p.vt_lpad = 96; p.stamps = stamps;
      int bm, bn;
      gemm_tile_for(PREC_BF16, p, &bm, &bn);
      auto go = [&]() { return launch_gemm(PREC_BF16, p, st); };
      for (int i = 0; i < 3; ++i) CK(go());
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CK(go());
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps, fl = 2.0 * BT * s.L * s.N * s.C;
      const double bytes = (double)BT * s.L * (s.C + s.N + (s.res1 ? s.N : 0)) * 2;
      printf("%-36s %s BM=%2d BN=%3d WGs=%6ld  %7.1f us  %6.1f TFLOP/s  %5.2f TB/s\n", s.name, flat == 2 ? "pipe" : flat ? "flat" : "per ", bm, bn,
             (long)p.B * ((p.L + bm - 1) / bm) * (s.N / bn), us, fl / us / 1e6, bytes / us / 1e6);
      unsigned long long h[8]; CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
      if (flat == 2) printf("      middle WG, 3rd tile [us]: main loop %.2f | bias/res/LN %.2f | film/silu -> out tile %.2f | copy-out (+ other column blocks) %.2f | commit next A %.2f\n", (h[1]-h[0])/100.0, (h[2]-h[1])/100.0, (h[3]-h[2])/100.0, (h[4]-h[3])/100.0, (h[5]-h[4])/100.0);
      if (flat < 2) printf("      middle WG [us]: stage A %.2f | main loop %.2f | bias/res/LN %.2f | out tile %.2f | copy-out %.2f\n", (h[1]-h[0])/100.0, (h[2]-h[1])/100.0, (h[3]-h[2])/100.0, (h[4]-h[3])/100.0, (h[5]-h[4])/100.0);
    }
  return 0;
}
