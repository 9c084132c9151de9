// bench_gemm.cpp — isolated timing of the fused GEMM kernel on the denoiser's real layer shapes
// (B=64, L=488).  Build: hipcc --offload-arch=gfx950 -O3 tools/bench_gemm.cpp <pkg>/build/gemm.o -o gpurun_out/bench_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/dhw_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Shape { const char* name; int L, N, C0, taps0, C1, taps1, ln, silu; };

int main(int argc, char** argv) {
  const int B = 64;
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  CK(gemm_init());
  const Shape shapes[] = {
      {"enc1.conv1      L488 128x3->64", 488, 64, 128, 3, 0, 0, 0, 1},
      {"enc1.conv2      L488 64x3->128", 488, 128, 64, 3, 0, 0, 0, 0},
      {"enc1.fc_skip    L488 128+128x3->128", 488, 128, 128, 1, 128, 3, 0, 0},
      {"dec1.conv1      L488 192x3->64", 488, 64, 192, 3, 0, 0, 0, 1},
      {"dec1.fc_skip    L488 128+192x3->128", 488, 128, 128, 1, 192, 3, 0, 0},
      {"skip_conv1      L488 128x3->192", 488, 192, 128, 3, 0, 0, 0, 0},
      {"enc2.fc_skip    L244 192+128x3->192", 244, 192, 192, 1, 128, 3, 0, 0},
      {"enc3.qkv_self   L244 192->576", 244, 576, 192, 1, 0, 0, 0, 0},
      {"enc3.dense+ln   L244 192->192", 244, 192, 192, 1, 0, 0, 1, 0},
      {"enc3.ffn1       L244 192->384", 244, 384, 192, 1, 0, 0, 0, 1},
      {"enc3.ffn2+ln    L244 384->192", 244, 192, 384, 1, 0, 0, 1, 0},
      {"dec3.conv1      L122 384x3->128", 122, 128, 384, 3, 0, 0, 0, 1},
      {"dec3.fc_skip    L122 256+384x3->256", 122, 256, 256, 1, 384, 3, 0, 0},
      {"att.qkv_self    L61  384->1152", 61, 1152, 384, 1, 0, 0, 0, 0},
      {"att.dense+ln    L61  384->384", 61, 384, 384, 1, 0, 0, 1, 0},
      {"att.ffn1        L61  384->768", 61, 768, 384, 1, 0, 0, 0, 1},
      {"att.ffn2+ln     L61  768->384", 61, 384, 768, 1, 0, 0, 1, 0},
      {"enc.text_dense  L30  384->384 ln", 30, 384, 384, 1, 0, 0, 1, 1},
  };
  void *A, *A2, *W, *W2, *O;
  float *bias, *film;
  const size_t big = (size_t)B * 512 * 1200 * 2;
  CK(hipMalloc(&A, big)); CK(hipMalloc(&A2, big)); CK(hipMalloc(&O, big));
  CK(hipMalloc(&W, 8 << 20)); CK(hipMalloc(&W2, 8 << 20));
  CK(hipMalloc(&bias, 1 << 16)); CK(hipMalloc(&film, 1 << 20));
  CK(hipMemset(bias, 0, 1 << 16)); CK(hipMemset(film, 0, 1 << 20));
  {  // non-trivial bf16 data (values in +-[1,2)) so clocks and MFMA power are realistic
    std::vector<unsigned short> h(big / 2);
    for (size_t i = 0; i < h.size(); ++i)
      h[i] = (unsigned short)(0x3f80 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0));
    CK(hipMemcpy(A, h.data(), big, hipMemcpyHostToDevice));
    CK(hipMemcpy(A2, h.data(), big, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 8 << 20, hipMemcpyHostToDevice));
    CK(hipMemcpy(W2, h.data(), 8 << 20, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    GemmParams p{};
    p.nseg = s.C1 ? 2 : 1;
    p.seg[0] = GemmSeg{A, W, s.C0, s.taps0, s.silu};
    p.seg[1] = GemmSeg{A2, W2, s.C1, s.taps1 ? s.taps1 : 1, 0};
    p.B = B; p.L = s.L; p.N = s.N; p.n_store = s.N;
    p.bias0 = bias; p.bias1 = bias;
    p.gam = film; p.bet = film + 4096; p.film_bs = 0; p.film_mode = s.C1 ? 2 : 1;
    p.ln = s.ln; p.out = O;
    int bm, bn;
    gemm_tile_for(PREC_BF16, p, &bm, &bn);
    for (int i = 0; i < 5; ++i) CK(launch_gemm(PREC_BF16, p, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(launch_gemm(PREC_BF16, p, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double K = (double)s.C0 * s.taps0 + (double)s.C1 * s.taps1;
    const double fl = 2.0 * B * s.L * s.N * K;
    printf("%-40s BM=%2d BN=%3d WGs=%5d  %7.2f us  %7.1f TFLOP/s\n", s.name, bm, bn,
           B * ((s.L + bm - 1) / bm) * (s.N / bn), us, fl / us / 1e6);
  }
  return 0;
}
