// bench_sgemm.cpp — the training step's GEMM kernel on its main shapes: time per launch and the phase stamps of one workgroup
// (the middle row tile): start | prologue done | loads of 4 K steps requested | first tile staged | K loop done | output written.
// Build: tools/build_tools.sh (links a -DDHW_STAMPS build of csrc/train.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/dhw_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dev_f32(size_t n) {
  float* p;
  CK(hipMalloc(&p, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = ((float)((i * 2654435761u) >> 20 & 0xfff) / 4096.0f - 0.5f) * 0.1f;
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  // form 0: A [M][K], B [K][N];  1: B^T [N][K] (nn.Linear forward);  2: A^T [K][M] (weight gradient, accumulating);
  // 3: Conv1d(k = 3) forward over samples of 480 rows, K = 3 Cin, weights [tap][N][Cin];  4: its weight gradient (M = Cout, N = Cin, K = rows, 3 taps as batch)
  struct Shape { int M, N, K, form; };
  const Shape shapes[] = {{15360, 128, 384, 3}, {7680, 192, 576, 3}, {3840, 256, 1152, 3}, {128, 128, 15360, 4}, {256, 384, 3840, 4},
                          {15360, 128, 128, 0}, {15360, 128, 128, 1}, {15360, 128, 384, 1}, {7680, 192, 192, 1}, {3840, 256, 256, 1}, {1920, 384, 384, 0},
                          {1920, 384, 384, 1}, {1920, 768, 384, 1}, {384, 384, 1920, 2}, {128, 128, 15360, 2}, {192, 192, 7680, 2}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 16 * 8));
  for (const Shape& s : shapes) {
    OpGemm g{};
    g.A = dev_f32((size_t)s.M * s.K + 64); g.B = dev_f32((size_t)s.K * s.N + 64); g.C = dev_f32((size_t)s.M * s.N + 64);
    if (s.form == 2) { g.sam = 1; g.sak = s.M; } else { g.sam = s.K; g.sak = 1; }
    if (s.form == 1) { g.sbk = 1; g.sbn = s.K; } else { g.sbk = s.N; g.sbn = 1; }
    g.scm = s.N; g.scn = 1;
    g.M = s.M; g.N = s.N; g.K = s.K; g.nzo = g.nzi = 1; g.taps = 1; g.alpha = 1.0f; g.accumulate = s.form == 2;
    if (s.form == 3) {   // x [M][Cin] rows shifted by tap - 1 inside 480-row samples; W [3][N][Cin]
      const int cin = s.K / 3;
      g.sam = cin; g.sak = 1; g.sbk = 1; g.sbn = cin; g.sbt = (long)s.N * cin; g.taps = 3; g.a_shift = -1; g.a_tap_shift = 1; g.lr = 480;
    } else if (s.form == 4) {   // dW[t][M][N] += dy^T [K][M] x[K + t - 1][N]
      g.sam = 1; g.sak = s.M; g.sbk = s.N; g.sbn = 1; g.nzi = 3; g.sczi = (long)s.M * s.N; g.b_shift = -1; g.b_z_shift = 1; g.lr = 480; g.accumulate = 1;
      hipFree(g.C); g.C = dev_f32((size_t)3 * s.M * s.N + 64);
    }
    g.stamps = stamps;
    CK(hipMemset(stamps, 0, 16 * 8));
    for (int i = 0; i < 3; ++i) CK(launch_sgemm(g, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(launch_sgemm(g, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / reps;
    printf("%6d x %4d x %6d %s: %6.2f us/launch %6.1f TFLOP/s | workgroup [us]:", s.M, s.N, s.K, s.form == 0 ? "AB " : s.form == 1 ? "ABt" : s.form == 2 ? "AtB" : s.form == 3 ? "cnv" : "cwg", us,
           2.0 * s.M * s.N * s.K * (s.form == 4 ? 3 : 1) / us / 1e6);
    for (int k = 1; k <= 5; ++k) printf(" %.2f", (double)(h[k] - h[0]) / 100.0);
    printf("\n");
    CK(hipFree((void*)g.A)); CK(hipFree((void*)g.B)); CK(hipFree(g.C));
  }
  return 0;
}
