// bench_conv.cpp — isolated timing + per-stage stamps of the fused ConvBlock kernel on the six real block shapes.
// Build: hipcc -c tools/bench_conv.cpp -o tools/bin/bench_conv.o; hipcc tools/bin/bench_conv.o <pkg>/build/convblock.o <pkg>/build/enclayer.o -o tools/bin/bench_conv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffusion-handwriting-generation.pytorch_amd/csrc/dhw_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void* dev_rand(size_t bytes, bool f32 = false) {
  void* p; CK(hipMalloc(&p, bytes));
  if (f32) { std::vector<float> h(bytes / 4); for (size_t i = 0; i < h.size(); ++i) h[i] = ((float)((i * 2654435761u) >> 20 & 0xfff) / 4096.0f - 0.5f) * 0.1f; CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)); }
  else { std::vector<unsigned short> h(bytes / 2); for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3d00 + (((i * 2654435761u) >> 22) & 0x7f) + ((i & 1) ? 0x8000 : 0)); CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)); }
  return p;
}
int main(int argc, char** argv) {
  const int B = getenv("BENCH_B") ? atoi(getenv("BENCH_B")) : 64, reps = argc > 1 ? atoi(argv[1]) : 30;
  CK(convblock_init());
  struct Cfg { const char* n; int L, cin, cout, up; };   // up: channels of the skip activation (fused Upsample + skip_conv input) or 0
  const Cfg cfgs[] = {{"enc1", 488, 128, 128, 0}, {"enc2", 244, 128, 192, 0}, {"enc4", 122, 192, 256, 0}, {"dec3", 122, 384, 256, 0}, {"dec2", 244, 256, 192, 0}, {"dec1", 488, 192, 128, 0},
                      {"dec3+up", 122, 384, 256, 256}, {"dec2+up", 244, 256, 192, 192}, {"dec1+up", 488, 192, 128, 128}};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 8));
  for (const Cfg& c : cfgs) {
    ConvBlockParams p{};
    const size_t rows = (size_t)B * c.L + 128;
    p.x = dev_rand(rows * c.cin * 2); p.B = B; p.L = c.L; p.Cin = c.cin; p.Cout = c.cout;
    p.w_c1 = dev_rand((size_t)3 * c.cin * c.cout); p.w_c2 = dev_rand((size_t)3 * c.cout * c.cout); p.w_fc = dev_rand((size_t)2 * c.cout * c.cout); p.w_skip = dev_rand((size_t)6 * c.cin * c.cout);
    p.b_c1 = (float*)dev_rand(4096, true); p.b_c2 = p.b_c1; p.b_fc = p.b_c1; p.b_skip = p.b_c1;
    p.film = (float*)dev_rand(1 << 20, true); p.film_bs = 0; p.film_tot = 9280; p.f1 = 0; p.f2 = 256; p.f3 = 512;
    if (c.up) { p.up_h = dev_rand(rows * c.up * 2); p.up_cin = c.up; p.up_w = dev_rand((size_t)6 * c.up * c.cin); p.up_b = p.b_c1; p.up_low = p.x; }
    p.out = dev_rand(rows * c.cout * 2); p.out_f32 = 0; p.pool = nullptr; p.stamps = stamps;
    p.stagger = getenv("STAMP_WG") ? atoi(getenv("STAMP_WG")) : 0;   // (stamped builds reuse the field: which workgroup writes the stamps)
    CK(hipMemset(stamps, 0, 256 * 8));
    for (int i = 0; i < 3; ++i) CK(launch_convblock(PREC_BF16, p, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(launch_convblock(PREC_BF16, p, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hw[256]; CK(hipMemcpy(hw, stamps, sizeof hw, hipMemcpyDeviceToHost));
    unsigned long long* h = hw;   // wave 0
    if (getenv("STAMP_WAVES")) {   // every wave's stamps relative to wave 0's first
      for (int w = 0; w < 8; ++w) { printf("    wave %d:", w); for (int sl = 0; sl < 16; ++sl) printf(" %6.2f", hw[w * 16 + sl] ? ((double)hw[w * 16 + sl] - (double)hw[0]) / 100.0 : -1.0); printf("\n"); }
    }
    printf("%s L=%d %d->%d: %.2f us/launch (%d WGs); WG0 stamps [us]: stage0 %.2f | s1 run %.2f epi+bar %.2f | s2 run %.2f epi+bar %.2f | fc %.2f skip %.2f | bar %.2f out %.2f\n",
           c.n, c.L, c.cin, c.cout, ms * 1e3 / reps, B * ((c.L + 61) / 62),
           (h[1]-h[0])/100.0, (h[2]-h[1])/100.0, (h[3]-h[2])/100.0, (h[4]-h[3])/100.0, (h[5]-h[4])/100.0, (h[6]-h[5])/100.0, (h[7]-h[6])/100.0, (h[8]-h[7])/100.0, (h[9]-h[8])/100.0);
    if (c.up) printf("    fused input stage: staged %.2f | skip_conv run %.2f | epilogue %.2f | barrier+tail %.2f\n", (h[10]-h[0])/100.0, (h[11]-h[10])/100.0, (h[12]-h[11])/100.0, (h[1]-h[12])/100.0);
  }
  return 0;
}
