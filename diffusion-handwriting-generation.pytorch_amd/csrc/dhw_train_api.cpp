// dhw_train_api.cpp — C-ABI of the training step's first slice (include/dhw_train.h): loss / perturbation / optimizer
// kernels and the ConvBlock forward + backward built from the generic MFMA GEMM (forward and data-gradient
// convolutions, the latter with transposed / tap-flipped packed weights) and the kernels of train.hip.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dhw_train.h"
#include "abi_guard.h"
#include "dhw_kernels.h"

namespace {

thread_local ErrBuf g_terr;   // per calling thread, as dhw_train_last_error() is documented

int tfail(int code, const char* fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  g_terr.vsetf(fmt, ap);
  va_end(ap);
  return code;
}
// the body of every extern "C" entry point runs inside this: no exception leaves the library (abi_guard.h)
#define TRAIN_GUARD(fn, R, ...) \
  return abi_guard<R>(fn, [&](const char* f_, const char* w_) { return tfail(DHW_ERR_INTERNAL, "%s: internal error: %s", f_, w_); }, [&]() -> R __VA_ARGS__)
#define THIP(call)                                                                            \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return tfail(DHW_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));  \
  } while (0)

// scratch allocations of one dhw_train_convblock call, released on every exit path
struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() {
    hipDeviceSynchronize();
    for (void* p : ptrs) hipFree(p);
  }
  float* f32(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 4) * sizeof(float)) != hipSuccess) return nullptr;
    hipMemset(p, 0, (n ? n : 4) * sizeof(float));
    ptrs.push_back(p);
    return (float*)p;
  }
  float* upload(const std::vector<float>& v) {
    float* p = f32(v.size());
    if (p && hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
  }
  // row-major Wf[N][K] -> MFMA-fragment order [N/16][K/32][64 lanes][8], fp32 (as dhw_api.cpp upload_packed)
  float* packed(const std::vector<float>& wf, int N, int K) {
    std::vector<float> pk((size_t)N * K);
    size_t o = 0;
    for (int nt = 0; nt < N / 16; ++nt)
      for (int kc = 0; kc < K / 32; ++kc)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) pk[o++] = wf[(size_t)(nt * 16 + (l & 15)) * K + kc * 32 + 8 * (l >> 4) + j];
    return upload(pk);
  }
};

// forward convolution matrix: Wf[co][tap * cin + ci] = W[co][ci][tap]
std::vector<float> fwd_matrix(const float* w, int cout, int cin, int taps) {
  std::vector<float> f((size_t)cout * cin * taps);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i)
      for (int t = 0; t < taps; ++t) f[(size_t)o * cin * taps + t * cin + i] = w[((size_t)o * cin + i) * taps + t];
  return f;
}
// data-gradient matrix: dX[r][ci] = sum_{tap', co} dY[r + tap' - 1][co] * W[co][ci][taps - 1 - tap']  =>  Wd[ci][tap' * cout + co]
std::vector<float> dgrad_matrix(const float* w, int cout, int cin, int taps) {
  std::vector<float> f((size_t)cout * cin * taps);
  for (int i = 0; i < cin; ++i)
    for (int t = 0; t < taps; ++t)
      for (int o = 0; o < cout; ++o) f[(size_t)i * cout * taps + t * cout + o] = w[((size_t)o * cin + i) * taps + (taps - 1 - t)];
  return f;
}

int conv_gemm(const float* in, int B, int L, int C, int taps, const float* w_packed, const float* bias, int N, float* out, hipStream_t st) {
  GemmParams p{};
  p.nseg = 1;
  p.seg[0] = GemmSeg{in, w_packed, C, taps, 0};
  p.B = B;
  p.L = L;
  p.N = N;
  p.n_store = N;
  p.bias0 = bias;
  p.film_div = 1;
  p.out = out;
  hipError_t e = launch_gemm(PREC_F32, p, st);
  if (e != hipSuccess) return tfail(DHW_ERR_HIP, "conv gemm %d x%d -> %d: %s", C, taps, N, hipGetErrorString(e));
  return 0;
}

}  // namespace

extern "C" {

const char* dhw_train_last_error(void) { return g_terr.c_str(); }

int dhw_train_perturb(const float* x, const float* eps, const float* alphas, int B, int L, float* out, void* hip_stream) {
  TRAIN_GUARD("dhw_train_perturb", int, {
    if (!x || !eps || !alphas || !out || B < 1 || L < 1) return tfail(DHW_ERR_ARG, "dhw_train_perturb: bad argument");
    THIP(launch_perturb(x, eps, alphas, B, L, out, (hipStream_t)hip_stream));
    return 0;
  });
}

int dhw_train_loss(const float* eps, const float* score_pred, const float* pen, const float* pen_pred, const float* alphas, int B, int L,
                   float* out3, float* d_score, float* d_pen_pred, void* hip_stream) {
  TRAIN_GUARD("dhw_train_loss", int, {
    if (!eps || !score_pred || !pen || !pen_pred || !alphas || !out3 || B < 1 || L < 1) return tfail(DHW_ERR_ARG, "dhw_train_loss: bad argument");
    THIP(launch_loss(eps, score_pred, pen, pen_pred, alphas, B, L, out3, d_score, d_pen_pred, (hipStream_t)hip_stream));
    return 0;
  });
}

int dhw_train_adam(int nbuf, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float max_norm, float* grad_norm_out, void* hip_stream) {
  TRAIN_GUARD("dhw_train_adam", int, {
    if (nbuf < 1 || !p || !g || !m || !v || !n || step < 1) return tfail(DHW_ERR_ARG, "dhw_train_adam: bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    // the squared-norm scalar: allocated once per DEVICE and kept for the life of the process (a hipMalloc / hipFree pair and a
    // stream synchronisation per update used to sit here).  One scalar per device: calls on different streams of one device must
    // not overlap, and the first call on a device allocates, so it must not be made under stream capture — dhw_train_adam_dev,
    // which takes the scalar from the caller, is the form for graphs and concurrent streams (include/dhw_train.h).
    static std::mutex sq_mu;
    static std::map<int, float*> sq_by_device;
    float* sq = nullptr;
    if (max_norm > 0.f || grad_norm_out) {
      int dev = 0;
      THIP(hipGetDevice(&dev));
      {
        std::lock_guard<std::mutex> lk(sq_mu);
        float*& slot = sq_by_device[dev];
        if (!slot) THIP(hipMalloc((void**)&slot, sizeof(float)));
        sq = slot;
      }
      THIP(hipMemsetAsync(sq, 0, sizeof(float), st));
      for (int i = 0; i < nbuf; ++i) THIP(launch_sqnorm(g[i], n[i], sq, st));
    }
    for (int i = 0; i < nbuf; ++i)
      THIP(launch_adam(p[i], g[i], m[i], v[i], n[i], lr, beta1, beta2, eps, weight_decay, step, max_norm > 0.f ? sq : nullptr, max_norm, st));
    if (sq && grad_norm_out) THIP(hipMemcpyAsync(grad_norm_out, sq, sizeof(float), hipMemcpyDeviceToDevice, st));   // (squared norm; stream-ordered)
    return 0;
  });
}

int dhw_train_convblock(int device, int B, int L, int cin, int cout, const float* x, const float* sigma, const float* dout,
                        const dhw_convblock_weights* w, float* out, float* dx, float* dsigma, const dhw_convblock_grads* g, void* hip_stream) {
  TRAIN_GUARD("dhw_train_convblock", int, {
    if (!x || !sigma || !dout || !w || !out || !dx || !dsigma || !g) return tfail(DHW_ERR_ARG, "dhw_train_convblock: null pointer");
    const int c1 = cout / 2;
    if (B < 1 || L < 2 || (L & 1) || cin % 32 || c1 % 32 || cout % 64) return tfail(DHW_ERR_ARG, "dhw_train_convblock: unsupported shape B=%d L=%d %d -> %d", B, L, cin, cout);
    THIP(hipSetDevice(device));
    if (gemm_init() != hipSuccess) return tfail(DHW_ERR_HIP, "kernel attribute setup failed");
    hipStream_t st = (hipStream_t)hip_stream;
    Scratch s;
    const long rows = (long)B * L, slack = 64;
    const int tot = c1 + 2 * cout, cols = 2 * tot;          // FiLM table columns: gamma1|gamma2|gamma3|beta1|beta2|beta3
    const int go[3] = {0, c1, c1 + cout}, bo[3] = {tot, tot + c1, tot + c1 + cout};

    // ---- weights: forward and data-gradient matrices in MFMA-fragment order
    float* wf1 = s.packed(fwd_matrix(w->conv1_w, c1, cin, 3), c1, 3 * cin);
    float* wf2 = s.packed(fwd_matrix(w->conv2_w, cout, c1, 3), cout, 3 * c1);
    float* wff = s.packed(fwd_matrix(w->fc_w, cout, cout, 1), cout, cout);
    float* wfs = s.packed(fwd_matrix(w->skip_w, cout, cin, 3), cout, 3 * cin);
    float* wd1 = s.packed(dgrad_matrix(w->conv1_w, c1, cin, 3), cin, 3 * c1);
    float* wd2 = s.packed(dgrad_matrix(w->conv2_w, cout, c1, 3), c1, 3 * cout);
    float* wdf = s.packed(dgrad_matrix(w->fc_w, cout, cout, 1), cout, cout);
    float* wds = s.packed(dgrad_matrix(w->skip_w, cout, cin, 3), cin, 3 * cout);
    float* b1 = s.upload(std::vector<float>(w->conv1_b, w->conv1_b + c1));
    float* b2 = s.upload(std::vector<float>(w->conv2_b, w->conv2_b + cout));
    float* bf = s.upload(std::vector<float>(w->fc_b, w->fc_b + cout));
    float* bsk = s.upload(std::vector<float>(w->skip_b, w->skip_b + cout));
    float* wcat = s.upload(std::vector<float>(w->film_w, w->film_w + (size_t)cols * 32));
    float* bcat = s.upload(std::vector<float>(w->film_b, w->film_b + cols));
    // ---- activations kept for the backward pass (every buffer with slack rows for the GEMM tiles)
    auto act = [&](int C) { return s.f32((size_t)(rows + slack) * C); };
    float *xs = act(cin), *sx = act(cin), *u1 = act(c1), *a1 = act(c1), *h1 = act(c1), *u2 = act(cout), *a2 = act(cout), *h2 = act(cout),
          *u3 = act(cout), *a3 = act(cout), *sk = act(cout), *dos = act(cout);
    float *du3 = act(cout), *dh2 = act(cout), *du2 = act(cout), *dh1 = act(c1), *du1 = act(c1), *dsx = act(cin), *dxs = act(cin);
    float* film = s.f32((size_t)B * cols);
    float* dfilm = s.f32((size_t)B * cols);
    float* zb = s.f32(cout);   // zero bias for the data-gradient convolutions
    if (!wf1 || !wf2 || !wff || !wfs || !wd1 || !wd2 || !wdf || !wds || !b1 || !b2 || !bf || !bsk || !wcat || !bcat || !dxs || !film || !dfilm || !zb)
      return tfail(DHW_ERR_HIP, "dhw_train_convblock: out of device memory");
    THIP(hipMemcpyAsync(xs, x, rows * cin * 4, hipMemcpyDeviceToDevice, st));      // (the caller's buffers carry no slack rows)
    THIP(hipMemcpyAsync(dos, dout, rows * cout * 4, hipMemcpyDeviceToDevice, st));
    int rc;

    // ---- forward (cnn.py:64-87), every pre-activation kept
    THIP(launch_film(sigma, B, wcat, bcat, cols, film, st));
    THIP(launch_silu_fwd(xs, rows * cin, sx, st));
    if ((rc = conv_gemm(sx, B, L, cin, 3, wf1, b1, c1, u1, st))) return rc;
    THIP(launch_film_silu_fwd(u1, film, cols, go[0], bo[0], B, L, c1, a1, h1, st));
    if ((rc = conv_gemm(h1, B, L, c1, 3, wf2, b2, cout, u2, st))) return rc;
    THIP(launch_film_silu_fwd(u2, film, cols, go[1], bo[1], B, L, cout, a2, h2, st));
    if ((rc = conv_gemm(h2, B, L, cout, 1, wff, bf, cout, u3, st))) return rc;
    THIP(launch_film_silu_fwd(u3, film, cols, go[2], bo[2], B, L, cout, a3, nullptr, st));
    if ((rc = conv_gemm(xs, B, L, cin, 3, wfs, bsk, cout, sk, st))) return rc;
    THIP(launch_add(a3, sk, rows * cout, out, st));

    // ---- backward
    const size_t n1 = (size_t)c1 * cin * 3, n2 = (size_t)cout * c1 * 3, nf = (size_t)cout * cout, ns = (size_t)cout * cin * 3;
    THIP(hipMemsetAsync(g->conv1_w, 0, n1 * 4, st)); THIP(hipMemsetAsync(g->conv1_b, 0, c1 * 4, st));
    THIP(hipMemsetAsync(g->conv2_w, 0, n2 * 4, st)); THIP(hipMemsetAsync(g->conv2_b, 0, cout * 4, st));
    THIP(hipMemsetAsync(g->fc_w, 0, nf * 4, st));    THIP(hipMemsetAsync(g->fc_b, 0, cout * 4, st));
    THIP(hipMemsetAsync(g->skip_w, 0, ns * 4, st));  THIP(hipMemsetAsync(g->skip_b, 0, cout * 4, st));
    // out = FiLM3(fc(h2)) + conv_skip(x): both branches see dout
    THIP(launch_film_bwd(dos, nullptr, u3, film, cols, go[2], B, L, cout, 0, du3, dfilm, cols, go[2], bo[2], st));
    THIP(launch_wgrad(du3, h2, B, L, cout, cout, 1, g->fc_w, st));
    THIP(launch_colsum(du3, rows, cout, g->fc_b, st));
    if ((rc = conv_gemm(du3, B, L, cout, 1, wdf, zb, cout, dh2, st))) return rc;
    // h2 = SiLU(FiLM2(conv2(h1)))
    THIP(launch_film_bwd(dh2, a2, u2, film, cols, go[1], B, L, cout, 1, du2, dfilm, cols, go[1], bo[1], st));
    THIP(launch_wgrad(du2, h1, B, L, cout, c1, 3, g->conv2_w, st));
    THIP(launch_colsum(du2, rows, cout, g->conv2_b, st));
    if ((rc = conv_gemm(du2, B, L, cout, 3, wd2, zb, c1, dh1, st))) return rc;
    // h1 = SiLU(FiLM1(conv1(SiLU(x))))
    THIP(launch_film_bwd(dh1, a1, u1, film, cols, go[0], B, L, c1, 1, du1, dfilm, cols, go[0], bo[0], st));
    THIP(launch_wgrad(du1, sx, B, L, c1, cin, 3, g->conv1_w, st));
    THIP(launch_colsum(du1, rows, c1, g->conv1_b, st));
    if ((rc = conv_gemm(du1, B, L, c1, 3, wd1, zb, cin, dsx, st))) return rc;
    // skip branch, then dx = conv_skip^T(dout) + conv1^T(..) * SiLU'(x)
    THIP(launch_wgrad(dos, xs, B, L, cout, cin, 3, g->skip_w, st));
    THIP(launch_colsum(dos, rows, cout, g->skip_b, st));
    if ((rc = conv_gemm(dos, B, L, cout, 3, wds, zb, cin, dxs, st))) return rc;
    THIP(launch_silu_bwd_add(dsx, xs, rows * cin, dxs, st));
    THIP(hipMemcpyAsync(dx, dxs, rows * cin * 4, hipMemcpyDeviceToDevice, st));
    // the six FiLM Linears
    THIP(launch_film_linear_bwd(dfilm, sigma, wcat, B, cols, g->film_w, g->film_b, dsigma, st));
    THIP(hipStreamSynchronize(st));
    return 0;
  });
}

int dhw_train_adam_dev(int nbuf, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                       const float* hyper, float* sqnorm, void* hip_stream) {
  TRAIN_GUARD("dhw_train_adam_dev", int, {
    if (nbuf < 1 || !p || !g || !m || !v || !n || !hyper || !sqnorm) return tfail(DHW_ERR_ARG, "dhw_train_adam_dev: bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    THIP(hipMemsetAsync(sqnorm, 0, sizeof(float), st));
    for (int i = 0; i < nbuf; ++i) THIP(launch_sqnorm(g[i], n[i], sqnorm, st));
    for (int i = 0; i < nbuf; ++i) THIP(launch_adam_dev(p[i], g[i], m[i], v[i], n[i], hyper, sqnorm, st));
    return 0;
  });
}

int dhw_train_draw(const uint64_t* rng, int B, int L, float* eps, long long n_keep, int keep_per_sample, float p, float* keep, void* hip_stream) {
  TRAIN_GUARD("dhw_train_draw", int, {
    if (!rng || !eps || !keep || B < 1 || L < 1 || n_keep < 1 || keep_per_sample < 4 || keep_per_sample % 4 || n_keep % keep_per_sample || p < 0.f || p >= 1.f)
      return tfail(DHW_ERR_ARG, "dhw_train_draw: bad argument");
    THIP(launch_train_draw(rng, B, L, eps, n_keep, keep_per_sample, p, keep, (hipStream_t)hip_stream));
    return 0;
  });
}

int dhw_op_keep_mask(const uint64_t* rng, int site, long long n, int per_sample, float p, float* keep, void* hip_stream) {
  TRAIN_GUARD("dhw_op_keep_mask", int, {
    if (!rng || !keep || site < 3 || n < 1 || per_sample < 4 || per_sample % 4 || n % per_sample || p < 0.f || p >= 1.f)
      return tfail(DHW_ERR_ARG, "dhw_op_keep_mask: bad argument");
    THIP(launch_keep_mask(rng, site, n, per_sample, p, keep, (hipStream_t)hip_stream));
    return 0;
  });
}

int dhw_op_film_table(const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int total, float* film, void* hip_stream) {
  TRAIN_GUARD("dhw_op_film_table", int, {
    if (!sigma || !flat || !woff || !boff || !film || B < 1 || total < 1) return tfail(DHW_ERR_ARG, "dhw_op_film_table: bad argument");
    THIP(launch_film_table(0, sigma, flat, woff, boff, B, total, film, nullptr, nullptr, (hipStream_t)hip_stream));
    return 0;
  });
}
int dhw_op_film_table_bwd(const float* dfilm, const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int total,
                          float* grad_flat, float* dsigma, void* hip_stream) {
  TRAIN_GUARD("dhw_op_film_table_bwd", int, {
    if (!dfilm || !sigma || !flat || !woff || !boff || !grad_flat || !dsigma || B < 1 || total < 1)
      return tfail(DHW_ERR_ARG, "dhw_op_film_table_bwd: bad argument");
    THIP(launch_film_table(1, sigma, flat, woff, boff, B, total, const_cast<float*>(dfilm), grad_flat, dsigma, (hipStream_t)hip_stream));
    return 0;
  });
}

#define OPCHECK(cond, name) if (!(cond)) return tfail(DHW_ERR_ARG, name ": bad argument")

static int gemm_from_desc(const dhw_gemm_desc* d, OpGemm& g) {
  OPCHECK(d && d->A && d->B && d->C && d->M > 0 && d->N > 0 && d->K > 0 && d->nzo > 0 && d->nzi > 0 && d->lr >= 0 && d->taps >= 1, "dhw_op_gemm");
  if ((d->a_shift || d->b_shift || d->a_tap_shift || d->b_z_shift) && d->lr < 1) return tfail(DHW_ERR_ARG, "dhw_op_gemm: a shift needs lr (rows per sample)");
  if (d->taps > 1 && (d->K % d->taps || (d->K / d->taps) % 32)) return tfail(DHW_ERR_ARG, "dhw_op_gemm: taps > 1 needs K / taps to be a multiple of 32");
  g.A = d->A; g.sam = d->sam; g.sak = d->sak; g.sazo = d->sazo; g.sazi = d->sazi; g.a_shift = d->a_shift; g.a_tap_shift = d->a_tap_shift;
  g.B = d->B; g.sbk = d->sbk; g.sbn = d->sbn; g.sbzo = d->sbzo; g.sbzi = d->sbzi; g.sbt = d->sbt; g.b_shift = d->b_shift; g.b_z_shift = d->b_z_shift;
  g.C = d->C; g.scm = d->scm; g.scn = d->scn; g.sczo = d->sczo; g.sczi = d->sczi;
  g.M = d->M; g.N = d->N; g.K = d->K; g.nzo = d->nzo; g.nzi = d->nzi; g.lr = d->lr; g.taps = d->taps;
  g.bias = d->bias; g.alpha = d->alpha; g.accumulate = d->accumulate; g.bf16 = d->bf16 ? 1 : 0; g.rowsum = d->rowsum; g.addend = d->addend; g.act_out = d->act_out; g.dsilu_of = d->dsilu_of; g.stamps = nullptr;
  g.film_g = d->film_gamma; g.film_b = d->film_beta; g.film_ps = d->film_pstride; g.film_rows = d->film_rows; g.film_act = d->film_act; g.film_out = d->film_out; g.film_add = d->film_addend;
  if (g.film_out && (!g.film_g || !g.film_b || g.film_rows < 1 || g.accumulate)) return tfail(DHW_ERR_ARG, "dhw_op_gemm: film_out needs gamma, beta, film_rows >= 1 and accumulate = 0");
  // (the rider's second output and its addend are addressed without a batch offset, train.hip "unbatched GEMMs": every batch of a
  // batched launch would write the same film_out rows)
  if (g.film_out && (long long)g.nzo * g.nzi != 1) return tfail(DHW_ERR_ARG, "dhw_op_gemm: film_out is for unbatched GEMMs (nzo * nzi = 1), got %d x %d", g.nzo, g.nzi);
  return 0;
}
int dhw_op_gemm(const dhw_gemm_desc* d, void* hip_stream) {
  TRAIN_GUARD("dhw_op_gemm", int, {
    OpGemm g;
    if (int rc = gemm_from_desc(d, g)) return rc;
    THIP(launch_sgemm(g, (hipStream_t)hip_stream));
    return 0;
  });
}
int dhw_op_gemm2(const dhw_gemm_desc* d0, const dhw_gemm_desc* d1, void* hip_stream) {
  TRAIN_GUARD("dhw_op_gemm2", int, {
    OpGemm g0, g1;
    if (int rc = gemm_from_desc(d0, g0)) return rc;
    if (int rc = gemm_from_desc(d1, g1)) return rc;
    int nl = 2;
    THIP(launch_sgemm_pair(g0, g1, (hipStream_t)hip_stream, &nl));
    return nl;
  });
}
int dhw_op_gemm_group(const dhw_gemm_desc* d, int n, void* hip_stream) {
  TRAIN_GUARD("dhw_op_gemm_group", int, {
    OPCHECK(d && n >= 1 && n <= 6, "dhw_op_gemm_group");
    OpGemm g[6];
    for (int i = 0; i < n; ++i)
      if (int rc = gemm_from_desc(d + i, g[i])) return rc;
    int nl = n;
    THIP(launch_sgemm_group(g, n, (hipStream_t)hip_stream, &nl));
    return nl;
  });
}
int dhw_op_unary(int kind, const float* x, long long n, float* y, void* st) {
  TRAIN_GUARD("dhw_op_unary", int, {
    OPCHECK(x && y && n > 0 && (kind == 0 || kind == 1), "dhw_op_unary");
    THIP(launch_unary(kind, x, n, y, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_unary_bwd(int kind, const float* dy, const float* x, long long n, float* dx, int accumulate, void* st) {
  TRAIN_GUARD("dhw_op_unary_bwd", int, {
    OPCHECK(dy && x && dx && n > 0 && (kind == 0 || kind == 1), "dhw_op_unary_bwd");
    THIP(launch_unary_bwd(kind, dy, x, n, dx, accumulate, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_add(const float* a, const float* b, long long n, float* out, int accumulate, void* st) {
  TRAIN_GUARD("dhw_op_add", int, {
    OPCHECK(a && out && n > 0, "dhw_op_add");
    THIP(launch_add2(a, b, n, out, accumulate, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_add_rows(const float* x, const float* table, int B, int L, int C, float* out, void* st) {
  TRAIN_GUARD("dhw_op_add_rows", int, {
    OPCHECK(x && table && out && B > 0 && L > 0 && C > 0, "dhw_op_add_rows");
    THIP(launch_add_rows(x, table, (long)B * L * C, (long)L * C, out, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_film(const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, float* y, void* st) {
  TRAIN_GUARD("dhw_op_film", int, {
    OPCHECK(x && gamma && beta && y && B > 0 && L > 0 && C > 0, "dhw_op_film");
    THIP(launch_film_fwd(x, gamma, beta, pstride, B, L, C, y, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_film_bwd(const float* dy, const float* x, const float* gamma, long long pstride, int B, int L, int C, float* dx, int accumulate,
                    float* dgamma, float* dbeta, void* st) {
  TRAIN_GUARD("dhw_op_film_bwd", int, {
    OPCHECK(dy && x && gamma && dx && dgamma && dbeta && B > 0 && L > 0 && C > 0, "dhw_op_film_bwd");
    THIP(launch_film_bwd2(dy, x, gamma, pstride, B, L, C, dx, accumulate, dgamma, dbeta, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_film_act(const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, int act, const float* addend, float* y,
                    void* st) {
  TRAIN_GUARD("dhw_op_film_act", int, {
    OPCHECK(x && gamma && beta && y && B > 0 && L > 0 && C > 0 && C % 4 == 0 && pstride % 4 == 0, "dhw_op_film_act");
    THIP(launch_film_act_fwd(x, gamma, beta, pstride, B, L, C, act, addend, y, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_film_act_bwd(const float* dy, const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, int act, float* dx,
                        int accumulate, float* dgamma, float* dbeta, void* st) {
  TRAIN_GUARD("dhw_op_film_act_bwd", int, {
    OPCHECK(dy && x && gamma && beta && dx && dgamma && dbeta && B > 0 && L > 0 && C > 0, "dhw_op_film_act_bwd");
    THIP(launch_film_act_bwd(dy, x, gamma, beta, pstride, B, L, C, act, dx, accumulate, dgamma, dbeta, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_ln_film(const float* x, int B, int L, int C, const float* gamma, const float* beta, long long pstride, const float* addend, float* y, float* act_out,
                   const float* pe, float* pe_out, float* mean, float* rstd, void* st) {
  TRAIN_GUARD("dhw_op_ln_film", int, {
    OPCHECK(x && gamma && beta && y && mean && rstd && B > 0 && L > 0 && C > 0 && (!pe_out || pe), "dhw_op_ln_film");
    THIP(launch_ln_film_fwd(x, (long)B * L, C, gamma, beta, pstride, L, addend, y, act_out, pe, pe_out, mean, rstd, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_ln_film_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, long long pstride, int B, int L, int C,
                       float* dx, int accumulate, float* dgamma, float* dbeta, void* st) {
  TRAIN_GUARD("dhw_op_ln_film_bwd", int, {
    OPCHECK(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && B > 0 && L > 0 && C > 0 && C <= 512, "dhw_op_ln_film_bwd");
    THIP(launch_ln_film_bwd(dy, x, mean, rstd, gamma, pstride, B, L, C, dx, accumulate, dgamma, dbeta, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_layernorm(const float* x, long long rows, int C, float* y, float* mean, float* rstd, void* st) {
  TRAIN_GUARD("dhw_op_layernorm", int, {
    OPCHECK(x && y && mean && rstd && rows > 0 && C > 0, "dhw_op_layernorm");
    THIP(launch_ln_fwd(x, rows, C, y, mean, rstd, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_layernorm_bwd(const float* dy, const float* y, const float* rstd, long long rows, int C, float* dx, int accumulate, void* st) {
  TRAIN_GUARD("dhw_op_layernorm_bwd", int, {
    OPCHECK(dy && y && rstd && dx && rows > 0 && C > 0, "dhw_op_layernorm_bwd");
    THIP(launch_ln_bwd(dy, y, rstd, rows, C, dx, accumulate, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_softmax(const float* s, long long rows, int cols, long long rows_per_sample, const float* mask, float scale, float* p, void* st) {
  TRAIN_GUARD("dhw_op_softmax", int, {
    OPCHECK(s && p && rows > 0 && cols > 0 && rows_per_sample > 0, "dhw_op_softmax");
    THIP(launch_softmax_fwd(s, rows, cols, rows_per_sample, mask, scale, p, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_softmax_bwd(const float* dp, const float* p, long long rows, int cols, float scale, float* ds, void* st) {
  TRAIN_GUARD("dhw_op_softmax_bwd", int, {
    OPCHECK(dp && p && ds && rows > 0 && cols > 0, "dhw_op_softmax_bwd");
    THIP(launch_softmax_bwd(dp, p, rows, cols, scale, ds, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_resample(int mode, const float* x, long long rows_out, int C, float* y, int accumulate, void* st) {
  TRAIN_GUARD("dhw_op_resample", int, {
    OPCHECK(x && y && rows_out > 0 && C > 0 && mode >= 0 && mode <= 3, "dhw_op_resample");
    THIP(launch_pool(mode, x, rows_out * C, C, y, accumulate, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_embedding(const int64_t* ids, const float* table, long long rows, int C, float* y, void* st) {
  TRAIN_GUARD("dhw_op_embedding", int, {
    OPCHECK(ids && table && y && rows > 0 && C > 0, "dhw_op_embedding");
    THIP(launch_embed(0, ids, table, rows * C, C, y, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_embedding_bwd(const int64_t* ids, const float* dy, long long rows, int C, float* dtable, void* st) {
  TRAIN_GUARD("dhw_op_embedding_bwd", int, {
    OPCHECK(ids && dy && dtable && rows > 0 && C > 0, "dhw_op_embedding_bwd");
    THIP(launch_embed(1, ids, dy, rows * C, C, dtable, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_mask_mul(const float* x, const float* mask, float scale, long long n, float* y, int accumulate, void* st) {
  TRAIN_GUARD("dhw_op_mask_mul", int, {
    OPCHECK(x && mask && y && n > 0, "dhw_op_mask_mul");
    THIP(launch_mask_mul(x, mask, scale, n, y, accumulate, (hipStream_t)st));
    return 0;
  });
}
int dhw_op_colsum(const float* dy, long long rows, int C, float* db, void* st) {
  TRAIN_GUARD("dhw_op_colsum", int, {
    OPCHECK(dy && db && rows > 0 && C > 0, "dhw_op_colsum");
    THIP(launch_colsum(dy, rows, C, db, (hipStream_t)st));
    return 0;
  });
}

}  // extern "C"
