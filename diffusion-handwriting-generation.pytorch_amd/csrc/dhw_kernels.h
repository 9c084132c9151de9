// dhw_kernels.h — host-side launch interface of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { PREC_BF16 = 0, PREC_F32 = 1 };

// ---------------------------------------------------------------- fused GEMM
// out[b, m, n] = epilogue( sum_seg sum_tap sum_c  act_seg[b, m + tap - halo, c] * W_seg[n][tap*C + c] )
// All activations are C-last [B, L, C] of the handle's element type.
struct GemmSeg {
  const void* A;   // activations [B*L (+slack), C]
  const void* W;   // packed weights: [N/16][taps*C/32][64 lanes][8] of the element type
  int C;           // input channels (multiple of 32)
  int taps;        // 1 (Linear) or 3 (Conv1d k=3, zero 'same' padding inside each sample)
  int silu;        // apply SiLU to the activations while staging
};

struct GemmParams {
  GemmSeg seg[2];
  int nseg;
  int B, L;              // samples, rows per sample
  int N;                 // output channels (multiple of 16)
  int n_store;           // columns [0,n_store) -> out (row stride n_store); [n_store,N) -> vt (transposed)
  const float* bias0;    // [N] added after segment 0
  const float* bias1;    // [N] added after segment 1 (nseg == 2) or null
  const float* posb;     // [>=L][posb_cols] position bias (PE·W), added for n < posb_cols; or null
  int posb_cols;
  const float* gam;      // FiLM gamma/beta for this layer: gam[(b / film_div) * film_bs + n]
  const float* bet;
  long film_bs;          // row stride of the FiLM table (0 inside the sampling loop; may be negative)
  int film_div;          // samples sharing one FiLM row (1 = per-sample sigma; B = batched all-steps text pass)
  int film_mode;         // 0 none; 1 after (LN); 2 between segment 0 and 1 (ConvBlock: FiLM3(fc)+conv_skip)
  const void* res1;      // [B*L, N] added before LN, or null
  const void* res2;      // added after FiLM: [B*L, N], or [B*L/2, N] when res2_half (nearest x2 upsample)
  int res2_half;
  int ln;                // LayerNorm over the N channels (eps 1e-6, no affine); needs BN == N
  int ln_n;              // channels the statistics run over: columns [ln_n, N) are zero padding (model widths below the kernels'
                         // physical ones, dhw_api.cpp pad_weights) and are written as 0; 0 = N
  float ln_inv;          // 1 / ln_n (filled in by launch_gemm)
  int silu_out;
  int relu6_out;         // clamp to [0, 6] (MobileNetV2 pointwise convolutions of the StyleExtractor, style.hip)
  void* out;             // [B*L, n_store] element type, or fp32 when out_f32
  int out_f32;
  void* pool;            // optional second output: AvgPool1d(2) over rows, [B*L/2, N]
  void* vt;              // [B][N-n_store][vt_lpad] element type (keys contiguous)
  int vt_lpad;
  unsigned long long* stamps;   // diagnostics only: phase times (s_memrealtime) of the middle workgroup, or null
};

// returns hipError; chooses the tile from (prec, L, N, ln)
hipError_t launch_gemm(int prec, const GemmParams& p, hipStream_t st);
// tile actually chosen (for tests / work accounting)
void gemm_tile_for(int prec, const GemmParams& p, int* BM, int* BN);
hipError_t gemm_init();


// heads (+ optional fused scheduler step).  x: fp32 [rows, C] (dec1 output).
struct HeadsParams {
  const float* x; long rows; int C;
  const float* w_out; const float* b_out;   // [2,C],[2]
  const float* w_pen; const float* b_pen;   // [1,C],[1]
  float* eps; float* pen;                   // [rows,2], [rows]  (pen may be null)
  // fused scheduler step (null xt = none): xt <- step(xt, eps, z), reference operation order
  float* xt;               // [rows,2] fp32 sampler state, updated in place
  const float* z;          // [rows,2] external noise or null (=> Philox)
  int mode;                // 0 new (utils/nn.py:110-112), 1 standard (utils/nn.py:84-87)
  int add_noise;           // standard mode: bool(i) (inference.py:92); new mode: always 1
  float k0;                // sqrt(1 - abar_i)
  float k1;                // new: sqrt(1 - beta_i);  standard: 1 / sqrt(1 - beta_i)
  float k2;                // new: sqrt(1 - abar_next);  standard: sqrt(beta_i)
  float k3;                // standard: beta_i
  const uint64_t* seed_ptr; int sample_off; int L; int iter;   // Philox: device [seed, first_sample]; counter = (sample, pos, iter)
  float* out3;             // optional [rows,3] final cat(x, pen)
};

// ---------------------------------------------------------------- fused ConvBlock (cnn.py:64-87), one launch
struct ConvBlockParams {
  const void* x;                      // block input [B*L, Cin]
  const float* strokes;               // enc1 only (or null): x = in_w·strokes + in_b is evaluated while staging (model.py:139)
  const float *in_w, *in_b;           // input_dense weight [Cin,2], bias [Cin]
  // decoder blocks (or null): x = Upsample(up_low) + skip_conv(up_h) (model.py:169-175) is evaluated while staging:
  const void* up_h; int up_cin;       //   skip-connection activation [B*L, up_cin]
  const void* up_w; const float* up_b;   // skip_conv weight packed as a 3-tap GEMM segment [Cin x 3*up_cin], bias [Cin]
  const void* up_low;                 //   half-resolution decoder state [B*L/2, Cin]
  int B, L, Cin, Cout;
  const void *w_c1, *w_c2, *w_fc, *w_skip;     // packed as for the GEMM kernel
  const float *b_c1, *b_c2, *b_fc, *b_skip;
  const float* film; long film_bs; int film_tot;   // gamma row = film + b*film_bs, beta row = gamma + film_tot
  int f1, f2, f3;                     // FiLM offsets of affine1..3
  void* out; int out_f32;             // [B*L, Cout]
  void* pool;                         // optional AvgPool1d(2) side output [B*L/2, Cout]
  int fuse_heads;                     // dec1 in the sampling loop: evaluate the eps/pen heads + scheduler step from the fp32
  HeadsParams hp;                     //   output tile in LDS (hp.x/rows/C unused); `out` may then be null (no activation write)
  unsigned long long* stamps;         // diagnostics only: per-stage s_memrealtime of workgroup 0, or null
  int stagger;                        // two-workgroups-per-CU variants: the second half of the grid starts this many x 0.5 us late
};
hipError_t launch_convblock(int prec, const ConvBlockParams& p, hipStream_t st);
hipError_t convblock_init();

// ---------------------------------------------------------------- fused EncoderLayer stroke side (model.py:37-58), two launches
struct EncLayerParams {
  int B, Lk, Lt, d, heads;
  const void* x;                                   // layer input [B*Lk, d]
  const void *w_q1, *w_d1, *w_qkv2, *w_d2, *w_f1, *w_f2;   // packed as for the GEMM kernel
  const float *b_q1, *b_d1, *b_qkv2, *b_d2, *b_f1, *b_f2;
  const float *pb_q1, *pb_qk2;                     // PE·W tables [>=Lk][d], [>=Lk][2d]
  const float* film; long film_bs; int film_tot; int f1, f2, f3;
  const void* k1; const void* vt1; int lpadT;      // text keys [B*Lt, d]; values: bf16 kernels v1 [B*Lt, d] (rows, like k1), fp32 V^T [B][d][lpadT]
  const int64_t* text;                             // key padding mask source [B, Lt] (0 = pad)
  void* x2;                                        // [B*Lk, d]     written by enc_a, read by enc_bc
  void* qk2; void* vt2; int lpadX;                 // bf16 kernels: [q2 | k2 | v2] rows [B*Lk, 3d] (vt2 unused); fp32: [B*Lk, 2d] and V^T [B][d][lpadX]
  void* out; void* pool;                           // [B*Lk, d], optional [B*Lk/2, d]
  int bm_min;                                      // 0, or the smallest row tile to use (32 when enc_bc chains into a pooled layer)
  int dbg;                                         // diagnostics only: bit0 = skip the attention stage (a = q)
  unsigned long long* stamps;                      // diagnostics only: per-stage s_memrealtime of workgroup 0 (16 slots per kernel) or null
};
// What an enc_bc (or ConvBlock) workgroup goes on to compute for its own rows after its own block — the stages up to the
// next self-attention are row-local, so they need no launch boundary:
//   mode 1: enc_a of the next EncoderLayer on the block's output tile (same width);
//   mode 2: AvgPool1d(2) -> Linear (att_dense, model.py:160-162) -> enc_a of the first attention layer.
struct EncChain {
  int mode;                  // 0 = none
  EncLayerParams a;          // the next layer (a.x unused: the tile is already in LDS)
  const void* w_dense; const float* b_dense;   // mode 2: packed weight [a.d x d], bias [a.d]
  void* dense_out;           // mode 2: Linear output [B*Lk/2, a.d] (kept for debug taps)
};
// A ConvBlock whose output feeds an EncoderLayer continues into that layer's enc_a on its own output tile (mode 1).
bool convblock_chain_supported(int prec, const ConvBlockParams& p, const EncChain& chain);
hipError_t launch_convblock_chain(int prec, const ConvBlockParams& p, const EncChain& chain, hipStream_t st);
bool convblock_chain_auto(const ConvBlockParams& p);   // the chain is a measured win for this launch geometry (enc4 on the asymmetric 32-row tiles)
bool enclayer_supported(int prec, int d, int heads);
// whether enc_bc of a (d, B, Lk) layer can continue with `mode`; mode 2 needs EncLayerParams.bm_min = 32 on that layer
bool enclayer_chain_supported(int prec, int d, int B, int Lk, int mode, int d_next);
hipError_t launch_enclayer(int prec, const EncLayerParams& p, int which, hipStream_t st, const EncChain* chain = nullptr);
hipError_t enclayer_init();

// ---------------------------------------------------------------- attention
struct AttnParams {
  const void* Q; int ldq;      // Q[(b*Lq + q)*ldq + h*D + d]
  const void* K; int ldk; int koff;  // K[(b*Lk + k)*ldk + koff + h*D + d]
  const void* Vt; int lpad;    // Vt[((b*H + h)*D + d)*lpad + k]
  const int64_t* text; int ldt; // key padding mask: key k masked (score += -1e9) iff text[b*ldt + k] == 0; null = none
  void* out; int ldo;          // out[(b*Lq + q)*ldo + h*D + d]
  int B, H, D, Lq, Lk;
};
hipError_t launch_attn(int prec, const AttnParams& p, hipStream_t st);

// ---------------------------------------------------------------- small kernels
// sigma_ffn: sig32[n,32] = W2 SiLU(W1 SiLU(sigma) + b1) + b2        (model.py:83,134)
hipError_t launch_sigma_ffn(const float* sigma, int n, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* sig32, hipStream_t st);
// FiLM table: film[n, cols] = sig32[n,32] · Wcat[cols,32]^T + bcat   (conditioning.py:16-18, all layers at once)
hipError_t launch_film(const float* sig32, int n, const float* wcat, const float* bcat, int cols,
                       float* film, hipStream_t st);
// t_n = LN(emb[text])  (text_style.py:96-97), element type out
// (n_true: channels the statistics run over; the table's columns [n_true, dim) are zero padding and stay 0)
hipError_t launch_embed_ln(int prec, const int64_t* text, int rows, const float* emb, int dim, int n_true, int vocab,
                           void* out, hipStream_t st);
// out[b, r, c] = in[b % in_B, r, c] * gam[(b / div)*bs + c] + bet[(b / div)*bs + c]
hipError_t launch_film_apply(int prec, const void* in, int in_B, int B, int rows, int dim, const float* gam,
                             const float* bet, long bs, int div, void* out, hipStream_t st);
// fp32 -> element type
hipError_t launch_cast(int prec, const float* in, long n, void* out, hipStream_t st);
// x0[b,l,:] = W[:,0]*s0 + W[:,1]*s1 + bias   (model.py:139)
hipError_t launch_input_dense(int prec, const float* strokes, long rows, const float* w, const float* b,
                              int C, void* out, hipStream_t st);

hipError_t launch_heads(const HeadsParams& p, hipStream_t st);
// x_T ~ N(0,1) from Philox, same keying as the per-step draws (iter = -1)
hipError_t launch_randn_init(float* xt, long rows, int L, const uint64_t* seed_ptr, int sample_off, hipStream_t st, int iter = -1);
// seed_ptr[0] = seed, seed_ptr[1] = first_sample (by-value kernel arguments: no host buffer lifetime)
hipError_t launch_set_seed(uint64_t* seed_ptr, uint64_t seed, int64_t first_sample, hipStream_t st);
// one-time per-process kernel attribute setup (dynamic LDS > 64 KiB)
hipError_t gemm_init();

// ---------------------------------------------------------------- StyleExtractor spatial kernels (style.hip): NHWC, channels padded to Cp
hipError_t launch_style_stem(int prec, const float* img, int B, int H, int W, const float* w, const float* bias, int Cp,
                             void* out, hipStream_t st);   // -> [B, ceil(H/2), ceil(W/2), Cp]
hipError_t launch_style_dw(int prec, const void* in, int B, int H, int W, int stride, const float* w, const float* bias,
                           int Cp, void* out, hipStream_t st);
hipError_t launch_style_pool(int prec, const void* in, int B, int H, int W, int C, int NB, float* out, hipStream_t st);

// ---------------------------------------------------------------- fused text side (textside.hip): one workgroup per (step, prompt) pair
struct TextStyleParams {
  int n;                  // (sampler step, prompt) pairs of this launch
  int in_B;               // prompts behind the sigma-independent inputs: pair i reads prompt i % in_B
  int Lt, S5;             // tokens (<= 32), style rows (<= 80)
  const void* sty_n;      // LN(style_ffn(style))  [in_B][S5][384]
  const void* t_n;        // LN(emb(text))         [in_B][Lt][384]
  const float* film; long film_bs; int film_div; int film_tot;   // gamma row of pair i = film + (i / film_div) * film_bs, beta = gamma + film_tot
  int f1, f2, f3, f4;     // FiLM offsets of text_style_model.affine1..4
  const void *w_q8, *w_kv8, *w_d8, *w_tf1, *w_tf3;   // packed as for the GEMM kernel ([384x384], [768x384] K|V, [384x384], [768x384], [384x768])
  const float *b_q8, *b_kv8, *b_d8, *b_tf1, *b_tf3;
  void* text_out;         // [n][Lt][384]
};
struct TextLayerParams {
  int n, Lt, d;           // pairs, tokens (<= 32), layer width (192 / 256 / 384)
  const void* text_out;   // [n][Lt][384]
  const void* w_td; const float* b_td;       // text_dense [d x 384]
  const float* film; long film_bs; int film_div; int film_tot; int f0;
  const void* w_kv; const float* b_kv;       // stacked K|V projection of the cross attention [2d x d]
  const float* pb_k1;     // PE·Wk table [>= Lt][d]
  void* k1;               // [n][Lt][d]
  void* vt1; int lpadT;   // v1 [n][Lt][d], rows like k1 (lpadT unused)
  int pairs;              // pairs per workgroup: 0 = the launcher's choice, 1, 2 (2 needs an even film_div: textside.hip)
};
bool textside_supported(int prec, int Lt, int S5, int dt);
hipError_t launch_text_style(int prec, const TextStyleParams& p, hipStream_t st);
hipError_t launch_text_layer(int prec, const TextLayerParams& p, hipStream_t st);
hipError_t textside_init();

// ---------------------------------------------------------------- training step, first slice (train.hip); fp32, rows C-last
hipError_t launch_perturb(const float* x, const float* eps, const float* alphas, int B, int L, float* out, hipStream_t st);
hipError_t launch_loss(const float* eps, const float* pred, const float* pen, const float* pen_pred, const float* alphas, int B, int L,
                       float* out3, float* d_pred, float* d_pen, hipStream_t st);
hipError_t launch_sqnorm(const float* g, long n, float* out, hipStream_t st);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                       const float* sqnorm, float max_norm, hipStream_t st);
hipError_t launch_film_silu_fwd(const float* u, const float* film, long film_bs, int goff, int boff, int B, int L, int C, float* a, float* h, hipStream_t st);
hipError_t launch_film_bwd(const float* d, const float* a, const float* u, const float* film, long film_bs, int goff, int B, int L, int C, int act,
                           float* du, float* dfilm, long dfilm_bs, int dgoff, int dboff, hipStream_t st);
hipError_t launch_silu_bwd_add(const float* d_sx, const float* x, long n, float* dx, hipStream_t st);
hipError_t launch_silu_fwd(const float* x, long n, float* out, hipStream_t st);
hipError_t launch_add(const float* a, const float* b, long n, float* out, hipStream_t st);
hipError_t launch_colsum(const float* dy, long rows, int C, float* db, hipStream_t st);
hipError_t launch_wgrad(const float* dy, const float* x, int B, int L, int Cout, int Cin, int taps, float* dw, hipStream_t st);
hipError_t launch_film_linear_bwd(const float* dfilm, const float* sigma, const float* wcat, int B, int cols, float* dw, float* db, float* dsigma, hipStream_t st);

// generic fp32 ops of the training step (train.hip): see include/dhw_train.h (dhw_op_*)
struct OpGemm {
  const float* A; long sam, sak, sazo, sazi; int a_shift, a_tap_shift;
  const float* B; long sbk, sbn, sbzo, sbzi, sbt; int b_shift, b_z_shift;
  float* C; long scm, scn, sczo, sczi;
  int M, N, K, nzo, nzi, lr, taps;
  const float* bias; float alpha; int accumulate, bf16;
  const float* dsilu_of; // or null: the product is multiplied by SiLU'(dsilu_of[m][n]) (C's layout) before it is written / added: the data gradient
                         // of a Linear / Conv1d whose input was SiLU(u) lands directly in du
  float* act_out;        // or null: SiLU of the value written to C, same layout (the activation that follows a Linear / Conv1d; not with accumulate / split-K)
  const float* addend;   // or null: C = alpha A B + bias + addend (C's layout; a residual add in the output pass; no split-K then)
  float* rowsum;   // or null: rowsum[m] += sum_k A(0,m,k) (batch z = 0) — a Linear / Conv1d bias gradient out of its weight-gradient GEMM
  unsigned long long* stamps;   // diagnostics only (tools/bench_sgemm.cpp, -DDHW_STAMPS builds): s_memrealtime of one workgroup's phases, or null
  // FiLM (+ SiLU) (+ addend) of the value written to C as a further output (film_out null: none); dhw_gemm_desc in include/dhw_train.h
  const float* film_g = nullptr; const float* film_b = nullptr; long film_ps = 0; int film_rows = 1, film_act = 0; float* film_out = nullptr; const float* film_add = nullptr;
};
hipError_t launch_sgemm(const OpGemm& g, hipStream_t st);
hipError_t launch_sgemm_pair(const OpGemm& a, const OpGemm& b, hipStream_t st, int* launches = nullptr);   // two independent GEMMs, one launch where possible
hipError_t launch_sgemm_group(const OpGemm* g, int n, hipStream_t st, int* launches = nullptr);            // up to 6 independent GEMMs, one launch where possible
hipError_t launch_film_table(int dir, const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int TOT, float* film,
                             float* gflat, float* dsigma, hipStream_t st);
hipError_t launch_keep_mask(const uint64_t* rng, int site, long n, int per_sample, float p, float* keep, hipStream_t st);
hipError_t launch_train_draw(const uint64_t* rng, int B, int L, float* eps, long n_keep, int keep_per_sample, float p, float* keep, hipStream_t st);
hipError_t launch_adam_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, const float* sqnorm, hipStream_t st);
hipError_t launch_unary(int kind, const float* x, long n, float* y, hipStream_t st);
hipError_t launch_unary_bwd(int kind, const float* dy, const float* x, long n, float* dx, int accumulate, hipStream_t st);
hipError_t launch_add2(const float* a, const float* b, long n, float* out, int accumulate, hipStream_t st);
hipError_t launch_add_rows(const float* x, const float* table, long n, long per_sample, float* out, hipStream_t st);
hipError_t launch_film_fwd(const float* x, const float* gam, const float* bet, long pstride, int B, int L, int C, float* y, hipStream_t st);
hipError_t launch_film_bwd2(const float* d, const float* u, const float* gam, long pstride, int B, int L, int C, float* du, int accumulate, float* dgam,
                            float* dbet, hipStream_t st);
hipError_t launch_film_act_fwd(const float* x, const float* gam, const float* bet, long pstride, int B, int L, int C, int act, const float* addend, float* y,
                               hipStream_t st);
hipError_t launch_film_act_bwd(const float* d, const float* u, const float* gam, const float* bet, long pstride, int B, int L, int C, int act, float* du,
                               int accumulate, float* dgam, float* dbet, hipStream_t st);
hipError_t launch_ln_film_fwd(const float* x, long rows, int C, const float* gam, const float* bet, long pstride, int L, const float* addend, float* y,
                              float* act_out, const float* pe, float* pe_out, float* mean, float* rstd, hipStream_t st);
hipError_t launch_ln_film_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gam, long pstride, int B, int L, int C,
                              float* dx, int accumulate, float* dgam, float* dbet, hipStream_t st);
hipError_t launch_ln_fwd(const float* x, long rows, int C, float* y, float* mean, float* rstd, hipStream_t st);
hipError_t launch_ln_bwd(const float* dy, const float* y, const float* rstd, long rows, int C, float* dx, int accumulate, hipStream_t st);
hipError_t launch_softmax_fwd(const float* s, long rows, int cols, long rows_per_sample, const float* mask, float scale, float* p, hipStream_t st);
hipError_t launch_softmax_bwd(const float* dp, const float* p, long rows, int cols, float scale, float* ds, hipStream_t st);
hipError_t launch_pool(int mode, const float* x, long n_out, int C, float* y, int accumulate, hipStream_t st);
hipError_t launch_embed(int bwd, const int64_t* ids, const float* src, long n, int C, float* dst, hipStream_t st);
hipError_t launch_mask_mul(const float* x, const float* mask, float scale, long n, float* y, int accumulate, hipStream_t st);
