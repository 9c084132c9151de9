/* dhw_train.h — C-ABI of the first slice of the training step in libdhw_hip.so (BASELINE configs[4]).
 *
 * What of the reference's train_step (train.py:26-67) runs here as hand-written HIP, each piece pinned by a fixture the
 * imported reference generated (oracle/make_golden_r2.py):
 *
 *   dhw_train_perturb    <- x_perturbed = sqrt(abar) x + sqrt(1 - abar) eps                      (train.py:41-43)
 *   dhw_train_loss       <- loss_fn (loss.py:5-37) and d loss / d(score_pred, pen_lifts_pred)     (loss.backward())
 *   dhw_train_adam       <- clip_grad_norm_(max_norm) (utils/clip_grad.py:42-43) + torch.optim.Adam (configs/best.yml:33-38)
 *   dhw_train_convblock  <- ConvBlock.forward + its autograd backward (cnn.py:64-87): the first block of the denoiser's
 *                           backward pass as ONE fused entry point (weights packed per call).
 *   dhw_op_*             <- the generic fp32 operations from which dhg_amd/train_model.py assembles the WHOLE
 *                           DiffusionModel forward (train mode) and backward — every parameter gradient of the
 *                           reference's loss.backward() (train.py:55-60).
 *
 * fp32; activations are C-last [B*L, C] device buffers; weights of dhw_train_convblock are HOST pointers in torch layouts
 * (packed per call: this entry point is a gradient-parity vehicle, not yet a tuned trainer).
 */
#ifndef DHW_TRAIN_H
#define DHW_TRAIN_H

#include "dhw.h"

#ifdef __cplusplus
extern "C" {
#endif

/* x, eps, out: f32 [B,L,2]; alphas: f32 [B] */
int dhw_train_perturb(const float* x, const float* eps, const float* alphas, int B, int L, float* out, void* hip_stream);

/* eps, score_pred: [B,L,2]; pen, pen_pred: [B,L]; alphas: [B].  out3 (device) = {loss, score_loss, pen_lifts_loss};
 * d_score [B,L,2] and d_pen_pred [B,L] (either may be NULL) = gradients of out3[0]. */
int dhw_train_loss(const float* eps, const float* score_pred, const float* pen, const float* pen_pred, const float* alphas,
                   int B, int L, float* out3, float* d_score, float* d_pen_pred, void* hip_stream);

/* One optimizer step on `nbuf` flat parameter buffers: global-norm clip (max_norm <= 0: none) over ALL of them, then Adam.
 * p/g/m/v: arrays of device pointers, n: element counts; step = 1 for the first update (bias correction).
 * The squared gradient norm goes through ONE scratch scalar per device that the library allocates at the first call on that
 * device and keeps: (a) that first call must not be made under stream capture, (b) calls on different streams of one device
 * must not overlap.  dhw_train_adam_dev below has neither constraint (the caller owns the scalar). */
int dhw_train_adam(int nbuf, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                   float* grad_norm_out /* device, 1 float, or NULL */, void* hip_stream);

/* The same update with nothing on the host's critical path, for capture into a hipGraph: the scalars come from DEVICE memory
 * hyper[9] = {lr, beta1, beta2, eps, weight_decay, 1 - beta1^step, 1 - beta2^step, max_norm (<= 0: no clipping), grad_scale} and the
 * squared global norm of the gradient BUFFER is left in sqnorm (device, 1 float).  grad_scale: the gradient is grad_scale * g — 1 /
 * world_size after a SUM all-reduce, so that no separate pass divides the buffer (norm and clip use the scaled gradient).  No
 * allocation, no synchronisation. */
int dhw_train_adam_dev(int nbuf, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                       const float* hyper, float* sqnorm, void* hip_stream);

/* The random draws of one update on the device, from the sampler's counter-based Philox generator (dhw.h: the same draw for
 * any batch sharding): eps ~ N(0,1) [B,L,2] (train.py:39) and the Dropout(p) keep-mask (1 = kept) over n_keep elements,
 * keep_per_sample of them per batch sample (text_style.py:97: p = 0.3 on [B, S, 1280]).  rng: DEVICE uint64[2] = {seed,
 * draw index}; sample b of draw d is stream d * B + b, so ranks pass disjoint draw indices.  Capturable. */
int dhw_train_draw(const uint64_t* rng, int B, int L, float* eps, long long n_keep, int keep_per_sample, float p, float* keep, void* hip_stream);

typedef struct {   /* HOST pointers, torch layouts */
  const float *conv1_w, *conv1_b;   /* [C/2, Cin, 3], [C/2] */
  const float *conv2_w, *conv2_b;   /* [C, C/2, 3], [C] */
  const float *fc_w, *fc_b;         /* [C, C], [C] */
  const float *skip_w, *skip_b;     /* [C, Cin, 3], [C] */
  const float *film_w, *film_b;     /* the six 32 -> c Linears stacked as rows gamma1|gamma2|gamma3|beta1|beta2|beta3: [2(C/2+2C), 32], [2(C/2+2C)] */
} dhw_convblock_weights;

typedef struct {   /* DEVICE pointers, the same layouts */
  float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *fc_w, *fc_b, *skip_w, *skip_b, *film_w, *film_b;
} dhw_convblock_grads;

/* out = ConvBlock(x, sigma) and, for the upstream gradient dout, dx, dsigma and every parameter gradient.
 * x [B*L,Cin], sigma [B,32], dout / out [B*L,C], dx [B*L,Cin], dsigma [B,32]: device, f32.  L even, Cin and C/2 multiples of 32. */
int dhw_train_convblock(int device, int B, int L, int cin, int cout, const float* x, const float* sigma, const float* dout,
                        const dhw_convblock_weights* w, float* out, float* dx, float* dsigma, const dhw_convblock_grads* g,
                        void* hip_stream);

/* ---- generic fp32 operations of the training step (DiffusionModel.forward in train mode + its autograd backward) -------
 * The host side (dhg_amd/train_model.py) chains these the way autograd chains the reference's modules; every pointer is a
 * DEVICE pointer to f32 (ids: int64), activations are C-last rows [B*L, C], weights stay in torch layouts (no packing:
 * the GEMM takes strides).  `accumulate` != 0 adds into the destination (gradient fan-in) instead of overwriting it. */

typedef struct {
  /* C[z][m][n] (+)= alpha * sum_k A(z,m,k) B(z,k,n) (+ bias[n]);  z = zo * nzi + zi  (two batch levels, e.g. sample x head),
   * K = taps * Kt and k = tap * Kt + kk  (taps = 1: plain GEMM; taps = 3: the three Conv1d taps in one contraction):
   *   A(z,m,k) = A[zo*sazo + zi*sazi + (m + sa)*sam + kk*sak],  sa = a_shift + tap*a_tap_shift,
   *              taken as 0 unless (m mod lr) + sa is in [0, lr)
   *   B(z,k,n) = B[zo*sbzo + zi*sbzi + tap*sbt + (kk + sb)*sbk + n*sbn],  sb = b_shift + zi*b_z_shift,
   *              taken as 0 unless (kk mod lr) + sb is in [0, lr)
   * lr = rows per sample for the Conv1d taps ('same' zero padding inside each sample; 0 = no shifting).  One description
   * covers nn.Linear / Conv1d forward, data gradient and weight gradient and the per-head attention products.
   * taps > 1 needs Kt to be a multiple of 32. */
  const float* A; long long sam, sak, sazo, sazi; int a_shift, a_tap_shift;
  const float* B; long long sbk, sbn, sbzo, sbzi, sbt; int b_shift, b_z_shift;
  float* C; long long scm, scn, sczo, sczi;
  int M, N, K, nzo, nzi, lr, taps;
  const float* bias; float alpha; int accumulate;
  int bf16;   /* 0: exact-f32 MFMA.  1: the operands (fp32 in memory) are rounded to bf16 on their way into LDS and contracted on
                 the bf16 MFMA with fp32 accumulation — mixed-precision training with fp32 master weights */
  float* act_out;        /* NULL, or an array laid out as C that receives SiLU(value written to C): the activation that follows the Linear /
                            Conv1d in ff_network / ConvBlock, written in the same pass (accumulate must be 0) */
  const float* addend;   /* NULL, or an array laid out as C: C = alpha A B + bias + addend — a residual add in the GEMM's output pass */
  const float* dsilu_of; /* NULL, or an array laid out as C: the product is multiplied by SiLU'(dsilu_of) before it is written / added — the
                            data gradient of a Linear / Conv1d whose input was SiLU(u) goes straight into du (no bias / addend with it) */
  float* rowsum;   /* NULL, or [M]: rowsum[m] += sum_k A(0,m,k) (batch z = 0 only) — the bias gradient of a Linear / Conv1d comes out
                      of its weight-gradient GEMM (A = dy^T) instead of a second pass over dy (dhw_op_colsum).  The sums are taken from
                      the operand tile as staged for the MFMA: with bf16 = 1 that is dy ROUNDED to bf16 (fp32 accumulation), i.e. the
                      bias gradient carries the same operand rounding as the weight gradient of its layer (relative 2^-9 per term,
                      sqrt(K)-averaged); with bf16 = 0 it is exact fp32.  act_out / addend / dsilu_of are applied in fp32 either way. */
  /* FiLM (+ SiLU) of the value written to C as a further output of the same pass (appended in round 4; film_out = NULL: none):
       film_out[m][n] = act(film_gamma[(m / film_rows) * film_pstride + n] * c + film_beta[...]) + film_addend[m][n]
     with c the value written to C, act = SiLU when film_act, film_addend NULL or laid out as C — a ConvBlock's
     SiLU(affine(conv(.))) / affine3(fc(.)) + conv_skip(.) (cnn.py:70-86) without a pass of its own.  accumulate must be 0, and the
     GEMM must be UNBATCHED (nzo * nzi = 1: film_out / film_addend carry no batch offset) — a batched descriptor with film_out is
     rejected with DHW_ERR_ARG by dhw_op_gemm / dhw_op_gemm2 / dhw_op_gemm_group. */
  const float* film_gamma; const float* film_beta; long long film_pstride; int film_rows; int film_act;
  float* film_out; const float* film_addend;
} dhw_gemm_desc;

int dhw_op_gemm(const dhw_gemm_desc* g, void* hip_stream);
/* two INDEPENDENT GEMMs (neither reads what the other writes, distinct outputs) as one launch where their forms allow it — a
   layer's weight gradient (first) and data gradient (second), which both read dy — otherwise as two launches; same results as
   two dhw_op_gemm calls.  Returns the number of launches issued (1 or 2), or a negative error code */
int dhw_op_gemm2(const dhw_gemm_desc* g0, const dhw_gemm_desc* g1, void* hip_stream);
/* n <= 6 INDEPENDENT GEMMs (g[0..n)), dispatched in that order: one launch where every one is an fp32 16-byte-load form, else n;
   returns the number of launches issued, or a negative error code */
int dhw_op_gemm_group(const dhw_gemm_desc* g, int n, void* hip_stream);
/* kind 0: SiLU, 1: sigmoid.  Backward: kind 0 takes the forward INPUT x, kind 1 the forward OUTPUT y. */
int dhw_op_unary(int kind, const float* x, long long n, float* y, void* hip_stream);
int dhw_op_unary_bwd(int kind, const float* dy, const float* x_or_y, long long n, float* dx, int accumulate, void* hip_stream);
/* out (+)= a + b   (b may be NULL) */
int dhw_op_add(const float* a, const float* b, long long n, float* out, int accumulate, void* hip_stream);
/* out[b][l][c] = x[b][l][c] + table[l][c]: the constant positional encodings (model.py:48, text_style.py:56) */
int dhw_op_add_rows(const float* x, const float* table, int B, int L, int C, float* out, void* hip_stream);
/* AffineTransformLayer (conditioning.py:20-26) given its gamma / beta rows [B][C] (row stride pstride) */
int dhw_op_film(const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, float* y, void* hip_stream);
int dhw_op_film_bwd(const float* dy, const float* x, const float* gamma, long long pstride, int B, int L, int C, float* dx, int accumulate,
                    float* dgamma, float* dbeta /* += */, void* hip_stream);
/* nn.LayerNorm(C, eps=1e-6, elementwise_affine=False) over each row; rstd [rows] is kept for the backward, which takes y */
/* Fused element-wise chains (one HBM pass each way; the intermediate is recomputed in the backward):
 *   film_act : y = act ? SiLU(x gamma[b] + beta[b]) : x gamma[b] + beta[b]     (conditioning.py:23-26 [+ the SiLU that follows in cnn.py:70-80])
 *   ln_film  : y = LayerNorm(x) gamma[b] + beta[b]                             (model.py:25 + conditioning.py:23-26, as every EncoderLayer chains them)
 * Both take an optional addend (same shape as y): y += addend, the residual add that follows them in ConvBlock / EncoderLayer; its gradient is dy.
 * gamma / beta: per-sample rows [B][pstride]; the backward ADDS into dgamma / dbeta (same layout) and writes or adds dx. */
int dhw_op_film_act(const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, int act, const float* addend /* or NULL */,
                    float* y, void* hip_stream);
int dhw_op_film_act_bwd(const float* dy, const float* x, const float* gamma, const float* beta, long long pstride, int B, int L, int C, int act, float* dx,
                        int accumulate, float* dgamma, float* dbeta, void* hip_stream);
int dhw_op_ln_film(const float* x, int B, int L, int C, const float* gamma, const float* beta, long long pstride, const float* addend /* or NULL */, float* y,
                   float* act_out /* or NULL: SiLU(y) */, const float* pe /* [L][C] */, float* pe_out /* or NULL: y + pe[l] */, float* mean, float* rstd,
                   void* hip_stream);
int dhw_op_ln_film_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, long long pstride, int B, int L, int C,
                       float* dx, int accumulate, float* dgamma, float* dbeta, void* hip_stream);
int dhw_op_layernorm(const float* x, long long rows, int C, float* y, float* mean, float* rstd, void* hip_stream);
int dhw_op_layernorm_bwd(const float* dy, const float* y, const float* rstd, long long rows, int C, float* dx, int accumulate, void* hip_stream);
/* attention.py:16-22: P = softmax(S * scale + mask * -1e9) over `cols` keys; rows = B*H*Lq, mask [B][cols] or NULL */
int dhw_op_softmax(const float* s, long long rows, int cols, long long rows_per_sample, const float* mask, float scale, float* p, void* hip_stream);
int dhw_op_softmax_bwd(const float* dp, const float* p, long long rows, int cols, float scale, float* ds, void* hip_stream);
/* mode 0: AvgPool1d(2) forward (rows_out = rows_in / 2), 1: its backward (rows_out = 2 rows_in), 2: nearest x2 Upsample
 * forward (rows_out = 2 rows_in), 3: its backward (rows_out = rows_in / 2).  L is even, so sample boundaries are kept. */
int dhw_op_resample(int mode, const float* x, long long rows_out, int C, float* y, int accumulate, void* hip_stream);
/* nn.Embedding forward (rows ids -> [rows, C]) and backward (dtable += scatter of dy) */
int dhw_op_embedding(const int64_t* ids, const float* table, long long rows, int C, float* y, void* hip_stream);
int dhw_op_embedding_bwd(const int64_t* ids, const float* dy, long long rows, int C, float* dtable, void* hip_stream);
/* y (+)= x * mask * scale: nn.Dropout with a supplied keep-mask (scale = 1 / (1 - p)), and its backward */
int dhw_op_mask_mul(const float* x, const float* mask, float scale, long long n, float* y, int accumulate, void* hip_stream);
/* Every AffineTransformLayer's gamma / beta Linear (conditioning.py:16-18) in one launch: film[b][j] = flat[boff[j]] + sum_k
 * sigma[b][k] flat[woff[j] + k] for the `total` output channels j of all the Linears; woff / boff (device int64[total]) give
 * each channel's weight row and bias inside the flat parameter buffer, so the parameters stay in state_dict order.  Backward:
 * grad_flat[woff[j] + k] += sum_b dfilm[b][j] sigma[b][k], grad_flat[boff[j]] += sum_b dfilm[b][j], dsigma[b][k] += sum_j
 * dfilm[b][j] flat[woff[j] + k].  sigma / dsigma [B,32], film / dfilm [B,total]. */
int dhw_op_film_table(const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int total, float* film, void* hip_stream);
int dhw_op_film_table_bwd(const float* dfilm, const float* sigma, const float* flat, const int64_t* woff, const int64_t* boff, int B, int total,
                          float* grad_flat, float* dsigma, void* hip_stream);
/* A Dropout(p) keep-mask drawn on the device for dropout site number `site` >= 3 of the model (EncoderLayer.drop, model.py:23,
 * 47-56); rng as in dhw_train_draw (sites 1 and 2 are its eps and style mask), per_sample elements per batch sample. */
int dhw_op_keep_mask(const uint64_t* rng, int site, long long n, int per_sample, float p, float* keep, void* hip_stream);
/* db[c] += sum over rows of dy[r][c]  (bias gradients) */
int dhw_op_colsum(const float* dy, long long rows, int C, float* db, void* hip_stream);

const char* dhw_train_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DHW_TRAIN_H */
