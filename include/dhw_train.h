/* dhw_train.h — C-ABI of the first slice of the training step in libdhw_hip.so (BASELINE configs[4]).
 *
 * What of the reference's train_step (train.py:26-67) runs here as hand-written HIP, each piece pinned by a fixture the
 * imported reference generated (oracle/make_golden_r2.py):
 *
 *   dhw_train_perturb    <- x_perturbed = sqrt(abar) x + sqrt(1 - abar) eps                      (train.py:41-43)
 *   dhw_train_loss       <- loss_fn (loss.py:5-37) and d loss / d(score_pred, pen_lifts_pred)     (loss.backward())
 *   dhw_train_adam       <- clip_grad_norm_(max_norm) (utils/clip_grad.py:42-43) + torch.optim.Adam (configs/best.yml:33-38)
 *   dhw_train_convblock  <- ConvBlock.forward + its autograd backward (cnn.py:64-87): the first block of the denoiser's
 *                           backward pass; the EncoderLayer / TextStyleEncoder backward kernels are not built yet, so a
 *                           whole train_step cannot run natively — see DESIGN.md.
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
 * p/g/m/v: arrays of device pointers, n: element counts; step = 1 for the first update (bias correction). */
int dhw_train_adam(int nbuf, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_norm,
                   float* grad_norm_out /* device, 1 float, or NULL */, void* hip_stream);

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

const char* dhw_train_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DHW_TRAIN_H */
