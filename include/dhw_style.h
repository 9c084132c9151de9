/* dhw_style.h — C-ABI of the StyleExtractor front end in libdhw_hip.so.
 *
 * Replaces, one-to-one (paths relative to diffusion_handwriting_generation/ in the reference):
 *
 *   dhw_style_create / dhw_style_load / dhw_style_finalize
 *        <- StyleExtractor.__init__ (text_style.py:14-31): torchvision `mobilenet_v2(...).features` in eval mode.
 *           Weight interchange is torchvision's own MobileNetV2 state_dict, key by key (`features.N...`;
 *           `classifier.*` and `num_batches_tracked` are accepted and ignored); BatchNorm (eval, eps 1e-5) is folded
 *           into the convolutions at finalize.
 *   dhw_style_forward
 *        <- StyleExtractor.forward (text_style.py:43-59): x/127.5 - 1, repeat to 3 channels, features,
 *           AvgPool2d(3, 3), AdaptiveAvgPool2d((1, 14)), squeeze, permute -> [B, 14, 1280].
 *
 * PARITY UNPINNED: torchvision is not importable in the build container and the pretrained weights cannot be fetched, so
 * no reference output exists for this component; it is checked against the build's own PyTorch restatement of the
 * MobileNetV2 architecture (oracle/mobilenet_ref.py) with random-init weights.
 *
 * Conventions as in dhw.h: plain pointers and sizes, 0 or a negative dhw_status, device pointers for tensors, work is
 * enqueued on the given HIP stream.
 */
#ifndef DHW_STYLE_H
#define DHW_STYLE_H

#include "dhw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dhw_style dhw_style;

/* precision: DHW_PREC_BF16 (bf16 activations / pointwise weights, fp32 accumulate) or DHW_PREC_F32. */
int dhw_style_create(dhw_style** out, int precision, int device);

/* One tensor of torchvision's MobileNetV2 state_dict (HOST pointer, torch-native layout: Conv2d.weight [Cout,Cin/groups,kh,kw],
 * BatchNorm2d weight / bias / running_mean / running_var [C]).  Unknown `features.*` key or wrong shape -> DHW_ERR_KEY. */
int dhw_style_load(dhw_style*, const char* key, const void* host_ptr, int dtype, const int64_t* shape, int ndim);

/* Check that every tensor is present, fold BatchNorm, pad channels, pack for the MFMA GEMM and upload. */
int dhw_style_finalize(dhw_style*);

int dhw_style_num_keys(dhw_style*);
int dhw_style_key_info(dhw_style*, int i, const char** key, int64_t shape[4], int* ndim);

/* img: f32 [B,1,H,W], raw grey levels 0..255 (text_style.py:50-51); out: f32 [B,14,1280].  H, W >= 96 so that the
 * feature map is at least 3 x 3 (AvgPool2d(3, 3)). */
int dhw_style_forward(dhw_style*, const float* img, int B, int H, int W, float* out, void* hip_stream);

/* test hook: the feature map of the last forward, NHWC fp32 [B, H/32, W/32, 1280]; returns floats written */
int64_t dhw_style_debug_features(dhw_style*, float* host_dst, int64_t max_floats, int64_t shape_out[4]);

const char* dhw_style_last_error(dhw_style*);
void dhw_style_destroy(dhw_style*);

#ifdef __cplusplus
}
#endif
#endif /* DHW_STYLE_H */
