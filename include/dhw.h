/* dhw.h — C-ABI of libdhw_hip.so: the MI355X (gfx950) reverse-diffusion
 * handwriting sampler.
 *
 * The reference (sleep3r/Diffusion-Handwriting-Generation.pytorch) is pure
 * Python and has no FFI; the hot path sits behind two Python surfaces, which
 * these entry points replace one-to-one (paths relative to
 * diffusion_handwriting_generation/ in the reference):
 *
 *   dhw_create / dhw_load / dhw_finalize
 *        <- DiffusionModel.__init__ (model.py:64-119) + strict state_dict load
 *           (checkpoint.py:92-130): weight interchange is the reference's own
 *           state_dict, key by key, torch-native layouts.
 *   dhw_forward  <- DiffusionModel.forward (model.py:121-182)
 *   dhw_sample   <- the T-step loop inlined in infer() (inference.py:80-96),
 *                   incl. get_beta_set (utils/nn.py:19-39) and the step
 *                   functions (utils/nn.py:64-112)
 *   dhw_schedule <- get_beta_set + cumprod (utils/nn.py:19-39, inference.py:81)
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every function
 * returns 0 on success or a negative dhw_status; nothing throws across the
 * ABI: every entry point of this library (dhw.h, dhw_debug.h, dhw_style.h,
 * dhw_train.h) runs inside a catch-all barrier that turns a C++ exception into
 * DHW_ERR_INTERNAL + a message (tests/test_host_cpu.py provokes one).  All tensor arguments of dhw_forward / dhw_sample are DEVICE pointers to
 * contiguous row-major buffers owned by the caller; work is enqueued on the
 * given HIP stream and is asynchronous w.r.t. the host.  The library owns its
 * packed weights, FiLM tables and workspace (sized at create from the dims).
 * One handle per device; a handle is not re-entrant (one in-flight call);
 * different handles are independent (batch shards across GPUs use one each).
 */
#ifndef DHW_H
#define DHW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dhw_handle dhw_handle;

typedef enum {
  DHW_OK = 0,
  DHW_ERR_ARG = -1,       /* bad argument / unsupported dims */
  DHW_ERR_KEY = -2,       /* unknown, duplicate or missing state_dict key, or shape mismatch */
  DHW_ERR_HIP = -3,       /* HIP runtime error (no device, launch failure, OOM) */
  DHW_ERR_STATE = -4,     /* call order (e.g. forward before all weights are loaded) */
  DHW_ERR_INTERNAL = -5   /* a C++ exception (std::bad_alloc, std::out_of_range, ...) or an internal inconsistency inside the library:
                             caught at the ABI (csrc/abi_guard.h), reported through dhw_last_error; the process and the handle stay
                             alive (the failed call's outputs are undefined) */
} dhw_status;

/* dtype codes for dhw_load */
enum { DHW_F32 = 0, DHW_BF16 = 1, DHW_F16 = 2, DHW_F64 = 3 };

/* compute precision of the denoiser */
enum { DHW_PREC_BF16 = 0,  /* bf16 activations + weights, fp32 accumulate/LN/softmax/state (perf mode) */
       DHW_PREC_F32 = 1 }; /* fp32 everywhere, exact-f32 MFMA (parity mode) */

typedef struct {
  int num_layers;   /* bottleneck EncoderLayers (model.py:66; shipped configs use 2) */
  int c1, c2, c3;   /* 128 / 192 / 256 (model.py:67-69); c1 must be 128 and c3 256; c2 any multiple of 12 up to 192
                       (3 / 6 / 8 attention heads, model.py:88-106): widths below 192 run zero-padded inside the 192-wide kernels */
  int max_B;        /* largest batch a call may pass */
  int max_L;        /* largest stroke length (multiple of 8, model.py:169-175) */
  int max_Lt;       /* largest token count */
  int S;            /* style rows: style_vector is [B,S,1280] (S = 14 in use, 1 in the reference test) */
  int precision;    /* DHW_PREC_* */
} dhw_dims;

/* Create a handle on HIP device `device`. */
int dhw_create(dhw_handle** out, const dhw_dims* dims, int device);

/* Hand over one state_dict tensor (HOST pointer, torch-native layout:
 * Linear.weight [out,in], Conv1d.weight [Cout,Cin,3]).  The library copies,
 * casts and repacks; the caller's buffer is free after return.  Unknown key or
 * wrong shape -> DHW_ERR_KEY (strict, like checkpoint.py:83-87).  Loading a key
 * again replaces it (and invalidates the packed copy until the next finalize). */
int dhw_load(dhw_handle*, const char* key, const void* host_ptr, int dtype,
             const int64_t* shape, int ndim);

/* Check that every key of the state_dict is present (DHW_ERR_KEY names the
 * first missing one), pack for the MFMA kernels and upload.  Called implicitly
 * by dhw_forward / dhw_sample when needed. */
int dhw_finalize(dhw_handle*);

/* Number of state_dict keys the handle expects, and the i-th key/shape. */
int dhw_num_keys(dhw_handle*);
int dhw_key_info(dhw_handle*, int i, const char** key, int64_t shape[3], int* ndim);

/* == DiffusionModel.forward(strokes, text, sigma, style_vector) -> (eps, pen)
 * strokes f32 [B,L,2]; text int64 [B,Lt] (0 = pad); sigma f32 [B];
 * style f32 [B,S,1280]; eps_out f32 [B,L,2]; pen_out f32 [B,L] in (0,1).
 * L % 8 == 0, B <= max_B, L <= max_L, Lt <= max_Lt. */
int dhw_forward(dhw_handle*, const float* strokes, const int64_t* text, const float* sigma,
                const float* style, int B, int L, int Lt,
                float* eps_out, float* pen_out, void* hip_stream);

/* == inference.py:80-96 for a batch.  mode 0 = "new" (default), 1 = "standard".
 * noise: f32 [T+1,B,L,2] in consumption order (noise[0] = x_T, noise[1+k] = the
 * draw of the k-th loop iteration) or NULL to draw N(0,1) on the device from
 * (seed, first_sample + b, iteration, position) — identical for any sharding.
 * out: f32 [B,L,3] = cat(x_0, pen of the LAST denoiser call). */
int dhw_sample(dhw_handle*, const int64_t* text, const float* style, int B, int L, int Lt,
               int T, int mode, const float* noise, uint64_t seed, int64_t first_sample,
               float* out, void* hip_stream);

/* Host-only: beta_i = 0.02 + exp(linspace(ln 1e-5, ln 0.4, T)), abar = cumprod(1-beta), fp32. */
int dhw_schedule(int T, float* beta_out, float* alpha_bar_out);

/* Algorithmic work of one denoiser call per sample at (L, Lt): FLOPs and the
 * block-boundary activation bytes; used by bench.py. */
int dhw_work(dhw_handle*, int L, int Lt, double* flops_out, double* bytes_out);

const char* dhw_last_error(dhw_handle*);   /* valid until the next call on the handle; NULL handle -> global */
const char* dhw_version(void);
void dhw_destroy(dhw_handle*);

#ifdef __cplusplus
}
#endif
#endif /* DHW_H */
