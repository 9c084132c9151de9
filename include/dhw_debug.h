/* dhw_debug.h — test / measurement hooks of libdhw_hip.so (not part of the
 * drop-in boundary; used by tests/ and bench.py only). */
#ifndef DHW_DEBUG_H
#define DHW_DEBUG_H

#include "dhw.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Copy the named intermediate activation of the LAST dhw_forward call to the
 * host as fp32, C-last [B, rows, cols].  Names follow the reference's module
 * names ("enc1", "enc3", "att_layers.0", "text_style_model", "skip_conv3",
 * "sigma_ffn", "input_dense", "att_dense", ...).  Synchronises the device.
 * Returns the number of floats written, or a negative dhw_status. */
int64_t dhw_debug_read(dhw_handle*, const char* name, float* host_dst, int64_t max_floats,
                       int64_t shape_out[3]);

/* Per-kernel timing: when enabled every launch is bracketed by HIP events on
 * the launch stream (graph replay is disabled while profiling). */
int dhw_profile_enable(dhw_handle*, int on);
int dhw_profile_reset(dhw_handle*);
/* Resolve pending events; returns the number of distinct kernel labels. */
int dhw_profile_count(dhw_handle*);
int dhw_profile_get(dhw_handle*, int i, const char** label, double* total_ms, int64_t* launches,
                    double* flops_per_launch_sum, double* bytes_per_launch_sum);

/* Number of prompt sub-batches dhw_sample runs concurrently (parallel graph branches on side streams);
 * clamped to the count allocated at create (env DHW_STREAMS, default 1).  Returns the value in effect. */
int dhw_set_streams(dhw_handle*, int n);

/* Use (1) or bypass (0) hipGraph replay of the sampling loop. Default 1. */
int dhw_set_graph(dhw_handle*, int on);
/* Number of dhw_sample shapes that run every denoiser call as ONE persistent launch (csrc/persist.h; OFF by default — it measured
   slower than the eleven launches, DESIGN 13.3 — env DHW_PERSIST=1 at dhw_create turns that form on).  0 = every call so far was
   launched kernel by kernel. */
int dhw_debug_persist_plans(dhw_handle*);
/* Diagnostics (DHW_PERSIST_TRACE=1 at dhw_create): 100 MHz time stamps of the last persistent step of the last dhw_sample call,
   [workgroup][16 phases][4] = {ticket drawn, inputs ready, body done, (phase 0: kernel entry; phase 1: exit | xcc << 56)}.
   Returns the number of workgroups, 0 without a trace buffer. */
int dhw_debug_persist_trace(dhw_handle*, unsigned long long* host_dst, int64_t max_words);

/* Teacher forcing of dhw_sample, for long-schedule parity tests (the random-init reverse process grows by 1/sqrt(1 - beta)
 * per step, so a free-running T = 1000 trajectory leaves every meaningful range; SURVEY 7 "hard parts"): with every > 0 the
 * next dhw_sample calls run eagerly and, in front of step k*every (k = 1, 2, ... while k*every < T), copy the state x
 * [B,L,2] reached so far to capture_dev[k-1] and continue from reset_dev[k-1] (device buffers of (T-1)/every x [B,L,2] floats,
 * owned by the caller, alive until the calls have completed).  every = 0 switches it off. */
int dhw_debug_set_teacher(dhw_handle*, const float* reset_dev, float* capture_dev, int every);

/* The device noise generator on its own: the N(0,1) draws of `B` samples x `L` positions x 2 for sampler iteration
 * `iter` (-1 = x_T, k >= 0 = the draw step k adds) under (seed, first_sample), copied to host_dst [B,L,2].
 * Exactly the values dhw_sample(noise = NULL) consumes.  Synchronises the device. */
int dhw_debug_randn(dhw_handle*, uint64_t seed, int64_t first_sample, int B, int L, int iter, float* host_dst);

/* The logical workgroup id the fused kernels derive from blockIdx (csrc/dhw_common.h xcd_swizzle): a bijection of
 * [0, nwg) that gives each of the 8 XCDs a contiguous id range.  Host-side copy for tests; needs no device. */
int dhw_debug_xcd_swizzle(int block_id, int nwg);

/* The self-attention stage of EncoderLayer `layer` (0 = enc3, 1 = enc5, 2 + i = att_layers.i) timed on its own, on the
 * activations the last dhw_forward left in the workspace: enc_bc_kernel launched `iters` times with the stage and `iters`
 * times with the stage skipped, mean launch time of each in microseconds (HIP events on `hip_stream`); flops_out = the
 * stage's QK^T + PV FLOPs, 4 B Lk^2 d.  bf16 handles with the fused kernels only; synchronises the stream. */
int dhw_debug_attention_time(dhw_handle*, int layer, int iters, double* us_with, double* us_without, double* flops_out,
                             void* hip_stream);

/* Raise a C++ exception inside the guarded body of an entry point: kind 1 = std::out_of_range (a std::map::at miss), 2 =
 * std::bad_alloc, 3 = a non-std exception.  Returns DHW_ERR_INTERNAL with the message in dhw_last_error(handle) — the test of
 * "nothing throws across the ABI" (include/dhw.h).  Needs no device; the handle may be NULL (message in the global slot). */
int dhw_debug_raise(dhw_handle*, int kind);

#ifdef __cplusplus
}
#endif
#endif
