// dhw_api.cpp — host side of libdhw_hip.so: the C-ABI of include/dhw.h.
//
// Owns: the state_dict intake (strict, by key name), weight repacking into
// MFMA-fragment order, the sigma-FiLM table, PE·W position-bias tables, the
// activation workspace, the denoiser launch sequence (== DiffusionModel.forward,
// reference model.py:121-182) and the T-step sampler (== inference.py:80-96)
// with hipGraph replay.  No torch types, no exceptions across the ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dhw.h"
#include "abi_guard.h"
#include "xcd_swizzle.h"
#include "../../include/dhw_debug.h"
#include "dhw_kernels.h"
#include "persist.h"

namespace {

constexpr int SIG = 32, SIG_HID = 2048, VOCAB = 73, STYLE_CH = 256;
constexpr int SLACK_ROWS = 64;   // every activation buffer is over-allocated so tile over-reads stay in bounds

ErrBuf g_err;   // errors without a handle (dhw_create, dhw_schedule); a fixed buffer: recording an error never throws

struct KeySpec { std::string key; std::vector<int64_t> shape; };

// ---------------------------------------------------------------- state_dict inventory (mirrors spec.py)
void add_linear(std::vector<KeySpec>& s, const std::string& n, int cin, int cout) {
  s.push_back({n + ".weight", {cout, cin}});
  s.push_back({n + ".bias", {cout}});
}
void add_conv(std::vector<KeySpec>& s, const std::string& n, int cin, int cout) {
  s.push_back({n + ".weight", {cout, cin, 3}});
  s.push_back({n + ".bias", {cout}});
}
void add_affine(std::vector<KeySpec>& s, const std::string& n, int c) {
  add_linear(s, n + ".gamma_emb", SIG, c);
  add_linear(s, n + ".beta_emb", SIG, c);
}
void add_convblock(std::vector<KeySpec>& s, const std::string& n, int cin, int cout) {
  add_affine(s, n + ".affine1", cout / 2);
  add_affine(s, n + ".affine2", cout);
  add_affine(s, n + ".affine3", cout);
  add_conv(s, n + ".conv_skip", cin, cout);
  add_conv(s, n + ".conv1", cin, cout / 2);
  add_conv(s, n + ".conv2", cout / 2, cout);
  add_linear(s, n + ".fc", cout, cout);
}
void add_mha(std::vector<KeySpec>& s, const std::string& n, int d) {
  for (const char* w : {".wq", ".wk", ".wv", ".dense"}) add_linear(s, n + w, d, d);
}
void add_enclayer(std::vector<KeySpec>& s, const std::string& n, int dinp, int d) {
  add_linear(s, n + ".text_dense", dinp, d);
  add_linear(s, n + ".ffn.1", d, 2 * d);
  add_linear(s, n + ".ffn.3", 2 * d, d);
  add_mha(s, n + ".mha", d);
  add_mha(s, n + ".mha2", d);
  for (int k = 0; k < 4; ++k) add_affine(s, n + ".affine" + std::to_string(k), d);
}
std::vector<KeySpec> build_spec(int nl, int c1, int c2, int c3) {
  std::vector<KeySpec> s;
  const int dt = 2 * c2;
  add_linear(s, "input_dense", 2, c1);
  add_linear(s, "sigma_ffn.1", 1, SIG_HID);
  add_linear(s, "sigma_ffn.3", SIG_HID, c1 / 4);
  add_convblock(s, "enc1", c1, c1);
  add_convblock(s, "enc2", c1, c2);
  add_enclayer(s, "enc3", dt, c2);
  add_convblock(s, "enc4", c2, c3);
  add_enclayer(s, "enc5", dt, c3);
  add_conv(s, "skip_conv1", c1, c2);
  add_conv(s, "skip_conv2", c2, c3);
  add_conv(s, "skip_conv3", c3, dt);
  const std::string t = "text_style_model";
  s.push_back({t + ".emb.weight", {VOCAB, dt}});
  add_linear(s, t + ".style_ffn.1", STYLE_CH, 4 * c2);
  add_linear(s, t + ".style_ffn.3", 4 * c2, dt);
  add_linear(s, t + ".text_ffn.1", dt, 2 * dt);
  add_linear(s, t + ".text_ffn.3", 2 * dt, dt);
  add_mha(s, t + ".mha", dt);
  for (int k = 1; k <= 4; ++k) add_affine(s, t + ".affine" + std::to_string(k), dt);
  add_linear(s, "att_dense", 2 * c1, dt);
  for (int i = 0; i < nl; ++i) add_enclayer(s, "att_layers." + std::to_string(i), dt, dt);
  add_convblock(s, "dec3", dt, c3);
  add_convblock(s, "dec2", c3, c2);
  add_convblock(s, "dec1", c2, c1);
  add_linear(s, "output_dense", c1, 2);
  add_linear(s, "pen_lifts_dense.0", c1, 1);
  return s;
}

// ---------------------------------------------------------------- small utilities
uint16_t f2bf(float f) {   // round-to-nearest-even; NaN stays NaN
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
float h2f(uint16_t h) {   // IEEE half -> float
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  float v;
  if (e == 0) v = std::ldexp((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = std::ldexp((float)(m | 1024), (int)e - 25);
  return s ? -v : v;
}

struct ProfRec { int label; hipEvent_t a, b; double flops, bytes; };
struct ProfAgg { std::string label; double ms = 0, flops = 0, bytes = 0; int64_t n = 0; };

struct Tap { void* p; int rows; int cols; bool f32; };
// dhw_debug_read's view of the last call: one slot per named activation, names resolved ONCE at dhw_create (build_names);
// a launch only touches slot ids.  (Round 4 kept a std::map<std::string, Tap> filled with names concatenated at every launch.)
struct TapSlot { std::string name; Tap t{}; bool set = false; };

// ConvBlocks and EncoderLayers by index: nothing on the launch path is looked up by name.
enum { CB_ENC1, CB_ENC2, CB_ENC4, CB_DEC3, CB_DEC2, CB_DEC1, CB_N };
constexpr const char* kConvName[CB_N] = {"enc1", "enc2", "enc4", "dec3", "dec2", "dec1"};
// EncoderLayer li: 0 = enc3, 1 = enc5, 2 + i = att_layers.i
enum { TAP_SIGMA_FFN, TAP_INPUT_DENSE, TAP_TS, TAP_TS_STYLE, TAP_TS_T2, TAP_ATT_DENSE, TAP_UP3, TAP_UP2, TAP_UP1, TAP_CONV0 };
inline int tap_conv(int id) { return TAP_CONV0 + id; }
inline int tap_el(int li, int which) { return TAP_CONV0 + CB_N + 3 * li + which; }   // which: 0 = layer output, 1 = .x2, 2 = .x3

// One full activation workspace.  dhw_sample splits a prompt batch into independent sub-batches, each
// with its own workspace on its own (captured) stream, so several small kernels are in flight at once.
// Every buffer is a NAMED POINTER set by alloc_workspace / ensure_plane (dhw_create, or the first dhw_sample of a longer
// schedule): a buffer the launch sequence needs and the allocation code forgot is reported by need() as DHW_ERR_INTERNAL
// before anything is launched.  (Round 4: a std::map<std::string, void*> looked up with .at(name + ".k1") at every launch;
// a renamed buffer threw std::out_of_range through dhw_forward and aborted the host process.)
struct ConvBufs { void *h1 = nullptr, *h2 = nullptr, *out = nullptr; };
struct TextBufs { void *s1 = nullptr, *k8 = nullptr, *vt8 = nullptr, *t1 = nullptr, *q8 = nullptr, *a8 = nullptr, *t2 = nullptr, *tf_h = nullptr, *text_out = nullptr; };
struct EncTextBufs { void *tl = nullptr, *k1 = nullptr, *vt1 = nullptr; };
struct EncBufs {
  EncTextBufs t, tT;   // the layer's text-side projections: per call, and the all-steps plane
  void *q1 = nullptr, *a1 = nullptr, *x2 = nullptr, *qk2 = nullptr, *vt2 = nullptr, *a2 = nullptr, *x3 = nullptr, *f = nullptr, *out = nullptr;
};
struct Workspace {
  void *sty_in = nullptr, *sty_h = nullptr, *sty_n = nullptr, *t_n = nullptr;
  TextBufs ts, tsT;    // sigma-dependent text side: per call, and the all-steps plane (".T")
  void *x0 = nullptr, *enc1_pool = nullptr, *enc3_pool = nullptr, *enc5_pool = nullptr, *att_dense = nullptr;
  void* xd[3] = {nullptr, nullptr, nullptr};   // decoder inputs xd3, xd2, xd1 (DHW_FUSE_UP=0 only)
  ConvBufs cb[CB_N];
  std::vector<EncBufs> el;
  float* d_xt = nullptr;   // fp32 sampler state [B*L, 2]
  long cap_B = 0;          // prompts this workspace was sized for
  long plane_cap = 0;      // (steps x prompts) the all-steps text plane (".T" buffers) is sized for
};
constexpr int MAX_STREAMS = 8;

// one ConvBlock / EncoderLayer worth of packed weights
struct ConvBlockW {
  void *w_c1, *w_c2, *w_fc, *w_skip;
  float *b_c1, *b_c2, *b_fc, *b_skip;
  int cin, cout, f1, f2, f3;   // FiLM offsets
};
struct EncLayerW {
  void *w_td, *w_kv1, *w_q1, *w_d1, *w_qkv2, *w_d2, *w_f1, *w_f2;
  float *b_td, *b_kv1, *b_q1, *b_d1, *b_qkv2, *b_d2, *b_f1, *b_f2;
  float *pb_k1, *pb_q1, *pb_qk2;   // PE·W tables
  int d, heads, f0, f1, f2, f3;
  float pos_factor;
};

}  // namespace

struct dhw_handle {
  dhw_dims dims{};      // PHYSICAL dims: what the kernels, workspaces and packed weights are sized for (c2 = 192)
  dhw_dims ldims{};     // the caller's dims (the reference's constructor arguments): c2 may be any multiple of 12 up to 192
  bool padded = false;  // ldims.c2 < dims.c2: weights are embedded into the physical shapes with zero padding (pad_weights)
  std::vector<KeySpec> pspec;                  // physical shapes, same key order as spec
  std::vector<std::vector<float>> phys_w;      // padded copies of host_w (padded handles only)
  int device = 0;
  int prec = 0;
  size_t es = 2;
  ErrBuf err;
  bool lookup_fail = false;   // a weight / FiLM name the packing code asked for does not exist (W, film_offset): finalize fails
  std::vector<KeySpec> spec;
  std::map<std::string, int> key_index;
  std::vector<std::vector<float>> host_w;
  std::vector<char> loaded;
  bool packed = false;
  std::vector<void*> allocs;

  // FiLM
  std::map<std::string, int> film_off;
  int film_tot = 0;
  float *d_film_w = nullptr, *d_film_b = nullptr;
  float *d_sig32 = nullptr, *d_film = nullptr, *d_sigma_in = nullptr;
  // dhw_sample: one FiLM table [T, 2*film_tot] per schedule length T, allocated once and never moved, so a cached graph
  // for T keeps reading ITS table whatever other T values are sampled in between (a single shared, re-grown buffer let a
  // replayed graph read another schedule's table).  d_film_T = the table of the call being enqueued.
  struct FilmT {
    float *d_sigma = nullptr, *d_sig32 = nullptr, *d_film = nullptr;
    std::vector<float> h_sigma;   // source of the async upload: must outlive the call
    bool ready = false;           // table computed for the current weights
  };
  std::map<int, FilmT> film_T;
  float* d_film_T = nullptr;

  // small fp32 weights
  float *sg_w1, *sg_b1, *sg_w2, *sg_b2, *in_w, *in_b, *out_w, *out_b, *pen_w, *pen_b, *emb;

  ConvBlockW enc1, enc2, enc4, dec3, dec2, dec1;
  std::vector<EncLayerW> el;   // enc3, enc5, att_layers...
  void *w_sf1, *w_sf3, *w_q8, *w_kv8, *w_d8, *w_tf1, *w_tf3, *w_attd, *w_sk1, *w_sk2, *w_sk3;
  float *b_sf1, *b_sf3, *b_q8, *b_kv8, *b_d8, *b_tf1, *b_tf3, *b_attd, *b_sk1, *b_sk2, *b_sk3;
  int f_ts1, f_ts2, f_ts3, f_ts4;

  // workspaces (ws[0] serves dhw_forward; dhw_sample uses ws[0..nstreams))
  std::vector<Workspace> ws;
  // sub-batches dhw_sample forks onto side streams (<= nstreams_alloc).  Default 1: on ROCm 7.2 parallel
  // hipGraph branches replay serially, and the split only shrinks every launch (measured 65 -> 99 ms at 4).
  int nstreams = 1;
  int nstreams_alloc = 1;
  hipStream_t sub_streams[MAX_STREAMS] = {};
  std::vector<TapSlot> taps;            // indexed by tap id (TAP_*, tap_conv, tap_el)
  std::vector<std::string> el_name;     // "enc3", "enc5", "att_layers.i"
  int lpadT = 0, lpadS = 0, lpadX[3] = {0, 0, 0};

  // profiling
  bool prof = false;
  std::vector<std::string> prof_labels;
  std::vector<ProfRec> prof_recs;
  std::vector<ProfAgg> prof_agg;

  // graph cache for dhw_sample: the graph only touches library-owned staging buffers, so it is keyed by the
  // problem shape alone and replays for any caller pointers
  bool use_graph = true;
  // teacher forcing of dhw_sample (dhw_debug_set_teacher): every `teach_every` steps x is captured and replaced
  int teach_every = 0;
  const float* teach_reset = nullptr;
  float* teach_capture = nullptr;
  bool fuse_heads = true;       // dec1 evaluates heads + scheduler step (env DHW_FUSE_HEADS=0 -> separate launch)
  bool plane = true;            // all-steps text plane in dhw_sample (env DHW_PLANE=0 -> text side inside every step)
  bool fuse_up = true;          // decoder ConvBlocks evaluate Upsample + skip_conv while staging (env DHW_FUSE_UP=0 -> separate GEMM)
  bool chain = true;            // row-local stages continue across layer boundaries inside one launch (env DHW_CHAIN=0 -> off)
  bool fuse = true;             // fused block kernels (env DHW_FUSE=0 -> one launch per GEMM, for A/B runs)
  bool fuse_text = true;        // fused text-side kernels (textside.hip; env DHW_FUSE_TEXT=0 -> generic GEMM / attention launches)
  int text_pairs = 0;           // (step, prompt) pairs per workgroup of text_layer_kernel: 0 = by size, env DHW_TEXT_PAIRS = 1 / 2 forces one form
  std::map<std::vector<uint64_t>, hipGraphExec_t> graphs;
  int64_t* d_text_stage = nullptr;
  float* d_style_stage = nullptr;
  float* d_out_stage = nullptr;
  float* d_noise_stage = nullptr;
  size_t noise_stage_cap = 0;
  uint64_t* d_seed = nullptr;   // [seed, first_sample] read by the noise kernels

  // One persistent launch per denoiser call inside dhw_sample's graph (persist.h): env DHW_PERSIST=1.  OFF by default: measured
  // 359 us per call against 328 us for the eleven launches (profiles/r04_persistent_step_trace.log) — bit-identical
  // samples, but every hand-off costs what a kernel boundary costs and the merged kernel's bodies compile worse.  One StepPlan
  // per sampler step, built by running the launch sequence in record mode, keyed like the graphs.
  bool persist = false;
  int persist_grid = 0;               // resident workgroups to start = the device's CU count
  unsigned* d_step_sync = nullptr;    // tickets / per-sample counters (zero between launches)
  size_t step_sync_words = 0;
  unsigned* h_step_err = nullptr;     // host-mapped error word of the step kernels (bounded spins), and its device address
  unsigned* d_step_err = nullptr;
  struct StepPlans { StepPlan* dev = nullptr; bool ok = false; };
  unsigned long long* d_step_trace = nullptr;   // diagnostics (DHW_PERSIST_TRACE=1): [workgroup][phase][4] stamps of the LAST step of a call
  std::map<std::vector<uint64_t>, StepPlans> plans;

  int last_B = 0, last_L = 0, last_Lt = 0;
};

namespace {

int fail(dhw_handle* h, int code, const char* fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  g_err.vsetf(fmt, ap);
  va_end(ap);
  if (h) h->err.set(g_err.c_str());
  return code;
}

// The body of every extern "C" entry point runs inside this: no exception leaves the library (abi_guard.h).
#define DHW_GUARD(h, fn, R, ...) \
  return abi_guard<R>(fn, [&](const char* f_, const char* w_) { return fail((h), DHW_ERR_INTERNAL, "%s: internal error: %s", f_, w_); }, [&]() -> R __VA_ARGS__)

#define HIPCK(h, call)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(h, DHW_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

int dev_alloc(dhw_handle* h, void** p, size_t bytes, bool zero = true) {
  HIPCK(h, hipMalloc(p, bytes ? bytes : 16));
  h->allocs.push_back(*p);
  if (zero) HIPCK(h, hipMemset(*p, 0, bytes ? bytes : 16));
  return 0;
}

// A weight by state_dict key (finalize-time only).  The packing code names the same keys build_spec declares, so a miss is a
// programming error: it is recorded (dhw_finalize then returns DHW_ERR_INTERNAL) and an empty tensor is returned, which the
// upload helpers reject by size — nothing throws.
const std::vector<float>& W(dhw_handle* h, const std::string& key) {
  static const std::vector<float> none;
  auto it = h->key_index.find(key);
  if (it == h->key_index.end()) {
    if (!h->lookup_fail) fail(h, DHW_ERR_INTERNAL, "internal: the packing code asked for an unknown weight '%s'", key.c_str());
    h->lookup_fail = true;
    return none;
  }
  return (h->padded ? h->phys_w : h->host_w)[it->second];
}
int film_offset(dhw_handle* h, const std::string& name) {
  auto it = h->film_off.find(name);
  if (it == h->film_off.end()) {
    if (!h->lookup_fail) fail(h, DHW_ERR_INTERNAL, "internal: no FiLM layer named '%s'", name.c_str());
    h->lookup_fail = true;
    return 0;
  }
  return it->second;
}

int upload_f32(dhw_handle* h, const std::vector<float>& v, float** out) {
  if (h->lookup_fail) return DHW_ERR_INTERNAL;
  int rc = dev_alloc(h, (void**)out, v.size() * 4, false);
  if (rc) return rc;
  HIPCK(h, hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// Pack a row-major weight matrix Wf[N][K] into MFMA-fragment order
// [N/16][K/32][64 lanes][8]: lane l holds Wf[nt*16 + (l&15)][kc*32 + 8*(l>>4) + j].
int upload_packed(dhw_handle* h, const std::vector<float>& wf, int N, int K, void** out) {
  if (h->lookup_fail) return DHW_ERR_INTERNAL;
  if (N % 16 || K % 32 || (size_t)N * K != wf.size()) return fail(h, DHW_ERR_INTERNAL, "pack: bad shape %d x %d", N, K);
  const size_t n = (size_t)N * K;
  std::vector<float> pk(n);
  size_t o = 0;
  for (int nt = 0; nt < N / 16; ++nt)
    for (int kc = 0; kc < K / 32; ++kc)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) pk[o++] = wf[(size_t)(nt * 16 + (l & 15)) * K + kc * 32 + 8 * (l >> 4) + j];
  int rc = dev_alloc(h, out, n * h->es, false);
  if (rc) return rc;
  if (h->prec == PREC_F32) {
    HIPCK(h, hipMemcpy(*out, pk.data(), n * 4, hipMemcpyHostToDevice));
  } else {
    std::vector<uint16_t> b(n);
    for (size_t i = 0; i < n; ++i) b[i] = f2bf(pk[i]);
    HIPCK(h, hipMemcpy(*out, b.data(), n * 2, hipMemcpyHostToDevice));
  }
  return 0;
}

// Conv1d weight [Cout][Cin][3] -> GEMM matrix [Cout][tap*Cin + c]
std::vector<float> conv_flat(const std::vector<float>& w, int cout, int cin) {
  std::vector<float> f((size_t)cout * cin * 3);
  for (int n = 0; n < cout; ++n)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < 3; ++t) f[(size_t)n * cin * 3 + t * cin + c] = w[((size_t)n * cin + c) * 3 + t];
  return f;
}
std::vector<float> vcat(std::initializer_list<const std::vector<float>*> vs) {
  std::vector<float> r;
  for (auto v : vs) r.insert(r.end(), v->begin(), v->end());
  return r;
}

// Sinusoidal table PE[pos][dim] exactly as attention.py:15-23 evaluates it in fp32.
// (dim_true < dim: the model's width is below the kernels' physical one; the table keeps the row stride dim, the values of the
// dim_true-wide encoding sit in columns [0, dim_true) and the rest is zero — the tail padding of pad_weights)
std::vector<float> pe_table(int n, int dim, float pos_factor, int dim_true = 0) {
  if (dim_true <= 0) dim_true = dim;
  const int half = dim_true / 2;
  const float negc = (float)(-(std::log(10000.0) / (half - 1)));
  std::vector<float> pe((size_t)n * dim);
  for (int j = 0; j < half; ++j) {
    const float f = expf((float)j * negc);
    for (int t = 0; t < n; ++t) {
      const float e = ((float)t * f) * pos_factor;
      pe[(size_t)t * dim + j] = sinf(e);
      pe[(size_t)t * dim + half + j] = cosf(e);
    }
  }
  return pe;
}
// posb[pos][n] = sum_k PE[pos][k] * Wm[n][k]   (Wm: [N][dim] row-major)
std::vector<float> pe_times_w(const std::vector<float>& pe, int n, int dim, const std::vector<float>& wm, int N) {
  std::vector<float> r((size_t)n * N);
  for (int t = 0; t < n; ++t)
    for (int o = 0; o < N; ++o) {
      double a = 0;
      const float* p = &pe[(size_t)t * dim];
      const float* w = &wm[(size_t)o * dim];
      for (int k = 0; k < dim; ++k) a += (double)p[k] * (double)w[k];
      r[(size_t)t * N + o] = (float)a;
    }
  return r;
}

// ---------------------------------------------------------------- model widths below the kernels' (c2 < 192)
// The reference's constructor takes any c2 divisible by 12 (model.py:64-71: 3 heads at c2, 6 at 2*c2, 8 in the TextStyleEncoder,
// text_style.py:78).  The kernels are built for c2 = 192: head dims 64 and 48, LayerNorm widths 192 / 384.  A smaller model is
// EMBEDDED into those shapes: every c2-derived channel axis is zero padded at the tail (c2/2 -> 96, c2 -> 192, 2*c2 -> 384,
// 4*c2 -> 768) and the q / k / v projections' output axis (= the attention dense's input axis) head by head (head h's c2/3 or
// c2/4 channels at the front of its 64- or 48-wide slot).  With zero weights, biases and FiLM rows in the padding every padded
// channel stays exactly 0 through convolutions, SiLU, residuals, pooling and attention; what is left to handle is (1) LayerNorm:
// statistics over the true width, padding written as 0 (GemmParams::ln_n, embed_ln's n_true), (2) the attention's 1/sqrt(depth):
// the kernels scale by the physical head dim, so wq / its bias (and with them PE·Wq) carry sqrt(physical / true), (3) the
// positional encodings, evaluated for the true width (pe_table).  Such a handle runs the one-launch-per-GEMM path (fuse = false):
// the fused block kernels keep their compile-time LayerNorm widths.
int true_width(const dhw_handle* h, int n) {
  if (!h->padded) return n;
  const int c2 = h->ldims.c2;
  switch (n) {
    case 96: return c2 / 2;
    case 192: return c2;
    case 384: return 2 * c2;
    case 768: return 4 * c2;
    default: return n;   // c1 = 128 and c3 = 256 are fixed, so 32 / 64 / 128 / 256 / 512 are never c2-derived
  }
}

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int pad_weights(dhw_handle* h) {
  h->phys_w.assign(h->spec.size(), {});
  for (size_t i = 0; i < h->spec.size(); ++i) {
    const KeySpec& lk = h->spec[i];
    const KeySpec& pk = h->pspec[i];
    const std::string& key = lk.key;
    if (lk.shape.size() != pk.shape.size() || lk.shape.size() > 3) return fail(h, DHW_ERR_ARG, "pad: rank of %s", key.c_str());
    // heads of the attention module this tensor belongs to (0 = not a q/k/v/dense tensor of an attention)
    int heads = 0;
    if (key.find(".mha") != std::string::npos)
      heads = key.compare(0, 4, "enc3") == 0 ? 3 : key.compare(0, 4, "enc5") == 0 ? 4 : key.compare(0, 10, "att_layers") == 0 ? 6 : 8;
    const bool is_dense = key.find(".dense.") != std::string::npos;
    const bool is_q = key.find(".wq.") != std::string::npos;
    int head_axis = -1;   // axis laid out head by head
    if (heads) head_axis = is_dense ? (ends_with(key, ".weight") ? 1 : -1) : 0;
    int64_t ls[3] = {1, 1, 1}, ps[3] = {1, 1, 1};
    for (size_t a = 0; a < lk.shape.size(); ++a) { ls[a] = lk.shape[a]; ps[a] = pk.shape[a]; }
    std::vector<int64_t> map[3];
    float qscale = 1.0f;
    for (int a = 0; a < 3; ++a) {
      map[a].resize(ls[a]);
      if (a == head_axis && ls[a] != ps[a]) {
        const int64_t dl = ls[a] / heads, dp = ps[a] / heads;
        if (dl * heads != ls[a] || dp * heads != ps[a] || dl > dp) return fail(h, DHW_ERR_ARG, "pad: head split of %s", key.c_str());
        for (int64_t j = 0; j < ls[a]; ++j) map[a][j] = (j / dl) * dp + j % dl;
        if (is_q) qscale = std::sqrt((float)dp / (float)dl);
      } else {
        if (ls[a] > ps[a]) return fail(h, DHW_ERR_ARG, "pad: %s is wider than its physical shape", key.c_str());
        for (int64_t j = 0; j < ls[a]; ++j) map[a][j] = j;
      }
    }
    const std::vector<float>& src = h->host_w[i];
    std::vector<float>& dst = h->phys_w[i];
    dst.assign((size_t)(ps[0] * ps[1] * ps[2]), 0.f);
    size_t o = 0;
    for (int64_t x = 0; x < ls[0]; ++x)
      for (int64_t y = 0; y < ls[1]; ++y)
        for (int64_t z = 0; z < ls[2]; ++z) dst[(size_t)((map[0][x] * ps[1] + map[1][y]) * ps[2] + map[2][z])] = src[o++] * qscale;
  }
  return 0;
}

int pack_convblock(dhw_handle* h, const std::string& n, int cin, int cout, ConvBlockW& cb) {
  cb.cin = cin;
  cb.cout = cout;
  int rc;
  if ((rc = upload_packed(h, conv_flat(W(h, n + ".conv1.weight"), cout / 2, cin), cout / 2, 3 * cin, &cb.w_c1))) return rc;
  if ((rc = upload_packed(h, conv_flat(W(h, n + ".conv2.weight"), cout, cout / 2), cout, 3 * (cout / 2), &cb.w_c2))) return rc;
  if ((rc = upload_packed(h, W(h, n + ".fc.weight"), cout, cout, &cb.w_fc))) return rc;
  if ((rc = upload_packed(h, conv_flat(W(h, n + ".conv_skip.weight"), cout, cin), cout, 3 * cin, &cb.w_skip))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".conv1.bias"), &cb.b_c1))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".conv2.bias"), &cb.b_c2))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".fc.bias"), &cb.b_fc))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".conv_skip.bias"), &cb.b_skip))) return rc;
  cb.f1 = film_offset(h, n + ".affine1");
  cb.f2 = film_offset(h, n + ".affine2");
  cb.f3 = film_offset(h, n + ".affine3");
  return 0;
}

int pack_enclayer(dhw_handle* h, const std::string& n, int d, int heads, float pf, int max_lk, EncLayerW& e) {
  e.d = d;
  e.heads = heads;
  e.pos_factor = pf;
  const int dt = 2 * h->dims.c2;
  int rc;
  auto& wq1 = W(h, n + ".mha.wq.weight");
  auto& wk1 = W(h, n + ".mha.wk.weight");
  auto& wv1 = W(h, n + ".mha.wv.weight");
  auto& wq2 = W(h, n + ".mha2.wq.weight");
  auto& wk2 = W(h, n + ".mha2.wk.weight");
  auto& wv2 = W(h, n + ".mha2.wv.weight");
  if ((rc = upload_packed(h, W(h, n + ".text_dense.weight"), d, dt, &e.w_td))) return rc;
  if ((rc = upload_packed(h, vcat({&wk1, &wv1}), 2 * d, d, &e.w_kv1))) return rc;
  if ((rc = upload_packed(h, wq1, d, d, &e.w_q1))) return rc;
  if ((rc = upload_packed(h, W(h, n + ".mha.dense.weight"), d, d, &e.w_d1))) return rc;
  if ((rc = upload_packed(h, vcat({&wq2, &wk2, &wv2}), 3 * d, d, &e.w_qkv2))) return rc;
  if ((rc = upload_packed(h, W(h, n + ".mha2.dense.weight"), d, d, &e.w_d2))) return rc;
  if ((rc = upload_packed(h, W(h, n + ".ffn.1.weight"), 2 * d, d, &e.w_f1))) return rc;
  if ((rc = upload_packed(h, W(h, n + ".ffn.3.weight"), d, 2 * d, &e.w_f2))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".text_dense.bias"), &e.b_td))) return rc;
  if ((rc = upload_f32(h, vcat({&W(h, n + ".mha.wk.bias"), &W(h, n + ".mha.wv.bias")}), &e.b_kv1))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".mha.wq.bias"), &e.b_q1))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".mha.dense.bias"), &e.b_d1))) return rc;
  if ((rc = upload_f32(h, vcat({&W(h, n + ".mha2.wq.bias"), &W(h, n + ".mha2.wk.bias"), &W(h, n + ".mha2.wv.bias")}), &e.b_qkv2))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".mha2.dense.bias"), &e.b_d2))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".ffn.1.bias"), &e.b_f1))) return rc;
  if ((rc = upload_f32(h, W(h, n + ".ffn.3.bias"), &e.b_f2))) return rc;
  // (x + PE)·W = x·W + PE·W: the PE term is a per-position bias table (model.py:40-50, attention.py:15-23)
  const auto pe_t = pe_table(h->dims.max_Lt + SLACK_ROWS, d, 1.0f, true_width(h, d));   // text_pe_gen: pos_factor 1 (model.py:22)
  const auto pe_x = pe_table(max_lk + SLACK_ROWS, d, pf, true_width(h, d));             // stroke_pe_gen
  if ((rc = upload_f32(h, pe_times_w(pe_t, h->dims.max_Lt + SLACK_ROWS, d, wk1, d), &e.pb_k1))) return rc;
  if ((rc = upload_f32(h, pe_times_w(pe_x, max_lk + SLACK_ROWS, d, wq1, d), &e.pb_q1))) return rc;
  if ((rc = upload_f32(h, pe_times_w(pe_x, max_lk + SLACK_ROWS, d, vcat({&wq2, &wk2}), 2 * d), &e.pb_qk2))) return rc;
  e.f0 = film_offset(h, n + ".affine0");
  e.f1 = film_offset(h, n + ".affine1");
  e.f2 = film_offset(h, n + ".affine2");
  e.f3 = film_offset(h, n + ".affine3");
  return 0;
}

int act_alloc(dhw_handle* h, void** slot, long rows, int cols, bool f32 = false) {
  const size_t bytes = (size_t)(rows + SLACK_ROWS) * cols * (f32 ? 4 : h->es);
  return dev_alloc(h, slot, bytes, true);
}

int pad32(int x) { return ((x + 31) / 32) * 32; }

// width of EncoderLayer li (0 = enc3 at c2, 1 = enc5 at c3, the bottleneck layers at 2 c2) and its stroke rows at length L
int el_width(const dhw_dims& d, int li) { return li == 0 ? d.c2 : li == 1 ? d.c3 : 2 * d.c2; }
long el_rows(long L, int li) { return li == 0 ? L / 2 : li == 1 ? L / 4 : L / 8; }

int alloc_workspace(dhw_handle* h, Workspace& w, long B) {
  w.cap_B = B;
  const dhw_dims& d = h->dims;
  const long L = d.max_L, Lt = d.max_Lt, S5 = d.S * 5;
  const int c1 = d.c1, c2 = d.c2, c3 = d.c3, dt = 2 * c2;
  int rc;
#define AA(slot, rows, cols) if ((rc = act_alloc(h, &(slot), rows, cols))) return rc
  AA(w.sty_in, B * S5, STYLE_CH); AA(w.sty_h, B * S5, 4 * c2); AA(w.sty_n, B * S5, dt); AA(w.ts.s1, B * S5, dt);
  AA(w.ts.k8, B * S5, dt);
  AA(w.t_n, B * Lt, dt); AA(w.ts.t1, B * Lt, dt); AA(w.ts.q8, B * Lt, dt); AA(w.ts.a8, B * Lt, dt); AA(w.ts.t2, B * Lt, dt);
  AA(w.ts.tf_h, B * Lt, 2 * dt); AA(w.ts.text_out, B * Lt, dt);
  h->lpadS = pad32((int)S5);
  h->lpadT = pad32((int)Lt);
  AA(w.ts.vt8, B * dt, h->lpadS);
  AA(w.x0, B * L, c1);
  struct CB { long rows; int cout; };
  const CB cbs[CB_N] = {{L, c1}, {L / 2, c2}, {L / 4, c3}, {L / 4, c3}, {L / 2, c2}, {L, c1}};
  for (int i = 0; i < CB_N; ++i) {
    const CB& c = cbs[i];
    AA(w.cb[i].h1, B * c.rows, c.cout / 2);
    AA(w.cb[i].h2, B * c.rows, c.cout);
    if (i == CB_DEC1) { if ((rc = act_alloc(h, &w.cb[i].out, B * c.rows, c.cout, true))) return rc; }   // dec1's output feeds the heads in fp32
    else AA(w.cb[i].out, B * c.rows, c.cout);
  }
  AA(w.enc1_pool, B * L / 2, c1);
  h->lpadX[0] = pad32((int)(L / 2));
  h->lpadX[1] = pad32((int)(L / 4));
  h->lpadX[2] = pad32((int)(L / 8));
  w.el.assign(2 + d.num_layers, EncBufs{});
  for (size_t i = 0; i < w.el.size(); ++i) {
    EncBufs& e = w.el[i];
    const int dm = el_width(d, (int)i), lp = h->lpadX[i < 2 ? i : 2];
    const long rows = el_rows(L, (int)i);
    AA(e.t.tl, B * Lt, dm); AA(e.t.k1, B * Lt, dm); AA(e.t.vt1, B * dm, h->lpadT);
    AA(e.q1, B * rows, dm); AA(e.a1, B * rows, dm); AA(e.x2, B * rows, dm);
    // (qk2: the bf16 fused kernels keep [q2 | k2 | v2] rows; the other paths use 2 dm columns of it and the transposed vt2)
    AA(e.qk2, B * rows, 3 * dm); AA(e.vt2, B * dm, lp); AA(e.a2, B * rows, dm);
    AA(e.x3, B * rows, dm); AA(e.f, B * rows, 2 * dm); AA(e.out, B * rows, dm);
  }
  AA(w.enc3_pool, B * L / 4, c2); AA(w.enc5_pool, B * L / 8, c3);
  AA(w.att_dense, B * L / 8, dt);
  AA(w.xd[0], B * L / 4, dt); AA(w.xd[1], B * L / 2, c3); AA(w.xd[2], B * L, c2);
#undef AA
  if ((rc = dev_alloc(h, (void**)&w.d_xt, (size_t)(B * L + SLACK_ROWS) * 2 * 4))) return rc;
  return 0;
}

// All-steps text plane: every sigma-dependent text-side activation for `steps` sampler steps x `B` prompts.
int ensure_plane(dhw_handle* h, Workspace& w, long steps, long B) {
  if (steps * B <= w.plane_cap) return 0;
  const dhw_dims& d = h->dims;
  const long n = steps * B, Lt = d.max_Lt, S5 = d.S * 5;
  const int dt = 2 * d.c2;
  int rc;
#define AA(slot, rows, cols) if ((rc = act_alloc(h, &(slot), rows, cols))) return rc
  // with the fused text-side kernels (every call of this handle qualifies) only text_out and the layers' K / V exist
  const bool fused = h->fuse && h->fuse_text && textside_supported(h->prec, (int)Lt, (int)S5, dt);
  if (!fused) {
    AA(w.tsT.s1, n * S5, dt); AA(w.tsT.k8, n * S5, dt); AA(w.tsT.vt8, n * dt, h->lpadS);
    AA(w.tsT.t1, n * Lt, dt); AA(w.tsT.q8, n * Lt, dt); AA(w.tsT.a8, n * Lt, dt); AA(w.tsT.t2, n * Lt, dt);
    AA(w.tsT.tf_h, n * Lt, 2 * dt);
  }
  AA(w.tsT.text_out, n * Lt, dt);
  for (size_t i = 0; i < w.el.size(); ++i) {
    const int dm = el_width(d, (int)i);
    if (!fused) AA(w.el[i].tT.tl, n * Lt, dm);
    AA(w.el[i].tT.k1, n * Lt, dm); AA(w.el[i].tT.vt1, n * dm, h->lpadT);
  }
#undef AA
  w.plane_cap = n;   // (a grown plane leaks the smaller one until destroy)
  return 0;
}

int alloc_shared(dhw_handle* h) {
  const dhw_dims& d = h->dims;
  const long B = d.max_B, L = d.max_L, Lt = d.max_Lt, S5 = d.S * 5;
  int rc;
  if ((rc = dev_alloc(h, (void**)&h->d_sigma_in, B * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_sig32, B * SIG * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_film, (size_t)B * 2 * h->film_tot * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_seed, 16))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_text_stage, (size_t)(B * Lt + 64) * 8))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_style_stage, (size_t)(B * S5 + SLACK_ROWS) * STYLE_CH * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&h->d_out_stage, (size_t)(B * L + SLACK_ROWS) * 3 * 4))) return rc;
  return 0;
}

void build_film_layout(dhw_handle* h) {
  int off = 0;
  for (const KeySpec& k : h->pspec) {
    const std::string suf = ".gamma_emb.weight";
    if (k.key.size() > suf.size() && k.key.compare(k.key.size() - suf.size(), suf.size(), suf) == 0) {
      h->film_off[k.key.substr(0, k.key.size() - suf.size())] = off;
      off += (int)k.shape[0];
    }
  }
  h->film_tot = off;
}

// ---------------------------------------------------------------- profiling wrapper
struct Launch {
  dhw_handle* h;
  hipStream_t st;
  int rec = -1;
  Launch(dhw_handle* h_, hipStream_t st_, const char* label, double flops = 0, double bytes = 0) : h(h_), st(st_) {
    if (!h->prof) return;
    int id = -1;
    for (size_t i = 0; i < h->prof_labels.size(); ++i)
      if (h->prof_labels[i] == label) id = (int)i;
    if (id < 0) { id = (int)h->prof_labels.size(); h->prof_labels.push_back(label); }
    ProfRec r{id, nullptr, nullptr, flops, bytes};
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, st);
    h->prof_recs.push_back(r);
    rec = (int)h->prof_recs.size() - 1;
  }
  ~Launch() {
    if (rec >= 0) hipEventRecord(h->prof_recs[rec].b, st);
  }
};

// ---------------------------------------------------------------- the denoiser launch sequence
struct Ctx {
  dhw_handle* h;
  Workspace* ws;
  hipStream_t st;
  int B, L, Lt, S5;
  const float* film;   // row 0 of the FiLM table to use
  long film_bs;        // FiLM row stride (0 in the sampling loop)
  int err = 0;
  int film_div = 1;    // samples per FiLM row
  int in_B = 0;        // batch of the sigma-independent inputs (0 = B); the text plane replicates them over steps
  bool planeT = false; // the text side writes the all-steps plane (".T" buffers) instead of the per-call ones
  const HeadsParams* fhp = nullptr;   // sampling loop: dec1 evaluates the heads + scheduler step itself
  bool fuse_input = false;  // enc1 evaluates input_dense while staging (sampling loop); forward() keeps the tap
  bool use_plane = false;   // stroke path reads the text K/V of step `plane_step` from the plane
  long plane_step = 0;
  // record mode (persist.h): the fused launches of stroke_path are appended to `rec` as phases instead of being launched;
  // anything the persistent kernel has no phase for sets rec_fail
  std::vector<StepPhase>* rec = nullptr;
  bool rec_fail = false;
};

// append one phase of the persistent per-step kernel (record mode)
void rec_phase(Ctx& c, int kind, int L, const ConvBlockParams* cb, const EncLayerParams* el, const EncChain* nx) {
  int rows = 0, lv = 0;
  if (!step_kind_geometry(kind, L, &rows, &lv) || (int)c.rec->size() >= STEP_MAX_PHASES) { c.rec_fail = true; return; }
  StepPhase ph{};
  ph.kind = kind;
  ph.rows = rows;
  ph.tps = (lv + rows - 1) / rows;
  if (cb) ph.cb = *cb;
  if (el) ph.el = *el;
  if (nx) ph.nx = *nx;
  c.rec->push_back(ph);
}

// A workspace pointer the launch sequence is about to hand to a kernel.  All of them are set when the workspace is allocated
// (alloc_workspace at dhw_create, ensure_plane); one that is still null here is a bug in that code: it becomes a status
// (every launch helper checks c.err first) — not a throw across the ABI and not a null dereference on the device.
void* need(Ctx& c, void* p, const char* what) {
  if (!p && !c.err) c.err = fail(c.h, DHW_ERR_INTERNAL, "internal: workspace buffer '%s' was never allocated", what);
  return p;
}
#define WS(c, field) need((c), (c).ws->field, #field)
#define TS(c, field) need((c), ((c).planeT ? (c).ws->tsT : (c).ws->ts).field, (c).planeT ? #field ".T" : #field)          /* sigma-dependent text side */
#define CBB(c, id, field) need((c), (c).ws->cb[id].field, #field)                                                        /* ConvBlock id */
#define ELB(c, li, field) need((c), (c).ws->el[li].field, #field)                                                        /* EncoderLayer li */
#define ELT(c, li, field) need((c), ((c).planeT ? (c).ws->el[li].tT : (c).ws->el[li].t).field, (c).planeT ? #field ".T" : #field)   /* its text projections, as the text side writes them */
#define ELK(c, li, field) need((c), ((c).use_plane ? (c).ws->el[li].tT : (c).ws->el[li].t).field, (c).use_plane ? #field ".T" : #field)   /* ... as the stroke side reads them */

GemmParams gp_base(const Ctx& c, int L, int N) {
  GemmParams p{};
  p.nseg = 1;
  p.B = c.B;
  p.L = L;
  p.N = N;
  p.n_store = N;
  p.film_bs = c.film_bs;
  p.film_div = c.film_div;
  return p;
}
// All-steps text plane: a GEMM with no per-sample structure (no position bias, no transposed-V output) can see each
// sampler step as ONE long "sample" of film_div*L rows sharing one FiLM row, so 64-row tiles run across prompts.
// Measured slower than per-prompt 32-row tiles (29.60 vs 29.28 ms/step: the 64x384 LayerNorm tile runs at one wave per
// SIMD), so it is opt-in (DHW_FLAT_TEXT=1).
GemmParams gp_text(const Ctx& c, int L, int N) {
  GemmParams p = gp_base(c, L, N);
  static const bool flat = getenv("DHW_FLAT_TEXT") && atoi(getenv("DHW_FLAT_TEXT")) != 0;
  if (c.film_div > 1 && flat) {
    p.B = c.B / c.film_div;
    p.L = L * c.film_div;
    p.film_div = 1;
  }
  return p;
}
void set_film(const Ctx& c, GemmParams& p, int off, int mode) {
  p.gam = c.film + off;
  p.bet = c.film + c.h->film_tot + off;
  p.film_mode = mode;
}
double gemm_flops(const GemmParams& p) {
  double k = 0;
  for (int s = 0; s < p.nseg; ++s) k += (double)p.seg[s].C * p.seg[s].taps;
  return 2.0 * p.B * p.L * p.N * k;
}
double gemm_bytes(const dhw_handle* h, const GemmParams& p) {   // algorithmic: activations in + out once, weights once
  double b = 0;
  for (int s = 0; s < p.nseg; ++s) b += (double)p.B * p.L * p.seg[s].C * h->es + (double)p.N * p.seg[s].C * p.seg[s].taps * h->es;
  b += (double)p.B * p.L * p.N * (p.out_f32 ? 4 : h->es);
  if (p.res1) b += (double)p.B * p.L * p.N * h->es;
  if (p.res2) b += (double)p.B * p.L * p.N * h->es / (p.res2_half ? 2 : 1);
  if (p.pool) b += (double)p.B * p.L * p.N * h->es / 2;
  return b;
}
void run_gemm(Ctx& c, const char* label, const GemmParams& p) {
  if (c.rec) { c.rec_fail = true; return; }
  if (c.err) return;
  Launch l(c.h, c.st, label, gemm_flops(p), gemm_bytes(c.h, p));
  GemmParams q = p;
  if (q.ln) q.ln_n = true_width(c.h, q.N);
  hipError_t e = launch_gemm(c.h->prec, q, c.st);
  if (e != hipSuccess) c.err = fail(c.h, DHW_ERR_HIP, "gemm %s: %s", label, hipGetErrorString(e));
}
void run_attn(Ctx& c, const char* label, const AttnParams& p) {
  if (c.rec) { c.rec_fail = true; return; }
  if (c.err) return;
  Launch l(c.h, c.st, label, 4.0 * p.B * p.H * (double)p.Lq * p.Lk * p.D,
           (double)p.B * p.H * p.D * (2.0 * p.Lq + 2.0 * p.Lk) * c.h->es);
  hipError_t e = launch_attn(c.h->prec, p, c.st);
  if (e != hipSuccess) c.err = fail(c.h, DHW_ERR_HIP, "attn %s: %s", label, hipGetErrorString(e));
}
#define RUN_SMALL(c, label, call)                                                                  \
  do {                                                                                             \
    if ((c).rec) (c).rec_fail = true;                                                              \
    else if (!(c).err) {                                                                                \
      Launch l_((c).h, (c).st, label);                                                             \
      hipError_t e_ = (call);                                                                      \
      if (e_ != hipSuccess) (c).err = fail((c).h, DHW_ERR_HIP, "%s: %s", label, hipGetErrorString(e_)); \
    }                                                                                              \
  } while (0)

void tap(Ctx& c, int id, void* p, int rows, int cols, bool f32 = false) {
  TapSlot& s = c.h->taps[id];
  s.t = Tap{p, rows, cols, f32};
  s.set = true;
}
void taps_clear(dhw_handle* h) {
  for (TapSlot& s : h->taps) s.set = false;
}

// decoder input produced inside the block: Upsample(low) + skip_conv(hskip)  (model.py:169-175)
struct UpIn { const void* hskip; const void* w; const float* b; int cin; const void* low; };

// cnn.py:64-87 as one fused launch (or three fused GEMM launches)
// chain: the EncoderLayer half the block's workgroups continue with (EncChain mode 1), or null; *chained reports
// whether the launch took it
// chain_auto: take the chain only where convblock_chain_auto says it pays for this launch geometry
void conv_block(Ctx& c, int id, const ConvBlockW& w, const void* x, int L, void* out, bool out_f32,
                void* pool, const float* strokes = nullptr, const UpIn* up = nullptr, const EncChain* chain = nullptr,
                bool* chained = nullptr, bool chain_auto = false) {
  dhw_handle* h = c.h;
  const char* n = kConvName[id];
  if (h->fuse) {
    ConvBlockParams q{};
    q.strokes = strokes; q.in_w = h->in_w; q.in_b = h->in_b;
    if (up) { q.up_h = up->hskip; q.up_cin = up->cin; q.up_w = up->w; q.up_b = up->b; q.up_low = up->low; }
    q.x = x; q.B = c.B; q.L = L; q.Cin = w.cin; q.Cout = w.cout;
    q.w_c1 = w.w_c1; q.w_c2 = w.w_c2; q.w_fc = w.w_fc; q.w_skip = w.w_skip;
    q.b_c1 = w.b_c1; q.b_c2 = w.b_c2; q.b_fc = w.b_fc; q.b_skip = w.b_skip;
    q.film = c.film; q.film_bs = c.film_bs; q.film_tot = h->film_tot;
    q.f1 = w.f1; q.f2 = w.f2; q.f3 = w.f3;
    q.out = out; q.out_f32 = out_f32; q.pool = pool;
    if (c.fhp && id == CB_DEC1) {
      q.fuse_heads = 1;
      q.hp = *c.fhp;
      q.hp.w_out = h->out_w; q.hp.b_out = h->out_b; q.hp.w_pen = h->pen_w; q.hp.b_pen = h->pen_b;
      q.hp.L = L;
      q.out = nullptr;   // the fp32 activation never leaves LDS
    }
    if (c.rec) {
      // the phase kinds the persistent kernel is built with (persist.h): the reference's widths, the canonical row tiles
      const bool ch = chain && chain->mode == 1 && convblock_chain_supported(h->prec, q, *chain) && (!chain_auto || convblock_chain_auto(q));
      if (chained) *chained = ch;
      int kind = -1;
      if (id == CB_ENC1 && strokes && !up) kind = PK_CONV_ENC1;
      else if (id == CB_ENC2 && ch && !up && !strokes) kind = PK_CONV_ENC2A;
      else if (id == CB_ENC4 && !(chain && chain->mode) && !up && !strokes) kind = PK_CONV_ENC4;
      else if (id == CB_DEC3 && up) kind = PK_CONV_DEC3;
      else if (id == CB_DEC2 && up) kind = PK_CONV_DEC2;
      else if (id == CB_DEC1 && up && q.fuse_heads) kind = PK_CONV_DEC1;
      if (kind < 0 || h->prec != PREC_BF16 || (L & 1)) { c.rec_fail = true; return; }
      rec_phase(c, kind, c.L, &q, nullptr, ch ? chain : nullptr);
      return;
    }
    if (!c.err) {
      const double rows = (double)c.B * L;
      const double upf = up ? 3.0 * up->cin * w.cin : 0.0;   // skip_conv MACs per row
      const bool ch = chain && chain->mode && convblock_chain_supported(h->prec, q, *chain) && (!chain_auto || convblock_chain_auto(q));
      if (chained) *chained = ch;
      const double dd = w.cout;
      const double chf = ch ? 2.0 * rows * dd * dd * 5 + 4.0 * rows * c.Lt * dd : 0.0, chb = ch ? rows * dd * 4 * h->es + 5.0 * dd * dd * h->es : 0.0;
      Launch l(h, c.st, ch ? "convblock.fused+a" : "convblock.fused", 2.0 * rows * (4.5 * w.cin * w.cout + 2.5 * w.cout * w.cout + upf) + chf,
               rows * ((up ? up->cin + 0.5 * w.cin : w.cin) * h->es + w.cout * (out_f32 ? 4.0 : (double)h->es) * (pool ? 1.5 : 1.0)) +
                   (4.5 * w.cin * w.cout + 2.5 * w.cout * w.cout + upf) * h->es + chb);
      hipError_t e = ch ? launch_convblock_chain(h->prec, q, *chain, c.st) : launch_convblock(h->prec, q, c.st);
      if (e != hipSuccess) c.err = fail(h, DHW_ERR_HIP, "convblock %s: %s", n, hipGetErrorString(e));
    }
    tap(c, tap_conv(id), out, L, w.cout, out_f32);
    return;
  }
  if (c.rec) { c.rec_fail = true; return; }
  {  // h1 = SiLU(FiLM1(conv1(SiLU(x))))
    GemmParams p = gp_base(c, L, w.cout / 2);
    p.seg[0] = GemmSeg{x, w.w_c1, w.cin, 3, 1};
    p.bias0 = w.b_c1;
    set_film(c, p, w.f1, 1);
    p.silu_out = 1;
    p.out = CBB(c, id, h1);
    run_gemm(c, "convblock.conv1", p);
  }
  {  // h2 = SiLU(FiLM2(conv2(h1)))
    GemmParams p = gp_base(c, L, w.cout);
    p.seg[0] = GemmSeg{CBB(c, id, h1), w.w_c2, w.cout / 2, 3, 0};
    p.bias0 = w.b_c2;
    set_film(c, p, w.f2, 1);
    p.silu_out = 1;
    p.out = CBB(c, id, h2);
    run_gemm(c, "convblock.conv2", p);
  }
  {  // out = FiLM3(fc(h2)) + conv_skip(x)
    GemmParams p = gp_base(c, L, w.cout);
    p.nseg = 2;
    p.seg[0] = GemmSeg{CBB(c, id, h2), w.w_fc, w.cout, 1, 0};
    p.seg[1] = GemmSeg{x, w.w_skip, w.cin, 3, 0};
    p.bias0 = w.b_fc;
    p.bias1 = w.b_skip;
    set_film(c, p, w.f3, 2);
    p.out = out;
    p.out_f32 = out_f32;
    p.pool = pool;
    run_gemm(c, "convblock.fc_skip", p);
  }
  tap(c, tap_conv(id), out, L, w.cout, out_f32);
}

// the layer's text values are kept as rows [B*Lt, d] (fused bf16 EncoderLayer kernels) instead of transposed [B][d][lpadT]
bool v_rows(const dhw_handle* h, const EncLayerW& w) { return h->fuse && h->prec == PREC_BF16 && enclayer_supported(h->prec, w.d, w.heads); }

// model.py:37-58.  The text-side projections (tl, k1, vt1) are produced by enc_layer_text.
void enc_layer_text(Ctx& c, int li, const EncLayerW& w) {
  dhw_handle* h = c.h;
  const int dt = 2 * h->dims.c2;
  {  // tl = FiLM0(LN(text_dense(SiLU(text))))
    GemmParams p = gp_text(c, c.Lt, w.d);
    p.seg[0] = GemmSeg{TS(c, text_out), w.w_td, dt, 1, 1};
    p.bias0 = w.b_td;
    p.ln = 1;
    set_film(c, p, w.f0, 1);
    p.out = ELT(c, li, tl);
    run_gemm(c, "enc.text_dense", p);
  }
  if (v_rows(h, w)) {
    // the fused bf16 EncoderLayer kernels read the values as rows [B*Lt, d] (attn_core.h): K and V as two launches of the
    // stacked [2d x d] weight's halves (this generic text path only runs with DHW_FUSE_TEXT=0 / unsupported text shapes)
    for (int half = 0; half < 2; ++half) {
      GemmParams p = gp_base(c, c.Lt, w.d);
      p.seg[0] = GemmSeg{ELT(c, li, tl), (const char*)w.w_kv1 + (size_t)half * w.d * w.d * h->es, w.d, 1, 0};
      p.bias0 = w.b_kv1 + half * w.d;
      if (half == 0) { p.posb = w.pb_k1; p.posb_cols = w.d; }
      p.out = half ? ELT(c, li, vt1) : ELT(c, li, k1);
      run_gemm(c, half ? "enc.v_text" : "enc.k_text", p);
    }
  } else {  // k1 = Wk(tl + PE), v1 = Wv(tl)   (values carry no PE: model.py:46)
    GemmParams p = gp_base(c, c.Lt, 2 * w.d);
    p.seg[0] = GemmSeg{ELT(c, li, tl), w.w_kv1, w.d, 1, 0};
    p.bias0 = w.b_kv1;
    p.posb = w.pb_k1;
    p.posb_cols = w.d;
    p.n_store = w.d;
    p.out = ELT(c, li, k1);
    p.vt = ELT(c, li, vt1);
    p.vt_lpad = h->lpadT;
    run_gemm(c, "enc.kv_text", p);
  }
}

// parameters of the fused EncoderLayer kernels for layer n (x may be null when the tile is handed over in LDS)
EncLayerParams enc_params(Ctx& c, int li, const EncLayerW& w, const void* x, int Lk, int lpad, const int64_t* text,
                          void* pool) {
  dhw_handle* h = c.h;
  const int d = w.d;
  // text keys/values of this layer: per-call buffers, or step `plane_step` of the all-steps plane
  const char* k1p = (const char*)ELK(c, li, k1);
  const char* vt1p = (const char*)ELK(c, li, vt1);
  if (c.use_plane) {
    k1p += (size_t)c.plane_step * c.B * c.Lt * d * h->es;
    vt1p += (size_t)c.plane_step * c.B * (v_rows(h, w) ? c.Lt : h->lpadT) * d * h->es;
  }
  EncLayerParams q{};
  q.B = c.B; q.Lk = Lk; q.Lt = c.Lt; q.d = d; q.heads = w.heads;
  q.x = x;
  q.w_q1 = w.w_q1; q.w_d1 = w.w_d1; q.w_qkv2 = w.w_qkv2; q.w_d2 = w.w_d2; q.w_f1 = w.w_f1; q.w_f2 = w.w_f2;
  q.b_q1 = w.b_q1; q.b_d1 = w.b_d1; q.b_qkv2 = w.b_qkv2; q.b_d2 = w.b_d2; q.b_f1 = w.b_f1; q.b_f2 = w.b_f2;
  q.pb_q1 = w.pb_q1; q.pb_qk2 = w.pb_qk2;
  q.film = c.film; q.film_bs = c.film_bs; q.film_tot = h->film_tot; q.f1 = w.f1; q.f2 = w.f2; q.f3 = w.f3;
  q.k1 = k1p; q.vt1 = vt1p; q.lpadT = h->lpadT; q.text = text;
  q.x2 = ELB(c, li, x2); q.qk2 = ELB(c, li, qk2); q.vt2 = ELB(c, li, vt2); q.lpadX = lpad;
  q.out = ELB(c, li, out); q.pool = pool;
  return q;
}

// skip_a: this layer's enc_a half was already evaluated by the previous launch (EncChain); chain: what this layer's
// enc_bc launch continues with (or null)
void enc_layer(Ctx& c, int li, const EncLayerW& w, const void* x, int Lk, int lpad, const int64_t* text,
               void* pool, bool skip_a = false, const EncChain* chain = nullptr, int bm_min = 0) {
  dhw_handle* h = c.h;
  const int d = w.d;
  const char* k1p = (const char*)ELK(c, li, k1);
  const char* vt1p = (const char*)ELK(c, li, vt1);
  if (c.use_plane) {
    k1p += (size_t)c.plane_step * c.B * c.Lt * d * h->es;
    vt1p += (size_t)c.plane_step * c.B * (v_rows(h, w) ? c.Lt : h->lpadT) * d * h->es;
  }
  if (h->fuse && enclayer_supported(h->prec, d, w.heads)) {
    EncLayerParams q = enc_params(c, li, w, x, Lk, lpad, text, pool);
    q.bm_min = bm_min;
    const double rows = (double)c.B * Lk, dd = d;
    if (c.rec) {
      const bool chained = chain && chain->mode;
      if (h->prec != PREC_BF16) { c.rec_fail = true; return; }
      if (!skip_a) {
        if (d == 256) rec_phase(c, PK_A256, c.L, nullptr, &q, nullptr);
        else c.rec_fail = true;
      }
      int kind = -1;
      if (d == 192 && !chained && skip_a) kind = PK_BC192;
      else if (d == 256 && chained && chain->mode == 2 && !skip_a && bm_min == 32 && !(Lk & 1)) kind = PK_BC256_N2;
      else if (d == 384 && chained && chain->mode == 1 && skip_a) kind = PK_BC384_N1;
      else if (d == 384 && !chained && skip_a) kind = PK_BC384;
      if (kind < 0) { c.rec_fail = true; return; }
      rec_phase(c, kind, c.L, nullptr, &q, chained ? chain : nullptr);
      return;
    }
    for (int which = skip_a ? 1 : 0; which < 2 && !c.err; ++which) {
      double fl = which == 0 ? 2.0 * rows * dd * dd * 5 + 4.0 * rows * c.Lt * dd
                             : 2.0 * rows * dd * dd * 5 + 4.0 * rows * Lk * dd;
      double by = (which == 0 ? rows * dd * 5 : rows * dd * (5 + (pool ? 0.5 : 0.0))) * h->es + 5.0 * dd * dd * h->es;
      const EncChain* ch = which == 1 ? chain : nullptr;
      if (ch && ch->mode) {   // + the chained layer's enc_a (+ att_dense)
        const double r2 = (double)c.B * ch->a.Lk, d2 = ch->a.d;
        fl += 2.0 * r2 * d2 * d2 * 5 + 4.0 * r2 * c.Lt * d2 + (ch->mode == 2 ? 2.0 * r2 * dd * d2 : 0.0);
        by += r2 * d2 * 5 * h->es + 5.0 * d2 * d2 * h->es;
      }
      Launch l(h, c.st, which == 0 ? "enc.fused_a" : (ch && ch->mode ? "enc.fused_bc+a" : "enc.fused_bc"), fl, by);
      hipError_t e = launch_enclayer(h->prec, q, which, c.st, ch);
      if (e != hipSuccess) c.err = fail(h, DHW_ERR_HIP, "enclayer %s/%d: %s", h->el_name[li].c_str(), which, hipGetErrorString(e));
    }
    tap(c, tap_el(li, 1), ELB(c, li, x2), Lk, d);
    tap(c, tap_el(li, 0), ELB(c, li, out), Lk, d);
    return;
  }
  if (c.rec) { c.rec_fail = true; return; }
  {  // q1 = Wq(x + PE)
    GemmParams p = gp_base(c, Lk, d);
    p.seg[0] = GemmSeg{x, w.w_q1, d, 1, 0};
    p.bias0 = w.b_q1;
    p.posb = w.pb_q1;
    p.posb_cols = d;
    p.out = ELB(c, li, q1);
    run_gemm(c, "enc.q_cross", p);
  }
  {
    AttnParams a{};
    a.Q = ELB(c, li, q1); a.ldq = d;
    a.K = k1p; a.ldk = d; a.koff = 0;
    a.Vt = vt1p; a.lpad = h->lpadT;
    a.text = text; a.ldt = c.Lt;
    a.out = ELB(c, li, a1); a.ldo = d;
    a.B = c.B; a.H = w.heads; a.D = d / w.heads; a.Lq = Lk; a.Lk = c.Lt;
    run_attn(c, "attn.cross", a);
  }
  {  // x2 = FiLM1(LN(dense(a1))) + x
    GemmParams p = gp_base(c, Lk, d);
    p.seg[0] = GemmSeg{ELB(c, li, a1), w.w_d1, d, 1, 0};
    p.bias0 = w.b_d1;
    p.ln = 1;
    set_film(c, p, w.f1, 1);
    p.res2 = x;
    p.out = ELB(c, li, x2);
    run_gemm(c, "enc.dense_cross", p);
  }
  {  // q2,k2 = W(x2 + PE), v2 = Wv x2
    GemmParams p = gp_base(c, Lk, 3 * d);
    p.seg[0] = GemmSeg{ELB(c, li, x2), w.w_qkv2, d, 1, 0};
    p.bias0 = w.b_qkv2;
    p.posb = w.pb_qk2;
    p.posb_cols = 2 * d;
    p.n_store = 2 * d;
    p.out = ELB(c, li, qk2);
    p.vt = ELB(c, li, vt2);
    p.vt_lpad = lpad;
    run_gemm(c, "enc.qkv_self", p);
  }
  {
    AttnParams a{};
    a.Q = ELB(c, li, qk2); a.ldq = 2 * d;
    a.K = ELB(c, li, qk2); a.ldk = 2 * d; a.koff = d;
    a.Vt = ELB(c, li, vt2); a.lpad = lpad;
    a.text = nullptr;
    a.out = ELB(c, li, a2); a.ldo = d;
    a.B = c.B; a.H = w.heads; a.D = d / w.heads; a.Lq = Lk; a.Lk = Lk;
    run_attn(c, "attn.self", a);
  }
  {  // x3 = FiLM2(LN(x2 + dense(a2)))
    GemmParams p = gp_base(c, Lk, d);
    p.seg[0] = GemmSeg{ELB(c, li, a2), w.w_d2, d, 1, 0};
    p.bias0 = w.b_d2;
    p.res1 = ELB(c, li, x2);
    p.ln = 1;
    set_film(c, p, w.f2, 1);
    p.out = ELB(c, li, x3);
    run_gemm(c, "enc.dense_self", p);
  }
  {  // f = SiLU(W1 SiLU(x3) + b1)
    GemmParams p = gp_base(c, Lk, 2 * d);
    p.seg[0] = GemmSeg{ELB(c, li, x3), w.w_f1, d, 1, 1};
    p.bias0 = w.b_f1;
    p.silu_out = 1;
    p.out = ELB(c, li, f);
    run_gemm(c, "enc.ffn1", p);
  }
  {  // out = FiLM3(LN(W2 f + b2 + x3))
    GemmParams p = gp_base(c, Lk, d);
    p.seg[0] = GemmSeg{ELB(c, li, f), w.w_f2, 2 * d, 1, 0};
    p.bias0 = w.b_f2;
    p.res1 = ELB(c, li, x3);
    p.ln = 1;
    set_film(c, p, w.f3, 1);
    p.out = ELB(c, li, out);
    p.pool = pool;
    run_gemm(c, "enc.ffn2", p);
  }
  tap(c, tap_el(li, 1), ELB(c, li, x2), Lk, d);
  tap(c, tap_el(li, 2), ELB(c, li, x3), Lk, d);
  tap(c, tap_el(li, 0), ELB(c, li, out), Lk, d);
}

// sigma-independent prefix of TextStyleEncoder (text_style.py:92-97 up to the LayerNorms; Dropout is identity in eval)
void text_style_static(Ctx& c, const int64_t* text, const float* style) {
  dhw_handle* h = c.h;
  const int c2 = h->dims.c2, dt = 2 * c2;
  RUN_SMALL(c, "cast.style", launch_cast(h->prec, style, (long)c.B * c.S5 * STYLE_CH, WS(c, sty_in), c.st));
  {
    GemmParams p = gp_base(c, c.S5, 4 * c2);
    p.seg[0] = GemmSeg{WS(c, sty_in), h->w_sf1, STYLE_CH, 1, 1};
    p.bias0 = h->b_sf1;
    p.silu_out = 1;
    p.out = WS(c, sty_h);
    run_gemm(c, "style.ffn1", p);
  }
  {
    GemmParams p = gp_base(c, c.S5, dt);
    p.seg[0] = GemmSeg{WS(c, sty_h), h->w_sf3, 4 * c2, 1, 0};
    p.bias0 = h->b_sf3;
    p.ln = 1;
    p.out = WS(c, sty_n);
    run_gemm(c, "style.ffn2_ln", p);
  }
  RUN_SMALL(c, "embed_ln", launch_embed_ln(h->prec, text, c.B * c.Lt, h->emb, dt, true_width(h, dt), VOCAB, WS(c, t_n), c.st));
}

// sigma-dependent part of TextStyleEncoder (text_style.py:94-104) + the per-layer text projections
void text_style_dynamic(Ctx& c) {
  dhw_handle* h = c.h;
  const int c2 = h->dims.c2, dt = 2 * c2;
  const float* g = c.film;
  const float* bt = c.film + h->film_tot;
  const int in_B = c.in_B ? c.in_B : c.B;
  if (h->fuse && h->fuse_text && textside_supported(h->prec, c.Lt, c.S5, dt)) {
    // one workgroup per (step, prompt) pair, every intermediate in LDS (textside.hip)
    TextStyleParams q{};
    q.n = c.B; q.in_B = in_B; q.Lt = c.Lt; q.S5 = c.S5;
    q.sty_n = WS(c, sty_n); q.t_n = WS(c, t_n);
    q.film = c.film; q.film_bs = c.film_bs; q.film_div = c.film_div; q.film_tot = h->film_tot;
    q.f1 = h->f_ts1; q.f2 = h->f_ts2; q.f3 = h->f_ts3; q.f4 = h->f_ts4;
    q.w_q8 = h->w_q8; q.w_kv8 = h->w_kv8; q.w_d8 = h->w_d8; q.w_tf1 = h->w_tf1; q.w_tf3 = h->w_tf3;
    q.b_q8 = h->b_q8; q.b_kv8 = h->b_kv8; q.b_d8 = h->b_d8; q.b_tf1 = h->b_tf1; q.b_tf3 = h->b_tf3;
    q.text_out = TS(c, text_out);
    if (!c.err) {
      const double n = c.B, ddt = dt;
      Launch l(h, c.st, "ts.fused", n * (2.0 * c.S5 * ddt * 2 * ddt + 2.0 * c.Lt * ddt * ddt * 2 + 4.0 * c.Lt * c.S5 * ddt + 2.0 * c.Lt * ddt * 2 * ddt * 2),
               n * c.Lt * ddt * h->es + (double)in_B * (c.S5 + c.Lt) * ddt * h->es + 8.0 * ddt * ddt * h->es);
      hipError_t e = launch_text_style(h->prec, q, c.st);
      if (e != hipSuccess) c.err = fail(h, DHW_ERR_HIP, "text_style fused: %s", hipGetErrorString(e));
    }
    if (!c.planeT) tap(c, TAP_TS, TS(c, text_out), c.Lt, dt);
    for (size_t i = 0; i < h->el.size() && !c.err; ++i) {
      const EncLayerW& w = h->el[i];
      TextLayerParams t{};
      t.n = c.B; t.Lt = c.Lt; t.d = w.d;
      t.text_out = TS(c, text_out);
      t.w_td = w.w_td; t.b_td = w.b_td;
      t.film = c.film; t.film_bs = c.film_bs; t.film_div = c.film_div; t.film_tot = h->film_tot; t.f0 = w.f0;
      t.w_kv = w.w_kv1; t.b_kv = w.b_kv1; t.pb_k1 = w.pb_k1;
      t.k1 = ELT(c, (int)i, k1); t.vt1 = ELT(c, (int)i, vt1); t.lpadT = h->lpadT;
      if (c.err) break;
      t.pairs = h->text_pairs;
      const double n = c.B, dd = w.d;
      Launch l(h, c.st, "enc.text_fused", n * c.Lt * (2.0 * dt * dd + 4.0 * dd * dd), n * c.Lt * (dt + 2.0 * dd) * h->es + (dt * dd + 2.0 * dd * dd) * h->es);
      hipError_t e = launch_text_layer(h->prec, t, c.st);
      if (e != hipSuccess) c.err = fail(h, DHW_ERR_HIP, "text layer %s: %s", h->el_name[i].c_str(), hipGetErrorString(e));
    }
    return;
  }
  RUN_SMALL(c, "film.style", launch_film_apply(h->prec, WS(c, sty_n), in_B, c.B, c.S5, dt, g + h->f_ts1, bt + h->f_ts1, c.film_bs, c.film_div, TS(c, s1), c.st));
  RUN_SMALL(c, "film.text", launch_film_apply(h->prec, WS(c, t_n), in_B, c.B, c.Lt, dt, g + h->f_ts2, bt + h->f_ts2, c.film_bs, c.film_div, TS(c, t1), c.st));
  {
    GemmParams p = gp_text(c, c.Lt, dt);
    p.seg[0] = GemmSeg{TS(c, t1), h->w_q8, dt, 1, 0};
    p.bias0 = h->b_q8;
    p.out = TS(c, q8);
    run_gemm(c, "ts.q", p);
  }
  {
    GemmParams p = gp_base(c, c.S5, 2 * dt);
    p.seg[0] = GemmSeg{TS(c, s1), h->w_kv8, dt, 1, 0};
    p.bias0 = h->b_kv8;
    p.n_store = dt;
    p.out = TS(c, k8);
    p.vt = TS(c, vt8);
    p.vt_lpad = h->lpadS;
    run_gemm(c, "ts.kv", p);
  }
  {
    AttnParams a{};
    a.Q = TS(c, q8); a.ldq = dt;
    a.K = TS(c, k8); a.ldk = dt; a.koff = 0;
    a.Vt = TS(c, vt8); a.lpad = h->lpadS;
    a.out = TS(c, a8); a.ldo = dt;
    a.B = c.B; a.H = 8; a.D = dt / 8; a.Lq = c.Lt; a.Lk = c.S5;
    run_attn(c, "attn.text_style", a);
  }
  {
    GemmParams p = gp_text(c, c.Lt, dt);
    p.seg[0] = GemmSeg{TS(c, a8), h->w_d8, dt, 1, 0};
    p.bias0 = h->b_d8;
    p.res1 = TS(c, t1);
    p.ln = 1;
    set_film(c, p, h->f_ts3, 1);
    p.out = TS(c, t2);
    run_gemm(c, "ts.dense", p);
  }
  {
    GemmParams p = gp_text(c, c.Lt, 2 * dt);
    p.seg[0] = GemmSeg{TS(c, t2), h->w_tf1, dt, 1, 1};
    p.bias0 = h->b_tf1;
    p.silu_out = 1;
    p.out = TS(c, tf_h);
    run_gemm(c, "ts.ffn1", p);
  }
  {
    GemmParams p = gp_text(c, c.Lt, dt);
    p.seg[0] = GemmSeg{TS(c, tf_h), h->w_tf3, 2 * dt, 1, 0};
    p.bias0 = h->b_tf3;
    p.ln = 1;
    set_film(c, p, h->f_ts4, 1);
    p.out = TS(c, text_out);
    run_gemm(c, "ts.ffn2", p);
  }
  if (!c.planeT) {
    tap(c, TAP_TS_STYLE, TS(c, s1), c.S5, dt);
    tap(c, TAP_TS_T2, TS(c, t2), c.Lt, dt);
    tap(c, TAP_TS, TS(c, text_out), c.Lt, dt);
  }
  for (size_t i = 0; i < h->el.size(); ++i) enc_layer_text(c, (int)i, h->el[i]);
}

// the stroke path of DiffusionModel.forward (model.py:139-182); the heads are launched by the caller
void stroke_path(Ctx& c, const float* strokes, const int64_t* text) {
  dhw_handle* h = c.h;
  const dhw_dims& d = h->dims;
  const int L = c.L, dt = 2 * d.c2;
  const bool fin = c.fuse_input && h->fuse;
  if (!fin) {
    RUN_SMALL(c, "input_dense", launch_input_dense(h->prec, strokes, (long)c.B * L, h->in_w, h->in_b, d.c1, WS(c, x0), c.st));
    tap(c, TAP_INPUT_DENSE, WS(c, x0), L, d.c1);
  }
  conv_block(c, CB_ENC1, h->enc1, WS(c, x0), L, CBB(c, CB_ENC1, out), false, WS(c, enc1_pool), fin ? strokes : nullptr);
  // Everything between two self-attentions is row-local: enc2 / enc4 continue into the first half of enc3 / enc5,
  // enc5's second half into AvgPool + att_dense + the first attention layer's first half, and every attention layer's
  // second half into the next layer's first half.
  const bool chain_ok = h->fuse && h->chain && h->prec == PREC_BF16;
  const int nl = d.num_layers;
  bool a3 = false, a5 = false;
  // enc2 / enc4 can continue into enc3.a / enc5.a the same way (bit 0 / bit 1), but the ConvBlock's row tiling (62 / 46 rows)
  // is a worse fit for the layer than its own: r1 measured 23.22 ms (off) / 23.21 (enc3) / 23.40 (enc5, both); r3, after the kernels
  // changed: 19.54 (off) / 19.40 (enc3: bit 0) / 19.62 (enc5: bit 1) / 19.45 (both), three alternating runs each -> enc3 only
  // r5: with enc4 on the asymmetric 32-row tiles (B = 64 at L / 4 = 122: the layer's own tiling) the enc5 chain wins, 18.02 -> 17.89 ms: the default
  // (no DHW_CHAIN_CONV) takes bit 1 exactly there (convblock_chain_auto); an explicit DHW_CHAIN_CONV forces / forbids it for any tiling
  static const bool conv_chain_env = getenv("DHW_CHAIN_CONV") != nullptr;
  static const int conv_chain = conv_chain_env ? atoi(getenv("DHW_CHAIN_CONV")) : 3;
  {
    EncChain ch{};
    if (chain_ok && (conv_chain & 1)) { ch.mode = 1; ch.a = enc_params(c, 0, h->el[0], nullptr, L / 2, h->lpadX[0], text, nullptr); }
    conv_block(c, CB_ENC2, h->enc2, WS(c, enc1_pool), L / 2, CBB(c, CB_ENC2, out), false, nullptr, nullptr, nullptr, ch.mode ? &ch : nullptr, &a3);
  }
  enc_layer(c, 0, h->el[0], CBB(c, CB_ENC2, out), L / 2, h->lpadX[0], text, WS(c, enc3_pool), a3);
  {
    EncChain ch{};
    // (record mode: the persistent step kernel has enc4 and enc5.a as two phases)
    if (chain_ok && (conv_chain & 2) && !c.rec) { ch.mode = 1; ch.a = enc_params(c, 1, h->el[1], nullptr, L / 4, h->lpadX[1], text, nullptr); }
    conv_block(c, CB_ENC4, h->enc4, WS(c, enc3_pool), L / 4, CBB(c, CB_ENC4, out), false, nullptr, nullptr, nullptr, ch.mode ? &ch : nullptr, &a5, !conv_chain_env);
  }
  EncChain ch5{};
  if (chain_ok && nl > 0 && enclayer_supported(h->prec, dt, h->el[2].heads) && enclayer_chain_supported(h->prec, d.c3, c.B, L / 4, 2, dt)) {
    ch5.mode = 2;
    ch5.a = enc_params(c, 2, h->el[2], nullptr, L / 8, h->lpadX[2], text, nullptr);
    ch5.w_dense = h->w_attd; ch5.b_dense = h->b_attd; ch5.dense_out = WS(c, att_dense);
  }
  enc_layer(c, 1, h->el[1], CBB(c, CB_ENC4, out), L / 4, h->lpadX[1], text, WS(c, enc5_pool), a5, ch5.mode ? &ch5 : nullptr,
            ch5.mode ? 32 : 0);
  if (!ch5.mode) {
    GemmParams p = gp_base(c, L / 8, dt);
    p.seg[0] = GemmSeg{WS(c, enc5_pool), h->w_attd, d.c3, 1, 0};
    p.bias0 = h->b_attd;
    p.out = WS(c, att_dense);
    run_gemm(c, "att_dense", p);
  }
  tap(c, TAP_ATT_DENSE, WS(c, att_dense), L / 8, dt);
  const void* x = WS(c, att_dense);
  bool a_done = ch5.mode != 0;   // this layer's first half was evaluated by the previous launch
  for (int i = 0; i < nl; ++i) {
    EncChain chn{};
    if (chain_ok && i + 1 < nl && enclayer_chain_supported(h->prec, dt, c.B, L / 8, 1, dt)) {
      chn.mode = 1;
      chn.a = enc_params(c, 3 + i, h->el[3 + i], nullptr, L / 8, h->lpadX[2], text, nullptr);
    }
    enc_layer(c, 2 + i, h->el[2 + i], x, L / 8, h->lpadX[2], text, nullptr, a_done, chn.mode ? &chn : nullptr);
    a_done = chn.mode != 0;
    x = ELB(c, 2 + i, out);
  }
  // decoder: x = Upsample(previous) + skip_conv(encoder output of the same resolution), then the ConvBlock (model.py:169-175)
  struct UP { int tap; const void* skip_in; void* w; float* b; int cin, cout, L; const void* low; int cb; };
  const UP ups[3] = {
      {TAP_UP3, ELB(c, 1, out), h->w_sk3, h->b_sk3, d.c3, dt, L / 4, x, CB_DEC3},
      {TAP_UP2, ELB(c, 0, out), h->w_sk2, h->b_sk2, d.c2, d.c3, L / 2, CBB(c, CB_DEC3, out), CB_DEC2},
      {TAP_UP1, CBB(c, CB_ENC1, out), h->w_sk1, h->b_sk1, d.c1, d.c2, L, CBB(c, CB_DEC2, out), CB_DEC1}};
  const ConvBlockW* decs[3] = {&h->dec3, &h->dec2, &h->dec1};
  const bool fup = h->fuse && h->fuse_up && h->prec == PREC_BF16;
  for (int i = 0; i < 3; ++i) {
    const UP& u = ups[i];
    if (fup) {   // the decoder block evaluates upsample(x) + skip_conv(h) while staging its input
      const UpIn in{u.skip_in, u.w, u.b, u.cin, u.low};
      h->taps[u.tap].set = false;
      conv_block(c, u.cb, *decs[i], nullptr, u.L, CBB(c, u.cb, out), i == 2, nullptr, nullptr, &in);
      continue;
    }
    void* xd = need(c, c.ws->xd[i], "xd");
    GemmParams p = gp_base(c, u.L, u.cout);   // upsample(x) + skip_conv(h)  (model.py:169-175)
    p.seg[0] = GemmSeg{u.skip_in, u.w, u.cin, 3, 0};
    p.bias0 = u.b;
    p.res2 = u.low;
    p.res2_half = 1;
    p.out = xd;
    run_gemm(c, "skip_conv_up", p);
    tap(c, u.tap, xd, u.L, u.cout);
    conv_block(c, u.cb, *decs[i], xd, u.L, CBB(c, u.cb, out), i == 2, nullptr);
  }
}

int check_shapes(dhw_handle* h, int B, int L, int Lt) {
  const dhw_dims& d = h->dims;
  if (B < 1 || B > d.max_B || L < 8 || L > d.max_L || L % 8 || Lt < 1 || Lt > d.max_Lt)
    return fail(h, DHW_ERR_ARG, "shape out of range: B=%d (max %d) L=%d (max %d, multiple of 8) Lt=%d (max %d)", B, d.max_B, L, d.max_L, Lt, d.max_Lt);
  return 0;
}

int ensure_film_T(dhw_handle* h, int T, dhw_handle::FilmT** out) {
  dhw_handle::FilmT& ft = h->film_T[T];
  *out = &ft;
  if (ft.d_film) return 0;
  int rc;
  if ((rc = dev_alloc(h, (void**)&ft.d_sigma, (size_t)T * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&ft.d_sig32, (size_t)T * SIG * 4))) return rc;
  if ((rc = dev_alloc(h, (void**)&ft.d_film, (size_t)T * 2 * h->film_tot * 4))) return rc;
  return 0;
}

void schedule_host(int T, std::vector<float>& beta, std::vector<float>& alpha) {
  // utils/nn.py:19-39 in fp32: torch.linspace evaluates fma(step, i, start) below the midpoint and
  // fma(-step, n-1-i, end) above it (probed against torch 2.10 CPU); then exp, + 0.02,
  // cumprod(1 - beta) (inference.py:81).
  beta.resize(T);
  alpha.resize(T);
  const float lo = (float)std::log(1e-5), hi = (float)std::log(0.4);
  const float step = T > 1 ? (hi - lo) / (float)(T - 1) : 0.f;
  const int half = T / 2;
  double a = 1.0;   // torch's CPU cumprod accumulates float inputs in double (acc_type) and rounds each output
  for (int i = 0; i < T; ++i) {
    // (a one-point linspace is its START: torch.linspace(a, b, 1) = [a]; the symmetric form alone gave the end point for T = 1)
    const float x = T == 1 ? lo : i < half ? fmaf(step, (float)i, lo) : fmaf(-step, (float)(T - 1 - i), hi);
    beta[i] = 0.02f + expf(x);
    a = a * (double)(1.0f - beta[i]);
    alpha[i] = (float)a;
  }
}



void destroy_impl(dhw_handle* h);

// Names are resolved HERE, once per handle: the EncoderLayers' module names and the table dhw_debug_read searches.
void build_names(dhw_handle* h) {
  const int nel = 2 + h->dims.num_layers;
  h->el_name.clear();
  h->el_name.push_back("enc3");
  h->el_name.push_back("enc5");
  for (int i = 0; i < h->dims.num_layers; ++i) h->el_name.push_back("att_layers." + std::to_string(i));
  h->taps.assign(TAP_CONV0 + CB_N + 3 * nel, TapSlot{});
  h->taps[TAP_SIGMA_FFN].name = "sigma_ffn";
  h->taps[TAP_INPUT_DENSE].name = "input_dense";
  h->taps[TAP_TS].name = "text_style_model";
  h->taps[TAP_TS_STYLE].name = "text_style_model.style";
  h->taps[TAP_TS_T2].name = "text_style_model.t2";
  h->taps[TAP_ATT_DENSE].name = "att_dense";
  h->taps[TAP_UP3].name = "skip_conv3+up";
  h->taps[TAP_UP2].name = "skip_conv2+up";
  h->taps[TAP_UP1].name = "skip_conv1+up";
  for (int i = 0; i < CB_N; ++i) h->taps[tap_conv(i)].name = kConvName[i];
  for (int li = 0; li < nel; ++li) {
    h->taps[tap_el(li, 0)].name = h->el_name[li];
    h->taps[tap_el(li, 1)].name = h->el_name[li] + ".x2";
    h->taps[tap_el(li, 2)].name = h->el_name[li] + ".x3";
  }
}

// Every per-call buffer of a freshly allocated workspace must exist (the all-steps plane comes later, ensure_plane): a buffer
// the allocation code forgot is a dhw_create error, not something a launch discovers.
int verify_workspace(dhw_handle* h, const Workspace& w) {
  std::vector<std::pair<const char*, const void*>> all = {
      {"sty_in", w.sty_in}, {"sty_h", w.sty_h}, {"sty_n", w.sty_n}, {"t_n", w.t_n}, {"s1", w.ts.s1}, {"k8", w.ts.k8}, {"vt8", w.ts.vt8},
      {"t1", w.ts.t1}, {"q8", w.ts.q8}, {"a8", w.ts.a8}, {"t2", w.ts.t2}, {"tf_h", w.ts.tf_h}, {"text_out", w.ts.text_out}, {"x0", w.x0},
      {"enc1.pool", w.enc1_pool}, {"enc3.pool", w.enc3_pool}, {"enc5.pool", w.enc5_pool}, {"att_dense", w.att_dense},
      {"xd3", w.xd[0]}, {"xd2", w.xd[1]}, {"xd1", w.xd[2]}, {"x_t", w.d_xt}};
  for (int i = 0; i < CB_N; ++i) { all.push_back({"convblock h1", w.cb[i].h1}); all.push_back({"convblock h2", w.cb[i].h2}); all.push_back({"convblock out", w.cb[i].out}); }
  if ((int)w.el.size() != 2 + h->dims.num_layers) return fail(h, DHW_ERR_INTERNAL, "internal: workspace has %d EncoderLayers, the model %d", (int)w.el.size(), 2 + h->dims.num_layers);
  for (const EncBufs& e : w.el)
    for (auto kv : std::initializer_list<std::pair<const char*, const void*>>{{"tl", e.t.tl}, {"k1", e.t.k1}, {"vt1", e.t.vt1}, {"q1", e.q1}, {"a1", e.a1}, {"x2", e.x2},
                                                                               {"qk2", e.qk2}, {"vt2", e.vt2}, {"a2", e.a2}, {"x3", e.x3}, {"f", e.f}, {"out", e.out}})
      all.push_back(kv);
  for (auto& kv : all)
    if (!kv.second) return fail(h, DHW_ERR_INTERNAL, "internal: workspace buffer '%s' was not allocated", kv.first);
  return 0;
}

}  // namespace

// ================================================================= C-ABI
extern "C" {

const char* dhw_version(void) { return "dhw-hip 0.1 (gfx950)"; }

const char* dhw_last_error(dhw_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int dhw_schedule(int T, float* beta_out, float* alpha_bar_out) {
  DHW_GUARD(nullptr, "dhw_schedule", int, {
    if (T < 1 || !beta_out || !alpha_bar_out) return fail(nullptr, DHW_ERR_ARG, "dhw_schedule: bad args");
    std::vector<float> b, a;
    schedule_host(T, b, a);
    std::memcpy(beta_out, b.data(), T * 4);
    std::memcpy(alpha_bar_out, a.data(), T * 4);
    return 0;
  });
}

int dhw_create(dhw_handle** out, const dhw_dims* dims, int device) {
  DHW_GUARD(nullptr, "dhw_create", int, {
    if (!out || !dims) return fail(nullptr, DHW_ERR_ARG, "dhw_create: null argument");
    *out = nullptr;
    const dhw_dims& d = *dims;
    if (d.c1 != 128 || d.c3 != 256) return fail(nullptr, DHW_ERR_ARG, "c1 must be 128 and c3 256 (reference conditioning.py:9-10, model.py:103)");
    if (d.c2 < 12 || d.c2 > 192 || d.c2 % 12)
      return fail(nullptr, DHW_ERR_ARG, "c2 must be a multiple of 12 (model.py:88-106: 3 / 6 / 8 attention heads) and at most 192, the width the kernels are built for");
    if (d.num_layers < 0 || d.num_layers > 16 || d.max_B < 1 || d.max_L < 8 || d.max_L % 8 || d.max_Lt < 1 || d.S < 1 || (d.S * 1280) % STYLE_CH)
      return fail(nullptr, DHW_ERR_ARG, "dhw_create: bad dims");
    if (d.precision != DHW_PREC_BF16 && d.precision != DHW_PREC_F32) return fail(nullptr, DHW_ERR_ARG, "bad precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, DHW_ERR_HIP, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, DHW_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    // (owned here until the handle is complete: an exception on the way — caught by the guard — must not leak it)
    struct Hold {
      dhw_handle* p;
      ~Hold() { if (p) dhw_destroy(p); }
    } hold{new dhw_handle()};
    dhw_handle* h = hold.p;
    h->ldims = d;
    h->dims = d;
    h->dims.c2 = 192;
    h->padded = d.c2 != 192;
    h->device = device;
    h->prec = d.precision == DHW_PREC_F32 ? PREC_F32 : PREC_BF16;
    h->es = h->prec == PREC_F32 ? 4 : 2;
    h->spec = build_spec(d.num_layers, d.c1, d.c2, d.c3);
    h->pspec = build_spec(d.num_layers, d.c1, h->dims.c2, d.c3);
    for (size_t i = 0; i < h->spec.size(); ++i) h->key_index[h->spec[i].key] = (int)i;
    h->host_w.resize(h->spec.size());
    h->loaded.assign(h->spec.size(), 0);
    build_film_layout(h);
    build_names(h);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, DHW_ERR_HIP, "hipSetDevice failed");
    {
      const char* e = getenv("DHW_STREAMS");
      if (e && atoi(e) >= 1) h->nstreams = std::min(atoi(e), MAX_STREAMS);
      h->nstreams = std::max(1, std::min(h->nstreams, d.max_B));
      h->nstreams_alloc = h->nstreams;
    }
    int rc = alloc_shared(h);
    h->ws.resize(h->nstreams);
    // ws[0] serves dhw_forward at the full batch; the others only ever see ceil(max_B / nstreams) prompts
    for (int i = 0; !rc && i < h->nstreams; ++i)
      if (!(rc = alloc_workspace(h, h->ws[i], i == 0 ? d.max_B : (d.max_B + h->nstreams - 1) / h->nstreams))) rc = verify_workspace(h, h->ws[i]);
    for (int i = 1; !rc && i < h->nstreams; ++i)
      if (hipStreamCreateWithFlags(&h->sub_streams[i], hipStreamNonBlocking) != hipSuccess) rc = fail(h, DHW_ERR_HIP, "stream create failed");
    if (const char* e = getenv("DHW_FUSE")) h->fuse = atoi(e) != 0;
    if (const char* e = getenv("DHW_PLANE")) h->plane = atoi(e) != 0;
    if (const char* e = getenv("DHW_FUSE_TEXT")) h->fuse_text = atoi(e) != 0;
    if (const char* e = getenv("DHW_FUSE_HEADS")) h->fuse_heads = atoi(e) != 0;
    if (const char* e = getenv("DHW_FUSE_UP")) h->fuse_up = atoi(e) != 0;
    if (const char* e = getenv("DHW_CHAIN")) h->chain = atoi(e) != 0;
    if (const char* e = getenv("DHW_PERSIST")) h->persist = atoi(e) != 0;
    if (const char* e = getenv("DHW_TEXT_PAIRS")) h->text_pairs = atoi(e) == 1 ? 1 : atoi(e) == 2 ? 2 : 0;
    if (h->padded) h->fuse = false;   // (pad_weights: the fused block kernels have compile-time LayerNorm widths)
    if (!rc && h->prec == PREC_BF16 && h->persist) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) != hipSuccess || persist_init() != hipSuccess) rc = fail(h, DHW_ERR_HIP, "persistent kernel setup failed: %s", hipGetErrorString(hipGetLastError()));
      else {
        h->persist_grid = prop.multiProcessorCount;
        if (hipHostMalloc((void**)&h->h_step_err, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&h->d_step_err, h->h_step_err, 0) != hipSuccess)
          rc = fail(h, DHW_ERR_HIP, "host-mapped error word: %s", hipGetErrorString(hipGetLastError()));
        else *h->h_step_err = 0;
      }
    }
    if (!rc && enclayer_init() != hipSuccess) rc = fail(h, DHW_ERR_HIP, "kernel attribute setup failed: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && convblock_init() != hipSuccess) rc = fail(h, DHW_ERR_HIP, "kernel attribute setup failed: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && textside_init() != hipSuccess) rc = fail(h, DHW_ERR_HIP, "kernel attribute setup failed: %s", hipGetErrorString(hipGetLastError()));
    if (!rc && gemm_init() != hipSuccess) rc = fail(h, DHW_ERR_HIP, "kernel attribute setup failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc) { ErrBuf keep = h->err; g_err = keep; return rc; }   // (~Hold destroys the half-built handle; its message survives in the global slot)
    hold.p = nullptr;
    *out = h;
    return 0;
  });
}

void dhw_destroy(dhw_handle* h) {
  if (!h) return;
  try {
    destroy_impl(h);
  } catch (...) {   // (nothing below is expected to throw; the ABI's promise holds regardless)
  }
}

}  // extern "C"

namespace {
void destroy_impl(dhw_handle* h) {
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
  for (auto& r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (int i = 1; i < MAX_STREAMS; ++i)
    if (h->sub_streams[i]) hipStreamDestroy(h->sub_streams[i]);
  for (void* p : h->allocs) hipFree(p);
  if (h->h_step_err) hipHostFree(h->h_step_err);
  delete h;
}
}  // namespace

extern "C" {

int dhw_num_keys(dhw_handle* h) { return h ? (int)h->spec.size() : DHW_ERR_ARG; }

int dhw_key_info(dhw_handle* h, int i, const char** key, int64_t shape[3], int* ndim) {
  DHW_GUARD(h, "dhw_num_keys", int, {
    if (!h || i < 0 || i >= (int)h->spec.size()) return fail(h, DHW_ERR_ARG, "dhw_key_info: index out of range");
    const KeySpec& k = h->spec[i];
    if (key) *key = k.key.c_str();
    if (ndim) *ndim = (int)k.shape.size();
    if (shape) for (size_t j = 0; j < 3; ++j) shape[j] = j < k.shape.size() ? k.shape[j] : 1;
    return 0;
  });
}

int dhw_load(dhw_handle* h, const char* key, const void* host_ptr, int dtype, const int64_t* shape, int ndim) {
  DHW_GUARD(h, "dhw_load", int, {
    if (!h || !key || !host_ptr || !shape) return fail(h, DHW_ERR_ARG, "dhw_load: null argument");
    auto it = h->key_index.find(key);
    if (it == h->key_index.end()) return fail(h, DHW_ERR_KEY, "unexpected key in state_dict: %s", key);
    const KeySpec& k = h->spec[it->second];
    bool ok = ndim == (int)k.shape.size();
    for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == k.shape[i];
    if (!ok) return fail(h, DHW_ERR_KEY, "size mismatch for %s", key);
    size_t n = 1;
    for (int64_t s : k.shape) n *= (size_t)s;
    std::vector<float>& dst = h->host_w[it->second];
    dst.resize(n);
    switch (dtype) {
      case DHW_F32: std::memcpy(dst.data(), host_ptr, n * 4); break;
      case DHW_BF16: for (size_t i = 0; i < n; ++i) dst[i] = bf2f(((const uint16_t*)host_ptr)[i]); break;
      case DHW_F16: for (size_t i = 0; i < n; ++i) dst[i] = h2f(((const uint16_t*)host_ptr)[i]); break;
      case DHW_F64: for (size_t i = 0; i < n; ++i) dst[i] = (float)((const double*)host_ptr)[i]; break;
      default: return fail(h, DHW_ERR_ARG, "dhw_load: unknown dtype %d", dtype);
    }
    h->loaded[it->second] = 1;
    h->packed = false;
    return 0;
  });
}

#define UPF(dst, key) if ((rc = upload_f32(h, W(h, key), &h->dst))) return rc
int dhw_finalize(dhw_handle* h) {
  DHW_GUARD(h, "dhw_finalize", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    if (h->packed) return 0;
    for (size_t i = 0; i < h->spec.size(); ++i)
      if (!h->loaded[i]) return fail(h, DHW_ERR_KEY, "missing key in state_dict: %s", h->spec[i].key.c_str());
    HIPCK(h, hipSetDevice(h->device));
    HIPCK(h, hipDeviceSynchronize());
    for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);   // device is idle here (synchronised above)
    h->graphs.clear();
    const dhw_dims& d = h->dims;
    const int c1 = d.c1, c2 = d.c2, c3 = d.c3, dt = 2 * c2;
    int rc;
    if (h->padded && (rc = pad_weights(h))) return rc;
    // (re-packing leaks the previous packed copies until destroy; weights are loaded once in practice)
    {  // FiLM: all gamma/beta projections concatenated -> [2*TOT, 32]
      std::vector<float> w((size_t)2 * h->film_tot * SIG), b((size_t)2 * h->film_tot);
      for (auto& kv : h->film_off) {
        const auto& gw = W(h, kv.first + ".gamma_emb.weight");
        const auto& gb = W(h, kv.first + ".gamma_emb.bias");
        const auto& bw = W(h, kv.first + ".beta_emb.weight");
        const auto& bb = W(h, kv.first + ".beta_emb.bias");
        std::copy(gw.begin(), gw.end(), w.begin() + (size_t)kv.second * SIG);
        std::copy(gb.begin(), gb.end(), b.begin() + kv.second);
        std::copy(bw.begin(), bw.end(), w.begin() + (size_t)(h->film_tot + kv.second) * SIG);
        std::copy(bb.begin(), bb.end(), b.begin() + h->film_tot + kv.second);
      }
      if ((rc = upload_f32(h, w, &h->d_film_w))) return rc;
      if ((rc = upload_f32(h, b, &h->d_film_b))) return rc;
    }
    UPF(sg_w1, "sigma_ffn.1.weight"); UPF(sg_b1, "sigma_ffn.1.bias"); UPF(sg_w2, "sigma_ffn.3.weight"); UPF(sg_b2, "sigma_ffn.3.bias");
    UPF(in_w, "input_dense.weight"); UPF(in_b, "input_dense.bias");
    UPF(out_w, "output_dense.weight"); UPF(out_b, "output_dense.bias");
    UPF(pen_w, "pen_lifts_dense.0.weight"); UPF(pen_b, "pen_lifts_dense.0.bias");
    UPF(emb, "text_style_model.emb.weight");
    const std::string t = "text_style_model";
    UPF(b_sf1, t + ".style_ffn.1.bias"); UPF(b_sf3, t + ".style_ffn.3.bias"); UPF(b_q8, t + ".mha.wq.bias");
    UPF(b_d8, t + ".mha.dense.bias"); UPF(b_tf1, t + ".text_ffn.1.bias"); UPF(b_tf3, t + ".text_ffn.3.bias");
    UPF(b_attd, "att_dense.bias"); UPF(b_sk1, "skip_conv1.bias"); UPF(b_sk2, "skip_conv2.bias"); UPF(b_sk3, "skip_conv3.bias");
    if ((rc = upload_f32(h, vcat({&W(h, t + ".mha.wk.bias"), &W(h, t + ".mha.wv.bias")}), &h->b_kv8))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".style_ffn.1.weight"), 4 * c2, STYLE_CH, &h->w_sf1))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".style_ffn.3.weight"), dt, 4 * c2, &h->w_sf3))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".mha.wq.weight"), dt, dt, &h->w_q8))) return rc;
    if ((rc = upload_packed(h, vcat({&W(h, t + ".mha.wk.weight"), &W(h, t + ".mha.wv.weight")}), 2 * dt, dt, &h->w_kv8))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".mha.dense.weight"), dt, dt, &h->w_d8))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".text_ffn.1.weight"), 2 * dt, dt, &h->w_tf1))) return rc;
    if ((rc = upload_packed(h, W(h, t + ".text_ffn.3.weight"), dt, 2 * dt, &h->w_tf3))) return rc;
    if ((rc = upload_packed(h, W(h, "att_dense.weight"), dt, 2 * c1, &h->w_attd))) return rc;
    if ((rc = upload_packed(h, conv_flat(W(h, "skip_conv1.weight"), c2, c1), c2, 3 * c1, &h->w_sk1))) return rc;
    if ((rc = upload_packed(h, conv_flat(W(h, "skip_conv2.weight"), c3, c2), c3, 3 * c2, &h->w_sk2))) return rc;
    if ((rc = upload_packed(h, conv_flat(W(h, "skip_conv3.weight"), dt, c3), dt, 3 * c3, &h->w_sk3))) return rc;
    h->f_ts1 = film_offset(h, t + ".affine1");
    h->f_ts2 = film_offset(h, t + ".affine2");
    h->f_ts3 = film_offset(h, t + ".affine3");
    h->f_ts4 = film_offset(h, t + ".affine4");
    if ((rc = pack_convblock(h, "enc1", c1, c1, h->enc1))) return rc;
    if ((rc = pack_convblock(h, "enc2", c1, c2, h->enc2))) return rc;
    if ((rc = pack_convblock(h, "enc4", c2, c3, h->enc4))) return rc;
    if ((rc = pack_convblock(h, "dec3", dt, c3, h->dec3))) return rc;
    if ((rc = pack_convblock(h, "dec2", c3, c2, h->dec2))) return rc;
    if ((rc = pack_convblock(h, "dec1", c2, c1, h->dec1))) return rc;
    h->el.assign(2 + d.num_layers, EncLayerW{});
    if ((rc = pack_enclayer(h, "enc3", c2, 3, 4.0f, d.max_L / 2, h->el[0]))) return rc;   // model.py:88
    if ((rc = pack_enclayer(h, "enc5", c3, 4, 2.0f, d.max_L / 4, h->el[1]))) return rc;   // model.py:90
    for (int i = 0; i < d.num_layers; ++i)
      if ((rc = pack_enclayer(h, "att_layers." + std::to_string(i), dt, 6, 1.0f, d.max_L / 8, h->el[2 + i]))) return rc;   // model.py:104-109
    HIPCK(h, hipDeviceSynchronize());
    if (h->lookup_fail) return DHW_ERR_INTERNAL;   // (message set by W / film_offset)
    h->packed = true;
    for (auto& kv : h->film_T) kv.second.ready = false;
    return 0;
  });
}

#undef UPF

static int launch_heads_for(Ctx& c, HeadsParams hp) {
  dhw_handle* h = c.h;
  hp.x = (const float*)CBB(c, CB_DEC1, out);
  hp.rows = (long)c.B * c.L;
  hp.C = h->dims.c1;
  hp.w_out = h->out_w; hp.b_out = h->out_b; hp.w_pen = h->pen_w; hp.b_pen = h->pen_b;
  hp.L = c.L;
  RUN_SMALL(c, "heads_step", launch_heads(hp, c.st));
  return c.err;
}

int dhw_forward(dhw_handle* h, const float* strokes, const int64_t* text, const float* sigma, const float* style,
                int B, int L, int Lt, float* eps_out, float* pen_out, void* hip_stream) {
  DHW_GUARD(h, "dhw_forward", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    if (!strokes || !text || !sigma || !style || !eps_out || !pen_out) return fail(h, DHW_ERR_ARG, "dhw_forward: null pointer");
    int rc = check_shapes(h, B, L, Lt);
    if (rc) return rc;
    if ((rc = dhw_finalize(h))) return rc;
    HIPCK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    Ctx c{h, &h->ws[0], st, B, L, Lt, h->dims.S * 5, h->d_film, 2L * h->film_tot};
    taps_clear(h);
    RUN_SMALL(c, "sigma_ffn", launch_sigma_ffn(sigma, B, h->sg_w1, h->sg_b1, h->sg_w2, h->sg_b2, h->d_sig32, st));
    RUN_SMALL(c, "film_table", launch_film(h->d_sig32, B, h->d_film_w, h->d_film_b, 2 * h->film_tot, h->d_film, st));
    tap(c, TAP_SIGMA_FFN, h->d_sig32, 1, SIG, true);
    text_style_static(c, text, style);
    text_style_dynamic(c);
    stroke_path(c, strokes, text);
    HeadsParams hp{};
    hp.eps = eps_out;
    hp.pen = pen_out;
    launch_heads_for(c, hp);
    h->last_B = B; h->last_L = L; h->last_Lt = Lt;
    return c.err;
  });
}

// sampler steps whose text side is precomputed together (bounds the plane's memory for long schedules)
static int plane_chunk(int T) { return std::min(T, 64); }

// One prompt sub-batch [b0, b0+Bs) of a B-prompt batch, enqueued on `st` with workspace `w`.
// d_plans: device array of T StepPlans (persist.h) -> every denoiser call is ONE persistent launch; rec_out: record mode —
// nothing is launched, the T plans are built on the host (rec_out->size() != T afterwards: this shape has no persistent form).
static int sample_enqueue(dhw_handle* h, Workspace* w, int b0, int Bs, int B, const int64_t* text, const float* style,
                          int L, int Lt, int T, int mode, const float* noise, float* out, hipStream_t st,
                          const std::vector<float>& beta, const std::vector<float>& alpha, const StepPlan* d_plans = nullptr,
                          std::vector<StepPlan>* rec_out = nullptr) {
  const long rows = (long)Bs * L;
  const size_t step_stride = (size_t)B * L * 2;   // one noise draw for the whole batch
  text += (size_t)b0 * Lt;
  style += (size_t)b0 * h->dims.S * 1280;
  out += (size_t)b0 * L * 3;
  if (noise) noise += (size_t)b0 * L * 2;
  Ctx c{h, w, st, Bs, L, Lt, h->dims.S * 5, h->d_film_T, 0};
  c.fuse_input = true;
  // x_T
  if (rec_out) {
  } else if (noise) {
    hipError_t e = hipMemcpyAsync(w->d_xt, noise, rows * 2 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "memcpy x_T: %s", hipGetErrorString(e));
  } else {
    RUN_SMALL(c, "randn_init", launch_randn_init(w->d_xt, rows, L, h->d_seed, b0, st));
  }
  if (!rec_out) text_style_static(c, text, style);   // sigma-independent: once per sample batch, not per step
  const int TC = plane_chunk(T);
  for (int step = 0, i = T - 1; i >= 0; --i, ++step) {
    if (!rec_out && h->teach_every > 0 && step > 0 && step % h->teach_every == 0) {
      // teacher forcing (tests only): x after `step` steps -> capture[k], x := reset[k]
      const size_t k = (size_t)(step / h->teach_every - 1), off = (k * B + b0) * (size_t)L * 2;
      hipError_t e = hipMemcpyAsync(h->teach_capture + off, w->d_xt, rows * 2 * 4, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(w->d_xt, h->teach_reset + off, rows * 2 * 4, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "teacher copy: %s", hipGetErrorString(e));
    }
    if (!rec_out && h->plane && step % TC == 0) {
      // The text side (TextStyleEncoder + every layer's text K/V, text_style.py:91-104, model.py:38-42) depends on
      // (text, style, sigma_i) only and the sigma schedule is known: evaluate it for the next `ns` steps in ONE
      // batched pass (ns*Bs "samples", FiLM row per step) instead of 16 small launches inside every step.
      const int ns = std::min(TC, T - step);
      Ctx cp = c;
      cp.B = ns * Bs;
      cp.in_B = Bs;
      cp.film = h->d_film_T + (size_t)i * 2 * h->film_tot;   // step `step + k` uses schedule index i - k
      cp.film_bs = -2L * h->film_tot;
      cp.film_div = Bs;
      cp.planeT = true;
      text_style_dynamic(cp);
      if (cp.err) return cp.err;
    }
    c.film = h->d_film_T + (size_t)i * 2 * h->film_tot;
    c.use_plane = h->plane;
    c.plane_step = step % TC;
    if (!h->plane) {
      if (rec_out) return 0;   // (the persistent form reads the text K/V from the all-steps plane)
      text_style_dynamic(c);
    }
    HeadsParams hp{};
    hp.eps = nullptr;
    hp.pen = nullptr;
    hp.xt = w->d_xt;
    hp.z = noise ? noise + (size_t)(1 + step) * step_stride : nullptr;
    hp.mode = mode;
    hp.seed_ptr = h->d_seed;
    hp.sample_off = b0;
    hp.iter = step;
    const float a = alpha[i], b = beta[i];
    const float a_next = i > 1 ? alpha[i - 1] : 1.0f;   // inference.py:87
    hp.k0 = sqrtf(1.0f - a);
    if (mode == 0) {
      hp.k1 = sqrtf(1.0f - b);
      hp.k2 = sqrtf(1.0f - a_next);
      hp.add_noise = 1;
    } else {
      hp.k1 = 1.0f / sqrtf(1.0f - b);
      hp.k2 = sqrtf(b);
      hp.k3 = b;
      hp.add_noise = i != 0;   // inference.py:92
    }
    if (i == 0) hp.out3 = out;
    const bool fh = h->fuse && h->fuse_heads;
    c.fhp = fh ? &hp : nullptr;
    if (rec_out) {
      if (!fh) return 0;
      std::vector<StepPhase> phases;
      c.rec = &phases;
      c.rec_fail = false;
      stroke_path(c, w->d_xt, text);
      c.rec = nullptr;
      if (c.rec_fail || c.err || phases.empty()) return c.err;
      StepPlan sp{};
      sp.nphase = (int)phases.size();
      sp.B = Bs;
      sp.spx = (Bs + STEP_XCDS - 1) / STEP_XCDS;
      sp.sync = h->d_step_sync;
      sp.err = h->d_step_err;
      for (size_t k = 0; k < phases.size(); ++k) { sp.ph[k] = phases[k]; sp.cum_tps[k + 1] = sp.cum_tps[k] + phases[k].tps; }
      rec_out->push_back(sp);
      continue;
    }
    if (d_plans) {
      Launch l(h, st, "step.persistent");
      hipError_t e = launch_step(d_plans + step, h->persist_grid, st);
      if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "persistent step %d: %s", step, hipGetErrorString(e));
      continue;
    }
    stroke_path(c, w->d_xt, text);
    if (!fh) launch_heads_for(c, hp);
    if (c.err) return c.err;
  }
  return c.err;
}

// The StepPlans of one dhw_sample shape (cached like the graphs): built by running the enqueue in record mode, uploaded once.
// Returns null when this shape / configuration has no persistent form (the caller then launches kernel by kernel).
static const StepPlan* ensure_step_plans(dhw_handle* h, const std::vector<uint64_t>& key, int B, int L, int Lt, int T, int mode,
                                         const float* noise, const std::vector<float>& beta, const std::vector<float>& alpha) {
  if (!h->persist || h->nstreams != 1 || h->prec != PREC_BF16 || !h->fuse || !h->plane || !h->fuse_heads || !h->fuse_up || !h->chain) return nullptr;
  auto it = h->plans.find(key);
  if (it != h->plans.end()) return it->second.ok ? it->second.dev : nullptr;
  dhw_handle::StepPlans& sp = h->plans[key];
  const size_t need = step_sync_words(B);
  if (need > h->step_sync_words) {   // (earlier plans keep the smaller buffer: it is never freed before destroy)
    if (dev_alloc(h, (void**)&h->d_step_sync, need * sizeof(unsigned))) return nullptr;
    h->step_sync_words = need;
  }
  std::vector<StepPlan> host;
  if (sample_enqueue(h, &h->ws[0], 0, B, B, h->d_text_stage, h->d_style_stage, L, Lt, T, mode, noise, h->d_out_stage, nullptr, beta, alpha, nullptr, &host) ||
      (int)host.size() != T)
    return nullptr;
  if (getenv("DHW_PERSIST_TRACE") && atoi(getenv("DHW_PERSIST_TRACE"))) {
    if (!h->d_step_trace && dev_alloc(h, (void**)&h->d_step_trace, (size_t)h->persist_grid * STEP_MAX_PHASES * 4 * 8)) return nullptr;
    host.back().trace = h->d_step_trace;
  }
  if (dev_alloc(h, (void**)&sp.dev, host.size() * sizeof(StepPlan), false)) return nullptr;
  if (hipMemcpy(sp.dev, host.data(), host.size() * sizeof(StepPlan), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  sp.ok = true;
  return sp.dev;
}

// All sub-batches of one dhw_sample call.  On a capturing stream the sub-batches fork onto the handle's
// side streams (parallel graph branches) and join back; eagerly (profiling) they run one after another.
static int sample_enqueue_all(dhw_handle* h, bool fork, int B, const int64_t* text, const float* style, int L, int Lt,
                              int T, int mode, const float* noise, float* out, hipStream_t st,
                              const std::vector<float>& beta, const std::vector<float>& alpha, const StepPlan* d_plans = nullptr) {
  const int ns = std::min(h->nstreams, B);
  const int per = (B + ns - 1) / ns;
  taps_clear(h);
  if (!fork || ns == 1) {
    for (int s = 0, b0 = 0; b0 < B; ++s, b0 += per) {
      int rc = sample_enqueue(h, &h->ws[s], b0, std::min(per, B - b0), B, text, style, L, Lt, T, mode, noise, out, st, beta, alpha, ns == 1 ? d_plans : nullptr);
      if (rc) return rc;
    }
    return 0;
  }
  hipEvent_t fork_ev, join_ev[MAX_STREAMS] = {};
  if (hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming) != hipSuccess) return fail(h, DHW_ERR_HIP, "event create failed");
  int rc = 0;
  if (hipEventRecord(fork_ev, st) != hipSuccess) rc = fail(h, DHW_ERR_HIP, "fork record failed");
  for (int s = 1, b0 = per; !rc && b0 < B; ++s, b0 += per) {
    hipStream_t ss = h->sub_streams[s];
    if (hipStreamWaitEvent(ss, fork_ev, 0) != hipSuccess) { rc = fail(h, DHW_ERR_HIP, "fork wait failed"); break; }
    rc = sample_enqueue(h, &h->ws[s], b0, std::min(per, B - b0), B, text, style, L, Lt, T, mode, noise, out, ss, beta, alpha);
    if (rc) break;
    if (hipEventCreateWithFlags(&join_ev[s], hipEventDisableTiming) != hipSuccess || hipEventRecord(join_ev[s], ss) != hipSuccess)
      rc = fail(h, DHW_ERR_HIP, "join record failed");
  }
  if (!rc) rc = sample_enqueue(h, &h->ws[0], 0, std::min(per, B), B, text, style, L, Lt, T, mode, noise, out, st, beta, alpha);
  for (int s = 1; s < MAX_STREAMS; ++s)
    if (join_ev[s]) {
      if (!rc && hipStreamWaitEvent(st, join_ev[s], 0) != hipSuccess) rc = fail(h, DHW_ERR_HIP, "join wait failed");
      hipEventDestroy(join_ev[s]);
    }
  hipEventDestroy(fork_ev);
  return rc;
}

int dhw_sample(dhw_handle* h, const int64_t* text, const float* style, int B, int L, int Lt, int T, int mode,
               const float* noise, uint64_t seed, int64_t first_sample, float* out, void* hip_stream) {
  DHW_GUARD(h, "dhw_sample", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    if (!text || !style || !out) return fail(h, DHW_ERR_ARG, "dhw_sample: null pointer");
    if (T < 1 || (mode != 0 && mode != 1)) return fail(h, DHW_ERR_ARG, "dhw_sample: bad T/mode");
    int rc = check_shapes(h, B, L, Lt);
    if (rc) return rc;
    if ((rc = dhw_finalize(h))) return rc;
    HIPCK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (h->h_step_err && *(volatile unsigned*)h->h_step_err) {
      // a persistent step kernel gave up waiting (bounded spin, persist.h): its results were wrong; say so and fall back for good
      const unsigned code = *(volatile unsigned*)h->h_step_err;
      *(volatile unsigned*)h->h_step_err = 0;
      h->persist = false;
      hipDeviceSynchronize();
      for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
      h->graphs.clear();
      if (h->d_step_sync) hipMemset(h->d_step_sync, 0, h->step_sync_words * sizeof(unsigned));
      if (code >= 0x100u)
        return fail(h, DHW_ERR_HIP, "persistent step kernel: XCD %u owns samples but no workgroup of the launch ran there in an EARLIER call (partitioned / "
                    "CU-masked device?): those samples were never computed; persistent launches are now disabled for this handle", code - 0x100u);
      return fail(h, DHW_ERR_HIP, "persistent step kernel timed out waiting for phase %u in an EARLIER call (its samples were invalid); "
                  "persistent launches are now disabled for this handle", code - 1);
    }
    dhw_handle::FilmT* ft = nullptr;
    if ((rc = ensure_film_T(h, T, &ft))) return rc;
    h->d_film_T = ft->d_film;
    if (h->plane) {
      const int ns = std::min(h->nstreams, B), per = (B + ns - 1) / ns;
      for (int s = 0; s < ns; ++s)
        if ((rc = ensure_plane(h, h->ws[s], plane_chunk(T), per))) return rc;
    }
    std::vector<float> beta, alpha;
    schedule_host(T, beta, alpha);
    if (!ft->ready) {
      // once per (weights, T): sigma_i = sqrt(abar_i) -> sigma MLP -> FiLM table [T, 2*TOT]; uploaded on the caller's
      // stream from a buffer the handle owns, so it is ordered against everything else this call enqueues
      ft->h_sigma.resize(T);
      for (int i = 0; i < T; ++i) ft->h_sigma[i] = sqrtf(alpha[i]);   // inference.py:89
      HIPCK(h, hipMemcpyAsync(ft->d_sigma, ft->h_sigma.data(), T * 4, hipMemcpyHostToDevice, st));
      Ctx c{h, &h->ws[0], st, B, L, Lt, h->dims.S * 5, ft->d_film, 0};
      RUN_SMALL(c, "sigma_ffn", launch_sigma_ffn(ft->d_sigma, T, h->sg_w1, h->sg_b1, h->sg_w2, h->sg_b2, ft->d_sig32, st));
      RUN_SMALL(c, "film_table", launch_film(ft->d_sig32, T, h->d_film_w, h->d_film_b, 2 * h->film_tot, ft->d_film, st));
      if (c.err) return c.err;
      ft->ready = true;
    }
    {
      hipError_t e = launch_set_seed(h->d_seed, seed, first_sample, st);
      if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "set_seed: %s", hipGetErrorString(e));
    }

    // stage the caller's tensors into library-owned buffers (tiny D2D copies, outside the graph)
    const size_t rows = (size_t)B * L;
    HIPCK(h, hipMemcpyAsync(h->d_text_stage, text, (size_t)B * Lt * 8, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemcpyAsync(h->d_style_stage, style, (size_t)B * h->dims.S * 1280 * 4, hipMemcpyDeviceToDevice, st));
    const float* nz = nullptr;
    if (noise) {
      const size_t need = (size_t)(T + 1) * rows * 2;
      if (need > h->noise_stage_cap) {
        HIPCK(h, hipDeviceSynchronize());
        for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);   // they captured the old staging pointer
        h->graphs.clear();
        h->plans.clear();   // (so did the step plans)
        if ((rc = dev_alloc(h, (void**)&h->d_noise_stage, need * 4, false))) return rc;
        h->noise_stage_cap = need;
      }
      HIPCK(h, hipMemcpyAsync(h->d_noise_stage, noise, need * 4, hipMemcpyDeviceToDevice, st));
      nz = h->d_noise_stage;
    }

    const bool graph = h->use_graph && !h->prof && !h->teach_every;
    if (!graph) {
      // eager launches: sub-batches still fork onto the side streams (concurrent kernels of different sub-batches);
      // profiling keeps one stream so the per-launch events bracket one kernel each
      rc = sample_enqueue_all(h, !h->prof, B, h->d_text_stage, h->d_style_stage, L, Lt, T, mode, nz, h->d_out_stage, st, beta, alpha);
    } else {
      const std::vector<uint64_t> key = {(uint64_t)B, (uint64_t)L, (uint64_t)Lt, (uint64_t)T, (uint64_t)mode, (uint64_t)(nz != nullptr), (uint64_t)h->nstreams, (uint64_t)h->plane, (uint64_t)h->fuse_heads, (uint64_t)h->fuse_up, (uint64_t)h->chain, (uint64_t)h->persist};
      auto it = h->graphs.find(key);
      if (it == h->graphs.end()) {
        const StepPlan* d_plans = ensure_step_plans(h, key, B, L, Lt, T, mode, nz, beta, alpha);   // (before the capture: it uploads)
        hipStream_t cs;
        HIPCK(h, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { hipStreamDestroy(cs); return fail(h, DHW_ERR_HIP, "begin capture: %s", hipGetErrorString(e)); }
        rc = sample_enqueue_all(h, true, B, h->d_text_stage, h->d_style_stage, L, Lt, T, mode, nz, h->d_out_stage, cs, beta, alpha, d_plans);
        hipGraph_t g = nullptr;
        e = hipStreamEndCapture(cs, &g);
        if (rc == 0 && e != hipSuccess) rc = fail(h, DHW_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e));
        hipGraphExec_t ex = nullptr;
        if (rc == 0) {
          e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
          if (e != hipSuccess) rc = fail(h, DHW_ERR_HIP, "graph instantiate failed: %s", hipGetErrorString(e));
        }
        if (g) hipGraphDestroy(g);
        hipStreamDestroy(cs);
        if (rc) return rc;
        it = h->graphs.emplace(key, ex).first;
      }
      HIPCK(h, hipGraphLaunch(it->second, st));
    }
    if (rc == 0) HIPCK(h, hipMemcpyAsync(out, h->d_out_stage, rows * 3 * 4, hipMemcpyDeviceToDevice, st));
    h->last_B = B; h->last_L = L; h->last_Lt = Lt;
    return rc;
  });
}

int dhw_work(dhw_handle* h, int L, int Lt, double* flops_out, double* bytes_out) {
  DHW_GUARD(h, "dhw_work", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    const dhw_dims& d = h->ldims;   // the model's own widths: zero padding (pad_weights) is not algorithmic work
    const double c1 = d.c1, c2 = d.c2, c3 = d.c3, dt = 2 * c2, S5 = d.S * 5;
    auto cb = [](double L_, double ci, double co) { return 2 * L_ * (3 * ci * co + 1.5 * ci * co + 1.5 * co * co + co * co); };
    auto el = [&](double Lk, double dm, double heads) {
      double f = 2 * Lt * dt * dm + 2 * Lt * dm * dm * 2;            // text_dense, k1, v1
      f += 2 * Lk * dm * dm * 2 + 2 * Lk * dm * dm * 4;              // q1, dense1, qkv2, dense2
      f += 2 * Lk * dm * 2 * dm * 2;                                 // ffn
      f += 4 * Lk * Lt * dm + 4 * Lk * Lk * dm;                      // SDPA cross + self
      (void)heads;
      return f;
    };
    double f = 0;
    f += 2 * S5 * (STYLE_CH * 4 * c2 + 4 * c2 * dt) + 2 * Lt * dt * dt * 2 + 2 * S5 * dt * dt * 2 + 4 * Lt * S5 * dt + 2 * Lt * dt * 2 * dt * 2;
    f += 2 * L * 2 * c1;
    f += cb(L, c1, c1) + cb(L / 2, c1, c2) + cb(L / 4, c2, c3) + cb(L / 4, dt, c3) + cb(L / 2, c3, c2) + cb(L, c2, c1);
    f += el(L / 2, c2, 3) + el(L / 4, c3, 4) + d.num_layers * el(L / 8, dt, 6);
    f += 2 * (L / 8) * c3 * dt;
    f += 2 * 3 * ((L / 4) * c3 * dt + (L / 2) * c2 * c3 + L * c1 * c2);
    f += 2 * L * c1 * 3;
    // block-boundary activation bytes: every top-level block reads its inputs and writes its outputs once
    const double es = (double)h->es;
    double by = 0;
    by += L * 2 * 4 + L * 3 * 4;                                                  // strokes in, eps+pen out (fp32)
    by += es * (L * c1 * 2 + (L / 2) * (c1 + c2) + (L / 4) * (c2 + c3) + (L / 4) * (dt + c3) + (L / 2) * (c3 + c2) + L * (c2 + c1));   // ConvBlocks
    by += es * 2 * ((L / 2) * c2 + (L / 4) * c3 + d.num_layers * (L / 8) * dt);   // EncoderLayers
    by += es * ((L / 8) * (c3 + dt));                                             // att_dense
    by += es * ((L / 4) * (c3 + dt + dt) + (L / 2) * (c2 + c3 + c3) + L * (c1 + c2 + c2));   // skip convs + upsample add
    by += es * (S5 * STYLE_CH + Lt * dt * (2 + 2 * (2 + d.num_layers)));          // text/style encoder + per-layer text reads
    if (flops_out) *flops_out = f;
    if (bytes_out) *bytes_out = by;
    return 0;
  });
}

// ---------------------------------------------------------------- debug / measurement hooks
int64_t dhw_debug_read(dhw_handle* h, const char* name, float* host_dst, int64_t max_floats, int64_t shape_out[3]) {
  DHW_GUARD(h, "dhw_debug_read", int64_t, {
    if (!h || !name || !host_dst) return fail(h, DHW_ERR_ARG, "dhw_debug_read: null argument");
    const Tap* tp = nullptr;
    for (const TapSlot& sl : h->taps)
      if (sl.set && sl.name == name) tp = &sl.t;
    if (!tp) return fail(h, DHW_ERR_ARG, "no activation named %s", name);
    const Tap& t = *tp;
    const int64_t n = (int64_t)h->last_B * t.rows * t.cols;
    if (n > max_floats) return fail(h, DHW_ERR_ARG, "buffer too small for %s", name);
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(h, DHW_ERR_HIP, "sync failed: %s", hipGetErrorString(hipGetLastError()));
    if (shape_out) { shape_out[0] = h->last_B; shape_out[1] = t.rows; shape_out[2] = t.cols; }
    if (t.f32 || h->prec == PREC_F32) {
      if (hipMemcpy(host_dst, t.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DHW_ERR_HIP, "memcpy failed");
    } else {
      std::vector<uint16_t> tmp(n);
      if (hipMemcpy(tmp.data(), t.p, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DHW_ERR_HIP, "memcpy failed");
      for (int64_t i = 0; i < n; ++i) host_dst[i] = bf2f(tmp[i]);
    }
    return n;
  });
}

int dhw_debug_xcd_swizzle(int block_id, int nwg) { return xcd_swizzle(block_id, nwg); }

// Tests of the no-throw barrier itself (needs no device, handle may be null): raise a C++ exception INSIDE the guarded body of an
// entry point, exactly where a std::map::at / vector::resize / new of the host code would.  Must come back as DHW_ERR_INTERNAL.
int dhw_debug_raise(dhw_handle* h, int kind) {
  DHW_GUARD(h, "dhw_debug_raise", int, {
    if (kind == DHW_RAISE_OUT_OF_RANGE) {
      std::map<std::string, int> m;
      return m.at("a buffer that was never allocated");
    }
    if (kind == DHW_RAISE_BAD_ALLOC) throw std::bad_alloc();
    if (kind == DHW_RAISE_UNKNOWN) throw 42;
    return fail(h, DHW_ERR_ARG, "dhw_debug_raise: kind %d", kind);
  });
}

int dhw_debug_randn(dhw_handle* h, uint64_t seed, int64_t first_sample, int B, int L, int iter, float* host_dst) {
  DHW_GUARD(h, "dhw_debug_xcd_swizzle", int, {
    if (!h || !host_dst || B < 1 || L < 1 || iter < -1 || (long)B * L > (long)h->dims.max_B * h->dims.max_L)
      return fail(h, DHW_ERR_ARG, "dhw_debug_randn: bad argument");
    HIPCK(h, hipSetDevice(h->device));
    HIPCK(h, hipDeviceSynchronize());
    const long rows = (long)B * L;
    hipError_t e = launch_set_seed(h->d_seed, seed, first_sample, nullptr);
    if (e == hipSuccess) e = launch_randn_init(h->ws[0].d_xt, rows, L, h->d_seed, 0, nullptr, iter);
    if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "randn: %s", hipGetErrorString(e));
    HIPCK(h, hipMemcpy(host_dst, h->ws[0].d_xt, rows * 2 * 4, hipMemcpyDeviceToHost));
    return 0;
  });
}

// The self-attention stage of an EncoderLayer's second kernel on its own (bench.py, roofline.by_function.attention; north_star:
// "MFMA utilisation for attention against the chip's peak").  enc_bc_kernel of layer `layer` (0 = enc3, 1 = enc5, 2.. = the
// bottleneck layers) is launched `iters` times back to back WITH its attention stage and `iters` times with the stage skipped
// (EncLayerParams.dbg bit 0, csrc/enc_bc_core.h) on the buffers the LAST dhw_forward(B, L, Lt) left in the workspace; the
// difference of the two mean launch times (HIP events on the stream) is the time of QK^T + softmax + PV + K / V staging.
// The product kernel, unmodified: no stamps, no extra instantiation.  flops_out = 4 B Lk^2 d (QK^T and PV over all heads).
int dhw_debug_attention_time(dhw_handle* h, int layer, int iters, double* us_with, double* us_without, double* flops_out, void* hip_stream) {
  DHW_GUARD(h, "dhw_debug_attention_time", int, {
    if (!h || !us_with || !us_without || iters < 1 || layer < 0) return fail(h, DHW_ERR_ARG, "dhw_debug_attention_time: bad argument");
    if (!h->packed || !h->last_B || layer >= (int)h->el.size()) return fail(h, DHW_ERR_STATE, "dhw_debug_attention_time: run dhw_forward first (layer %d of %d)", layer, (int)h->el.size());
    const EncLayerW& w = h->el[layer];
    if (!h->fuse || h->prec != PREC_BF16 || !enclayer_supported(h->prec, w.d, w.heads)) return fail(h, DHW_ERR_STATE, "dhw_debug_attention_time: the fused bf16 EncoderLayer kernels are not in use on this handle");
    HIPCK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int B = h->last_B, L = h->last_L;
    const int Lk = (int)el_rows(L, layer);
    Ctx c{h, &h->ws[0], st, B, L, h->last_Lt, h->dims.S * 5, h->d_film, 2L * h->film_tot};
    const void* x = layer == 0 ? CBB(c, CB_ENC2, out) : layer == 1 ? CBB(c, CB_ENC4, out) : layer == 2 ? WS(c, att_dense) : ELB(c, layer - 1, out);
    EncLayerParams q = enc_params(c, layer, w, x, Lk, h->lpadX[layer < 2 ? layer : 2], h->d_text_stage, nullptr);
    // (the measured launches write the layer's `out` again: same inputs, same values; with the stage skipped, different ones —
    // the workspace is scratch between calls)
    if (c.err) return c.err;
    hipEvent_t e0, e1;
    HIPCK(h, hipEventCreate(&e0));
    HIPCK(h, hipEventCreate(&e1));
    double us[2] = {0, 0};
    int rc = 0;
    for (int mode = 0; mode < 2 && !rc; ++mode) {
      q.dbg = mode;   // 0: with the attention stage, 1: skipped
      for (int it = 0; it < 3 + iters && !rc; ++it) {
        if (it == 3 && hipEventRecord(e0, st) != hipSuccess) rc = fail(h, DHW_ERR_HIP, "event record failed");
        hipError_t e = launch_enclayer(h->prec, q, 1, st, nullptr);
        if (e != hipSuccess) rc = fail(h, DHW_ERR_HIP, "enc_bc launch: %s", hipGetErrorString(e));
      }
      float ms = 0.f;
      if (!rc && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess))
        rc = fail(h, DHW_ERR_HIP, "event timing failed: %s", hipGetErrorString(hipGetLastError()));
      us[mode] = (double)ms * 1e3 / iters;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (rc) return rc;
    *us_with = us[0];
    *us_without = us[1];
    if (flops_out) *flops_out = 4.0 * B * (double)Lk * Lk * w.d;
    return 0;
  });
}

int dhw_profile_enable(dhw_handle* h, int on) {
  DHW_GUARD(h, "dhw_profile_enable", int, {
    if (!h) return DHW_ERR_ARG;
    h->prof = on != 0;
    return 0;
  });
}
int dhw_profile_reset(dhw_handle* h) {
  DHW_GUARD(h, "dhw_profile_reset", int, {
    if (!h) return DHW_ERR_ARG;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (auto& r : h->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    h->prof_recs.clear();
    h->prof_agg.clear();
    return 0;
  });
}
int dhw_profile_count(dhw_handle* h) {
  DHW_GUARD(h, "dhw_profile_count", int, {
    if (!h) return DHW_ERR_ARG;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    h->prof_agg.assign(h->prof_labels.size(), ProfAgg{});
    for (size_t i = 0; i < h->prof_labels.size(); ++i) h->prof_agg[i].label = h->prof_labels[i];
    for (auto& r : h->prof_recs) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
      ProfAgg& a = h->prof_agg[r.label];
      a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.n += 1;
    }
    return (int)h->prof_agg.size();
  });
}
int dhw_profile_get(dhw_handle* h, int i, const char** label, double* total_ms, int64_t* launches, double* flops_sum,
                    double* bytes_sum) {
  DHW_GUARD(h, "dhw_profile_get", int, {
    if (!h || i < 0 || i >= (int)h->prof_agg.size()) return DHW_ERR_ARG;
    const ProfAgg& a = h->prof_agg[i];
    if (label) *label = a.label.c_str();
    if (total_ms) *total_ms = a.ms;
    if (launches) *launches = a.n;
    if (flops_sum) *flops_sum = a.flops;
    if (bytes_sum) *bytes_sum = a.bytes;
    return 0;
  });
}
int dhw_set_streams(dhw_handle* h, int n) {
  DHW_GUARD(h, "dhw_set_streams", int, {
    if (!h || n < 1) return DHW_ERR_ARG;
    h->nstreams = std::min(n, h->nstreams_alloc);
    return h->nstreams;
  });
}
// shapes of dhw_sample that run as one persistent launch per denoiser call (persist.h): cached plans that are in use
int dhw_debug_persist_plans(dhw_handle* h) {
  DHW_GUARD(h, "dhw_debug_persist_plans", int, {
    if (!h) return DHW_ERR_ARG;
    int n = 0;
    for (auto& kv : h->plans) n += kv.second.ok ? 1 : 0;
    return n;
  });
}
// diagnostics: the stamps of the last persistent step (see persist.hip, PTRACE) -> host_dst[workgroups * STEP_MAX_PHASES * 4]; returns
// the number of workgroups (0 = no trace buffer: DHW_PERSIST_TRACE was not set when the plans were built)
int dhw_debug_persist_trace(dhw_handle* h, unsigned long long* host_dst, int64_t max_words) {
  DHW_GUARD(h, "dhw_debug_persist_trace", int, {
    if (!h || !host_dst) return DHW_ERR_ARG;
    if (!h->d_step_trace) return 0;
    const int64_t n = (int64_t)h->persist_grid * STEP_MAX_PHASES * 4;
    if (max_words < n) return DHW_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_dst, h->d_step_trace, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return DHW_ERR_HIP;
    return h->persist_grid;
  });
}
int dhw_set_graph(dhw_handle* h, int on) {
  DHW_GUARD(h, "dhw_set_graph", int, {
    if (!h) return DHW_ERR_ARG;
    h->use_graph = on != 0;
    return 0;
  });
}

int dhw_debug_set_teacher(dhw_handle* h, const float* reset_dev, float* capture_dev, int every) {
  DHW_GUARD(h, "dhw_debug_set_teacher", int, {
    if (!h) return DHW_ERR_ARG;
    if (every < 0 || (every > 0 && (!reset_dev || !capture_dev))) return fail(h, DHW_ERR_ARG, "dhw_debug_set_teacher: bad arguments");
    h->teach_every = every;
    h->teach_reset = every ? reset_dev : nullptr;
    h->teach_capture = every ? capture_dev : nullptr;
    return 0;
  });
}

}  // extern "C"
