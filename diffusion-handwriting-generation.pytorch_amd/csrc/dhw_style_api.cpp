// dhw_style_api.cpp — C-ABI of the StyleExtractor front end (include/dhw_style.h): torchvision MobileNetV2 `features`
// with BatchNorm folded, NHWC activations with channels padded to the GEMM tile, the 1x1 convolutions on the generic MFMA
// GEMM kernel (gemm.hip) and the spatial kernels of style.hip.  Reference: text_style.py:11-59.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dhw_style.h"
#include "abi_guard.h"
#include "dhw_kernels.h"

namespace {

struct SKey { std::string key; std::vector<int64_t> shape; };

// torchvision.models.mobilenet_v2 inverted_residual_setting: expansion t, output channels c, repeats n, first stride s
struct Setting { int t, c, n, s; };
const Setting kSettings[] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2}, {6, 96, 3, 1}, {6, 160, 3, 2}, {6, 320, 1, 1}};
constexpr int kStem = 32, kLast = 1280, kBins = 14;
constexpr float kBnEps = 1e-5f;

struct BlockDesc { int idx, t, cin, hid, cout, stride; bool res; };

std::vector<BlockDesc> block_descs() {
  std::vector<BlockDesc> v;
  int cin = kStem, idx = 1;
  for (const Setting& s : kSettings)
    for (int i = 0; i < s.n; ++i) {
      const int stride = i == 0 ? s.s : 1;
      v.push_back({idx++, s.t, cin, cin * s.t, s.c, stride, stride == 1 && cin == s.c});
      cin = s.c;
    }
  return v;
}

void add_bn(std::vector<SKey>& k, const std::string& n, int c) {
  for (const char* s : {"weight", "bias", "running_mean", "running_var"}) k.push_back({n + "." + s, {c}});
}

std::vector<SKey> build_keys() {
  std::vector<SKey> k;
  k.push_back({"features.0.0.weight", {kStem, 3, 3, 3}});
  add_bn(k, "features.0.1", kStem);
  for (const BlockDesc& b : block_descs()) {
    const std::string p = "features." + std::to_string(b.idx) + ".conv.";
    int j = 0;
    if (b.t != 1) {
      k.push_back({p + "0.0.weight", {b.hid, b.cin, 1, 1}});
      add_bn(k, p + "0.1", b.hid);
      j = 1;
    }
    k.push_back({p + std::to_string(j) + ".0.weight", {b.hid, 1, 3, 3}});
    add_bn(k, p + std::to_string(j) + ".1", b.hid);
    k.push_back({p + std::to_string(j + 1) + ".weight", {b.cout, b.hid, 1, 1}});
    add_bn(k, p + std::to_string(j + 2), b.cout);
  }
  k.push_back({"features.18.0.weight", {kLast, 320, 1, 1}});
  add_bn(k, "features.18.1", kLast);
  return k;
}

// channel padding: what the generic GEMM's column tiles (64 / 96 / 128 / 192 / 256 / 384) divide
int padc(int c) { return c <= 64 ? 64 : (c == 96 ? 96 : ((c + 63) / 64) * 64); }

uint16_t f2bf(float f) {
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

ErrBuf g_err;

}  // namespace

struct dhw_style {
  int device = 0, prec = PREC_BF16, es = 2;
  ErrBuf err;
  bool lookup_fail = false;   // the packing code asked for a key build_keys does not declare
  std::vector<SKey> spec;
  std::map<std::string, int> key_index;
  std::vector<std::vector<float>> host_w;
  std::vector<char> loaded;
  bool packed = false;
  std::vector<void*> allocs;

  float *stem_w = nullptr, *stem_b = nullptr;
  struct Block {
    BlockDesc d;
    int cin_p, hid_p, cout_p;
    void* w_exp = nullptr; float* b_exp = nullptr;
    float *w_dw = nullptr, *b_dw = nullptr;
    void* w_proj = nullptr; float* b_proj = nullptr;
  };
  std::vector<Block> blocks;
  void* w_last = nullptr; float* b_last = nullptr;

  void* buf[3] = {nullptr, nullptr, nullptr};
  size_t buf_bytes = 0;
  const void* feat = nullptr;   // feature map of the last forward
  int fB = 0, fH = 0, fW = 0;
};

namespace {

int fail(dhw_style* h, int code, const char* fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  (h ? h->err : g_err).vsetf(fmt, ap);
  va_end(ap);
  return code;
}
// the body of every extern "C" entry point runs inside this: no exception leaves the library (abi_guard.h)
#define STYLE_GUARD(h, fn, R, ...) \
  return abi_guard<R>(fn, [&](const char* f_, const char* w_) { return fail((h), DHW_ERR_INTERNAL, "%s: internal error: %s", f_, w_); }, [&]() -> R __VA_ARGS__)

#define SHIP(h, call)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return fail(h, DHW_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));               \
  } while (0)

int dev_alloc(dhw_style* h, void** p, size_t bytes) {
  SHIP(h, hipMalloc(p, bytes ? bytes : 16));
  h->allocs.push_back(*p);
  SHIP(h, hipMemset(*p, 0, bytes ? bytes : 16));
  return 0;
}
int upload_f32(dhw_style* h, const std::vector<float>& v, float** out) {
  int rc = dev_alloc(h, (void**)out, v.size() * 4);
  if (rc) return rc;
  SHIP(h, hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return 0;
}
// row-major Wf[N][K] -> MFMA-fragment order [N/16][K/32][64 lanes][8] in the handle's element type (as dhw_api.cpp)
int upload_packed(dhw_style* h, const std::vector<float>& wf, int N, int K, void** out) {
  const size_t n = (size_t)N * K;
  std::vector<float> pk(n);
  size_t o = 0;
  for (int nt = 0; nt < N / 16; ++nt)
    for (int kc = 0; kc < K / 32; ++kc)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) pk[o++] = wf[(size_t)(nt * 16 + (l & 15)) * K + kc * 32 + 8 * (l >> 4) + j];
  int rc = dev_alloc(h, out, n * h->es);
  if (rc) return rc;
  if (h->prec == PREC_F32) {
    SHIP(h, hipMemcpy(*out, pk.data(), n * 4, hipMemcpyHostToDevice));
  } else {
    std::vector<uint16_t> b(n);
    for (size_t i = 0; i < n; ++i) b[i] = f2bf(pk[i]);
    SHIP(h, hipMemcpy(*out, b.data(), n * 2, hipMemcpyHostToDevice));
  }
  return 0;
}

// (finalize-time only; a miss is a programming error: recorded, finalize returns DHW_ERR_INTERNAL, nothing throws)
const std::vector<float>& W(dhw_style* h, const std::string& k) {
  static const std::vector<float> none(4096, 0.f);   // (long enough for every per-channel read of the packing loops)
  auto it = h->key_index.find(k);
  if (it == h->key_index.end()) {
    if (!h->lookup_fail) fail(h, DHW_ERR_INTERNAL, "internal: the packing code asked for an unknown weight '%s'", k.c_str());
    h->lookup_fail = true;
    return none;
  }
  return h->host_w[it->second];
}

// eval-mode BatchNorm2d as a per-channel affine: y = x * scale + shift
void bn_affine(dhw_style* h, const std::string& n, int c, std::vector<float>& scale, std::vector<float>& shift) {
  const auto &g = W(h, n + ".weight"), &b = W(h, n + ".bias"), &m = W(h, n + ".running_mean"), &v = W(h, n + ".running_var");
  scale.resize(c);
  shift.resize(c);
  for (int i = 0; i < c; ++i) {
    scale[i] = g[i] / std::sqrt(v[i] + kBnEps);
    shift[i] = b[i] - m[i] * scale[i];
  }
}

// 1x1 convolution [cout][cin] + BN -> padded row-major [cout_p][cin_p] + bias[cout_p]
int pack_pointwise(dhw_style* h, const std::string& wkey, const std::string& bn, int cout, int cin, int cout_p, int cin_p, void** w_out,
                   float** b_out) {
  std::vector<float> sc, sh;
  bn_affine(h, bn, cout, sc, sh);
  const auto& w = W(h, wkey);
  std::vector<float> wf((size_t)cout_p * cin_p, 0.f), bias(cout_p, 0.f);
  for (int o = 0; o < cout; ++o) {
    for (int i = 0; i < cin; ++i) wf[(size_t)o * cin_p + i] = w[(size_t)o * cin + i] * sc[o];
    bias[o] = sh[o];
  }
  int rc = upload_packed(h, wf, cout_p, cin_p, w_out);
  return rc ? rc : upload_f32(h, bias, b_out);
}

// depthwise 3x3 [c][1][3][3] + BN -> [9][c_p] + bias[c_p]
int pack_depthwise(dhw_style* h, const std::string& wkey, const std::string& bn, int c, int c_p, float** w_out, float** b_out) {
  std::vector<float> sc, sh;
  bn_affine(h, bn, c, sc, sh);
  const auto& w = W(h, wkey);
  std::vector<float> wf((size_t)9 * c_p, 0.f), bias(c_p, 0.f);
  for (int i = 0; i < c; ++i) {
    for (int t = 0; t < 9; ++t) wf[(size_t)t * c_p + i] = w[(size_t)i * 9 + t] * sc[i];
    bias[i] = sh[i];
  }
  int rc = upload_f32(h, wf, w_out);
  return rc ? rc : upload_f32(h, bias, b_out);
}

int run_pointwise(dhw_style* h, const void* in, int B, long rows, int cin_p, int cout_p, const void* w, const float* b, bool relu6,
                  const void* res, void* out, hipStream_t st) {
  GemmParams p{};
  p.nseg = 1;
  p.seg[0] = GemmSeg{in, w, cin_p, 1, 0};
  p.B = B;
  p.L = (int)rows;
  p.N = cout_p;
  p.n_store = cout_p;
  p.bias0 = b;
  p.film_div = 1;
  p.res1 = res;
  p.relu6_out = relu6 ? 1 : 0;
  p.out = out;
  hipError_t e = launch_gemm(h->prec, p, st);
  if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "pointwise conv %d -> %d: %s", cin_p, cout_p, hipGetErrorString(e));
  return 0;
}

int ensure_buffers(dhw_style* h, int B, int H, int W) {
  // the widest activation: the expanded 96-channel map at half resolution (features.2) = 24 H W elements per image
  const size_t H1 = (H + 1) / 2, W1 = (W + 1) / 2;
  size_t need = ((size_t)B * H1 * W1 + 64) * 96;
  need = std::max(need, ((size_t)B * ((H + 31) / 32 + 1) * ((W + 31) / 32 + 1) + 64) * (size_t)kLast);
  need *= h->es;
  if (need <= h->buf_bytes) return 0;
  SHIP(h, hipDeviceSynchronize());
  for (int i = 0; i < 3; ++i) {
    int rc = dev_alloc(h, &h->buf[i], need);   // (a grown buffer leaks the smaller one until destroy)
    if (rc) return rc;
  }
  h->buf_bytes = need;
  return 0;
}

}  // namespace

extern "C" {

const char* dhw_style_last_error(dhw_style* h) { return h ? h->err.c_str() : g_err.c_str(); }

int dhw_style_create(dhw_style** out, int precision, int device) {
  STYLE_GUARD(nullptr, "dhw_style_create", int, {
    if (!out) return fail(nullptr, DHW_ERR_ARG, "dhw_style_create: null out");
    *out = nullptr;
    if (precision != DHW_PREC_BF16 && precision != DHW_PREC_F32) return fail(nullptr, DHW_ERR_ARG, "dhw_style_create: bad precision %d", precision);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device || device < 0)
      return fail(nullptr, DHW_ERR_HIP, "no HIP device %d (the StyleExtractor has no CPU path)", device);
    struct Hold {
      dhw_style* p;
      ~Hold() { delete p; }
    } hold{new dhw_style};
    dhw_style* h = hold.p;
    h->device = device;
    h->prec = precision == DHW_PREC_F32 ? PREC_F32 : PREC_BF16;
    h->es = h->prec == PREC_F32 ? 4 : 2;
    h->spec = build_keys();
    for (size_t i = 0; i < h->spec.size(); ++i) h->key_index[h->spec[i].key] = (int)i;
    h->host_w.resize(h->spec.size());
    h->loaded.assign(h->spec.size(), 0);
    if (hipSetDevice(device) != hipSuccess || gemm_init() != hipSuccess)
      return fail(nullptr, DHW_ERR_HIP, "device setup failed: %s", hipGetErrorString(hipGetLastError()));
    hold.p = nullptr;
    *out = h;
    return 0;
  });
}

void dhw_style_destroy(dhw_style* h) {
  if (!h) return;
  try {
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    delete h;
  } catch (...) {
  }
}

int dhw_style_num_keys(dhw_style* h) { return h ? (int)h->spec.size() : DHW_ERR_ARG; }

int dhw_style_key_info(dhw_style* h, int i, const char** key, int64_t shape[4], int* ndim) {
  STYLE_GUARD(h, "dhw_style_key_info", int, {
    if (!h || i < 0 || i >= (int)h->spec.size()) return DHW_ERR_ARG;
    if (key) *key = h->spec[i].key.c_str();
    if (ndim) *ndim = (int)h->spec[i].shape.size();
    if (shape)
      for (size_t k = 0; k < h->spec[i].shape.size(); ++k) shape[k] = h->spec[i].shape[k];
    return 0;
  });
}

int dhw_style_load(dhw_style* h, const char* key, const void* host_ptr, int dtype, const int64_t* shape, int ndim) {
  STYLE_GUARD(h, "dhw_style_load", int, {
    if (!h || !key || !host_ptr || (!shape && ndim > 0)) return fail(h, DHW_ERR_ARG, "dhw_style_load: null argument");
    const std::string k = key;
    // the parts of torchvision's MobileNetV2 state_dict the feature extractor does not use
    if (k.rfind("classifier.", 0) == 0 || (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0)) return 0;
    auto it = h->key_index.find(k);
    if (it == h->key_index.end()) return fail(h, DHW_ERR_KEY, "unexpected key in MobileNetV2 state_dict: %s", key);
    const SKey& s = h->spec[it->second];
    bool ok = ndim == (int)s.shape.size();
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) { ok = shape[i] == s.shape[i]; n *= (size_t)s.shape[i]; }
    if (!ok) return fail(h, DHW_ERR_KEY, "size mismatch for %s", key);
    std::vector<float>& dst = h->host_w[it->second];
    dst.resize(n);
    switch (dtype) {
      case DHW_F32: std::memcpy(dst.data(), host_ptr, n * 4); break;
      case DHW_F64: for (size_t i = 0; i < n; ++i) dst[i] = (float)((const double*)host_ptr)[i]; break;
      case DHW_BF16: for (size_t i = 0; i < n; ++i) dst[i] = bf2f(((const uint16_t*)host_ptr)[i]); break;
      default: return fail(h, DHW_ERR_ARG, "dhw_style_load: unsupported dtype %d", dtype);
    }
    h->loaded[it->second] = 1;
    h->packed = false;
    return 0;
  });
}

int dhw_style_finalize(dhw_style* h) {
  STYLE_GUARD(h, "dhw_style_finalize", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    if (h->packed) return 0;
    for (size_t i = 0; i < h->spec.size(); ++i)
      if (!h->loaded[i]) return fail(h, DHW_ERR_KEY, "missing key in MobileNetV2 state_dict: %s", h->spec[i].key.c_str());
    SHIP(h, hipSetDevice(h->device));
    SHIP(h, hipDeviceSynchronize());
    int rc;
    {  // stem: the 3 input channels carry the same grey image -> sum the kernel over them
      std::vector<float> sc, sh;
      bn_affine(h, "features.0.1", kStem, sc, sh);
      const auto& w = W(h, "features.0.0.weight");
      const int cp = padc(kStem);
      std::vector<float> wf((size_t)9 * cp, 0.f), bias(cp, 0.f);
      for (int o = 0; o < kStem; ++o) {
        for (int t = 0; t < 9; ++t) {
          float s = 0.f;
          for (int ci = 0; ci < 3; ++ci) s += w[((size_t)o * 3 + ci) * 9 + t];
          wf[(size_t)t * cp + o] = s * sc[o];
        }
        bias[o] = sh[o];
      }
      if ((rc = upload_f32(h, wf, &h->stem_w)) || (rc = upload_f32(h, bias, &h->stem_b))) return rc;
    }
    h->blocks.clear();
    for (const BlockDesc& d : block_descs()) {
      dhw_style::Block b;
      b.d = d;
      b.cin_p = padc(d.cin);
      b.hid_p = padc(d.hid);
      b.cout_p = padc(d.cout);
      const std::string p = "features." + std::to_string(d.idx) + ".conv.";
      int j = 0;
      if (d.t != 1) {
        if ((rc = pack_pointwise(h, p + "0.0.weight", p + "0.1", d.hid, d.cin, b.hid_p, b.cin_p, &b.w_exp, &b.b_exp))) return rc;
        j = 1;
      }
      if ((rc = pack_depthwise(h, p + std::to_string(j) + ".0.weight", p + std::to_string(j) + ".1", d.hid, b.hid_p, &b.w_dw, &b.b_dw))) return rc;
      if ((rc = pack_pointwise(h, p + std::to_string(j + 1) + ".weight", p + std::to_string(j + 2), d.cout, d.hid, b.cout_p, b.hid_p, &b.w_proj, &b.b_proj))) return rc;
      h->blocks.push_back(b);
    }
    if ((rc = pack_pointwise(h, "features.18.0.weight", "features.18.1", kLast, 320, kLast, padc(320), &h->w_last, &h->b_last))) return rc;
    if (h->lookup_fail) return DHW_ERR_INTERNAL;
    h->packed = true;
    return 0;
  });
}

int dhw_style_forward(dhw_style* h, const float* img, int B, int H, int W, float* out, void* hip_stream) {
  STYLE_GUARD(h, "dhw_style_forward", int, {
    if (!h) return fail(nullptr, DHW_ERR_ARG, "null handle");
    if (!img || !out) return fail(h, DHW_ERR_ARG, "dhw_style_forward: null pointer");
    if (B < 1 || H < 96 || W < 96) return fail(h, DHW_ERR_ARG, "dhw_style_forward: needs B >= 1 and an image of at least 96 x 96 (got %d x %d x %d)", B, H, W);
    int rc = dhw_style_finalize(h);
    if (rc) return rc;
    SHIP(h, hipSetDevice(h->device));
    if ((rc = ensure_buffers(h, B, H, W))) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    int hh = (H + 1) / 2, ww = (W + 1) / 2, cur = 0;
    hipError_t e = launch_style_stem(h->prec, img, B, H, W, h->stem_w, h->stem_b, padc(kStem), h->buf[0], st);
    if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "stem: %s", hipGetErrorString(e));
    for (const dhw_style::Block& b : h->blocks) {
      const int a = (cur + 1) % 3, c = (cur + 2) % 3;
      const void* hidden = h->buf[cur];
      if (b.d.t != 1) {
        if ((rc = run_pointwise(h, h->buf[cur], B, (long)hh * ww, b.cin_p, b.hid_p, b.w_exp, b.b_exp, true, nullptr, h->buf[a], st))) return rc;
        hidden = h->buf[a];
      }
      const int ho = b.d.stride == 2 ? (hh + 1) / 2 : hh, wo = b.d.stride == 2 ? (ww + 1) / 2 : ww;
      e = launch_style_dw(h->prec, hidden, B, hh, ww, b.d.stride, b.w_dw, b.b_dw, b.hid_p, h->buf[c], st);
      if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "depthwise %d: %s", b.d.idx, hipGetErrorString(e));
      // projection (no activation) + the residual when the block keeps shape; written over the dead expanded map
      if ((rc = run_pointwise(h, h->buf[c], B, (long)ho * wo, b.hid_p, b.cout_p, b.w_proj, b.b_proj, false, b.d.res ? h->buf[cur] : nullptr, h->buf[a], st)))
        return rc;
      cur = a;
      hh = ho;
      ww = wo;
    }
    const int nxt = (cur + 1) % 3;
    if ((rc = run_pointwise(h, h->buf[cur], B, (long)hh * ww, padc(320), kLast, h->w_last, h->b_last, true, nullptr, h->buf[nxt], st))) return rc;
    h->feat = h->buf[nxt];
    h->fB = B; h->fH = hh; h->fW = ww;
    e = launch_style_pool(h->prec, h->buf[nxt], B, hh, ww, kLast, kBins, out, st);
    if (e != hipSuccess) return fail(h, DHW_ERR_HIP, "pool: %s", hipGetErrorString(e));
    return 0;
  });
}

int64_t dhw_style_debug_features(dhw_style* h, float* host_dst, int64_t max_floats, int64_t shape_out[4]) {
  STYLE_GUARD(h, "dhw_style_debug_features", int64_t, {
    if (!h || !host_dst || !h->feat) return fail(h, DHW_ERR_ARG, "dhw_style_debug_features: no forward yet");
    const int64_t n = (int64_t)h->fB * h->fH * h->fW * kLast;
    if (n > max_floats) return fail(h, DHW_ERR_ARG, "buffer too small");
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(h, DHW_ERR_HIP, "sync failed");
    if (shape_out) { shape_out[0] = h->fB; shape_out[1] = h->fH; shape_out[2] = h->fW; shape_out[3] = kLast; }
    if (h->prec == PREC_F32) {
      if (hipMemcpy(host_dst, h->feat, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DHW_ERR_HIP, "memcpy failed");
    } else {
      std::vector<uint16_t> tmp(n);
      if (hipMemcpy(tmp.data(), h->feat, n * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DHW_ERR_HIP, "memcpy failed");
      for (int64_t i = 0; i < n; ++i) host_dst[i] = bf2f(tmp[i]);
    }
    return n;
  });
}

}  // extern "C"
