// Model-level C ABI: the layer graphs of the FCOS detector, the A2J pose network and the HandNet glue in C++, on top of
// the op-level entry points of this library.  A host that is not Python binds these (hn_create /
// hn_load_weight / hn_fcos_forward / hn_a2j_forward / hn_destroy); the Python engines (hn_amd/*_engine.py) issue the
// same launches with the same descriptors, so both hosts produce bit-identical results
// (tests/test_model_abi_gpu.py).
//
//   hn_create        configuration -> handle
//   hn_load_weight   one entry of a REFERENCE-layout state_dict, fp32 host data, by name
//   hn_finalize      BatchNorm / FrozenBatchNorm folding in fp64 (fcos.py:737 body; a2j/resnet.py:35,67-71), NCHW -> [Cout][R][S][Cin]
//                    repacking, hi/lo fp16 splitting, upload (the only place weights are allocated)
//   hn_fcos_forward  fcos_utils/fcos.py:675-767 (eval)      hn_a2j_forward   a2j/a2j.py:243-250
//   hn_handnet_forward  handnet_pipeline/handnet_pipeline.py:58-116
// Precision per hn_model_config: f16x3 (default), f16x1 (f16_terms = 1) or the exact f32 MFMA (precision = HN_PRECISION_F32).
// Memory: activations live in one arena per model that is sized by a dry pass over the graph and (re)allocated only
// when a forward needs more than any earlier one -- steady-state calls neither allocate nor synchronise.
#include "hn_common.h"

#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

namespace {

constexpr double kBnEps = 1e-5;
constexpr int kCrop = 176;
constexpr int64_t kConvWorkspaceBytes = 32ll << 20;  // split-K workspace, as hn_amd/ops.py

struct HostT {
  std::vector<float> v;
  std::vector<int64_t> shape;
};

// packed convolution: w [cout][r][s][cin] fp32 (cin padded to 4), optional bias, optional split bank
struct ConvW {
  int cout = 0, r = 0, s = 0, cin = 0, stride = 1, pad = 0, dil = 1;
  std::vector<float> hw, hb;          // host copies (until upload)
  std::vector<_Float16> hw16;
  bool has_bias = false;
  float* w = nullptr;
  float* bias = nullptr;
  _Float16* w16 = nullptr;
};

struct T {  // device activation: dense NHWC fp32, or S32 split ([n][h][w][c/32][2][32] fp16); ps = pixel stride in elements
  char* p = nullptr;
  int n = 0, h = 0, w = 0, c = 0, ps = 0;
  bool split = false;
  size_t bytes = 0;   // size of the arena block this tensor OWNS (0 for channel slices / borrowed tensors)
};

// Activation arena with liveness: a tensor's block is handed back (give) once its last consumer has been
// ENQUEUED -- launches of a forward are ordered on one stream, so the next producer may overwrite it -- and take() reuses
// the smallest free block that fits before it bumps the high-water mark.  The dry pass replays exactly the same take / give
// sequence, so the layout it sizes is the layout the real pass gets.  (Bump-only, the FCOS graph at batch 32 needed ~20 GB.)
struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
  bool dry = false;
  bool overflow = false;   // a real pass asked for more than the planned bytes (run_planned fails the call)
  struct Block { size_t off, bytes; };
  std::vector<Block> free_blocks;
  char* origin() const { return dry ? (char*)0x1000 : base; }
  void reset() { off = 0; overflow = false; free_blocks.clear(); }
  char* take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < free_blocks.size(); ++i)
      if (free_blocks[i].bytes >= bytes && (best < 0 || free_blocks[i].bytes < free_blocks[best].bytes)) best = (int)i;
    if (best >= 0) {
      const Block b = free_blocks[best];
      if (b.bytes > bytes) free_blocks[best] = Block{b.off + bytes, b.bytes - bytes};
      else free_blocks.erase(free_blocks.begin() + best);
      return origin() + b.off;
    }
    const size_t a = (off + 255) & ~(size_t)255;
    if (!dry && a + bytes > cap) {   // never hand out memory beyond the arena: alias its start and report
      overflow = true;
      return base;
    }
    off = a + bytes;
    return origin() + a;
  }
  void give(char* p, size_t bytes) {
    if (!p || !bytes) return;
    free_blocks.push_back(Block{(size_t)(p - origin()), (bytes + 255) & ~(size_t)255});
  }
};

}  // namespace

struct hn_model {
  hn_model_config cfg;
  std::map<std::string, HostT> sd;
  bool finalized = false;
  std::vector<void*> owned;  // device allocations of the weights
  // ---- A2J ----
  ConvW a_stem;
  ConvW a_stem16;  // w16 = [64][7][2][32] stem rows (depth-only model: the three identical input channels folded into one), bias
  struct Bneck { ConvW c1, c2, c3, ds; bool has_ds = false; int layer = 0; };
  std::vector<Bneck> a_blocks;
  ConvW a_cls[4], a_reg[3], a_dep[3], a_regdep1, a_cls_out, a_reg_out, a_dep_out;
  // ---- FCOS ----
  ConvW f_stem16;  // w16 = [64][7][2][32] stem rows, bias
  ConvW f_stem;    // exact-f32 mode: [64][7][7][4] fp32 bank of conv1 + bn1
  struct Basic { ConvW c1, c2, ds; bool has_ds = false; int layer = 0; bool last = false; };
  std::vector<Basic> f_blocks;
  ConvW f_inner[3], f_layer[3], f_tower0, f_cls_t[3], f_reg_t[3], f_cls_out, f_reg_out, f_ext_out;  // f_ext_out: cfg.ext only
  float* f_gn0_gamma = nullptr; float* f_gn0_beta = nullptr;  // [512]
  float* f_gn_gamma[3] = {nullptr, nullptr, nullptr}; float* f_gn_beta[3] = {nullptr, nullptr, nullptr};  // [512] cls | reg
  // ---- run-time state ----
  Arena arena;
  std::map<std::string, size_t> plan;  // arena bytes per (entry, n, h, w)
  void* last_stream = nullptr;         // stream of the previous forward (the arena is per model, not per stream)
  bool has_last_stream = false;
  hipEvent_t done = nullptr;           // recorded at the end of every forward: a forward on ANOTHER stream waits for it
};

namespace {

#define HN_TRY(expr)            \
  do {                          \
    const int st_ = (expr);     \
    if (st_ != HN_OK) return st_; \
  } while (0)

// ------------------------------------------------------------------------------------------------------------------
// weight packing (host, fp64) -- restates hn_amd/weights.py
// ------------------------------------------------------------------------------------------------------------------
const HostT* find(const hn_model* m, const std::string& k) {
  auto it = m->sd.find(k);
  return it == m->sd.end() ? nullptr : &it->second;
}

int need(const hn_model* m, const std::string& k, const HostT** out, int ndim) {
  const HostT* t = find(m, k);
  if (!t) return hn::fail(HN_ERR_ARG, "weight '%s' was not loaded", k.c_str());
  if ((int)t->shape.size() != ndim) return hn::fail(HN_ERR_ARG, "weight '%s' has %d dims, expected %d", k.c_str(), (int)t->shape.size(), ndim);
  *out = t;
  return HN_OK;
}

// (Frozen)BatchNorm eval: y = x * scale + shift
int bn_scale_shift(const hn_model* m, const std::string& name, std::vector<double>& scale, std::vector<double>& shift) {
  const HostT *w, *b, *rm, *rv;
  HN_TRY(need(m, name + ".weight", &w, 1));
  HN_TRY(need(m, name + ".bias", &b, 1));
  HN_TRY(need(m, name + ".running_mean", &rm, 1));
  HN_TRY(need(m, name + ".running_var", &rv, 1));
  const size_t c = w->v.size();
  scale.resize(c);
  shift.resize(c);
  for (size_t i = 0; i < c; ++i) {
    scale[i] = (double)w->v[i] / sqrt((double)rv->v[i] + kBnEps);
    shift[i] = (double)b->v[i] - (double)rm->v[i] * scale[i];
  }
  return HN_OK;
}

// hi = fp16(v), lo = fp16(v - hi); bank [cout][(cin/32)*r*s][2][32], k tiles channel-block outer / tap inner
int split_bank(const std::vector<float>& w, int cout, int r, int s, int cin, std::vector<_Float16>& out, const char* what) {
  if (cin % 32) return hn::fail(HN_ERR_ARG, "%s: split bank needs cin %% 32 == 0", what);
  const int taps = r * s, cbs = cin / 32;
  out.resize((size_t)cout * cbs * taps * 64);
  for (int o = 0; o < cout; ++o)
    for (int cb = 0; cb < cbs; ++cb)
      for (int t = 0; t < taps; ++t) {
        const float* src = &w[((size_t)o * taps + t) * cin + cb * 32];
        _Float16* dst = &out[(((size_t)o * cbs + cb) * taps + t) * 64];
        for (int e = 0; e < 32; ++e) {
          const float v = src[e];
          if (!(fabsf(v) <= 65504.f))
            return hn::fail(HN_ERR_ARG, "%s: value %g outside the fp16 range (f16x3 range contract); this model needs the f32 mode", what, (double)v);
          const _Float16 h = (_Float16)v;
          dst[e] = h;
          dst[32 + e] = (_Float16)(v - (float)h);
        }
      }
  return HN_OK;
}

// weight [cout][cin][r][s] (torch) (+ bias) (+ BN) -> ConvW; sum_cin collapses the input channels (A2J stem, a2j/a2j.py:199)
int pack_conv(const hn_model* m, const std::string& wname, const std::string& bname, const std::string& bnname, int stride,
              int pad, int dil, bool sum_cin, ConvW& cw) {
  const HostT* w;
  HN_TRY(need(m, wname, &w, 4));
  const int cout = (int)w->shape[0], cin0 = (int)w->shape[1], r = (int)w->shape[2], s = (int)w->shape[3];
  const int cin = sum_cin ? 1 : cin0;
  const int cp = (cin + 3) / 4 * 4;
  std::vector<double> scale, shift;
  const bool bn = !bnname.empty();
  if (bn) HN_TRY(bn_scale_shift(m, bnname, scale, shift));
  const HostT* b = nullptr;
  if (!bname.empty()) HN_TRY(need(m, bname, &b, 1));
  cw.cout = cout; cw.r = r; cw.s = s; cw.cin = cp; cw.stride = stride; cw.pad = pad; cw.dil = dil;
  cw.hw.assign((size_t)cout * r * s * cp, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int y = 0; y < r; ++y)
      for (int x = 0; x < s; ++x)
        for (int c = 0; c < cin; ++c) {
          double v = 0.0;
          if (sum_cin) {
            for (int cc = 0; cc < cin0; ++cc) v += (double)w->v[(((size_t)o * cin0 + cc) * r + y) * s + x];
          } else {
            v = (double)w->v[(((size_t)o * cin0 + c) * r + y) * s + x];
          }
          if (bn) v *= scale[o];
          cw.hw[(((size_t)o * r + y) * s + x) * cp + c] = (float)v;
        }
  cw.has_bias = bn || b;
  if (cw.has_bias) {
    cw.hb.resize(cout);
    for (int o = 0; o < cout; ++o) {
      double v = b ? (double)b->v[o] : 0.0;
      if (bn) v = b ? v * scale[o] + shift[o] : shift[o];
      cw.hb[o] = (float)v;
    }
  }
  // (the exact-f32 mode keeps the fp32 bank: no split, no fp16-range condition on the folded weights)
  if (cp % 32 == 0 && m->cfg.precision != HN_PRECISION_F32) HN_TRY(split_bank(cw.hw, cout, r, s, cp, cw.hw16, wname.c_str()));
  return HN_OK;
}

int concat_cout(const ConvW& a, const ConvW& b, ConvW& out, const char* what, bool f32 = false) {
  if (a.r != b.r || a.s != b.s || a.cin != b.cin || a.stride != b.stride || a.pad != b.pad || a.dil != b.dil)
    return hn::fail(HN_ERR_ARG, "%s: stacked convolutions differ in geometry", what);
  out = a;
  out.cout = a.cout + b.cout;
  out.hw.insert(out.hw.end(), b.hw.begin(), b.hw.end());
  out.has_bias = a.has_bias || b.has_bias;
  if (out.has_bias) {
    out.hb.assign(out.cout, 0.f);
    for (int i = 0; i < a.cout && a.has_bias; ++i) out.hb[i] = a.hb[i];
    for (int i = 0; i < b.cout && b.has_bias; ++i) out.hb[a.cout + i] = b.hb[i];
  }
  out.hw16.clear();
  if (out.cin % 32 == 0 && !f32) HN_TRY(split_bank(out.hw, out.cout, out.r, out.s, out.cin, out.hw16, what));
  return HN_OK;
}

// R x R stem (cin <= 4) for hn_conv_stem_f16x3: each filter ROW is one 32-deep k tile, k = kx*4 + c
// sum_cin: collapse the input channels into one (the A2J stem sees the depth map replicated 3x, a2j/a2j.py:199)
int pack_stem_split(const hn_model* m, const std::string& wname, const std::string& bnname, ConvW& cw, bool sum_cin = false) {
  const HostT* w;
  HN_TRY(need(m, wname, &w, 4));
  const int cout = (int)w->shape[0], cin = (int)w->shape[1], r = (int)w->shape[2], s = (int)w->shape[3];
  if (r != s || r > 8 || cin > 4) return hn::fail(HN_ERR_ARG, "stem filter must be R x R with R <= 8 and Cin <= 4");
  std::vector<double> scale, shift;
  HN_TRY(bn_scale_shift(m, bnname, scale, shift));
  std::vector<float> rows((size_t)cout * r * 32, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int ky = 0; ky < r; ++ky)
      for (int kx = 0; kx < r; ++kx)
        if (sum_cin) {
          double acc = 0.0;
          for (int c = 0; c < cin; ++c) acc += (double)w->v[(((size_t)o * cin + c) * r + ky) * s + kx];
          rows[((size_t)o * r + ky) * 32 + kx * 4] = (float)(acc * scale[o]);
        } else {
          for (int c = 0; c < cin; ++c)
            rows[((size_t)o * r + ky) * 32 + kx * 4 + c] = (float)((double)w->v[(((size_t)o * cin + c) * r + ky) * s + kx] * scale[o]);
        }
  cw.cout = cout; cw.r = r; cw.s = r; cw.cin = 4; cw.stride = 2; cw.pad = r / 2; cw.dil = 1;
  cw.has_bias = true;
  cw.hb.resize(cout);
  for (int o = 0; o < cout; ++o) cw.hb[o] = (float)shift[o];
  return split_bank(rows, cout, r, 1, 32, cw.hw16, wname.c_str());
}

int upload(hn_model* m, ConvW& cw) {
  auto put = [&](const void* src, size_t bytes, void** dst) -> int {
    HN_CHECK_HIP(hipMalloc(dst, bytes));
    m->owned.push_back(*dst);
    HN_CHECK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return HN_OK;
  };
  // the fp32 bank is read by the f32 kernel only (the A2J stem, Cin = 4): convolutions with a split bank run on
  // hn_conv2d_nhwc_f16x3* and never touch it -- uploading both doubled the weight memory
  if (!cw.hw.empty() && cw.hw16.empty()) HN_TRY(put(cw.hw.data(), cw.hw.size() * 4, (void**)&cw.w));
  if (cw.has_bias) HN_TRY(put(cw.hb.data(), cw.hb.size() * 4, (void**)&cw.bias));
  if (!cw.hw16.empty()) HN_TRY(put(cw.hw16.data(), cw.hw16.size() * 2, (void**)&cw.w16));
  cw.hw.clear(); cw.hw.shrink_to_fit();
  cw.hw16.clear(); cw.hw16.shrink_to_fit();
  return HN_OK;
}

int upload_vec(hn_model* m, const std::vector<float>& v, float** dst) {
  HN_CHECK_HIP(hipMalloc((void**)dst, v.size() * 4));
  m->owned.push_back(*dst);
  HN_CHECK_HIP(hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// graph execution helpers (mirror hn_amd/ops.py: same descriptors -> same kernels -> same bits)
// ------------------------------------------------------------------------------------------------------------------
struct Ctx {
  hn_model* m;
  void* stream;
  bool dry;
  char* ws;  // split-K workspace
};

T alloc(Ctx& cx, int n, int h, int w, int c, bool split) {
  T t;
  t.n = n; t.h = h; t.w = w; t.c = c; t.split = split;
  t.ps = split ? 2 * c : c;
  t.bytes = (size_t)n * h * w * c * 4;  // S32 has the same bytes as fp32
  t.p = cx.m->arena.take(t.bytes);
  return t;
}

// the last consumer of `t` has been enqueued: its block may be reused (no-op for slices / borrowed tensors)
void release(Ctx& cx, T& t) {
  cx.m->arena.give(t.p, t.bytes);
  t.bytes = 0;
}

char* alloc_bytes(Ctx& cx, size_t bytes) { return cx.m->arena.take(bytes); }

inline void out_size(int h, int w, int r, int s, int stride, int pad, int dil, int& oh, int& ow) {
  oh = (h + 2 * pad - dil * (r - 1) - 1) / stride + 1;
  ow = (w + 2 * pad - dil * (s - 1) - 1) / stride + 1;
}

// hn_model_config.f16_terms of the model whose forward is being enqueued on this thread (set by run_planned): every
// descriptor of the graph carries it (0 = f16x3; 1 = the f16x1 throughput mode)
thread_local int t_terms = 0;

// hn_handnet_forward_xyz: the conversion the aggregation's epilogue performs for the forward being enqueued on this thread
// (null: plain hn_a2j_aggregate_f32)
struct ConvertReq {
  const int64_t* crop_box;
  const float* paras;
  const hn_convert_opts* opts;
  float* image_uvd;
  float* xyz_mm;
};
thread_local const ConvertReq* t_convert = nullptr;

int aggregate(Ctx& cx, const T& cls, const T& reg, const T& dep, const int32_t* valid, int k, float* keypoints) {
  hn_model* m = cx.m;
  if (t_convert)
    return hn_a2j_aggregate_convert_f32((const float*)cls.p, (const float*)reg.p, (const float*)dep.p, valid, k, cls.h, cls.w,
                                        m->cfg.num_joints, 16, t_convert->crop_box, (float)kCrop, (float)kCrop, t_convert->paras,
                                        t_convert->opts, keypoints, t_convert->image_uvd, t_convert->xyz_mm, cx.stream);
  return hn_a2j_aggregate_f32((const float*)cls.p, (const float*)reg.p, (const float*)dep.p, valid, k, cls.h, cls.w,
                              m->cfg.num_joints, 16, keypoints, cx.stream);
}

hn_conv_desc make_desc(const T& x, const ConvW& cw, int relu_cols) {
  hn_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.terms = t_terms;
  d.n = x.n; d.h = x.h; d.w = x.w; d.cin = cw.cin; d.cout = cw.cout; d.r = cw.r; d.s = cw.s;
  d.stride = cw.stride; d.pad = cw.pad; d.dil = cw.dil;
  int oh, ow;
  out_size(x.h, x.w, cw.r, cw.s, cw.stride, cw.pad, cw.dil, oh, ow);
  d.oh = oh; d.ow = ow;
  d.relu_cols = relu_cols;
  d.in_pix_stride = x.ps == (x.split ? 2 * cw.cin : cw.cin) ? 0 : x.ps;
  return d;
}

// one f16x3 convolution with the split-K workspace (ops.conv2d_nhwc with w16): S32 in, S32 or fp32 out
int conv16(Ctx& cx, const T& x, const ConvW& cw, bool relu, bool out_split, const T* res, bool res_up, T& y) {
  if (!x.split || x.c != cw.cin) return hn::fail(HN_ERR_ARG, "model graph: conv input mismatch (c %d vs cin %d)", x.c, cw.cin);
  hn_conv_desc d = make_desc(x, cw, relu ? cw.cout : 0);
  y = alloc(cx, x.n, d.oh, d.ow, cw.cout, out_split);
  d.out_split = out_split ? 1 : 0;
  if (res) {
    d.res_mode = res_up ? 2 : 1;
    if (res_up) { d.res_h = res->h; d.res_w = res->w; }
    d.res_split = res->split ? 1 : 0;
    d.res_pix_stride = res->ps;
  }
  d.splitk = 1;
  if (cx.dry) return HN_OK;
  return hn_conv2d_nhwc_f16x3_ws(&d, x.p, cw.w16, cw.bias, res ? res->p : nullptr, y.p, cx.ws, kConvWorkspaceBytes, cx.stream);
}

// channel-block slice of an S32 tensor (no copy)
T slice_blocks(const T& x, int b0, int b1) {
  T s = x;
  s.p = x.p + (size_t)b0 * 64 * 2;
  s.c = (b1 - b0) * 32;
  s.bytes = 0;
  return s;
}

// ops.conv2d_nhwc_multi: independent convolutions of different shapes as ONE launch (hn_conv2d_nhwc_f16x3_multi); every member
// computes what its own conv16 call would, bit for bit
struct MultiItem {
  T x;
  const ConvW* w = nullptr;
  bool relu = false, out_split = true;
  const T* res = nullptr;
  T y;   // out
};

int conv_multi(Ctx& cx, MultiItem* it, int count) {
  if (count < 1 || count > HN_CONV_MULTI_MAX) return hn::fail(HN_ERR_ARG, "model graph: multi launch of %d members", count);
  hn_conv_multi mm;
  memset(&mm, 0, sizeof(mm));
  mm.count = count;
  for (int i = 0; i < count; ++i) {
    const ConvW& cw = *it[i].w;
    if (!it[i].x.split || it[i].x.c != cw.cin) return hn::fail(HN_ERR_ARG, "model graph: multi member %d input mismatch", i);
    hn_conv_desc d = make_desc(it[i].x, cw, it[i].relu ? cw.cout : 0);
    it[i].y = alloc(cx, it[i].x.n, d.oh, d.ow, cw.cout, it[i].out_split);
    d.out_split = it[i].out_split ? 1 : 0;
    if (it[i].res) {
      d.res_mode = 1;
      d.res_split = it[i].res->split ? 1 : 0;
      d.res_pix_stride = it[i].res->ps;
    }
    d.splitk = 1;
    mm.desc[i] = d;
    mm.x16[i] = it[i].x.p; mm.w16[i] = cw.w16; mm.bias[i] = cw.bias;
    mm.residual[i] = it[i].res ? it[i].res->p : nullptr;
    mm.y[i] = it[i].y.p;
  }
  if (cx.dry) return HN_OK;
  return hn_conv2d_nhwc_f16x3_multi(&mm, cx.ws, kConvWorkspaceBytes, cx.stream);
}

struct GroupSpec {
  int count = 0;
  T x[HN_CONV_MAX_GROUP];
  const ConvW* w[HN_CONV_MAX_GROUP];
  char* y[HN_CONV_MAX_GROUP];           // output base (already offset to the member's channel)
  float* gn[HN_CONV_MAX_GROUP];
};

// ops.conv2d_nhwc_grouped: members share channels / filter / batch / pixel strides; outputs given by the caller
int conv_grouped(Ctx& cx, const GroupSpec& g, int relu_cols, bool out_split, int out_pix_stride, int gn_units) {
  const ConvW& c0 = *g.w[0];
  hn_conv_desc d = make_desc(g.x[0], c0, relu_cols);
  d.out_split = out_split ? 1 : 0;
  d.out_pix_stride = out_pix_stride == (out_split ? 2 : 1) * c0.cout ? 0 : out_pix_stride;
  d.splitk = -1;
  hn_conv_group grp;
  memset(&grp, 0, sizeof(grp));
  grp.count = g.count;
  grp.gn_units = gn_units;
  for (int i = 0; i < g.count; ++i) {
    grp.x16[i] = g.x[i].p; grp.w16[i] = g.w[i]->w16; grp.bias[i] = g.w[i]->bias; grp.y[i] = g.y[i];
    grp.gn_partial[i] = g.gn[i];
    grp.h[i] = g.x[i].h; grp.w[i] = g.x[i].w;
  }
  if (cx.dry) return HN_OK;
  return hn_conv2d_nhwc_f16x3_grouped(&d, &grp, cx.stream);
}

// ops.conv3x3_thin_levels: one <= 16-channel 3x3 filter bank on every level (the FCOS head outputs), dense fp32 outputs
int conv_thin_levels(Ctx& cx, const GroupSpec& g, int relu_cols) {
  const ConvW& c0 = *g.w[0];
  if (hn::env_flags().no_thin || t_terms == 1 || c0.cout > 16 || g.count > HN_FCOS_MAX_LEVELS) return conv_grouped(cx, g, relu_cols, false, c0.cout, 0);
  hn_thin_levels lv;
  memset(&lv, 0, sizeof(lv));
  lv.count = g.count;
  for (int i = 0; i < g.count; ++i) {
    lv.x16[i] = g.x[i].p; lv.y[i] = (float*)g.y[i]; lv.h[i] = g.x[i].h; lv.w[i] = g.x[i].w;
  }
  if (cx.dry) return HN_OK;
  const int dense = 2 * g.x[0].c;
  return hn_conv3x3_thin_f16x3_levels(&lv, g.x[0].n, g.x[0].c, c0.cout, c0.w16, c0.bias, relu_cols, g.x[0].ps == dense ? 0 : g.x[0].ps,
                                      cx.stream);
}

// ops.conv3x3_thin_levels_group: up to three of them as one launch where they run the tap kernel
int conv_thin_levels_group(Ctx& cx, const GroupSpec* const* gs, const int* relu_cols, int count) {
  bool thin = !hn::env_flags().no_thin && t_terms != 1 && gs[0]->count <= HN_FCOS_MAX_LEVELS;
  for (int m = 0; m < count; ++m) thin = thin && gs[m]->w[0]->cout <= 16;
  if (!thin) {
    for (int m = 0; m < count; ++m) HN_TRY(conv_thin_levels(cx, *gs[m], relu_cols[m]));
    return HN_OK;
  }
  hn_thin_member mem[3];
  memset(mem, 0, sizeof(mem));
  for (int m = 0; m < count; ++m) {
    const GroupSpec& g = *gs[m];
    mem[m].lv.count = g.count;
    for (int i = 0; i < g.count; ++i) {
      mem[m].lv.x16[i] = g.x[i].p; mem[m].lv.y[i] = (float*)g.y[i]; mem[m].lv.h[i] = g.x[i].h; mem[m].lv.w[i] = g.x[i].w;
    }
    mem[m].cout = g.w[0]->cout; mem[m].relu_cols = relu_cols[m]; mem[m].w16 = g.w[0]->w16; mem[m].bias = g.w[0]->bias;
  }
  if (cx.dry) return HN_OK;
  const T& x0 = gs[0]->x[0];
  return hn_conv3x3_thin_f16x3_levels_group(mem, count, x0.n, x0.c, x0.ps == 2 * x0.c ? 0 : x0.ps, cx.stream);
}

// one exact-f32 convolution (ops.conv2d_nhwc without w16 -> hn_conv2d_nhwc_f32): fp32 NHWC in and out; in_scale / in_shift =
// the GroupNorm of the PREVIOUS layer applied (with its ReLU) while the input is staged
int conv32(Ctx& cx, const T& x, const ConvW& cw, int relu_cols, const T* res, bool res_up, const float* in_scale,
           const float* in_shift, int affine_stride, T& y) {
  if (x.split || x.c != cw.cin) return hn::fail(HN_ERR_ARG, "model graph (f32): conv input mismatch (c %d vs cin %d)", x.c, cw.cin);
  hn_conv_desc d = make_desc(x, cw, relu_cols);
  d.terms = 0;
  y = alloc(cx, x.n, d.oh, d.ow, cw.cout, false);
  if (res) {
    d.res_mode = res_up ? 2 : 1;
    if (res_up) { d.res_h = res->h; d.res_w = res->w; }
  }
  if (in_scale) {
    d.in_affine = 1;
    d.in_affine_stride = affine_stride == cw.cin ? 0 : affine_stride;
  }
  if (cx.dry) return HN_OK;
  return hn_conv2d_nhwc_f32(&d, (const float*)x.p, cw.w, cw.bias, res ? (const float*)res->p : nullptr, in_scale, in_shift,
                            (float*)y.p, cx.stream);
}

// channel slice of a dense fp32 tensor (no copy)
T slice_channels(const T& x, int c0, int c1) {
  T s = x;
  s.p = x.p + (size_t)c0 * 4;
  s.c = c1 - c0;
  s.bytes = 0;
  return s;
}

int maxpool32(Ctx& cx, const T& x, T& y) {
  y = alloc(cx, x.n, (x.h + 2 - 3) / 2 + 1, (x.w + 2 - 3) / 2 + 1, x.c, false);
  if (cx.dry) return HN_OK;
  return hn_maxpool3x3s2_nhwc_f32((const float*)x.p, (float*)y.p, x.n, x.h, x.w, x.c, y.h, y.w, cx.stream);
}

// A2J in the exact-f32 mode (A2JEngine(precision="f32"): every convolution on the f32-MFMA kernel, heads ungrouped)
int a2j_graph_f32(Ctx& cx, const T& crops /* fp32 [k][176][176][4] */, const int32_t* valid, float* keypoints) {
  hn_model* m = cx.m;
  const int k = crops.n;
  T s0, x;
  HN_TRY(conv32(cx, crops, m->a_stem, m->a_stem.cout, nullptr, false, nullptr, nullptr, 0, s0));
  HN_TRY(maxpool32(cx, s0, x));
  release(cx, s0);
  T x3;
  for (size_t i = 0; i < m->a_blocks.size(); ++i) {
    auto& b = m->a_blocks[i];
    T o, o2, idn, y;
    HN_TRY(conv32(cx, x, b.c1, b.c1.cout, nullptr, false, nullptr, nullptr, 0, o));
    HN_TRY(conv32(cx, o, b.c2, b.c2.cout, nullptr, false, nullptr, nullptr, 0, o2));
    if (b.has_ds) HN_TRY(conv32(cx, x, b.ds, 0, nullptr, false, nullptr, nullptr, 0, idn));
    else idn = x;
    HN_TRY(conv32(cx, o2, b.c3, b.c3.cout, &idn, false, nullptr, nullptr, 0, y));
    release(cx, o);
    release(cx, o2);
    if (b.has_ds) release(cx, idn);
    if (x.p != x3.p) release(cx, x);
    x = y;
    const bool last_of_layer = i + 1 == m->a_blocks.size() || m->a_blocks[i + 1].layer != b.layer;
    if (b.layer == 3 && last_of_layer) x3 = x;
  }
  const T x4 = x;
  T c, rd;
  HN_TRY(conv32(cx, x3, m->a_cls[0], 256, nullptr, false, nullptr, nullptr, 0, c));
  HN_TRY(conv32(cx, x4, m->a_regdep1, 512, nullptr, false, nullptr, nullptr, 0, rd));
  T r = slice_channels(rd, 0, 256), dd = slice_channels(rd, 256, 512);
  for (int i = 1; i <= 3; ++i) {
    T t2;
    HN_TRY(conv32(cx, c, m->a_cls[i], 256, nullptr, false, nullptr, nullptr, 0, t2));
    release(cx, c);
    c = t2;
  }
  T cls, reg, dep;
  HN_TRY(conv32(cx, c, m->a_cls_out, 0, nullptr, false, nullptr, nullptr, 0, cls));
  for (int i = 0; i < 3; ++i) {
    T t2;
    HN_TRY(conv32(cx, r, m->a_reg[i], 256, nullptr, false, nullptr, nullptr, 0, t2));
    release(cx, r);
    r = t2;
  }
  for (int i = 0; i < 3; ++i) {
    T t2;
    HN_TRY(conv32(cx, dd, m->a_dep[i], 256, nullptr, false, nullptr, nullptr, 0, t2));
    release(cx, dd);
    dd = t2;
  }
  HN_TRY(conv32(cx, r, m->a_reg_out, 0, nullptr, false, nullptr, nullptr, 0, reg));
  HN_TRY(conv32(cx, dd, m->a_dep_out, 0, nullptr, false, nullptr, nullptr, 0, dep));
  if (cx.dry) return HN_OK;
  return aggregate(cx, cls, reg, dep, valid, k, keypoints);
}

// ------------------------------------------------------------------------------------------------------------------
// A2J (hn_amd/a2j_engine.py)
// ------------------------------------------------------------------------------------------------------------------
// The A2J layers behind the stem for SMALL batches (A2JEngine._trunk_multi + heads, k <= A2JEngine.MULTI_MAX_CROPS): launch-bound,
// so independent convolutions share launches -- the downsample beside conv1 of a block, one stage of the classification head
// (which reads x3 only) in each of layer4's launches, the regression tower beside the depth tower.
constexpr int kA2jMultiMaxCrops = 4;
int a2j_tail_small(Ctx& cx, T x, const int32_t* valid, float* keypoints) {
  hn_model* m = cx.m;
  const int k = x.n;
  struct Stage { const ConvW* w; bool relu, out_split; };
  const Stage st[5] = {{&m->a_cls[0], true, true}, {&m->a_cls[1], true, true}, {&m->a_cls[2], true, true},
                       {&m->a_cls[3], true, true}, {&m->a_cls_out, false, false}};
  auto launch = [&](MultiItem* it, int n) -> int {
    if (n > 1) return conv_multi(cx, it, n);
    return conv16(cx, it[0].x, *it[0].w, it[0].relu, it[0].out_split, it[0].res, false, it[0].y);
  };
  T c, x3;
  int stage = 0;
  bool have_c = false;
  for (auto& b : m->a_blocks) {
    const bool in_l4 = b.layer == 4;
    if (in_l4 && !have_c) { c = x3 = x; have_c = true; }
    auto ride = [&](MultiItem* it, int& n) -> bool {
      if (!in_l4 || stage >= 5) return false;
      it[n].x = c; it[n].w = st[stage].w; it[n].relu = st[stage].relu; it[n].out_split = st[stage].out_split;
      ++n;
      return true;
    };
    MultiItem it[3];
    int n = 0;
    it[n].x = x; it[n].w = &b.c1; it[n].relu = true; ++n;
    if (b.has_ds) { it[n].x = x; it[n].w = &b.ds; it[n].relu = false; ++n; }
    bool rides = ride(it, n);
    HN_TRY(launch(it, n));
    T o = it[0].y;
    const T idn = b.has_ds ? it[1].y : x;
    if (rides) { c = it[n - 1].y; ++stage; }
    MultiItem i2[2];
    n = 0;
    i2[n].x = o; i2[n].w = &b.c2; i2[n].relu = true; ++n;
    rides = ride(i2, n);
    HN_TRY(launch(i2, n));
    o = i2[0].y;
    if (rides) { c = i2[n - 1].y; ++stage; }
    MultiItem i3[2];
    n = 0;
    i3[n].x = o; i3[n].w = &b.c3; i3[n].relu = true; i3[n].res = &idn; ++n;
    rides = ride(i3, n);
    HN_TRY(launch(i3, n));
    x = i3[0].y;
    if (rides) { c = i3[n - 1].y; ++stage; }
  }
  while (stage < 5) {
    T y;
    HN_TRY(conv16(cx, c, *st[stage].w, st[stage].relu, st[stage].out_split, nullptr, false, y));
    c = y;
    ++stage;
  }
  const T cls = c;
  T rd;
  HN_TRY(conv16(cx, x, m->a_regdep1, true, true, nullptr, false, rd));
  T r = slice_blocks(rd, 0, 8), dd = slice_blocks(rd, 8, 16);
  for (int i = 0; i < 3; ++i) {
    MultiItem it[2];
    it[0].x = r; it[0].w = &m->a_reg[i]; it[0].relu = true;
    it[1].x = dd; it[1].w = &m->a_dep[i]; it[1].relu = true;
    HN_TRY(conv_multi(cx, it, 2));
    r = it[0].y; dd = it[1].y;
  }
  MultiItem io[2];
  io[0].x = r; io[0].w = &m->a_reg_out; io[0].relu = false; io[0].out_split = false;
  io[1].x = dd; io[1].w = &m->a_dep_out; io[1].relu = false; io[1].out_split = false;
  HN_TRY(conv_multi(cx, io, 2));
  if (cx.dry) return HN_OK;
  return aggregate(cx, cls, io[0].y, io[1].y, valid, k, keypoints);
}

// valid_rw: the per-crop flags when they are the model's own to update (hn_handnet_forward's has_hand: a crop with non-finite
// pixels becomes 2 = NaN keypoints, like the Python engine); null for a caller's read-only flags
int a2j_graph(Ctx& cx, const T& crops /* fp32 [k][176][176][4] */, const int32_t* valid, float* keypoints,
              int32_t* valid_rw = nullptr) {
  hn_model* m = cx.m;
  if (m->cfg.precision == HN_PRECISION_F32) return a2j_graph_f32(cx, crops, valid, keypoints);
  const int k = crops.n;
  // stem: conv1 + bn1 + relu + maxpool as ONE split-precision kernel on the crops' stem image (like hn_amd/a2j_engine.py)
  const int border = 3;
  char* img16 = alloc_bytes(cx, (size_t)2 * k * (crops.h + 2 * border) * (crops.w + 2 * border) * 4 * 2);
  if (!cx.dry) HN_TRY(hn_stem_image_nhwc4_valid((const float*)crops.p, k, crops.h, crops.w, border, img16, valid_rw, cx.stream));
  int sh, sw;
  out_size(crops.h + 2 * border, crops.w + 2 * border, 7, 7, 2, 0, 1, sh, sw);
  T x = alloc(cx, k, (sh + 2 - 3) / 2 + 1, (sw + 2 - 3) / 2 + 1, 64, true);
  if (!cx.dry)
    HN_TRY(hn_conv_stem_pool_f16x3_terms(img16, k, crops.h, crops.w, border, 7, 2, 64, m->a_stem16.w16, m->a_stem16.bias, x.p, t_terms,
                                         cx.stream));
  if (k <= kA2jMultiMaxCrops) return a2j_tail_small(cx, x, valid, keypoints);
  T x3;
  for (size_t i = 0; i < m->a_blocks.size(); ++i) {
    auto& b = m->a_blocks[i];
    T o, o2, idn, y;
    HN_TRY(conv16(cx, x, b.c1, true, true, nullptr, false, o));
    HN_TRY(conv16(cx, o, b.c2, true, true, nullptr, false, o2));
    if (b.has_ds) HN_TRY(conv16(cx, x, b.ds, false, true, nullptr, false, idn));
    else idn = x;
    HN_TRY(conv16(cx, o2, b.c3, true, true, &idn, false, y));
    release(cx, o);
    release(cx, o2);
    if (b.has_ds) release(cx, idn);
    if (x.p != x3.p) release(cx, x);   // the block input is dead (x3 is read again by the classification head)
    x = y;
    const bool last_of_layer = i + 1 == m->a_blocks.size() || m->a_blocks[i + 1].layer != b.layer;
    if (b.layer == 3 && last_of_layer) x3 = x;
  }
  const T x4 = x;
  // heads (grouped launches, a2j_engine.heads)
  T c, rd;
  HN_TRY(conv16(cx, x3, m->a_cls[0], true, true, nullptr, false, c));
  HN_TRY(conv16(cx, x4, m->a_regdep1, true, true, nullptr, false, rd));
  T r = slice_blocks(rd, 0, 8), dd = slice_blocks(rd, 8, 16);
  {
    T c2;
    HN_TRY(conv16(cx, c, m->a_cls[1], true, true, nullptr, false, c2));
    c = c2;
    GroupSpec g;
    g.count = 2;
    T r2 = alloc(cx, k, r.h, r.w, 256, true), d2 = alloc(cx, k, r.h, r.w, 256, true);
    g.x[0] = r; g.x[1] = dd; g.w[0] = &m->a_reg[0]; g.w[1] = &m->a_dep[0];
    g.y[0] = r2.p; g.y[1] = d2.p; g.gn[0] = g.gn[1] = nullptr;
    HN_TRY(conv_grouped(cx, g, 256, true, 512, 0));
    r = r2; dd = d2;
  }
  for (int i = 1; i <= 2; ++i) {
    GroupSpec g;
    g.count = 3;
    T c2 = alloc(cx, k, c.h, c.w, 256, true), r2 = alloc(cx, k, c.h, c.w, 256, true), d2 = alloc(cx, k, c.h, c.w, 256, true);
    g.x[0] = c; g.x[1] = r; g.x[2] = dd;
    g.w[0] = &m->a_cls[i + 1]; g.w[1] = &m->a_reg[i]; g.w[2] = &m->a_dep[i];
    g.y[0] = c2.p; g.y[1] = r2.p; g.y[2] = d2.p; g.gn[0] = g.gn[1] = g.gn[2] = nullptr;
    HN_TRY(conv_grouped(cx, g, 256, true, 512, 0));
    c = c2; r = r2; dd = d2;
  }
  const int aj = m->a_cls_out.cout;
  T cls = alloc(cx, k, c.h, c.w, aj, false), dep = alloc(cx, k, c.h, c.w, aj, false), reg;
  {
    GroupSpec g;
    g.count = 2;
    g.x[0] = c; g.x[1] = dd; g.w[0] = &m->a_cls_out; g.w[1] = &m->a_dep_out;
    g.y[0] = cls.p; g.y[1] = dep.p; g.gn[0] = g.gn[1] = nullptr;
    HN_TRY(conv_grouped(cx, g, 0, false, aj, 0));
  }
  HN_TRY(conv16(cx, r, m->a_reg_out, false, false, nullptr, false, reg));
  if (cx.dry) return HN_OK;
  return aggregate(cx, cls, reg, dep, valid, k, keypoints);
}

// ------------------------------------------------------------------------------------------------------------------
// FCOS (hn_amd/fcos_engine.py, default f16x3 path with lock-step grouped heads)
// ------------------------------------------------------------------------------------------------------------------
struct Geometry {
  int oh, ow, ph, pw;
};

// torchvision GeneralizedRCNNTransform.resize: float / tensor = tensor.reciprocal() * float in fp32, then
// floor(size * scale) in double
Geometry geometry(const hn_model_config& c, int h, int w) {
  const float inv_min = 1.0f / (float)(h < w ? h : w), inv_max = 1.0f / (float)(h < w ? w : h);
  const float a = inv_min * (float)c.min_size, b = inv_max * (float)c.max_size;
  const double scale = (double)(a < b ? a : b);
  Geometry g;
  g.oh = (int)((double)h * scale);
  g.ow = (int)((double)w * scale);
  g.ph = (g.oh + 31) / 32 * 32;
  g.pw = (g.ow + 31) / 32 * 32;
  return g;
}

struct FcosOut {
  float* boxes; float* scores; int32_t *labels, *sides, *level, *count;
  int cap;
  int32_t* contacts = nullptr;  // ext=True outputs (fcos.py:637-647), optional
  float* dxdymags = nullptr;
};

// a batch of differently sized images (torchvision batch_images, fcos.py:702-709): host tables of the device image
// pointers and sizes; every image is resized on its own into the common canvas, the boxes are rescaled per image
struct ImageList {
  const float* const* images;   // HOST array of n DEVICE pointers to [3][h_i][w_i]
  const int32_t *hs, *ws;       // HOST
};

Geometry list_canvas(const hn_model_config& c, const ImageList& ls, int n) {
  Geometry g{0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const Geometry gi = geometry(c, ls.hs[i], ls.ws[i]);
    g.oh = gi.oh > g.oh ? gi.oh : g.oh;
    g.ow = gi.ow > g.ow ? gi.ow : g.ow;
  }
  g.ph = (g.oh + 31) / 32 * 32;
  g.pw = (g.ow + 31) / 32 * 32;
  return g;
}

// The detector's network in the exact-f32 mode (FCOSEngine(precision="f32"): every convolution on the f32-MFMA kernel,
// GroupNorm statistics by a separate pass, its affine + ReLU applied while the next convolution stages its input; one
// FPN level after the other, FCOSEngine.head_level).  Fills lv (+ ext_lv) like the split-precision graph does.
int fcos_net_f32(Ctx& cx, const float* rgb, int n, int h, int w, const Geometry& g, const float* const* ptrs_dev,
                 const int32_t* geom_dev, const float* mean, const float* stdv, bool want_ext, hn_fcos_levels& lv, T* ext_lv,
                 int* hw) {
  hn_model* m = cx.m;
  T canvas = alloc(cx, n, g.ph, g.pw, 4, false);
  if (!cx.dry) {
    if (ptrs_dev) HN_TRY(hn_fcos_preprocess_list(ptrs_dev, geom_dev, canvas.p, 0, n, g.ph, g.pw, 0, mean, stdv, cx.stream));
    else HN_TRY(hn_fcos_preprocess_f32(rgb, (float*)canvas.p, n, h, w, g.oh, g.ow, g.ph, g.pw, mean, stdv, cx.stream));
  }
  T s0, x;
  HN_TRY(conv32(cx, canvas, m->f_stem, m->f_stem.cout, nullptr, false, nullptr, nullptr, 0, s0));
  release(cx, canvas);
  HN_TRY(maxpool32(cx, s0, x));
  release(cx, s0);
  T feats_c[3];
  int nf = 0;
  for (auto& b : m->f_blocks) {
    T o, idn, y;
    HN_TRY(conv32(cx, x, b.c1, b.c1.cout, nullptr, false, nullptr, nullptr, 0, o));
    if (b.has_ds) HN_TRY(conv32(cx, x, b.ds, 0, nullptr, false, nullptr, nullptr, 0, idn));
    else idn = x;
    HN_TRY(conv32(cx, o, b.c2, b.c2.cout, &idn, false, nullptr, nullptr, 0, y));
    release(cx, o);
    if (b.has_ds) release(cx, idn);
    bool saved = false;
    for (int f = 0; f < nf; ++f) saved = saved || feats_c[f].p == x.p;
    if (!saved) release(cx, x);
    x = y;
    if (b.last && b.layer >= 2) feats_c[nf++] = x;
  }
  T lat5, lat4, lat3;
  HN_TRY(conv32(cx, feats_c[2], m->f_inner[2], 0, nullptr, false, nullptr, nullptr, 0, lat5));
  HN_TRY(conv32(cx, feats_c[1], m->f_inner[1], 0, &lat5, true, nullptr, nullptr, 0, lat4));
  HN_TRY(conv32(cx, feats_c[0], m->f_inner[0], 0, &lat4, true, nullptr, nullptr, 0, lat3));
  const T lat[3] = {lat3, lat4, lat5};
  memset(&lv, 0, sizeof(lv));
  lv.num_levels = 3;
  for (int l = 0; l < 3; ++l) {
    T feat, t0;
    HN_TRY(conv32(cx, lat[l], m->f_layer[l], 0, nullptr, false, nullptr, nullptr, 0, feat));
    const int fh = feat.h, fw = feat.w;
    hw[l] = fh * fw;
    float* part = (float*)alloc_bytes(cx, (size_t)hn_groupnorm_scratch_floats(n, hw[l], 512, 64) * 4);
    float* sc0 = (float*)alloc_bytes(cx, (size_t)n * 512 * 4);
    float* sh0 = (float*)alloc_bytes(cx, (size_t)n * 512 * 4);
    HN_TRY(conv32(cx, feat, m->f_tower0, 0, nullptr, false, nullptr, nullptr, 0, t0));
    if (!cx.dry)
      HN_TRY(hn_groupnorm_affine_f32((const float*)t0.p, m->f_gn0_gamma, m->f_gn0_beta, n, hw[l], 512, 64, 1e-5f, part, sc0, sh0,
                                     cx.stream));
    T tower_out[2];
    const float *tsc[2], *tsh[2];
    int tstride[2];
    for (int tw = 0; tw < 2; ++tw) {   // 0: classification tower, 1: regression tower (fcos.py:267-329, 373-395)
      T xin = slice_channels(t0, tw * 256, tw * 256 + 256);
      const float *sc = sc0 + tw * 256, *sh = sh0 + tw * 256;
      int stride = 512;
      for (int layer = 0; layer < 3; ++layer) {
        T y;
        HN_TRY(conv32(cx, xin, tw == 0 ? m->f_cls_t[layer] : m->f_reg_t[layer], 0, nullptr, false, sc, sh, stride, y));
        float* nsc = (float*)alloc_bytes(cx, (size_t)n * 256 * 4);
        float* nsh = (float*)alloc_bytes(cx, (size_t)n * 256 * 4);
        if (!cx.dry)
          HN_TRY(hn_groupnorm_affine_f32((const float*)y.p, m->f_gn_gamma[layer] + tw * 256, m->f_gn_beta[layer] + tw * 256, n,
                                         hw[l], 256, 32, 1e-5f, part, nsc, nsh, cx.stream));
        release(cx, xin);   // (no-op for the slice of t0)
        xin = y;
        sc = nsc; sh = nsh; stride = 256;
      }
      tower_out[tw] = xin; tsc[tw] = sc; tsh[tw] = sh; tstride[tw] = stride;
    }
    T cls_lr, reg_ctr;
    HN_TRY(conv32(cx, tower_out[0], m->f_cls_out, 0, nullptr, false, tsc[0], tsh[0], tstride[0], cls_lr));
    if (want_ext) HN_TRY(conv32(cx, tower_out[0], m->f_ext_out, 3, nullptr, false, tsc[0], tsh[0], tstride[0], ext_lv[l]));
    HN_TRY(conv32(cx, tower_out[1], m->f_reg_out, 4, nullptr, false, tsc[1], tsh[1], tstride[1], reg_ctr));
    release(cx, tower_out[0]);
    release(cx, tower_out[1]);
    release(cx, t0);
    release(cx, feat);
    lv.h[l] = fh; lv.w[l] = fw; lv.stride[l] = g.ph / fh;
    lv.cls_lr[l] = (const float*)cls_lr.p; lv.reg_ctr[l] = (const float*)reg_ctr.p;
  }
  return HN_OK;
}

// The detector's network in the default split-precision mode (FCOSEngine, f16x3 / f16x1): fills lv (+ ext_lv) and hw.
int fcos_net_f16(Ctx& cx, const float* rgb, int n, int h, int w, const Geometry& g, const float* const* ptrs_dev,
                 const int32_t* geom_dev, const float* mean, const float* stdv, bool want_ext, hn_fcos_levels& lv, T* ext_lv,
                 int* hw) {
  hn_model* m = cx.m;
  const int border = 3;
  char* img16 = alloc_bytes(cx, (size_t)2 * n * (g.ph + 2 * border) * (g.pw + 2 * border) * 4 * 2);
  if (!cx.dry) {
    if (ptrs_dev) HN_TRY(hn_fcos_preprocess_list(ptrs_dev, geom_dev, img16, 1, n, g.ph, g.pw, border, mean, stdv, cx.stream));
    else HN_TRY(hn_fcos_preprocess_split(rgb, img16, n, h, w, g.oh, g.ow, g.ph, g.pw, border, mean, stdv, cx.stream));
  }
  int sh, sw;
  out_size(g.ph + 2 * border, g.pw + 2 * border, 7, 7, 2, 0, 1, sh, sw);
  // conv1 + bn1 + relu + 3x3/2 max pooling as ONE kernel (like hn_amd/fcos_engine.py): the half-resolution map is never stored
  T x = alloc(cx, n, (sh + 2 - 3) / 2 + 1, (sw + 2 - 3) / 2 + 1, 64, true);
  if (!cx.dry)
    HN_TRY(hn_conv_stem_pool_f16x3_terms(img16, n, g.ph, g.pw, border, 7, 2, 64, m->f_stem16.w16, m->f_stem16.bias, x.p, t_terms,
                                         cx.stream));
  T feats_c[3];
  int nf = 0;
  for (auto& b : m->f_blocks) {
    T o, idn, y;
    if (b.has_ds) {   // both read the block input: one grid (FCOSEngine.backbone)
      MultiItem it[2];
      it[0].x = x; it[0].w = &b.c1; it[0].relu = true;
      it[1].x = x; it[1].w = &b.ds; it[1].relu = false;
      HN_TRY(conv_multi(cx, it, 2));
      o = it[0].y; idn = it[1].y;
    } else {
      HN_TRY(conv16(cx, x, b.c1, true, true, nullptr, false, o));
      idn = x;
    }
    HN_TRY(conv16(cx, o, b.c2, true, true, &idn, false, y));
    release(cx, o);
    if (b.has_ds) release(cx, idn);
    bool saved = false;   // C3 / C4 / C5 are read again by the FPN laterals
    for (int f = 0; f < nf; ++f) saved = saved || feats_c[f].p == x.p;
    if (!saved) release(cx, x);
    x = y;
    if (b.last && b.layer >= 2) feats_c[nf++] = x;
  }
  T lat5, lat4, lat3;
  HN_TRY(conv16(cx, feats_c[2], m->f_inner[2], false, true, nullptr, false, lat5));
  HN_TRY(conv16(cx, feats_c[1], m->f_inner[1], false, true, &lat5, true, lat4));
  HN_TRY(conv16(cx, feats_c[0], m->f_inner[0], false, true, &lat4, true, lat3));
  const T lat[3] = {lat3, lat4, lat5};
  const int L = 3;
  T feats[3];
  {
    GroupSpec gs;
    gs.count = L;
    for (int l = 0; l < L; ++l) {
      feats[l] = alloc(cx, n, lat[l].h, lat[l].w, 256, true);
      gs.x[l] = lat[l]; gs.w[l] = &m->f_layer[l]; gs.y[l] = feats[l].p; gs.gn[l] = nullptr;
    }
    HN_TRY(conv_grouped(cx, gs, 0, true, 512, 0));
  }
  // ---- heads in lock-step over levels and towers (FCOSEngine.heads_grouped) ----
  float* parts[3];
  for (int l = 0; l < L; ++l) {
    hw[l] = feats[l].h * feats[l].w;
    if (hw[l] < 32) return hn::fail(HN_ERR_ARG, "image %dx%d too small for the grouped FCOS heads (a level has %d points)", h, w, hw[l]);
    parts[l] = (float*)alloc_bytes(cx, (size_t)hn_groupnorm_rows32_scratch_floats((int64_t)n * hw[l], 512) * 4);
  }
  auto new_t = [&](T* t) {
    for (int l = 0; l < L; ++l) t[l] = alloc(cx, n, feats[l].h, feats[l].w, 512, false);
  };
  T t[3];
  new_t(t);
  {
    GroupSpec gs;
    gs.count = L;
    for (int l = 0; l < L; ++l) { gs.x[l] = feats[l]; gs.w[l] = &m->f_tower0; gs.y[l] = t[l].p; gs.gn[l] = parts[l]; }
    HN_TRY(conv_grouped(cx, gs, 0, false, 512, 64));
  }
  float *scale[3], *shift[3];
  for (int l = 0; l < L; ++l) {
    scale[l] = (float*)alloc_bytes(cx, (size_t)n * 512 * 4);
    shift[l] = (float*)alloc_bytes(cx, (size_t)n * 512 * 4);
  }
  auto finalize = [&](const float* gamma, const float* beta) -> int {   // one launch over the levels
    if (cx.dry) return HN_OK;
    hn_gn_levels lv;
    lv.count = L;
    for (int l = 0; l < L; ++l) { lv.hw[l] = hw[l]; lv.partial[l] = parts[l]; lv.scale[l] = scale[l]; lv.shift[l] = shift[l]; }
    return hn_groupnorm_finalize_rows32_levels(&lv, gamma, beta, n, 512, 64, 1e-5f, cx.stream);
  };
  auto activate = [&](T* a) -> int {  // GroupNorm affine + ReLU + split: S32 [n][h][w][16][2][32], cls blocks 0-7, reg 8-15
    hn_split_levels lv;
    lv.count = L;
    for (int l = 0; l < L; ++l) {
      a[l] = alloc(cx, n, t[l].h, t[l].w, 512, true);
      lv.hw[l] = hw[l]; lv.x[l] = (const float*)t[l].p; lv.scale[l] = scale[l]; lv.shift[l] = shift[l]; lv.y16[l] = a[l].p;
    }
    const int rc = cx.dry ? HN_OK : hn_affine_split_f32_levels(&lv, 1, n, 512, 512, 512, 1024, cx.stream);   // one launch unless a level is cache-sized
    for (int l = 0; l < L; ++l) release(cx, t[l]);   // the raw tower output has been consumed
    return rc;
  };
  HN_TRY(finalize(m->f_gn0_gamma, m->f_gn0_beta));
  T a[3];
  for (int layer = 0; layer < 3; ++layer) {
    HN_TRY(activate(a));
    new_t(t);
    GroupSpec gs;
    gs.count = 2 * L;
    for (int l = 0; l < L; ++l) {
      gs.x[l] = slice_blocks(a[l], 0, 8); gs.w[l] = &m->f_cls_t[layer]; gs.y[l] = t[l].p; gs.gn[l] = parts[l];
      gs.x[L + l] = slice_blocks(a[l], 8, 16); gs.w[L + l] = &m->f_reg_t[layer]; gs.y[L + l] = t[l].p + 256 * 4;
      gs.gn[L + l] = (float*)((char*)parts[l] + 16 * 32);
    }
    HN_TRY(conv_grouped(cx, gs, 0, false, 512, 64));
    for (int l = 0; l < L; ++l) release(cx, a[l]);   // the activated input of this tower layer
    HN_TRY(finalize(m->f_gn_gamma[layer], m->f_gn_beta[layer]));
  }
  // the last GroupNorm apply pass rides on the head-output kernels' fragments where the fused P form takes the problem
  // (fcos_engine.heads_grouped; HN_FUSE_LAST_GN=0: A/B)
  hn_thin_levels tl;
  memset(&tl, 0, sizeof(tl));
  tl.count = L;
  for (int l = 0; l < L; ++l) { tl.h[l] = t[l].h; tl.w[l] = t[l].w; }
  const bool fuse_gn = !hn::env_flags().no_thin && !hn::env_flags().no_fuse_last_gn && t_terms != 1 && !want_ext &&
                       hn_conv3x3_thin_affine_applies(&tl, n, 256, m->f_cls_out.cout) &&
                       hn_conv3x3_thin_affine_applies(&tl, n, 256, m->f_reg_out.cout);
  memset(&lv, 0, sizeof(lv));
  lv.num_levels = L;
  T cls_lr[3], reg_ctr[3];
  if (fuse_gn) {
    hn_thin_affine ac, ar;
    hn_thin_levels yc = tl, yr = tl;
    ac.in_pix_stride = ar.in_pix_stride = 512;
    ac.affine_stride = ar.affine_stride = 512;
    for (int l = 0; l < L; ++l) {
      cls_lr[l] = alloc(cx, n, t[l].h, t[l].w, m->f_cls_out.cout, false);
      reg_ctr[l] = alloc(cx, n, t[l].h, t[l].w, 5, false);
      ac.x[l] = (const float*)t[l].p; ac.scale[l] = scale[l]; ac.shift[l] = shift[l];
      ar.x[l] = (const float*)t[l].p + 256; ar.scale[l] = scale[l] + 256; ar.shift[l] = shift[l] + 256;
      yc.y[l] = (float*)cls_lr[l].p; yr.y[l] = (float*)reg_ctr[l].p;
      lv.h[l] = t[l].h; lv.w[l] = t[l].w; lv.stride[l] = g.ph / t[l].h;
      lv.cls_lr[l] = (const float*)cls_lr[l].p; lv.reg_ctr[l] = (const float*)reg_ctr[l].p;
    }
    if (!cx.dry) {
      HN_TRY(hn_conv3x3_thin_affine_f16x3_levels(&yc, &ac, n, 256, m->f_cls_out.cout, m->f_cls_out.w16, m->f_cls_out.bias, 0, cx.stream));
      HN_TRY(hn_conv3x3_thin_affine_f16x3_levels(&yr, &ar, n, 256, 5, m->f_reg_out.w16, m->f_reg_out.bias, 4, cx.stream));
    }
    for (int l = 0; l < L; ++l) release(cx, t[l]);
  } else {
    HN_TRY(activate(a));
    GroupSpec gc, gr;
    gc.count = gr.count = L;
    const int ccls = m->f_cls_out.cout;
    for (int l = 0; l < L; ++l) {
      cls_lr[l] = alloc(cx, n, a[l].h, a[l].w, ccls, false);
      reg_ctr[l] = alloc(cx, n, a[l].h, a[l].w, 5, false);
      gc.x[l] = slice_blocks(a[l], 0, 8); gc.w[l] = &m->f_cls_out; gc.y[l] = cls_lr[l].p; gc.gn[l] = nullptr;
      gr.x[l] = slice_blocks(a[l], 8, 16); gr.w[l] = &m->f_reg_out; gr.y[l] = reg_ctr[l].p; gr.gn[l] = nullptr;
      lv.h[l] = a[l].h; lv.w[l] = a[l].w; lv.stride[l] = g.ph / a[l].h;
      lv.cls_lr[l] = (const float*)cls_lr[l].p; lv.reg_ctr[l] = (const float*)reg_ctr[l].p;
    }
    GroupSpec ge;
    if (want_ext) {  // ext heads: relu(hand_dydx_layer)[3] | hand_contact_state_layer[5] from the cls tower (fcos.py:255-264)
      ge.count = L;
      for (int l = 0; l < L; ++l) {
        ext_lv[l] = alloc(cx, n, a[l].h, a[l].w, 8, false);
        ge.x[l] = slice_blocks(a[l], 0, 8); ge.w[l] = &m->f_ext_out; ge.y[l] = ext_lv[l].p; ge.gn[l] = nullptr;
      }
    }
    // one launch for the two (ext: three) filter banks where they run the tap kernel (a single frame); else one each
    const GroupSpec* members[3] = {&gc, want_ext ? &ge : &gr, &gr};
    const int relus[3] = {0, want_ext ? 3 : 4, 4};
    HN_TRY(conv_thin_levels_group(cx, members, relus, want_ext ? 3 : 2));
  }
  return HN_OK;
}

int fcos_graph(Ctx& cx, const float* rgb, int n, int h, int w, const FcosOut& out, const ImageList* ls = nullptr) {
  hn_model* m = cx.m;
  const Geometry g = ls ? list_canvas(m->cfg, *ls, n) : geometry(m->cfg, h, w);
  const int want_cap = (g.ph / 8) * (g.pw / 8) + (g.ph / 16) * (g.pw / 16) + (g.ph / 32) * (g.pw / 32);
  if (out.cap != want_cap)
    return hn::fail(HN_ERR_ARG, "detection arrays must have hn_fcos_capacity%s = %d rows per image (got %d)", ls ? "_list" : "",
                    want_cap, out.cap);
  // the transform's normalisation (fcos.py:501-505): hn_model_config.image_mean / image_std, default ImageNet's
  float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  if (m->cfg.image_std[0] != 0.f || m->cfg.image_std[1] != 0.f || m->cfg.image_std[2] != 0.f)
    for (int c = 0; c < 3; ++c) { mean[c] = m->cfg.image_mean[c]; stdv[c] = m->cfg.image_std[c]; }
  const bool f32 = m->cfg.precision == HN_PRECISION_F32;
  float* ratios_dev = nullptr;
  const float** ptrs_dev = nullptr;
  int32_t* geom_dev = nullptr;
  if (ls) {   // device tables: image pointers, (h, w, oh, ow) rows, (ratio_h, ratio_w) rows; uploaded from pageable host memory,
              // which hipMemcpyAsync stages before it returns
    ptrs_dev = (const float**)alloc_bytes(cx, (size_t)n * 8);
    geom_dev = (int32_t*)alloc_bytes(cx, (size_t)n * 16);
    ratios_dev = (float*)alloc_bytes(cx, (size_t)n * 8);
    if (!cx.dry) {
      std::vector<int32_t> geom(4 * (size_t)n);
      std::vector<float> ratios(2 * (size_t)n);
      for (int i = 0; i < n; ++i) {
        const Geometry gi = geometry(m->cfg, ls->hs[i], ls->ws[i]);
        geom[4 * i] = ls->hs[i]; geom[4 * i + 1] = ls->ws[i]; geom[4 * i + 2] = gi.oh; geom[4 * i + 3] = gi.ow;
        ratios[2 * i] = (float)ls->hs[i] / (float)gi.oh;       // resize_boxes (fcos.py:770-783): fp32 / fp32
        ratios[2 * i + 1] = (float)ls->ws[i] / (float)gi.ow;
      }
      hipStream_t st = (hipStream_t)cx.stream;
      HN_CHECK_HIP(hipMemcpyAsync((void*)ptrs_dev, ls->images, (size_t)n * 8, hipMemcpyHostToDevice, st));
      HN_CHECK_HIP(hipMemcpyAsync(geom_dev, geom.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
      HN_CHECK_HIP(hipMemcpyAsync(ratios_dev, ratios.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
  }
  hn_fcos_levels lv;
  memset(&lv, 0, sizeof(lv));
  T ext_lv[3];
  int hw[3] = {0, 0, 0};
  if (f32) {
    HN_TRY(fcos_net_f32(cx, rgb, n, h, w, g, ptrs_dev, geom_dev, mean, stdv, out.contacts != nullptr, lv, ext_lv, hw));
  } else {
    HN_TRY(fcos_net_f16(cx, rgb, n, h, w, g, ptrs_dev, geom_dev, mean, stdv, out.contacts != nullptr, lv, ext_lv, hw));
  }
  const int cap = hw[0] + hw[1] + hw[2];
  if (out.cap != cap) return hn::fail(HN_ERR_ARG, "internal: %d anchor points but capacity %d", cap, out.cap);
  float* cb = (float*)alloc_bytes(cx, (size_t)n * cap * 16);
  float* cs = (float*)alloc_bytes(cx, (size_t)n * cap * 4);
  int32_t* cl = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
  int32_t* cd = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
  int32_t* cv = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
  int32_t* cc = (int32_t*)alloc_bytes(cx, (size_t)n * 4);
  int32_t* keep = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
  int32_t* point = out.contacts ? (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4) : nullptr;
  char* scratch = alloc_bytes(cx, (size_t)hn_fcos_nms_scratch_bytes(n, cap));
  const int64_t cand_ws_bytes = hn_fcos_candidates_ws_bytes(n, cap);
  char* cand_ws = alloc_bytes(cx, (size_t)cand_ws_bytes);
  if (cx.dry) return HN_OK;
  HN_TRY(hn_fcos_candidates_ws(&lv, n, m->cfg.num_classes, 0.7f /* fcos.py:600 */, cb, cs, cl, cd, cv, point, cc, cap, cand_ws,
                               cand_ws_bytes, cx.stream));
  // resize_boxes (fcos.py:770-783): fp32 / fp32
  if (ls) {
    HN_TRY(hn_fcos_nms_ratios(cb, cs, cl, cd, cv, cc, n, cap, 0.3 /* fcos.py:635 */, ratios_dev, scratch, out.boxes, out.scores,
                              out.labels, out.sides, out.level, keep, out.count, cx.stream));
  } else {
    const float ratio_h = (float)h / (float)g.oh, ratio_w = (float)w / (float)g.ow;
    HN_TRY(hn_fcos_nms(cb, cs, cl, cd, cv, cc, n, cap, 0.3 /* fcos.py:635 */, ratio_h, ratio_w, scratch, out.boxes, out.scores,
                       out.labels, out.sides, out.level, keep, out.count, cx.stream));
  }
  if (out.contacts) {
    const float* ext_ptrs[HN_FCOS_MAX_LEVELS] = {(const float*)ext_lv[0].p, (const float*)ext_lv[1].p, (const float*)ext_lv[2].p, nullptr, nullptr};
    HN_TRY(hn_fcos_ext_gather(&lv, ext_ptrs, keep, point, out.count, n, cap, out.contacts, out.dxdymags, cx.stream));
  }
  return HN_OK;
}

// dry pass (sizes the arena) + real pass
template <class F>
int run_planned(hn_model* m, const std::string& key0, void* stream, F&& graph) {
  HN_CHECK_ARG(m && m->finalized, "model is not finalized (hn_finalize)");
  t_terms = m->cfg.f16_terms == 1 ? 1 : 0;
  // the take / give sequence of a graph depends on the library's kernel-form switches (hn_set_form): a plan is only valid for
  // the switch state it was sized under
  const hn::EnvFlags& ef = hn::env_flags();
  const int sw = (ef.no_thin ? 1 : 0) | (ef.thin_tap ? 2 : 0) | (ef.thin_flat ? 4 : 0) | (ef.no_fuse_last_gn ? 8 : 0);
  const std::string key = key0 + "/" + std::to_string(sw);
  auto it = m->plan.find(key);
  size_t bytes;
  if (it == m->plan.end()) {
    Ctx dry{m, stream, true, nullptr};
    m->arena.dry = true;
    m->arena.reset();
    (void)m->arena.take(kConvWorkspaceBytes);
    HN_TRY(graph(dry));
    bytes = m->arena.off + 256;
    m->plan[key] = bytes;
  } else {
    bytes = it->second;
  }
  if (bytes > m->arena.cap) {  // first call with a larger problem: the one place a forward allocates (and synchronises)
    HN_CHECK_HIP(hipDeviceSynchronize());
    if (m->arena.base) HN_CHECK_HIP(hipFree(m->arena.base));
    m->arena.base = nullptr;
    m->arena.cap = 0;
    HN_CHECK_HIP(hipMalloc((void**)&m->arena.base, bytes));
    m->arena.cap = bytes;
  }
  // The arena is shared by every forward of this model: calls on different streams would use the same buffers without
  // ordering, so a forward on another stream than the previous one first waits (on the device) for the event the previous
  // forward recorded at its end.  Steady-state callers keep one stream and pay nothing; no raw stream handle of an earlier
  // call is ever touched again, and nothing here synchronises the host (legal under stream capture as well, where the
  // event bookkeeping is skipped: a captured step is replayed on one stream by construction).
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing((hipStream_t)stream, &cap_st);
  const bool capturing = cap_st != hipStreamCaptureStatusNone;
  if (!capturing) {
    if (!m->done) HN_CHECK_HIP(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
    if (m->has_last_stream && m->last_stream != stream) HN_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, m->done, 0));
  }
  m->arena.dry = false;
  m->arena.reset();
  Ctx cx{m, stream, false, nullptr};
  cx.ws = m->arena.take(kConvWorkspaceBytes);
  const int st = graph(cx);
  if (!capturing) {
    // (recorded also after a failed graph: whatever it enqueued still uses the arena)
    HN_CHECK_HIP(hipEventRecord(m->done, (hipStream_t)stream));
    m->last_stream = stream;
    m->has_last_stream = true;
  }
  if (st == HN_OK && m->arena.overflow)
    return hn::fail(HN_ERR_ARG, "model arena overflow: the graph asked for more than the %zu bytes planned for '%s' (results of this "
                    "call are invalid)", m->arena.cap, key.c_str());
  return st;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
extern "C" int hn_create(const hn_model_config* cfg, hn_model** out) {
  HN_CHECK_ARG(cfg && out, "hn_create: null pointer");
  HN_CHECK_ARG(cfg->parts & (HN_MODEL_FCOS | HN_MODEL_A2J), "hn_create: parts must name HN_MODEL_FCOS and / or HN_MODEL_A2J");
  HN_CHECK_ARG(cfg->num_classes >= 1 && cfg->num_classes <= 64 && cfg->num_joints >= 1, "bad class / joint count");
  HN_CHECK_ARG(cfg->f16_terms == 0 || cfg->f16_terms == 1 || cfg->f16_terms == 3, "f16_terms must be 0, 1 or 3");
  HN_CHECK_ARG(cfg->precision == HN_PRECISION_SPLIT || cfg->precision == HN_PRECISION_F32, "precision must be HN_PRECISION_SPLIT or HN_PRECISION_F32");
  HN_CHECK_ARG(cfg->precision != HN_PRECISION_F32 || cfg->f16_terms != 1, "f16_terms = 1 (f16x1) and HN_PRECISION_F32 exclude each other");
  for (int c = 0; c < 3; ++c)
    HN_CHECK_ARG(cfg->image_std[c] == cfg->image_std[c] && ((cfg->image_std[0] == 0.f) == (cfg->image_std[c] == 0.f)),
                 "image_std must be three non-zero values (or all zero for the default normalisation)");
  hn_model* m = new hn_model();
  m->cfg = *cfg;
  if (m->cfg.min_size <= 0) m->cfg.min_size = 800;     // fcos.py:460-461
  if (m->cfg.max_size <= 0) m->cfg.max_size = 1333;
  *out = m;
  return HN_OK;
}

extern "C" int hn_load_weight(hn_model* m, const char* name, const float* data, const int64_t* shape, int ndim) {
  HN_CHECK_ARG(m && name && data && (shape || ndim == 0), "hn_load_weight: null pointer");
  HN_CHECK_ARG(!m->finalized, "hn_load_weight after hn_finalize");
  HN_CHECK_ARG(ndim >= 0 && ndim <= 4, "weights have 0..4 dims");
  std::string key(name);
  if (key.rfind("a2j.", 0) == 0) key = key.substr(4);  // Lightning checkpoints prefix A2J with 'a2j.' (a2j/a2j.py:277)
  HostT t;
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) {
    HN_CHECK_ARG(shape[i] > 0, "bad shape");
    t.shape.push_back(shape[i]);
    numel *= shape[i];
  }
  t.v.assign(data, data + numel);
  m->sd[key] = std::move(t);
  return HN_OK;
}

extern "C" int hn_finalize(hn_model* m) {
  HN_CHECK_ARG(m && !m->finalized, "hn_finalize: null or already finalized");
  // a failed earlier attempt (missing weight, fp16-range error) may have left partial state: start clean, so that a
  // second hn_finalize after the missing weight was loaded builds the graph once
  m->a_blocks.clear();
  m->f_blocks.clear();
  for (void* q : m->owned) (void)hipFree(q);
  m->owned.clear();
  const bool f32 = m->cfg.precision == HN_PRECISION_F32;   // exact mode: fp32 banks only, no fp16-range condition
  if (m->cfg.parts & HN_MODEL_A2J) {
    const std::string p = "Backbone.model.";
    HN_TRY(pack_conv(m, p + "conv1.weight", "", p + "bn1", 2, 3, 1, !m->cfg.rgbd, m->a_stem));
    if (!f32) HN_TRY(pack_stem_split(m, p + "conv1.weight", p + "bn1", m->a_stem16, !m->cfg.rgbd));
    const int planes[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 1}, dils[4] = {1, 1, 1, 2};
    (void)planes;
    for (int li = 1; li <= 4; ++li)
      for (int b = 0; b < blocks[li - 1]; ++b) {
        const std::string q = p + "layer" + std::to_string(li) + "." + std::to_string(b) + ".";
        const int st = b == 0 ? strides[li - 1] : 1, dl = b == 0 ? 1 : dils[li - 1];  // a2j/resnet.py:133-147
        hn_model::Bneck k;
        k.layer = li;
        HN_TRY(pack_conv(m, q + "conv1.weight", "", q + "bn1", 1, 0, 1, false, k.c1));
        HN_TRY(pack_conv(m, q + "conv2.weight", "", q + "bn2", st, dl, dl, false, k.c2));
        HN_TRY(pack_conv(m, q + "conv3.weight", "", q + "bn3", 1, 0, 1, false, k.c3));
        k.has_ds = find(m, q + "downsample.0.weight") != nullptr;
        if (k.has_ds) HN_TRY(pack_conv(m, q + "downsample.0.weight", "", q + "downsample.1", st, 0, 1, false, k.ds));
        m->a_blocks.push_back(std::move(k));
      }
    auto head = [&](const std::string& name, ConvW* convs4, ConvW& outc) -> int {
      for (int i = 1; i <= 4; ++i) {
        const std::string c = name + ".conv" + std::to_string(i), bn = name + ".bn" + std::to_string(i);
        HN_TRY(pack_conv(m, c + ".weight", c + ".bias", bn, 1, 1, 1, false, convs4[i - 1]));
      }
      return pack_conv(m, name + ".output.weight", name + ".output.bias", "", 1, 1, 1, false, outc);
    };
    ConvW reg4[4], dep4[4];
    HN_TRY(head("classificationModel", m->a_cls, m->a_cls_out));
    HN_TRY(head("regressionModel", reg4, m->a_reg_out));
    HN_TRY(head("DepthRegressionModel", dep4, m->a_dep_out));
    HN_TRY(concat_cout(reg4[0], dep4[0], m->a_regdep1, "A2J regression+depth conv1", f32));
    for (int i = 0; i < 3; ++i) { m->a_reg[i] = reg4[i + 1]; m->a_dep[i] = dep4[i + 1]; }
    HN_CHECK_ARG(m->a_cls_out.cout == 16 * m->cfg.num_joints, "checkpoint does not match num_joints");
    HN_TRY(upload(m, m->a_stem));
    if (!f32) HN_TRY(upload(m, m->a_stem16));
    for (auto& k : m->a_blocks) {
      HN_TRY(upload(m, k.c1)); HN_TRY(upload(m, k.c2)); HN_TRY(upload(m, k.c3));
      if (k.has_ds) HN_TRY(upload(m, k.ds));
    }
    for (int i = 0; i < 4; ++i) HN_TRY(upload(m, m->a_cls[i]));
    for (int i = 0; i < 3; ++i) { HN_TRY(upload(m, m->a_reg[i])); HN_TRY(upload(m, m->a_dep[i])); }
    HN_TRY(upload(m, m->a_regdep1)); HN_TRY(upload(m, m->a_cls_out)); HN_TRY(upload(m, m->a_reg_out)); HN_TRY(upload(m, m->a_dep_out));
  }
  if (m->cfg.parts & HN_MODEL_FCOS) {
    const std::string p = "backbone.body.";
    if (f32) HN_TRY(pack_conv(m, p + "conv1.weight", "", p + "bn1", 2, 3, 1, false, m->f_stem));
    else HN_TRY(pack_stem_split(m, p + "conv1.weight", p + "bn1", m->f_stem16));
    const int blocks[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 2};
    for (int li = 1; li <= 4; ++li)
      for (int b = 0; b < blocks[li - 1]; ++b) {
        const std::string q = p + "layer" + std::to_string(li) + "." + std::to_string(b) + ".";
        const int st = b == 0 ? strides[li - 1] : 1;
        hn_model::Basic k;
        k.layer = li;
        k.last = b == blocks[li - 1] - 1;
        HN_TRY(pack_conv(m, q + "conv1.weight", "", q + "bn1", st, 1, 1, false, k.c1));
        HN_TRY(pack_conv(m, q + "conv2.weight", "", q + "bn2", 1, 1, 1, false, k.c2));
        k.has_ds = find(m, q + "downsample.0.weight") != nullptr;
        if (k.has_ds) HN_TRY(pack_conv(m, q + "downsample.0.weight", "", q + "downsample.1", st, 0, 1, false, k.ds));
        m->f_blocks.push_back(std::move(k));
      }
    const std::string f = "backbone.fpn.";
    for (int i = 0; i < 3; ++i) {
      const std::string a = f + "inner_blocks." + std::to_string(i), b = f + "layer_blocks." + std::to_string(i);
      HN_TRY(pack_conv(m, a + ".weight", a + ".bias", "", 1, 0, 1, false, m->f_inner[i]));
      HN_TRY(pack_conv(m, b + ".weight", b + ".bias", "", 1, 1, 1, false, m->f_layer[i]));
    }
    const std::string c = "head.classification_head", r = "head.regression_head";
    auto tconv = [&](const std::string& t, int i, ConvW& cw) {
      const std::string n = t + ".conv." + std::to_string(3 * i);
      return pack_conv(m, n + ".weight", n + ".bias", "", 1, 1, 1, false, cw);
    };
    ConvW c0, r0;
    HN_TRY(tconv(c, 0, c0));
    HN_TRY(tconv(r, 0, r0));
    HN_TRY(concat_cout(c0, r0, m->f_tower0, "FCOS tower layer 0", f32));
    for (int i = 1; i < 4; ++i) { HN_TRY(tconv(c, i, m->f_cls_t[i - 1])); HN_TRY(tconv(r, i, m->f_reg_t[i - 1])); }
    auto gn = [&](int i, const char* k, std::vector<float>& v) -> int {  // cls | reg stacked
      v.clear();
      for (const std::string* t : {&c, &r}) {
        const HostT* x;
        HN_TRY(need(m, *t + ".conv." + std::to_string(3 * i + 1) + "." + k, &x, 1));
        v.insert(v.end(), x->v.begin(), x->v.end());
      }
      return HN_OK;
    };
    std::vector<float> v;
    HN_TRY(gn(0, "weight", v)); HN_TRY(upload_vec(m, v, &m->f_gn0_gamma));
    HN_TRY(gn(0, "bias", v)); HN_TRY(upload_vec(m, v, &m->f_gn0_beta));
    for (int i = 1; i < 4; ++i) {
      HN_TRY(gn(i, "weight", v)); HN_TRY(upload_vec(m, v, &m->f_gn_gamma[i - 1]));
      HN_TRY(gn(i, "bias", v)); HN_TRY(upload_vec(m, v, &m->f_gn_beta[i - 1]));
    }
    ConvW cl, lr, br, bc;
    HN_TRY(pack_conv(m, c + ".cls_logits.weight", c + ".cls_logits.bias", "", 1, 1, 1, false, cl));
    HN_TRY(pack_conv(m, c + ".hand_lr_layer.weight", c + ".hand_lr_layer.bias", "", 1, 1, 1, false, lr));
    HN_TRY(pack_conv(m, r + ".bbox_reg.weight", r + ".bbox_reg.bias", "", 1, 1, 1, false, br));
    HN_TRY(pack_conv(m, r + ".bbox_ctrness.weight", r + ".bbox_ctrness.bias", "", 1, 1, 1, false, bc));
    HN_TRY(concat_cout(cl, lr, m->f_cls_out, "FCOS cls_logits+hand_lr", f32));
    HN_TRY(concat_cout(br, bc, m->f_reg_out, "FCOS bbox_reg+ctrness", f32));
    HN_CHECK_ARG(m->f_cls_out.cout == m->cfg.num_classes + 2, "checkpoint does not match num_classes");
    if (m->cfg.ext) {
      ConvW dx, ct;
      HN_TRY(pack_conv(m, c + ".hand_dydx_layer.weight", c + ".hand_dydx_layer.bias", "", 1, 1, 1, false, dx));
      HN_TRY(pack_conv(m, c + ".hand_contact_state_layer.weight", c + ".hand_contact_state_layer.bias", "", 1, 1, 1, false, ct));
      HN_TRY(concat_cout(dx, ct, m->f_ext_out, "FCOS ext heads", f32));
      HN_CHECK_ARG(m->f_ext_out.cout == 8, "ext heads must have 3 + 5 output channels");
      HN_TRY(upload(m, m->f_ext_out));
    }
    if (f32) HN_TRY(upload(m, m->f_stem));
    else HN_TRY(upload(m, m->f_stem16));
    for (auto& k : m->f_blocks) {
      HN_TRY(upload(m, k.c1)); HN_TRY(upload(m, k.c2));
      if (k.has_ds) HN_TRY(upload(m, k.ds));
    }
    for (int i = 0; i < 3; ++i) {
      HN_TRY(upload(m, m->f_inner[i])); HN_TRY(upload(m, m->f_layer[i])); HN_TRY(upload(m, m->f_cls_t[i])); HN_TRY(upload(m, m->f_reg_t[i]));
    }
    HN_TRY(upload(m, m->f_tower0)); HN_TRY(upload(m, m->f_cls_out)); HN_TRY(upload(m, m->f_reg_out));
  }
  m->sd.clear();
  m->finalized = true;
  return HN_OK;
}

extern "C" int64_t hn_fcos_capacity(const hn_model* m, int h, int w) {
  if (!m || h <= 0 || w <= 0) return 0;
  const Geometry g = geometry(m->cfg, h, w);
  return (int64_t)(g.ph / 8) * (g.pw / 8) + (int64_t)(g.ph / 16) * (g.pw / 16) + (int64_t)(g.ph / 32) * (g.pw / 32);
}

extern "C" int64_t hn_fcos_capacity_list(const hn_model* m, const int32_t* hs, const int32_t* ws, int n) {
  if (!m || !hs || !ws || n <= 0) return 0;
  for (int i = 0; i < n; ++i)
    if (hs[i] <= 0 || ws[i] <= 0) return 0;
  const ImageList ls{nullptr, hs, ws};
  const Geometry g = list_canvas(m->cfg, ls, n);
  return (int64_t)(g.ph / 8) * (g.pw / 8) + (int64_t)(g.ph / 16) * (g.pw / 16) + (int64_t)(g.ph / 32) * (g.pw / 32);
}

extern "C" int hn_fcos_forward_list(hn_model* m, const float* const* images, const int32_t* hs, const int32_t* ws, int n,
                                    float* det_boxes, float* det_scores, int32_t* det_labels, int32_t* det_sides,
                                    int32_t* det_level, int32_t* det_count, int cap, void* stream) {
  HN_CHECK_ARG(m && images && hs && ws && det_boxes && det_scores && det_labels && det_sides && det_level && det_count,
               "hn_fcos_forward_list: null pointer");
  HN_CHECK_ARG(m->cfg.parts & HN_MODEL_FCOS, "model was created without HN_MODEL_FCOS");
  HN_CHECK_ARG(n > 0, "empty image list");
  for (int i = 0; i < n; ++i) HN_CHECK_ARG(images[i] && hs[i] > 0 && ws[i] > 0, "image %d: null pointer or empty", i);
  const ImageList ls{images, hs, ws};
  const Geometry g = list_canvas(m->cfg, ls, n);
  const std::string key = "fcosl:" + std::to_string(n) + "x" + std::to_string(g.ph) + "x" + std::to_string(g.pw);
  FcosOut out{det_boxes, det_scores, det_labels, det_sides, det_level, det_count, cap};
  return run_planned(m, key, stream, [&](Ctx& cx) -> int { return fcos_graph(cx, nullptr, n, 0, 0, out, &ls); });
}

extern "C" int hn_a2j_forward(hn_model* m, const float* crops, int k, int h, int w, const int32_t* valid, float* keypoints,
                              void* stream) {
  HN_CHECK_ARG(m && crops && keypoints, "hn_a2j_forward: null pointer");
  HN_CHECK_ARG(m->cfg.parts & HN_MODEL_A2J, "model was created without HN_MODEL_A2J");
  HN_CHECK_ARG(!m->cfg.rgbd, "RGB-D crops are NHWC inside the pipeline: use hn_handnet_forward");
  HN_CHECK_ARG(k > 0 && h >= 32 && w >= 32, "bad crop batch");
  const std::string key = "a2j:" + std::to_string(k) + "x" + std::to_string(h) + "x" + std::to_string(w);
  return run_planned(m, key, stream, [&](Ctx& cx) -> int {
    T x;
    x.n = k; x.h = h; x.w = w; x.c = 4; x.ps = 4; x.split = false;
    x.p = cx.m->arena.take((size_t)k * h * w * 16);
    // flags of the call's own, like A2JEngine.forward: the stem marks a crop that holds NaN / inf pixels (2) and the
    // aggregation writes its NaN row (a2j/a2j.py:243-250 returns NaN for such a crop); the caller's `valid` stays read-only
    int32_t* flags = (int32_t*)alloc_bytes(cx, (size_t)k * 4);
    if (!cx.dry) {
      HN_TRY(hn_pack_depth_nhwc(crops, (float*)x.p, k, h * w, 4, cx.stream));
      if (valid) HN_CHECK_HIP(hipMemcpyAsync(flags, valid, (size_t)k * 4, hipMemcpyDeviceToDevice, (hipStream_t)cx.stream));
      else HN_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)flags, 1, (size_t)k, (hipStream_t)cx.stream));
    }
    return a2j_graph(cx, x, flags, keypoints, flags);
  });
}

extern "C" int hn_fcos_forward(hn_model* m, const float* rgb, int n, int h, int w, float* det_boxes, float* det_scores,
                               int32_t* det_labels, int32_t* det_sides, int32_t* det_level, int32_t* det_count, int cap,
                               void* stream) {
  HN_CHECK_ARG(m && rgb && det_boxes && det_scores && det_labels && det_sides && det_level && det_count, "hn_fcos_forward: null pointer");
  HN_CHECK_ARG(m->cfg.parts & HN_MODEL_FCOS, "model was created without HN_MODEL_FCOS");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad image batch");
  const std::string key = "fcos:" + std::to_string(n) + "x" + std::to_string(h) + "x" + std::to_string(w);
  FcosOut out{det_boxes, det_scores, det_labels, det_sides, det_level, det_count, cap};
  return run_planned(m, key, stream, [&](Ctx& cx) -> int { return fcos_graph(cx, rgb, n, h, w, out); });
}

extern "C" int hn_fcos_forward_ext(hn_model* m, const float* rgb, int n, int h, int w, float* det_boxes, float* det_scores,
                                   int32_t* det_labels, int32_t* det_sides, int32_t* det_level, int32_t* det_count,
                                   int32_t* det_contacts, float* det_dxdymags, int cap, void* stream) {
  HN_CHECK_ARG(m && rgb && det_boxes && det_scores && det_labels && det_sides && det_level && det_count && det_contacts && det_dxdymags,
               "hn_fcos_forward_ext: null pointer");
  HN_CHECK_ARG((m->cfg.parts & HN_MODEL_FCOS) && m->cfg.ext, "model was created without HN_MODEL_FCOS / ext = 1");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad image batch");
  const std::string key = "fcos_ext:" + std::to_string(n) + "x" + std::to_string(h) + "x" + std::to_string(w);
  FcosOut out{det_boxes, det_scores, det_labels, det_sides, det_level, det_count, cap};
  out.contacts = det_contacts;
  out.dxdymags = det_dxdymags;
  return run_planned(m, key, stream, [&](Ctx& cx) -> int { return fcos_graph(cx, rgb, n, h, w, out); });
}

extern "C" int hn_handnet_forward(hn_model* m, const float* rgb, const float* depth, int n, int h, int w, float* keypoints,
                                  int64_t* crop_box, int32_t* has_hand, void* stream) {
  HN_CHECK_ARG(m && rgb && depth && keypoints && crop_box && has_hand, "hn_handnet_forward: null pointer");
  HN_CHECK_ARG((m->cfg.parts & (HN_MODEL_FCOS | HN_MODEL_A2J)) == (HN_MODEL_FCOS | HN_MODEL_A2J), "model needs both HN_MODEL_FCOS and HN_MODEL_A2J");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad image batch");
  const int cap = (int)hn_fcos_capacity(m, h, w);
  const std::string key = "handnet:" + std::to_string(n) + "x" + std::to_string(h) + "x" + std::to_string(w);
  return run_planned(m, key, stream, [&](Ctx& cx) -> int {
    FcosOut out;
    out.cap = cap;
    out.boxes = (float*)alloc_bytes(cx, (size_t)n * cap * 16);
    out.scores = (float*)alloc_bytes(cx, (size_t)n * cap * 4);
    out.labels = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
    out.sides = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
    out.level = (int32_t*)alloc_bytes(cx, (size_t)n * cap * 4);
    out.count = (int32_t*)alloc_bytes(cx, (size_t)n * 4);
    HN_TRY(fcos_graph(cx, rgb, n, h, w, out));
    T crops;
    crops.n = n; crops.h = kCrop; crops.w = kCrop; crops.c = 4; crops.ps = 4; crops.split = false;
    crops.p = alloc_bytes(cx, (size_t)n * kCrop * kCrop * 16);
    if (!cx.dry)
      HN_TRY(hn_crop_resize(out.boxes, out.labels, out.count, cap, m->cfg.num_classes - 1, depth, n, m->cfg.rgbd ? 4 : 1,
                            m->cfg.rgbd ? 1 : 0, h, w, kCrop, 4, crop_box, has_hand, (float*)crops.p, cx.stream));
    return a2j_graph(cx, crops, has_hand, keypoints, has_hand);
  });
}

extern "C" int hn_handnet_forward_xyz(hn_model* m, const float* rgb, const float* depth, int n, int h, int w, const float* paras,
                                      const hn_convert_opts* opts, float* keypoints, float* image_uvd, float* xyz_mm,
                                      int64_t* crop_box, int32_t* has_hand, void* stream) {
  HN_CHECK_ARG(image_uvd || xyz_mm, "hn_handnet_forward_xyz: no converted output requested (use hn_handnet_forward)");
  HN_CHECK_ARG(!xyz_mm || paras || (opts && opts->sample_paras), "hn_handnet_forward_xyz: camera xyz needs the intrinsics (fx, fy, cx, cy)");
  HN_CHECK_ARG(!opts || !opts->sample_box, "hn_handnet_forward_xyz: the boxes are the detector's (opts->sample_box must be NULL)");
  const ConvertReq req{crop_box, paras, opts, image_uvd, xyz_mm};
  t_convert = &req;     // (this host thread's forward; the aggregation's launch reads it)
  const int rc = hn_handnet_forward(m, rgb, depth, n, h, w, keypoints, crop_box, has_hand, stream);
  t_convert = nullptr;
  return rc;
}

extern "C" int hn_destroy(hn_model* m) {
  if (!m) return HN_OK;
  for (void* p : m->owned) (void)hipFree(p);
  if (m->arena.base) (void)hipFree(m->arena.base);
  if (m->done) (void)hipEventDestroy(m->done);
  delete m;
  return HN_OK;
}
