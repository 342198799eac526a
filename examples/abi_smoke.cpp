// A host that is NOT Python: binds include/handnet_hip.h directly (plain pointers, sizes, POD structs), runs
// a split-fp16 convolution and an NMS through the op-level C ABI and checks both against host loops; given a model
// blob (argv[1], written by tests/test_abi_gpu.py: a reference-layout A2J state_dict, seeded crops and the keypoints the
// REFERENCE produced for them) it also runs a whole A2J forward through the model-level ABI
// (hn_create / hn_load_weight / hn_finalize / hn_a2j_forward / hn_destroy) and checks it against that fixture.
//   hipcc --offload-arch=gfx950 -Iinclude examples/abi_smoke.cpp -Lhandnet-pipeline_amd/csrc -lhandnet_hip -o abi_smoke
// Exit code 0 and "abi_smoke ok" on success.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "handnet_hip.h"

#define HIP_OK(e)                                                                  \
  do {                                                                             \
    hipError_t err_ = (e);                                                         \
    if (err_ != hipSuccess) {                                                      \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(err_), __FILE__, __LINE__); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define HN_OK_OR_DIE(e)                                                            \
  do {                                                                             \
    if ((e) != 0) {                                                                \
      std::fprintf(stderr, "handnet error: %s (%s:%d)\n", hn_last_error(), __FILE__, __LINE__); \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

static float frand(unsigned& s) {  // deterministic LCG in [-1, 1)
  s = s * 1664525u + 1013904223u;
  return (float)((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

// blob: "HNB1", int32 tensors, then per tensor {int32 name_len, name, int32 ndim, int64 shape[ndim], float data[]},
// then int32 k, h, w, float crops[k*h*w], int32 joints, float expected[k*joints*3]
static int run_a2j_model(const char* path, hipStream_t st) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); return 10; }
  auto rd = [&](void* dst, size_t bytes) { return std::fread(dst, 1, bytes, f) == bytes; };
  char magic[4];
  int32_t count = 0;
  if (!rd(magic, 4) || std::memcmp(magic, "HNB1", 4) != 0 || !rd(&count, 4)) return 11;
  hn_model_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.parts = HN_MODEL_A2J; cfg.num_classes = 3; cfg.num_joints = 21;
  hn_model* m = nullptr;
  HN_OK_OR_DIE(hn_create(&cfg, &m));
  std::vector<float> buf;
  for (int i = 0; i < count; ++i) {
    int32_t len = 0, ndim = 0;
    if (!rd(&len, 4)) return 12;
    std::vector<char> name(len + 1, 0);
    int64_t shape[4] = {0, 0, 0, 0};
    if (!rd(name.data(), len) || !rd(&ndim, 4) || ndim > 4 || !rd(shape, 8 * (size_t)ndim)) return 12;
    size_t numel = 1;
    for (int d = 0; d < ndim; ++d) numel *= (size_t)shape[d];
    buf.resize(numel);
    if (!rd(buf.data(), numel * 4)) return 12;
    HN_OK_OR_DIE(hn_load_weight(m, name.data(), buf.data(), shape, ndim));
  }
  HN_OK_OR_DIE(hn_finalize(m));
  int32_t k = 0, h = 0, w = 0, joints = 0;
  if (!rd(&k, 4) || !rd(&h, 4) || !rd(&w, 4)) return 13;
  std::vector<float> crops((size_t)k * h * w);
  if (!rd(crops.data(), crops.size() * 4) || !rd(&joints, 4)) return 13;
  std::vector<float> expect((size_t)k * joints * 3), got(expect.size());
  if (!rd(expect.data(), expect.size() * 4)) return 13;
  std::fclose(f);
  float *dc, *dk;
  HIP_OK(hipMalloc(&dc, crops.size() * 4));
  HIP_OK(hipMalloc(&dk, got.size() * 4));
  HIP_OK(hipMemcpy(dc, crops.data(), crops.size() * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep)   // the second call re-uses the plan and the arena: no allocation
    HN_OK_OR_DIE(hn_a2j_forward(m, dc, k, h, w, nullptr, dk, st));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(got.data(), dk, got.size() * 4, hipMemcpyDeviceToHost));
  double err = 0.0;
  for (size_t i = 0; i < got.size(); ++i) err = std::fmax(err, std::fabs((double)got[i] - expect[i]));
  std::printf("a2j model forward (%d crops, C++ layer graph): max |keypoint - reference| %.3g (bound 1e-3)\n", k, err);
  HN_OK_OR_DIE(hn_destroy(m));
  return err < 1e-3 ? 0 : 14;
}

int main(int argc, char** argv) {
  if (hn_abi_version() != HN_ABI_VERSION) {
    std::fprintf(stderr, "ABI mismatch: header %d, library %d\n", HN_ABI_VERSION, hn_abi_version());
    return 1;
  }
  // ---- convolution: n=2, 9x7, 64 -> 40 channels, 3x3, pad 1, bias + ReLU, fp32 output ----
  const int n = 2, h = 9, w = 7, cin = 64, cout = 40, r = 3;
  unsigned seed = 12345;
  std::vector<float> x((size_t)n * h * w * cin), wt((size_t)cout * r * r * cin), bias(cout);
  for (auto& v : x) v = frand(seed);
  for (auto& v : wt) v = frand(seed) * 0.05f;
  for (auto& v : bias) v = frand(seed) * 0.1f;
  // weights in the split layout fp16 [cout][(cin/32)*r*r][2][32]: channel block outer, tap inner, hi run | lo run
  std::vector<_Float16> w16((size_t)cout * r * r * cin * 2);
  for (int o = 0; o < cout; ++o)
    for (int cb = 0; cb < cin / 32; ++cb)
      for (int t = 0; t < r * r; ++t)
        for (int c = 0; c < 32; ++c) {
          const float v = wt[((size_t)o * r * r + t) * cin + cb * 32 + c];
          const _Float16 hi = (_Float16)v;
          const size_t base = (((size_t)o * (cin / 32) + cb) * r * r + t) * 64;
          w16[base + c] = hi;
          w16[base + 32 + c] = (_Float16)(v - (float)hi);
        }
  float *dx, *db, *dy;
  void *dx16, *dw16;
  HIP_OK(hipMalloc(&dx, x.size() * 4));
  HIP_OK(hipMalloc(&dx16, x.size() * 4));
  HIP_OK(hipMalloc(&dw16, w16.size() * 2));
  HIP_OK(hipMalloc(&db, bias.size() * 4));
  HIP_OK(hipMalloc(&dy, (size_t)n * h * w * cout * 4));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  // fp32 NHWC -> S32 split activations (no affine)
  HN_OK_OR_DIE(hn_affine_split_f32(dx, nullptr, nullptr, 0, n, h * w, cin, cin, 0, dx16, 2 * cin, st));
  hn_conv_desc d;
  std::memset(&d, 0, sizeof(d));
  d.n = n; d.h = h; d.w = w; d.cin = cin; d.cout = cout; d.r = r; d.s = r; d.stride = 1; d.pad = 1; d.dil = 1;
  d.oh = h; d.ow = w; d.relu_cols = cout;
  HN_OK_OR_DIE(hn_conv2d_nhwc_f16x3(&d, dx16, dw16, db, nullptr, dy, st));
  HIP_OK(hipStreamSynchronize(st));
  std::vector<float> y((size_t)n * h * w * cout);
  HIP_OK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
  double max_err = 0.0, max_ref = 0.0;
  for (int b = 0; b < n; ++b)
    for (int oy = 0; oy < h; ++oy)
      for (int ox = 0; ox < w; ++ox)
        for (int o = 0; o < cout; ++o) {
          double acc = bias[o];
          for (int ky = 0; ky < r; ++ky)
            for (int kx = 0; kx < r; ++kx) {
              const int iy = oy + ky - 1, ix = ox + kx - 1;
              if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
              for (int c = 0; c < cin; ++c)
                acc += (double)x[(((size_t)b * h + iy) * w + ix) * cin + c] * wt[((size_t)o * r * r + ky * r + kx) * cin + c];
            }
          if (acc < 0) acc = 0;
          const double got = y[(((size_t)b * h + oy) * w + ox) * cout + o];
          max_err = std::fmax(max_err, std::fabs(got - acc));
          max_ref = std::fmax(max_ref, std::fabs(acc));
        }
  std::printf("conv f16x3: max |err| %.3g on |ref| <= %.3g\n", max_err, max_ref);
  if (!(max_err <= 2e-5 * std::fmax(1.0, max_ref))) return 4;

  // ---- NMS: four boxes, descending scores; box 1 overlaps box 0 (IoU 0.68), box 3 overlaps box 2 ----
  const float boxes[16] = {0, 0, 10, 10, 1, 1, 11, 11, 20, 20, 30, 30, 21, 20, 31, 30};
  const float scores[4] = {0.9f, 0.8f, 0.7f, 0.95f};
  float *dbx, *dsc;
  int32_t *dkeep, *dnum;
  void* scratch;
  HIP_OK(hipMalloc(&dbx, sizeof(boxes)));
  HIP_OK(hipMalloc(&dsc, sizeof(scores)));
  HIP_OK(hipMalloc(&dkeep, 4 * sizeof(int32_t)));
  HIP_OK(hipMalloc(&dnum, sizeof(int32_t)));
  HIP_OK(hipMalloc(&scratch, (size_t)hn_fcos_nms_scratch_bytes(1, 4)));
  HIP_OK(hipMemcpy(dbx, boxes, sizeof(boxes), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsc, scores, sizeof(scores), hipMemcpyHostToDevice));
  HN_OK_OR_DIE(hn_nms(dbx, dsc, 4, 0.3, scratch, dkeep, dnum, st));
  HIP_OK(hipStreamSynchronize(st));
  int32_t keep[4] = {-1, -1, -1, -1}, num = -1;
  HIP_OK(hipMemcpy(keep, dkeep, sizeof(keep), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&num, dnum, sizeof(num), hipMemcpyDeviceToHost));
  std::printf("nms: kept %d -> [%d, %d]\n", num, keep[0], keep[1]);
  if (num != 2 || keep[0] != 3 || keep[1] != 0) return 5;   // score order: 3 (suppresses 2), 0 (suppresses 1)

  // a bad argument must fail loudly, not crash
  d.cin = 48;
  if (hn_conv2d_nhwc_f16x3(&d, dx16, dw16, db, nullptr, dy, st) == 0 || std::strlen(hn_last_error()) == 0) return 6;
  if (argc > 1) {
    const int rc = run_a2j_model(argv[1], st);
    if (rc) return rc;
  }
  std::printf("abi_smoke ok\n");
  return 0;
}
