// HBM-bound helpers of the A2J / ResNet path: 3x3/2 max pooling, depth -> NHWC packing
// and the softmax-weighted anchor aggregation (a2j/anchor.py:57-82).
#include "hn_common.h"

#include <float.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one thread = one output pixel x 4 channels; consecutive threads walk channels first,
// so a wave reads/writes contiguous 16-B pieces (coalesced NHWC).
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int n, int h, int w, int c4, int oh, int ow) {
  const long total = (long)n * oh * ow * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    long pix = i / c4;
    const int x_ = (int)(pix % ow);
    pix /= ow;
    const int y_ = (int)(pix % oh);
    const int img = (int)(pix / oh);
    f32x4 m = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = y_ * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = x_ * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)w) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)img * h + iy) * w + ix) * c4 * 4 + cc * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = hn::max_nan(m[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

__global__ __launch_bounds__(256) void pack_depth_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         long npix, int c4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix * c4; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c4;
    const int cc = (int)(i - pix * c4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cc == 0) v[0] = src[pix];
    *reinterpret_cast<f32x4*>(dst + i * 4) = v;
  }
}

// kJointSplit workgroups per crop, each owning a contiguous range of joints (the joints are independent; one workgroup
// per crop was 28 us of the batch-1 frame on one CU).  Thread (g, a, jl): anchor-in-cell a, joint j = j0 + jl; the G
// thread groups take the cells p = g, g+G, ... (a single group walked all fh*fw cells with two dependent loads each: 60 us
// of pure latency per launch).
// Pass 1: per-joint max of the logits (exact, order independent).
// Pass 2: e = exp(x - max); partial sums of e, e*(anchor+offset), e*depth per (group, anchor, joint), then
// combined through LDS in a fixed order (group-major, then anchor): bitwise reproducible, and independent of the joint split.
// crop-(u,v,d) of one joint -> image (u,v,d) and / or camera xyz in millimetres (a2j/a2j.py:17-34 convert_joints +
// datasets3d/a2jdataset.py:31-38 uvd2xyz).  ONE function for the stand-alone kernel and the aggregation's epilogue: the two are
// bit-identical by construction.  (Every operation is a single rounding in the reference's order; nothing here can contract
// into an fma: each product feeds a division.)
struct ConvertSpec {
  const long long* box;   // [n,4] padded crop boxes (x1,y1,x2,y2), or nullptr: no conversion (unless box_f32)
  const float* box_f32;   // [n,4] the DATASET's float boxes (the evaluation caller, a2j/a2j.py:339-346; a2jdataset.py:293): used instead
  const float* sample_paras;   // [n,4] per-sample (fx, fy, cx, cy) on the device (a2jdataset.py:279,293): used instead of fx..cy
  float crop_w, crop_h;
  int has_paras;          // camera intrinsics given: xyz_mm is written
  float fx, fy, cx, cy;
  int clamp_kp;           // clamp the crop-(u,v,d) to [0, crop_w] first (ros_demo.py:283 clamps all three to [0, 176])
  int clamp_h, clamp_w;   // > 0: clamp box x1,y1 to [0, clamp_h] and x2,y2 to [0, clamp_w] first (ros_demo.py:279-280, as written there)
  float* image_uvd;       // [n,J,3] or nullptr
  float* xyz_mm;          // [n,J,3] or nullptr
};

__device__ __forceinline__ void convert_one(const ConvertSpec& cs, int img, long joint_row, float ku, float kv, float kd) {
  float x0, y0, x1, y1;
  if (cs.box_f32) {
    x0 = cs.box_f32[img * 4 + 0], y0 = cs.box_f32[img * 4 + 1], x1 = cs.box_f32[img * 4 + 2], y1 = cs.box_f32[img * 4 + 3];
  } else {
    x0 = (float)cs.box[img * 4 + 0], y0 = (float)cs.box[img * 4 + 1], x1 = (float)cs.box[img * 4 + 2], y1 = (float)cs.box[img * 4 + 3];
  }
  if (cs.clamp_h > 0) {
    x0 = fminf(fmaxf(x0, 0.f), (float)cs.clamp_h);
    y0 = fminf(fmaxf(y0, 0.f), (float)cs.clamp_h);
    x1 = fminf(fmaxf(x1, 0.f), (float)cs.clamp_w);
    y1 = fminf(fmaxf(y1, 0.f), (float)cs.clamp_w);
  }
  if (cs.clamp_kp) {   // torch.clamp keeps NaN; fminf / fmaxf would drop it
    ku = ku != ku ? ku : fminf(fmaxf(ku, 0.f), cs.crop_w);
    kv = kv != kv ? kv : fminf(fmaxf(kv, 0.f), cs.crop_w);
    kd = kd != kd ? kd : fminf(fmaxf(kd, 0.f), cs.crop_w);
  }
  const float u = ku * (x1 - x0) / cs.crop_w + x0;
  const float v = kv * (y1 - y0) / cs.crop_h + y0;
  if (cs.image_uvd) {
    float* o = cs.image_uvd + joint_row * 3;
    o[0] = u;
    o[1] = v;
    o[2] = kd;
  }
  if (cs.xyz_mm && cs.has_paras) {
    float* o = cs.xyz_mm + joint_row * 3;
    float fx = cs.fx, fy = cs.fy, cx = cs.cx, cy = cs.cy;
    if (cs.sample_paras) fx = cs.sample_paras[img * 4 + 0], fy = cs.sample_paras[img * 4 + 1], cx = cs.sample_paras[img * 4 + 2], cy = cs.sample_paras[img * 4 + 3];
    o[0] = (u - cx) * kd / fx * 1000.f;
    o[1] = (v - cy) * kd / fy * 1000.f;
    o[2] = kd * 1000.f;
  }
}

constexpr int kAnchorsPerCell = 16;
constexpr int kMaxGroups = 9;   // (round 4: 3 -> 9 cell groups: on an 11 x 11 map a thread owns 14 cells = ONE batch of loads per pass
                                // instead of three, 18 -> ~11 us at one crop; the group count fixes the summation order, so it is the
                                // same for every batch size)
constexpr int kJointSplit = 3;

__global__ __launch_bounds__(1024) void a2j_aggregate_kernel(const float* __restrict__ cls,
                                                             const float* __restrict__ reg,
                                                             const float* __restrict__ dep,
                                                             const int* __restrict__ valid, int fh, int fw,
                                                             int J, int Jw, int stride, int G, float* __restrict__ out,
                                                             const ConvertSpec cs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [G][4][A*Jw]
  const int k = blockIdx.x;
  const int j0 = blockIdx.y * Jw;
  const int jn = min(Jw, J - j0);      // joints of this workgroup (the last one may have fewer, or none)
  if (jn <= 0) return;
  const int AJ = kAnchorsPerCell * J;   // channels per cell in memory
  const int AW = kAnchorsPerCell * Jw;  // thread slots per cell group
  const int g = threadIdx.x / AW, cl = threadIdx.x - g * AW;
  const int a = cl / Jw, jl = cl - a * Jw;
  const bool active = g < G && jl < jn;
  const int c = a * J + j0 + jl;        // channel in memory
  if (valid && valid[k] != 1) {  // uniform per workgroup: 0 = no crop (zero row), 2 = non-finite crop (NaN row, as the
    // reference's network returns for it: every cell of the 11 x 11 maps sees every pixel of the crop)
    if ((int)threadIdx.x < jn * 3) {
      const float fill = valid[k] == 0 ? 0.f : __builtin_nanf("");
      const long at = ((long)k * J + j0) * 3 + threadIdx.x;
      out[at] = fill;
      if ((cs.box || cs.box_f32) && cs.image_uvd) cs.image_uvd[at] = fill;
      if ((cs.box || cs.box_f32) && cs.xyz_mm && cs.has_paras) cs.xyz_mm[at] = fill;
    }
    return;
  }
  const int cells = fh * fw;
  const float* clsk = cls + (long)k * cells * AJ;
  const float* depk = dep + (long)k * cells * AJ;
  const float* regk = reg + (long)k * cells * AJ * 2;

  // Both cell loops fetch kBatch (14: three rounds for the 41 cells a thread owns on an 11 x 11 map) cells' operands before they use any (same operations in the same order): written as
  // load -> use per cell, each of the 2 x 41 iterations waited for its own L2 round trip -- 27 us per launch.
  constexpr int kBatch = 14;
  float mx = -FLT_MAX;
  if (active)
    for (int pb = g; pb < cells; pb += kBatch * G) {
      float v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int p = pb + u * G;
        v[u] = p < cells ? clsk[(long)p * AJ + c] : -FLT_MAX;
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) mx = fmaxf(mx, v[u]);
    }
  if (g < G) lds[g * AW + cl] = mx;
  __syncthreads();
  if ((int)threadIdx.x < AW) {   // (group 0's threads: slot cl = threadIdx.x) maximum over the groups, then over the anchors below
    float m = lds[threadIdx.x];
    for (int gg = 1; gg < G; ++gg) m = fmaxf(m, lds[gg * AW + threadIdx.x]);
    lds[threadIdx.x] = m;
  }
  __syncthreads();
  float mj = -FLT_MAX;
  if (active)
    for (int aa = 0; aa < kAnchorsPerCell; ++aa) mj = fmaxf(mj, lds[aa * Jw + jl]);
  __syncthreads();

  if (active) {
    float s = 0.f, s0 = 0.f, s1 = 0.f, sd = 0.f;
    const float p0 = 2.f + 4.f * (float)(a >> 2), p1 = 2.f + 4.f * (float)(a & 3);
    for (int pb = g; pb < cells; pb += kBatch * G) {
      float vc[kBatch], vd[kBatch];
      float2 vr[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int p = pb + u * G;
        const long q = (long)(p < cells ? p : g) * AJ + c;   // (clamped: the tail batch's extra cells are not used)
        vc[u] = clsk[q];
        vr[u] = *reinterpret_cast<const float2*>(regk + q * 2);
        vd[u] = depk[q];
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int p = pb + u * G;
        if (p < cells) {
          const int hh = p / fw, ww = p - hh * fw;
          const float e = expf(vc[u] - mj);
          const float a0 = (float)(hh * stride) + p0, a1 = (float)(ww * stride) + p1;
          s += e;
          s0 += e * (a0 + vr[u].x);
          s1 += e * (a1 + vr[u].y);
          sd += e * vd[u];
        }
      }
    }
    float* o = lds + (long)g * 4 * AW;
    o[cl] = s;
    o[AW + cl] = s0;
    o[2 * AW + cl] = s1;
    o[3 * AW + cl] = sd;
  }
  __syncthreads();
  // fixed-order combination in two steps: thread (quantity q, anchor a, joint jl) of the first 4 * AW sums its slot over the
  // groups 0 .. G-1, then thread jj sums the 16 anchors of its joint (G + 16 dependent additions instead of 16 G)
  for (int s4 = threadIdx.x; s4 < 4 * AW; s4 += blockDim.x) {
    float acc = lds[s4];
    for (int gg = 1; gg < G; ++gg) acc += lds[(long)gg * 4 * AW + s4];
    lds[s4] = acc;   // (group 0's slot: only this thread reads or writes it here)
  }
  __syncthreads();
  if ((int)threadIdx.x < jn) {
    const int jj = threadIdx.x;
    float t = 0.f, t0 = 0.f, t1 = 0.f, td = 0.f;
    for (int aa = 0; aa < kAnchorsPerCell; ++aa) {
      t += lds[aa * Jw + jj];
      t0 += lds[AW + aa * Jw + jj];
      t1 += lds[2 * AW + aa * Jw + jj];
      td += lds[3 * AW + aa * Jw + jj];
    }
    float* o = out + ((long)k * J + j0 + jj) * 3;
    const float ku = t0 / t, kv = t1 / t, kd = td / t;
    o[0] = ku;
    o[1] = kv;
    o[2] = kd;
    // convert_joints + uvd2xyz in the aggregation's epilogue (what every caller does next: ros_demo.py:289,
    // 329-330, a2j/a2j.py:341-348) -- from the registers that hold the joint, no second launch
    if (cs.box || cs.box_f32) convert_one(cs, k, (long)k * J + j0 + jj, ku, kv, kd);
  }
}

// the stand-alone form (the reference form of the fused epilogue above), one thread per joint: out = camera xyz in mm when
// intrinsics are given, else image (u,v,d)
__global__ __launch_bounds__(256) void convert_joints_kernel(const float* __restrict__ kp, const int* __restrict__ valid, int n,
                                                             int J, const ConvertSpec cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * J) return;
  const int img = i / J;
  if (valid && valid[img] == 0) {
    if (cs.image_uvd) cs.image_uvd[(long)i * 3 + 0] = cs.image_uvd[(long)i * 3 + 1] = cs.image_uvd[(long)i * 3 + 2] = 0.f;
    if (cs.xyz_mm && cs.has_paras) cs.xyz_mm[(long)i * 3 + 0] = cs.xyz_mm[(long)i * 3 + 1] = cs.xyz_mm[(long)i * 3 + 2] = 0.f;
    return;
  }
  convert_one(cs, img, i, kp[(long)i * 3 + 0], kp[(long)i * 3 + 1], kp[(long)i * 3 + 2]);
}

}  // namespace

extern "C" int hn_convert_joints_f32(const float* kp, const int64_t* crop_box, const int32_t* valid, int n, int joints,
                                     float crop_w, float crop_h, const float* paras, float* out, void* stream) {
  HN_CHECK_ARG(kp && crop_box && out, "hn_convert_joints_f32: null pointer");
  HN_CHECK_ARG(n >= 0 && joints > 0 && crop_w > 0.f && crop_h > 0.f, "bad dims");
  if (n == 0) return HN_OK;
  const int total = n * joints;
  ConvertSpec cs{};
  cs.box = (const long long*)crop_box;
  cs.crop_w = crop_w;
  cs.crop_h = crop_h;
  cs.has_paras = paras ? 1 : 0;
  cs.fx = paras ? paras[0] : 1.f;
  cs.fy = paras ? paras[1] : 1.f;
  cs.cx = paras ? paras[2] : 0.f;
  cs.cy = paras ? paras[3] : 0.f;
  cs.image_uvd = paras ? nullptr : out;
  cs.xyz_mm = paras ? out : nullptr;
  hipLaunchKernelGGL(convert_joints_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, kp, valid, n, joints, cs);
  HN_CHECK_LAUNCH("convert_joints_kernel");
  return HN_OK;
}

extern "C" int hn_convert_joints_samples_f32(const float* kp, const float* box_f32, const float* sample_paras, const int32_t* valid,
                                             int n, int joints, float crop_w, float crop_h, float* out_image_uvd, float* out_xyz_mm,
                                             void* stream) {
  HN_CHECK_ARG(kp && box_f32, "hn_convert_joints_samples_f32: null pointer");
  HN_CHECK_ARG(out_image_uvd || out_xyz_mm, "hn_convert_joints_samples_f32: no output requested");
  HN_CHECK_ARG(!out_xyz_mm || sample_paras, "hn_convert_joints_samples_f32: camera xyz needs the samples' intrinsics");
  HN_CHECK_ARG(n >= 0 && joints > 0 && crop_w > 0.f && crop_h > 0.f, "bad dims");
  if (n == 0) return HN_OK;
  const int total = n * joints;
  ConvertSpec cs{};
  cs.box_f32 = box_f32;
  cs.sample_paras = sample_paras;
  cs.crop_w = crop_w;
  cs.crop_h = crop_h;
  cs.has_paras = sample_paras ? 1 : 0;
  cs.fx = cs.fy = 1.f;
  cs.image_uvd = out_image_uvd;
  cs.xyz_mm = out_xyz_mm;
  hipLaunchKernelGGL(convert_joints_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, kp, valid, n, joints, cs);
  HN_CHECK_LAUNCH("convert_joints_kernel");
  return HN_OK;
}

extern "C" int hn_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n, int h, int w, int c, int oh, int ow,
                                        void* stream) {
  HN_CHECK_ARG(x && y, "hn_maxpool3x3s2_nhwc_f32: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "bad dims (c must be a multiple of 4)");
  HN_CHECK_ARG(oh == (h + 2 - 3) / 2 + 1 && ow == (w + 2 - 3) / 2 + 1, "output size mismatch");
  const long total = (long)n * oh * ow * (c / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n, h, w, c / 4, oh, ow);
  HN_CHECK_LAUNCH("maxpool3x3s2_kernel");
  return HN_OK;
}

extern "C" int hn_pack_depth_nhwc(const float* src, float* dst, int n, int hw, int cpad, void* stream) {
  HN_CHECK_ARG(src && dst, "hn_pack_depth_nhwc: null pointer");
  HN_CHECK_ARG(n > 0 && hw > 0 && cpad >= 4 && cpad % 4 == 0, "bad dims");
  const long npix = (long)n * hw;
  const long total = npix * (cpad / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_depth_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, npix, cpad / 4);
  HN_CHECK_LAUNCH("pack_depth_kernel");
  return HN_OK;
}

static int launch_aggregate(const float* cls, const float* reg, const float* dep, const int32_t* valid, int k, int fh, int fw,
                            int joints, int stride, float* out, const ConvertSpec& cs, void* stream) {
  // the cell-group count depends on the joint count only (it fixes the summation order): min(9, 1024 / (16 * joints per workgroup))
  const int split = joints >= 2 * kJointSplit ? kJointSplit : 1;
  const int jw = (joints + split - 1) / split;
  const int aw = kAnchorsPerCell * jw;
  int groups = 1024 / aw;            // cell groups of a workgroup (16 * jw thread slots each)
  groups = groups > kMaxGroups ? kMaxGroups : groups;
  groups = groups < 1 ? 1 : groups;
  const int threads = ((groups * aw + 63) / 64) * 64;
  hipLaunchKernelGGL(a2j_aggregate_kernel, dim3(k, split), dim3(threads), groups * 4 * aw * sizeof(float),
                     (hipStream_t)stream, cls, reg, dep, valid, fh, fw, joints, jw, stride, groups, out, cs);
  HN_CHECK_LAUNCH("a2j_aggregate_kernel");
  return HN_OK;
}

extern "C" int hn_a2j_aggregate_f32(const float* cls, const float* reg, const float* dep, const int32_t* valid,
                                    int k, int fh, int fw, int joints, int stride, float* out, void* stream) {
  HN_CHECK_ARG(cls && reg && dep && out, "hn_a2j_aggregate_f32: null pointer");
  HN_CHECK_ARG(k >= 0 && fh > 0 && fw > 0 && stride > 0, "bad dims");
  HN_CHECK_ARG(joints > 0 && joints * kAnchorsPerCell <= 1024, "joints must be in [1, 64]");
  if (k == 0) return HN_OK;
  return launch_aggregate(cls, reg, dep, valid, k, fh, fw, joints, stride, out, ConvertSpec{}, stream);
}

extern "C" int hn_a2j_aggregate_convert_f32(const float* cls, const float* reg, const float* dep, const int32_t* valid, int k,
                                            int fh, int fw, int joints, int stride, const int64_t* crop_box, float crop_w,
                                            float crop_h, const float* paras, const hn_convert_opts* opts, float* out_uvd,
                                            float* out_image_uvd, float* out_xyz_mm, void* stream) {
  HN_CHECK_ARG(cls && reg && dep && out_uvd, "hn_a2j_aggregate_convert_f32: null pointer");
  HN_CHECK_ARG(crop_box || (opts && opts->sample_box), "hn_a2j_aggregate_convert_f32: no boxes (crop_box or opts->sample_box)");
  HN_CHECK_ARG(out_image_uvd || out_xyz_mm, "hn_a2j_aggregate_convert_f32: no converted output requested");
  HN_CHECK_ARG(!out_xyz_mm || paras || (opts && opts->sample_paras),
               "hn_a2j_aggregate_convert_f32: camera xyz needs the intrinsics (fx, fy, cx, cy)");
  HN_CHECK_ARG(k >= 0 && fh > 0 && fw > 0 && stride > 0 && crop_w > 0.f && crop_h > 0.f, "bad dims");
  HN_CHECK_ARG(joints > 0 && joints * kAnchorsPerCell <= 1024, "joints must be in [1, 64]");
  if (k == 0) return HN_OK;
  ConvertSpec cs{};
  cs.box = (const long long*)crop_box;
  cs.crop_w = crop_w;
  cs.crop_h = crop_h;
  cs.box_f32 = opts ? opts->sample_box : nullptr;
  cs.sample_paras = opts ? opts->sample_paras : nullptr;
  cs.has_paras = (paras || cs.sample_paras) ? 1 : 0;
  cs.fx = paras ? paras[0] : 1.f;
  cs.fy = paras ? paras[1] : 1.f;
  cs.cx = paras ? paras[2] : 0.f;
  cs.cy = paras ? paras[3] : 0.f;
  cs.clamp_kp = opts ? opts->clamp_keypoints : 0;
  cs.clamp_h = opts ? opts->clamp_box_h : 0;
  cs.clamp_w = opts ? opts->clamp_box_w : 0;
  HN_CHECK_ARG((cs.clamp_h > 0) == (cs.clamp_w > 0), "clamp_box_h and clamp_box_w are given together");
  cs.image_uvd = out_image_uvd;
  cs.xyz_mm = out_xyz_mm;
  return launch_aggregate(cls, reg, dep, valid, k, fh, fw, joints, stride, out_uvd, cs, stream);
}

// ---------------------------------------------------------------------------------------
// The lifter's input (include/handnet_hip.h): per frame and axis (x - mean) / std over the J joints, population std.
// One wave per frame; sums in fp64 in joint order (numpy's pairwise fp32 / fp64 summation differs from ANY fixed order by
// rounding only; the tests compare with a function-by-function fp64 restatement of the caller's chain).
// ---------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void joints2d_standardize_kernel(const float* __restrict__ uvd, const int* __restrict__ valid,
                                                                  int J, float* __restrict__ out) {
  const int img = blockIdx.x, lane = threadIdx.x;
  float* o = out + (long)img * J * 2;
  if (valid && valid[img] != 1) {
    for (int i = lane; i < J * 2; i += 64) o[i] = 0.f;
    return;
  }
  const float* p = uvd + (long)img * J * 3;
  // every lane computes both axes' statistics itself (J <= 64 values each: 2 x 21 loads that hit the same cache line)
  double m0 = 0.0, m1 = 0.0;
  for (int j = 0; j < J; ++j) {
    m0 += (double)p[j * 3 + 0];
    m1 += (double)p[j * 3 + 1];
  }
  m0 /= J;
  m1 /= J;
  double v0 = 0.0, v1 = 0.0;
  for (int j = 0; j < J; ++j) {
    const double a = (double)p[j * 3 + 0] - m0, b = (double)p[j * 3 + 1] - m1;
    v0 += a * a;
    v1 += b * b;
  }
  const double s0 = sqrt(v0 / J), s1 = sqrt(v1 / J);
  for (int j = lane; j < J; j += 64) {
    o[j * 2 + 0] = (float)(((double)p[j * 3 + 0] - m0) / s0);
    o[j * 2 + 1] = (float)(((double)p[j * 3 + 1] - m1) / s1);
  }
}
}  // namespace

extern "C" int hn_joints2d_standardize_f32(const float* image_uvd, const int32_t* valid, int n, int joints, float* out,
                                           void* stream) {
  HN_CHECK_ARG(image_uvd && out, "hn_joints2d_standardize_f32: null pointer");
  HN_CHECK_ARG(n >= 0 && joints > 1, "bad dims");
  if (n == 0) return HN_OK;
  hipLaunchKernelGGL(joints2d_standardize_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, image_uvd, valid, joints, out);
  HN_CHECK_LAUNCH("joints2d_standardize_kernel");
  return HN_OK;
}

// ---------------------------------------------------------------------------------------
// Per-frame result records of the N > 1 all-gather (hn_amd/dist.py) and the always-on non-finite check
// ---------------------------------------------------------------------------------------
namespace {

constexpr int kRecHead = 40;  // 4 x int64 box, int32 has_hand, int32 row flag

// one thread per 4-byte word of a record
__global__ __launch_bounds__(256) void pack_records_kernel(const float* __restrict__ kp, const long long* __restrict__ box,
                                                           const int* __restrict__ has, int n, int rows, int j3,
                                                           int rec_words, unsigned* __restrict__ rec,
                                                           const float* __restrict__ e0, const float* __restrict__ e1) {
  const long total = (long)rows * rec_words;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / rec_words), w = (int)(i - (long)r * rec_words);
    unsigned v = 0u;
    if (r < n) {
      if (w < 8) {
        const long long b = box[(long)r * 4 + (w >> 1)];
        v = (unsigned)((unsigned long long)b >> ((w & 1) * 32));
      } else if (w == 8) {
        v = (unsigned)has[r];
      } else if (w == 9) {
        v = 1u;
      } else if (w - 10 < j3) {
        v = __float_as_uint(kp[(long)r * j3 + (w - 10)]);
      } else if (e0 && w - 10 < 2 * j3) {           // wide records: image (u,v,d) behind the crop (u,v,d) ...
        v = __float_as_uint(e0[(long)r * j3 + (w - 10 - j3)]);
      } else if (e1 && w - 10 < 3 * j3) {           // ... then camera xyz in mm
        v = __float_as_uint(e1[(long)r * j3 + (w - 10 - 2 * j3)]);
      }
    }
    rec[i] = v;
  }
}

__global__ __launch_bounds__(256) void unpack_records_kernel(const unsigned* __restrict__ rec, int rows, int j3, int rec_words,
                                                             float* __restrict__ kp, long long* __restrict__ box,
                                                             int* __restrict__ has, int* __restrict__ valid) {
  const long total = (long)rows * rec_words;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / rec_words), w = (int)(i - (long)r * rec_words);
    const unsigned v = rec[i];
    if (w < 8) {
      if ((w & 1) == 0) {
        const unsigned hi = rec[i + 1];
        box[(long)r * 4 + (w >> 1)] = (long long)(((unsigned long long)hi << 32) | v);
      }
    } else if (w == 8) {
      has[r] = (int)v;
    } else if (w == 9) {
      valid[r] = (int)v;
    } else if (w - 10 < j3) {
      kp[(long)r * j3 + (w - 10)] = __uint_as_float(v);
    }
  }
}

__global__ __launch_bounds__(256) void nonfinite_count_kernel(const float* __restrict__ x, long count, int* __restrict__ flag) {
  int bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    bad += !(fabsf(x[i]) <= 3.402823466e38f) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(flag, bad);
}

}  // namespace

extern "C" int hn_pack_records_ex(const float* keypoints, const int64_t* crop_box, const int32_t* has_hand, int n, int rows,
                                  int joints, int rec_bytes, const float* extra0, const float* extra1, void* records,
                                  void* stream) {
  HN_CHECK_ARG(records && (n == 0 || (keypoints && crop_box && has_hand)), "hn_pack_records: null pointer");
  HN_CHECK_ARG(n >= 0 && rows >= n && joints > 0, "bad dims");
  HN_CHECK_ARG(extra0 || !extra1, "hn_pack_records_ex: extra1 without extra0");
  const int fields = 1 + (extra0 ? 1 : 0) + (extra1 ? 1 : 0);
  HN_CHECK_ARG(rec_bytes >= kRecHead + 12 * joints * fields && rec_bytes % 8 == 0,
               "rec_bytes must be a multiple of 8 >= 40 + 12*joints per keypoint field");
  if (rows == 0) return HN_OK;
  const long total = (long)rows * (rec_bytes / 4);
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_records_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, keypoints, (const long long*)crop_box,
                     has_hand, n, rows, joints * 3, rec_bytes / 4, (unsigned*)records, extra0, extra1);
  HN_CHECK_LAUNCH("pack_records_kernel");
  return HN_OK;
}

extern "C" int hn_pack_records(const float* keypoints, const int64_t* crop_box, const int32_t* has_hand, int n, int rows,
                               int joints, int rec_bytes, void* records, void* stream) {
  return hn_pack_records_ex(keypoints, crop_box, has_hand, n, rows, joints, rec_bytes, nullptr, nullptr, records, stream);
}

extern "C" int hn_unpack_records(const void* records, int rows, int joints, int rec_bytes, float* keypoints,
                                 int64_t* crop_box, int32_t* has_hand, int32_t* valid, void* stream) {
  HN_CHECK_ARG(records && keypoints && crop_box && has_hand && valid, "hn_unpack_records: null pointer");
  HN_CHECK_ARG(rows >= 0 && joints > 0, "bad dims");
  HN_CHECK_ARG(rec_bytes >= kRecHead + 12 * joints && rec_bytes % 8 == 0, "rec_bytes must be a multiple of 8 >= 40 + 12*joints");
  if (rows == 0) return HN_OK;
  const long total = (long)rows * (rec_bytes / 4);
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(unpack_records_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned*)records, rows,
                     joints * 3, rec_bytes / 4, keypoints, (long long*)crop_box, has_hand, valid);
  HN_CHECK_LAUNCH("unpack_records_kernel");
  return HN_OK;
}

extern "C" int hn_nonfinite_count_f32(const float* x, int64_t count, int32_t* flag, void* stream) {
  HN_CHECK_ARG(x && flag && count >= 0, "hn_nonfinite_count_f32: bad arguments");
  if (count == 0) return HN_OK;
  const int grid = (int)((count + 255) / 256 < 256 ? (count + 255) / 256 : 256);
  hipLaunchKernelGGL(nonfinite_count_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long)count, flag);
  HN_CHECK_LAUNCH("nonfinite_count_kernel");
  return HN_OK;
}

// ---------------------------------------------------------------------------------------
// fp32 NHWC(4) image -> the stem image of hn_conv_stem_f16x3 / hn_conv_stem_pool_f16x3: two fp16 planes (hi, lo) of
// [n][h + 2b][w + 2b][4] with a zero border of b pixels (the A2J crops; the FCOS image is written in this form by the
// preprocess kernel).  One thread per bordered pixel: 16 B read, 8 B + 8 B written.
// ---------------------------------------------------------------------------------------
namespace {
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stem_image_kernel(const float* __restrict__ x, _Float16* __restrict__ dst, int n, int h,
                                                         int w, int b, int* range_flag, int* __restrict__ valid) {
  const int hb = h + 2 * b, wb = w + 2 * b;
  const long total = (long)n * hb * wb;
  _Float16* lo_plane = dst + total * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % wb);
    const long t = i / wb;
    const int yy = (int)(t % hb);
    const int img = (int)(t / hb);
    f16x4s hi = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0}, lo = hi;
    const int sy = yy - b, sx = xx - b;
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)img * h + sy) * w + sx) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (range_flag) hn::range_note_input(range_flag, v[e]);
        float ve = v[e];
        // a non-finite pixel: mark the image (every writer stores the same value; images with valid == 0 stay 0) and keep
        // the value OUT of the network -- the aggregation writes the NaN row of a marked image whatever the convolutions
        // compute for it, and with no NaN of the caller's inside, a NaN activation always means a defect (range contract)
        if (valid && !(fabsf(ve) <= 3.402823466e38f)) {
          if (valid[img] == 1) valid[img] = 2;
          ve = 0.f;
        }
        const _Float16 hh = (_Float16)ve;
        hi[e] = hh;
        lo[e] = (_Float16)(ve - (float)hh);
      }
    }
    *reinterpret_cast<f16x4s*>(dst + i * 4) = hi;
    *reinterpret_cast<f16x4s*>(lo_plane + i * 4) = lo;
  }
}
}  // namespace

extern "C" int hn_stem_image_nhwc4(const float* x, int n, int h, int w, int border, void* dst16, void* stream) {
  return hn_stem_image_nhwc4_valid(x, n, h, w, border, dst16, nullptr, stream);
}

extern "C" int hn_stem_image_nhwc4_valid(const float* x, int n, int h, int w, int border, void* dst16, int32_t* valid,
                                         void* stream) {
  HN_CHECK_ARG(x && dst16, "hn_stem_image_nhwc4: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && border >= 0 && border <= 4, "bad dims");
  HN_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)dst16 % 8 == 0, "unaligned tensors");
  const long total = (long)n * (h + 2 * border) * (w + 2 * border);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(stem_image_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)dst16, n, h, w, border,
                     hn::range_flag_ptr(), valid);
  HN_CHECK_LAUNCH("stem_image_kernel");
  return HN_OK;
}
