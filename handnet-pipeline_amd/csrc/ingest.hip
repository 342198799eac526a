// Caller-side ingest of the reference's live loop, fused into one kernel:
//   ros_demo.py:230-231   16UC1 depth (millimetres)  -> float32 / 1000.0          (32FC1 depth is passed through)
//   ros_demo.py:266       bgr8 HWC uint8 -> cv2.COLOR_BGR2RGB -> transpose(2,0,1) -> float32 / 255.0
//   ros_demo.py:267-269   depth -> [1,1,H,W]; RGB-D model: cat([rgb, depth], dim=1)
// Reads 3 + 2 bytes per pixel (from device memory or straight from PINNED HOST memory: the pointers only have to be
// readable by the device) and writes the fp32 planar RGB tensor the preprocess kernel takes and the metres depth map the
// crop kernel takes: 1.5 MB per 640x480 frame cross PCIe instead of the 4.9 MB of the fp32 feed.  Both divisions are IEEE
// divisions of exactly representable integers by 255.0f / 1000.0f: bit-identical to numpy's float32 arithmetic.
// HBM-bound (17 B per pixel); four pixels per lane, 16-byte stores.
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float u8_to_unit(unsigned v) { return __fdiv_rn((float)v, 255.0f); }

// depth_kind: 0 none, 1 uint16 millimetres, 2 float32 metres
template <bool VEC4>
__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ bgr, const void* __restrict__ depth, int depth_kind,
                                                     float* __restrict__ rgb, float* __restrict__ depth_m,
                                                     float* __restrict__ rgbd, int n, long hw) {
  constexpr int PX = VEC4 ? 4 : 1;
  const long groups = hw / PX;   // (VEC4: hw % 4 == 0 and every base pointer is 16-byte aligned; checked on the host)
  const long total = (long)n * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long img = i / groups;
    const long p = (i - img * groups) * PX;   // first pixel of this lane within the image
    float r[PX], g[PX], b[PX], d[PX];
    if (VEC4) {
      // 12 bytes = pixels p .. p+3 as three dwords: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
      const unsigned* src = reinterpret_cast<const unsigned*>(bgr + (img * hw + p) * 3);
      const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
      b[0] = u8_to_unit(w0 & 255u);         g[0] = u8_to_unit((w0 >> 8) & 255u);  r[0] = u8_to_unit((w0 >> 16) & 255u);
      b[1] = u8_to_unit(w0 >> 24);          g[1] = u8_to_unit(w1 & 255u);         r[1] = u8_to_unit((w1 >> 8) & 255u);
      b[2] = u8_to_unit((w1 >> 16) & 255u); g[2] = u8_to_unit(w1 >> 24);          r[2] = u8_to_unit(w2 & 255u);
      b[3] = u8_to_unit((w2 >> 8) & 255u);  g[3] = u8_to_unit((w2 >> 16) & 255u); r[3] = u8_to_unit(w2 >> 24);
    } else {
      const uint8_t* src = bgr + (img * hw + p) * 3;
      b[0] = u8_to_unit(src[0]);
      g[0] = u8_to_unit(src[1]);
      r[0] = u8_to_unit(src[2]);
    }
    if (depth_kind == 1) {
      const uint16_t* ds = reinterpret_cast<const uint16_t*>(depth) + img * hw + p;
      if (VEC4) {
        const unsigned long long q = *reinterpret_cast<const unsigned long long*>(ds);
#pragma unroll
        for (int e = 0; e < PX; ++e) d[e] = __fdiv_rn((float)(unsigned)((q >> (16 * e)) & 0xFFFFull), 1000.0f);
      } else {
        d[0] = __fdiv_rn((float)ds[0], 1000.0f);
      }
    } else if (depth_kind == 2) {
      const float* ds = reinterpret_cast<const float*>(depth) + img * hw + p;
      if (VEC4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(ds);
#pragma unroll
        for (int e = 0; e < PX; ++e) d[e] = q[e];
      } else {
        d[0] = ds[0];
      }
    }
    float* o = rgb ? rgb + img * 3 * hw + p : nullptr;
    float* o4 = rgbd ? rgbd + img * 4 * hw + p : nullptr;
    if (VEC4) {
      const f32x4 vr = {r[0], r[1], r[2], r[3]}, vg = {g[0], g[1], g[2], g[3]}, vb = {b[0], b[1], b[2], b[3]};
      if (o) {
        *reinterpret_cast<f32x4*>(o) = vr;
        *reinterpret_cast<f32x4*>(o + hw) = vg;
        *reinterpret_cast<f32x4*>(o + 2 * hw) = vb;
      }
      if (o4) {
        *reinterpret_cast<f32x4*>(o4) = vr;
        *reinterpret_cast<f32x4*>(o4 + hw) = vg;
        *reinterpret_cast<f32x4*>(o4 + 2 * hw) = vb;
      }
      if (depth_kind) {
        const f32x4 vd = {d[0], d[1], d[2], d[3]};
        if (depth_m) *reinterpret_cast<f32x4*>(depth_m + img * hw + p) = vd;
        if (o4) *reinterpret_cast<f32x4*>(o4 + 3 * hw) = vd;
      }
    } else {
      if (o) {
        o[0] = r[0];
        o[hw] = g[0];
        o[2 * hw] = b[0];
      }
      if (o4) {
        o4[0] = r[0];
        o4[hw] = g[0];
        o4[2 * hw] = b[0];
      }
      if (depth_kind) {
        if (depth_m) depth_m[img * hw + p] = d[0];
        if (o4) o4[3 * hw] = d[0];
      }
    }
  }
}

}  // namespace

extern "C" int hn_ingest_u8bgr_u16mm(const uint8_t* bgr, const void* depth, int depth_kind, float* rgb, float* depth_m,
                                     float* rgbd, int n, int h, int w, void* stream) {
  HN_CHECK_ARG(bgr && (rgb || rgbd), "hn_ingest_u8bgr_u16mm: null image pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0, "hn_ingest_u8bgr_u16mm: bad dims");
  HN_CHECK_ARG(depth_kind >= 0 && depth_kind <= 2, "hn_ingest_u8bgr_u16mm: depth_kind must be 0 (none), 1 (uint16 mm) or 2 (float32 m)");
  HN_CHECK_ARG((depth_kind == 0) == (depth == nullptr), "hn_ingest_u8bgr_u16mm: depth pointer and depth_kind disagree");
  HN_CHECK_ARG(depth_kind == 0 || depth_m || rgbd, "hn_ingest_u8bgr_u16mm: a depth input needs depth_m or rgbd");
  HN_CHECK_ARG(!rgbd || depth_kind != 0, "hn_ingest_u8bgr_u16mm: the RGB-D output needs a depth input");
  const long hw = (long)h * w;
  const bool aligned = (uintptr_t)bgr % 4 == 0 && (uintptr_t)depth % 16 == 0 && (uintptr_t)rgb % 16 == 0 &&
                       (uintptr_t)depth_m % 16 == 0 && (uintptr_t)rgbd % 16 == 0;
  const bool vec = hw % 4 == 0 && aligned;
  const long total = (long)n * (vec ? hw / 4 : hw);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (vec)
    hipLaunchKernelGGL(ingest_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, bgr, depth, depth_kind, rgb, depth_m, rgbd,
                       n, hw);
  else
    hipLaunchKernelGGL(ingest_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, bgr, depth, depth_kind, rgb, depth_m,
                       rgbd, n, hw);
  HN_CHECK_LAUNCH("ingest_kernel");
  return HN_OK;
}
