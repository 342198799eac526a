// FCOS pre/post-processing and the HandNet crop stage (all HBM/latency-bound, integer or
// order-sensitive fp32 work -- built with -ffp-contract=off so every multiply and add
// rounds separately, exactly like the reference's chain of torch ops).
//
//   hn_fcos_preprocess_f32 : torchvision GeneralizedRCNNTransform (fcos_utils/fcos.py:709)
//   hn_fcos_candidates     : fcos_utils/fcos.py:591-628 + det_utils.py:266-294 +
//                            anchor_utils.py:82-132 (anchors are generated in-kernel)
//   hn_fcos_nms / hn_nms   : torchvision.ops.batched_nms / nms (call site fcos.py:635),
//                            resize_boxes (fcos.py:770-783)
//   hn_crop_resize         : handnet_pipeline/handnet_pipeline.py:74-105
#include "hn_common.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// preprocess: normalize -> bilinear (align_corners=False, scale = in/out) -> zero pad
// ---------------------------------------------------------------------------------------
struct Norm3 {
  float mean[3], stdv[3];
};

__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  // ATen's area_pixel_compute_source_index; its builds (AVX2 host code and nvcc device code
  // alike) contract scale*(dst+0.5)-0.5 into one fused multiply-add, so do the same here
  float real = fmaf(scale, (float)dst + 0.5f, -0.5f);
  if (real < 0.f) real = 0.f;
  i0 = min((int)floorf(real), in_size - 1);
  l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
  i1 = min(i0 + 1, in_size - 1);
  l0 = 1.f - l1;
}

// The interpolation itself, in the operation order of ATen's CPU kernel (UpSampleKernel.cpp: horizontal pairs first, each
// `w0 * a + w1 * b` contracted by its build into fma(a, w0, b * w1)): with this order and the fused source index above the
// canvas is BIT-IDENTICAL to F.interpolate(..., mode="bilinear", align_corners=False) of the torch build in this image
// (tests/test_fcos_gpu.py::test_preprocess_matches_transform), so a constant image stays constant and the score ties of
// such a frame are the oracle's ties.  (This file is built with -ffp-contract=off: only the explicit fmaf fuse.)
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, float wx0, float wx1, float wy0,
                                        float wy1) {
  const float r0 = fmaf(v00, wx0, v01 * wx1);
  const float r1 = fmaf(v10, wx0, v11 * wx1);
  return fmaf(r0, wy0, r1 * wy1);
}

// Batches of differently sized images (torchvision batch_images, fcos_utils/fcos.py:702-709): image `img` is
// its own [3][h][w] buffer srcs[img] with geometry geom[img] = {h, w, oh, ow}; every image is resized on its own
// and lands in the top-left corner of the common zero-padded canvas.  Null tables = one dense [n][3][h][w] batch.
struct ImageGeom {
  const float* const* srcs;
  const int* geom;
};

__device__ __forceinline__ const float* image_geometry(const ImageGeom& g, const float* src, int img, int& h, int& w,
                                                       int& oh, int& ow, float& scale_h, float& scale_w) {
  if (!g.geom) return src + (long)img * 3 * h * w;
  const int* q = g.geom + img * 4;
  h = q[0]; w = q[1]; oh = q[2]; ow = q[3];
  scale_h = (float)h / (float)oh;
  scale_w = (float)w / (float)ow;
  return g.srcs[img];
}

__global__ __launch_bounds__(256) void fcos_preprocess_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int n, int h, int w, int oh, int ow, int ph, int pw,
                                                              float scale_h, float scale_w, Norm3 nm, ImageGeom g) {
  const long total = (long)n * ph * pw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % pw);
    const long t = i / pw;
    const int oy = (int)(t % ph);
    const int img = (int)(t / ph);
    const float* base = image_geometry(g, src, img, h, w, oh, ow, scale_h, scale_w);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (oy < oh && ox < ow) {
      int y0, y1, x0, x1;
      float wy0, wy1, wx0, wx1;
      src_index(scale_h, oy, h, y0, y1, wy0, wy1);
      src_index(scale_w, ox, w, x0, x1, wx0, wx1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pl = base + (long)c * h * w;
        const float m = nm.mean[c], s = nm.stdv[c];
        const float v00 = (pl[(long)y0 * w + x0] - m) / s, v01 = (pl[(long)y0 * w + x1] - m) / s;
        const float v10 = (pl[(long)y1 * w + x0] - m) / s, v11 = (pl[(long)y1 * w + x1] - m) / s;
        o[c] = bilerp(v00, v01, v10, v11, wx0, wx1, wy0, wy1);
      }
    }
    *reinterpret_cast<f32x4*>(dst + i * 4) = o;
  }
}

// Same arithmetic, but the result is written as the stem's split image: two fp16 planes (hi, lo) of
// [n][ph + 2b][pw + 2b][4] with a zero border of b pixels (hn_conv_stem_f16x3).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// IDX = unsigned for fewer than 2^31 canvas pixels (three 32-bit divisions per pixel instead of three 64-bit ones: the
// kernel is bound by its index arithmetic and the twelve IEEE divisions of the normalisation, not by its 20 bytes per pixel)
template <typename IDX>
__global__ __launch_bounds__(256) void fcos_preprocess_split_kernel(const float* __restrict__ src,
                                                                    _Float16* __restrict__ dst, int n, int h, int w,
                                                                    int oh, int ow, int ph, int pw, int b,
                                                                    float scale_h, float scale_w, Norm3 nm,
                                                                    int* range_flag, ImageGeom g) {
  const int hb = ph + 2 * b, wb = pw + 2 * b;
  const long total = (long)n * hb * wb;
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < (IDX)total; i += (IDX)gridDim.x * blockDim.x) {
    const IDX t = i / (IDX)wb;
    const int ox = (int)(i - t * (IDX)wb) - b;
    const IDX img_ = t / (IDX)hb;
    const int oy = (int)(t - img_ * (IDX)hb) - b;
    const int img = (int)img_;
    const float* base = image_geometry(g, src, img, h, w, oh, ow, scale_h, scale_w);
    float o[3] = {0.f, 0.f, 0.f};
    if ((unsigned)oy < (unsigned)oh && (unsigned)ox < (unsigned)ow) {
      int y0, y1, x0, x1;
      float wy0, wy1, wx0, wx1;
      src_index(scale_h, oy, h, y0, y1, wy0, wy1);
      src_index(scale_w, ox, w, x0, x1, wx0, wx1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pl = base + (long)c * h * w;
        const float m = nm.mean[c], s = nm.stdv[c];
        const float v00 = (pl[(long)y0 * w + x0] - m) / s, v01 = (pl[(long)y0 * w + x1] - m) / s;
        const float v10 = (pl[(long)y1 * w + x0] - m) / s, v11 = (pl[(long)y1 * w + x1] - m) / s;
        o[c] = bilerp(v00, v01, v10, v11, wx0, wx1, wy0, wy1);
      }
    }
    f16x4 hi, lo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (range_flag) hn::range_note_input(range_flag, o[c]);
      hi[c] = (_Float16)o[c];
      lo[c] = (_Float16)(o[c] - (float)hi[c]);
    }
    hi[3] = lo[3] = (_Float16)0.f;
    *reinterpret_cast<f16x4*>(dst + (long)i * 4) = hi;
    *reinterpret_cast<f16x4*>(dst + (total + (long)i) * 4) = lo;
  }
}

// The dense-batch case as tiles (round 3): the kernel above is VALU-bound -- twelve IEEE divisions per output pixel for a
// normalisation that has one value per SOURCE pixel, and the upsampling 480 -> 800 reads every source value ~2.8 times.  A
// workgroup owns an 8 x 128 tile of the bordered canvas, normalises the source rectangle under it ONCE into LDS
// ((p - mean) / std: the same division on the same value), and interpolates from LDS with the same expressions in the same
// order: bit-identical output, ~3x fewer instructions, the kernel becomes a store stream.
constexpr int kPreTH = 8, kPreTW = 128;   // LDS patch: <8, 96> rows x columns for scales up to ~0.7 (the 480 -> 800 case), <16, 160> beyond

template <int kPreSR, int kPreSC>
__global__ __launch_bounds__(256) void fcos_preprocess_split_tiled_kernel(const float* __restrict__ src, _Float16* __restrict__ dst,
                                                                          int n, int h, int w, int oh, int ow, int ph, int pw,
                                                                          int b, float scale_h, float scale_w, Norm3 nm,
                                                                          int* range_flag) {
  __shared__ float tile[3][kPreSR][kPreSC];
  const int hb = ph + 2 * b, wb = pw + 2 * b;
  const int img = blockIdx.z, cy0 = blockIdx.y * kPreTH, cx0 = blockIdx.x * kPreTW;   // canvas coordinates of the tile
  const int tid = threadIdx.x;
  // output rows / columns of the tile that lie inside the resized image
  const int oy_lo = max(cy0 - b, 0), oy_hi = min(cy0 - b + kPreTH, oh) - 1;
  const int ox_lo = max(cx0 - b, 0), ox_hi = min(cx0 - b + kPreTW, ow) - 1;
  int ys0 = 0, xs0 = 0;
  if (oy_lo <= oy_hi && ox_lo <= ox_hi) {
    int i0, i1, ys1, xs1;
    float l0, l1;
    src_index(scale_h, oy_lo, h, ys0, i1, l0, l1);
    src_index(scale_h, oy_hi, h, i0, ys1, l0, l1);
    src_index(scale_w, ox_lo, w, xs0, i1, l0, l1);
    src_index(scale_w, ox_hi, w, i0, xs1, l0, l1);
    const int rows = ys1 - ys0 + 1, cols = xs1 - xs0 + 1;   // <= kPreSR x kPreSC (the host checks the scales)
    const float* base = src + (long)img * 3 * h * w;
    for (int e = tid; e < 3 * rows * cols; e += 256) {
      const int c = e / (rows * cols), r = e - c * (rows * cols);
      const int yy = r / cols, xx = r - yy * cols;
      tile[c][yy][xx] = (base[(long)c * h * w + (long)(ys0 + yy) * w + (xs0 + xx)] - nm.mean[c]) / nm.stdv[c];
    }
  }
  __syncthreads();
  const long total = (long)n * hb * wb;
  // a lane owns TWO adjacent canvas columns (one 16-byte store per plane and row; wb is even: pw % 32 == 0) of rows
  // (tid >> 6) + 4 k
  const int cxa = cx0 + (tid & 63) * 2;
  int x0[2] = {0, 0}, x1[2] = {0, 0};
  float wx0[2] = {0.f, 0.f}, wx1[2] = {0.f, 0.f};
  bool xin[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ox = cxa + j - b;
    xin[j] = cxa + j < wb && (unsigned)ox < (unsigned)ow;
    if (xin[j]) src_index(scale_w, ox, w, x0[j], x1[j], wx0[j], wx1[j]);
  }
#pragma unroll
  for (int k = 0; k < kPreTH / 4; ++k) {
    const int cy = cy0 + (tid >> 6) + 4 * k, oy = cy - b;
    if (cy >= hb || cxa >= wb) continue;
    const bool yin = (unsigned)oy < (unsigned)oh;
    int y0 = 0, y1 = 0;
    float wy0 = 0.f, wy1 = 0.f;
    if (yin) src_index(scale_h, oy, h, y0, y1, wy0, wy1);
    typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
    f16x8_ hi, lo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float o[3] = {0.f, 0.f, 0.f};
      if (yin && xin[j]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v00 = tile[c][y0 - ys0][x0[j] - xs0], v01 = tile[c][y0 - ys0][x1[j] - xs0];
          const float v10 = tile[c][y1 - ys0][x0[j] - xs0], v11 = tile[c][y1 - ys0][x1[j] - xs0];
          o[c] = bilerp(v00, v01, v10, v11, wx0[j], wx1[j], wy0, wy1);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (range_flag) hn::range_note_input(range_flag, o[c]);
        const _Float16 hh = (_Float16)o[c];
        hi[j * 4 + c] = hh;
        lo[j * 4 + c] = (_Float16)(o[c] - (float)hh);
      }
      hi[j * 4 + 3] = lo[j * 4 + 3] = (_Float16)0.f;
    }
    const long i = ((long)img * hb + cy) * wb + cxa;
    if (cxa + 1 < wb) {
      *reinterpret_cast<f16x8_*>(dst + i * 4) = hi;
      *reinterpret_cast<f16x8_*>(dst + (total + i) * 4) = lo;
    } else {   // (odd canvas width: the last column alone)
      *reinterpret_cast<f16x4*>(dst + i * 4) = f16x4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<f16x4*>(dst + (total + i) * 4) = f16x4{lo[0], lo[1], lo[2], lo[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------
// candidates: score / argmax / threshold / decode / ordered compaction
// ---------------------------------------------------------------------------------------
struct LevelTable {
  int num_levels;
  int h[HN_FCOS_MAX_LEVELS], w[HN_FCOS_MAX_LEVELS], stride[HN_FCOS_MAX_LEVELS];
  int start[HN_FCOS_MAX_LEVELS + 1];  // first point index of each level
  const float* cls_lr[HN_FCOS_MAX_LEVELS];
  const float* reg_ctr[HN_FCOS_MAX_LEVELS];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// One anchor point: score = sqrt(sigmoid(cls) * sigmoid(ctr)), max / argmax over the classes (ties keep the lowest class
// index, torch.max), threshold, and -- for a passing point only -- the hand side and the decoded box
// (fcos.py:591-628, det_utils.py:266-294, anchor_utils.py:82-132).  Shared by the one-workgroup-per-image kernel and the
// chunked pair below: one source, one arithmetic.
struct CandPoint {
  float score, x0, y0, x1, y1;
  int label, side, lvl;
};
__device__ __forceinline__ bool eval_point(const LevelTable& lt, int img, int i, int num_classes, int cw, float thresh,
                                           CandPoint& o) {
  int lv = 0;
  while (lv + 1 < lt.num_levels && i >= lt.start[lv + 1]) ++lv;
  o.lvl = lv;
  const int q = i - lt.start[lv];
  const int hw = lt.h[lv] * lt.w[lv];
  const float* pc = lt.cls_lr[lv] + ((long)img * hw + q) * cw;
  const float* pr = lt.reg_ctr[lv] + ((long)img * hw + q) * 5;
  const float sctr = sigmoidf_(pr[4]);
  float best = -1.f;
  int lab = 0;
  for (int c = 0; c < num_classes; ++c) {
    const float sc = sqrtf(sigmoidf_(pc[c]) * sctr);
    if (sc > best) {  // ties keep the lowest class index (torch.max)
      best = sc;
      lab = c;
    }
  }
  o.score = best;
  o.label = lab;
  o.side = 0;
  o.x0 = o.y0 = o.x1 = o.y1 = 0.f;
  const bool pass = best > thresh;
  if (pass) {
    const float s0 = sigmoidf_(pc[num_classes]), s1 = sigmoidf_(pc[num_classes + 1]);
    o.side = s1 > s0 ? 1 : 0;
    const int gy = q / lt.w[lv], gx = q - gy * lt.w[lv];
    const float st = (float)lt.stride[lv];
    const float half = rintf(st * 0.5f);  // base anchor [-s/2, -s/2, s/2, s/2].round()
    const float ax0 = (float)(gx * lt.stride[lv]) - half, ay0 = (float)(gy * lt.stride[lv]) - half;
    const float ax1 = (float)(gx * lt.stride[lv]) + half, ay1 = (float)(gy * lt.stride[lv]) + half;
    const float cx = 0.5f * (ax0 + ax1), cy = 0.5f * (ay0 + ay1);
    const float bw = ax1 - ax0, bh = ay1 - ay0;
    o.x0 = cx - pr[0] * bw;
    o.y0 = cy - pr[1] * bh;
    o.x1 = cx + pr[2] * bw;
    o.y1 = cy + pr[3] * bh;
  }
  return pass;
}

// One workgroup per image walks the points in anchor order, kU x 1024 at a time: the kU sub-blocks of an iteration are
// evaluated first (their ~10 dependent-latency global loads per point in flight together) and compacted afterwards in
// sub-block order, so the output order is the anchor order.  (kU = 1, the round-1/2 form, spent 63 us at batch 1 on 18
// serial load -> ballot -> barrier rounds; kU = 4 needs 5.)
constexpr int kCandUnroll = 4;
__global__ __launch_bounds__(1024) void fcos_candidates_kernel(const LevelTable lt, int num_classes, float thresh,
                                                               float* __restrict__ cand_boxes,
                                                               float* __restrict__ cand_scores,
                                                               int* __restrict__ cand_labels,
                                                               int* __restrict__ cand_sides,
                                                               int* __restrict__ cand_level,
                                                               int* __restrict__ cand_point,
                                                               int* __restrict__ cand_count, int cap) {
  constexpr int U = kCandUnroll;
  __shared__ int wave_counts[U][16];
  __shared__ int base_s;
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = lt.start[lt.num_levels];
  const int cw = num_classes + 2;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int start = 0; start < P; start += U * (int)blockDim.x) {
    bool pass[U];
    float score[U], bx0[U], by0[U], bx1[U], by1[U];
    int label[U], side[U], lvl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = start + u * (int)blockDim.x + tid;
      pass[u] = false;
      score[u] = bx0[u] = by0[u] = bx1[u] = by1[u] = 0.f;
      label[u] = side[u] = lvl[u] = 0;
      if (i < P) {
        CandPoint cp;
        pass[u] = eval_point(lt, img, i, num_classes, cw, thresh, cp);
        score[u] = cp.score; label[u] = cp.label; side[u] = cp.side; lvl[u] = cp.lvl;
        bx0[u] = cp.x0; by0[u] = cp.y0; bx1[u] = cp.x1; by1[u] = cp.y1;
      }
    }
    int lane_prefix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long bal = __ballot(pass[u]);
      lane_prefix[u] = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_counts[u][wave] = __popcll(bal);
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    int base = base_s;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int wave_off = 0, total = 0;
      for (int k = 0; k < nw; ++k) {
        if (k < wave) wave_off += wave_counts[u][k];
        total += wave_counts[u][k];
      }
      if (pass[u]) {
        const int pos = base + wave_off + lane_prefix[u];
        if (pos < cap) {
          const long o = (long)img * cap + pos;
          cand_boxes[o * 4 + 0] = bx0[u];
          cand_boxes[o * 4 + 1] = by0[u];
          cand_boxes[o * 4 + 2] = bx1[u];
          cand_boxes[o * 4 + 3] = by1[u];
          cand_scores[o] = score[u];
          cand_labels[o] = label[u];
          cand_sides[o] = side[u];
          cand_level[o] = lvl[u];
          if (cand_point) cand_point[o] = start + u * (int)blockDim.x + tid;
        }
      }
      base += total;
    }
    __syncthreads();
    if (tid == 0) base_s = base;
    __syncthreads();
  }
  if (tid == 0) cand_count[img] = min(base_s, cap);
}

// The same compaction with the points of an image spread over ceil(P / 1024) workgroups (hn_fcos_candidates_ws): the single
// workgroup above is five serial rounds of load -> ballot -> barrier per image, 60 us at batch 1 whatever the number of
// candidates, on one CU.  Kernel 1 counts the passing points of each 1024-point chunk into the workspace; kernel 2
// evaluates the chunk again (the head outputs are L2-resident), adds the counts of the chunks before it and writes its
// candidates at their final, anchor-ordered positions.  Two launches, no inter-workgroup waiting, identical output.
__global__ __launch_bounds__(1024) void fcos_candidates_count_kernel(const LevelTable lt, int num_classes, float thresh,
                                                                     int* __restrict__ chunk_counts) {
  __shared__ int wave_counts[16];
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = lt.start[lt.num_levels];
  const int i = chunk * 1024 + tid;
  CandPoint cp;
  const bool pass = i < P && eval_point(lt, img, i, num_classes, num_classes + 2, thresh, cp);
  const unsigned long long bal = __ballot(pass);
  if (lane == 0) wave_counts[wave] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int total = 0;
    for (int k = 0; k < 16; ++k) total += wave_counts[k];
    chunk_counts[img * gridDim.x + chunk] = total;
  }
}

__global__ __launch_bounds__(1024) void fcos_candidates_scatter_kernel(const LevelTable lt, int num_classes, float thresh,
                                                                       const int* __restrict__ chunk_counts,
                                                                       float* __restrict__ cand_boxes,
                                                                       float* __restrict__ cand_scores,
                                                                       int* __restrict__ cand_labels,
                                                                       int* __restrict__ cand_sides,
                                                                       int* __restrict__ cand_level,
                                                                       int* __restrict__ cand_point,
                                                                       int* __restrict__ cand_count, int cap) {
  __shared__ int wave_counts[16];
  __shared__ int base_s;
  const int img = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = lt.start[lt.num_levels];
  const int i = chunk * 1024 + tid;
  CandPoint cp;
  const bool pass = i < P && eval_point(lt, img, i, num_classes, num_classes + 2, thresh, cp);
  const unsigned long long bal = __ballot(pass);
  const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_counts[wave] = __popcll(bal);
  if (wave == 0) {   // candidates of the chunks before this one
    int s = 0;
    for (int c = lane; c < chunk; c += 64) s += chunk_counts[img * chunks + c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) base_s = s;
  }
  __syncthreads();
  int wave_off = 0, total = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < wave) wave_off += wave_counts[k];
    total += wave_counts[k];
  }
  const int base = base_s;
  if (pass) {
    const int pos = base + wave_off + lane_prefix;
    if (pos < cap) {
      const long o = (long)img * cap + pos;
      cand_boxes[o * 4 + 0] = cp.x0;
      cand_boxes[o * 4 + 1] = cp.y0;
      cand_boxes[o * 4 + 2] = cp.x1;
      cand_boxes[o * 4 + 3] = cp.y1;
      cand_scores[o] = cp.score;
      cand_labels[o] = cp.label;
      cand_sides[o] = cp.side;
      cand_level[o] = cp.lvl;
      if (cand_point) cand_point[o] = i;
    }
  }
  if (chunk == chunks - 1 && tid == 0) cand_count[img] = min(base + total, cap);
}

// ext=True outputs of the kept detections (fcos.py:299-320 head maths, :605-607,631-647 gather):
//   ext[l] [n][h][w][8] = relu(hand_dydx_layer)[3] then hand_contact_state_layer[5], raw conv outputs
//   dxdymags = [mag, 0.1 * dx / max(||(dx,dy)||, 1e-12), 0.1 * dy / ...]; contacts = argmax sigmoid (first max)
struct ExtTable {
  int num_levels;
  int hw[HN_FCOS_MAX_LEVELS];
  int start[HN_FCOS_MAX_LEVELS + 1];
  const float* ext[HN_FCOS_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void fcos_ext_gather_kernel(const ExtTable et, const int* __restrict__ det_keep,
                                                               const int* __restrict__ cand_point,
                                                               const int* __restrict__ det_count, int n, int cap,
                                                               int* __restrict__ contacts,
                                                               float* __restrict__ dxdymags) {
  const long total = (long)n * cap;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / cap), d = (int)(i - (long)img * cap);
    if (d >= det_count[img]) continue;
    const int pt = cand_point[(long)img * cap + det_keep[i]];
    int lvl = 0;
    while (lvl + 1 < et.num_levels && pt >= et.start[lvl + 1]) ++lvl;
    const float* e = et.ext[lvl] + ((long)img * et.hw[lvl] + (pt - et.start[lvl])) * 8;
    const float dx = e[1], dy = e[2];
    const float denom = fmaxf(sqrtf(dx * dx + dy * dy), 1e-12f);
    dxdymags[i * 3 + 0] = e[0];
    dxdymags[i * 3 + 1] = 0.1f * (dx / denom);
    dxdymags[i * 3 + 2] = 0.1f * (dy / denom);
    int best = 0;
    float bs = sigmoidf_(e[3]);
    for (int c = 1; c < 5; ++c) {
      const float sc = sigmoidf_(e[3 + c]);
      if (sc > bs) {
        bs = sc;
        best = c;
      }
    }
    contacts[i] = best;
  }
}

// ---------------------------------------------------------------------------------------
// NMS
// ---------------------------------------------------------------------------------------
constexpr int kSortLds = 2048;  // keys sorted in LDS up to this many (padded) entries
constexpr int kBitK = 512;      // ... and resolved by the bitmask form up to this many (its records and rows live where the kept list would)
constexpr int kRankSort = 1024; // ... by rank counting up to this many (one key per thread), by the bitonic network beyond
static_assert(2 * kRankSort <= kSortLds, "the rank sort writes into the upper half of the key array");
constexpr int kKeptRec = 6;     // x1 y1 x2 y2 area label(as float bits)
static_assert(kBitK * kKeptRec * 4 + kBitK * (kBitK / 64) * 8 <= kSortLds * kKeptRec * 4, "records + rows of the bitmask form must fit the kept list's LDS");

__host__ __device__ inline int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
  unsigned u = __float_as_uint(score);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending-sortable
  u = ~u;                                       // descending score
  return ((unsigned long long)u << 32) | (unsigned)idx;
}

struct NmsArgs {
  const float* boxes;    // [n][cap][4]
  const float* scores;   // [n][cap]
  const int* labels;     // [n][cap] or null (plain nms)
  const int* sides;      // or null
  const int* level;      // or null
  const int* count;      // [n] or null (then k_fixed)
  int k_fixed;
  int cap;
  double thr;  // torchvision passes iou_threshold as a C++ double (0.3, not 0.3f)
  float ratio_h, ratio_w;
  const float* ratios;   // optional per-image [n][2] = (ratio_h, ratio_w) (batches of differently sized images)
  int rescale;           // multiply output boxes by the ratios
  char* scratch;
  long scratch_stride;   // bytes per image
  int pad_cap;           // pow2 >= cap
  float* det_boxes;
  float* det_scores;
  int* det_labels;
  int* det_sides;
  int* det_level;
  int* det_keep;
  int* det_count;
};

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1,
                                       float bx2, float by2, float barea, double thr) {
  const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
  const float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (aarea + barea - inter);
  return (double)ovr > thr;  // torchvision CPU kernel compares the fp32 IoU with a double threshold
}

__global__ __launch_bounds__(1024) void nms_kernel(const NmsArgs a) {
  __shared__ unsigned long long keys_lds[kSortLds];
  __shared__ __attribute__((aligned(16))) float kept_lds[kSortLds * kKeptRec];   // (the bitmask form keeps 8-byte words in it)
  __shared__ float red[16];
  __shared__ float maxc_s;
  const int img = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.count ? min(a.count[img], a.cap) : a.k_fixed;
  const float* boxes = a.boxes + (long)img * a.cap * 4;
  const float* scores = a.scores + (long)img * a.cap;
  const int* labels = a.labels ? a.labels + (long)img * a.cap : nullptr;
  if (K <= 0) {
    if (tid == 0 && a.det_count) a.det_count[img] = 0;
    return;
  }
  char* scr = a.scratch + (long)img * a.scratch_stride;
  unsigned long long* keys_g = reinterpret_cast<unsigned long long*>(scr);
  const int npad = pow2_at_least(K);
  unsigned long long* keys = npad <= kSortLds ? keys_lds : keys_g;
  // The greedy pass re-reads the kept list once per 64-candidate tile, one record per iteration with a
  // data-dependent exit: from global memory that is one L2 round trip per kept box (~170 us for 250
  // candidates); the usual case (K <= 2048) keeps it in LDS.
  float* kept = npad <= kSortLds ? kept_lds : reinterpret_cast<float*>(scr + (long)a.pad_cap * 8);

  // only as many waves as the padded sort needs stay: every __syncthreads() below then rendezvous 4 waves
  // instead of 16 for the usual ~250 candidates (terminated waves do not take part in s_barrier)
  // (the bitmask form below has ~K^2 / 64 independent work items and five barriers in all: four threads per padded key there)
  const int nwant = npad <= kBitK ? 4 * npad : npad;
  const int nthreads = nwant < 64 ? 64 : (nwant < (int)blockDim.x ? nwant : (int)blockDim.x);
  if (tid >= nthreads) return;
  // keys + max coordinate (torchvision's coordinate trick needs boxes.max())
  float mx = -3.402823466e38f;
  for (int i = tid; i < npad; i += nthreads) {
    keys[i] = i < K ? make_key(scores[i], i) : ~0ull;
    if (i < K) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (long)i * 4);
      mx = fmaxf(fmaxf(fmaxf(mx, b[0]), fmaxf(b[1], b[2])), b[3]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int k = 1; k < (nthreads >> 6); ++k) m = fmaxf(m, red[k]);
    maxc_s = m;
  }
  // Up to kRankSort keys: RANK sort -- the keys are distinct (the candidate index is their low word), so the position of a key
  // in the sorted list is the number of smaller keys; a thread owns at most one key (nthreads >= npad >= K), counts over all K with
  // wave-uniform LDS reads (a broadcast, no bank conflict) and drops its key at that position of a second array: two barriers
  // instead of the bitonic network's 36-55 (round 4: the sort was about half of the kernel's 66 us at ~260 candidates).  The
  // order is the same total order, so everything downstream is bit-identical.
  if (npad <= kRankSort) {   // (the keys are visible: the barrier of the maximum above)
    unsigned long long* sorted_lds = keys_lds + kRankSort;   // the upper half of the key array is free at these sizes
    unsigned long long mine = ~0ull;
    int rank = 0;
    if (tid < K) {
      mine = keys_lds[tid];
      int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      int j = 0;
      for (; j + 4 <= K; j += 4) {
        r0 += keys_lds[j] < mine;
        r1 += keys_lds[j + 1] < mine;
        r2 += keys_lds[j + 2] < mine;
        r3 += keys_lds[j + 3] < mine;
      }
      for (; j < K; ++j) r0 += keys_lds[j] < mine;
      rank = (r0 + r1) + (r2 + r3);
    }
    if (tid < K) sorted_lds[rank] = mine;
    keys = sorted_lds;
  } else
  // bitonic sort (ascending keys = descending score, ties by ascending index)
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < npad; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            keys[i] = y;
            keys[ixj] = x;
          }
        }
      }
    }
  }
  __syncthreads();
  __shared__ unsigned long long alive_w[4];
  __shared__ unsigned long long mask_w[4][64];
  __shared__ int nk_s;
  __shared__ unsigned long long kept_w[kBitK / 64];
  // batched_nms: coordinate trick iff boxes.numel() <= 4000, else per-class on raw boxes
  const bool classwise = labels != nullptr;
  const bool trick = classwise && (4 * K <= 4000);
  const float offs_unit = maxc_s + 1.0f;
  const double thr = a.thr;
  int nk = 0;
  bool resolved = false;
  if (npad <= kBitK) {
    // Up to kBitK candidates: the BITMASK form.  All waves compute the suppression matrix at once -- bit j of word (i, w) says
    // "candidate i, if kept, suppresses the later candidate j = 64 w + bit" (the same iou_gt on the same operands as the greedy
    // tiles below, for every pair i < j instead of for the pairs the greedy order happens to reach) -- and ONE wave then walks
    // the candidates in score order with the removed set in registers: candidate i is kept iff its bit is still clear, and a
    // kept candidate ORs its row into the set.  Same decisions as the tiles (a candidate is dropped iff a KEPT earlier one
    // overlaps it), so survivors and their order are bit-identical; what goes is the per-tile chain of global gathers and
    // rendezvous (round 4, ~260 candidates: 73 -> 41 us of the kernel in isolation).
    float* rec = kept_lds;                                                                  // [K][6]: x1 y1 x2 y2 area label
    unsigned long long* rows = reinterpret_cast<unsigned long long*>(kept_lds + kBitK * kKeptRec);   // [K][nw]
    const int nw = (K + 63) >> 6;
    unsigned long long* sk = keys_lds + kRankSort;   // = keys, as an LDS pointer (no flat accesses in the sequential walk)
    if (tid < K) {
      const int idx = (int)(unsigned)(sk[tid] & 0xFFFFFFFFull);
      const f32x4 raw = *reinterpret_cast<const f32x4*>(boxes + (long)idx * 4);
      const int lab = labels ? labels[idx] : 0;
      float x1 = raw[0], y1 = raw[1], x2 = raw[2], y2 = raw[3];
      if (trick) {
        const float off = (float)lab * offs_unit;
        x1 = x1 + off;
        y1 = y1 + off;
        x2 = x2 + off;
        y2 = y2 + off;
      }
      float* r = rec + tid * kKeptRec;
      r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2; r[4] = (x2 - x1) * (y2 - y1); r[5] = __int_as_float(lab);
    }
    __syncthreads();
    // word-major work list: word w holds rows 0 .. min(K, 64 (w + 1)) - 1 (a row only suppresses LATER candidates), so the
    // 64 lanes of a wave share w and read the same record j at a time (an LDS broadcast)
    int total = 0;
    for (int w = 0; w < nw; ++w) total += min(K, 64 * (w + 1));
    for (int item = tid; item < total; item += nthreads) {
      int w = 0, i = item;
      while (i >= min(K, 64 * (w + 1))) {
        i -= min(K, 64 * (w + 1));
        ++w;
      }
      const float* ri = rec + i * kKeptRec;
      const float ix1 = ri[0], iy1 = ri[1], ix2 = ri[2], iy2 = ri[3], ia = ri[4];
      const int il = __float_as_int(ri[5]);
      unsigned long long word = 0ull;
      const int j0 = 64 * w, jn = min(64, K - j0);
      for (int jj = 0; jj < jn; ++jj) {
        const float* rj = rec + (j0 + jj) * kKeptRec;
        const int jl = __float_as_int(rj[5]);
        if (j0 + jj > i && (trick || !classwise || jl == il) &&
            iou_gt(ix1, iy1, ix2, iy2, ia, rj[0], rj[1], rj[2], rj[3], rj[4], thr))
          word |= 1ull << jj;
      }
      rows[i * nw + w] = word;
    }
    __syncthreads();
    if (wave == 0) {
      // lane l < nw holds word l of the removed set; the rows of the next four candidates are fetched while the current four
      // are decided (their addresses do not depend on the decisions).  Nothing else touches memory in this loop: a row only
      // has bits of LATER candidates, so bit i of the set is final once candidate i has been visited -- the kept set is
      // simply the complement of the removed set at the end.
      unsigned long long remv = 0ull, cur[4], nxt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[u] = (lane < nw && u < K) ? rows[u * nw + lane] : 0ull;   // (u < 4: word 0 is their first)
      for (int i0 = 0; i0 < K; i0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)   // (words in front of a row's own were never written: they read as zero)
          nxt[u] = (lane < nw && lane >= ((i0 + 4 + u) >> 6) && i0 + 4 + u < K) ? rows[(i0 + 4 + u) * nw + lane] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u;   // (rows beyond K were fetched as zeros: visiting them changes nothing)
          const int wi = i >> 6;
          const unsigned lo = __builtin_amdgcn_readlane((unsigned)remv, wi);
          const unsigned hi = __builtin_amdgcn_readlane((unsigned)(remv >> 32), wi);
          const unsigned long long rw = ((unsigned long long)hi << 32) | lo;
          if (!((rw >> (i & 63)) & 1ull)) remv |= cur[u];   // (wave-uniform condition)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
      }
      if (lane < nw) {
        const int left = K - 64 * lane;
        kept_w[lane] = ~remv & (left >= 64 ? ~0ull : ((1ull << left) - 1ull));
      }
    }
    __syncthreads();
    // compaction, all threads: kept candidate t goes to position (kept candidates before t); the list lands in the lower half
    // of the key array (the unsorted keys are no longer needed), where the output loop below reads it
    if (tid < K) {
      const int wt = tid >> 6;
      const unsigned long long kw = kept_w[wt];
      if ((kw >> (tid & 63)) & 1ull) {
        int pos = __popcll(kw & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < wt; ++w) pos += __popcll(kept_w[w]);
        keys_lds[pos] = sk[tid] & 0xFFFFFFFFull;
      }
    }
    for (int w = 0; w < nw; ++w) nk += __popcll(kept_w[w]);
    keys = keys_lds;
    __syncthreads();
    resolved = true;
  }
  // Greedy pass, 64 candidates (one tile) at a time in score order.  Up to four waves share the two
  // IoU-heavy parts of a tile -- the check against the boxes kept so far (wave w takes kept entries w, w+W, ..)
  // and the in-tile suppression masks (wave w takes 64/W of the tile's columns) -- and wave 0 then resolves the
  // tile sequentially.  Same comparisons as a single wave, only distributed: results are bit-identical.
  const int W = (nthreads >> 6) < 4 ? (nthreads >> 6) : 4;
  if (wave >= W) return;
  for (int t0 = 0; !resolved && t0 < K; t0 += 64) {
    const int t = t0 + lane;
    const bool has = t < K;
    const int idx = has ? (int)(unsigned)(keys[t] & 0xFFFFFFFFull) : 0;
    f32x4 raw = {0.f, 0.f, 0.f, 0.f};
    int lab = 0;
    if (has) {
      raw = *reinterpret_cast<const f32x4*>(boxes + (long)idx * 4);
      lab = labels ? labels[idx] : 0;
    }
    float x1 = raw[0], y1 = raw[1], x2 = raw[2], y2 = raw[3];
    if (trick) {
      const float off = (float)lab * offs_unit;
      x1 = x1 + off;
      y1 = y1 + off;
      x2 = x2 + off;
      y2 = y2 + off;
    }
    const float area = (x2 - x1) * (y2 - y1);
    bool alive = has;
    // against the boxes kept so far (wave-uniform reads of the kept list), this wave's share
    for (int q = wave; q < nk; q += W) {
      const float* kr = kept + (long)q * kKeptRec;
      const float kx1 = kr[0], ky1 = kr[1], kx2 = kr[2], ky2 = kr[3], ka = kr[4];
      const int kl = __float_as_int(kr[5]);
      if (alive && (trick || !classwise || kl == lab) && iou_gt(kx1, ky1, kx2, ky2, ka, x1, y1, x2, y2, area, thr))
        alive = false;
    }
    // inside the tile: mask of later lanes this lane would suppress, this wave's share of the columns
    unsigned long long mask = 0ull;
    const int jn = 64 / W;
    for (int j = wave * jn; j < (wave + 1) * jn; ++j) {
      const float jx1 = __shfl(x1, j), jy1 = __shfl(y1, j), jx2 = __shfl(x2, j), jy2 = __shfl(y2, j);
      const float ja = __shfl(area, j);
      const int jl = __shfl(lab, j);
      const bool jhas = (t0 + j) < K;
      if (j > lane && jhas && (trick || !classwise || jl == lab) &&
          iou_gt(x1, y1, x2, y2, area, jx1, jy1, jx2, jy2, ja, thr))
        mask |= 1ull << j;
    }
    const unsigned long long my_alive = __ballot(alive);
    if (W > 1) {
      if (lane == 0) alive_w[wave] = my_alive;
      mask_w[wave][lane] = mask;
      __syncthreads();
    }
    if (wave == 0) {
      unsigned long long alive_bits = my_alive;
      for (int w = 1; w < W; ++w) {
        alive_bits &= alive_w[w];
        mask |= mask_w[w][lane];
      }
      for (int i = 0; i < 64; ++i) {
        const unsigned long long mi = __shfl(mask, i);
        if ((alive_bits >> i) & 1ull) alive_bits &= ~mi;
      }
      const bool keep = (alive_bits >> lane) & 1ull;
      if (keep) {
        const int pos = nk + __popcll(alive_bits & ((1ull << lane) - 1ull));
        float* kr = kept + (long)pos * kKeptRec;
        kr[0] = x1; kr[1] = y1; kr[2] = x2; kr[3] = y2; kr[4] = area; kr[5] = __int_as_float(lab);
        // the kept candidate's index goes where the sorted keys were: pos <= t, and this tile's keys have been read by every
        // lane already, so no unread key is overwritten.  The output rows are written after the loop, by all waves at once
        // (round 4: gathering scores / sides / levels here stalled wave 0 for a memory round trip per tile).
        keys[pos] = (unsigned long long)(unsigned)idx;
      }
      nk += __popcll(alive_bits);
      if (lane == 0) nk_s = nk;
      __threadfence_block();  // kept[] written above is read by every lane in the next tile
    }
    if (W > 1) {
      if (npad > kSortLds) __threadfence();   // kept[] / keys[] live in global scratch (K > 2048): make them visible to the other waves
      __syncthreads();
      nk = nk_s;
    }
  }
  // ---- output rows: kept candidate pos -> boxes (rescaled), score, label, side, level, index ----
  if (W == 1) __threadfence_block();
  for (int pos = wave * 64 + lane; pos < nk; pos += W * 64) {
    const int idx = (int)(unsigned)(keys[pos] & 0xFFFFFFFFull);
    const long o = (long)img * a.cap + pos;
    if (a.det_boxes) {
      const f32x4 raw = *reinterpret_cast<const f32x4*>(boxes + (long)idx * 4);
      float ox1 = raw[0], oy1 = raw[1], ox2 = raw[2], oy2 = raw[3];
      if (a.rescale) {
        const float rh = a.ratios ? a.ratios[img * 2] : a.ratio_h, rw = a.ratios ? a.ratios[img * 2 + 1] : a.ratio_w;
        ox1 = ox1 * rw; ox2 = ox2 * rw;
        oy1 = oy1 * rh; oy2 = oy2 * rh;
      }
      a.det_boxes[o * 4 + 0] = ox1; a.det_boxes[o * 4 + 1] = oy1;
      a.det_boxes[o * 4 + 2] = ox2; a.det_boxes[o * 4 + 3] = oy2;
    }
    if (a.det_scores) a.det_scores[o] = scores[idx];
    if (a.det_labels) a.det_labels[o] = labels ? labels[idx] : 0;
    if (a.det_sides) a.det_sides[o] = a.sides[(long)img * a.cap + idx];
    if (a.det_level) a.det_level[o] = a.level[(long)img * a.cap + idx];
    if (a.det_keep) a.det_keep[o] = idx;
  }
  if (wave != 0) return;
  if (lane == 0 && a.det_count) a.det_count[img] = nk;
}

// ---------------------------------------------------------------------------------------
// crop: top-1 hand box -> padded int box; nearest gather to out x out
// ---------------------------------------------------------------------------------------
__global__ void crop_box_kernel(const float* __restrict__ det_boxes, const int* __restrict__ det_labels,
                                const int* __restrict__ det_count, int cap, int hand_label, int n, int h, int w,
                                long long* __restrict__ crop_box, int* __restrict__ has_hand) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n) return;
  const int cnt = min(det_count[img], cap);
  int found = -1;
  for (int i = 0; i < cnt; ++i)
    if (det_labels[(long)img * cap + i] == hand_label) {
      found = i;
      break;
    }
  long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  int ok = 0;
  if (found >= 0) {
    const float* b = det_boxes + ((long)img * cap + found) * 4;
    b0 = (long long)b[0]; b1 = (long long)b[1]; b2 = (long long)b[2]; b3 = (long long)b[3];  // trunc toward 0
    const long long bw = b2 - b0, bh = b3 - b1;
    // python: box[0] = max(0, box[0] - 0.4 * w) on 0-d tensors: fp32 arithmetic, trunc on store
    const float pw = 0.4f * (float)bw, phh = 0.4f * (float)bh;
    const float t0 = (float)b0 - pw, t1 = (float)b1 - phh;
    const float t2 = (float)b2 + pw, t3 = (float)b3 + phh;
    b0 = t0 > 0.f ? (long long)t0 : 0;
    b1 = t1 > 0.f ? (long long)t1 : 0;
    b2 = t2 < (float)w ? (long long)t2 : (long long)w;
    b3 = t3 < (float)h ? (long long)t3 : (long long)h;
    // slice [b1 : b3+1, b0 : b2+1] clamped to the image must be non-empty
    const long long ch = (b3 + 1 < h ? b3 + 1 : h) - b1, cwid = (b2 + 1 < w ? b2 + 1 : w) - b0;
    ok = (ch > 0 && cwid > 0 && b1 >= 0 && b0 >= 0) ? 1 : 0;
  }
  if (!ok) b0 = b1 = b2 = b3 = 0;
  crop_box[(long)img * 4 + 0] = b0;
  crop_box[(long)img * 4 + 1] = b1;
  crop_box[(long)img * 4 + 2] = b2;
  crop_box[(long)img * 4 + 3] = b3;
  has_hand[img] = ok;
}

__global__ __launch_bounds__(256) void crop_gather_kernel(const float* __restrict__ depth,
                                                          const long long* __restrict__ crop_box,
                                                          const int* __restrict__ has_hand, int n, int h, int w,
                                                          int in_ch, int reorder, int out, int c4,
                                                          float* __restrict__ crops) {
  const long total = (long)n * out * out;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % out);
    const long t = i / out;
    const int oy = (int)(t % out);
    const int img = (int)(t / out);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (has_hand[img]) {
      const long long* b = crop_box + (long)img * 4;
      const int x1 = (int)b[0], y1 = (int)b[1];
      const int cw = (int)((b[2] + 1 < w ? b[2] + 1 : w) - b[0]);
      const int ch = (int)((b[3] + 1 < h ? b[3] + 1 : h) - b[1]);
      // F.interpolate(mode='nearest'): src = min(floor(dst * (float)in / out), in - 1)
      int sy, sx;
      if (ch == out) sy = oy; else if (out == 2 * ch) sy = oy >> 1;
      else sy = min((int)floorf((float)oy * ((float)ch / (float)out)), ch - 1);
      if (cw == out) sx = ox; else if (out == 2 * cw) sx = ox >> 1;
      else sx = min((int)floorf((float)ox * ((float)cw / (float)out)), cw - 1);
      const long pix = (long)(y1 + sy) * w + (x1 + sx);
      for (int c = 0; c < in_ch; ++c) {
        // RGBD: depth_crop[[2,1,0,3]] (handnet_pipeline.py:102): output channel c reads input channel perm[c]
        const int src_c = (reorder && c < 3) ? 2 - c : c;
        o[c] = depth[((long)img * in_ch + src_c) * h * w + pix];
      }
    }
    *reinterpret_cast<f32x4*>(crops + i * c4 * 4) = o;
    for (int q = 1; q < c4; ++q) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(crops + i * c4 * 4 + q * 4) = z;
    }
  }
}

int grid_for(long total, int block) {
  const long g = (total + block - 1) / block;
  return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

long nms_scratch_stride(int cap) {
  const long b = (long)pow2_at_least(cap) * 8 + (long)cap * kKeptRec * 4;
  return (b + 255) / 256 * 256;
}

}  // namespace

static int preprocess_run(const float* src, const float* const* srcs, const int32_t* geom, void* dst, int split, int n,
                          int h, int w, int oh, int ow, int ph, int pw, int border, const float mean[3],
                          const float stdv[3], void* stream) {
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean[c];
    nm.stdv[c] = stdv[c];
  }
  ImageGeom g;
  g.srcs = srcs;
  g.geom = geom;
  const float scale_h = geom ? 1.f : (float)h / (float)oh, scale_w = geom ? 1.f : (float)w / (float)ow;
  const float need_r = (kPreTH - 1) * scale_h + 3.f, need_c = (kPreTW - 1) * scale_w + 3.f;   // source rows / columns under a tile
  if (split && !geom && n <= 65535 && !hn::env_flags().pre_generic && need_r <= 16.f && need_c <= 160.f) {
    // dense batch, source rectangle of a tile fits an LDS patch: the tiled kernel (HN_PREPROCESS_GENERIC=1: A/B)
    const int hb = ph + 2 * border, wb = pw + 2 * border;
    const dim3 grid(hn::cdiv(wb, kPreTW), hn::cdiv(hb, kPreTH), n);
    if (need_r <= 8.f && need_c <= 96.f)
      hipLaunchKernelGGL((fcos_preprocess_split_tiled_kernel<8, 96>), grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n,
                         h, w, oh, ow, ph, pw, border, scale_h, scale_w, nm, hn::range_flag_ptr());
    else
      hipLaunchKernelGGL((fcos_preprocess_split_tiled_kernel<16, 160>), grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n,
                         h, w, oh, ow, ph, pw, border, scale_h, scale_w, nm, hn::range_flag_ptr());
    HN_CHECK_LAUNCH("fcos_preprocess_split_tiled_kernel");
    return HN_OK;
  }
  if (split) {
    const long total = (long)n * (ph + 2 * border) * (pw + 2 * border);
    if (total < ((long)1 << 31))
      hipLaunchKernelGGL(fcos_preprocess_split_kernel<unsigned>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                         (_Float16*)dst, n, h, w, oh, ow, ph, pw, border, scale_h, scale_w, nm, hn::range_flag_ptr(), g);
    else
      hipLaunchKernelGGL(fcos_preprocess_split_kernel<long>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                         (_Float16*)dst, n, h, w, oh, ow, ph, pw, border, scale_h, scale_w, nm, hn::range_flag_ptr(), g);
    HN_CHECK_LAUNCH("fcos_preprocess_split_kernel");
  } else {
    const long total = (long)n * ph * pw;
    hipLaunchKernelGGL(fcos_preprocess_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (float*)dst, n, h, w, oh, ow, ph, pw, scale_h, scale_w, nm, g);
    HN_CHECK_LAUNCH("fcos_preprocess_kernel");
  }
  return HN_OK;
}

extern "C" int hn_fcos_preprocess_f32(const float* src, float* dst, int n, int h, int w, int oh, int ow, int ph,
                                      int pw, const float mean[3], const float stdv[3], void* stream) {
  HN_CHECK_ARG(src && dst && mean && stdv, "hn_fcos_preprocess_f32: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ph >= oh && pw >= ow, "bad dims");
  return preprocess_run(src, nullptr, nullptr, dst, 0, n, h, w, oh, ow, ph, pw, 0, mean, stdv, stream);
}

extern "C" int hn_fcos_preprocess_split(const float* src, void* dst16, int n, int h, int w, int oh, int ow, int ph,
                                        int pw, int border, const float mean[3], const float stdv[3], void* stream) {
  HN_CHECK_ARG(src && dst16 && mean && stdv, "hn_fcos_preprocess_split: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ph >= oh && pw >= ow && border >= 0, "bad dims");
  return preprocess_run(src, nullptr, nullptr, dst16, 1, n, h, w, oh, ow, ph, pw, border, mean, stdv, stream);
}

// A batch of differently sized images (torchvision batch_images): srcs = DEVICE array of n device pointers to
// [3][h_i][w_i] images, geom = DEVICE int32 [n][4] = {h_i, w_i, oh_i, ow_i} (oh_i <= ph, ow_i <= pw: the caller
// has checked that, the kernel clips to the canvas).  split = 0: fp32 [n][ph][pw][4] canvas; 1: the stem image.
extern "C" int hn_fcos_preprocess_list(const float* const* srcs, const int32_t* geom, void* dst, int split, int n, int ph,
                                       int pw, int border, const float mean[3], const float stdv[3], void* stream) {
  HN_CHECK_ARG(srcs && geom && dst && mean && stdv, "hn_fcos_preprocess_list: null pointer");
  HN_CHECK_ARG(n > 0 && ph > 0 && pw > 0 && border >= 0 && (split == 0 || split == 1), "bad dims");
  return preprocess_run(nullptr, srcs, geom, dst, split, n, 1, 1, 1, 1, ph, pw, split ? border : 0, mean, stdv, stream);
}

extern "C" int64_t hn_fcos_candidates_ws_bytes(int n, int total_points) {
  return n > 0 && total_points > 0 ? (int64_t)n * ((total_points + 1023) / 1024) * 4 : 0;
}

static int candidates_run(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh, float* cand_boxes,
                          float* cand_scores, int32_t* cand_labels, int32_t* cand_sides, int32_t* cand_level,
                          int32_t* cand_point, int32_t* cand_count, int cap, void* workspace, int64_t workspace_bytes,
                          void* stream);

extern "C" int hn_fcos_candidates(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh,
                                  float* cand_boxes, float* cand_scores, int32_t* cand_labels, int32_t* cand_sides,
                                  int32_t* cand_level, int32_t* cand_point, int32_t* cand_count, int cap,
                                  void* stream) {
  return candidates_run(lv, n, num_classes, score_thresh, cand_boxes, cand_scores, cand_labels, cand_sides, cand_level,
                        cand_point, cand_count, cap, nullptr, 0, stream);
}

extern "C" int hn_fcos_candidates_ws(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh,
                                     float* cand_boxes, float* cand_scores, int32_t* cand_labels, int32_t* cand_sides,
                                     int32_t* cand_level, int32_t* cand_point, int32_t* cand_count, int cap,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  HN_CHECK_ARG(workspace && (uintptr_t)workspace % 4 == 0, "hn_fcos_candidates_ws: null / unaligned workspace");
  return candidates_run(lv, n, num_classes, score_thresh, cand_boxes, cand_scores, cand_labels, cand_sides, cand_level,
                        cand_point, cand_count, cap, workspace, workspace_bytes, stream);
}

static int candidates_run(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh, float* cand_boxes,
                          float* cand_scores, int32_t* cand_labels, int32_t* cand_sides, int32_t* cand_level,
                          int32_t* cand_point, int32_t* cand_count, int cap, void* workspace, int64_t workspace_bytes,
                          void* stream) {
  HN_CHECK_ARG(lv && cand_boxes && cand_scores && cand_labels && cand_sides && cand_level && cand_count,
               "hn_fcos_candidates: null pointer");
  HN_CHECK_ARG(lv->num_levels > 0 && lv->num_levels <= HN_FCOS_MAX_LEVELS, "bad level count");
  HN_CHECK_ARG(n > 0 && num_classes > 0 && num_classes <= 64 && cap > 0, "bad dims");
  LevelTable lt;
  lt.num_levels = lv->num_levels;
  lt.start[0] = 0;
  for (int l = 0; l < lv->num_levels; ++l) {
    HN_CHECK_ARG(lv->h[l] > 0 && lv->w[l] > 0 && lv->stride[l] > 0 && lv->cls_lr[l] && lv->reg_ctr[l], "bad level %d", l);
    lt.h[l] = lv->h[l];
    lt.w[l] = lv->w[l];
    lt.stride[l] = lv->stride[l];
    lt.cls_lr[l] = lv->cls_lr[l];
    lt.reg_ctr[l] = lv->reg_ctr[l];
    lt.start[l + 1] = lt.start[l] + lv->h[l] * lv->w[l];
  }
  for (int l = lv->num_levels; l < HN_FCOS_MAX_LEVELS; ++l) {
    lt.h[l] = lt.w[l] = lt.stride[l] = 0;
    lt.cls_lr[l] = lt.reg_ctr[l] = nullptr;
    lt.start[l + 1] = lt.start[lv->num_levels];
  }
  if (workspace) {
    const int P = lt.start[lv->num_levels], chunks = (P + 1023) / 1024;
    HN_CHECK_ARG(workspace_bytes >= hn_fcos_candidates_ws_bytes(n, P), "workspace too small: %lld < %lld bytes",
                 (long long)workspace_bytes, (long long)hn_fcos_candidates_ws_bytes(n, P));
    HN_CHECK_ARG(n <= 65535, "more than 65535 images");
    hipLaunchKernelGGL(fcos_candidates_count_kernel, dim3(chunks, n), dim3(1024), 0, (hipStream_t)stream, lt, num_classes,
                       score_thresh, (int*)workspace);
    HN_CHECK_LAUNCH("fcos_candidates_count_kernel");
    hipLaunchKernelGGL(fcos_candidates_scatter_kernel, dim3(chunks, n), dim3(1024), 0, (hipStream_t)stream, lt, num_classes,
                       score_thresh, (const int*)workspace, cand_boxes, cand_scores, cand_labels, cand_sides, cand_level,
                       cand_point, cand_count, cap);
    HN_CHECK_LAUNCH("fcos_candidates_scatter_kernel");
    return HN_OK;
  }
  hipLaunchKernelGGL(fcos_candidates_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, lt, num_classes,
                     score_thresh, cand_boxes, cand_scores, cand_labels, cand_sides, cand_level, cand_point, cand_count,
                     cap);
  HN_CHECK_LAUNCH("fcos_candidates_kernel");
  return HN_OK;
}

extern "C" int hn_fcos_ext_gather(const hn_fcos_levels* lv, const float* const* ext, const int32_t* det_keep,
                                  const int32_t* cand_point, const int32_t* det_count, int n, int cap,
                                  int32_t* det_contacts, float* det_dxdymags, void* stream) {
  HN_CHECK_ARG(lv && ext && det_keep && cand_point && det_count && det_contacts && det_dxdymags,
               "hn_fcos_ext_gather: null pointer");
  HN_CHECK_ARG(lv->num_levels > 0 && lv->num_levels <= HN_FCOS_MAX_LEVELS, "bad level count");
  HN_CHECK_ARG(n > 0 && cap > 0, "bad dims");
  ExtTable et;
  et.num_levels = lv->num_levels;
  et.start[0] = 0;
  for (int l = 0; l < HN_FCOS_MAX_LEVELS; ++l) {
    const bool on = l < lv->num_levels;
    if (on) HN_CHECK_ARG(lv->h[l] > 0 && lv->w[l] > 0 && ext[l], "bad level %d", l);
    et.hw[l] = on ? lv->h[l] * lv->w[l] : 0;
    et.ext[l] = on ? ext[l] : nullptr;
    et.start[l + 1] = et.start[l] + et.hw[l];
  }
  const long total = (long)n * cap;
  hipLaunchKernelGGL(fcos_ext_gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, et,
                     det_keep, cand_point, det_count, n, cap, det_contacts, det_dxdymags);
  HN_CHECK_LAUNCH("fcos_ext_gather_kernel");
  return HN_OK;
}

extern "C" int64_t hn_fcos_nms_scratch_bytes(int n, int cap) {
  if (n <= 0 || cap <= 0) return 0;
  return (int64_t)n * nms_scratch_stride(cap);
}

static int fcos_nms_run(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                        const int32_t* cand_sides, const int32_t* cand_level, const int32_t* cand_count, int n, int cap,
                        double iou_thresh, float ratio_h, float ratio_w, const float* ratios, void* scratch,
                        float* det_boxes, float* det_scores, int32_t* det_labels, int32_t* det_sides,
                        int32_t* det_level, int32_t* det_keep, int32_t* det_count, void* stream) {
  HN_CHECK_ARG(cand_boxes && cand_scores && cand_labels && cand_sides && cand_level && cand_count && scratch,
               "hn_fcos_nms: null input");
  HN_CHECK_ARG(det_boxes && det_scores && det_labels && det_sides && det_level && det_count, "hn_fcos_nms: null output");
  HN_CHECK_ARG(n > 0 && cap > 0 && cap <= (1 << 24), "bad dims");
  NmsArgs a;
  a.boxes = cand_boxes; a.scores = cand_scores; a.labels = cand_labels; a.sides = cand_sides; a.level = cand_level;
  a.count = cand_count; a.k_fixed = 0; a.cap = cap; a.thr = iou_thresh; a.ratio_h = ratio_h; a.ratio_w = ratio_w;
  a.ratios = ratios;
  a.rescale = 1; a.scratch = (char*)scratch; a.scratch_stride = nms_scratch_stride(cap); a.pad_cap = pow2_at_least(cap);
  a.det_boxes = det_boxes; a.det_scores = det_scores; a.det_labels = det_labels; a.det_sides = det_sides;
  a.det_level = det_level; a.det_keep = det_keep; a.det_count = det_count;
  hipLaunchKernelGGL(nms_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, a);
  HN_CHECK_LAUNCH("nms_kernel");
  return HN_OK;
}

extern "C" int hn_fcos_nms(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                           const int32_t* cand_sides, const int32_t* cand_level, const int32_t* cand_count, int n,
                           int cap, double iou_thresh, float ratio_h, float ratio_w, void* scratch, float* det_boxes,
                           float* det_scores, int32_t* det_labels, int32_t* det_sides, int32_t* det_level,
                           int32_t* det_keep, int32_t* det_count, void* stream) {
  return fcos_nms_run(cand_boxes, cand_scores, cand_labels, cand_sides, cand_level, cand_count, n, cap, iou_thresh,
                      ratio_h, ratio_w, nullptr, scratch, det_boxes, det_scores, det_labels, det_sides, det_level,
                      det_keep, det_count, stream);
}

// Same with one (ratio_h, ratio_w) pair per image: ratios = DEVICE fp32 [n][2] (resize_boxes of a batch whose
// images differ in size, fcos_utils/fcos.py:661-669,770-783).
extern "C" int hn_fcos_nms_ratios(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                                  const int32_t* cand_sides, const int32_t* cand_level, const int32_t* cand_count, int n,
                                  int cap, double iou_thresh, const float* ratios, void* scratch, float* det_boxes,
                                  float* det_scores, int32_t* det_labels, int32_t* det_sides, int32_t* det_level,
                                  int32_t* det_keep, int32_t* det_count, void* stream) {
  HN_CHECK_ARG(ratios, "hn_fcos_nms_ratios: null ratios");
  return fcos_nms_run(cand_boxes, cand_scores, cand_labels, cand_sides, cand_level, cand_count, n, cap, iou_thresh, 1.f,
                      1.f, ratios, scratch, det_boxes, det_scores, det_labels, det_sides, det_level, det_keep, det_count,
                      stream);
}

extern "C" int hn_nms(const float* boxes, const float* scores, int k, double iou_thresh, void* scratch, int32_t* keep,
                      int32_t* num_keep, void* stream) {
  HN_CHECK_ARG(boxes && scores && scratch && keep && num_keep, "hn_nms: null pointer");
  HN_CHECK_ARG(k >= 0 && k <= (1 << 24), "bad k");
  if (k == 0) {
    HN_CHECK_HIP(hipMemsetAsync(num_keep, 0, sizeof(int32_t), (hipStream_t)stream));
    return HN_OK;
  }
  NmsArgs a;
  a.boxes = boxes; a.scores = scores; a.labels = nullptr; a.sides = nullptr; a.level = nullptr; a.count = nullptr;
  a.k_fixed = k; a.cap = k; a.thr = iou_thresh; a.ratio_h = a.ratio_w = 1.f; a.ratios = nullptr; a.rescale = 0;
  a.scratch = (char*)scratch; a.scratch_stride = nms_scratch_stride(k); a.pad_cap = pow2_at_least(k);
  a.det_boxes = nullptr; a.det_scores = nullptr; a.det_labels = nullptr; a.det_sides = nullptr; a.det_level = nullptr;
  a.det_keep = keep; a.det_count = num_keep;
  hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  HN_CHECK_LAUNCH("nms_kernel");
  return HN_OK;
}

extern "C" int hn_crop_resize(const float* det_boxes, const int32_t* det_labels, const int32_t* det_count, int cap,
                              int hand_label, const float* depth, int n, int in_ch, int reorder_bgr, int h, int w,
                              int out, int cpad,
                              int64_t* crop_box, int32_t* has_hand, float* crops, void* stream) {
  HN_CHECK_ARG(det_boxes && det_labels && det_count && depth && crop_box && has_hand && crops,
               "hn_crop_resize: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && out > 0 && cap > 0 && cpad >= 4 && cpad % 4 == 0, "bad dims");
  HN_CHECK_ARG(in_ch >= 1 && in_ch <= 4, "depth image must have 1..4 channels (got %d)", in_ch);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(crop_box_kernel, dim3((n + 63) / 64), dim3(64), 0, st, det_boxes, det_labels, det_count, cap,
                     hand_label, n, h, w, (long long*)crop_box, has_hand);
  HN_CHECK_LAUNCH("crop_box_kernel");
  const long total = (long)n * out * out;
  hipLaunchKernelGGL(crop_gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, depth,
                     (const long long*)crop_box, has_hand, n, h, w, in_ch, reorder_bgr, out, cpad / 4, crops);
  HN_CHECK_LAUNCH("crop_gather_kernel");
  return HN_OK;
}
