// HBM-bound producers / consumers of the "S32" split activation format used by the f16x3
// convolutions (conv_igemm_f16x3.hip):   fp16 [N][H][W][C/32][2][32], per pixel and 32-channel
// block a 128-byte run hi[32] | lo[32] with hi = fp16(v), lo = fp16(v - hi).
//
//   hn_affine_split_f32   fp32 NHWC -> S32, optionally y = relu(x*scale[img][c] + shift[img][c])
//                         first: this is the GroupNorm(32,256)+ReLU of the FCOS towers
//                         (fcos_utils/fcos.py:232-239,352-359) applied between two convs
//   hn_unsplit_f32        S32 -> fp32 (hi + lo is exact in fp32)
//   hn_maxpool3x3s2_s32   3x3/2 max pooling on S32 (the max element's (hi, lo) pair is copied,
//                         so pooling commutes with the split exactly)
#include "hn_common.h"

#include <float.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 h0 = (_Float16)a[e], h1 = (_Float16)b[e];
    hi[e] = h0;
    hi[4 + e] = h1;
    lo[e] = (_Float16)(a[e] - (float)h0);
    lo[4 + e] = (_Float16)(b[e] - (float)h1);
  }
}

// one thread = one pixel x 8 channels: reads 32 B fp32, writes 16 B hi + 16 B lo
__global__ __launch_bounds__(256) void affine_split_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, long npix,
                                                           int hw, int c, int xs, int as,
                                                           _Float16* __restrict__ y, int ys, int* range_flag) {
  const int c8 = c >> 3;
  const long total = npix * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c8;
    const int ch = (int)(i - pix * c8) * 8;
    const float* src = x + pix * xs + ch;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
    if (scale) {
      const long o = (pix / hw) * as + ch;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + o), s1 = *reinterpret_cast<const f32x4*>(scale + o + 4);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + o), t1 = *reinterpret_cast<const f32x4*>(shift + o + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = a[e] * s0[e] + t0[e];
        b[e] = b[e] * s1[e] + t1[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = hn::relu(a[e]);
        b[e] = hn::relu(b[e]);
      }
    }
    if (range_flag) {
      bool bad = false;   // (NaN-aware: this pass is HBM-bound, and its input may be any fp32 tensor)
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= hn::range_bad(a[e]) | hn::range_bad(b[e]);
      if (bad) *range_flag = 1;
    }
    f16x8 hi, lo;
    split8(a, b, hi, lo);
    _Float16* dst = y + pix * ys + (ch >> 5) * 64 + (ch & 31);
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 32) = lo;
  }
}

// The same pass for channel counts whose 8-channel groups divide the block (C = 32 .. 2048, powers of two): one image
// per blockIdx.y, a lane keeps ONE channel group for its whole pixel loop, so the GroupNorm scale / shift of that group
// are loaded once, the per-item index math is a shift (the generic kernel pays two 64-bit divisions per 32 bytes), two
// pixels are in flight per iteration, and the streams bypass the caches (read once / written once).
template <bool AFFINE, int NT, int UNR>   // NT bit 0: streaming loads, bit 1: streaming stores; UNR pixels in flight per lane
__global__ __launch_bounds__(256) void affine_split_pow2_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int relu, int hw, int c8_log2,
                                                                int xs, int as, _Float16* __restrict__ y, int ys,
                                                                int* range_flag) {
  const int c8 = 1 << c8_log2;
  const int grp = threadIdx.x & (c8 - 1), ch = grp * 8;
  const int ppb = 256 >> c8_log2;                       // pixels per block and iteration
  const int img = blockIdx.y;
  f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
  if constexpr (AFFINE) {
    const long o = (long)img * as + ch;
    s0 = *reinterpret_cast<const f32x4*>(scale + o);
    s1 = *reinterpret_cast<const f32x4*>(scale + o + 4);
    t0 = *reinterpret_cast<const f32x4*>(shift + o);
    t1 = *reinterpret_cast<const f32x4*>(shift + o + 4);
  }
  const float* xi = x + (long)img * hw * xs + ch;
  _Float16* yi = y + (long)img * hw * ys + (ch >> 5) * 64 + (ch & 31);
  const int step = gridDim.x * ppb;
  auto load = [&](const float* q) {
    if constexpr (NT & 1) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
    else return *reinterpret_cast<const f32x4*>(q);
  };
  auto one = [&](f32x4 a, f32x4 b, int pix) {
    if constexpr (AFFINE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = a[e] * s0[e] + t0[e];
        b[e] = b[e] * s1[e] + t1[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = hn::relu(a[e]);
        b[e] = hn::relu(b[e]);
      }
    }
    if (range_flag) {
      bool bad = false;   // (NaN-aware: this pass is HBM-bound, and its input may be any fp32 tensor)
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= hn::range_bad(a[e]) | hn::range_bad(b[e]);
      if (bad) *range_flag = 1;
    }
    f16x8 hi, lo;
    split8(a, b, hi, lo);
    _Float16* dst = yi + (long)pix * ys;
    if constexpr (NT & 2) {
      __builtin_nontemporal_store(hi, reinterpret_cast<f16x8*>(dst));
      __builtin_nontemporal_store(lo, reinterpret_cast<f16x8*>(dst + 32));
    } else {
      *reinterpret_cast<f16x8*>(dst) = hi;
      *reinterpret_cast<f16x8*>(dst + 32) = lo;
    }
  };
  int pix = blockIdx.x * ppb + (threadIdx.x >> c8_log2);
  for (; pix + (UNR - 1) * step < hw; pix += UNR * step) {
    f32x4 a[UNR], b[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float* q = xi + (long)(pix + u * step) * xs;
      a[u] = load(q);
      b[u] = load(q + 4);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) one(a[u], b[u], pix + u * step);
  }
  for (; pix < hw; pix += step) {
    const float* q = xi + (long)pix * xs;
    one(load(q), load(q + 4), pix);
  }
}

// All FPN levels of one tower layer in ONE launch (small batch: three launches are 4.5 us of launch latency each).
// blockIdx.x walks the levels' block ranges one after the other; otherwise affine_split_pow2_kernel<true, 0, 2>.
struct SplitLevelTable {
  int count;
  int hw[HN_FCOS_MAX_LEVELS];
  int first_block[HN_FCOS_MAX_LEVELS + 1];
  const float* x[HN_FCOS_MAX_LEVELS];
  const float* scale[HN_FCOS_MAX_LEVELS];
  const float* shift[HN_FCOS_MAX_LEVELS];
  _Float16* y[HN_FCOS_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void affine_split_pow2_levels_kernel(const SplitLevelTable lt, int relu, int c8_log2, int xs,
                                                                       int as, int ys, int* range_flag) {
  // constant-index selects (a dynamic index into the by-value table would go through scratch memory)
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l < lt.count && (int)blockIdx.x >= lt.first_block[l]) lvl = l;
  int hw = lt.hw[0], b0 = lt.first_block[0], b1 = lt.first_block[1];
  const float *x = lt.x[0], *scale = lt.scale[0], *shift = lt.shift[0];
  _Float16* y = lt.y[0];
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l == lvl) {
      hw = lt.hw[l]; b0 = lt.first_block[l]; b1 = lt.first_block[l + 1];
      x = lt.x[l]; scale = lt.scale[l]; shift = lt.shift[l]; y = lt.y[l];
    }
  const int c8 = 1 << c8_log2;
  const int grp = threadIdx.x & (c8 - 1), ch = grp * 8;
  const int ppb = 256 >> c8_log2;
  const int img = blockIdx.y;
  const long o = (long)img * as + ch;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + o), s1 = *reinterpret_cast<const f32x4*>(scale + o + 4);
  const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + o), t1 = *reinterpret_cast<const f32x4*>(shift + o + 4);
  const float* xi = x + (long)img * hw * xs + ch;
  _Float16* yi = y + (long)img * hw * ys + (ch >> 5) * 64 + (ch & 31);
  const int step = (b1 - b0) * ppb;
  auto one = [&](f32x4 a, f32x4 b, int pix) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = a[e] * s0[e] + t0[e];
      b[e] = b[e] * s1[e] + t1[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = hn::relu(a[e]);
        b[e] = hn::relu(b[e]);
      }
    }
    if (range_flag) {
      bool bad = false;   // (NaN-aware: this pass is HBM-bound, and its input may be any fp32 tensor)
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= hn::range_bad(a[e]) | hn::range_bad(b[e]);
      if (bad) *range_flag = 1;
    }
    f16x8 hi, lo;
    split8(a, b, hi, lo);
    _Float16* dst = yi + (long)pix * ys;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 32) = lo;
  };
  int pix = ((int)blockIdx.x - b0) * ppb + (threadIdx.x >> c8_log2);
  for (; pix + step < hw; pix += 2 * step) {
    const float *q0 = xi + (long)pix * xs, *q1 = xi + (long)(pix + step) * xs;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(q0), b0v = *reinterpret_cast<const f32x4*>(q0 + 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(q1), b1v = *reinterpret_cast<const f32x4*>(q1 + 4);
    one(a0, b0v, pix);
    one(a1, b1v, pix + step);
  }
  for (; pix < hw; pix += step) {
    const float* q = xi + (long)pix * xs;
    one(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4), pix);
  }
}

__global__ __launch_bounds__(256) void unsplit_kernel(const _Float16* __restrict__ x, long npix, int c, int xs,
                                                      float* __restrict__ y, int ys) {
  const int c8 = c >> 3;
  const long total = npix * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c8;
    const int ch = (int)(i - pix * c8) * 8;
    const _Float16* src = x + pix * xs + (ch >> 5) * 64 + (ch & 31);
    const f16x8 hi = *reinterpret_cast<const f16x8*>(src);
    const f16x8 lo = *reinterpret_cast<const f16x8*>(src + 32);
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (float)hi[e] + (float)lo[e];
      b[e] = (float)hi[4 + e] + (float)lo[4 + e];
    }
    float* dst = y + pix * ys + ch;
    *reinterpret_cast<f32x4*>(dst) = a;
    *reinterpret_cast<f32x4*>(dst + 4) = b;
  }
}

__global__ __launch_bounds__(256) void maxpool_s32_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y,
                                                          int n, int h, int w, int c, int oh, int ow) {
  const int c8 = c >> 3;
  const long total = (long)n * oh * ow * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8) * 8;
    long pix = i / c8;
    const long opix = pix;
    const int x_ = (int)(pix % ow);
    pix /= ow;
    const int y_ = (int)(pix % oh);
    const int img = (int)(pix / oh);
    const int off = (cc >> 5) * 64 + (cc & 31);
    float best[8];
    f16x8 bh, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      best[e] = -FLT_MAX;
      bh[e] = (_Float16)0;
      bl[e] = (_Float16)0;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = y_ * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = x_ * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)w) continue;
        const _Float16* src = x + (((long)img * h + iy) * w + ix) * (2 * c) + off;
        const f16x8 hi = *reinterpret_cast<const f16x8*>(src);
        const f16x8 lo = *reinterpret_cast<const f16x8*>(src + 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (float)hi[e] + (float)lo[e];
          if (v > best[e] || v != v) {   // (a NaN wins and stays: torch max_pool2d propagates it)
            best[e] = v;
            bh[e] = hi[e];
            bl[e] = lo[e];
          }
        }
      }
    }
    _Float16* dst = y + opix * (2 * c) + off;
    *reinterpret_cast<f16x8*>(dst) = bh;
    *reinterpret_cast<f16x8*>(dst + 32) = bl;
  }
}

int grid_for(long total) {
  const long g = (total + 255) / 256;
  return (int)(g < 16384 ? (g > 0 ? g : 1) : 16384);
}

}  // namespace

extern "C" int hn_affine_split_f32(const float* x, const float* scale, const float* shift, int relu, int n, int hw,
                                   int c, int in_pix_stride, int affine_stride, void* y16, int out_pix_stride,
                                   void* stream) {
  HN_CHECK_ARG(x && y16, "hn_affine_split_f32: null pointer");
  HN_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale and shift must be given together");
  HN_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 32 == 0, "bad dims (c must be a multiple of 32)");
  const int xs = in_pix_stride ? in_pix_stride : c;
  const int as = affine_stride ? affine_stride : c;
  const int ys = out_pix_stride ? out_pix_stride : 2 * c;
  HN_CHECK_ARG(xs >= c && xs % 4 == 0 && as >= c && as % 4 == 0 && ys >= 2 * c && ys % 64 == 0, "bad strides");
  const long npix = (long)n * hw;
  const int c8 = c / 8;
  if ((c8 & (c8 - 1)) == 0 && c8 <= 256 && n <= 65535 && !hn::env_flags().split_generic) {
    int lg = 0;
    while ((1 << lg) < c8) ++lg;
    const int ppb = 256 >> lg;
    // enough blocks per image to fill the chip a few times over, two pixels per lane and iteration
    int gx = hn::cdiv(hw, 2 * ppb);  // (the grid size is not sensitive: 64 .. 512 blocks per image within 2 %)
    const int cap = hn::cdiv(16384, n);
    gx = gx < cap ? gx : cap;
    gx = gx > 0 ? gx : 1;
    // Streaming stores + four pixels in flight for tensors that no cache level keeps anyway (level 0: 382 -> 313 us, level 1:
    // 89 -> 73 us at batch 32); plain stores for small ones, which the next convolution finds in L2 / Infinity Cache
    // (sweep of NT bits x pixels in flight, profiles/r02_apply_pass_sweep.txt; streaming LOADS cost 8 % everywhere).  In the pipeline: +0.9 % at batch 32 over the
    // generic kernel, the same with either store form (tools/probes/exp/apply_ab.sh).
    const bool big = (int64_t)npix * c * 4 >= ((int64_t)128 << 20);
#define HN_SPLIT_LAUNCH(AFF, NTV, UNRV)                                                                                          \
  hipLaunchKernelGGL((affine_split_pow2_kernel<AFF, NTV, UNRV>), dim3(gx, n), dim3(256), 0, (hipStream_t)stream, x, scale, shift, \
                     relu, hw, lg, xs, as, (_Float16*)y16, ys, hn::range_flag_ptr())
    if (scale) {
      if (big) HN_SPLIT_LAUNCH(true, 2, 4); else HN_SPLIT_LAUNCH(true, 0, 2);
    } else {
      if (big) HN_SPLIT_LAUNCH(false, 2, 4); else HN_SPLIT_LAUNCH(false, 0, 2);
    }
#undef HN_SPLIT_LAUNCH
    HN_CHECK_LAUNCH("affine_split_pow2_kernel");
    return HN_OK;
  }
  hipLaunchKernelGGL(affine_split_kernel, dim3(grid_for(npix * (c / 8))), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, relu, npix, hw, c, xs, as, (_Float16*)y16, ys, hn::range_flag_ptr());
  HN_CHECK_LAUNCH("affine_split_kernel");
  return HN_OK;
}

extern "C" int hn_affine_split_f32_levels(const hn_split_levels* lv, int relu, int n, int c, int in_pix_stride,
                                          int affine_stride, int out_pix_stride, void* stream) {
  HN_CHECK_ARG(lv, "hn_affine_split_f32_levels: null pointer");
  HN_CHECK_ARG(lv->count >= 1 && lv->count <= HN_FCOS_MAX_LEVELS, "level count must be 1..%d", HN_FCOS_MAX_LEVELS);
  HN_CHECK_ARG(n > 0 && c > 0 && c % 32 == 0, "bad dims (c must be a multiple of 32)");
  const int xs = in_pix_stride ? in_pix_stride : c;
  const int as = affine_stride ? affine_stride : c;
  const int ys = out_pix_stride ? out_pix_stride : 2 * c;
  HN_CHECK_ARG(xs >= c && xs % 4 == 0 && as >= c && as % 4 == 0 && ys >= 2 * c && ys % 64 == 0, "bad strides");
  const int c8 = c / 8;
  bool merged = (c8 & (c8 - 1)) == 0 && c8 <= 256 && n <= 65535 && !hn::env_flags().split_generic;
  for (int l = 0; l < lv->count; ++l) {
    HN_CHECK_ARG(lv->x[l] && lv->y16[l] && lv->scale[l] && lv->shift[l] && lv->hw[l] > 0, "level %d: null pointer or hw <= 0", l);
    // tensors no cache keeps take the streaming-store form of the per-level launch (see hn_affine_split_f32)
    if ((int64_t)n * lv->hw[l] * c * 4 >= ((int64_t)128 << 20)) merged = false;
  }
  if (!merged) {
    for (int l = 0; l < lv->count; ++l) {
      const int rc = hn_affine_split_f32(lv->x[l], lv->scale[l], lv->shift[l], relu, n, lv->hw[l], c, in_pix_stride,
                                         affine_stride, lv->y16[l], out_pix_stride, stream);
      if (rc != HN_OK) return rc;
    }
    return HN_OK;
  }
  int lg = 0;
  while ((1 << lg) < c8) ++lg;
  const int ppb = 256 >> lg;
  SplitLevelTable lt;
  lt.count = lv->count;
  lt.first_block[0] = 0;
  const int cap = hn::cdiv(16384, n);
  for (int l = 0; l < HN_FCOS_MAX_LEVELS; ++l) {
    const bool on = l < lv->count;
    int gx = on ? hn::cdiv(lv->hw[l], 2 * ppb) : 0;
    gx = gx < cap ? gx : cap;
    gx = on && gx < 1 ? 1 : gx;
    lt.hw[l] = on ? lv->hw[l] : 0;
    lt.first_block[l + 1] = lt.first_block[l] + gx;
    lt.x[l] = on ? lv->x[l] : nullptr;
    lt.scale[l] = on ? lv->scale[l] : nullptr;
    lt.shift[l] = on ? lv->shift[l] : nullptr;
    lt.y[l] = on ? (_Float16*)lv->y16[l] : nullptr;
  }
  hipLaunchKernelGGL(affine_split_pow2_levels_kernel, dim3(lt.first_block[lv->count], n), dim3(256), 0, (hipStream_t)stream, lt,
                     relu, lg, xs, as, ys, hn::range_flag_ptr());
  HN_CHECK_LAUNCH("affine_split_pow2_levels_kernel");
  return HN_OK;
}

extern "C" int hn_unsplit_f32(const void* x16, int n, int hw, int c, int in_pix_stride, float* y, int out_pix_stride,
                              void* stream) {
  HN_CHECK_ARG(x16 && y, "hn_unsplit_f32: null pointer");
  HN_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 32 == 0, "bad dims (c must be a multiple of 32)");
  const int xs = in_pix_stride ? in_pix_stride : 2 * c;
  const int ys = out_pix_stride ? out_pix_stride : c;
  HN_CHECK_ARG(xs >= 2 * c && xs % 64 == 0 && ys >= c && ys % 4 == 0, "bad strides");
  const long npix = (long)n * hw;
  hipLaunchKernelGGL(unsplit_kernel, dim3(grid_for(npix * (c / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x16, npix, c, xs, y, ys);
  HN_CHECK_LAUNCH("unsplit_kernel");
  return HN_OK;
}

extern "C" int hn_maxpool3x3s2_s32(const void* x16, void* y16, int n, int h, int w, int c, int oh, int ow,
                                   void* stream) {
  HN_CHECK_ARG(x16 && y16, "hn_maxpool3x3s2_s32: null pointer");
  HN_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && c % 32 == 0, "bad dims (c must be a multiple of 32)");
  HN_CHECK_ARG(oh == (h + 2 - 3) / 2 + 1 && ow == (w + 2 - 3) / 2 + 1, "output size mismatch");
  const long total = (long)n * oh * ow * (c / 8);
  hipLaunchKernelGGL(maxpool_s32_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)x16, (_Float16*)y16, n, h, w, c, oh, ow);
  HN_CHECK_LAUNCH("maxpool_s32_kernel");
  return HN_OK;
}
