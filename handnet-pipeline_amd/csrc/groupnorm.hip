// GroupNorm statistics -> per-(image, channel) scale/shift tables.
//
// The FCOS towers are conv3x3 -> GroupNorm(32,256) -> ReLU (fcos_utils/fcos.py:232-239,
// 352-359).  The statistics span a whole feature level, so they cannot be folded into
// the conv; instead the NEXT conv applies relu(x*scale + shift) while it stages its
// input (hn_conv_desc.in_affine).  This file computes the tables:
//   pass 1  gn_partial : one workgroup per (64-pixel chunk, image): fp32 sum / sumsq per
//                        group, written to a [n][chunks][G][2] slab (no atomics ->
//                        bitwise reproducible)
//   pass 2  gn_finalize: one wave per (image, group): combine the chunk partials in fp64 in
//                        fixed order, mean / biased var / rstd, then
//                        scale = gamma*rstd, shift = beta - mean*scale.
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRowsPerChunk = 64;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int hw, int c, int groups,
                                                         int chunks, float* __restrict__ partial) {
  __shared__ float lds[2 * 256];
  const int chunk = blockIdx.x, img = blockIdx.y;
  const int cols = c >> 2;            // float4 columns; 256 % cols == 0 (checked on host)
  const int rows_par = 256 / cols;
  const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
  const int r0 = chunk * kRowsPerChunk;
  const int r1 = min(r0 + kRowsPerChunk, hw);
  float s = 0.f, ss = 0.f;
  for (int r = r0 + rl; r < r1; r += rows_par) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((long)img * hw + r) * c + col * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += v[e];
      ss += v[e] * v[e];
    }
  }
  lds[threadIdx.x] = s;
  lds[256 + threadIdx.x] = ss;
  __syncthreads();
  const int cpg4 = (c / groups) >> 2;  // float4 columns per group
  if ((int)threadIdx.x < groups) {
    const int g = threadIdx.x;
    float ts = 0.f, tss = 0.f;
    for (int rr = 0; rr < rows_par; ++rr)
      for (int q = 0; q < cpg4; ++q) {
        ts += lds[rr * cols + g * cpg4 + q];
        tss += lds[256 + rr * cols + g * cpg4 + q];
      }
    float* o = partial + (((long)img * chunks + chunk) * groups + g) * 2;
    o[0] = ts;
    o[1] = tss;
  }
}

// one wave per (image, group): lanes stride over the chunk partials (fixed assignment ->
// bitwise reproducible), fp64 butterfly reduction, then lanes 0..cpg-1 write the group's
// per-channel scale / shift.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int hw, int c,
                                                          int groups, int chunks, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int cpg = c / groups;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    const float* pp = partial + (((long)img * chunks + k) * groups + g) * 2;
    s += (double)pp[0];
    ss += (double)pp[1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  const double cnt = (double)hw * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int e = lane; e < cpg; e += 64) {
    const int ch = g * cpg + e;
    const float sc = gamma[ch] * rstd;
    scale[(long)img * c + ch] = sc;
    shift[(long)img * c + ch] = beta[ch] - (float)mean * sc;
  }
}

// Finalize from the partial sums a conv epilogue wrote (hn_conv2d_nhwc_f16x3_gn): record (rg, unit) holds
// {sum, sumsq} of the rows of 32-row group rg that belong to image (32*rg)/hw, then those of the next image.
// One wave per (image, group); fixed lane assignment + fp64 butterfly -> bitwise reproducible.
__global__ __launch_bounds__(256) void gn_finalize_rows32_kernel(const float* __restrict__ partial,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int hw, int c,
                                                                 int groups, float eps, float* __restrict__ scale,
                                                                 float* __restrict__ shift) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int cpg = c / groups, upg = cpg >> 3, units = c >> 3;
  const long row0 = (long)img * hw, row1 = row0 + hw - 1;
  const int rg0 = (int)(row0 >> 5), rg1 = (int)(row1 >> 5);
  const int items = (rg1 - rg0 + 1) * upg;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < items; k += 64) {
    const int rg = rg0 + k / upg, u = g * upg + k % upg;
    const int img_a = (int)(((long)rg << 5) / hw);
    const float* pp = partial + ((long)rg * units + u) * 4 + (img_a == img ? 0 : 2);
    s += (double)pp[0];
    ss += (double)pp[1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  const double cnt = (double)hw * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int e = lane; e < cpg; e += 64) {
    const int ch = g * cpg + e;
    const float sc = gamma[ch] * rstd;
    scale[(long)img * c + ch] = sc;
    shift[(long)img * c + ch] = beta[ch] - (float)mean * sc;
  }
}

// The same finalize for all FPN levels of one tower layer in ONE launch (blockIdx.z = level): at small batch the three
// per-level launches are 4.5 us each of pure launch latency (profiles/r03_b1_timeline.txt).
struct GnLevelTable {
  int count;
  int hw[HN_FCOS_MAX_LEVELS];
  const float* partial[HN_FCOS_MAX_LEVELS];
  float* scale[HN_FCOS_MAX_LEVELS];
  float* shift[HN_FCOS_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void gn_finalize_rows32_levels_kernel(const GnLevelTable lt, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, int c, int groups,
                                                                        float eps) {
  const int lvl = blockIdx.z;
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= groups) return;
  // constant-index selects: a dynamic index into the by-value table would send it through scratch memory
  int hw = lt.hw[0];
  const float* partial = lt.partial[0];
  float *scale = lt.scale[0], *shift = lt.shift[0];
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l == lvl) {
      hw = lt.hw[l];
      partial = lt.partial[l];
      scale = lt.scale[l];
      shift = lt.shift[l];
    }
  const int cpg = c / groups, upg = cpg >> 3, units = c >> 3;
  const long row0 = (long)img * hw, row1 = row0 + hw - 1;
  const int rg0 = (int)(row0 >> 5), rg1 = (int)(row1 >> 5);
  const int items = (rg1 - rg0 + 1) * upg;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < items; k += 64) {   // same lane assignment and order as gn_finalize_rows32_kernel: same bits
    const int rg = rg0 + k / upg, u = g * upg + k % upg;
    const int img_a = (int)(((long)rg << 5) / hw);
    const float* pp = partial + ((long)rg * units + u) * 4 + (img_a == img ? 0 : 2);
    s += (double)pp[0];
    ss += (double)pp[1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    ss += __shfl_xor(ss, o);
  }
  const double cnt = (double)hw * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int e = lane; e < cpg; e += 64) {
    const int ch = g * cpg + e;
    const float sc = gamma[ch] * rstd;
    scale[(long)img * c + ch] = sc;
    shift[(long)img * c + ch] = beta[ch] - (float)mean * sc;
  }
}

}  // namespace

extern "C" int hn_groupnorm_finalize_rows32_levels(const hn_gn_levels* lv, const float* gamma, const float* beta, int n,
                                                   int c, int groups, float eps, void* stream) {
  HN_CHECK_ARG(lv && gamma && beta, "hn_groupnorm_finalize_rows32_levels: null pointer");
  HN_CHECK_ARG(lv->count >= 1 && lv->count <= HN_FCOS_MAX_LEVELS, "level count must be 1..%d", HN_FCOS_MAX_LEVELS);
  HN_CHECK_ARG(n > 0 && c > 0 && groups > 0 && c % groups == 0, "bad dims");
  HN_CHECK_ARG((c / groups) % 8 == 0, "channels per group (%d) must be a multiple of 8", c / groups);
  GnLevelTable lt;
  lt.count = lv->count;
  for (int l = 0; l < HN_FCOS_MAX_LEVELS; ++l) {
    const bool on = l < lv->count;
    if (on) HN_CHECK_ARG(lv->partial[l] && lv->scale[l] && lv->shift[l] && lv->hw[l] >= 32, "level %d: null pointer or hw < 32", l);
    lt.hw[l] = on ? lv->hw[l] : 32;
    lt.partial[l] = on ? lv->partial[l] : nullptr;
    lt.scale[l] = on ? lv->scale[l] : nullptr;
    lt.shift[l] = on ? lv->shift[l] : nullptr;
  }
  hipLaunchKernelGGL(gn_finalize_rows32_levels_kernel, dim3((groups + 3) / 4, n, lv->count), dim3(256), 0, (hipStream_t)stream,
                     lt, gamma, beta, c, groups, eps);
  HN_CHECK_LAUNCH("gn_finalize_rows32_levels_kernel");
  return HN_OK;
}

extern "C" int64_t hn_groupnorm_rows32_scratch_floats(int64_t rows, int c) {
  if (rows <= 0 || c <= 0) return 0;
  return ((rows + 31) / 32) * (c / 8) * 4;
}

extern "C" int hn_groupnorm_finalize_rows32(const float* partial, const float* gamma, const float* beta, int n, int hw,
                                            int c, int groups, float eps, float* scale, float* shift, void* stream) {
  HN_CHECK_ARG(partial && gamma && beta && scale && shift, "hn_groupnorm_finalize_rows32: null pointer");
  HN_CHECK_ARG(n > 0 && hw >= 32 && c > 0 && groups > 0 && c % groups == 0, "bad dims (hw must be >= 32)");
  HN_CHECK_ARG((c / groups) % 8 == 0, "channels per group (%d) must be a multiple of 8", c / groups);
  hipLaunchKernelGGL(gn_finalize_rows32_kernel, dim3((groups + 3) / 4, n), dim3(256), 0, (hipStream_t)stream, partial,
                     gamma, beta, hw, c, groups, eps, scale, shift);
  HN_CHECK_LAUNCH("gn_finalize_rows32_kernel");
  return HN_OK;
}

extern "C" int64_t hn_groupnorm_scratch_floats(int n, int hw, int c, int groups) {
  (void)c;
  return (int64_t)n * hn::cdiv(hw, kRowsPerChunk) * groups * 2;
}

extern "C" int hn_groupnorm_affine_f32(const float* x, const float* gamma, const float* beta, int n, int hw, int c,
                                       int groups, float eps, float* partial, float* scale, float* shift,
                                       void* stream) {
  HN_CHECK_ARG(x && gamma && beta && partial && scale && shift, "hn_groupnorm_affine_f32: null pointer");
  HN_CHECK_ARG(n > 0 && hw > 0 && c > 0 && groups > 0 && c % groups == 0, "bad dims");
  HN_CHECK_ARG((c / groups) % 4 == 0, "channels per group (%d) must be a multiple of 4", c / groups);
  HN_CHECK_ARG(c <= 1024 && 256 % (c / 4) == 0, "c/4 (%d) must divide 256", c / 4);
  HN_CHECK_ARG(groups <= 256, "too many groups");
  const int chunks = hn::cdiv(hw, kRowsPerChunk);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, n), dim3(256), 0, st, x, hw, c, groups, chunks, partial);
  HN_CHECK_LAUNCH("gn_partial_kernel");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((groups + 3) / 4, n), dim3(256), 0, st, partial, gamma, beta, hw, c,
                     groups, chunks, eps, scale, shift);
  HN_CHECK_LAUNCH("gn_finalize_kernel");
  return HN_OK;
}
