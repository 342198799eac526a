// Chebyshev graph-convolution helpers of the Pose2Mesh lifter (pose2mesh/lib/models/backbones/
// cheby_graph_conv.py:5-42, meshnet.py:68-117).  The dense part of a graph conv (Linear + BatchNorm + ReLU)
// runs on the f16x3 convolution kernel as a 1x1 conv over [B][V] "pixels"; this file supplies what comes
// before and after it:
//   spmm_csr_kernel      x1 = L x0                                   (fp32, rows = (batch, vertex))
//   cheby3_basis_kernel  x2 = 2 L x1 - x0 and the split-fp16 (S32) operand [x0 | x1 | x2 | 0-pad] of the conv
//   feat_interp_add      block residual: linear interpolation along the FEATURE axis + add (+ nearest x2
//                        vertex up-sampling), meshnet.py:105-113
// All three are gather / elementwise kernels over at most 1152 x 256 values per sample: HBM/latency bound.
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// y[b][v][:] = sum_j L[v][j] * x[b][j][:]   (CSR, column indices ascending: fixed summation order)
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                       const float* __restrict__ values, const float* __restrict__ x,
                                                       float* __restrict__ y, int B, int V, int F4) {
  const long total = (long)B * V * F4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % F4);
    const long r = i / F4;
    const int v = (int)(r % V), b = (int)(r / V);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) {
      const float w = values[e];
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (((long)b * V + indices[e]) * F4 + c) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += w * xv[q];
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = acc;
  }
}

// One thread = 4 features of one (batch, vertex) row: x2 = 2*(L x1) - x0, then hi/lo fp16 of x0, x1, x2 are
// written at channels [0,F), [F,2F), [2F,3F) of the row's S32 record (32-channel blocks of hi[32] | lo[32]);
// threads with f >= F zero the padding channels [3F, Cpad).
__device__ __forceinline__ void store_split4(_Float16* row16, int ch, const f32x4& v) {
  f16x4 hi, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi[q] = (_Float16)v[q];
    lo[q] = (_Float16)(v[q] - (float)hi[q]);
  }
  _Float16* p = row16 + (ch >> 5) * 64 + (ch & 31);
  *reinterpret_cast<f16x4*>(p) = hi;
  *reinterpret_cast<f16x4*>(p + 32) = lo;
}

__global__ __launch_bounds__(256) void cheby3_basis_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                           const float* __restrict__ values, const float* __restrict__ x0,
                                                           const float* __restrict__ x1, _Float16* __restrict__ out16,
                                                           int B, int V, int F, int Cpad) {
  const int F4 = F >> 2, P4 = (Cpad - 3 * F) >> 2;  // real and padding float4 columns per row
  const int W4 = F4 + P4;
  const long total = (long)B * V * W4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W4);
    const long r = i / W4;
    _Float16* row16 = out16 + r * (long)Cpad * 2;
    if (c >= F4) {  // zero padding channels
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      store_split4(row16, 3 * F + (c - F4) * 4, z);
      continue;
    }
    const int v = (int)(r % V), b = (int)(r / V);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) {
      const float w = values[e];
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x1 + (((long)b * V + indices[e]) * F4 + c) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += w * xv[q];
    }
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(x0 + (r * F4 + c) * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(x1 + (r * F4 + c) * 4);
    f32x4 a2;
#pragma unroll
    for (int q = 0; q < 4; ++q) a2[q] = 2.f * acc[q] - a0[q];
    store_split4(row16, c * 4, a0);
    store_split4(row16, F + c * 4, a1);
    store_split4(row16, 2 * F + c * 4, a2);
  }
}

// out[(r*up + u)][j] = y[r][j] + lerp(xin[r][:], j), linear interpolation with align_corners = False from Fi to Fo
// samples (ATen area_pixel_compute_source_index: src = scale*(j+0.5)-0.5 clamped at 0, scale = Fi/Fo in fp32)
__global__ __launch_bounds__(256) void feat_interp_add_kernel(const float* __restrict__ xin, const float* __restrict__ y,
                                                              float* __restrict__ out, long rows, int Fi, int Fo, int up) {
  const float scale = (float)Fi / (float)Fo;
  const long total = rows * Fo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Fo);
    const long r = i / Fo;
    float src = fmaf(scale, (float)j + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < Fi - 1 ? 1 : 0);
    const float w1 = src - (float)i0, w0 = 1.f - w1;
    const float* xr = xin + r * Fi;
    const float val = y[i] + (w0 * xr[i0] + w1 * xr[i1]);
    for (int u = 0; u < up; ++u) out[(r * up + u) * Fo + j] = val;
  }
}

int grid_for(long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

}  // namespace

extern "C" int hn_spmm_csr_f32(const int32_t* indptr, const int32_t* indices, const float* values, int v,
                               const float* x, float* y, int batch, int f, void* stream) {
  HN_CHECK_ARG(indptr && indices && values && x && y, "hn_spmm_csr_f32: null pointer");
  HN_CHECK_ARG(v > 0 && batch > 0 && f > 0 && f % 4 == 0, "bad dims (features must be a multiple of 4)");
  const long total = (long)batch * v * (f / 4);
  hipLaunchKernelGGL(spmm_csr_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, indptr, indices, values,
                     x, y, batch, v, f / 4);
  HN_CHECK_LAUNCH("spmm_csr_kernel");
  return HN_OK;
}

extern "C" int hn_cheby3_basis_split(const int32_t* indptr, const int32_t* indices, const float* values, int v,
                                     const float* x0, const float* x1, void* out16, int batch, int f, int cpad,
                                     void* stream) {
  HN_CHECK_ARG(indptr && indices && values && x0 && x1 && out16, "hn_cheby3_basis_split: null pointer");
  HN_CHECK_ARG(v > 0 && batch > 0 && f > 0 && f % 4 == 0, "bad dims (features must be a multiple of 4)");
  HN_CHECK_ARG(cpad >= 3 * f && cpad % 32 == 0, "cpad must be a multiple of 32 and >= 3*f");
  const long total = (long)batch * v * ((cpad - 2 * f) / 4);
  hipLaunchKernelGGL(cheby3_basis_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, indptr, indices,
                     values, x0, x1, (_Float16*)out16, batch, v, f, cpad);
  HN_CHECK_LAUNCH("cheby3_basis_kernel");
  return HN_OK;
}

extern "C" int hn_feat_interp_add_f32(const float* xin, const float* y, float* out, int64_t rows, int fi, int fo, int up,
                                      void* stream) {
  HN_CHECK_ARG(xin && y && out, "hn_feat_interp_add_f32: null pointer");
  HN_CHECK_ARG(rows > 0 && fi > 0 && fo > 0 && up >= 1, "bad dims");
  hipLaunchKernelGGL(feat_interp_add_kernel, dim3(grid_for(rows * fo)), dim3(256), 0, (hipStream_t)stream, xin, y, out,
                     (long)rows, fi, fo, up);
  HN_CHECK_LAUNCH("feat_interp_add_kernel");
  return HN_OK;
}
