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

// ---------------------------------------------------------------------------------------------------------------------------
// One Chebyshev graph convolution (K = 3) as ONE kernel: basis -> Linear (+ folded BatchNorm) -> ReLU (-> block residual +
// vertex up-sampling).  The lifter is launch-bound at the live caller's batch (76 launches of ~10 us for ~0.3 GFLOP), so the
// three launches of a layer (spmm, basis, 1x1 convolution) and the residual pass behind a block become one:
//   phase 1  a tile of 16 rows (row = sample * V + vertex; 1024 threads: one (row, 4 features) item each at 256 features) of the basis [x0 | L x0 | (2 L L - I) x0 | 0] is gathered from x
//            (fp32, L2-resident: at most 1152 x 256 floats per sample) into LDS as fp16 hi / lo planes.  The second-order
//            term uses the matrix 2 L L - I precomputed at load (fp64 product, ~19 entries per row instead of a second
//            dependent gather over 7 x 7): the same polynomial as cheby_graph_conv.py:28-31, rounded once per coefficient.
//   phase 2  [16 x K] x [K x Fout] on the f16 MFMA in the split form (lo*hi + hi*lo + hi*hi per k tile, fp32 accumulate), the
//            filter fragments read straight from global memory, eight k tiles requested at a time (<= 768 x 256 values,
//            L2-resident), A fragments from LDS.  A workgroup owns at most four 16-column tiles (blockIdx.y = column group);
//            its sixteen waves are (column tile, k slice) pairs: 4 x 4 from 64 outputs on, 1 x 16 for the 3-channel output
//            layer; slices are added in slice order.
//   epilogue bias, ReLU, optional residual = linear interpolation of the block input along the FEATURE axis
//            (meshnet.py:105-113, ATen's align_corners = False source index), `up` copies of the row (nearest vertex
//            up-sampling), fp32 or S32 stores.
// ---------------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kGcRows = 16;      // rows per workgroup (one 16-row MFMA tile)
constexpr int kGcPad = 8;        // halves of padding per LDS row: consecutive rows start 16 bytes further along the banks
constexpr int kGcWaves = 16;     // 1024 threads: one basis item per thread at Fin = 256, one (column tile, k slice) per wave
constexpr int kGcKBatch = 8;     // k tiles whose filter fragments are requested before the first of them is used

struct GraphConvParams {
  const int* l_ptr; const int* l_idx; const float* l_val;       // L            (CSR, V rows)
  const int* q_ptr; const int* q_idx; const float* q_val;       // 2 L L - I    (CSR, V rows)
  const float* x;          // [rows][Fin] fp32
  const _Float16* w16;     // [Fout][K / 32][2][32], or in fragment order [ceil(Fout / 16)][K / 32][2][64 lanes][8] (w_frag)
  const float* bias;       // [Fout] or null
  const float* xin;        // [rows][Fi] block input for the residual, or null
  void* y;                 // fp32 [rows * up][Fout], or S32 [rows * up][Fout / 32][2][32]
  int rows, V, Fin, Fout, K, Fi, relu, up, out_split;
  int w_frag;              // 1: the filter bank is stored in MFMA fragment order (a wave's load is one contiguous KB)
  int nt_pow2;             // column tiles of ONE workgroup (a power of two <= 4; blockIdx.y = column group); k slices = 16 / nt_pow2
  int* range_flag;
};

// One round of the gathers of a basis row: NL neighbours of L from entry le and NQ neighbours of 2 L L - I from entry qe, all
// their loads issued together (first every index / coefficient, then every x row), then accumulated in CSR order -- a1 += over
// the L entries, a2 += over the others.  Written as load -> use per neighbour, every one of the ~26 gathers of a row waited
// for its own trip to the L2 (the first version of this kernel: 110 us per layer, whatever the graph size).
template <int NL, int NQ>
__device__ __forceinline__ void gather_round(const GraphConvParams& p, int le, int le1, int qe, int qe1, const float* __restrict__ xb,
                                             f32x4& a1, f32x4& a2) {
  int id[NL + NQ];
  float w[NL + NQ];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const bool in = le + u < le1;
    id[u] = in ? p.l_idx[le + u] : 0;
    w[u] = in ? p.l_val[le + u] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const bool in = qe + u < qe1;
    id[NL + u] = in ? p.q_idx[qe + u] : 0;
    w[NL + u] = in ? p.q_val[qe + u] : 0.f;
  }
  f32x4 xv[NL + NQ];
#pragma unroll
  for (int u = 0; u < NL + NQ; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xb + (long)id[u] * p.Fin);
#pragma unroll
  for (int u = 0; u < NL; ++u)
    if (le + u < le1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a1[q] += w[u] * xv[u][q];
    }
#pragma unroll
  for (int u = 0; u < NQ; ++u)
    if (qe + u < qe1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a2[q] += w[NL + u] * xv[NL + u][q];
    }
}

// a1 = (L x)[v], a2 = ((2 L L - I) x)[v] for the 4 features at xb: the four row pointers in one trip, then rounds of 8 + 8
// neighbours where the metadata is wave-uniform (scalar registers), 4 + 8 where every lane holds its own (a mesh vertex has ~7
// entries in L and ~19 in 2 L L - I: two or three rounds)
template <int NL, int NQ>
__device__ __forceinline__ void gather_basis(const GraphConvParams& p, int v, const float* __restrict__ xb, f32x4& a1, f32x4& a2) {
  int le = p.l_ptr[v], qe = p.q_ptr[v];
  const int le1 = p.l_ptr[v + 1], qe1 = p.q_ptr[v + 1];
  while (le < le1 || qe < qe1) {
    gather_round<NL, NQ>(p, le, le1, qe, qe1, xb, a1, a2);
    le += NL;
    qe += NQ;
  }
}

__global__ __launch_bounds__(kGcWaves * 64) void graph_conv_fused_kernel(const GraphConvParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 gc_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldk = p.K + kGcPad;
  _Float16* hi_pl = gc_lds;
  _Float16* lo_pl = gc_lds + kGcRows * ldk;
  float* red = reinterpret_cast<float*>(gc_lds + 2 * kGcRows * ldk);   // [16 waves][64 lanes][4]: k-slice partial tiles
  const int row0 = blockIdx.x * kGcRows;

  // the epilogue's operands depend on nothing the kernel computes: requested now, they arrive under phase 1 (fetched in the
  // epilogue they were two more dependent trips to memory at the end of every launch)
  const int nt_all = (p.Fout + 15) >> 4;
  const int nt = blockIdx.y * p.nt_pow2 + (wave & (p.nt_pow2 - 1)), ks = wave / p.nt_pow2;
  const int lr = lane & 15, lg = lane >> 4;
  const int col = nt * 16 + lr;
  const bool live = nt < nt_all && col < p.Fout;
  float bias = 0.f, res[4] = {0.f, 0.f, 0.f, 0.f};
  if (live && ks == 0) {
    if (p.bias) bias = p.bias[col];
    if (p.xin) {   // ATen area_pixel_compute_source_index, align_corners = False (as feat_interp_add_kernel)
      const float scale = (float)p.Fi / (float)p.Fout;
      float src = fmaf(scale, (float)col + 0.5f, -0.5f);
      src = src < 0.f ? 0.f : src;
      const int i0 = (int)src;
      const int i1 = i0 + (i0 < p.Fi - 1 ? 1 : 0);
      const float w1 = src - (float)i0, w0 = 1.f - w1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + lg * 4 + r;
        if (row < p.rows) {
          const float* xr = p.xin + (long)row * p.Fi;
          res[r] = w0 * xr[i0] + w1 * xr[i1];
        }
      }
    }
  }

  // ---- phase 1: the basis rows of this tile ----
  const int F4 = p.Fin >> 2;
  for (int i = tid; i < kGcRows * F4; i += kGcWaves * 64) {
    const int r = i / F4, c = i - r * F4;
    const int row = row0 + r;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
    if (row < p.rows) {
      const int b = row / p.V;
      int v = row - b * p.V;
      // 256 features: a wave's 64 items are ONE row, so its CSR metadata is wave-uniform -- said to the compiler, which then
      // fetches it with scalar loads (no vector registers, sixteen indices per instruction)
      const float* xb = p.x + (long)b * p.V * p.Fin + c * 4;
      if (F4 == 64) {
        v = __builtin_amdgcn_readfirstlane(v);
        a0 = *reinterpret_cast<const f32x4*>(xb + (long)v * p.Fin);
        gather_basis<8, 8>(p, v, xb, a1, a2);
      } else {
        a0 = *reinterpret_cast<const f32x4*>(xb + (long)v * p.Fin);
        gather_basis<4, 8>(p, v, xb, a1, a2);
      }
    }
    const f32x4* parts[3] = {&a0, &a1, &a2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f16x4 h, l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float val = (*parts[t])[q];
        h[q] = (_Float16)val;
        l[q] = (_Float16)(val - (float)h[q]);
      }
      const int k = t * p.Fin + c * 4;
      *reinterpret_cast<f16x4*>(hi_pl + r * ldk + k) = h;
      *reinterpret_cast<f16x4*>(lo_pl + r * ldk + k) = l;
    }
  }
  const int padk = p.K - 3 * p.Fin;      // zero channels behind the basis (K is a multiple of 32)
  for (int i = tid; i < kGcRows * (padk >> 2); i += kGcWaves * 64) {
    const int r = i / (padk >> 2), c = i - r * (padk >> 2);
    const f16x4 z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
    *reinterpret_cast<f16x4*>(hi_pl + r * ldk + 3 * p.Fin + c * 4) = z;
    *reinterpret_cast<f16x4*>(lo_pl + r * ldk + 3 * p.Fin + c * 4) = z;
  }
  __syncthreads();

  // ---- phase 2: [16 x K] x [K x Fout]; wave = (column tile nt, k slice ks) ----
  const int slices = kGcWaves / p.nt_pow2;
  const int ktiles = p.K >> 5;
  const int per = (ktiles + slices - 1) / slices;
  const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f16x8 zero8 = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
  if (nt < nt_all) {
    // standard layout: lane (column lr, k chunk lg) reads 16 bytes of its column's 64-byte run -- 16 runs of 64 bytes per wave
    // load; fragment order: the wave's 64 x 16 bytes are one contiguous KB (half the cache lines through the CU's 64 B/clk path)
    const _Float16* wq = p.w_frag ? p.w16 + ((long)nt * ktiles * 2 * 64 + lane) * 8
                                  : p.w16 + ((long)(live ? col : 0) * ktiles * 2) * 32 + lg * 8;
    const long kstep = p.w_frag ? 2 * 64 * 8 : 64, lo_off = p.w_frag ? 64 * 8 : 32;
    const bool have = p.w_frag ? true : live;        // (fragment order holds zero columns behind Fout)
    for (int kb = kt0; kb < kt1; kb += kGcKBatch) {
      f16x8 bh[kGcKBatch], bl[kGcKBatch];
#pragma unroll
      for (int u = 0; u < kGcKBatch; ++u) {
        const bool in = have && kb + u < kt1;
        bh[u] = in ? *reinterpret_cast<const f16x8*>(wq + (long)(kb + u) * kstep) : zero8;
        bl[u] = in ? *reinterpret_cast<const f16x8*>(wq + (long)(kb + u) * kstep + lo_off) : zero8;
      }
#pragma unroll
      for (int u = 0; u < kGcKBatch; ++u) {
        if (kb + u < kt1) {
          const int off = lr * ldk + (kb + u) * 32 + lg * 8;
          const f16x8 ah = *reinterpret_cast<const f16x8*>(hi_pl + off);
          const f16x8 al = *reinterpret_cast<const f16x8*>(lo_pl + off);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], acc, 0, 0, 0);
        }
      }
    }
  }
  if (slices > 1) {   // the k slices of a column tile are added in slice order by the wave of slice 0
    *reinterpret_cast<f32x4*>(red + ((long)wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (ks == 0)
      for (int s2 = 1; s2 < slices; ++s2) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((long)(s2 * p.nt_pow2 + (wave & (p.nt_pow2 - 1))) * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += o[q];
      }
  }
  if (ks != 0 || !live) return;

  // ---- epilogue: D[row = lg * 4 + r][col] ----
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + lg * 4 + r;
    if (row >= p.rows) continue;
    float v = acc[r] + bias;
    if (p.relu) v = hn::relu(v);
    if (p.xin) v = v + res[r];
    for (int u = 0; u < p.up; ++u) {
      const long orow = (long)row * p.up + u;
      if (p.out_split) {
        _Float16* o = (_Float16*)p.y + orow * 2 * p.Fout + (col >> 5) * 64 + (col & 31);
        const _Float16 h = (_Float16)v;
        o[0] = h;
        o[32] = (_Float16)(v - (float)h);
        bad |= hn::range_bad(v);
      } else {
        ((float*)p.y)[orow * p.Fout + col] = v;
      }
    }
  }
  if (bad && p.range_flag) *p.range_flag = 1;
}

}  // namespace

extern "C" int hn_graph_conv_cheby3_f16x3(const hn_graph_csr* L, const hn_graph_csr* L2, const float* x, int batch, int fin,
                                          const void* w16, int w_frag, const float* bias, int fout, int relu, const float* xin,
                                          int fi, int up, void* y, int out_split, void* stream) {
  HN_CHECK_ARG(L && L2 && x && w16 && y, "hn_graph_conv_cheby3_f16x3: null pointer");
  HN_CHECK_ARG(L->indptr && L->indices && L->values && L2->indptr && L2->indices && L2->values, "null CSR arrays");
  HN_CHECK_ARG(L->v > 0 && L2->v == L->v && batch > 0, "bad graph / batch");
  HN_CHECK_ARG(fin >= 4 && fin % 4 == 0 && fin <= 256, "Fin must be a multiple of 4 in [4, 256]");
  HN_CHECK_ARG(fout >= 1 && fout <= 16 * kGcWaves, "Fout must be in [1, 256]");
  HN_CHECK_ARG(up >= 1 && up <= 4 && (!xin || fi > 0), "bad residual / up-sampling arguments");
  HN_CHECK_ARG(!out_split || fout % 32 == 0, "the S32 output needs Fout % 32 == 0");
  HN_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)w16 % 16 == 0, "unaligned operands");
  GraphConvParams p;
  p.l_ptr = L->indptr; p.l_idx = L->indices; p.l_val = L->values;
  p.q_ptr = L2->indptr; p.q_idx = L2->indices; p.q_val = L2->values;
  p.x = x; p.w16 = (const _Float16*)w16; p.bias = bias; p.xin = xin; p.y = y;
  p.rows = batch * L->v; p.V = L->v; p.Fin = fin; p.Fout = fout; p.K = (3 * fin + 31) / 32 * 32; p.Fi = fi;
  p.relu = relu ? 1 : 0; p.up = up; p.out_split = out_split ? 1 : 0; p.w_frag = w_frag ? 1 : 0;
  p.range_flag = out_split ? hn::range_flag_ptr() : nullptr;
  // column tiles per workgroup: at most 4 (a CU pulls its workgroup's share of the filter bank through ONE 64 B/clk vector
  // memory path: the whole 768 x 256 bank per workgroup was 17 of a layer's 25 us; with four column groups every workgroup
  // repeats the basis gather -- 3.5 us, in parallel on other CUs -- and reads a quarter of the bank, its sixteen waves
  // sharing the k loop four ways)
  p.nt_pow2 = 1;
  while (p.nt_pow2 * 16 < fout && p.nt_pow2 < 4) p.nt_pow2 *= 2;
  const int col_groups = (fout + 16 * p.nt_pow2 - 1) / (16 * p.nt_pow2);
  const size_t lds = (size_t)2 * kGcRows * (p.K + kGcPad) * sizeof(_Float16) + (size_t)kGcWaves * 64 * 4 * sizeof(float);
  static bool attr_set[64] = {};    // (> 64 KB of dynamic LDS needs the attribute once per device)
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)graph_conv_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(graph_conv_fused_kernel, dim3((p.rows + kGcRows - 1) / kGcRows, col_groups), dim3(kGcWaves * 64), lds,
                     (hipStream_t)stream, p);
  HN_CHECK_LAUNCH("graph_conv_fused_kernel");
  return HN_OK;
}

namespace {
// fp32 rows [rows][f] -> S32 rows of cpad channels (zero padding behind f): the operand of the lifter's first Linear
__global__ __launch_bounds__(256) void pad_split_rows_kernel(const float* __restrict__ x, _Float16* __restrict__ out16, long rows,
                                                             int f, int cpad) {
  const long total = rows * cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad);
    const long r = i / cpad;
    const float v = c < f ? x[r * f + c] : 0.f;
    const _Float16 h = (_Float16)v;
    _Float16* o = out16 + r * 2 * cpad + (c >> 5) * 64 + (c & 31);
    o[0] = h;
    o[32] = (_Float16)(v - (float)h);
  }
}

// pose_combine of pose2mesh_net.py:20: [pose2d | pose3d / 1000 | 0-pad] per joint, fp32 [rows][fpad] (fpad = 8: the padded
// input of the first graph convolution)
__global__ __launch_bounds__(256) void lifter_combine_kernel(const float* __restrict__ pose2d, const float* __restrict__ pose3d,
                                                             float* __restrict__ out, long rows, int fpad, int joints,
                                                             int p3_stride) {
  const long total = rows * fpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % fpad);
    const long r = i / fpad;
    const long b = r / joints, j = r - b * joints;
    out[i] = c < 2 ? pose2d[r * 2 + c] : (c < 5 ? pose3d[b * p3_stride + j * 3 + (c - 2)] / 1000.f : 0.f);
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// A Linear layer on 1..4 rows (PoseNet at the live caller's batch: 42 -> 4096 -> 4096 x 4 -> 63): a matrix-VECTOR product, i.e.
// 67 MB of filter bank streamed once for 17 M MACs -- HBM-bound by three orders of magnitude, so the vector ALU does it: every
// wave streams whole rows of the bank (1 KB per load = 8 blocks of hi[32] | lo[32]; a lane holds eight hi OR eight lo halves and
// multiplies them with the matching fp32 activations from LDS; hi and lo partial sums meet in the wave reduction: w = hi + lo
// exactly, the activations stay fp32 -- MORE precise than the three-term MFMA form, which drops lo * lo).  The activations are
// staged in LDS once per workgroup, optionally through the pre-activation BatchNorm + ReLU of posenet.py:24-26 (scale / shift),
// so a PoseNet stage is two launches and no split pass.  Bias, fp32 residual, ReLU in the epilogue; fp32 in, fp32 out.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kLinRowsPerWave = 2, kLinMaxBatch = 4, kLinKLoads = 16;   // 16 loads x 256 channels = 4096 channels per pass

struct LinearParams {
  const float* x; int xs;            // [M][xs] fp32, k_real columns used
  const float* scale; const float* shift;   // [K] pre-activation affine (+ ReLU) applied while staging x, or null
  const _Float16* w16;               // [N][K / 32][2][32]
  const float* bias; const float* residual; int rs;   // [N] or null; [M][rs] fp32 or null
  float* y; int ys;                  // [M][ys]
  int M, K, k_real, N, relu;
};

__global__ __launch_bounds__(256, 2) void linear_rows_kernel(const LinearParams p) {
  extern __shared__ __attribute__((aligned(16))) float lin_x[];   // [M][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 3, chunk = lane & 3;   // 32-channel block of a load (lanes 4..7 of a block hold its lo halves), 8-channel chunk
  const int ktiles = p.K >> 5;
  const int n0 = (blockIdx.x * 4 + wave) * kLinRowsPerWave;
  // The filter rows do not depend on the activations: a 4096-channel pass of BOTH rows of this wave is requested first (32 loads
  // of 1 KB), the activations are staged under their latency.  (Staging first, then one row after the other: 22 us per
  // 4096 x 4096 layer, slower than the convolution kernel's 18.5 us; the bank must be in flight before anything waits.)
  f16x8 wv[kLinRowsPerWave][kLinKLoads];
  auto request = [&](int k0) {
#pragma unroll
    for (int rr = 0; rr < kLinRowsPerWave; ++rr) {
      const int n = n0 + rr;
      const _Float16* wrow = p.w16 + (long)(n < p.N ? n : 0) * ktiles * 64 + lane * 8;
#pragma unroll
      for (int u = 0; u < kLinKLoads; ++u) {
        const int kt = k0 + u * 8;             // first of the 8 blocks this load covers
        wv[rr][u] = (n < p.N && kt + blk < ktiles) ? *reinterpret_cast<const f16x8*>(wrow + (long)kt * 64) : f16x8{};
      }
    }
  };
  request(0);
  for (int i = tid; i < p.M * p.K; i += 256) {
    const int m = i / p.K, k = i - m * p.K;
    float v = k < p.k_real ? p.x[(long)m * p.xs + k] : 0.f;
    if (p.scale && k < p.k_real) v = hn::relu(v * p.scale[k] + p.shift[k]);
    lin_x[i] = v;
  }
  __syncthreads();
  float acc[kLinRowsPerWave][kLinMaxBatch];
#pragma unroll
  for (int rr = 0; rr < kLinRowsPerWave; ++rr)
#pragma unroll
    for (int m = 0; m < kLinMaxBatch; ++m) acc[rr][m] = 0.f;
  for (int k0 = 0; k0 < ktiles; k0 += 8 * kLinKLoads) {
    if (k0) request(k0);                       // (banks deeper than 4096 channels: further passes)
#pragma unroll
    for (int u = 0; u < kLinKLoads; ++u) {
      const int kt = k0 + u * 8 + blk;
      if (kt < ktiles) {
        const float* xk = lin_x + kt * 32 + chunk * 8;
#pragma unroll
        for (int m = 0; m < kLinMaxBatch; ++m)
          if (m < p.M) {
            float xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = xk[m * p.K + i];
#pragma unroll
            for (int rr = 0; rr < kLinRowsPerWave; ++rr)
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[rr][m] += (float)wv[rr][u][i] * xv[i];
          }
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < kLinRowsPerWave; ++rr) {
    const int n = n0 + rr;
    if (n >= p.N) break;                       // (wave-uniform)
#pragma unroll
    for (int m = 0; m < kLinMaxBatch; ++m) {
      if (m >= p.M) break;
      float v = acc[rr][m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) {
        if (p.bias) v += p.bias[n];
        if (p.residual) v += p.residual[(long)m * p.rs + n];
        if (p.relu) v = hn::relu(v);
        p.y[(long)m * p.ys + n] = v;
      }
    }
  }
}
}  // namespace

extern "C" int hn_linear_rows_f16x3(const float* x, int batch, int x_stride, int k_real, const float* scale, const float* shift,
                                    const void* w16, int k, int n, const float* bias, const float* residual, int res_stride,
                                    int relu, float* y, int y_stride, void* stream) {
  HN_CHECK_ARG(x && w16 && y, "hn_linear_rows_f16x3: null pointer");
  HN_CHECK_ARG(batch >= 1 && batch <= kLinMaxBatch, "hn_linear_rows_f16x3 takes 1..4 rows (larger batches: the convolution kernel)");
  HN_CHECK_ARG(k > 0 && k % 32 == 0 && k_real > 0 && k_real <= k && x_stride >= k_real && n > 0 && y_stride >= n, "bad dims");
  HN_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (!residual || res_stride >= n), "bad affine / residual arguments");
  HN_CHECK_ARG((size_t)batch * k * 4 <= 64 * 1024 && (uintptr_t)w16 % 16 == 0, "activations must fit 64 KB of LDS; aligned bank");
  LinearParams p;
  p.x = x; p.xs = x_stride; p.scale = scale; p.shift = shift; p.w16 = (const _Float16*)w16; p.bias = bias;
  p.residual = residual; p.rs = res_stride; p.y = y; p.ys = y_stride; p.M = batch; p.K = k; p.k_real = k_real; p.N = n;
  p.relu = relu ? 1 : 0;
  const int rows_per_wg = 4 * kLinRowsPerWave;
  hipLaunchKernelGGL(linear_rows_kernel, dim3((n + rows_per_wg - 1) / rows_per_wg), dim3(256), (size_t)batch * k * 4,
                     (hipStream_t)stream, p);
  HN_CHECK_LAUNCH("linear_rows_kernel");
  return HN_OK;
}

namespace {
// out['mesh'] of ros_demo.py:162,332-337: the vertices of the real mesh in original order (pred_mesh[:, graph_perm_reverse[:V]]),
// moved to the camera frame of the depth sensor -- (mesh * 1000 + joints3d[0]) / 1000 -- with y and z negated; numpy float32
// arithmetic, one rounding per operation (contraction is switched off inside the kernel)
__global__ __launch_bounds__(256) void mesh_finish_kernel(const float* __restrict__ mesh, const long long* __restrict__ perm,
                                                          const float* __restrict__ xyz_mm, const int* __restrict__ valid,
                                                          float* __restrict__ out, int n, int v0, int v, int joints) {
#pragma clang fp contract(off)   // (HIP's __fmul_rn / __fadd_rn are plain operators: without this the pair becomes one fma)
  const long total = (long)n * v * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const long r = i / 3;
    const int vv = (int)(r % v), img = (int)(r / v);
    float val = 0.f;
    if (!valid || valid[img] == 1) {
      const float m = mesh[((long)img * v0 + perm[vv]) * 3 + c];
      const float root = xyz_mm[(long)img * joints * 3 + c];               // joints3d[0]: the first joint
      const float scaled = m * 1000.f;
      const float moved = scaled + root;
      val = moved / 1000.f;
      if (c) val = -val;
    }
    out[i] = val;
  }
}
}  // namespace

extern "C" int hn_mesh_finish_f32(const float* mesh, const int64_t* perm, const float* xyz_mm, const int32_t* valid, int n, int v0,
                                  int v, int joints, float* out, void* stream) {
  HN_CHECK_ARG(mesh && perm && xyz_mm && out, "hn_mesh_finish_f32: null pointer");
  HN_CHECK_ARG(n > 0 && v0 > 0 && v > 0 && joints > 0, "bad dims");
  hipLaunchKernelGGL(mesh_finish_kernel, dim3(grid_for((long)n * v * 3)), dim3(256), 0, (hipStream_t)stream, mesh,
                     (const long long*)perm, xyz_mm, valid, out, n, v0, v, joints);
  HN_CHECK_LAUNCH("mesh_finish_kernel");
  return HN_OK;
}

extern "C" int hn_pad_split_rows_f32(const float* x, int64_t rows, int f, int cpad, void* out16, void* stream) {
  HN_CHECK_ARG(x && out16, "hn_pad_split_rows_f32: null pointer");
  HN_CHECK_ARG(rows > 0 && f > 0 && cpad >= f && cpad % 32 == 0, "bad dims (cpad: a multiple of 32 >= f)");
  hipLaunchKernelGGL(pad_split_rows_kernel, dim3(grid_for(rows * cpad)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)out16,
                     (long)rows, f, cpad);
  HN_CHECK_LAUNCH("pad_split_rows_kernel");
  return HN_OK;
}

extern "C" int hn_lifter_combine_f32(const float* pose2d, const float* pose3d, int batch, int joints, int pose3d_stride, int fpad,
                                     float* out, void* stream) {
  HN_CHECK_ARG(pose2d && pose3d && out, "hn_lifter_combine_f32: null pointer");
  HN_CHECK_ARG(batch > 0 && joints > 0 && fpad >= 5 && pose3d_stride >= 3 * joints, "bad dims");
  const long rows = (long)batch * joints;
  hipLaunchKernelGGL(lifter_combine_kernel, dim3(grid_for(rows * fpad)), dim3(256), 0, (hipStream_t)stream, pose2d, pose3d, out,
                     rows, fpad, joints, pose3d_stride);
  HN_CHECK_LAUNCH("lifter_combine_kernel");
  return HN_OK;
}

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
