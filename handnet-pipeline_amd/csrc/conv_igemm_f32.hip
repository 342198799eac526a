// Implicit-GEMM convolution on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
//   y[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),  m = (img, oh, ow), n = cout,
//   k = (r, s, c) with c fastest -- A is gathered on the fly from the NHWC input
//   (zero outside the image), W is the [Cout][R][S][Cin] filter bank.
//
// Replaces the cuDNN convolutions + folded (Frozen)BatchNorm + ReLU + residual add of
// the reference's ResNet-50 trunk / A2J heads (a2j/resnet.py:61-96, a2j/a2j.py:70-181)
// and of the ResNet-34-FPN + FCOS towers (fcos_utils/fcos.py:476,737,745).
//
// Design (MI355X): 256-thread workgroups (4 waves), BM x BN output tile, BK = 16.
// Both operands are staged global -> registers -> LDS as [row][k] images with the k
// run contiguous (row pitch 20 floats: conflict-free ds_read_b128), double buffered,
// one barrier per k tile.  A lane's ds_read_b128 delivers 4 k values that feed 4
// consecutive MFMAs (lanes 0-31 carry k = g*8+j, lanes 32-63 carry k = g*8+4+j; A and
// W use the same map so the permutation of k inside a group is harmless).  The f32
// MFMA is an exact k-ordered fmaf chain, so results are fp32-faithful.
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous range of
// output-pixel tiles together with all their channel tiles.
#include "hn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* in_scale;
  const float* in_shift;
  float* y;
  int N, H, W, Cin, Cout, R, S, stride, pad, dil, OH, OW;
  int M, Ktot, ktiles;
  int relu_cols, res_mode, res_h, res_w, in_affine;
  int xs, ys;  // pixel strides (floats) of x and y: channel-slice views of wider tensors
  int as;      // row stride (floats) of the in_scale / in_shift tables
  int out_split;  // write y as an S32 split tensor (ys is then in halfs)
  int vec_epi;    // 1: 16-byte epilogue through LDS (Cout % 8 == 0 and aligned strides)
  int tiles_m, tiles_n, nblocks;
};

constexpr int BK = 16;
constexpr int LDK = BK + 4;  // LDS row pitch in floats

template <int BM, int BN, int WM, int WN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int A_IT = BM / 64;
  constexpr int B_IT = (BN + 63) / 64;
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;

  // XCD-aware bijective remap: ids congruent mod 8 share an XCD; give each XCD a
  // contiguous run of logical tiles (channel tile fastest) for L2 reuse of A halos.
  int lid;
  {
    const int bid = blockIdx.x, nb = p.nblocks;
    const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int kc = tid & 3;     // 16-byte chunk column inside the k tile
  const int lrow = tid >> 2;  // 0..63

  // ---- per-thread gather state for its A rows and W rows ----
  int a_ih0[A_IT], a_iw0[A_IT], a_img[A_IT];
  long a_base[A_IT];
  bool a_ok[A_IT];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int m = m0 + lrow + it * 64;
    a_ok[it] = m < p.M;
    const int mm = a_ok[it] ? m : 0;
    const int img = mm / ohow;
    const int rem = mm - img * ohow;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    a_img[it] = img;
    a_ih0[it] = oh * p.stride - p.pad;
    a_iw0[it] = ow * p.stride - p.pad;
    a_base[it] = (long)img * p.H * p.W * p.xs;
  }
  long b_off[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int rown = lrow + it * 64;
    const int n = n0 + rown;
    b_ok[it] = (rown < BN) && (n < p.Cout);
    b_off[it] = (long)(b_ok[it] ? n : 0) * p.Ktot + kc * 4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int cur_r = 0, cur_s = 0, cur_c = 0;  // tap / channel of the NEXT tile to load (!SMALLC)
  f32x4 ra[A_IT], rb[B_IT];

  auto gload = [&](int t) {
    int r, s, c;
    bool kvalid = true;
    if constexpr (SMALLC) {
      const int qk = t * 4 + kc;
      kvalid = qk < (p.Ktot >> 2);
      const int cpt = p.Cin >> 2;
      const int tap = qk / cpt;
      c = (qk - tap * cpt) << 2;
      r = tap / p.S;
      s = tap - r * p.S;
    } else {
      r = cur_r;
      s = cur_s;
      c = cur_c + kc * 4;
      cur_c += BK;
      if (cur_c >= p.Cin) {
        cur_c = 0;
        if (++cur_s == p.S) {
          cur_s = 0;
          ++cur_r;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int ih = a_ih0[it] + r * p.dil, iw = a_iw0[it] + s * p.dil;
      const bool ok = a_ok[it] && kvalid && (unsigned)ih < (unsigned)p.H &&
                      (unsigned)iw < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v = *reinterpret_cast<const f32x4*>(p.x + a_base[it] + ((long)ih * p.W + iw) * p.xs + c);
        if (p.in_affine) {
          const long o = (long)a_img[it] * p.as + c;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.in_scale + o);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.in_shift + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = hn::relu(v[e] * sc[e] + sh[e]);
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[it] && kvalid) v = *reinterpret_cast<const f32x4*>(p.w + b_off[it] + (long)t * BK);
      rb[it] = v;
    }
  };

  auto sstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<f32x4*>(&As[buf * BM * LDK + (lrow + it * 64) * LDK + kc * 4]) = ra[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (lrow + it * 64 < BN)
        *reinterpret_cast<f32x4*>(&Bs[buf * BN * LDK + (lrow + it * 64) * LDK + kc * 4]) = rb[it];
  };

  const int arow = wm * (BM / WM) + (lane & 31);
  const int brow = wn * (BN / WN) + (lane & 31);
  const int koff = (lane >> 5) * 4;

  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * LDK;
    const float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(&Ab[(arow + i * 32) * LDK + g * 8 + koff]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(&Bb[(brow + j * 32) * LDK + g * 8 + koff]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    }
  };

  const int T = p.ktiles;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const bool more = (t + 1 < T);
    if (more) gload(t + 1);
    compute(t & 1);
    if (more) sstore((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: bias, residual, ReLU, NHWC store ----
  constexpr bool kPatchFits = 4 * 32 * (TN * 32) <= 2 * (BM + BN) * LDK;  // 4 wave patches inside smem
  if (kPatchFits && p.vec_epi) {
    // Vector path (Cout % 8 == 0): each wave transposes its accumulators through a private LDS
    // patch (32 rows x TN*32 columns per pass) so that a lane owns 8 consecutive channels of
    // one pixel: 16-byte bias / residual reads and fp32 or S32 (hi|lo) stores.
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    constexpr int PW = TN * 32, GROUPS = PW / 8;
    __syncthreads();
    float* patch = smem + wave * (32 * PW);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j) patch[prow * PW + j * 32 + (lane & 31)] = acc[i][j][r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < (32 * GROUPS) / 64; ++k) {
        const int q = lane + 64 * k;
        const int prow = q / GROUPS, g = q - prow * GROUPS;
        const int m = m0 + wm * (BM / WM) + i * 32 + prow;
        const int n = n0 + wn * (BN / WN) + g * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(&patch[prow * PW + g * 8]);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(&patch[prow * PW + g * 8 + 4]);
        if (m >= p.M || n >= p.Cout) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = c0[e];
          v[4 + e] = c1[e];
        }
        if (p.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += b0[e];
            v[4 + e] += b1[e];
          }
        }
        if (p.res_mode) {
          long rpix = m;
          if (p.res_mode == 2) {
            const int img = m / ohow;
            const int rem = m - img * ohow;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const int sh_ = (int)(((long)oh * p.res_h) / p.OH), sw_ = (int)(((long)ow * p.res_w) / p.OW);
            rpix = ((long)img * p.res_h + sh_) * p.res_w + sw_;
          }
          const float* q32 = p.res + rpix * p.Cout + n;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(q32), r1 = *reinterpret_cast<const f32x4*>(q32 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += r0[e];
            v[4 + e] += r1[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.relu_cols) v[e] = hn::relu(v[e]);
        if (p.out_split) {
          f16x8 hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const _Float16 h = (_Float16)v[e];
            hi[e] = h;
            lo[e] = (_Float16)(v[e] - (float)h);
          }
          _Float16* q16 = reinterpret_cast<_Float16*>(p.y) + (long)m * p.ys + (n >> 5) * 64 + (n & 31);
          *reinterpret_cast<f16x8*>(q16) = hi;
          *reinterpret_cast<f16x8*>(q16 + 32) = lo;
        } else {
          float* q32 = p.y + (long)m * p.ys + n;
          f32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = v[e];
            o1[e] = v[4 + e];
          }
          *reinterpret_cast<f32x4*>(q32) = o0;
          *reinterpret_cast<f32x4*>(q32 + 4) = o1;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  // Scalar path (ragged Cout or unaligned strides)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + row;
      if (m >= p.M) continue;
      long rbase = 0;
      if (p.res_mode == 1) {
        rbase = (long)m * p.Cout;
      } else if (p.res_mode == 2) {
        const int img = m / ohow;
        const int rem = m - img * ohow;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        // nearest: src = floor(dst * in / out)  (exact 2x in the FPN)
        const int sh_ = (int)(((long)oh * p.res_h) / p.OH), sw_ = (int)(((long)ow * p.res_w) / p.OW);
        rbase = (((long)img * p.res_h + sh_) * p.res_w + sw_) * p.Cout;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[n];
        if (p.res_mode) v += p.res[rbase + n];
        if (n < p.relu_cols) v = hn::relu(v);
        if (p.out_split) {
          _Float16* q = reinterpret_cast<_Float16*>(p.y) + (long)m * p.ys + (n >> 5) * 64 + (n & 31);
          const _Float16 h = (_Float16)v;
          q[0] = h;
          q[32] = (_Float16)(v - (float)h);
        } else {
          p.y[(long)m * p.ys + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvParams& p0, bool smallc, hipStream_t st) {
  ConvParams p = p0;
  p.tiles_m = hn::cdiv(p.M, BM);
  p.tiles_n = hn::cdiv(p.Cout, BN);
  p.nblocks = p.tiles_m * p.tiles_n;
  if (smallc)
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true>), dim3(p.nblocks), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false>), dim3(p.nblocks), dim3(256), 0, st, p);
  HN_CHECK_LAUNCH("conv_igemm_f32_kernel");
  return HN_OK;
}

int64_t nblocks_for(const hn_conv_desc* d, int bm, int bn) {
  const int64_t M = (int64_t)d->n * d->oh * d->ow;
  return (int64_t)hn::cdiv(M, bm) * hn::cdiv(d->cout, bn);
}

}  // namespace

extern "C" int hn_conv2d_pick_tile(const hn_conv_desc* d) {
  if (!d) return HN_TILE_64x64;
  if (d->tile != HN_TILE_AUTO) return d->tile;
  if (d->cout <= 32) return HN_TILE_128x32;
  const int64_t want = 2 * 256;  // >= 2 workgroups per CU
  if (d->cout > 64 && nblocks_for(d, 128, 128) >= want) return HN_TILE_128x128;
  if (nblocks_for(d, 128, 64) >= want) return HN_TILE_128x64;
  return HN_TILE_64x64;
}

extern "C" int hn_conv2d_nhwc_f32(const hn_conv_desc* d, const float* x, const float* w,
                                  const float* bias, const float* residual, const float* in_scale,
                                  const float* in_shift, float* y, void* stream) {
  HN_CHECK_ARG(d && x && w && y, "hn_conv2d_nhwc_f32: null pointer");
  HN_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad tensor dims");
  HN_CHECK_ARG(d->cin % 4 == 0, "cin (%d) must be a multiple of 4 (pad with zeros)", d->cin);
  HN_CHECK_ARG(d->r > 0 && d->s > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "bad filter geometry");
  const int oh = (d->h + 2 * d->pad - d->dil * (d->r - 1) - 1) / d->stride + 1;
  const int ow = (d->w + 2 * d->pad - d->dil * (d->s - 1) - 1) / d->stride + 1;
  HN_CHECK_ARG(oh == d->oh && ow == d->ow, "output size mismatch: desc %dx%d, computed %dx%d", d->oh, d->ow, oh, ow);
  HN_CHECK_ARG(d->res_mode >= 0 && d->res_mode <= 2, "bad res_mode %d", d->res_mode);
  HN_CHECK_ARG(d->res_mode == 0 || residual, "res_mode set but residual is null");
  HN_CHECK_ARG(d->res_mode != 2 || (d->res_h > 0 && d->res_w > 0), "res_mode 2 needs res_h/res_w");
  HN_CHECK_ARG(!d->in_affine || (in_scale && in_shift), "in_affine set but scale/shift null");
  HN_CHECK_ARG(d->in_pix_stride == 0 || (d->in_pix_stride >= d->cin && d->in_pix_stride % 4 == 0), "bad in_pix_stride");
  HN_CHECK_ARG(d->out_pix_stride == 0 || d->out_pix_stride >= (d->out_split ? 2 : 1) * d->cout, "bad out_pix_stride");
  HN_CHECK_ARG(d->in_affine_stride == 0 || (d->in_affine_stride >= d->cin && d->in_affine_stride % 4 == 0), "bad in_affine_stride");
  HN_CHECK_ARG(!d->out_split || d->cout % 32 == 0, "S32 output needs cout %% 32 == 0 (got %d)", d->cout);
  HN_CHECK_ARG(!d->res_split, "the f32 kernel takes fp32 residuals only");
  HN_CHECK_ARG((int64_t)d->n * d->h * d->w * d->cin < (int64_t)1 << 40, "input too large");
  HN_CHECK_ARG((int64_t)d->n * d->oh * d->ow < (int64_t)1 << 31, "too many output pixels");

  ConvParams p;
  p.x = x; p.w = w; p.bias = bias; p.res = residual; p.in_scale = in_scale; p.in_shift = in_shift; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.Cout = d->cout; p.R = d->r; p.S = d->s;
  p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.OH = d->oh; p.OW = d->ow;
  p.M = d->n * d->oh * d->ow;
  p.Ktot = d->r * d->s * d->cin;
  p.ktiles = hn::cdiv(p.Ktot, BK);
  p.relu_cols = d->relu_cols; p.res_mode = d->res_mode; p.res_h = d->res_h; p.res_w = d->res_w;
  p.in_affine = d->in_affine;
  p.xs = d->in_pix_stride ? d->in_pix_stride : d->cin;
  p.out_split = d->out_split;
  p.ys = d->out_pix_stride ? d->out_pix_stride : (d->out_split ? 2 : 1) * d->cout;
  p.as = d->in_affine_stride ? d->in_affine_stride : d->cin;
  p.vec_epi = (d->cout % 8 == 0) && (p.out_split || p.ys % 4 == 0) && ((uintptr_t)y % 16 == 0) &&
              (bias == nullptr || (uintptr_t)bias % 16 == 0) && (residual == nullptr || (uintptr_t)residual % 16 == 0);
  p.tiles_m = p.tiles_n = p.nblocks = 0;
  const bool smallc = (d->cin % BK) != 0;
  hipStream_t st = (hipStream_t)stream;
  switch (hn_conv2d_pick_tile(d)) {
    case HN_TILE_128x128: return launch<128, 128, 2, 2>(p, smallc, st);
    case HN_TILE_128x64: return launch<128, 64, 2, 2>(p, smallc, st);
    case HN_TILE_64x64: return launch<64, 64, 2, 2>(p, smallc, st);
    case HN_TILE_128x32: return launch<128, 32, 4, 1>(p, smallc, st);
    case HN_TILE_64x128: return launch<64, 128, 2, 2>(p, smallc, st);
    default: return hn::fail(HN_ERR_ARG, "unknown tile id %d", d->tile);
  }
}
