// Implicit-GEMM convolution on the gfx950 f16 MFMA with SPLIT operands ("f16x3").
//
// Every fp32 operand v is represented as hi + lo with hi = fp16(v), lo = fp16(v - hi)
// (22 significant bits) and each product a*b is evaluated as
//      a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (a_lo*b_lo ~ 2^-22 |ab| is dropped)
// by three v_mfma_f32_32x32x16_f16 accumulating in ONE fp32 accumulator.  fp16 x fp16
// products are exact in fp32, so the result carries fp32-grade error (measured on the A2J
// network: keypoints move by <= 2e-4 vs the fp32 reference, bound 1e-3) at 16/3 = 5.3x the
// f32-MFMA rate.  Plain fp16 / bf16 operands miss the bound by two orders.
//
//   activations: fp32 NHWC in HBM (unchanged); split in registers by the loader
//                (optionally after the GroupNorm-on-load affine + ReLU)
//   weights:     split once at load time, stored [Cout][Ktot/32][2][32] fp16 so that a
//                k tile of one filter row is one contiguous 128-byte run (hi 64 B | lo 64 B)
//
// Kernel structure (v2, driven by rocprof counters of v1: VALU ~= MFMA time with 14 %
// co-execution, 1/3 of LDS cycles lost to write bank conflicts):
//   * steady-state loop without divergent branches: spatial padding is read from a zero
//     page, ragged rows / channels are clamped (their results are never stored), so the
//     MFMAs of tile t and the fp32->hi/lo conversion of tile t+1 live in ONE basic block
//     and are interleaved in source order (masked sched_barrier), instead of running as separate phases;
//   * two register stages (tile t+1 being converted, tile t+2 in flight from HBM/L2);
//   * LDS images are UNPADDED [row][32 halfs] (64 B rows) with the 16-byte chunk index
//     XOR-swizzled by (row>>2)&3: conflict-free for ds_read_b128 (64-bank rule) AND for
//     ds_write_b128 (32-bank rule); 64 KB per 128x128 workgroup, 2 workgroups per CU.
// Same epilogue / descriptor / tiling conventions as conv_igemm_f32.hip (requires
// Cin % 32 == 0; the 3- and 4-channel stems stay on the f32 kernel).
#include "hn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 256 bytes of zeros: source of every out-of-image tap (and of its GroupNorm scale/shift)
__device__ __attribute__((aligned(256))) float g_zero_page[64];

struct ConvParams16 {
  const float* x;
  const _Float16* w;  // [Cout][Ktot/32][2][32]
  const float* bias;
  const float* res;
  const float* in_scale;
  const float* in_shift;
  float* y;
  int N, H, W, Cin, Cout, R, S, stride, pad, dil, OH, OW;
  int M, Ktot, ktiles;
  int relu_cols, res_mode, res_h, res_w, in_affine;
  int xs, ys, as;
  int tiles_m, tiles_n, nblocks;
};

constexpr int BK = 32;   // k values per tile
constexpr int LDH = 32;  // LDS row pitch in halfs (64 bytes, unpadded, swizzled)


template <int BM, int BN, int WM, int WN, bool AFFINE>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_f16x3_kernel(const ConvParams16 p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int ROWS_PASS = NT / 4;  // 4 lanes cover one 64-byte row (one plane)
  constexpr int A_IT = BM / ROWS_PASS;
  constexpr int B_IT = (BN + ROWS_PASS - 1) / ROWS_PASS;
  static_assert(A_IT >= 1 && BM % ROWS_PASS == 0, "BM too small for this workgroup size");
  constexpr int A_PLANE = BM * LDH, B_PLANE = BN * LDH;  // halfs
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 2 * (A_PLANE + B_PLANE)];
  _Float16* As = smem;                    // [buf][plane][BM][32]
  _Float16* Bs = smem + 2 * 2 * A_PLANE;  // [buf][plane][BN][32]

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
  const int ohow = p.OH * p.OW;

  // ---- loader geometry: thread = (row within pass, 16-byte chunk c of the 64-byte row) ----
  const int lc = tid & 3;
  const int lrow = tid >> 2;
  // A rows (clamped to M-1: results of rows >= M are never stored)
  int a_ih0[A_IT], a_iw0[A_IT];
  long a_base[A_IT], a_aff[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int m = m0 + lrow + it * ROWS_PASS;
    m = m < p.M ? m : p.M - 1;
    const int img = m / ohow;
    const int rem = m - img * ohow;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    a_ih0[it] = oh * p.stride - p.pad;
    a_iw0[it] = ow * p.stride - p.pad;
    a_base[it] = (long)img * p.H * p.W * p.xs + lc * 8;
    a_aff[it] = (long)img * p.as + lc * 8;
  }
  // W rows (clamped to Cout-1: columns >= Cout are never stored)
  const _Float16* b_ptr[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    int n = n0 + lrow + it * ROWS_PASS;
    n = n < p.Cout ? n : p.Cout - 1;
    b_ptr[it] = p.w + (long)n * p.Ktot * 2 + lc * 8;
  }
  // swizzled LDS store offsets (halfs) of this thread's chunk, per pass
  int a_st[A_IT], b_st[B_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int row = lrow + it * ROWS_PASS;
    a_st[it] = row * LDH + ((lc ^ ((row >> 2) & 3)) << 3);
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int row = lrow + it * ROWS_PASS;
    b_st[it] = row * LDH + ((lc ^ ((row >> 2) & 3)) << 3);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct Stage {
    f32x4 a[A_IT][2];
    f16x8 b[2][B_IT];
    const float* ps[AFFINE ? A_IT : 1];  // GroupNorm scale / shift rows of this tile (or the zero page)
    const float* pt[AFFINE ? A_IT : 1];
  };
  Stage st0, st1;

  int cur_r = 0, cur_s = 0, cur_c = 0;  // tap / channel block of the next tile to load
  int load_t = 0;                       // index of the next tile to load
  // per-row source pointers of the CURRENT tap, advanced by 32 channels per tile; padding taps
  // point at the zero page and do not advance (a_inc = 0).  Recomputed only when the tap changes.
  const float* a_ptr[A_IT];
  const float* a_ps[A_IT];
  const float* a_pt[A_IT];
  int a_inc[A_IT];

  // issue the global loads of tile `load_t` into `st` and advance the tile iterators
  auto gload = [&](Stage& st) {
    if (cur_c == 0) {  // wave-uniform: first channel block of a new tap
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int ih = a_ih0[it] + cur_r * p.dil, iw = a_iw0[it] + cur_s * p.dil;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        a_ptr[it] = ok ? p.x + a_base[it] + ((long)ih * p.W + iw) * p.xs : g_zero_page;
        a_inc[it] = ok ? BK : 0;
        if constexpr (AFFINE) {
          // padding taps read scale = shift = 0 from the zero page: relu(0*0 + 0) = 0
          a_ps[it] = ok ? p.in_scale + a_aff[it] : g_zero_page;
          a_pt[it] = ok ? p.in_shift + a_aff[it] : g_zero_page;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      st.a[it][0] = *reinterpret_cast<const f32x4*>(a_ptr[it]);
      st.a[it][1] = *reinterpret_cast<const f32x4*>(a_ptr[it] + 4);
      a_ptr[it] += a_inc[it];
      if constexpr (AFFINE) {
        st.ps[it] = a_ps[it];
        st.pt[it] = a_pt[it];
        a_ps[it] += a_inc[it];
        a_pt[it] += a_inc[it];
      }
    }
    const long boff = (long)load_t * (2 * BK);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      st.b[0][it] = *reinterpret_cast<const f16x8*>(b_ptr[it] + boff);
      st.b[1][it] = *reinterpret_cast<const f16x8*>(b_ptr[it] + boff + BK);
    }
    ++load_t;
    cur_c += BK;
    if (cur_c >= p.Cin) {
      cur_c = 0;
      if (++cur_s == p.S) {
        cur_s = 0;
        ++cur_r;
      }
    }
  };

  // GroupNorm-on-load: y = relu(x * scale + shift), applied when a value is split
  auto affine4 = [&](f32x4 v, const float* ps, const float* pt) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(ps);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(pt);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
    return v;
  };

  // split the fp32 A values of `st` into hi/lo and store A and W chunks into LDS buffer `buf`
  auto convert_store = [&](const Stage& st, int buf) {
    _Float16* Ah = As + (buf * 2) * A_PLANE;
    _Float16* Al = Ah + A_PLANE;
    _Float16* Bh = Bs + (buf * 2) * B_PLANE;
    _Float16* Bl = Bh + B_PLANE;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      f32x4 v0 = st.a[it][0], v1 = st.a[it][1];
      if constexpr (AFFINE) {
        v0 = affine4(v0, st.ps[it], st.pt[it]);
        v1 = affine4(v1, st.ps[it] + 4, st.pt[it] + 4);
      }
      f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const _Float16 h0 = (_Float16)v0[e];
        const _Float16 h1 = (_Float16)v1[e];
        hi[e] = h0;
        hi[4 + e] = h1;
        lo[e] = (_Float16)(v0[e] - (float)h0);
        lo[4 + e] = (_Float16)(v1[e] - (float)h1);
      }
      *reinterpret_cast<f16x8*>(&Ah[a_st[it]]) = hi;
      *reinterpret_cast<f16x8*>(&Al[a_st[it]]) = lo;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (BN % ROWS_PASS == 0 || lrow + it * ROWS_PASS < BN) {
        *reinterpret_cast<f16x8*>(&Bh[b_st[it]]) = st.b[0][it];
        *reinterpret_cast<f16x8*>(&Bl[b_st[it]]) = st.b[1][it];
      }
    }
  };

  // fragment read offsets: lane (r = lane&31, h = lane>>5) reads chunk (2*s + h) of its row
  const int arow = wm * (BM / WM) + (lane & 31);
  const int brow = wn * (BN / WN) + (lane & 31);
  const int lh = lane >> 5;
  int a_rd[TM][2], b_rd[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = arow + i * 32;
      a_rd[i][s] = row * LDH + (((2 * s + lh) ^ ((row >> 2) & 3)) << 3);
    }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = brow + j * 32;
      b_rd[j][s] = row * LDH + (((2 * s + lh) ^ ((row >> 2) & 3)) << 3);
    }

  auto compute = [&](int buf) {
    const _Float16* Ah = As + (buf * 2) * A_PLANE;
    const _Float16* Al = Ah + A_PLANE;
    const _Float16* Bh = Bs + (buf * 2) * B_PLANE;
    const _Float16* Bl = Bh + B_PLANE;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(&Ah[a_rd[i][s]]);
        al[i] = *reinterpret_cast<const f16x8*>(&Al[a_rd[i][s]]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(&Bh[b_rd[j][s]]);
        bl[j] = *reinterpret_cast<const f16x8*>(&Bl[b_rd[j][s]]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // small cross terms first, the dominant hi*hi last
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // one steady-state step: MFMAs of tile t (LDS buffer t&1) interleaved IN SOURCE ORDER with
  // the split of tile t+1 (held in `cur`): the fp32->hi/lo conversion is cut into NP pieces
  // of 2 elements and one piece follows every (MFMAS / NP)-th MFMA, so the VALU work issues in
  // the MFMA shadow (an MFMA blocks VALU issue for only 8 of its 32 cycles).  Masked
  // sched_barriers keep that VALU/MFMA order; LDS / global memory ops may still be moved.
  constexpr int MFMAS = TM * TN * 3 * (BK / 16);
  constexpr int NP = A_IT * 4;
  constexpr int SB_MEM_ONLY = 0x4 | 0x10 | 0x20 | 0x40 | 0x80 | 0x100 | 0x200;  // SALU, VMEM, DS may cross
  auto step = [&](const Stage& cur, int t) {
    const int buf = t & 1, nbuf = buf ^ 1;
    const _Float16* Ah = As + (buf * 2) * A_PLANE;
    const _Float16* Al = Ah + A_PLANE;
    const _Float16* Bh = Bs + (buf * 2) * B_PLANE;
    const _Float16* Bl = Bh + B_PLANE;
    _Float16* nAh = As + (nbuf * 2) * A_PLANE;
    _Float16* nAl = nAh + A_PLANE;
    _Float16* nBh = Bs + (nbuf * 2) * B_PLANE;
    _Float16* nBl = nBh + B_PLANE;
    // W chunks of tile t+1 were loaded one step ago: store them first
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (BN % ROWS_PASS == 0 || lrow + it * ROWS_PASS < BN) {
        *reinterpret_cast<f16x8*>(&nBh[b_st[it]]) = cur.b[0][it];
        *reinterpret_cast<f16x8*>(&nBl[b_st[it]]) = cur.b[1][it];
      }
    }
    // GroupNorm scale / shift of tile t+1: tiny L1/L2-resident table, loaded at the top of
    // the step and consumed by the pieces below (latency covered by the first MFMAs)
    f32x4 gsc[AFFINE ? A_IT : 1][2], gsh[AFFINE ? A_IT : 1][2];
    if constexpr (AFFINE) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        gsc[it][0] = *reinterpret_cast<const f32x4*>(cur.ps[it]);
        gsc[it][1] = *reinterpret_cast<const f32x4*>(cur.ps[it] + 4);
        gsh[it][0] = *reinterpret_cast<const f32x4*>(cur.pt[it]);
        gsh[it][1] = *reinterpret_cast<const f32x4*>(cur.pt[it] + 4);
      }
    }
    f16x8 hi[A_IT], lo[A_IT];
    auto piece = [&](int pidx) {  // 2 elements of row (pidx / 4)
      const int it = pidx >> 2, q = pidx & 3;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float v = cur.a[it][q >> 1][(q & 1) * 2 + e];
        if constexpr (AFFINE)
          v = fmaxf(v * gsc[it][q >> 1][(q & 1) * 2 + e] + gsh[it][q >> 1][(q & 1) * 2 + e], 0.f);
        const _Float16 h = (_Float16)v;
        hi[it][q * 2 + e] = h;
        lo[it][q * 2 + e] = (_Float16)(v - (float)h);
      }
      if (q == 3) {  // row complete: store it
        *reinterpret_cast<f16x8*>(&nAh[a_st[it]]) = hi[it];
        *reinterpret_cast<f16x8*>(&nAl[a_st[it]]) = lo[it];
      }
    };
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(&Ah[a_rd[i][s]]);
        al[i] = *reinterpret_cast<const f16x8*>(&Al[a_rd[i][s]]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(&Bh[b_rd[j][s]]);
        bl[j] = *reinterpret_cast<const f16x8*>(&Bl[b_rd[j][s]]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m == 0 ? al[i] : ah[i], m == 1 ? bl[j] : bh[j],
                                                               acc[i][j], 0, 0, 0);
            const int k = ((s * TM + i) * TN + j) * 3 + m;
            for (int pp = k * NP / MFMAS; pp < (k + 1) * NP / MFMAS; ++pp) piece(pp);
            __builtin_amdgcn_sched_barrier(SB_MEM_ONLY);
          }
    }
    __syncthreads();
  };

  const int T = p.ktiles;
  // prologue: tile 0 -> LDS buffer 0, tile 1 in flight in st1
  gload(st0);
  convert_store(st0, 0);
  if (T > 1) gload(st1);
  __syncthreads();
  int t = 0;
  for (; t + 2 < T; t += 2) {
    gload(st0);  // tile t+2
    step(st1, t);
    if (t + 3 < T) gload(st1);  // tile t+3
    step(st0, t + 1);
  }
  if (t + 1 < T) {  // exactly tiles t, t+1 remain; t+1 is in st1
    step(st1, t);
    ++t;
  }
  compute(t & 1);

  // ---- epilogue: bias, residual, ReLU, NHWC store (128-byte runs per pixel row) ----
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
        if (n < p.relu_cols) v = fmaxf(v, 0.f);
        p.y[(long)m * p.ys + n] = v;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch16(const ConvParams16& p0, hipStream_t st) {
  ConvParams16 p = p0;
  p.tiles_m = hn::cdiv(p.M, BM);
  p.tiles_n = hn::cdiv(p.Cout, BN);
  p.nblocks = p.tiles_m * p.tiles_n;
  if (p.in_affine)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, true>), dim3(p.nblocks), dim3(WM * WN * 64), 0, st, p);
  else
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, false>), dim3(p.nblocks), dim3(WM * WN * 64), 0, st, p);
  HN_CHECK_LAUNCH("conv_igemm_f16x3_kernel");
  return HN_OK;
}

int64_t nblocks16(const hn_conv_desc* d, int bm, int bn) {
  const int64_t M = (int64_t)d->n * d->oh * d->ow;
  return (int64_t)hn::cdiv(M, bm) * hn::cdiv(d->cout, bn);
}

}  // namespace

extern "C" int hn_conv2d_f16x3_pick_tile(const hn_conv_desc* d) {
  if (!d) return HN_TILE_64x64;
  if (d->tile != HN_TILE_AUTO) return d->tile;
  if (d->cout <= 32) return HN_TILE_128x32;
  const int64_t want = 2 * 256;
  if (d->cout > 64 && nblocks16(d, 128, 128) >= want) return HN_TILE_128x128;
  if (nblocks16(d, 128, 64) >= want) return HN_TILE_128x64;
  return HN_TILE_64x64;
}

extern "C" int hn_conv2d_nhwc_f16x3(const hn_conv_desc* d, const float* x, const void* w16, const float* bias,
                                    const float* residual, const float* in_scale, const float* in_shift, float* y,
                                    void* stream) {
  HN_CHECK_ARG(d && x && w16 && y, "hn_conv2d_nhwc_f16x3: null pointer");
  HN_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad tensor dims");
  HN_CHECK_ARG(d->cin % 32 == 0, "f16x3 conv needs cin %% 32 == 0 (got %d); use hn_conv2d_nhwc_f32", d->cin);
  HN_CHECK_ARG(d->r > 0 && d->s > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "bad filter geometry");
  const int oh = (d->h + 2 * d->pad - d->dil * (d->r - 1) - 1) / d->stride + 1;
  const int ow = (d->w + 2 * d->pad - d->dil * (d->s - 1) - 1) / d->stride + 1;
  HN_CHECK_ARG(oh == d->oh && ow == d->ow, "output size mismatch: desc %dx%d, computed %dx%d", d->oh, d->ow, oh, ow);
  HN_CHECK_ARG(d->res_mode >= 0 && d->res_mode <= 2, "bad res_mode %d", d->res_mode);
  HN_CHECK_ARG(d->res_mode == 0 || residual, "res_mode set but residual is null");
  HN_CHECK_ARG(d->res_mode != 2 || (d->res_h > 0 && d->res_w > 0), "res_mode 2 needs res_h/res_w");
  HN_CHECK_ARG(!d->in_affine || (in_scale && in_shift), "in_affine set but scale/shift null");
  HN_CHECK_ARG(d->in_pix_stride == 0 || (d->in_pix_stride >= d->cin && d->in_pix_stride % 4 == 0), "bad in_pix_stride");
  HN_CHECK_ARG(d->out_pix_stride == 0 || d->out_pix_stride >= d->cout, "bad out_pix_stride");
  HN_CHECK_ARG(d->in_affine_stride == 0 || (d->in_affine_stride >= d->cin && d->in_affine_stride % 4 == 0), "bad in_affine_stride");
  HN_CHECK_ARG((int64_t)d->n * d->oh * d->ow < (int64_t)1 << 31, "too many output pixels");

  ConvParams16 p;
  p.x = x; p.w = (const _Float16*)w16; p.bias = bias; p.res = residual; p.in_scale = in_scale; p.in_shift = in_shift; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.Cout = d->cout; p.R = d->r; p.S = d->s;
  p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.OH = d->oh; p.OW = d->ow;
  p.M = d->n * d->oh * d->ow;
  p.Ktot = d->r * d->s * d->cin;
  p.ktiles = p.Ktot / BK;
  p.relu_cols = d->relu_cols; p.res_mode = d->res_mode; p.res_h = d->res_h; p.res_w = d->res_w;
  p.in_affine = d->in_affine;
  p.xs = d->in_pix_stride ? d->in_pix_stride : d->cin;
  p.ys = d->out_pix_stride ? d->out_pix_stride : d->cout;
  p.as = d->in_affine_stride ? d->in_affine_stride : d->cin;
  p.tiles_m = p.tiles_n = p.nblocks = 0;
  hipStream_t st = (hipStream_t)stream;
  switch (hn_conv2d_f16x3_pick_tile(d)) {
    case HN_TILE_128x128: return launch16<128, 128, 2, 2>(p, st);
    case HN_TILE_128x64: return launch16<128, 64, 2, 2>(p, st);
    case HN_TILE_64x64: return launch16<64, 64, 2, 2>(p, st);
    case HN_TILE_128x32: return launch16<128, 32, 4, 1>(p, st);
    case HN_TILE_64x128: return launch16<64, 128, 2, 2>(p, st);
    case HN_TILE_256x128: return launch16<256, 128, 4, 2>(p, st);
    default: return hn::fail(HN_ERR_ARG, "unknown tile id %d", d->tile);
  }
}
