// Winograd F(2x2, 3x3) convolution in the split-fp16 ("f16x3") domain -- 3x3, stride 1, pad 1, dilation 1.
//
// Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A needs 16 multiplies per 2x2 output tile and channel pair instead of 36:
// 1.33 f16 MFMAs per algorithmic MAC instead of the 3 of the direct kernel (conv_igemm_f16x3.hip), which sits at the
// socket power cap where only the MFMA count moves throughput.  fp32-grade results are kept by doing
// every transform in fp32 and splitting the TRANSFORMED operands: U = G g G^T is computed in fp64 on the host and split
// into fp16 hi + lo once; V = B^T d B is computed in fp32 from the S32 input (hi + lo is exact in fp32) and split to
// hi + lo on the fly; the 16 frequency products accumulate in fp32 MFMA accumulators with the same three terms
// (lo*hi + hi*lo + hi*hi); the output transform is fp32 adds.
//
// Workgroup = 8 waves = one block of 4 x 8 tiles (8 x 16 output pixels of one image) x 128 output channels x all 16
// frequencies.  Wave w owns frequency row i = w >> 1 (4 frequencies) and the 64-channel half w & 1: 4 x (2 x 4) 16x16
// accumulator tiles = 128 AGPRs.  Per 32-channel chunk:
//   1. the 10 x 18 input patch (S32 rows of 128 B) arrives by LDS-DMA through a buffer descriptor (out-of-image pixels
//      are lanes with bit 31 set in their offset: the range check returns zeros), one chunk ahead;
//   2. every lane transforms one (tile, channel pair): 16 pixels -> 16 frequencies, splits them and writes them into
//      V[freq][tile][hi 32 | lo 32] -- the direct kernel's LDS row format and swizzle, so the A fragments are read the
//      same way;
//   3. each wave multiplies its 4 frequencies: A fragments from V, B fragments (U) straight from global memory (L2) in
//      fragment order (1 KB per wave instruction, no LDS: nobody else uses this wave's frequencies).
// After the last chunk the waves reduce along j in registers, exchange the 4 x 2 partial planes through LDS, finish
// along i and run the usual epilogue tail per (pixel, 8 channels): bias, residual, ReLU, fp32 or S32 store.
#include "../../../handnet-pipeline_amd/csrc/hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TBH = 4, TBW = 8;              // tiles per block
constexpr int T = TBH * TBW;                 // 32 tiles = 8 x 16 output pixels
constexpr int PH = 2 * TBH + 2, PW = 2 * TBW + 2;
constexpr int NPATCH = PH * PW;              // 180 patch pixels
constexpr int PPIECES = 24;                  // DMA pieces of 8 pixels: 3 per wave (the last 12 rows are padding, zero-filled)
static_assert(PPIECES * 8 >= NPATCH && PPIECES % 8 == 0, "every wave issues the same number of pieces (vmcnt accounting)");
constexpr int CB = 64;                       // output channels per block
constexpr int NTHR = 512;
constexpr int ROWH = 64;                     // halfs per V row: hi 32 | lo 32
constexpr int V_HALFS = 16 * T * ROWH;       // 64 KB
constexpr int PROW = 32;                     // floats per patch row (one pixel, 32 channels)
constexpr int PATCH_FLOATS = PPIECES * 8 * PROW;
constexpr int TS = CB + 4;                   // exchange row stride (floats): the 4 row groups of a C/D fragment land 16 banks apart
constexpr int SMEM_BYTES = V_HALFS * 2 + 2 * PATCH_FLOATS * 4;   // 112640 B; the output exchange (69632 B) reuses it

struct WinoParams {
  const float* x;      // fp32 activations [N][H][W][Cin] (pixel stride xs floats)
  const _Float16* u;   // transformed filters in fragment order (pack_wino_f16x3)
  const float* bias;
  const void* res;
  void* y;
  int N, H, W, Cin, Cout;
  int xs;              // input pixel stride in floats
  int ys, rs;          // output / residual pixel stride (floats or halfs)
  int relu_cols, res_mode, out_split, res_split;
  int by, bx;          // blocks per image along y / x
  unsigned x_records;
  int kchunks, ctiles; // Cin / 32, Cout / 16
  int* range_flag;
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void mfma16(f32x4& c, const f16x8& a, const f16x8& b) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// same tail as the direct kernel's epi_finish8 for the features used here
__device__ __forceinline__ void finish8(const WinoParams& p, long m, int n, float (&v)[8]) {
  if (p.res_mode) {
    if (p.res_split) {
      const _Float16* q16 = reinterpret_cast<const _Float16*>(p.res) + m * p.rs + (n >> 5) * 64 + (n & 31);
      const f16x8 rh = *reinterpret_cast<const f16x8*>(q16), rl = *reinterpret_cast<const f16x8*>(q16 + 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)rh[e] + (float)rl[e];
    } else {
      const float* q32 = reinterpret_cast<const float*>(p.res) + m * p.rs + n;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(q32), r1 = *reinterpret_cast<const f32x4*>(q32 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += r0[e];
        v[4 + e] += r1[e];
      }
    }
  }
  if (p.relu_cols >= p.Cout) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (p.relu_cols > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (n + e < p.relu_cols) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.out_split) {
    if (p.range_flag) {
#pragma unroll
      for (int e = 0; e < 8; ++e) hn::range_note(p.range_flag, v[e]);
    }
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 h = (_Float16)v[e];
      hi[e] = h;
      lo[e] = (_Float16)(v[e] - (float)h);
    }
    _Float16* q16 = reinterpret_cast<_Float16*>(p.y) + m * p.ys + (n >> 5) * 64 + (n & 31);
    *reinterpret_cast<f16x8*>(q16) = hi;
    *reinterpret_cast<f16x8*>(q16 + 32) = lo;
  } else {
    float* q32 = reinterpret_cast<float*>(p.y) + m * p.ys + n;
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

__global__ __launch_bounds__(NTHR) void wino_f16x3_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  _Float16* Vs = reinterpret_cast<_Float16*>(smem_raw);                 // [16][T][64]
  float* Ps = reinterpret_cast<float*>(smem_raw + V_HALFS * 2);         // [2][PPIECES * 8][32]
  float* Ts = reinterpret_cast<float*>(smem_raw);                       // [4][2][T][TS] after the k loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int bxi = b % p.bx;
  b /= p.bx;
  const int byi = b % p.by;
  const int img = b / p.by;
  const int y0 = byi * (2 * TBH), x0 = bxi * (2 * TBW);
  const int n0 = blockIdx.y * CB;

  // ---- patch DMA geometry: piece q = wave + 8 * k, lane -> (pixel q * 8 + (lane >> 3), 16-byte chunk lane & 7) ----
  constexpr int PPW = (PPIECES + 7) / 8;  // pieces per wave
  unsigned p_off[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int q = wave + 8 * k;
    const int px = q * 8 + (lane >> 3), chunk = lane & 7;
    const int py = px / PW, pxx = px - py * PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
    const bool ok = q < PPIECES && px < NPATCH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    p_off[k] = ok ? ((unsigned)((img * p.H + gy) * p.W + gx) * (unsigned)p.xs + (unsigned)(chunk * 4)) * 4u : 0x80000000u;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_records, 0x00020000);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_patch = [&](int kc, int buf) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave_u + 8 * k;
      if (q < PPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_void*)(Ps + buf * PATCH_FLOATS + q * 8 * PROW), 16, (int)p_off[k],
                                                 kc * 128, 0, 0);
    }
    asm volatile("" ::: "memory");  // the B loads below stay younger than the patch (vmcnt accounting at the barriers)
  };

  // ---- transform role: (tile, channel pair) ----
  const int tt = tid >> 4, cp = tid & 15;
  const int tty = tt / TBW, ttx = tt - tty * TBW;
  const int p_rd = ((2 * tty) * PW + 2 * ttx) * PROW + cp * 2;
  const int v_wr_hi = tt * ROWH + (((cp >> 2) ^ swz(tt)) << 3) + (cp & 3) * 2;
  const int v_wr_lo = tt * ROWH + (((4 + (cp >> 2)) ^ swz(tt)) << 3) + (cp & 3) * 2;
  // The transform of one chunk is done in two parts -- frequency columns j = 0, 1 and j = 2, 3 -- each part needs three of the
  // four patch columns.  row stage: t[i][c] = sum_r B^T[i][r] d[r][c] (c = jp .. jp + 2)
  auto row_stage = [&](const float* patch, int jp, float (&t)[4][3][2]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float d[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 q = *reinterpret_cast<const float2*>(patch + p_rd + (r * PW + c + jp) * PROW);
        d[r][0] = q.x;
        d[r][1] = q.y;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        t[0][c][e] = d[0][e] - d[2][e];
        t[1][c][e] = d[1][e] + d[2][e];
        t[2][c][e] = d[2][e] - d[1][e];
        t[3][c][e] = d[1][e] - d[3][e];
      }
    }
  };
  // column stage of frequency row i for the column pair jp (0: j = 0, 1; 1: j = 2, 3), split, store
  auto col_stage = [&](const float (&t)[4][3][2], int jp, int i) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      f16x2 hi, lo;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // jp 0: t holds columns 0..2: V0 = t0 - t2, V1 = t1 + t2.   jp 1: t holds columns 1..3: V2 = t2 - t1, V3 = t1 - t3
        const float v = jp == 0 ? (jj == 0 ? t[i][0][e] - t[i][2][e] : t[i][1][e] + t[i][2][e])
                                : (jj == 0 ? t[i][1][e] - t[i][0][e] : t[i][0][e] - t[i][2][e]);
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
      }
      _Float16* plane = Vs + (i * 4 + jp * 2 + jj) * T * ROWH;
      *reinterpret_cast<f16x2*>(plane + v_wr_hi) = hi;
      *reinterpret_cast<f16x2*>(plane + v_wr_lo) = lo;
    }
  };

  // ---- MFMA role: frequency row fr (4 frequencies), 32 output channels (2 column tiles), all 32 tiles (2 row tiles) ----
  const int fr = wave >> 1, half = wave & 1;
  const int lg = lane >> 4, lr = lane & 15;
  f32x4 acc[4][2][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[j][rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_rd[2][2];  // A fragment addresses (halfs) inside one frequency plane: row = rt * 16 + lr, chunk (pl * 4 + lg) ^ swz(row)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int row = rt * 16 + lr;
      a_rd[rt][pl] = row * ROWH + (((pl * 4 + lg) ^ swz(row)) << 3);
    }
  // B fragments: u[((f * ctiles + ct) * kchunks + kc) * 2 + plane][lane][8], prefetched one frequency ahead
  const int ct0 = (n0 + half * 32) >> 4;
  f16x8 bh[4][2], bl[4][2];  // slot j holds frequency j; refilled for the next chunk as soon as its MFMAs are issued
  auto load_b = [&](int kc, int j) {
    const int slot = j;
#ifdef WINO_NO_BLOAD
    if (kc > 0) return;
#endif
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const _Float16* up = p.u + ((((long)(fr * 4 + j) * p.ctiles + ct0 + ct) * p.kchunks + kc) * 2) * 512 + lane * 8;
      bh[slot][ct] = *reinterpret_cast<const f16x8*>(up);
      bl[slot][ct] = *reinterpret_cast<const f16x8*>(up + 512);
    }
  };
  f16x8 ah[2], al[2];
  auto load_a = [&](int j) {
    const _Float16* vp = Vs + (fr * 4 + j) * T * ROWH;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      ah[rt] = *reinterpret_cast<const f16x8*>(vp + a_rd[rt][0]);
      al[rt] = *reinterpret_cast<const f16x8*>(vp + a_rd[rt][1]);
    }
  };
  auto mfma_term = [&](int j, int term) {  // 4 independent accumulator tiles
#ifdef WINO_NO_MFMA
    return;
#endif
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        mfma16(acc[j][rt][ct], term == 0 ? al[rt] : ah[rt], term == 1 ? bl[j][ct] : bh[j][ct]);
  };
#define WINO_FENCE() __builtin_amdgcn_sched_barrier(0)
  // half a chunk: frequencies j0, j0 + 1 on the matrix pipe, one transform part (rows of t -> column pair jp) on the VALU,
  // in fixed alternation: 4 MFMAs (64 cycles of the pipe), one frequency row of the transform.
  auto half_chunk = [&](int j0, bool with_transform, const float* patch, int jp, int kn) {
#ifdef WINO_NO_TRANSFORM
    with_transform = false;
#endif
    float t[4][3][2];
    load_a(j0);
    if (with_transform) row_stage(patch, jp, t);
    WINO_FENCE();
    mfma_term(j0, 0);
    WINO_FENCE();
    if (with_transform) col_stage(t, jp, 0);
    WINO_FENCE();
    mfma_term(j0, 1);
    WINO_FENCE();
    if (with_transform) col_stage(t, jp, 1);
    WINO_FENCE();
    mfma_term(j0, 2);
    WINO_FENCE();
    load_a(j0 + 1);
    load_b(kn, j0);
    WINO_FENCE();
    mfma_term(j0 + 1, 0);
    WINO_FENCE();
    if (with_transform) col_stage(t, jp, 2);
    WINO_FENCE();
    mfma_term(j0 + 1, 1);
    WINO_FENCE();
    if (with_transform) col_stage(t, jp, 3);
    WINO_FENCE();
    mfma_term(j0 + 1, 2);
    WINO_FENCE();
    load_b(kn, j0 + 1);
  };

  // ---- prologue: patches 0 and 1 in flight, frequency columns 0, 1 of chunk 0 transformed before the loop ----
  dma_patch(0, 0);
  if (p.kchunks > 1) dma_patch(1, 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) load_b(0, j);
  if (p.kchunks > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(16 + PPW) : "memory");  // patch 0 landed: younger = patch 1, 16 B loads
  else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  {
    float t[4][3][2];
    row_stage(Ps, 0, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_stage(t, 0, i);
  }
  // Chunk kc:  [barrier 1: V columns 0, 1 of kc complete, columns 2, 3 free]
  //              frequencies 0, 1 of kc   ||  transform of kc, columns 2, 3           (patch kc)
  //            [barrier 2: V columns 2, 3 of kc complete, columns 0, 1 free, patch kc + 1 landed, patch kc's buffer free]
  //              patch kc + 2 starts;  frequencies 2, 3 of kc  ||  transform of kc + 1, columns 0, 1   (patch kc + 1)
  for (int kc = 0; kc < p.kchunks; ++kc) {
    const float* pcur = Ps + (kc & 1) * PATCH_FLOATS;
    const float* pnext = Ps + ((kc & 1) ^ 1) * PATCH_FLOATS;
    const bool more = kc + 1 < p.kchunks;
    const int kn = more ? kc + 1 : kc;  // the last chunk prefetches a fragment nobody uses
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    half_chunk(0, true, pcur, 1, kn);
    // Patch kc + 1 was started at this point of the previous chunk; the 16 vector-memory operations issued since are B
    // fragments (frequencies 2, 3 of kc, then 0, 1 of kc + 1), so "all but the 16 youngest" covers it.
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef WINO_NO_DMA
    if (kc + 2 < p.kchunks) dma_patch(kc + 2, kc & 1);
#endif
    if (more) half_chunk(2, true, pnext, 0, kn);
    else half_chunk(2, false, pnext, 0, kn);
  }
#undef WINO_FENCE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with V: LDS becomes the exchange area

  // ---- output transform, stage 1 (along j, in registers): s_b = sum_j A[j][b] m_j;  A^T = [[1,1,1,0],[0,1,-1,-1]] ----
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float m0 = acc[0][rt][ct][r], m1 = acc[1][rt][ct][r], m2 = acc[2][rt][ct][r], m3 = acc[3][rt][ct][r];
        const int tile = rt * 16 + lg * 4 + r;                 // C/D map: row = 4 * (lane >> 4) + reg, col = lane & 15
        const int col = half * 32 + ct * 16 + lr;
        Ts[((fr * 2 + 0) * T + tile) * TS + col] = m0 + m1 + m2;
        Ts[((fr * 2 + 1) * T + tile) * TS + col] = m1 - m2 - m3;
      }
  __syncthreads();
  // ---- stage 2 (along i) + epilogue tail: unit = (tile, a, b, 8-channel group) ----
#pragma unroll
  for (int k = 0; k < (T * 4 * (CB / 8)) / NTHR; ++k) {
    const int u = tid + NTHR * k;
    const int cg = u & 7, ab = (u >> 3) & 3, tile = u >> 5;
    const int a = ab >> 1, bb = ab & 1;
    const int ty = tile / TBW, tx = tile - ty * TBW;
    const int oy = y0 + 2 * ty + a, ox = x0 + 2 * tx + bb;
    const int n = n0 + cg * 8;
    const float* s0 = &Ts[(((a ? 1 : 0) * 2 + bb) * T + tile) * TS + cg * 8];      // i = a (0 or 1)
    const float* s1 = &Ts[(((a ? 2 : 1) * 2 + bb) * T + tile) * TS + cg * 8];      // i = a + 1
    const float* s2 = &Ts[(((a ? 3 : 2) * 2 + bb) * T + tile) * TS + cg * 8];      // i = a + 2
    float v[8];
#pragma unroll
    for (int h4 = 0; h4 < 2; ++h4) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(s0 + h4 * 4), q1 = *reinterpret_cast<const f32x4*>(s1 + h4 * 4),
                  q2 = *reinterpret_cast<const f32x4*>(s2 + h4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[h4 * 4 + e] = a ? (q0[e] - q1[e] - q2[e]) : (q0[e] + q1[e] + q2[e]);
    }
    if (oy >= p.H || ox >= p.W || n >= p.Cout) continue;
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += b0[e];
        v[4 + e] += b1[e];
      }
    }
    finish8(p, ((long)img * p.H + oy) * p.W + ox, n, v);
  }
}

}  // namespace

// Host-side: transformed, split filter bank in fragment order.
//   w  [cout][3][3][cin] fp32 (the layout of hn_conv2d_nhwc_f32)  ->
//   u16 fp16 [16 freq][cout / 16][cin / 32][2 planes][64 lanes][8]   with U_f = (G g G^T)[i][j], f = 4 i + j,
//   lane l of a fragment = (cout c = l & 15, channels 8 (l >> 4) .. + 7) of the 16 x 32 tile; computed in fp64.
extern "C" int64_t hn_wino_f16x3_bank_halfs(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cout % 16 || cin % 32) return 0;
  return (int64_t)16 * cout * cin * 2;
}

extern "C" int hn_wino_f16x3_pack(const float* w, int cout, int cin, void* u16_host) {
  HN_CHECK_ARG(w && u16_host, "hn_wino_f16x3_pack: null pointer");
  HN_CHECK_ARG(cout > 0 && cin > 0 && cout % 16 == 0 && cin % 32 == 0, "Winograd bank needs cout %% 16 == 0 and cin %% 32 == 0");
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  _Float16* out = (_Float16*)u16_host;
  const int ctiles = cout / 16, kch = cin / 32;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      double g[3][3], tmp[4][3], U[4][4];
      for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) g[r][s] = (double)w[(((size_t)co * 3 + r) * 3 + s) * cin + ci];
      for (int i = 0; i < 4; ++i)
        for (int s = 0; s < 3; ++s) tmp[i][s] = G[i][0] * g[0][s] + G[i][1] * g[1][s] + G[i][2] * g[2][s];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
      const int ct = co >> 4, c = co & 15, kc = ci >> 5, k = ci & 31, lane = c + 16 * (k >> 3), e = k & 7;
      for (int f = 0; f < 16; ++f) {
        const float v = (float)U[f >> 2][f & 3];
        if (!(fabsf(v) <= 65504.f)) return hn::fail(HN_ERR_ARG, "transformed filter value %g leaves the fp16 range", (double)v);
        const _Float16 h = (_Float16)v;
        const size_t base = ((((size_t)f * ctiles + ct) * kch + kc) * 2) * 512 + (size_t)lane * 8 + e;
        out[base] = h;
        out[base + 512] = (_Float16)(v - (float)h);
      }
    }
  return HN_OK;
}

// 3x3 / stride 1 / pad 1 convolution, S32 input, Winograd F(2x2,3x3).  d: the usual descriptor (r = s = 3, stride = dil =
// pad = 1, cin % 32 == 0, cout % 128 == 0); u16 = device copy of hn_wino_f16x3_pack's bank.
extern "C" int hn_conv3x3_wino_f16x3(const hn_conv_desc* d, const float* x, const void* u16, const float* bias,
                                     const void* residual, void* y, void* stream) {
  HN_CHECK_ARG(d && x && u16 && y, "hn_conv3x3_wino_f16x3: null pointer");
  HN_CHECK_ARG(d->r == 3 && d->s == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1, "Winograd path is 3x3 / stride 1 / pad 1 / dilation 1");
  HN_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin % 32 == 0 && d->cout % CB == 0, "needs cin %% 32 == 0 and cout %% %d == 0", CB);
  HN_CHECK_ARG(d->oh == d->h && d->ow == d->w, "output size mismatch");
  HN_CHECK_ARG(d->res_mode == 0 || (d->res_mode == 1 && residual), "residual: same-shape only");
  HN_CHECK_ARG(!d->in_affine, "no GroupNorm-on-load here");
  HN_CHECK_ARG(!d->out_split || d->cout % 32 == 0, "S32 output needs cout %% 32 == 0");
  WinoParams p;
  p.x = (const float*)x; p.u = (const _Float16*)u16; p.bias = bias; p.res = residual; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.Cout = d->cout;
  p.xs = d->in_pix_stride ? d->in_pix_stride : d->cin;
  p.out_split = d->out_split; p.res_split = d->res_split;
  p.ys = d->out_pix_stride ? d->out_pix_stride : (d->out_split ? 2 : 1) * d->cout;
  p.rs = d->res_pix_stride ? d->res_pix_stride : (d->res_split ? 2 : 1) * d->cout;
  p.relu_cols = d->relu_cols; p.res_mode = d->res_mode;
  p.by = hn::cdiv(d->h, 2 * TBH); p.bx = hn::cdiv(d->w, 2 * TBW);
  const int64_t xbytes = (int64_t)d->n * d->h * d->w * p.xs * 4;
  HN_CHECK_ARG(xbytes < ((int64_t)1 << 31), "input of 2 GB or more: use the direct kernel");
  p.x_records = (unsigned)xbytes;
  p.kchunks = d->cin / 32; p.ctiles = d->cout / 16;
  p.range_flag = hn::range_flag_ptr();
  HN_CHECK_ARG((uintptr_t)y % 16 == 0 && (bias == nullptr || (uintptr_t)bias % 16 == 0) && (residual == nullptr || (uintptr_t)residual % 16 == 0) &&
                   p.ys % 4 == 0 && p.rs % 4 == 0 && p.xs % 4 == 0 && (uintptr_t)x % 16 == 0, "16-byte aligned tensors / strides needed");
  static bool attr_set = false;
  if (!attr_set) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)wino_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    attr_set = true;
  }
  static_assert(4 * 2 * T * TS * 4 <= SMEM_BYTES && SMEM_BYTES <= 160 * 1024, "LDS budget");
  const int64_t blocks = (int64_t)d->n * p.by * p.bx;
  HN_CHECK_ARG(blocks < ((int64_t)1 << 31), "too many blocks");
  hipLaunchKernelGGL(wino_f16x3_kernel, dim3((unsigned)blocks, d->cout / CB), dim3(NTHR), SMEM_BYTES, (hipStream_t)stream, p);
  HN_CHECK_LAUNCH("wino_f16x3_kernel");
  return HN_OK;
}
