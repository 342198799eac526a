// 3x3 / stride 1 / pad 1 convolution with at most 16 output channels over up to HN_FCOS_MAX_LEVELS maps in one launch
// (round 3; the FCOS head OUTPUT convolutions: cls_logits + hand_lr (C + 2 channels), bbox_reg + bbox_ctrness (5), the
// ext=True heads (8): fcos_utils/fcos.py:247-264,362-363 -- 256 -> <= 8 channels on every FPN level, weights shared).
//
// On the implicit-GEMM kernel these layers ran the 128x32 tile at 50 TFLOP/s (0.02 of the f16 peak): 27 of 32 output
// columns are padding, and the A operand -- all of the traffic -- is re-gathered per filter row.  Here the output channels
// are ONE 16-row MFMA tile (W fragment = srcA), a workgroup of 8 waves owns 16 x 16 output pixels, and per 32-channel
// block the 18 x 18 input patch (41 KB) and the nine filter tiles (16 x 128 B each) are staged ONCE, double-buffered, so
// block cb + 1 loads while the 54 MFMAs per wave of block cb run from LDS: every input pixel is fetched once (+ halo).
//   * k order: channel block outer, taps inner, terms lo*hi, hi*lo, hi*hi -- the implicit-GEMM kernel's, bit-identical;
//   * padding pixels are zero-filled by the buffer descriptor's range check; bank swizzle chunk ^ ((row >> 1) & 7);
//   * lane = pixel, registers = 4 consecutive output channels: bias, ReLU on a channel prefix, fp32 store of the
//     Cout <= 16 real channels (dense [pixel][Cout] rows, the layout hn_fcos_candidates reads).
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kT = 16, kP = kT + 2, kPatchPix = kP * kP;
constexpr int kNT = 512;                                        // 8 waves
constexpr int kPatchRounds = (kPatchPix * 8 + kNT - 1) / kNT;   // 6 rounds of 512 sixteen-byte pieces
constexpr int kPatchBytes = kPatchRounds * kNT * 16;            // 49152
constexpr int kWRounds = (9 * 16 * 8 + kNT - 1) / kNT;          // 3 rounds: 9 taps x 16 rows x 8 chunks = 1152 pieces
constexpr int kWBytes = kWRounds * kNT * 16;                    // 24576
constexpr int kStageBytes = kPatchBytes + kWBytes;

struct ThinLevel {
  const _Float16* x;
  float* y;
  int h, w, ty, tx, first_block;
  unsigned x_records;
};
struct ThinParams {
  ThinLevel lv[HN_FCOS_MAX_LEVELS];
  int levels, n, cbs, cout, xs, ys, relu_cols;
  const _Float16* wt;   // [cout][cbs * 9][2][32]
  const float* bias;
  unsigned w_records;
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// one workgroup of the tap kernel: workgroup `block` of the launch described by p
__device__ __forceinline__ void thin_tap_body(const ThinParams p, const int block) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, px = lane & 15;
  // level of this workgroup: constant-index selects (a dynamic index into the by-value table would go through scratch)
  int li = 0;
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l < p.levels && block >= p.lv[l].first_block) li = l;
  ThinLevel L = p.lv[0];
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l == li) L = p.lv[l];
  const int lid = block - L.first_block;
  const int per_img = L.ty * L.tx;
  const int img = lid / per_img;
  const int rem = lid - img * per_img;
  const int tyi = rem / L.tx, txi = rem - tyi * L.tx;
  const int y0 = tyi * kT, x0 = txi * kT;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)L.x, 0, (int)L.x_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)p.w_records, 0x00020000);
  const int wave_base = __builtin_amdgcn_readfirstlane(wave) * 64;

  unsigned a_off[kPatchRounds];
#pragma unroll
  for (int r = 0; r < kPatchRounds; ++r) {
    const int piece = r * kNT + tid;
    int q = piece >> 3;
    const int pos = piece & 7;
    const bool in_patch = q < kPatchPix;
    q = in_patch ? q : kPatchPix - 1;
    const int py = q / kP, pxx = q - py * kP;
    const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
    const bool ok = in_patch && (unsigned)iy < (unsigned)L.h && (unsigned)ix < (unsigned)L.w;
    const int chunk = pos ^ swz(q);
    const unsigned off = (((unsigned)(img * L.h + iy) * (unsigned)L.w + (unsigned)ix) * (unsigned)p.xs + (unsigned)(chunk * 8)) * 2u;
    a_off[r] = ok ? off : 0x80000000u;   // out of range: zeros
  }
  unsigned b_off[kWRounds];
#pragma unroll
  for (int r = 0; r < kWRounds; ++r) {
    const int piece = r * kNT + tid;
    int row = piece >> 3;                 // tap * 16 + output channel
    const int pos = piece & 7;
    row = row < 9 * 16 ? row : 9 * 16 - 1;
    const int tap = row >> 4;
    int oc = row & 15;
    oc = oc < p.cout ? oc : p.cout - 1;   // rows >= cout duplicate the last real filter (their results are never stored)
    b_off[r] = (((unsigned)oc * (unsigned)(p.cbs * 9) + (unsigned)tap) * 64u + (unsigned)((pos ^ swz(row)) * 8)) * 2u;
  }
  auto dma_stage = [&](int cb, int stage) {
    char* base = smem + stage * kStageBytes;
#pragma unroll
    for (int r = 0; r < kPatchRounds; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_void*)(base + (r * kNT + wave_base) * 16), 16, (int)a_off[r], cb * 128, 0, 0);
#pragma unroll
    for (int r = 0; r < kWRounds; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(base + kPatchBytes + (r * kNT + wave_base) * 16), 16,
                                               (int)b_off[r], cb * 9 * 128, 0, 0);
  };

  // wave w owns output rows 2w, 2w + 1 of the tile (row tile i = output row 2w + i, lane pixel = column px)
  int pp0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) pp0[i] = (wave * 2 + i) * kP + px;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  dma_stage(0, 0);
#pragma unroll 1
  for (int cb = 0; cb < p.cbs; ++cb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // block cb landed for every wave; nobody reads the other stage any more
    if (cb + 1 < p.cbs) dma_stage(cb + 1, (cb + 1) & 1);
    const char* patch = smem + (cb & 1) * kStageBytes;
    const char* wb = patch + kPatchBytes;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const int row = tap * 16 + px;   // W fragment row: lane (px = output channel, lg = k chunk)
      const int sw = swz(row);
      const f16x8 bh = *reinterpret_cast<const f16x8*>(wb + row * 128 + ((lg ^ sw) << 4));
      const f16x8 bl = *reinterpret_cast<const f16x8*>(wb + row * 128 + (((4 + lg) ^ sw) << 4));
      f16x8 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pp = pp0[i] + dy * kP + dx;
        const int s = swz(pp);
        ah[i] = *reinterpret_cast<const f16x8*>(patch + pp * 128 + ((lg ^ s) << 4));
        al[i] = *reinterpret_cast<const f16x8*>(patch + pp * 128 + (((4 + lg) ^ s) << 4));
      }
      // term order of conv_igemm_f16x3_kernel: lo*hi, hi*lo, hi*hi (W = srcA: lane = pixel, registers = channels)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], acc[i], 0, 0, 0);
    }
  }

  // ---- epilogue: lane (px, lg) holds channels 4 lg .. 4 lg + 3 of pixel (row 2w + i, column px) ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = y0 + wave * 2 + i, ox = x0 + px;
    if (oy >= L.h || ox >= L.w) continue;
    float* dst = L.y + (((long)img * L.h + oy) * L.w + ox) * p.ys;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = lg * 4 + r;
      if (n >= p.cout) continue;
      float v = acc[i][r] + (p.bias ? p.bias[n] : 0.f);
      if (n < p.relu_cols) v = hn::relu(v);
      dst[n] = v;
    }
  }
}

__global__ __launch_bounds__(kNT, 2) void conv3x3_thin_kernel(const ThinParams p) { thin_tap_body(p, (int)blockIdx.x); }

// Up to three independent launches of the tap kernel as ONE grid (the FCOS head outputs of a single frame: cls + lr, ext, reg +
// ctr -- 89 workgroups each on 256 CUs, latency-bound: 17-18 us per launch alone).  Member m owns the workgroups
// [first[m], first[m + 1]); each workgroup runs exactly what its own launch would have run (bit-identical outputs).
constexpr int kThinGroupMax = 3;
struct ThinGroup {
  ThinParams m[kThinGroupMax];
  int first1, first2;   // first workgroup of members 1 and 2 (0x7fffffff: absent)
};
__global__ __launch_bounds__(kNT, 2) void conv3x3_thin_group_kernel(const ThinGroup g) {
  const int b = (int)blockIdx.x;
  if (b < g.first1) thin_tap_body(g.m[0], b);
  else if (b < g.first2) thin_tap_body(g.m[1], b - g.first1);
  else thin_tap_body(g.m[2], b - g.first2);
}

// ------------------------------------------------------------------------------------------------------------------
// P form (Cout <= 5, Cin % 128 == 0): the tap kernel above is bound by its LDS reads -- every tap re-reads the pixel
// fragments, one ds_read_b128 per MFMA, and 11 of the 16 MFMA rows are padding.  Turn the convolution inside out instead:
//   P[(tap, oc)][q] = sum_c W[oc][tap][c] * X[q][c]        one GEMM with 9 * Cout <= 45 rows (three 16-row MFMA tiles) over
//                                                          the FLAT pixel list q of a level -- no geometry, no halo, each
//                                                          pixel's 128 B per channel block go global -> registers ONCE;
//   y[oc][f] = bias + sum_tap P[(tap, oc)][f + d_tap]      nine shifted reads per output from an LDS ring of P columns,
//                                                          taps 0..8 in order, taps that fall into the zero padding skipped.
// A workgroup sweeps a contiguous range of flat output pixels in steps of 256 P columns (8 waves x 2 column tiles), the
// outputs trailing the P front by one image row + 1; the ring holds 256 + 2 (W + 1) columns.  The filter bank (Cin/32 x
// 6 KB, fragment-ready) is LDS-resident for the life of the workgroup.  MFMA work / 3, LDS reads / 9 against the tap kernel:
// the launch becomes a stream over the input (HBM bound).  The summation order differs from the implicit GEMM's (per tap
// over all channels, then over taps), so results agree to fp32 rounding, not bit for bit.
constexpr int kStepPx = 256, kMT = 3, kDepth = 4, kFlatMaxC = 5;

struct FlatLevel {
  const _Float16* x;    // S32 input (plain form)
  const float* xf;      // AFFINE form: raw fp32 [pixel][xs] conv output, already at the head's first channel
  const float* scale;   // AFFINE form: GroupNorm scale / shift tables [image][as], already at the head's first channel
  const float* shift;
  float* y;
  int h, w, total, first_unit, unit_len;
};
struct FlatParams {
  FlatLevel lv[HN_FCOS_MAX_LEVELS];
  int levels, cbs, cout, xs, relu_cols, ring, as;
  const _Float16* wt;
  const float* bias;
  int* range_flag;
};
// AFFINE: images whose scale / shift rows a workgroup keeps in LDS (its pixel range spans no more): as many as fit beside the
// filter bank and the ring, at most 8, at least 3
constexpr int kAffImagesMax = 8, kAffImagesMin = 3;

__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// AFFINE: the input is the RAW fp32 output of the last tower layer and the GroupNorm apply pass (y = relu(x * scale +
// shift), split into fp16 hi + lo: hn_affine_split_f32's arithmetic, expression for expression) happens on the fragments in
// registers -- the kernel streams at the HBM / L1 rate with its VALU idle, and the separate pass moved 2.4 GB per step
// (0.4 ms at batch 32) only to hand this kernel the same bytes again.  The scale / shift rows of the (at most eight)
// images a workgroup's pixel range touches sit in LDS: vmcnt retires in order, so a table load from memory at the point
// of use would wait for every older fragment prefetch and drain the stream.
template <bool AFFINE>
__global__ __launch_bounds__(kNT, 1) void conv3x3_thin_flat_kernel(const FlatParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, px = lane & 15;
  int li = 0;
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l < p.levels && (int)blockIdx.x >= p.lv[l].first_unit) li = l;
  FlatLevel L = p.lv[0];
#pragma unroll
  for (int l = 1; l < HN_FCOS_MAX_LEVELS; ++l)
    if (l == li) L = p.lv[l];
  const int unit = (int)blockIdx.x - L.first_unit;
  const int o0 = unit * L.unit_len, o1 = min(L.total, o0 + L.unit_len);
  const int halo = L.w + 1;
  const int p0 = max(0, o0 - halo), p1 = min(L.total, o1 + halo);
  const int steps = (p1 - p0 + kStepPx - 1) / kStepPx;
  const int cbs = p.cbs, cout = p.cout, ring = p.ring, rows = 9 * cout;
  char* wbank = smem;                                              // [cb][mt][hi|lo][lane] x 16 B
  float* ringp = reinterpret_cast<float*>(smem + cbs * (kMT * 2 * 1024));   // [rows][ring]
  float* afftab = ringp + ((rows * ring + 3) & ~3);                // AFFINE: [image - img0][scale | shift][cbs * 32]
  const int hw = L.h * L.w;
  const int img0 = p0 / hw;
  if constexpr (AFFINE) {
    const int cin = cbs * 32, nimg = (p1 - 1) / hw - img0 + 1;     // <= the table's capacity (the host sizes the ranges for that)
    for (int e = tid; e < nimg * 2 * (cin / 4); e += kNT) {
      const int c4 = e % (cin / 4), q = e / (cin / 4), which = q & 1, j = q >> 1;
      const float* src = (which ? L.shift : L.scale) + (size_t)(img0 + j) * p.as + c4 * 4;
      *reinterpret_cast<f32x4*>(afftab + (j * 2 + which) * cin + c4 * 4) = *reinterpret_cast<const f32x4*>(src);
    }
  }

  for (int e = tid; e < cbs * kMT * 2 * 64; e += kNT) {
    const int ln = e & 63, hl = (e >> 6) & 1, q = e >> 7;
    const int cb = q / kMT, mt = q - cb * kMT;
    const int r = mt * 16 + (ln & 15);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < rows) {
      const int tap = r / cout, oc = r - tap * cout;
      v = *reinterpret_cast<const uint4*>(p.wt + ((size_t)(oc * (cbs * 9) + cb * 9 + tap) * 64 + hl * 32 + (ln >> 4) * 8));
    }
    *reinterpret_cast<uint4*>(wbank + e * 16) = v;
  }

  // the wave's two column tiles of a step: flat pixels p0 + step * 256 + (2 wave + i) * 16 + px (clamped: columns past the
  // end of the level hold a duplicate that no output ever reads)
  const int col0 = (wave * 2) * 16 + px;
  const _Float16* xl = L.x + lg * 8;
  const float* xfl = L.xf + lg * 8;
  auto issue = [&](int step, int cb, f16x8 (&dst)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = min(p0 + step * kStepPx + col0 + i * 16, L.total - 1);
      if constexpr (AFFINE) {   // eight raw fp32 channels: the same 32 bytes per lane as hi + lo
        const float* src = xfl + (size_t)g * p.xs + cb * 32;
        dst[i][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(src));
        dst[i][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(src + 4));
      } else {
        const _Float16* src = xl + (size_t)g * p.xs + cb * 64;
        dst[i][0] = *reinterpret_cast<const f16x8*>(src);
        dst[i][1] = *reinterpret_cast<const f16x8*>(src + 32);
      }
    }
  };
  float bv[kFlatMaxC];   // read before the stream starts: a global load in the output phase would drain the prefetch queue
#pragma unroll
  for (int oc = 0; oc < kFlatMaxC; ++oc) bv[oc] = (p.bias && oc < cout) ? p.bias[oc] : 0.f;
  f16x8 xr[kDepth][2][2];
#pragma unroll
  for (int k = 0; k < kDepth; ++k) issue(0, k, xr[k]);
  wg_barrier_lds();   // filter bank in place

  int slot0 = col0;          // ring slot of the wave's first column this step (ring > 256 + 32)
  int e_lo = o0;
#pragma unroll 1
  for (int step = 0; step < steps; ++step) {
    f32x4 acc[2][kMT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* aff[2];   // AFFINE: this lane's rows of the LDS table (the image of each of its two pixels, its 8 channels)
    if constexpr (AFFINE) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int g = min(min(p0 + step * kStepPx + col0 + i * 16, p1 - 1), L.total - 1);   // (columns past p1 are never used)
        aff[i] = afftab + (g / hw - img0) * 2 * (cbs * 32) + lg * 8;
      }
    }
#pragma unroll 1
    for (int cb0 = 0; cb0 < cbs; cb0 += kDepth) {
#pragma unroll
      for (int k = 0; k < kDepth; ++k) {
        const int cb = cb0 + k;
        f16x8 xv[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { xv[i][0] = xr[k][i][0]; xv[i][1] = xr[k][i][1]; }
        if constexpr (AFFINE) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 a = __builtin_bit_cast(f32x4, xv[i][0]), b = __builtin_bit_cast(f32x4, xv[i][1]);
            const float* sp = aff[i] + cb * 32;
            const float* tp = sp + cbs * 32;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // hn_affine_split_f32 with relu = 1, expression for expression
              a[e] = a[e] * s0[e] + t0[e];
              b[e] = b[e] * s1[e] + t1[e];
              a[e] = hn::relu(a[e]);
              b[e] = hn::relu(b[e]);
              if (p.range_flag && (hn::range_bad(a[e]) | hn::range_bad(b[e]))) *p.range_flag = 1;
              const _Float16 h0 = (_Float16)a[e], h1 = (_Float16)b[e];
              hi[e] = h0;
              hi[4 + e] = h1;
              lo[e] = (_Float16)(a[e] - (float)h0);
              lo[4 + e] = (_Float16)(b[e] - (float)h1);
            }
            xv[i][0] = hi;
            xv[i][1] = lo;
          }
        }
        const bool wrap = cb0 + kDepth >= cbs;   // the block kDepth ahead belongs to the next step
        issue(wrap ? step + 1 : step, wrap ? k : cb + kDepth, xr[k]);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE: sunk to the end of the loop body they are one unit deep
        const char* wf = wbank + cb * (kMT * 2 * 1024) + lane * 16;
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(wf + (m * 2) * 1024);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(wf + (m * 2 + 1) * 1024);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xv[i][1], acc[i][m], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xv[i][0], acc[i][m], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xv[i][0], acc[i][m], 0, 0, 0);
        }
      }
    }
    wg_barrier_lds();   // the previous step's outputs have been read out of the slots this step overwrites
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int s = slot0 + i * 16;
      s = s >= ring ? s - ring : s;
#pragma unroll
      for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m * 16 + lg * 4 + r;
          if (row < rows) ringp[row * ring + s] = acc[i][m][r];
        }
    }
    slot0 += kStepPx;
    slot0 = slot0 >= ring ? slot0 - ring : slot0;
    wg_barrier_lds();

    // outputs whose nine P columns are all in the ring
    const int p_hi = p0 + (step + 1) * kStepPx;
    const int e_hi = p_hi >= p1 ? o1 : min(o1, p_hi - halo);
    const int f = e_lo + tid;
    if (f < e_hi) {
      const unsigned hw = (unsigned)(L.h * L.w);
      const unsigned rem = (unsigned)f % hw;
      const int y = (int)(rem / (unsigned)L.w), x = (int)rem - y * L.w;
      const int s0 = (int)((unsigned)(f - p0) % (unsigned)ring);
      float v[kFlatMaxC];
#pragma unroll
      for (int oc = 0; oc < kFlatMaxC; ++oc) v[oc] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if ((unsigned)(y + dy) < (unsigned)L.h && (unsigned)(x + dx) < (unsigned)L.w) {
          int s = s0 + dy * L.w + dx;
          s = s < 0 ? s + ring : s;
          s = s >= ring ? s - ring : s;
          const float* col = ringp + tap * cout * ring + s;
#pragma unroll
          for (int oc = 0; oc < kFlatMaxC; ++oc)
            if (oc < cout) v[oc] += col[oc * ring];
        }
      }
      float* dst = L.y + (size_t)f * cout;
#pragma unroll
      for (int oc = 0; oc < kFlatMaxC; ++oc)
        if (oc < cout) {
          float o = v[oc] + bv[oc];
          if (oc < p.relu_cols) o = hn::relu(o);
          dst[oc] = o;
        }
    }
    e_lo = max(e_lo, e_hi);
  }
}

// ring words per P row: 256 new columns + one image row + 1 on either side of the outputs; (4 * ring) % 64 == 16 keeps the
// four 4-row lane groups of an accumulator store on distinct banks
int flat_ring_words(int wmax) {
  int ring = kStepPx + 2 * (wmax + 1);
  while (ring % 16 != 4) ++ring;
  return ring;
}

}  // namespace

// 1 when hn_conv3x3_thin_f16x3_levels would run the P-form kernel on this problem, 0 for the tap kernel (tests, profiles)
extern "C" int hn_conv3x3_thin_uses_flat(const hn_thin_levels* lv, int n, int cin, int cout) {
  if (!lv || lv->count < 1 || lv->count > HN_FCOS_MAX_LEVELS || hn::env_flags().thin_tap) return 0;   // HN_THIN_FORM=tap: A/B
  if (cout > kFlatMaxC || cin % (32 * kDepth)) return 0;
  int wmax = 1;
  int64_t all = 0;
  for (int l = 0; l < lv->count; ++l) {
    wmax = lv->w[l] > wmax ? lv->w[l] : wmax;
    all += (int64_t)n * lv->h[l] * lv->w[l];
  }
  // below ~64 k pixels (a single frame: 18 k) a workgroup's range is one or two steps and its fixed costs -- the 48 KB filter
  // bank, the W + 1 columns of lead-in -- outweigh the saved LDS reads: the tap kernel is 13 us per launch faster at batch 1
  if (all < 65536 && !hn::env_flags().thin_flat) return 0;
  const int64_t lds = (int64_t)(cin / 32) * (kMT * 2 * 1024) + (int64_t)9 * cout * flat_ring_words(wmax) * 4;
  return lds <= 160 * 1024 ? 1 : 0;
}

// images the scale / shift table can hold for this problem (0: no room)
static int aff_table_images(int cin, int cout, int wmax) {
  const int64_t fixed = (int64_t)(cin / 32) * (kMT * 2 * 1024) + (int64_t)((9 * cout * flat_ring_words(wmax) + 3) & ~3) * 4;
  const int64_t k = (160 * 1024 - fixed) / ((int64_t)2 * cin * 4);
  return k < kAffImagesMin ? 0 : (int)(k > kAffImagesMax ? kAffImagesMax : k);
}

// aff != nullptr: the AFFINE form (x = raw fp32, GroupNorm apply fused); lv->x16 is then ignored
static int thin_flat_run(const hn_thin_levels* lv, const hn_thin_affine* aff, int n, int cin, int cout, const void* w16,
                         const float* bias, int relu_cols, int xs, hipStream_t st) {
  FlatParams p;
  p.levels = lv->count; p.cbs = cin / 32; p.cout = cout; p.xs = xs; p.relu_cols = relu_cols;
  p.wt = (const _Float16*)w16; p.bias = bias;
  p.as = aff ? aff->affine_stride : 0;
  p.range_flag = aff ? hn::range_flag_ptr() : nullptr;
  int wmax = 1;
  int64_t all = 0;
  for (int l = 0; l < lv->count; ++l) {
    const int64_t total = (int64_t)n * lv->h[l] * lv->w[l];
    HN_CHECK_ARG(total < ((int64_t)1 << 30), "level %d: too many pixels", l);
    all += total;
    wmax = lv->w[l] > wmax ? lv->w[l] : wmax;
  }
  p.ring = flat_ring_words(wmax);
  const int aff_images = aff ? aff_table_images(cin, cout, wmax) : 0;
  const int lds = p.cbs * (kMT * 2 * 1024) + ((9 * cout * p.ring + 3) & ~3) * 4 + aff_images * 2 * cin * 4;
  static int cus[64] = {};
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  HN_CHECK_ARG(dev >= 0 && dev < 64, "device index out of range");
  if (!cus[dev]) {
    int c = 0;
    HN_CHECK_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_thin_flat_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_thin_flat_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    cus[dev] = c > 0 ? c : 256;
  }
  // one range of flat pixels per workgroup, one workgroup per CU (the LDS allows no more): the shortest range length
  // with which all levels fit in one wave of workgroups
  int64_t unit_len = (all + cus[dev] - 1) / cus[dev];
  unit_len = unit_len < kStepPx ? kStepPx : (unit_len + 63) / 64 * 64;
  for (;;) {
    int64_t units = 0;
    for (int l = 0; l < lv->count; ++l) units += ((int64_t)n * lv->h[l] * lv->w[l] + unit_len - 1) / unit_len;
    if (units <= cus[dev] || unit_len >= all) break;
    unit_len += 64;
  }
  int units = 0;
  for (int l = 0; l < HN_FCOS_MAX_LEVELS; ++l) {
    FlatLevel& t = p.lv[l];
    if (l < lv->count) {
      t.x = aff ? nullptr : (const _Float16*)lv->x16[l];
      t.xf = aff ? aff->x[l] : nullptr; t.scale = aff ? aff->scale[l] : nullptr; t.shift = aff ? aff->shift[l] : nullptr;
      t.y = lv->y[l]; t.h = lv->h[l]; t.w = lv->w[l];
      t.total = n * t.h * t.w;
      int64_t ul = unit_len;
      if (aff) {   // a range of R pixels touches at most (R - 2) / HW + 2 images: keep that within the LDS table
        const int64_t cap = (int64_t)(aff_images - 2) * t.h * t.w + 2 - 2 * (t.w + 1);
        ul = ul < cap ? ul : cap / 64 * 64;
        HN_CHECK_ARG(ul >= 64, "level %d: map too small for the fused GroupNorm apply (hn_conv3x3_thin_affine_applies)", l);
      }
      t.first_unit = units; t.unit_len = (int)ul;
      units += (int)((t.total + ul - 1) / ul);
    } else {
      t.x = nullptr; t.xf = t.scale = t.shift = nullptr; t.y = nullptr; t.h = t.w = t.total = 1; t.first_unit = 0x7fffffff; t.unit_len = 1;
    }
  }
  if (aff) hipLaunchKernelGGL(conv3x3_thin_flat_kernel<true>, dim3(units), dim3(kNT), lds, st, p);
  else hipLaunchKernelGGL(conv3x3_thin_flat_kernel<false>, dim3(units), dim3(kNT), lds, st, p);
  HN_CHECK_LAUNCH("conv3x3_thin_flat_kernel");
  return HN_OK;
}

// Does the fused form take this problem?  The plain P form must (hn_conv3x3_thin_uses_flat), the table must fit beside
// the filter bank and the ring, and every map must be large enough that a 64-pixel range stays within the table's images.
extern "C" int hn_conv3x3_thin_affine_applies(const hn_thin_levels* lv, int n, int cin, int cout) {
  if (!hn_conv3x3_thin_uses_flat(lv, n, cin, cout)) return 0;
  int wmax = 1;
  for (int l = 0; l < lv->count; ++l) wmax = lv->w[l] > wmax ? lv->w[l] : wmax;
  const int k = aff_table_images(cin, cout, wmax);
  if (!k) return 0;
  for (int l = 0; l < lv->count; ++l)
    if ((int64_t)(k - 2) * lv->h[l] * lv->w[l] + 2 - 2 * (lv->w[l] + 1) < 64) return 0;
  return 1;
}

extern "C" int hn_conv3x3_thin_affine_f16x3_levels(const hn_thin_levels* lv, const hn_thin_affine* aff, int n, int cin, int cout,
                                                   const void* w16, const float* bias, int relu_cols, void* stream) {
  HN_CHECK_ARG(lv && aff && w16, "hn_conv3x3_thin_affine_f16x3_levels: null pointer");
  HN_CHECK_ARG(lv->count >= 1 && lv->count <= HN_FCOS_MAX_LEVELS, "level count must be 1..%d", HN_FCOS_MAX_LEVELS);
  HN_CHECK_ARG(n > 0 && cin > 0 && cout >= 1 && relu_cols >= 0 && relu_cols <= cout, "bad dims");
  HN_CHECK_ARG(hn_conv3x3_thin_affine_applies(lv, n, cin, cout), "the fused form does not take this problem "
               "(hn_conv3x3_thin_affine_applies): run hn_affine_split_f32_levels + hn_conv3x3_thin_f16x3_levels");
  HN_CHECK_ARG(aff->in_pix_stride >= cin && aff->in_pix_stride % 4 == 0 && aff->affine_stride >= cin && aff->affine_stride % 4 == 0 &&
                   (uintptr_t)w16 % 16 == 0, "bad strides / unaligned filter bank");
  for (int l = 0; l < lv->count; ++l) {
    HN_CHECK_ARG(aff->x[l] && aff->scale[l] && aff->shift[l] && lv->y[l] && lv->h[l] > 0 && lv->w[l] > 0,
                 "level %d: null pointer or empty map", l);
    HN_CHECK_ARG((uintptr_t)aff->x[l] % 16 == 0 && (uintptr_t)aff->scale[l] % 16 == 0 && (uintptr_t)aff->shift[l] % 16 == 0,
                 "level %d: unaligned tensor", l);
  }
  return thin_flat_run(lv, aff, n, cin, cout, w16, bias, relu_cols, aff->in_pix_stride, (hipStream_t)stream);
}

#define HN_TRY_THIN(expr)       \
  do {                          \
    const int st_ = (expr);     \
    if (st_ != HN_OK) return st_; \
  } while (0)

// the tap kernel's launch description: scalars (checked), then the level table and the workgroup count
static int thin_tap_scalars(const hn_thin_levels* lv, int n, int cin, int cout, const void* w16, const float* bias, int relu_cols,
                            int in_pix_stride, ThinParams& p) {
  HN_CHECK_ARG(lv && w16, "hn_conv3x3_thin_f16x3_levels: null pointer");
  HN_CHECK_ARG(lv->count >= 1 && lv->count <= HN_FCOS_MAX_LEVELS, "level count must be 1..%d", HN_FCOS_MAX_LEVELS);
  HN_CHECK_ARG(n > 0 && cin > 0 && cin % 32 == 0 && cout >= 1 && cout <= 16, "need cin %% 32 == 0 and 1 <= cout <= 16");
  HN_CHECK_ARG(relu_cols >= 0 && relu_cols <= cout, "bad relu_cols");
  p.levels = lv->count; p.n = n; p.cbs = cin / 32; p.cout = cout;
  p.xs = in_pix_stride ? in_pix_stride : 2 * cin;
  p.ys = cout; p.relu_cols = relu_cols;
  p.wt = (const _Float16*)w16; p.bias = bias;
  HN_CHECK_ARG(p.xs >= 2 * cin && p.xs % 64 == 0 && (uintptr_t)w16 % 16 == 0, "bad input pixel stride / unaligned filter bank");
  const int64_t wbytes = (int64_t)cout * p.cbs * 9 * 128;
  HN_CHECK_ARG(wbytes < ((int64_t)1 << 31), "filter bank too large");
  p.w_records = (unsigned)wbytes;
  for (int l = 0; l < lv->count; ++l) {
    HN_CHECK_ARG(lv->x16[l] && lv->y[l] && lv->h[l] > 0 && lv->w[l] > 0, "level %d: null pointer or empty map", l);
    HN_CHECK_ARG((uintptr_t)lv->x16[l] % 16 == 0, "level %d: unaligned input", l);
  }
  return HN_OK;
}

static int thin_tap_levels(const hn_thin_levels* lv, int n, ThinParams& p, int* nblocks) {
  int blocks = 0;
  for (int l = 0; l < HN_FCOS_MAX_LEVELS; ++l) {
    ThinLevel& t = p.lv[l];
    if (l < lv->count) {
      const int64_t xbytes = (int64_t)n * lv->h[l] * lv->w[l] * p.xs * 2;
      HN_CHECK_ARG(xbytes < ((int64_t)1 << 31), "level %d: input too large for 32-bit offsets", l);
      t.x = (const _Float16*)lv->x16[l]; t.y = lv->y[l]; t.h = lv->h[l]; t.w = lv->w[l];
      t.ty = hn::cdiv(t.h, kT); t.tx = hn::cdiv(t.w, kT);
      t.first_block = blocks;
      t.x_records = (unsigned)xbytes;
      HN_CHECK_ARG((int64_t)blocks + (int64_t)n * t.ty * t.tx < (int64_t)1 << 30, "too many tiles");
      blocks += n * t.ty * t.tx;
    } else {
      t.x = nullptr; t.y = nullptr; t.h = t.w = t.ty = t.tx = 1; t.first_block = 0x7fffffff; t.x_records = 0;
    }
  }
  *nblocks = blocks;
  return HN_OK;
}

static int thin_tap_prepare() {
  constexpr int LDS_BYTES = 2 * kStageBytes;
  static_assert(LDS_BYTES <= 160 * 1024 - 2048, "one 8-wave workgroup per CU");
  static bool attr_set[64] = {};
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_thin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_thin_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  return HN_OK;
}

extern "C" int hn_conv3x3_thin_f16x3_levels(const hn_thin_levels* lv, int n, int cin, int cout, const void* w16, const float* bias,
                                            int relu_cols, int in_pix_stride, void* stream) {
  ThinParams p;
  HN_TRY_THIN(thin_tap_scalars(lv, n, cin, cout, w16, bias, relu_cols, in_pix_stride, p));
  if (hn_conv3x3_thin_uses_flat(lv, n, cin, cout))
    return thin_flat_run(lv, nullptr, n, cin, cout, w16, bias, relu_cols, p.xs, (hipStream_t)stream);
  int blocks = 0;
  HN_TRY_THIN(thin_tap_levels(lv, n, p, &blocks));
  HN_TRY_THIN(thin_tap_prepare());
  hipLaunchKernelGGL(conv3x3_thin_kernel, dim3(blocks), dim3(kNT), 2 * kStageBytes, (hipStream_t)stream, p);
  HN_CHECK_LAUNCH("conv3x3_thin_kernel");
  return HN_OK;
}

extern "C" int hn_conv3x3_thin_f16x3_levels_group(const hn_thin_member* members, int count, int n, int cin, int in_pix_stride,
                                                  void* stream) {
  HN_CHECK_ARG(members && count >= 1 && count <= kThinGroupMax, "hn_conv3x3_thin_f16x3_levels_group: 1..%d members", kThinGroupMax);
  // the P form (large pixel counts: HBM-bound streams, nothing to gain from sharing a grid) and single members: member by member
  bool flat = false;
  for (int m = 0; m < count; ++m)
    flat = flat || hn_conv3x3_thin_uses_flat(&members[m].lv, n, cin, members[m].cout);
  if (flat || count == 1 || hn::env_flags().thin_nogroup) {
    for (int m = 0; m < count; ++m)
      HN_TRY_THIN(hn_conv3x3_thin_f16x3_levels(&members[m].lv, n, cin, members[m].cout, members[m].w16, members[m].bias,
                                               members[m].relu_cols, in_pix_stride, stream));
    return HN_OK;
  }
  ThinGroup g;
  int first[kThinGroupMax + 1] = {0, 0x7fffffff, 0x7fffffff, 0};
  int total = 0;
  for (int m = 0; m < kThinGroupMax; ++m) {
    const hn_thin_member& mm = members[m < count ? m : 0];   // (absent members repeat member 0: never selected)
    int blocks = 0;
    HN_TRY_THIN(thin_tap_scalars(&mm.lv, n, cin, mm.cout, mm.w16, mm.bias, mm.relu_cols, in_pix_stride, g.m[m]));
    HN_TRY_THIN(thin_tap_levels(&mm.lv, n, g.m[m], &blocks));
    if (m < count) {
      first[m] = total;
      total += blocks;
      HN_CHECK_ARG(total < (1 << 30), "too many tiles");
    }
  }
  g.first1 = count > 1 ? first[1] : 0x7fffffff;
  g.first2 = count > 2 ? first[2] : 0x7fffffff;
  HN_TRY_THIN(thin_tap_prepare());
  hipLaunchKernelGGL(conv3x3_thin_group_kernel, dim3(total), dim3(kNT), 2 * kStageBytes, (hipStream_t)stream, g);
  HN_CHECK_LAUNCH("conv3x3_thin_group_kernel");
  return HN_OK;
}
