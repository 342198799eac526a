// Stem convolution (7x7 / stride 2, 4-channel stem image) + bias + ReLU + 3x3 / stride-2 / pad-1 max pooling as a DIRECT
// convolution from an LDS-resident image patch (round 3).
//
// Replaces, for the ResNet stems of this path (torchvision resnet34 conv1..maxpool at fcos_utils/fcos.py:737 and
// a2j/resnet.py:155-158), the implicit-GEMM form of the same fusion (conv_igemm_f16x3_kernel<..., POOL>): there every
// one of the 7 filter rows re-gathered its 256 x 64-byte im2col rows through LDS-DMA -- 280 KB per workgroup for a
// 22 KB image patch, because a stride-2 7x7 window overlaps its neighbours 3.5x vertically and 4x horizontally.  Here a
// workgroup (4 waves) loads the patch ONCE (35 rows x 40 pixels x 2 planes = 22 KB) plus the whole filter bank
// (64 x 7 x 128 B = 56 KB), waits once, and runs all 7 k tiles out of LDS with no further DMA and no barrier:
//   * output tile = a 15 x 17 patch of conv pixels (row-major, 255 of the 256 MFMA rows) = 7 x 8 pooled pixels,
//   * A fragment of conv pixel (oy, ox), filter row ky, lane group g: the 16 bytes at patch[(2 oy + ky)][2 ox + 2 g]
//     (k = kx * 4 + c, two pixels per 8-half chunk) -- a plain ds_read_b128 at a per-lane address,
//   * W fragments from the resident bank; MFMA operands swapped (lane = pixel, registers = channels) and the three
//     terms per k tile in the order of the implicit-GEMM kernel (lo*hi, hi*lo, hi*hi), so results are BIT-IDENTICAL to
//     hn_conv_stem_f16x3 + hn_maxpool3x3s2_s32,
//   * epilogue: bias + ReLU patch -> LDS (fp32), 3x3 / stride-2 max over the patch, S32 (hi | lo) store of the pooled map.
// Conv pixels outside the map (the pooling's padding ring, partial patches at the bottom / right edge) read clamped
// image rows / pieces and are zeroed before pooling (every pooling window holds a real pixel and ReLU >= 0).
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int kPR = 15, kPC = 17;            // conv patch (rows x columns)
constexpr int kQR = 7, kQC = 8;              // pooled pixels per patch
constexpr int kIR = 2 * (kPR - 1) + 7;       // 35 image rows
constexpr int kIP = (2 * (kPC - 1) + 8) / 2; // 20 sixteen-byte pieces (2 pixels each) per image row
constexpr int kPlanePieces = kIR * kIP;      // 700 pieces per plane
constexpr int kPatchPieces = ((2 * kPlanePieces + 255) / 256) * 256;   // both planes, rounded up to whole DMA rounds: 1536
constexpr int kWPieces = 64 * 7 * 8;         // 64 output channels x 7 filter rows x 128 bytes
constexpr int kCout = 64;
constexpr int kPitch = kCout + 4;            // fp32 patch pitch of the pooling stage

struct StemParams {
  const _Float16* x;   // stem image: planes hi, lo of [n][hb][wb][4]
  const _Float16* w;   // [64][7][2][32]
  const float* bias;
  _Float16* y;         // pooled S32 [n][poh][pow][64/32][2][32]
  int n, hb, wb, oh, ow, poh, pow_, ty, tx;
  long plane;          // halfs between the hi and the lo plane
  int terms;          // 3: the split-precision product; 1: hi*hi only (the f16x1 throughput mode)
  int* range_flag;
};

__global__ __launch_bounds__(256, 2) void conv_stem_pool_direct_kernel(const StemParams p) {
  // LDS: image patch (24 KB, piece-linear: plane, row, piece) | filter bank (56 KB, the global layout) -- 80 KB; the
  // fp32 pooling patch (256 x 68 x 4 = 68 KB) reuses it after the MFMAs
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  _Float16* patch = reinterpret_cast<_Float16*>(smem);
  _Float16* wbank = reinterpret_cast<_Float16*>(smem + kPatchPieces * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, px = lane & 15;
  int lid;
  {  // XCD-aware remap (blocks are dealt round-robin over the 8 XCDs): each XCD works on a contiguous run of patches
    const int bid = blockIdx.x, nb = gridDim.x;
    const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int per_img = p.ty * p.tx;
  const int img = lid / per_img;
  const int rem = lid - img * per_img;
  const int pty = rem / p.tx, ptx = rem - pty * p.tx;
  const int oy0 = 2 * kQR * pty - 1, ox0 = 2 * kQC * ptx - 1;   // first conv pixel of the patch (may be -1)
  const int r0 = 2 * oy0, c0p = ox0;                            // first image row / first PIECE column (2 pixels per piece)

  // ---- one round of loads: image patch (both planes) then the filter bank; 16 bytes per lane, wave-linear LDS destination ----
  const int wave_base = __builtin_amdgcn_readfirstlane(wave) * 64;
#pragma unroll
  for (int rnd = 0; rnd < kPatchPieces / 256; ++rnd) {
    const int q = rnd * 256 + tid;
    int pl = q >= kPlanePieces ? 1 : 0;
    int k = q - pl * kPlanePieces;
    if (k >= kPlanePieces) { pl = 1; k = kPlanePieces - 1; }    // padding pieces of the last round: any valid source
    const int row = k / kIP, pc = k - row * kIP;
    int sr = r0 + row, sc = c0p + pc;
    sr = sr < 0 ? 0 : (sr < p.hb ? sr : p.hb - 1);
    sc = sc < 0 ? 0 : (sc < (p.wb >> 1) ? sc : (p.wb >> 1) - 1);
    const _Float16* src = p.x + pl * p.plane + (((long)img * p.hb + sr) * p.wb + sc * 2) * 4;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + (rnd * 256 + wave_base) * 16), 16, 0, 0);
  }
#pragma unroll
  for (int rnd = 0; rnd < kWPieces / 256; ++rnd) {
    const _Float16* src = p.w + (long)(rnd * 256 + tid) * 8;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(reinterpret_cast<char*>(wbank) + (rnd * 256 + wave_base) * 16), 16, 0,
                                     0);
  }

  // ---- per-lane fragment addresses (bytes into the patch / bank), computed under the loads ----
  // wave w owns conv-patch rows [64 w, 64 w + 64): tile i, lane pixel px -> patch pixel (pr, pc)
  int a_off[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = wave * 64 + i * 16 + px;
    ok[i] = row < kPR * kPC;
    row = ok[i] ? row : kPR * kPC - 1;
    const int pr = row / kPC, pc = row - pr * kPC;
    const int oy = oy0 + pr, ox = ox0 + pc;
    ok[i] = ok[i] && (unsigned)oy < (unsigned)p.oh && (unsigned)ox < (unsigned)p.ow;
    a_off[i] = ((2 * pr) * kIP + pc + lg) * 16;   // filter row ky adds ky * kIP pieces; the lo plane kPlanePieces pieces
  }
  // W fragment of column tile j: output channel 16 j + px, chunk lg of the hi run (lo run 64 bytes on)
  const int b_off = (px * 7) * 128 + lg * 16;   // + j * 16 * 7 * 128 + ky * 128

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const char* pa = reinterpret_cast<const char*>(patch);
  const char* pb = reinterpret_cast<const char*>(wbank);
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(pa + a_off[i] + ky * (kIP * 16));
      al[i] = *reinterpret_cast<const f16x8*>(pa + a_off[i] + ky * (kIP * 16) + kPlanePieces * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *reinterpret_cast<const f16x8*>(pb + b_off + j * (16 * 7 * 128) + ky * 128);
      bl[j] = *reinterpret_cast<const f16x8*>(pb + b_off + j * (16 * 7 * 128) + ky * 128 + 64);
    }
    // same term order per accumulator as conv_igemm_f16x3_kernel (lo*hi, hi*lo, hi*hi); W fragment = srcA (lane = pixel)
    if (p.terms == 3) {   // (wave-uniform; terms == 1 issues the hi*hi products alone, like the TERMS = 1 kernels)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
  }

  // ---- epilogue: bias + ReLU (zero for conv pixels outside the map) -> fp32 patch in LDS -> 3x3 / stride-2 max -> S32 ----
  __syncthreads();   // every wave is done reading the image patch / filter bank
  float* fp = reinterpret_cast<float*>(smem);
  const int nsub = (lg & 1) * 16 + (lg >> 1) * 8;   // channel offset inside a pair of column tiles after the row exchange
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 64 + i * 16 + px;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // rows {1, 3} of x <-> rows {0, 2} of y: eight consecutive channels per lane
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
        x[r] = __uint_as_float(s[0]);
        y[r] = __uint_as_float(s[1]);
      }
      const int n = jp * 32 + nsub;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = ok[i] ? hn::relu(x[e] + b0[e]) : 0.f;
        y[e] = ok[i] ? hn::relu(y[e] + b1[e]) : 0.f;
      }
      *reinterpret_cast<f32x4*>(fp + row * kPitch + n) = x;
      *reinterpret_cast<f32x4*>(fp + row * kPitch + n + 4) = y;
    }
  }
  __syncthreads();
  for (int item = tid; item < kQR * kQC * (kCout / 8); item += 256) {
    const int c8 = item & (kCout / 8 - 1), pp = item / (kCout / 8);
    const int pr = pp / kQC, pc = pp - pr * kQC;
    const int gy = kQR * pty + pr, gx = kQC * ptx + pc;
    if (gy >= p.poh || gx >= p.pow_) continue;
    f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0;   // ReLU output: 0 is the identity of max here
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* src = fp + ((2 * pr + dy) * kPC + 2 * pc + dx) * kPitch + c8 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m0[e] = hn::max_nan(m0[e], a[e]);
          m1[e] = hn::max_nan(m1[e], b[e]);
        }
      }
    f16x8 hi, lo;
    if (p.range_flag) {
      float mg = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) mg = hn::max_nan(mg, hn::range_mag(m0[e], m1[e]));
      if (!(mg <= 65504.f)) *p.range_flag = 1;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 h0 = (_Float16)m0[e], h1 = (_Float16)m1[e];
      hi[e] = h0;
      hi[4 + e] = h1;
      lo[e] = (_Float16)(m0[e] - (float)h0);
      lo[4 + e] = (_Float16)(m1[e] - (float)h1);
    }
    _Float16* dst = p.y + (((long)img * p.poh + gy) * p.pow_ + gx) * (2 * kCout) + (c8 >> 2) * 64 + (c8 & 3) * 8;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 32) = lo;
  }
}

}  // namespace

namespace hn {

// hn_conv_stem_pool_f16x3 for the shape every ResNet stem of this path has (7x7 / stride 2 / pad 3, 64 output channels)
int stem_pool_direct(const void* x16, int n, int ph, int pw, const void* w16, const float* bias, void* y, int terms, hipStream_t st) {
  const int hb = ph + 6, wb = pw + 6;
  StemParams p;
  p.x = (const _Float16*)x16; p.w = (const _Float16*)w16; p.bias = bias; p.y = (_Float16*)y;
  p.n = n; p.hb = hb; p.wb = wb;
  p.oh = (hb - 7) / 2 + 1; p.ow = (wb - 7) / 2 + 1;
  p.poh = (p.oh + 2 - 3) / 2 + 1; p.pow_ = (p.ow + 2 - 3) / 2 + 1;
  p.ty = hn::cdiv(p.poh, kQR); p.tx = hn::cdiv(p.pow_, kQC);
  p.plane = (long)n * hb * wb * 4;
  p.terms = terms == 1 ? 1 : 3;
  p.range_flag = hn::range_flag_ptr();
  HN_CHECK_ARG(wb % 2 == 0 && (uintptr_t)x16 % 16 == 0 && (uintptr_t)w16 % 16 == 0 && (uintptr_t)y % 16 == 0 &&
                   (uintptr_t)bias % 16 == 0, "stem image rows / tensors must be 16-byte aligned");
  HN_CHECK_ARG((int64_t)n * p.ty * p.tx < (int64_t)1 << 31, "too many patches");
  constexpr int LDS_BYTES = kPatchPieces * 16 + kWPieces * 16;
  static_assert(LDS_BYTES <= 80 * 1024 && 256 * kPitch * 4 <= LDS_BYTES, "two workgroups per CU; the pooling patch reuses the space");
  static bool attr_set[64] = {};
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_stem_pool_direct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(conv_stem_pool_direct_kernel, dim3(n * p.ty * p.tx), dim3(256), LDS_BYTES, st, p);
  HN_CHECK_LAUNCH("conv_stem_pool_direct_kernel");
  return HN_OK;
}

}  // namespace hn
