// 3x3 / stride 1 / pad 1 convolution with 64 output channels as a DIRECT convolution from an LDS-resident 2-D halo patch
// (round 3; the ResNet-34 layer1 convolutions of the FCOS trunk, torchvision resnet34 at fcos_utils/fcos.py:737, and the
// 64 -> 64 3x3 of the A2J layer1 bottlenecks, a2j/resnet.py:78-96).
//
// Why a second kernel for this shape: with only 64 output columns the implicit-GEMM kernel (conv_igemm_f16x3.hip, 256x64 tile)
// moves 40 KB of operands through LDS-DMA per 256 x 64 x 32 MACs -- 83 % of the time its MFMAs need, on a k loop of just 18
// tiles -- and re-gathers every input pixel nine times (once per tap): 298 TFLOP/s, 0.12 of the f16 peak, 9 % of the
// batch-32 step.  The row-shared form (v7) fetches a pixel three times but does not fit two 256-row workgroups on a CU.
// Here a workgroup (4 waves, 16 x 16 output pixels x 64 channels) stages the 18 x 18 input patch of ONE 32-channel block
// once (41 KB) and reads all nine taps' A fragments from it at (dy, dx) offsets; only the 8 KB filter tile of a tap is
// streamed (a 4-stage ring, three taps ahead).  Operand traffic per workgroup: 226 KB instead of 720 KB; LDS 76 KB, two workgroups per CU.
//   * k order: 32-channel block outer, taps (dy, dx) inner, terms lo*hi, hi*lo, hi*hi -- the implicit-GEMM kernel's order,
//     so results are bit-identical to it;
//   * patch pixels outside the image are zero-filled by the buffer descriptor's range check (offset bit 31), so padding
//     costs nothing; the patch and the filter tiles use the bank swizzle of the big kernel (chunk ^ ((row >> 1) & 7));
//   * MFMA operands swapped (lane = pixel, registers = channels) + v_permlane16_swap: the epilogue works from registers
//     with 16-byte accesses (bias, S32 residual of the output's own shape, ReLU, S32 store).
#include "hn_common.h"

#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kT = 16;                 // output tile edge
constexpr int kP = kT + 2;             // patch edge
constexpr int kPatchPix = kP * kP;     // 324 pixels x 128 B
constexpr int kPatchRounds = (kPatchPix * 8 + 255) / 256;   // 11 DMA rounds of 256 sixteen-byte pieces
constexpr int kPatchBytes = kPatchRounds * 256 * 16;        // 45056 (the tail of the last round lands in padding)
constexpr int kWBytes = 64 * 128;      // one tap's filter tile: 64 output channels x (hi 64 B | lo 64 B)

struct HaloParams {
  const _Float16* x;   // S32 [n][h][w][cin/32][2][32] (pixel stride xs halfs)
  const _Float16* wt;  // [64][(cin/32) * 9][2][32]
  const float* bias;
  const _Float16* res; // S32 residual of the output's shape, or null
  _Float16* y;         // S32 [n][h][w][2][2][32]
  int n, h, wd, cbs, xs, ys, rs, relu, ty, tx;
  unsigned x_records, w_records;
  int* range_flag;
  unsigned long long* stamps;   // diagnostics (HN_HALO_STAMPS=1): s_memtime at 4 points per workgroup, else null
};

__device__ unsigned long long g_halo_stamps[4 * 8192];

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const HaloParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* patch = smem;                       // [324 pixels][8 chunks of 16 B], chunk c of pixel q at position c ^ swz(q)
  char* wst = smem + kPatchBytes;           // four filter-tile stages (ring)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, px = lane & 15;
  int lid;
  {
    const int bid = blockIdx.x, nb = gridDim.x;
    const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int per_img = p.ty * p.tx;
  const int img = lid / per_img;
  const int rem = lid - img * per_img;
  const int tyi = rem / p.tx, txi = rem - tyi * p.tx;
  const int y0 = tyi * kT, x0 = txi * kT;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)p.w_records, 0x00020000);
  const int wave_base = __builtin_amdgcn_readfirstlane(wave) * 64;

  // ---- loop-invariant DMA offsets: patch piece (round r, lane) -> pixel q = piece / 8, LDS position pos = piece % 8 ----
  unsigned a_off[kPatchRounds];
#pragma unroll
  for (int r = 0; r < kPatchRounds; ++r) {
    const int piece = r * 256 + tid;
    int q = piece >> 3;
    const int pos = piece & 7;
    const bool in_patch = q < kPatchPix;
    q = in_patch ? q : kPatchPix - 1;
    const int py = q / kP, pxx = q - py * kP;
    const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
    const bool ok = in_patch && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.wd;
    const int chunk = pos ^ swz(q);   // source chunk that belongs at this position: 0-3 hi run, 4-7 lo run
    const unsigned off = (((unsigned)(img * p.h + iy) * (unsigned)p.wd + (unsigned)ix) * (unsigned)p.xs + (unsigned)(chunk * 8)) * 2u;
    a_off[r] = ok ? off : 0x80000000u;   // out of range: the descriptor's range check returns zeros
  }
  // filter tile piece (round r of 2, lane): row = output channel, position pos -> chunk pos ^ swz(row)
  unsigned b_off[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int piece = r * 256 + tid;
    const int row = piece >> 3, pos = piece & 7;
    b_off[r] = ((unsigned)row * (unsigned)(p.cbs * 9) * 64u + (unsigned)((pos ^ swz(row)) * 8)) * 2u;
  }
  auto dma_patch = [&](int cb) {
#pragma unroll
    for (int r = 0; r < kPatchRounds; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_void*)(patch + (r * 256 + wave_base) * 16), 16, (int)a_off[r], cb * 128, 0, 0);
  };
  auto dma_w = [&](int kt, int stage) {   // kt = cb * 9 + tap
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(wst + stage * kWBytes + (r * 256 + wave_base) * 16), 16,
                                               (int)b_off[r], kt * 128, 0, 0);
  };

  // ---- fragment addresses: wave w owns output rows 4w .. 4w+3 (row tile i = output row 4w + i, lane pixel = column px) ----
  int pp0[4];   // patch pixel of the (dy, dx) = (0, 0) tap of this lane's pixel in row tile i
#pragma unroll
  for (int i = 0; i < 4; ++i) pp0[i] = (wave * 4 + i) * kP + px;
  const int brow = px;   // W fragment of column tile j: output channel 16 j + px
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Filter tiles run THREE taps ahead in a 4-stage ring (8 KB each): the L2 -> LDS latency of a tile (~2.5 k cycles) is
  // longer than a tap's 48 MFMAs (768 cycles), so with the usual one-step lookahead every tap waited for its filter tile
  // (measured: the workgroup lived 62 k cycles for 14 k cycles of MFMA work).  Counted vmcnt waits: at the top of tap kt
  // the two younger tiles (2 DMA instructions each) may still be in flight.
  if (p.stamps && tid == 0 && blockIdx.x < 8192) p.stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
  dma_patch(0);
  const int ktiles = p.cbs * 9;
  dma_w(0, 0);
  if (1 < ktiles) dma_w(1, 1);
  if (2 < ktiles) dma_w(2, 2);
  // epilogue operands are requested under the k loop: the bias here, the S32 residual two taps before the end (the kernel
  // has the registers for it: 64 for the residual of a lane's 64 outputs), so the epilogue starts without a memory latency
  const int nsub = (lg & 1) * 16 + (lg >> 1) * 8;   // channel offset of this lane inside a pair of column tiles (see the epilogue)
  const bool has_res = p.res != nullptr;
  long opix[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = y0 + wave * 4 + i, ox = x0 + px;
    ok[i] = oy < p.h && ox < p.wd;
    opix[i] = ((long)img * p.h + (oy < p.h ? oy : p.h - 1)) * p.wd + (ox < p.wd ? ox : p.wd - 1);
  }
  f32x4 b0[2], b1[2];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int n = jp * 32 + nsub;
    b0[jp] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    b1[jp] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f16x8 rh[8], rl[8];
  int tap = 0, cb = 0;
#pragma unroll 1
  for (int kt = 0; kt < ktiles; ++kt) {
    if (tap == 0 && kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the patch of this channel block
    else if (kt + 1 >= ktiles) {
      // last tile: it is older than the 16 residual loads issued one tap ago, which may stay in flight
      if (has_res && ktiles >= 2 && tap != 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (kt + 2 >= ktiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();   // tile kt (and the patch) landed for every wave; nobody reads stage (kt + 3) & 3 = (kt - 1) & 3 any more
    if (has_res && kt + 2 == ktiles) {   // all residual pieces of this lane, one tap ahead of the end
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int n = jp * 32 + nsub;
          const _Float16* q = p.res + opix[i] * p.rs + (n >> 5) * 64 + (n & 31);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rh[i * 2 + jp]) : "v"(q) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(rl[i * 2 + jp]) : "v"(q) : "memory");
        }
    }
    if (kt == 0 && p.stamps && tid == 0 && blockIdx.x < 8192) p.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
    if (kt + 3 < ktiles) dma_w(kt + 3, (kt + 3) & 3);
    const int dy = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), dx = tap - dy * 3;
    const char* ws = wst + (kt & 3) * kWBytes;
    f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = pp0[i] + dy * kP + dx;
      const int sw = swz(pp);
      ah[i] = *reinterpret_cast<const f16x8*>(patch + pp * 128 + ((lg ^ sw) << 4));
      al[i] = *reinterpret_cast<const f16x8*>(patch + pp * 128 + (((4 + lg) ^ sw) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 16 + brow;
      const int sw = swz(row);
      bh[j] = *reinterpret_cast<const f16x8*>(ws + row * 128 + ((lg ^ sw) << 4));
      bl[j] = *reinterpret_cast<const f16x8*>(ws + row * 128 + (((4 + lg) ^ sw) << 4));
    }
    // term order of conv_igemm_f16x3_kernel: lo*hi, hi*lo, hi*hi; W fragment = srcA (lane = pixel, registers = channels)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    if (++tap == 9) {
      tap = 0;
      ++cb;
      if (cb < p.cbs) {   // next channel block: every wave is done with the patch, then reload it (its filter tiles are in flight)
        __syncthreads();
        dma_patch(cb);
      }
    }
  }

  if (p.stamps && tid == 0 && blockIdx.x < 8192) p.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
  // ---- epilogue: bias, residual (S32, output's shape), ReLU, S32 store; eight consecutive channels per lane ----
  if (has_res) {
    if (ktiles < 2) {   // (never for the shapes routed here: the prefetch point did not exist)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int n = jp * 32 + nsub;
          const _Float16* q = p.res + opix[i] * p.rs + (n >> 5) * 64 + (n & 31);
          rh[i * 2 + jp] = *reinterpret_cast<const f16x8*>(q);
          rl[i * 2 + jp] = *reinterpret_cast<const f16x8*>(q + 32);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the asm loads above are invisible to the compiler's counters
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // rows {1, 3} of x <-> rows {0, 2} of y (16-lane rows)
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
        x[r] = __uint_as_float(s[0]);
        y[r] = __uint_as_float(s[1]);
      }
      if (!ok[i]) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = x[e] + b0[jp][e];
        v[4 + e] = y[e] + b1[jp][e];
      }
      if (has_res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rh[i * 2 + jp][e] + (float)rl[i * 2 + jp][e];
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = hn::relu(v[e]);
      }
      f16x8 hi, lo;
      if (p.range_flag) hn::range_note_n<8>(p.range_flag, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
      }
      const int n = jp * 32 + nsub;
      _Float16* q = p.y + opix[i] * p.ys + (n >> 5) * 64 + (n & 31);
      *reinterpret_cast<f16x8*>(q) = hi;
      *reinterpret_cast<f16x8*>(q + 32) = lo;
    }
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 8192) p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace

extern "C" int hnx_debug_halo_stamps(unsigned long long* host, int count) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_halo_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}

namespace hn {

bool conv3x3_halo_applies(const hn_conv_desc* d, bool has_gn, bool has_group, const void* residual) {
  return !env_flags().no_halo && !has_gn && !has_group && d->cout == 64 && d->cin % 32 == 0 && d->cin <= 256 && d->r == 3 && d->s == 3 &&
         d->stride == 1 && d->pad == 1 && d->dil == 1 && d->out_split == 1 && d->relu_cols == 64 &&
         (d->res_mode == 0 || (d->res_mode == 1 && d->res_split == 1 && residual)) && d->tile == HN_TILE_AUTO &&
         // worth it once the grid fills the chip twice over (few tiles: the 128x64 row-shared form has less padding waste)
         (int64_t)d->n * cdiv(d->h, kT) * cdiv(d->w, kT) >= 512;
}

// the operand conditions of the halo kernel (16-byte alignment, S32 pixel strides, extents below 2 GB): a call that the shape
// predicate above routes here but that does not meet them takes the implicit-GEMM kernels instead of failing
bool conv3x3_halo_operands_ok(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                              const void* y) {
  const int xs = d->in_pix_stride ? d->in_pix_stride : 2 * d->cin;
  const int ys = d->out_pix_stride ? d->out_pix_stride : 2 * d->cout;
  const int rs = d->res_pix_stride ? d->res_pix_stride : 2 * d->cout;
  const int64_t xbytes = (int64_t)d->n * d->h * d->w * xs * 2, wbytes = (int64_t)64 * (d->cin / 32) * 9 * 128;
  return xbytes < ((int64_t)1 << 31) && wbytes < ((int64_t)1 << 31) && (uintptr_t)x16 % 16 == 0 && (uintptr_t)w16 % 16 == 0 &&
         (uintptr_t)y % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0) &&
         xs % 64 == 0 && ys % 64 == 0 && rs % 64 == 0;
}

int conv3x3_halo(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                 hipStream_t st) {
  HaloParams p;
  p.x = (const _Float16*)x16; p.wt = (const _Float16*)w16; p.bias = bias; p.res = (const _Float16*)residual; p.y = (_Float16*)y;
  p.n = d->n; p.h = d->h; p.wd = d->w; p.cbs = d->cin / 32;
  p.xs = d->in_pix_stride ? d->in_pix_stride : 2 * d->cin;
  p.ys = d->out_pix_stride ? d->out_pix_stride : 2 * d->cout;
  p.rs = d->res_pix_stride ? d->res_pix_stride : 2 * d->cout;
  p.relu = 1;
  p.ty = cdiv(d->h, kT); p.tx = cdiv(d->w, kT);
  p.range_flag = range_flag_ptr();
  p.stamps = nullptr;
  if (env_flags().halo_stamps) {
    void* sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(g_halo_stamps)) == hipSuccess) p.stamps = (unsigned long long*)sp;
  }
  const int64_t xbytes = (int64_t)d->n * d->h * d->w * p.xs * 2, wbytes = (int64_t)64 * p.cbs * 9 * 128;
  HN_CHECK_ARG(xbytes < ((int64_t)1 << 31) && wbytes < ((int64_t)1 << 31), "operand too large for the halo kernel");
  HN_CHECK_ARG((uintptr_t)x16 % 16 == 0 && (uintptr_t)w16 % 16 == 0 && (uintptr_t)y % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
                   (!residual || (uintptr_t)residual % 16 == 0) && p.xs % 64 == 0 && p.ys % 64 == 0 && p.rs % 64 == 0,
               "halo kernel needs 16-byte aligned tensors and S32 pixel strides");
  p.x_records = (unsigned)xbytes;
  p.w_records = (unsigned)wbytes;
  constexpr int LDS_BYTES = kPatchBytes + 4 * kWBytes;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
  static bool attr_set[64] = {};
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(d->n * p.ty * p.tx), dim3(256), LDS_BYTES, st, p);
  HN_CHECK_LAUNCH("conv3x3_halo_kernel");
  return HN_OK;
}

}  // namespace hn
