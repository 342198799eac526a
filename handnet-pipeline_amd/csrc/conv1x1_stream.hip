// 1x1 / stride-1 convolutions with a SHORT k (Cin = 64 or 128) and many output channels (Cout % 256 == 0) on many pixels, as a
// streaming kernel (round 5): the FPN P3 lateral (torchvision FeaturePyramidNetwork inner_blocks[0], 128 -> 256 at 100 x 136,
// called at fcos_utils/fcos.py:737) and the 64 -> 256 / 128 -> 512 expansions of the A2J bottlenecks (a2j/resnet.py:78-96).
//
// Why a third convolution kernel: with 2-4 k tiles the implicit-GEMM kernel (conv_igemm_f16x3.hip) spends a workgroup's life in
// its fixed phases (index math, pipeline fill, a 10 k-cycle epilogue for 6 k cycles of k loop) and stages the activation tile
// once per 128-column tile: the P3 lateral ran at 2.3 TB/s of algorithmic traffic, the layer is HBM-bound by a wide margin
// (1.8 KB moved per pixel for 98 k MACs).  Here
//   * the FILTER BANK lives in REGISTERS: wave w of an 8-wave workgroup owns output channels 32 w .. 32 w + 31 of a 256-channel
//     column group and keeps their fragments (2 column tiles x Cin / 32 blocks x (hi, lo) = 32 or 64 VGPRs) for the whole launch;
//   * the ACTIVATIONS stream: a workgroup walks 32 KB tiles (128 pixels at Cin = 64, 64 at Cin = 128; persistent, two workgroups
//     per CU), each tile one linear LDS-DMA copy into one of two stages, one tile ahead (64 KB in flight per CU); every wave
//     reads every pixel's
//     fragments from LDS (conflict-free through a per-pixel XOR of the 16-byte chunk position, applied in the DMA's source
//     address) and multiplies them with its own filter fragments;
//   * the epilogue works from registers like the big kernel's (swapped MFMA operands + v_permlane16_swap: eight consecutive
//     channels per lane): bias, residual (same shape or the FPN's nearest-neighbour top-down read, fp32 or S32), ReLU, fp32 or
//     S32 stores of whole 128-byte runs.
// k order and term order are the implicit-GEMM kernel's (channel block ascending; lo*hi, hi*lo, hi*hi into one fp32
// accumulator) and so is the epilogue arithmetic: results are BIT-IDENTICAL to it (tests/test_conv_gpu.py).
#include "hn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTileBytes = 32 * 1024;   // bytes per streamed tile: 128 pixels at Cin = 64, 64 pixels at Cin = 128
constexpr int kStages = 2;
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

struct StreamParams {
  const _Float16* x;   // S32 dense [M][Cin/32][2][32]
  const _Float16* w;   // [Cout][Cin/32][2][32]
  const float* bias;
  const void* res;     // fp32 or S32, or null
  void* y;             // fp32 or S32
  int M, Cout, relu, out_split, res_mode, res_split, res_h, res_w, OH, OW;
  int ys, rs;          // output / residual pixel strides (floats for fp32, halfs for S32)
  int tiles;           // ceil(M / pixels per tile)
  unsigned x_records;
  unsigned mg_ohow, sh_ohow, mg_ow, sh_ow, mg_oh, sh_oh;   // exact divisions of the top-down read (res_mode 2), see magic_u31
  int* range_flag;
};

// floor(n / d) for 0 <= n < 2^31 with the host's magic pair (conv_igemm_f16x3.hip: magic_u31); d == 1 is encoded as mg == 0
__device__ __forceinline__ int fastdiv(int n, unsigned mg, unsigned sh) { return mg ? (int)(__umulhi((unsigned)n, mg) >> sh) : n; }

// Two 8-wave workgroups per CU (the second launch bound: 128 registers per lane): one workgroup's waits -- the tile's DMA and,
// because stores share vmcnt with it, the acknowledgement of the previous tile's stores -- are covered by the other's work.
template <int CB>   // channel blocks of 32: Cin = 32 * CB
__global__ __launch_bounds__(kThreads, 4) void conv1x1_stream_kernel(const StreamParams p) {
  constexpr int kTilePix = kTileBytes / (CB * 128);
  constexpr int PIX_BYTES = CB * 128;                 // one pixel's S32 row
  constexpr int CHUNKS = CB * 8;                      // 16-byte chunks per pixel (power of two >= 16)
  constexpr int TILE_BYTES = kTileBytes;
  constexpr int PASSES = TILE_BYTES / (kThreads * 16);
  static_assert(CB == 2 || CB == 4, "Cin = 64 or 128");
  static_assert(TILE_BYTES % (kThreads * 16) == 0, "a tile is a whole number of DMA passes");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, px = lane & 15;
  const int n0 = blockIdx.y * 256 + wave * 32;        // this wave's 32 output channels

  // ---- the wave's filter fragments, resident for the whole launch: column tile j, channel block cb, plane (hi, lo) ----
  f16x8 bh[2][CB], bl[2][CB];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const _Float16* q = p.w + ((long)(n0 + j * 16 + px) * CB + cb) * 64 + lg * 8;
      bh[j][cb] = *reinterpret_cast<const f16x8*>(q);
      bl[j][cb] = *reinterpret_cast<const f16x8*>(q + 32);
    }
  const int nsub = (lg & 1) * 16 + (lg >> 1) * 8;     // channel offset of this lane after the row exchange (see conv_igemm_f16x3.hip)
  const int n = n0 + nsub;
  // the column group's bias row in LDS behind the stages (read per pixel group: 8 registers less than keeping it per lane)
  float* bias_lds = reinterpret_cast<float*>(smem + kStages * kTileBytes);
  if (tid < 256) bias_lds[tid] = p.bias ? p.bias[blockIdx.y * 256 + tid] : 0.f;   // visible after the loop's first barrier

  // ---- the activation stream: tile t = 64 consecutive pixels = TILE_BYTES contiguous bytes; piece (pass, tid) lands at the
  // wave-linear LDS position (pass * 512 + tid) * 16 and FETCHES chunk (position ^ (pixel & 15)) of its pixel ----
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_records, 0x00020000);
  // (pass ps adds 512 pieces = a whole number of 16-pixel groups, i.e. 8 KB of source and no change of the XOR: one register)
  static_assert((kThreads / CHUNKS) % 16 == 0, "a DMA pass covers whole 16-pixel groups");
  const unsigned src_off = (unsigned)((tid / CHUNKS) * PIX_BYTES + (((tid % CHUNKS) ^ ((tid / CHUNKS) & 15)) << 4));
  const int wave_base = __builtin_amdgcn_readfirstlane(wave) * 64 * 16;
  auto dma_tile = [&](int t, int stage) {
    // the part of the ragged last tile past the tensor's end is zero-filled by the descriptor's range check
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * TILE_BYTES + ps * kThreads * 16 + wave_base), 16,
                                               (int)src_off, t * TILE_BYTES + ps * kThreads * 16, 0, 0);
  };
  // fragment read offset of this lane inside a 16-pixel group of a stage: pixel px, chunk (cb * 8 + pl * 4 + lg) ^ px
  auto frag = [&](int stage, int pt, int cb, int pl) {
    const int pixel = pt * 16 + px;
    return *reinterpret_cast<const f16x8*>(smem + stage * TILE_BYTES + pixel * PIX_BYTES + (((cb * 8 + pl * 4 + lg) ^ px) << 4));
  };

  const int ohow = p.OH * p.OW;
  const int t0 = blockIdx.x, dt = gridDim.x;
  // Two stages: the tile after this one streams in while this one is multiplied.  (One wait for EVERYTHING at the top of a tile:
  // the epilogue's stores share vmcnt with the DMA, and the counter gives no order between the two kinds -- deeper rings would
  // be drained by it anyway.  The layer moves 1.8 KB per pixel for 24 MFMAs per wave: the waits are idle HBM time only if
  // fewer bytes are in flight than the memory system needs, and 64 KB per CU is enough.)
  dma_tile(t0, 0);
  int stage = 0;
  for (int t = t0; t < p.tiles; t += dt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // tile t has landed for every wave, and nobody reads the other stage (tile t - dt) any more
    if (t + dt < p.tiles) dma_tile(t + dt, stage ^ 1);   // (wave-uniform; no look-ahead past the tensor: nothing rests on the
    //                                                       range check of an offset beyond num_records)
#pragma unroll
    for (int pt = 0; pt < kTilePix / 16; ++pt) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      // term order of conv_igemm_f16x3_kernel: lo*hi, hi*lo, hi*hi; W fragment = srcA (lane = pixel, registers = channels).
      // Two channel blocks at a time (the scheduling barrier keeps the compiler from hoisting all of a pixel group's fragment
      // reads: 16 live fragment registers instead of 32, which is what fits two workgroups per CU at Cin = 128)
#pragma unroll
      for (int c2 = 0; c2 < CB; c2 += 2) {
        f16x8 ah[2], al[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          ah[c] = frag(stage, pt, c2 + c, 0);
          al[c] = frag(stage, pt, c2 + c, 1);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int cb = c2 + c;
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[0][cb], al[c], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[1][cb], al[c], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[0][cb], ah[c], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[1][cb], ah[c], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[0][cb], ah[c], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[1][cb], ah[c], acc1, 0, 0, 0);
        }
        if (c2 + 2 < CB) __builtin_amdgcn_sched_barrier(0);
      }
      // rows {1, 3} of acc0 <-> rows {0, 2} of acc1 (16-lane rows): the lane then holds channels n .. n + 7 of pixel px
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc0[r]), __float_as_uint(acc1[r]), false, false);
        acc0[r] = __uint_as_float(s[0]);
        acc1[r] = __uint_as_float(s[1]);
      }
      const int m = t * kTilePix + pt * 16 + px;
      if (m >= p.M) continue;
      float v[8];
      const f32x4 bias0 = *reinterpret_cast<const f32x4*>(bias_lds + wave * 32 + nsub);
      const f32x4 bias1 = *reinterpret_cast<const f32x4*>(bias_lds + wave * 32 + nsub + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // + 0 when there is no bias (exact)
        v[e] = acc0[e] + bias0[e];
        v[4 + e] = acc1[e] + bias1[e];
      }
      if (p.res_mode) {
        long rpix = m;
        if (p.res_mode == 2) {
          // (the implicit-GEMM epilogue's integer divisions, with host-computed magic numbers: the same quotients)
          const int img = fastdiv(m, p.mg_ohow, p.sh_ohow);
          const int rem = m - img * ohow;
          const int oh = fastdiv(rem, p.mg_ow, p.sh_ow), ow = rem - oh * p.OW;
          const int sh_ = fastdiv(oh * p.res_h, p.mg_oh, p.sh_oh), sw_ = fastdiv(ow * p.res_w, p.mg_ow, p.sh_ow);
          rpix = ((long)img * p.res_h + sh_) * p.res_w + sw_;
        }
        if (p.res_split) {
          const _Float16* q16 = reinterpret_cast<const _Float16*>(p.res) + rpix * p.rs + (n >> 5) * 64 + (n & 31);
          const f16x8 rh = *reinterpret_cast<const f16x8*>(q16), rl = *reinterpret_cast<const f16x8*>(q16 + 32);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rh[e] + (float)rl[e];
        } else {
          const float* q32 = reinterpret_cast<const float*>(p.res) + rpix * p.rs + n;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(q32), r1 = *reinterpret_cast<const f32x4*>(q32 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += r0[e];
            v[4 + e] += r1[e];
          }
        }
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = hn::relu(v[e]);
      }
      if (p.out_split) {
        if (p.range_flag) hn::range_note_n<8>(p.range_flag, v);
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
        float* q32 = reinterpret_cast<float*>(p.y) + (long)m * p.ys + n;
        *reinterpret_cast<f32x4*>(q32) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(q32 + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    }
    stage ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead tiles past the end must land before the LDS is released
}

}  // namespace

namespace hn {

static void magic_u31(unsigned d, unsigned& mg, unsigned& sh) {   // as in conv_igemm_f16x3.hip
  if (d <= 1) { mg = 0; sh = 0; return; }
  unsigned k = 0;
  while ((1ull << k) < d) ++k;
  mg = (unsigned)(((1ull << (31 + k)) + d - 1) / d);
  sh = k - 1;
}

// Shape predicate (host): 1x1 / stride 1 / pad 0, Cin 64 or 128, Cout a multiple of 256, dense S32 input, enough 32 KB tiles
// to give every CU several (below that the implicit-GEMM tiles fill the chip better), the vector epilogue's alignment.
bool conv1x1_stream_applies(const hn_conv_desc* d, bool has_gn, bool has_group) {
  const int64_t m = (int64_t)d->n * d->oh * d->ow;
  return !env_flags().no_stream && !has_gn && !has_group && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->dil == 1 &&
         (d->cin == 64 || d->cin == 128) && d->cout % 256 == 0 && d->terms != 1 && d->tile == HN_TILE_AUTO &&
         (d->in_pix_stride == 0 || d->in_pix_stride == 2 * d->cin) && (d->relu_cols == 0 || d->relu_cols >= d->cout) &&
         m >= (int64_t)65536 && (int64_t)d->oh * d->res_h < ((int64_t)1 << 31) && (int64_t)d->ow * d->res_w < ((int64_t)1 << 31) && m * d->cin * 4 < ((int64_t)1 << 31) - ((int64_t)64 << 20);
}

bool conv1x1_stream_operands_ok(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                                const void* y) {
  const int ys = d->out_pix_stride ? d->out_pix_stride : (d->out_split ? 2 : 1) * d->cout;
  const int rs = d->res_pix_stride ? d->res_pix_stride : (d->res_split ? 2 : 1) * d->cout;
  return (uintptr_t)x16 % 16 == 0 && (uintptr_t)w16 % 16 == 0 && (uintptr_t)y % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
         (!residual || (uintptr_t)residual % 16 == 0) && (d->out_split ? ys % 64 == 0 : ys % 4 == 0) &&
         (d->res_mode == 0 || (d->res_split ? rs % 64 == 0 : rs % 4 == 0));
}

int conv1x1_stream(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                   hipStream_t st) {
  StreamParams p;
  p.x = (const _Float16*)x16; p.w = (const _Float16*)w16; p.bias = bias; p.res = d->res_mode ? residual : nullptr; p.y = y;
  p.M = d->n * d->oh * d->ow; p.Cout = d->cout; p.relu = d->relu_cols > 0; p.out_split = d->out_split;
  p.res_mode = d->res_mode; p.res_split = d->res_split; p.res_h = d->res_h; p.res_w = d->res_w; p.OH = d->oh; p.OW = d->ow;
  p.ys = d->out_pix_stride ? d->out_pix_stride : (d->out_split ? 2 : 1) * d->cout;
  p.rs = d->res_pix_stride ? d->res_pix_stride : (d->res_split ? 2 : 1) * d->cout;
  const int tile_pix = kTileBytes / (d->cin * 4);
  p.tiles = cdiv(p.M, tile_pix);
  magic_u31((unsigned)(d->oh * d->ow), p.mg_ohow, p.sh_ohow);
  magic_u31((unsigned)d->ow, p.mg_ow, p.sh_ow);
  magic_u31((unsigned)d->oh, p.mg_oh, p.sh_oh);
  p.x_records = (unsigned)((int64_t)p.M * d->cin * 4);
  p.range_flag = range_flag_ptr();
  const int cb = d->cin / 32;
  const int lds = kStages * kTileBytes + 1024;   // + the bias row
  int dev = 0, cus = 256;
  HN_CHECK_HIP(hipGetDevice(&dev));
  static int cu_count[64] = {};
  if (dev >= 0 && dev < 64) {
    if (!cu_count[dev]) {
      hipDeviceProp_t prop;
      HN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
      cu_count[dev] = prop.multiProcessorCount;
    }
    cus = cu_count[dev];
  }
  const int groups = d->cout / 256;
  int gx = cus * 2 / groups;   // resident workgroups of the chip (two per CU), over all column groups
  gx = gx < 1 ? 1 : (gx > p.tiles ? p.tiles : gx);
  static bool attr_set[2][64] = {};
  auto set_attr = [&](const void* fn, int which) -> int {
    if (dev < 0 || dev >= 64 || !attr_set[which][dev]) {
      HN_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      if (dev >= 0 && dev < 64) attr_set[which][dev] = true;
    }
    return HN_OK;
  };
  if (cb == 2) {
    const int s = set_attr((const void*)conv1x1_stream_kernel<2>, 0);
    if (s != HN_OK) return s;
    hipLaunchKernelGGL(conv1x1_stream_kernel<2>, dim3(gx, groups), dim3(kThreads), lds, st, p);
  } else {
    const int s = set_attr((const void*)conv1x1_stream_kernel<4>, 1);
    if (s != HN_OK) return s;
    hipLaunchKernelGGL(conv1x1_stream_kernel<4>, dim3(gx, groups), dim3(kThreads), lds, st, p);
  }
  HN_CHECK_LAUNCH("conv1x1_stream_kernel");
  return HN_OK;
}

}  // namespace hn
