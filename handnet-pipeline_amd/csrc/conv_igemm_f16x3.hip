// Implicit-GEMM convolution on the gfx950 f16 MFMA with SPLIT operands ("f16x3").
//
// Every fp32 value v is represented as hi + lo with hi = fp16(v), lo = fp16(v - hi)
// (22 significant bits) and each product a*b is evaluated as
//      a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (a_lo*b_lo ~ 2^-22 |ab| is dropped)
// by three v_mfma_f32_16x16x32_f16 accumulating in ONE fp32 accumulator.  fp16 x fp16
// products are exact in fp32, so the result carries fp32-grade error (A2J keypoints move by
// <= 2e-4 vs the fp32 reference, bound 1e-3) at 16/3 = 5.3x the f32-MFMA rate; plain fp16 /
// bf16 operands miss the bound by two orders.
//
// Data layout ("S32" split tensors, same bytes as fp32):
//   activations  fp16 [N][H][W][C/32][2][32]  -- per pixel and 32-channel block a 128-byte run:
//                hi[32] | lo[32].  Written ONCE by the producer (conv epilogue, max-pool,
//                GroupNorm-apply pass), so consumers never convert.
//   weights      fp16 [Cout][(Cin/32)*R*S][2][32] -- the same run structure along k; k tiles are
//                ordered channel block OUTER, tap (r,s) INNER (weights.split_f16x3).
//
// Kernel (v5).  rocprof ablations of v2 (fp32 activations split in the loader; git history,
// commit 4caba32) showed the loader's conversion VALU (+27 %) and
// its ds_write_b128 traffic (+15 %) to be the largest costs, so since v3 there is neither:
//   * both operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR
//     staging, no VALU, no ds_write.  One wave instruction moves 8 rows x 128 B; the gather
//     (im2col row, zero page for padding taps) is expressed in the per-lane SOURCE address,
//     and so is the bank swizzle (the DMA destination is wave-linear): chunk cc of row r
//     lands at position cc ^ ((r >> 1) & 7), which makes every ds_read_b128 conflict-free;
//   * NBUF LDS stages (2 for the 128x128 tile, 3-4 for the small tiles whose steps are shorter
//     than the memory latency), DMA issued NBUF-1 steps ahead and retired with COUNTED vmcnt
//     waits, ONE barrier per 32-deep k tile placed MID-step so that MFMAs sit on both sides;
//   * MFMA shape 16x16x32 (v4): a register-only probe (tools/probes/mfma_peak*.hip) sustains
//     1.65-1.9 PFLOP/s with it on this chip against 1.2-1.45 PFLOP/s for 32x32x16; one
//     ds_read_b128 covers a 16-row tile's whole 32-deep k run.  A step is split by column halves;
//   * v5: the k loop is ONE basic block whose MFMAs and LDS reads are volatile asm in a fixed,
//     hand-interleaved order (HalfSched) with hand-counted lgkmcnt waits, a single set of A
//     fragments refilled in place, and a saturating prefetch instead of tail branches.  The
//     compiler-scheduled v4 loop serialised DMA issue -> fragment reads -> lgkmcnt(0) in front of
//     the second MFMA half and copied 32 fragment registers per step; v5 is 8-10 % faster in
//     steady state (tools/probes/exp/ab.sh: 422 vs 388 TFLOP/s on the 100x136x256->256 layer).
// What bounds it now is the socket power cap, not issue slots: while this kernel loops rocm-smi
// shows 1400 W (the cap) and sclk 1.84 GHz instead of 2.4 (tools/probes/exp/clocks.sh), so the
// clock-adjusted dense-f16 peak is ~1.9 PFLOP/s; ablations (no DMA: +22 %, no LDS reads: +23 %)
// show data movement energy, not MFMA issue, is what is left.
//   * v6: both operands through buffer descriptors (per-lane 32-bit offsets, padding by range check), see ConvParams16;
//   * v7 ("RS"): 3x3 / stride 1 / pad 1 layers stage the A operand once per filter ROW as a wide tile with zero-filled gap
//     slots at the image-row ends and read the three taps from it at slot offsets (see the kernel template's comment).
//   * v8 (round 3): MFMA operands swapped (lane = pixel, registers = channels) + v_permlane16_swap: the epilogue works from
//     registers with 16-byte accesses, no LDS transposition, no barrier.
// Epilogue: bias, residual (fp32 or S32), ReLU on a column prefix, output fp32 or S32.
// Requires Cin % 32 == 0 (the 4-channel stems stay on the f32 kernel).
#include "hn_common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// 256 bytes of zeros: source of every out-of-image tap
__device__ __attribute__((aligned(256))) _Float16 g_zero_page16[128];

struct ConvParams16 {
  const _Float16* x;  // S32 activations
  const _Float16* w;  // S32 weights
  const float* bias;
  const void* res;    // fp32 or S32
  void* y;            // fp32 or S32
  int N, H, W, Cin, Cout, R, S, stride, pad, dil, OH, OW;
  int M, Ktot, ktiles;
  int relu_cols, res_mode, res_h, res_w;
  int xs;             // input pixel stride in halfs
  int pitch;          // input row pitch in pixels (W for dense tensors; the bordered stem image is wider)
  long lo_off;        // halfs from a row's 64-byte hi run to its lo run (32 in S32; plane distance for the stem image)
  int ys;             // output pixel stride (floats for fp32 output, halfs for S32)
  int rs;             // residual pixel stride (same convention)
  int out_split, res_split;
  int vec_epi;        // 1: 16-byte epilogue through LDS (Cout % 8 == 0 and aligned strides)
  float* gn_partial;  // optional GroupNorm partial sums [ceil(M/32)][Cout/8][4] (see hn_conv2d_nhwc_f16x3_gn)
  // split-K (small-M layers): gridDim.y workgroups share an output tile, each sums kt_per k tiles into
  // split_ws[z][M][Cout] (fp32, no epilogue); splitk_reduce_kernel adds them in z order and finishes
  // grouped launch (hn_conv2d_nhwc_f16x3_grouped): gridDim.z same-shape problems with their own tensors
  int groups;
  const _Float16* gx[HN_CONV_MAX_GROUP];
  const _Float16* gw[HN_CONV_MAX_GROUP];
  const float* gbias[HN_CONV_MAX_GROUP];
  void* gy[HN_CONV_MAX_GROUP];
  float* ggn[HN_CONV_MAX_GROUP];
  int gH[HN_CONV_MAX_GROUP], gW[HN_CONV_MAX_GROUP], gOH[HN_CONV_MAX_GROUP], gOW[HN_CONV_MAX_GROUP];
  int gM[HN_CONV_MAX_GROUP], gnblocks[HN_CONV_MAX_GROUP];   // members may differ in spatial size (FPN levels)
  int gn_units;       // 8-channel units per row group in the GroupNorm slab (Cout/8 unless members share a slab)
  int splits, kt_per, splitk_mode;
  float* split_ws;
  int64_t split_ws_bytes;
  int tiles_m, tiles_n, nblocks;
  int small_mask;     // mixed grouped launch: bit g set = member g runs the 64-row per-tap form (see conv_igemm_f16x3_mixed_kernel)
  int* range_flag;    // f16x3 range contract (hn_range_check_enable): set to 1 when an S32 output value cannot be split
  // v6 operand addressing (BUF kernels): both operands are fetched through buffer descriptors, so a DMA's address is
  // <descriptor base> + <per-lane 32-bit offset, loop-invariant> + <wave-uniform SGPR offset of the k tile>, and a
  // padding tap is a lane whose offset has bit 31 set: the hardware range check returns zeros for it.
  // exact division of a row index m < 2^31 by OH*OW and by OW with one v_mul_hi (host-computed magic numbers): the
  // per-lane pixel decomposition in the prologue cost ~25 VALU per division, 2 divisions per DMA piece
  unsigned mg_ohow, sh_ohow, mg_ow, sh_ow;
  unsigned gmg_ohow[HN_CONV_MAX_GROUP], gsh_ohow[HN_CONV_MAX_GROUP], gmg_ow[HN_CONV_MAX_GROUP], gsh_ow[HN_CONV_MAX_GROUP];
  // row-shared A operand (RS kernels, 3x3 / stride 1 / pad 1): exact division by W + 1 and by H for the slot -> pixel map
  unsigned mg_w1, sh_w1, mg_h, sh_h;
  unsigned gmg_w1[HN_CONV_MAX_GROUP], gsh_w1[HN_CONV_MAX_GROUP], gmg_h[HN_CONV_MAX_GROUP], gsh_h[HN_CONV_MAX_GROUP];
  int terms;          // host: 3 (default) or 1 (hn_conv_desc.terms: the hi*hi-only throughput mode)
  int rs_ok;          // host: the row-shared A kernel may be used (set by conv16_run, refined in launch16)
  // fused 3x3 / stride-2 / pad-1 max pooling (POOL kernel, the FCOS stem): a workgroup computes a 15 x 17 patch of conv
  // pixels = 7 x 8 pooled pixels; pool_ty x pool_tx patches per image, pooled map pool_oh x pool_ow
  int pool_ty, pool_tx, pool_oh, pool_ow;
  unsigned mg_pt, sh_pt, mg_ptx, sh_ptx;   // exact division of the patch index by pool_ty * pool_tx and by pool_tx
  unsigned a_records; // bytes covered by the A descriptor (< 2^31 so that bit 31 is out of range)
  unsigned b_records;
  unsigned ga_records[HN_CONV_MAX_GROUP];
};

#define HN_TRY16(expr)          \
  do {                          \
    const int st_ = (expr);     \
    if (st_ != HN_OK) return st_; \
  } while (0)

constexpr int BK = 32;    // k values per tile
constexpr int ROWH = 64;  // halfs per LDS row (hi 32 | lo 32) = 128 bytes

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// floor(n / d) for 0 <= n < 2^31 with the host's magic pair (see magic_u31): d == 1 is encoded as mg == 0
__device__ __forceinline__ int fastdiv(int n, unsigned mg, unsigned sh) {
  return mg ? (int)(__umulhi((unsigned)n, mg) >> sh) : n;
}

// Issue plan of the second half of a k step.  MFMA k (term-major: term = k / (TM*TH), row tile i, column
// tile jj) is preceded by the memory instructions whose slot is k.  Instruction list q: the DPT LDS-DMA pieces of
// tile t+NBUF, the 2*TH half-0 W fragment reads of tile t+1, then the A fragments of tile t+1 -- lo[i] may
// be overwritten once term 0 (the only user of lo) is through row tile i, hi[i] once term 2 is.
// One MFMA whose place in the instruction stream is fixed: accumulator tied in an AGPR quad, and (volatile +
// memory clobber) neither other pinned MFMAs nor LDS reads / LDS-DMA move across it.  The builtin form let
// the scheduler hoist fragment reads over the loop back-edge or sink MFMAs past the barrier, and the
// allocator then rotated accumulators through copies (v_accvgpr_mov) in the hot loop.
// Operand order (v8): the W fragment `b` is srcA and the activation fragment `a` is srcB (both fragments have the same
// register layout: lane (r, g) holds row r, k = 8g..8g+7), so the accumulator holds D[channel 4*(lane>>4)+reg][pixel
// lane&15] -- four consecutive CHANNELS of one pixel per lane, which is what lets the epilogue store 16-byte runs
// without a transposition through LDS.  Same products, same k order: results are bit-identical to the a-b order.
__device__ __forceinline__ void mfma_pinned(f32x4& c, const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %1, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
}

// LDS read with a fixed place in the instruction stream; the compiler neither sees that it is asynchronous
// nor inserts waits for it -- the consumer waits with lgkm_wait<N>() (LDS reads return in issue order).
template <int OFF>
__device__ __forceinline__ void lds_read_pinned(f16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const _Float16* p) { return (unsigned)(size_t)(lds_void*)p; }

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// residual add, ReLU on a column prefix and the 16-byte store of 8 consecutive channels n..n+7 of output
// pixel m (fp32 or S32); shared by the conv epilogue and the split-K reduction
__device__ __forceinline__ void epi_finish8(const ConvParams16& p, int m, int n, float (&v)[8], int ohow,
                                            const f16x8* pre_hi = nullptr, const f16x8* pre_lo = nullptr) {
  if (pre_hi) {  // S32 residual of the same shape, fetched by the caller ahead of time
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)(*pre_hi)[e] + (float)(*pre_lo)[e];
  } else if (p.res_mode) {
    long rpix = m;
    if (p.res_mode == 2) {
      const int img = m / ohow;
      const int rem = m - img * ohow;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      const int sh_ = (int)(((long)oh * p.res_h) / p.OH), sw_ = (int)(((long)ow * p.res_w) / p.OW);
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
  if (p.relu_cols >= p.Cout) {  // the usual case (all columns): wave-uniform, 8 v_max
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = hn::relu(v[e]);
  } else if (p.relu_cols > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (n + e < p.relu_cols) v[e] = hn::relu(v[e]);
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

// z-ordered sum of the split-K partial tiles + bias + the common epilogue tail; one thread = 8 channels of a pixel
__device__ __forceinline__ void splitk_reduce_body(const ConvParams16& p) {
  const int units = p.Cout >> 3;
  const long total = (long)p.M * units;
  const int ohow = p.OH * p.OW;
  const long plane = (long)p.M * p.Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / units), n = (int)(i - (long)m * units) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = p.split_ws + (long)m * p.Cout + n;
    for (int z = 0; z < p.splits; ++z) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + z * plane), b = *reinterpret_cast<const f32x4*>(src + z * plane + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += a[e];
        v[4 + e] += b[e];
      }
    }
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += b0[e];
        v[4 + e] += b1[e];
      }
    }
    epi_finish8(p, m, n, v, ohow);
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams16 p) { splitk_reduce_body(p); }

// TERMS == 1 (f16x1): the lo fragments are never read -- NOT "read and ignored": an asynchronous LDS read into a register the
// compiler considers dead lands in whatever that register holds by then (seen as nondeterministic garbage) -- so the list has
// TH + TM reads; the MFMA slot space stays 3 * TM * TH with the slots of terms 0 / 1 empty.
template <int TM, int TH, int DPT, int TERMS = 3>
struct HalfSched {
  static constexpr int NM = 3 * TM * TH;
  static constexpr int NB = (TERMS == 3 ? 2 : 1) * TH;    // W fragment reads of a column half
  static constexpr int NAL = TERMS == 3 ? TM : 0;         // A lo fragment reads
  static constexpr int NMEM = DPT + NB + NAL + TM;
  static constexpr int earliest(int q) {
    return q < DPT + NB ? 0
           : q < DPT + NB + NAL ? (q - DPT - NB + 1) * TH
                                : 2 * TM * TH + (q - DPT - NB - NAL + 1) * TH;
  }
  static constexpr int slot(int q) {
    const int spread = (q * NM) / NMEM;
    return earliest(q) > spread ? earliest(q) : spread;
  }
};

// RS ("row-shared A", 3x3 / stride 1 / pad 1 / dilation 1 only): the three taps of a filter ROW read the same pixels
// shifted by one, so the A operand of a (channel block, filter row) is staged ONCE as a wide tile and the k steps s = 0, 1, 2
// read their fragments from it at slot offsets -1 / 0 / +1 -- a third of the L2 -> LDS traffic of the A operand, which is what
// holds the clock down at the power cap (tools/probes/exp/halo2.sh: +17 % on the tower layer with that traffic removed).
// Slot j of the wide tile holds the pixel with PADDED linear index u0 + j, where a padded image row has W + 1 entries and
// entry W is a gap that the DMA zero-fills (descriptor range check): the left neighbour of a pixel with ow = 0 and the right
// neighbour of one with ow = W - 1 are then the gap, with no per-tap masking of fragments.  Vertical padding and the slots
// past the tensor are per-lane invalid bits of the DMA piece, one per filter row.
// Two waves per SIMD (two 4-wave workgroups or one 8-wave workgroup per CU) are part of the design: one workgroup's
// prologue / epilogue runs under the other's MFMAs.  The second launch bound makes the register allocator keep to the
// 256 registers per lane that allows (the v8 epilogue once came out at 194 + 64 = 260 on the 128x128 tile: one workgroup
// per CU, -40 % on every short-k layer of that tile).  The 4-wave 256x128 sweep variant needs 128 accumulators: one wave.
// POOL (the FCOS stem, conv -> ReLU -> 3x3 / stride-2 / pad-1 max pooling fused; 256x64 tile only): the 256 rows of a tile
// are a 15 x 17 PATCH of conv pixels (row-major, row 255 unused) instead of 256 consecutive ones -- the conv pixels that 7 x 8
// pooled pixels need, one halo row / column shared with the neighbouring patches (1.16x the convolution work of the
// unfused form) -- and the epilogue pools the patch through LDS and stores only the pooled S32 tensor: the 64-channel
// conv map at half resolution (1.8 GB at batch 32) is neither written nor read back.
constexpr int kPoolRows = 15, kPoolCols = 17, kPoolPR = 7, kPoolPC = 8;
// TERMS = 3: the split-precision product (lo*hi + hi*lo + hi*hi, fp32-grade).  TERMS = 1 ("f16x1", the THROUGHPUT mode SURVEY D6
// plans beside the parity mode; never the default): only hi*hi is issued -- one MFMA per MAC on plain fp16 operands, identical
// data movement -- so that "what does the 1e-3 contract cost" has a measured answer (bench.py --precision f16x1).
// The kernel's body as a device function of (parameter block, workgroup coordinates): conv_igemm_f16x3_kernel passes its own
// kernel argument and blockIdx; conv_igemm_f16x3_multi_kernel (heterogeneous launches, below) the member's block and the
// member-local coordinates.  Always inlined: the single-problem kernel compiles to what it was.
template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, bool RS = false, bool POOL = false, int TERMS = 3, bool DYN = false>
__device__ __forceinline__ void conv_igemm_f16x3_body(const ConvParams16& p, const int blk_x, const int blk_y, const int blk_z) {
  static_assert(TERMS == 3 || TERMS == 1, "three terms (fp32-grade) or the hi*hi term alone");
  static_assert(NBUF >= 2 && NBUF <= 6, "2..6 LDS stages");
  static_assert(!POOL || (BUF && !RS && BM == 256 && BN == 64 && WN == 1), "the pooling epilogue is written for the 256x64 tile");
  static_assert(!RS || (BUF && NBUF == 2), "row-shared A needs the descriptor form and the 2-stage pipeline");
  // Grouped launch: workgroup z works on member z -- its own tensors and, for FPN levels, its own spatial size.
  // Only these fields differ per member; they live in a small local struct `o` (picked with constant-index
  // selects: a dynamic index into the kernel-argument arrays would send the whole parameter block through
  // scratch memory, -40 % on every convolution; copying the whole block and patching it spills 480 SGPRs).
  // Workgroups beyond a smaller member's tile count leave at once.
  struct {
    const _Float16 *x, *w;
    const float* bias;
    void* y;
    float* gn_partial;
    int H, W, pitch, OH, OW, M, nblocks;
    unsigned a_records, mg_ohow, sh_ohow, mg_ow, sh_ow, mg_w1, sh_w1, mg_h, sh_h;
  } o = {p.x, p.w, p.bias, p.y, p.gn_partial, p.H, p.W, p.pitch, p.OH, p.OW, p.M, p.nblocks, p.a_records,
         p.mg_ohow, p.sh_ohow, p.mg_ow, p.sh_ow, p.mg_w1, p.sh_w1, p.mg_h, p.sh_h};
  if (p.groups > 1) {
    // The member's fields are read straight from the kernel-argument SEGMENT (constant address space) with the uniform
    // index blockIdx.z: scalar loads with an SGPR offset.  (Indexing the by-value parameter `p` dynamically would copy
    // the whole block to scratch; constant-index select chains over all six members -- the round-1 form -- kept ~500
    // bytes of member tables live in SGPRs: 376-528 spilled SGPRs, i.e. ~630 v_writelane / v_readlane per workgroup
    // in the prologue of EVERY convolution, grouped or not.)
    typedef __attribute__((address_space(4))) const ConvParams16 KArgs;
    KArgs* kp = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int gz = blk_z;
    static_assert(HN_CONV_MAX_GROUP == 6, "member tables hold six entries");
    o.nblocks = kp->gnblocks[gz];
    if (blk_x >= o.nblocks) return;
    o.x = kp->gx[gz];
    o.w = kp->gw[gz];
    o.bias = kp->gbias[gz];
    o.y = kp->gy[gz];
    o.gn_partial = kp->ggn[gz];
    o.H = kp->gH[gz];
    o.W = kp->gW[gz];
    o.pitch = o.W;
    o.OH = kp->gOH[gz];
    o.OW = kp->gOW[gz];
    o.M = kp->gM[gz];
    o.a_records = kp->ga_records[gz];
    o.mg_ohow = kp->gmg_ohow[gz];
    o.sh_ohow = kp->gsh_ohow[gz];
    o.mg_ow = kp->gmg_ow[gz];
    o.sh_ow = kp->gsh_ow[gz];
    if constexpr (RS) {
      o.mg_w1 = kp->gmg_w1[gz];
      o.sh_w1 = kp->gsh_w1[gz];
      o.mg_h = kp->gmg_h[gz];
      o.sh_h = kp->gsh_h[gz];
    }
  }
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;  // 16x16 MFMA tiles per wave
  static_assert(TM >= 2 && TM % 2 == 0 && TN >= 2 && TN % 2 == 0, "wave tile must be a multiple of 32x32");
  constexpr int TH = TN / 2;                            // column tiles per half step
  constexpr int ROWS_PASS = NT / 8;  // 8 lanes (16 B each) cover one 128-byte row
  static_assert(BM % ROWS_PASS == 0 && BN % ROWS_PASS == 0, "tile rows must be a multiple of NT/8");
  constexpr int A_ROWS = RS ? BM + ROWS_PASS : BM;  // RS: BM + 2 neighbours + up to ROWS_PASS - 2 gap slots
  constexpr int A_IT = A_ROWS / ROWS_PASS, B_IT = BN / ROWS_PASS;
  constexpr int A_BUF = A_ROWS * ROWH, B_BUF = BN * ROWH;  // halfs per buffer
  // the RS stages exceed the 64 KB a static array may have: dynamic LDS there (launch16_impl sets the size)
  extern __shared__ __attribute__((aligned(1024))) _Float16 smem_dyn[];
  // (DYN: a per-tap form that shares its kernel -- and the dynamic LDS block -- with a row-shared form: the mixed grouped kernel)
  __shared__ __attribute__((aligned(1024))) _Float16 smem_static[(RS || DYN) ? 8 : NBUF * (A_BUF + B_BUF)];
  _Float16* smem = (RS || DYN) ? smem_dyn : smem_static;
  _Float16* As = smem;                 // [stage][A_ROWS][64]
  _Float16* Bs = smem + NBUF * A_BUF;  // [stage][BN][64]

  int lid;
  {
    const int bid = blk_x, nb = o.nblocks;
    const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // POOL: patch (image, patch row, patch column); conv pixel of tile row `row` = (oy0 + row / 17, ox0 + row % 17)
  int pl_img = 0, pl_ty = 0, pl_tx = 0, pl_oy0 = 0, pl_ox0 = 0;
  if constexpr (POOL) {
    pl_img = fastdiv(lid, p.mg_pt, p.sh_pt);
    const int rem = lid - pl_img * (p.pool_ty * p.pool_tx);
    pl_ty = fastdiv(rem, p.mg_ptx, p.sh_ptx);
    pl_tx = rem - pl_ty * p.pool_tx;
    pl_oy0 = 2 * kPoolPR * pl_ty - 1;
    pl_ox0 = 2 * kPoolPC * pl_tx - 1;
  }
  const int wm = wave / WN, wn = wave - wm * WN;
  const int ohow = o.OH * o.OW;


  // ---- DMA geometry: lane -> (row = tid >> 3 within a pass, LDS position pos = tid & 7) ----
  const int drow = tid >> 3, dpos = tid & 7;
  // !BUF (fallback for tensors of 2 GB and more): 64-bit per-lane pointers, bounds checks and a zero page per tap
  int a_ih0[A_IT], a_iw0[A_IT], a_cc[A_IT];  // a_cc: chunk offset inside the zero page
  const _Float16* a_row[A_IT];  // address of (img, ih0, iw0, channel 0) + swizzled chunk; may lie outside the image
  const _Float16* b_ptr[B_IT];
  // BUF: loop-invariant 32-bit byte offsets from the descriptor bases + one bit per filter tap that is SET when the tap
  // falls outside the image for this lane's pixel (tap index = r * S + s <= 31)
  unsigned a_off[A_IT], a_inv[A_IT], b_off[B_IT];
  // RS: padded linear index of the slot in front of the tile's first pixel (may be -1)
  int rs_u0 = 0;
  if constexpr (RS) {
    const int mw = fastdiv(m0, o.mg_ow, o.sh_ow);
    rs_u0 = mw * (o.W + 1) + (m0 - mw * o.W) - 1;
  }
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int row = drow + it * ROWS_PASS;
    if constexpr (RS) {
      const int u = rs_u0 + row;
      const int uc = u < 0 ? 0 : u;
      const int rowi = fastdiv(uc, o.mg_w1, o.sh_w1);        // global image-row index (img * H + oh)
      const int owp = uc - rowi * (o.W + 1);
      const int img = fastdiv(rowi, o.mg_h, o.sh_h);
      const int oh = rowi - img * o.H;
      const bool pix_ok = u >= 0 && owp < o.W && rowi < p.N * o.H;
      const int chunk = dpos ^ swz(row);
      a_off[it] = pix_ok ? (((unsigned)rowi * (unsigned)o.pitch + (unsigned)owp) * (unsigned)p.xs + (unsigned)((chunk & 3) * 8) +
                            (unsigned)(chunk >> 2) * (unsigned)p.lo_off) * 2u
                         : 0u;
      // bit r: filter row r reads image row oh + r - 1
      a_inv[it] = !pix_ok ? 7u : (oh == 0 ? 1u : 0u) | (oh == o.H - 1 ? 4u : 0u);
      continue;
    }
    if constexpr (POOL) {
      // conv pixels outside the map (the pooling's own padding ring, patches past the bottom / right edge, row 255) read a
      // clamped pixel: their results are zeroed in the epilogue (every pooling window holds a real pixel and ReLU >= 0)
      const int pr = row / kPoolCols, pc = row - pr * kPoolCols;
      int oy = pl_oy0 + pr, ox = pl_ox0 + pc;
      oy = oy < 0 ? 0 : (oy < o.OH ? oy : o.OH - 1);
      ox = ox < 0 ? 0 : (ox < o.OW ? ox : o.OW - 1);
      const int chunk = dpos ^ swz(row);
      a_off[it] = (((unsigned)(pl_img * o.H + oy * p.stride) * (unsigned)o.pitch + (unsigned)(ox * p.stride)) * (unsigned)p.xs +
                   (unsigned)((chunk & 3) * 8) + (unsigned)(chunk >> 2) * (unsigned)p.lo_off) * 2u;
      a_inv[it] = 0u;
      continue;
    }
    int m = m0 + row;
    m = m < o.M ? m : o.M - 1;  // rows >= M are never stored
    const int img = fastdiv(m, o.mg_ohow, o.sh_ohow);
    const int rem = m - img * ohow;
    const int oh = fastdiv(rem, o.mg_ow, o.sh_ow), ow = rem - oh * o.OW;
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    const int chunk = dpos ^ swz(row);  // source chunk that belongs at this LDS position: 0-3 hi run, 4-7 lo run
    if constexpr (BUF) {
      // every term stays below the descriptor's extent (< 2^31 bytes, checked on the host): 32-bit arithmetic
      a_off[it] = (((unsigned)(img * o.H + oh * p.stride) * (unsigned)o.pitch + (unsigned)(ow * p.stride)) * (unsigned)p.xs +
                   (unsigned)((chunk & 3) * 8) + (unsigned)(chunk >> 2) * (unsigned)p.lo_off) * 2u;
      // tap (r, s) is invalid iff its row or its column is: R + S tests instead of R * S.  col_bits has bit s set for
      // a bad column; a bad row sets the whole S-bit field of that row.  (No padding => nothing to test.)
      unsigned inv = 0;
      if (p.pad > 0) {
        if (p.R == 3 && p.S == 3) {  // wave-uniform: the usual filter, fully unrolled (no loop control per piece)
          const unsigned uw = (unsigned)o.W, uh = (unsigned)o.H;
          const unsigned col_bits = ((unsigned)iw0 >= uw ? 1u : 0u) | ((unsigned)(iw0 + p.dil) >= uw ? 2u : 0u) |
                                    ((unsigned)(iw0 + 2 * p.dil) >= uw ? 4u : 0u);
          inv = ((unsigned)ih0 >= uh ? 7u : col_bits) | (((unsigned)(ih0 + p.dil) >= uh ? 7u : col_bits) << 3) |
                (((unsigned)(ih0 + 2 * p.dil) >= uh ? 7u : col_bits) << 6);
        } else {
          unsigned col_bits = 0;
          for (int sx = 0; sx < p.S; ++sx) col_bits |= ((unsigned)(iw0 + sx * p.dil) >= (unsigned)o.W ? 1u : 0u) << sx;
          const unsigned row_full = (1u << p.S) - 1u;
          for (int r = 0; r < p.R; ++r) {
            const unsigned bits = (unsigned)(ih0 + r * p.dil) >= (unsigned)o.H ? row_full : col_bits;
            inv |= bits << (r * p.S);
          }
        }
      }
      a_inv[it] = inv;
    } else {
      a_ih0[it] = ih0;
      a_iw0[it] = iw0;
      a_cc[it] = chunk * 8;
      a_row[it] = o.x + (((long)img * o.H + ih0) * o.pitch + iw0) * p.xs + (chunk & 3) * 8 + (chunk >> 2) * p.lo_off;
    }
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int row = drow + it * ROWS_PASS;
    int n = n0 + row;
    n = n < p.Cout ? n : p.Cout - 1;  // columns >= Cout are never stored
    if constexpr (BUF)
      b_off[it] = ((unsigned)n * (unsigned)p.Ktot * 2u + (unsigned)((dpos ^ swz(row)) * 8)) * 2u;
    else
      b_ptr[it] = o.w + (long)n * p.Ktot * 2 + (dpos ^ swz(row)) * 8;
  }
  // buffer descriptors (wave-uniform by construction: kernel arguments / blockIdx.z selects)
  // the A descriptor starts pad rows + pad columns BEFORE the member's first pixel, so that a_off (computed from the
  // un-padded coordinates oh * stride, ow * stride) is never negative; the pitch is the member's own
  // (RS: one row up only -- the slot map takes care of the columns)
  const long a_shift = RS ? (long)o.pitch * p.xs * 2 : BUF ? ((long)p.pad * o.pitch + p.pad) * p.xs * 2 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(o.x) - a_shift), 0, (int)o.a_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)o.w, 0, (int)p.b_records, 0x00020000);
  // wave-uniform LDS row base of this wave's 8-row group inside a pass
  const int grp_row = __builtin_amdgcn_readfirstlane(wave) * 8;

  // K order: 32-channel block OUTER, filter taps INNER.  Consecutive k tiles then re-read almost
  // the same pixels (shifted by one tap), so the re-use distance across the ~64 workgroups of an
  // XCD is ~1 MB instead of ~8 MB and the 4 MB L2 serves it (tap-outer order re-fetched the
  // input 3-7x over the fabric: FETCH_SIZE, profiles/).  Per step the tap offset is wave-uniform
  // (SALU); per lane only the two bounds checks and one 64-bit add remain.
  // this workgroup's k tiles: [t_begin, t_end) (everything unless split-K)
  const int t_begin = p.splits > 1 ? blk_y * p.kt_per : 0;
  const int t_end = p.splits > 1 ? min(t_begin + p.kt_per, p.ktiles) : p.ktiles;
  int load_t = t_begin, cur_cb = t_begin / (p.R * p.S);
  int cur_r = (t_begin - cur_cb * p.R * p.S) / p.S, cur_s = t_begin - cur_cb * p.R * p.S - cur_r * p.S;

  // one DMA instruction (8 rows x 128 B per wave): A piece `it` gathers im2col rows, B piece `it` weight rows.
  // BUF: `uoff` / `boff` are the wave-uniform BYTE offsets of the k tile (SGPR soffset), `sh` = 31 - tap index; the
  // lane's own part is two VALU instructions (shift its invalid-tap bit to bit 31, OR it into the offset) -- no
  // compares, no 64-bit address arithmetic, no zero page: the range check of the descriptor supplies the zeros.
  auto dma_a_piece = [&](int it, _Float16* Ad, int dr, int ds, long uoff, int sh) {
    lds_void* dst = (lds_void*)(Ad + (it * ROWS_PASS + grp_row) * ROWH);
    if constexpr (BUF) {
      const unsigned voff = ((a_inv[it] << sh) & 0x80000000u) | a_off[it];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst, 16, (int)voff, (int)uoff, 0, 0);
    } else {
      const bool ok = (unsigned)(a_ih0[it] + dr) < (unsigned)o.H && (unsigned)(a_iw0[it] + ds) < (unsigned)o.W;
      const _Float16* src = ok ? a_row[it] + uoff : g_zero_page16 + a_cc[it];  // padding taps read zeros
      __builtin_amdgcn_global_load_lds((gbl_void*)src, dst, 16, 0, 0);
    }
  };
  auto dma_b_piece = [&](int it, _Float16* Bd, long boff) {
    lds_void* dst = (lds_void*)(Bd + (it * ROWS_PASS + grp_row) * ROWH);
    if constexpr (BUF)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, dst, 16, (int)b_off[it], (int)boff, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_void*)(b_ptr[it] + boff), dst, 16, 0, 0);
  };
  // wave-uniform offsets of the k tile the loader is at (halfs for the pointer form, bytes for the descriptor form)
  auto tile_offsets = [&](long& uoff, long& boff, int& sh) {
    const int dr = cur_r * p.dil, ds = cur_s * p.dil;
    uoff = ((long)dr * o.pitch + ds) * p.xs + (long)cur_cb * (2 * BK);
    boff = (long)load_t * (2 * BK);
    sh = 31 - (cur_r * p.S + cur_s);
    if constexpr (BUF) {
      uoff *= 2;
      boff *= 2;
    }
  };
  // branch-free advance of (tap, channel block) to the next k tile: keeps the hot loop one basic block
  // saturates at the last tile: the pipeline keeps issuing (redundant, never read) loads of it past the end
  // instead of branching around the DMA
  auto advance_tile = [&]() {
    const int adv = load_t + 1 < t_end ? 1 : 0;
    load_t += adv;
    const int s1 = cur_s + adv;
    const int ws = s1 == p.S ? 1 : 0;
    cur_s = ws ? 0 : s1;
    const int r1 = cur_r + ws;
    const int wr = r1 == p.R ? 1 : 0;
    cur_r = wr ? 0 : r1;
    cur_cb += wr;
  };
  // RS: the A loader walks (channel block, filter row) pairs on its own, two row tiles ahead of the MFMAs
  int a_r = 0, a_cb = t_begin / (p.R * p.S), a_q = 0;
  const int a_qn = (t_end - t_begin) / 3;
  auto rs_a_offsets = [&](long& uoff, int& sh) {
    uoff = (((long)a_r * o.pitch) * p.xs + (long)a_cb * (2 * BK)) * 2;
    sh = 31 - a_r;
  };
  auto rs_a_advance = [&]() {  // saturates at the last row tile, like advance_tile
    const int adv = a_q + 1 < a_qn ? 1 : 0;
    a_q += adv;
    const int r1 = a_r + adv;
    const int wr = r1 == 3 ? 1 : 0;
    a_r = wr ? 0 : r1;
    a_cb += wr;
  };
  auto rs_dma_a_tile = [&](int stage) {
    long uoff;
    int sh;
    rs_a_offsets(uoff, sh);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) dma_a_piece(it, As + stage * A_BUF, 0, 0, uoff, sh);
    rs_a_advance();
  };
  auto dma_tile = [&](int buf) {
    const int dr = cur_r * p.dil, ds = cur_s * p.dil;
    long uoff, boff;
    int sh;
    tile_offsets(uoff, boff, sh);
    _Float16* Ad = As + buf * A_BUF;
    _Float16* Bd = Bs + buf * B_BUF;
    if constexpr (!RS) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) dma_a_piece(it, Ad, dr, ds, uoff, sh);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) dma_b_piece(it, Bd, boff);
    advance_tile();
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (halfs): 16x16x32 operand map -- lane (r = lane&15, g = lane>>4) holds
  // row r of the tile and k = 8g..8g+7, i.e. chunk pl*4 + g of its LDS row
  const int lg = lane >> 4;
  int a_rd[TM][2], b_rd[TN][2];  // [tile][plane]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int row = wm * (BM / WM) + i * 16 + (lane & 15);
      a_rd[i][pl] = row * ROWH + (((pl * 4 + lg) ^ swz(row)) << 3);
    }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int row = wn * (BN / WN) + j * 16 + (lane & 15);
      b_rd[j][pl] = row * ROWH + (((pl * 4 + lg) ^ swz(row)) << 3);
    }

  struct AFrag {
    f16x8 h[TM], l[TM];
  };
  struct BFrag {
    f16x8 h[TH], l[TH];
  };
  // LDS byte addresses of this lane's fragment chunks in stage 0; row tile i / column tile j adds
  // i * TILE_OFF (the swizzle depends on row bits 1..3 only), a stage adds A_BUF / B_BUF halfs
  constexpr int TILE_OFF = 16 * ROWH * 2;
  const unsigned a_rd_hi = lds_addr(As + a_rd[0][0]), a_rd_lo = lds_addr(As + a_rd[0][1]);
  const unsigned b_rd_hi = lds_addr(Bs + b_rd[0][0]), b_rd_lo = lds_addr(Bs + b_rd[0][1]);
  // RS: fragment addresses per (row tile, tap s, plane): output row `row` sits in slot c = u(m0 + row) - u0, tap s reads
  // slot c + s - 1; the swizzle follows the slot, so the three taps need their own addresses
  // (kept as the centre slot per row tile; the three taps' addresses are rebuilt per step -- 4 VALU each -- because 24
  // loop-invariant address registers do not fit next to the fragments at two workgroups per CU)
  int a_c[RS ? TM : 1];
  const unsigned as_base = lds_addr(As);
  if constexpr (RS) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
      m = m < o.M ? m : o.M - 1;
      const int mw = fastdiv(m, o.mg_ow, o.sh_ow);
      int c = mw * (o.W + 1) + (m - mw * o.W) - rs_u0;
      a_c[i] = c < A_ROWS - 1 ? c : A_ROWS - 2;  // never taken when the host's gap bound holds
    }
  }
  // LDS byte address of chunk (pl * 4 + lg) of slot a_c[i] + d in A stage `stage`
  auto rs_addr = [&](int i, int d, int pl, int stage) {
    const int slot = a_c[i] + d;
    return as_base + (unsigned)(stage * (A_BUF * 2)) + (unsigned)(slot * (ROWH * 2)) + (unsigned)((((pl * 4 + lg) ^ swz(slot)) << 4));
  };
  BFrag b0, b1;
  AFrag af;  // ONE set of A fragments: the next tile's are read into each register after its last use
  constexpr int DPT = A_IT + B_IT;  // DMA instructions each wave issues per k tile
  using S2 = HalfSched<TM, TH, DPT, TERMS>;
  // all of this wave's DMA has landed and all of its LDS reads have returned; then rendezvous
  auto drain_and_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const int T = t_end - t_begin;
  // One step = tile t in LDS stage `cs`; its A fragments are in `af`, its column-half-0 W fragments in
  // b0; tiles t+1 .. t+NBUF-1 are in flight or landed (stage `ns` holds t+1):
  //   first half : W fragments of half 1 are read (b1) || MFMAs of column half 0
  //   wait       : counted vmcnt -- only tile t+1 has to be complete, the NBUF-2 younger tiles stay in
  //                flight across the barrier -- and all LDS reads of stage `cs` have returned
  //   barrier    : tile t+1 is complete for every wave and nobody reads stage `cs` any more
  //   second half: MFMAs of column half 1, one memory instruction issued in front of each (HalfSched):
  //                the DMA pieces of tile t+NBUF into stage `cs`, then the fragments of tile t+1
  // The body is ONE basic block with a fixed instruction order: MFMAs and LDS reads are volatile asm
  // (the builtin forms let the compiler put lgkmcnt(0) in front of MFMAs that needed no read, hoist
  // reads over the back-edge and rotate accumulators through copies), the LDS waits are counted by
  // hand (reads return in issue order), and past the end of k the DMA re-loads the last tile into a
  // stage nobody reads instead of branching.
  auto step_main = [&](int cs, int ns) {
    const unsigned bcur_hi = b_rd_hi + cs * (B_BUF * 2), bcur_lo = b_rd_lo + cs * (B_BUF * 2);
    static_for<0, TH>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      lds_read_pinned<(TH + jj) * TILE_OFF>(b1.h[jj], bcur_hi);
      if constexpr (TERMS == 3) lds_read_pinned<(TH + jj) * TILE_OFF>(b1.l[jj], bcur_lo);
    });
    // outstanding reads, oldest first: b0 (2*TH), af.l (TM), af.h (TM) of this tile, then b1 (2*TH)
    static_for<0, S2::NM>([&](auto K) {
      constexpr int k = decltype(K)::value;
      constexpr int term = k / (TM * TH), i = (k / TH) % TM, jj = k % TH;
      if constexpr (TERMS == 3) {
        if constexpr (k == 0) lgkm_wait<TM + 2 * TH>();                             // all of af.l (and b0)
        if constexpr (term == 1 && jj == 0) lgkm_wait<2 * TH + (TM - 1 - i)>();     // af.h[i]
      } else {
        if constexpr (term == 2 && jj == 0) lgkm_wait<TH + (TM - 1 - i)>();         // b0 and af.h[i]; younger: af.h[i+1..], b1
      }
      if constexpr (TERMS == 3 || term == 2) mfma_pinned(acc[i][jj], term == 0 ? af.l[i] : af.h[i], term == 1 ? b0.l[jj] : b0.h[jj]);
    });
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NBUF - 2) * DPT) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int dr = cur_r * p.dil, ds = cur_s * p.dil;
    long uoff, boff;
    int sh;
    tile_offsets(uoff, boff, sh);
    _Float16* Ad = As + cs * A_BUF;
    _Float16* Bd = Bs + cs * B_BUF;
    const unsigned anx_hi = a_rd_hi + ns * (A_BUF * 2), anx_lo = a_rd_lo + ns * (A_BUF * 2);
    const unsigned bnx_hi = b_rd_hi + ns * (B_BUF * 2), bnx_lo = b_rd_lo + ns * (B_BUF * 2);
    static_for<0, S2::NM + 1>([&](auto K) {
      constexpr int k = decltype(K)::value;
      static_for<0, S2::NMEM>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        if constexpr (S2::slot(q) == k) {
          if constexpr (q < A_IT) {
            dma_a_piece(q, Ad, dr, ds, uoff, sh);
          } else if constexpr (q < DPT) {
            dma_b_piece(q - A_IT, Bd, boff);
          } else if constexpr (q < DPT + S2::NB) {
            if constexpr (TERMS == 3) {
              constexpr int jj = (q - DPT) >> 1;
              if constexpr ((q - DPT) & 1)
                lds_read_pinned<jj * TILE_OFF>(b0.l[jj], bnx_lo);
              else
                lds_read_pinned<jj * TILE_OFF>(b0.h[jj], bnx_hi);
            } else {
              lds_read_pinned<(q - DPT) * TILE_OFF>(b0.h[q - DPT], bnx_hi);
            }
          } else if constexpr (q < DPT + S2::NB + S2::NAL) {
            constexpr int i = q - DPT - S2::NB;
            lds_read_pinned<i * TILE_OFF>(af.l[i], anx_lo);
          } else {
            constexpr int i = q - DPT - S2::NB - S2::NAL;
            lds_read_pinned<i * TILE_OFF>(af.h[i], anx_hi);
          }
        }
      });
      if constexpr (k < S2::NM) {
        constexpr int term = k / (TM * TH), i = (k / TH) % TM, jj = k % TH;
        if constexpr (TERMS == 3 || term == 2) mfma_pinned(acc[i][TH + jj], term == 0 ? af.l[i] : af.h[i], term == 1 ? b1.l[jj] : b1.h[jj]);
      }
    });
    advance_tile();
  };
  // RS step, phase PH = tap s of tile t: as step_main, except that (a) the A pieces are issued in phase 2 only -- the wide
  // tile of row tile q + 2 into the A stage `aq` that row tile q has just finished with -- and (b) the next tile's A
  // fragments come from the slot set of tap (PH + 1) % 3, in the other A stage after phase 2.  NBUF == 2: the mid-step
  // wait is vmcnt(0), so the varying number of DMA instructions per step needs no accounting.
  auto step_rs = [&](auto PHc, int cs, int ns, int aq) {
    constexpr int PH = decltype(PHc)::value;
    constexpr int NPH = (PH + 1) % 3;
    constexpr int A_CNT = PH == 2 ? A_IT : 0;
    constexpr int DPT_PH = A_CNT + B_IT;
    using S3 = HalfSched<TM, TH, DPT_PH, TERMS>;
    const unsigned bcur_hi = b_rd_hi + cs * (B_BUF * 2), bcur_lo = b_rd_lo + cs * (B_BUF * 2);
    static_for<0, TH>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      lds_read_pinned<(TH + jj) * TILE_OFF>(b1.h[jj], bcur_hi);
      if constexpr (TERMS == 3) lds_read_pinned<(TH + jj) * TILE_OFF>(b1.l[jj], bcur_lo);
    });
    static_for<0, S3::NM>([&](auto K) {
      constexpr int k = decltype(K)::value;
      constexpr int term = k / (TM * TH), i = (k / TH) % TM, jj = k % TH;
      if constexpr (TERMS == 3) {
        if constexpr (k == 0) lgkm_wait<TM + 2 * TH>();
        if constexpr (term == 1 && jj == 0) lgkm_wait<2 * TH + (TM - 1 - i)>();
      } else {
        if constexpr (term == 2 && jj == 0) lgkm_wait<TH + (TM - 1 - i)>();
      }
      if constexpr (TERMS == 3 || term == 2) mfma_pinned(acc[i][jj], term == 0 ? af.l[i] : af.h[i], term == 1 ? b0.l[jj] : b0.h[jj]);
    });
    // Phase 0 follows the step that put a wide A tile in flight as its YOUNGEST DMA instructions (W pieces first, A pieces
    // last, below): only the W tile of the next step has to be complete here, the A tile -- first read three steps from
    // now, and coming from HBM rather than L2 -- stays in flight across this barrier and is retired by phase 1's wait.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PH == 0 ? A_IT : 0) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    long uoff_b, boff, uoff_a = 0;
    int sh_b, sh_a = 0;
    tile_offsets(uoff_b, boff, sh_b);
    if constexpr (PH == 2) rs_a_offsets(uoff_a, sh_a);
    _Float16* Ad = As + aq * A_BUF;
    _Float16* Bd = Bs + cs * B_BUF;
    const int an = PH == 2 ? aq ^ 1 : aq;  // A stage of tile t + 1
    unsigned anx[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) anx[i][pl] = rs_addr(i, NPH - 1, pl, an);
    const unsigned bnx_hi = b_rd_hi + ns * (B_BUF * 2), bnx_lo = b_rd_lo + ns * (B_BUF * 2);
    static_for<0, S3::NM + 1>([&](auto K) {
      constexpr int k = decltype(K)::value;
      static_for<0, S3::NMEM>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        if constexpr (S3::slot(q) == k) {
          if constexpr (q < B_IT) {
            dma_b_piece(q, Bd, boff);
          } else if constexpr (q < DPT_PH) {
            dma_a_piece(q - B_IT, Ad, 0, 0, uoff_a, sh_a);
          } else if constexpr (q < DPT_PH + S3::NB) {
            if constexpr (TERMS == 3) {
              constexpr int jj = (q - DPT_PH) >> 1;
              if constexpr ((q - DPT_PH) & 1)
                lds_read_pinned<jj * TILE_OFF>(b0.l[jj], bnx_lo);
              else
                lds_read_pinned<jj * TILE_OFF>(b0.h[jj], bnx_hi);
            } else {
              lds_read_pinned<(q - DPT_PH) * TILE_OFF>(b0.h[q - DPT_PH], bnx_hi);
            }
          } else if constexpr (q < DPT_PH + S3::NB + S3::NAL) {
            constexpr int i = q - DPT_PH - S3::NB;
            lds_read_pinned<0>(af.l[i], anx[i][1]);
          } else {
            constexpr int i = q - DPT_PH - S3::NB - S3::NAL;
            lds_read_pinned<0>(af.h[i], anx[i][0]);
          }
        }
      });
      if constexpr (k < S3::NM) {
        constexpr int term = k / (TM * TH), i = (k / TH) % TM, jj = k % TH;
        if constexpr (TERMS == 3 || term == 2) mfma_pinned(acc[i][TH + jj], term == 0 ? af.l[i] : af.h[i], term == 1 ? b1.l[jj] : b1.h[jj]);
      }
    });
    advance_tile();
    if constexpr (PH == 2) rs_a_advance();
  };
  if constexpr (RS) {
    // prologue: row tile 0 + W tile 0 (waited for), then row tile 1 + W tile 1 in flight; fragments of tile 0 (tap 0)
    rs_dma_a_tile(0);
    dma_tile(0);
    drain_and_barrier();
    dma_tile(1);
    rs_dma_a_tile(1);  // youngest, like in phase 2 of the loop: phase 0's counted wait leaves exactly these in flight
    static_for<0, TH>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      lds_read_pinned<jj * TILE_OFF>(b0.h[jj], b_rd_hi);
      if constexpr (TERMS == 3) lds_read_pinned<jj * TILE_OFF>(b0.l[jj], b_rd_lo);
    });
    if constexpr (TERMS == 3)
      static_for<0, TM>([&](auto I) { lds_read_pinned<0>(af.l[decltype(I)::value], rs_addr(decltype(I)::value, -1, 1, 0)); });
    static_for<0, TM>([&](auto I) { lds_read_pinned<0>(af.h[decltype(I)::value], rs_addr(decltype(I)::value, -1, 0, 0)); });
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), see below
    // one row tile (3 k steps) per iteration; the W stage parity flips from one row tile to the next
    int aq = 0, cs = 0;
    for (int q3 = 0; q3 < a_qn; ++q3) {
      step_rs(std::integral_constant<int, 0>{}, cs, cs ^ 1, aq);
      step_rs(std::integral_constant<int, 1>{}, cs ^ 1, cs, aq);
      step_rs(std::integral_constant<int, 2>{}, cs, cs ^ 1, aq);
      cs ^= 1;
      aq ^= 1;
    }
  } else {
  // prologue: tile 0 -> stage 0 (waited for); tiles 1..NBUF-1 are put in flight behind it
  dma_tile(0);
  drain_and_barrier();
  for (int i = 1; i < NBUF; ++i) dma_tile(i);
  static_for<0, TH>([&](auto JJ) {  // same issue order as inside a step: b0, af.l, af.h
    constexpr int jj = decltype(JJ)::value;
    lds_read_pinned<jj * TILE_OFF>(b0.h[jj], b_rd_hi);
    if constexpr (TERMS == 3) lds_read_pinned<jj * TILE_OFF>(b0.l[jj], b_rd_lo);
  });
  if constexpr (TERMS == 3)
    static_for<0, TM>([&](auto I) { lds_read_pinned<decltype(I)::value * TILE_OFF>(af.l[decltype(I)::value], a_rd_lo); });
  static_for<0, TM>([&](auto I) { lds_read_pinned<decltype(I)::value * TILE_OFF>(af.h[decltype(I)::value], a_rd_hi); });
  // A wait the compiler's counter model can see (the asm ones it cannot): every kernel-argument load it still has
  // in flight retires HERE.  Otherwise the compiler may defer that wait to the first use inside the loop, where it
  // becomes an s_waitcnt lgkmcnt(0) per iteration that also drains the pinned LDS reads (seen once while adding
  // code to the epilogue: -13 % at batch 32).
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  int cs = 0, ns = 1;
  for (int t = 0; t < T; ++t) {
    step_main(cs, ns);
    cs = ns;
    ns = ns + 1 == NBUF ? 0 : ns + 1;
  }
  }  // !RS
  // the pinned MFMAs / reads are opaque to the compiler's hazard and counter tracking: retire everything
  // before the epilogue touches the accumulators
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail loads must land before LDS is reused / freed

  // ---- epilogue: bias, residual, ReLU, NHWC store ----
  // A split-K workgroup stores its raw fp32 partial tile into plane blockIdx.y of the workspace instead
  // (dense [M][Cout], no bias / residual / ReLU): the reduction kernel finishes the job.
  ConvParams16 q = p;
  q.x = o.x; q.w = o.w; q.bias = o.bias; q.y = o.y; q.gn_partial = o.gn_partial;
  q.H = o.H; q.W = o.W; q.pitch = o.pitch; q.OH = o.OH; q.OW = o.OW; q.M = o.M;
  if (p.splits > 1) {
    q.y = p.split_ws + (long)blk_y * o.M * p.Cout;
    q.ys = p.Cout;
    q.bias = nullptr;
    q.res_mode = 0;
    q.relu_cols = 0;
    q.out_split = 0;
    q.gn_partial = nullptr;
  }
  const float* res32 = reinterpret_cast<const float*>(q.res);
  const _Float16* res16 = reinterpret_cast<const _Float16*>(q.res);
  float* y32 = reinterpret_cast<float*>(q.y);
  _Float16* y16 = reinterpret_cast<_Float16*>(q.y);
  // v8 (round 3): the MFMAs run with SWAPPED operands (the W fragment as srcA, the activation fragment as srcB; the two
  // fragment layouts are identical, so the main loop is unchanged), which makes a lane own output PIXEL (lane & 15) of a
  // 16-row tile and, per column tile j, the four consecutive channels 16j + 4*(lane >> 4) + reg.  One v_permlane16_swap
  // per register (gfx950) then exchanges 16-lane rows between the accumulators of two neighbouring column tiles, after
  // which every lane holds EIGHT consecutive channels of its pixel: bias / residual are read and fp32 or S32 results
  // written with 16-byte accesses straight from registers.  The round-1/2 epilogue got the same ownership by transposing
  // every accumulator through LDS (64 ds_write_b32 + 16 ds_read_b128 per lane and a workgroup barrier, ~12.5-14 k cycles
  // per 128x128 tile, profiles/r02_conv_phase_stamps.txt) -- for the short-k layers (ResNet-34 layer1, the stem, all of
  // A2J) that was a third to a half of a workgroup's life.
  const int px = lane & 15;
  // channel offset of this lane inside a PAIR of column tiles after the row exchange (see swap8 below)
  const int nsub = (lg & 1) * 16 + (lg >> 1) * 8;
  const int n_wave = n0 + wn * (BN / WN);
  // rows {1, 3} of x <-> rows {0, 2} of y (16-lane rows): lane (px, g) ends up with channels
  //   pair base + 16 * (g & 1) + 8 * (g >> 1) + [0, 8)   as   x[0..3], y[0..3]
  auto swap8 = [&](f32x4& x, f32x4& y) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (__float_as_uint, not __builtin_bit_cast: the latter on an ext-vector ELEMENT reads element 0 with this compiler)
      const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
      x[r] = __uint_as_float(s[0]);
      y[r] = __uint_as_float(s[1]);
    }
  };
  if constexpr (POOL) {
    // conv + bias + ReLU patch -> LDS (fp32 [256][64], pitch 68 floats), then 7 x 8 pooled pixels x 8 channel groups
    constexpr int NP = TN / 2;
    constexpr int PITCH = BN + 4;
    static_assert(BM * PITCH * 4 <= NBUF * (A_BUF + B_BUF) * 2, "the pooling patch reuses the operand stages");
    float* patch = reinterpret_cast<float*>(smem);
    __syncthreads();  // every wave is done with the operand tiles in LDS
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (BM / WM) + i * 16 + px;
      const int pr = row / kPoolCols, pc = row - pr * kPoolCols;
      const int oy = pl_oy0 + pr, ox = pl_ox0 + pc;
      const bool ok = row < kPoolRows * kPoolCols && (unsigned)oy < (unsigned)o.OH && (unsigned)ox < (unsigned)o.OW;
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        f32x4 x = acc[i][2 * jp], y = acc[i][2 * jp + 1];
        swap8(x, y);
        const int n = jp * 32 + nsub;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(q.bias + n), b1 = *reinterpret_cast<const f32x4*>(q.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[e] = ok ? hn::relu(x[e] + b0[e]) : 0.f;
          y[e] = ok ? hn::relu(y[e] + b1[e]) : 0.f;
        }
        *reinterpret_cast<f32x4*>(patch + row * PITCH + n) = x;
        *reinterpret_cast<f32x4*>(patch + row * PITCH + n + 4) = y;
      }
    }
    __syncthreads();
    _Float16* y16p = reinterpret_cast<_Float16*>(q.y);
    for (int item = tid; item < kPoolPR * kPoolPC * (BN / 8); item += NT) {
      const int c8 = item & (BN / 8 - 1), pp = item / (BN / 8);
      const int pr = pp / kPoolPC, pc = pp - pr * kPoolPC;
      const int gy = kPoolPR * pl_ty + pr, gx = kPoolPC * pl_tx + pc;
      if (gy >= p.pool_oh || gx >= p.pool_ow) continue;
      f32x4 m0v = {0.f, 0.f, 0.f, 0.f}, m1v = m0v;   // ReLU output: 0 is the identity of max here
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* src = patch + ((2 * pr + dy) * kPoolCols + 2 * pc + dx) * PITCH + c8 * 8;
          const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            m0v[e] = hn::max_nan(m0v[e], a[e]);
            m1v[e] = hn::max_nan(m1v[e], b[e]);
          }
        }
      f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (p.range_flag && !(hn::range_mag(m0v[e], m1v[e]) <= 65504.f)) *p.range_flag = 1;
        const _Float16 h0 = (_Float16)m0v[e], h1 = (_Float16)m1v[e];
        hi[e] = h0;
        hi[4 + e] = h1;
        lo[e] = (_Float16)(m0v[e] - (float)h0);
        lo[4 + e] = (_Float16)(m1v[e] - (float)h1);
      }
      _Float16* dst = y16p + (((long)pl_img * p.pool_oh + gy) * p.pool_ow + gx) * p.ys + (c8 >> 2) * 64 + (c8 & 3) * 8;
      *reinterpret_cast<f16x8*>(dst) = hi;
      *reinterpret_cast<f16x8*>(dst + 32) = lo;
    }
    return;
  }
  if (p.vec_epi) {
    constexpr int NP = TN / 2;        // column-tile pairs per wave
    f32x4 bias0[NP], bias1[NP];
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int n = n_wave + jp * 32 + nsub;
      bias0[jp] = f32x4{0.f, 0.f, 0.f, 0.f};
      bias1[jp] = bias0[jp];
      if (q.bias && n < p.Cout) {
        bias0[jp] = *reinterpret_cast<const f32x4*>(q.bias + n);
        bias1[jp] = *reinterpret_cast<const f32x4*>(q.bias + n + 4);
      }
    }
    // S32 residual of the output's own shape (every ResNet block): ALL of this lane's 16-byte pieces are requested
    // here, before the first use, so the loop below waits for ONE memory latency
    constexpr bool PREF = TM * NP <= 8;   // <= 64 VGPRs of prefetched residual (the fragments are dead by now)
    f16x8 rpre_h[PREF ? TM * NP : 1], rpre_l[PREF ? TM * NP : 1];
    const bool use_pre = PREF && q.res_mode == 1 && q.res_split;
    if constexpr (PREF) {
      if (use_pre) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) {
            int m = m0 + wm * (BM / WM) + i * 16 + px;
            m = m < o.M ? m : o.M - 1;
            int n = n_wave + jp * 32 + nsub;
            n = n < p.Cout ? n : 0;
            const _Float16* q16 = res16 + (long)m * q.rs + (n >> 5) * 64 + (n & 31);
            rpre_h[i * NP + jp] = *reinterpret_cast<const f16x8*>(q16);
            rpre_l[i * NP + jp] = *reinterpret_cast<const f16x8*>(q16 + 32);
          }
      }
    }
    // The bias / residual loads above retire HERE, once, in a form the compiler's counter model sees.  Otherwise it
    // cannot tell at the joins below whether they are still in flight, and because loads and stores share vmcnt it
    // puts s_waitcnt vmcnt(0) in front of later bias uses -- i.e. a store batch would wait for the previous batch's
    // stores to be acknowledged by memory (~1 k cycles each).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int i = 0; i < TM; i += 2) {  // 32 rows (two 16-row tiles) per pass: one GroupNorm row group
      // GroupNorm statistics of this 32-row group, per 8-channel unit, split at the image boundary
      // (a group touches at most two images when OH*OW >= 32): [sum, sumsq] of image A, then of image A+1
      float gsum[NP][4];
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int e = 0; e < 4; ++e) gsum[jp][e] = 0.f;
      const int m_grp = m0 + wm * (BM / WM) + i * 16;
      const int m_split = (fastdiv(m_grp < o.M ? m_grp : o.M - 1, o.mg_ohow, o.sh_ohow) + 1) * ohow;  // first row of the next image
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = m_grp + ii * 16 + px;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          f32x4 x = acc[i + ii][2 * jp], y = acc[i + ii][2 * jp + 1];
          swap8(x, y);
          const int n = n_wave + jp * 32 + nsub;
          if (m >= o.M || n >= p.Cout) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // + 0 when there is no bias (exact)
            v[e] = x[e] + bias0[jp][e];
            v[4 + e] = y[e] + bias1[jp][e];
          }
          if (q.gn_partial) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1 += v[e];
              s2 += v[e] * v[e];
            }
            const bool second = m >= m_split;
            gsum[jp][0] += second ? 0.f : s1;
            gsum[jp][1] += second ? 0.f : s2;
            gsum[jp][2] += second ? s1 : 0.f;
            gsum[jp][3] += second ? s2 : 0.f;
          }
          if constexpr (PREF) {
            if (use_pre) epi_finish8(q, m, n, v, ohow, &rpre_h[(i + ii) * NP + jp], &rpre_l[(i + ii) * NP + jp]);
            else epi_finish8(q, m, n, v, ohow);
          } else {
            epi_finish8(q, m, n, v, ohow);
          }
        }
      }
      if (q.gn_partial) {
        // the 16 lanes of a row (equal lane >> 4) hold the same channel unit for 16 different pixels: fixed-order butterfly
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
#pragma unroll
          for (int ofs = 1; ofs < 16; ofs <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) gsum[jp][e] += __shfl_xor(gsum[jp][e], ofs);
          const int n = n_wave + jp * 32 + nsub;
          if (px == 0 && n < p.Cout && m_grp < o.M) {
            f32x4 o4 = {gsum[jp][0], gsum[jp][1], gsum[jp][2], gsum[jp][3]};
            *reinterpret_cast<f32x4*>(q.gn_partial + ((long)(m_grp >> 5) * p.gn_units + (n >> 3)) * 4) = o4;
          }
        }
      }
    }
    return;
  }
  // Scalar path (ragged Cout such as the 5-channel FCOS outputs, or unaligned fp32 strides): the raw (swapped) MFMA
  // layout -- lane = pixel (lane & 15), registers = channels 16j + 4*(lane >> 4) + r
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WM) + i * 16 + px;
    if (m >= o.M) continue;
    long rpix = 0;
    if (p.res_mode == 1) {
      rpix = (long)m;
    } else if (p.res_mode == 2) {
      const int img = m / ohow;
      const int rem = m - img * ohow;
      const int oh = rem / o.OW, ow = rem - oh * o.OW;
      const int sh_ = (int)(((long)oh * p.res_h) / o.OH), sw_ = (int)(((long)ow * p.res_w) / o.OW);
      rpix = ((long)img * p.res_h + sh_) * p.res_w + sw_;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_wave + j * 16 + lg * 4 + r;
        if (n >= p.Cout) continue;
        float v = acc[i][j][r];
        if (q.bias) v += q.bias[n];
        if (p.res_mode) {
          if (p.res_split) {
            const _Float16* q = res16 + rpix * p.rs + (n >> 5) * 64 + (n & 31);
            v += (float)q[0] + (float)q[32];
          } else {
            v += res32[rpix * p.rs + n];
          }
        }
        if (n < p.relu_cols) v = hn::relu(v);
        if (p.out_split) {
          if (p.range_flag) hn::range_note(p.range_flag, v);
          _Float16* q = y16 + (long)m * p.ys + (n >> 5) * 64 + (n & 31);
          const _Float16 h = (_Float16)v;
          q[0] = h;
          q[32] = (_Float16)(v - (float)h);
        } else {
          y32[(long)m * p.ys + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, bool RS = false, bool POOL = false, int TERMS = 3>
__global__ __launch_bounds__(WM* WN * 64, (BM * BN / (WM * WN) > 64 * 64 ? 1 : 2))
void conv_igemm_f16x3_kernel(const ConvParams16 p) {
  conv_igemm_f16x3_body<BM, BN, WM, WN, NBUF, BUF, RS, POOL, TERMS>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ---- grouped launch whose members carry their own tile shape (round 5).  The FCOS tower layers at small batch are ONE grouped
// launch over (tower, FPN level) members of very different sizes -- at batch 1: 214 + 54 + 14 tiles of 128 x 128 per tower, 564
// in all on the chip's 512 slots: one full round and a second one that is 10 % full, i.e. the layer takes the time of three
// tiles per CU where 2.2 would do (MFMA-bound workgroups: a CU's two slots share its matrix pipes).  Equal tiles cannot be
// packed better; HALF tiles at the END of the dispatch order can: the small members (the stride-16 / stride-32 levels, a
// quarter of the rows) run the 64 x 128 per-tap form -- same k order, bit-identical results -- so that the last round is filled
// with half-size workgroups.  Both bodies live in one kernel (256 threads, the row-shared form's dynamic LDS block);
// blockIdx.z = member as in the plain grouped launch.
__global__ __launch_bounds__(256, 2) void conv_igemm_f16x3_mixed_kernel(const ConvParams16 p) {
  typedef __attribute__((address_space(4))) const ConvParams16 KArgs;
  KArgs* kp = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  if ((kp->small_mask >> blockIdx.z) & 1)
    conv_igemm_f16x3_body<64, 128, 2, 2, 3, true, false, false, 3, true>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
  else
    conv_igemm_f16x3_body<128, 128, 2, 2, 2, true, true, false, 3>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ---- heterogeneous launches (hn_conv2d_nhwc_f16x3_multi): up to HN_CONV_MULTI_MAX INDEPENDENT convolutions of different shapes
// (channels, filter, stride, residual, split-K plan: anything) in ONE grid.  The kernel argument is a table of complete
// parameter blocks; a workgroup finds its member from the prefix sums of the members' workgroup counts, copies that member's
// block out of the kernel-argument segment (scalar loads at a uniform offset) and runs the ordinary body on it with
// member-local coordinates.  What it buys: small grids that are independent of each other (the downsample 1x1 beside conv1 of
// a residual block, the classification head beside layer4 of the A2J trunk) fill each other's idle CUs and cost one launch --
// and ONE reduction launch for all their split-K members -- instead of one each.
constexpr int kMultiMax = HN_CONV_MULTI_MAX;
struct MultiParams16 {
  int count;
  int start[kMultiMax + 1];   // first workgroup of member i (member grids are nblocks x splits, split-major)
  ConvParams16 m[kMultiMax];
};

__device__ __forceinline__ void load_member16(const MultiParams16& unused_by_value_copy, int g, ConvParams16& p) {
  (void)unused_by_value_copy;
  typedef __attribute__((address_space(4))) const unsigned KW;
  typedef __attribute__((address_space(4))) const char KB;
  KB* base = (KB*)__builtin_amdgcn_kernarg_segment_ptr();
  KW* src = (KW*)(base + offsetof(MultiParams16, m) + (size_t)g * sizeof(ConvParams16));
  unsigned* dst = reinterpret_cast<unsigned*>(&p);
  static_assert(sizeof(ConvParams16) % 4 == 0, "parameter block is copied word by word");
#pragma unroll
  for (int i = 0; i < (int)(sizeof(ConvParams16) / 4); ++i) dst[i] = src[i];
}

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(WM* WN * 64, (BM * BN / (WM * WN) > 64 * 64 ? 1 : 2))
void conv_igemm_f16x3_multi_kernel(const MultiParams16 mp) {
  typedef __attribute__((address_space(4))) const MultiParams16 KM;
  KM* km = (KM*)__builtin_amdgcn_kernarg_segment_ptr();
  const int bid = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < kMultiMax; ++i) g = (i < km->count && bid >= km->start[i]) ? i : g;
  int local = bid - km->start[g];
  ConvParams16 p;
  load_member16(mp, g, p);
  int by = 0;
  while (local >= p.nblocks) {   // wave-uniform: at most 16 splits
    local -= p.nblocks;
    ++by;
  }
  conv_igemm_f16x3_body<BM, BN, WM, WN, NBUF, true, false, false, 3>(p, local, by, 0);
}

// the reductions of a multi launch's split-K members as ONE launch: gridDim.y = member
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const MultiParams16 mp) {
  ConvParams16 p;
  load_member16(mp, (int)blockIdx.y, p);
  if (p.splits <= 1) return;
  splitk_reduce_body(p);
}

// ---- host-side launch planning, shared by the launcher and by hn_conv2d_f16x3_uses_rs (ONE definition of each decision) ----
// Split-K: S workgroups share an output tile, each sums ceil(ktiles / S) k tiles into its own fp32 plane, a second launch
// adds the planes in z order.  Round 4: the split count comes from a small cost model fitted to tools/splitk_sweep.py
// (profiles/r04_splitk_sweep_b1.txt) instead of "fill 512 slots whenever the grid is below 256 workgroups" -- that rule split
// grids of ~200 workgroups three ways (1.25 rounds of workgroups plus a reduction that streams three planes: ResNet-34 layer2
// at batch 1 30.8 us against 20.4 us unsplit) and gave 11 x 11 layers 16 planes whose serial reads cost the reduction more
// than the shorter k loops gave back.
//   time(S) = ceil(nblocks * S / slots) * (t_fix + ceil(ktiles / S) * t_k)  +  (S > 1: t_red0 + S * max(t_plane_min, plane_bytes / bw))
// slots = resident workgroups of the tile form on the chip; t_fix / t_k = fixed time of a workgroup (prologue, pipeline fill,
// epilogue) and time per k tile, from pairs of sweep rows (the 4-wave tiles spend 12-16 us outside their k loop at these grid
// sizes, which is why halving a 36-tile loop does not pay for a reduction; the 2-wave 32x64 tile 3 us).
struct SplitModel { int slots; double t_fix, t_k; };
static SplitModel split_model(int bm, int bn) {
  const int cus = 256;
  if (bm == 32 && bn == 64) return {3 * cus, 3.0, 0.27};
  if (bm == 64 && bn == 64) return {3 * cus, 12.0, 0.12};
  if (bm == 64 && bn == 128) return {2 * cus, 16.0, 0.12};
  if (bm == 128 && bn == 64) return {2 * cus, 16.0, 0.20};
  if (bm == 128 && bn == 32) return {2 * cus, 12.0, 0.15};
  return {2 * cus, 20.0, 0.40};   // 128x128 and the 256-row tiles
}
static void plan_splits(ConvParams16& p, int bm, int bn) {   // needs p.nblocks; sets p.splits / p.kt_per
  p.splits = 1;
  p.kt_per = p.ktiles;
  if (p.groups > 1) return;  // grouped problems never split (the grid is already groups x larger)
  if (!(p.split_ws && p.splitk_mode >= 0 && p.vec_epi && !p.gn_partial)) return;
  const int64_t plane_bytes = (int64_t)p.M * p.Cout * 4;
  int want = 1;
  if (p.splitk_mode >= 2) {   // sweeps (tools/splitk_sweep.py): exactly this many
    want = p.splitk_mode < 16 ? p.splitk_mode : 16;
    want = want < p.ktiles ? want : p.ktiles;
  } else if (hn::env_flags().splitk_fill512) {   // the round-1..3 rule (A/B reference): fill 512 slots below 256 workgroups
    const int min_tiles = p.splitk_mode > 0 ? 8 : 128, min_per = p.splitk_mode > 0 ? 4 : 16;
    if (p.nblocks >= 256 || p.ktiles < min_tiles) return;
    want = hn::cdiv(512, p.nblocks);
    want = want < p.ktiles / min_per ? want : p.ktiles / min_per;
    want = want < 16 ? want : 16;
  } else {
    // eager callers that pay for the second launch on the host (desc.splitk = 0) split long loops only
    const int min_tiles = p.splitk_mode > 0 ? 8 : 128, min_per = p.splitk_mode > 0 ? 4 : 16;
    if (p.ktiles < min_tiles) return;
    SplitModel sm = split_model(bm, bn);
    const hn::Tuning& tn = hn::tuning();   // development: scale factors of the model's constants (hn_set_tuning; all 1.0)
    sm.t_fix *= tn.splitk_fix;
    sm.t_k *= tn.splitk_tk;
    const double t_red0 = 4.5 * tn.splitk_red0, t_plane_min = 0.3 * tn.splitk_plane, bw = 3.0e6;   // us, us per plane, bytes per us
    double best = (double)hn::cdiv(p.nblocks, sm.slots) * (sm.t_fix + p.ktiles * sm.t_k);
    for (int s = 2; s <= 16 && p.ktiles / s >= min_per; ++s) {
      const int kt = hn::cdiv(p.ktiles, s), se = hn::cdiv(p.ktiles, kt);
      if (se != s) continue;   // (the same plan as a smaller s)
      const double per_plane = (double)plane_bytes / bw;
      const double t = (double)hn::cdiv((int64_t)p.nblocks * s, sm.slots) * (sm.t_fix + kt * sm.t_k) + t_red0 +
                       s * (per_plane > t_plane_min ? per_plane : t_plane_min);
      if (t < best) {
        best = t;
        want = s;
      }
    }
  }
  if ((int64_t)want * plane_bytes > p.split_ws_bytes) want = (int)(p.split_ws_bytes / plane_bytes);
  if (want > 1) {
    p.kt_per = hn::cdiv(p.ktiles, want);
    p.splits = hn::cdiv(p.ktiles, p.kt_per);  // every split has at least one tile
  }
}

// Row-shared A operand: 3x3 / stride 1 / pad 1 / dilation 1 on a dense-row tensor, single pass (no split-K), a tile form
// whose two wide A stages fit beside the W stages with eight waves per CU, and every (member's) image row long enough
// that the gap slots of a wide tile fit (a tile of BM + 1 pixels crosses at most BM / W + 1 row ends).
static bool rs_tile_form(int bm, int bn, int waves, int nbuf, bool buf) {
  const int lds = nbuf * (bm + waves * 8 + bn) * ROWH * 2;
  return buf && nbuf == 2 && (waves == 4 || waves == 8) && lds <= (waves == 4 ? 80 : 160) * 1024 - 2048;
}
static bool rs_geometry(const ConvParams16& p) {
  return p.R == 3 && p.S == 3 && p.stride == 1 && p.dil == 1 && p.pad == 1 && p.pitch == p.W && p.OH == p.H && p.OW == p.W &&
         !hn::env_flags().no_rs;
}
static bool rs_will_run(const ConvParams16& p, int bm, int bn, int waves, int nbuf, bool buf) {   // needs p.splits
  if (!(p.rs_ok && rs_geometry(p) && rs_tile_form(bm, bn, waves, nbuf, buf)) || p.splits != 1) return false;
  const int spare = waves * 8 - 2;  // A_ROWS - BM - 2 gap slots
  if ((bm + 1) / p.W + 1 > spare) return false;
  for (int g = 0; p.groups > 1 && g < p.groups; ++g)
    if ((bm + 1) / p.gW[g] + 1 > spare) return false;
  return true;
}

// Mixed grouped launch (conv_igemm_f16x3_mixed_kernel): which members take the 64-row form.  Only where the last round of
// 128 x 128 tiles would be at most a QUARTER full on a grid of at most three rounds (the FCOS towers / tower0 at batch 1 and 2:
// 2.277 -> 2.247 ms and 3.153 -> 3.114 ms per step in the frame; with a fuller last round -- batch 3 and 4 -- the per-tap 64-row
// form's lower efficiency costs more than the packing gains, +70 / +80 us, tools/probes/exp/mixed_tiles.sh): the members with at
// most half the rows of the largest one.  0 = the plain grouped launch.
static int mixed_small_mask(const ConvParams16& p, int tiles_n) {
  if (p.groups <= 1 || p.terms != 3 || hn::env_flags().no_mixed) return 0;
  const int slots = 512;
  int64_t total = 0;
  int max_m = 0;
  for (int g = 0; g < p.groups; ++g) {
    total += (int64_t)hn::cdiv(p.gM[g], 128) * tiles_n;
    max_m = max_m > p.gM[g] ? max_m : p.gM[g];
  }
  const int64_t tail = total % slots;
  if (total <= slots || total > 3 * slots || tail == 0 || tail > slots / 4) return 0;
  int mask = 0;
  for (int g = 0; g < p.groups; ++g)
    if ((int64_t)p.gM[g] * 2 <= max_m) mask |= 1 << g;
  return mask;
}

template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, int TERMS = 3>
int launch16_impl(const ConvParams16& p0, hipStream_t st) {
  ConvParams16 p = p0;
  p.tiles_m = hn::cdiv(p.M, BM);
  p.tiles_n = hn::cdiv(p.Cout, BN);
  p.nblocks = p.tiles_m * p.tiles_n;
  plan_splits(p, BM, BN);
  int grid_x = p.nblocks;
  if (p.groups > 1) {
    grid_x = 0;
    for (int g = 0; g < p.groups; ++g) {
      p.gnblocks[g] = hn::cdiv(p.gM[g], BM) * p.tiles_n;
      grid_x = grid_x > p.gnblocks[g] ? grid_x : p.gnblocks[g];
    }
  }
  if (p.pool_ty > 0) {  // fused conv + ReLU + max pooling (hn_conv_stem_pool_f16x3): one workgroup per 15 x 17 patch
    if constexpr (BUF && BM == 256 && BN == 64 && WM == 4 && WN == 1 && NBUF == 2) {
      p.tiles_n = 1;
      p.nblocks = p.N * p.pool_ty * p.pool_tx;
      p.tiles_m = p.nblocks;
      p.splits = 1;
      p.kt_per = p.ktiles;
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, true, false, true, TERMS>), dim3(p.nblocks),
                         dim3(WM * WN * 64), 0, st, p);
      HN_CHECK_LAUNCH("conv_igemm_f16x3_kernel<pool>");
      return HN_OK;
    } else {
      return hn::fail(HN_ERR_ARG, "the pooling epilogue exists for the 256x64 tile only");
    }
  }
  // Row-shared A operand: 3x3 / stride 1 / pad 1 / dilation 1 on a dense-row tensor, no split-K, and every (member's) image
  // row long enough that the gap slots of a wide tile fit (a tile of BM + 1 pixels crosses at most BM / W + 1 row ends).
  bool rs = false;
  constexpr int RS_LDS_BYTES = NBUF * (BM + WM * WN * 8 + BN) * ROWH * 2;
  // eight waves per CU must remain: two 4-wave workgroups or one 8-wave workgroup (rs_tile_form, evaluated at compile time
  // here so that only the eligible tile forms instantiate the RS kernel)
  if constexpr (BUF && NBUF == 2 && (WM * WN == 4 || WM * WN == 8) && RS_LDS_BYTES <= (WM * WN == 4 ? 80 : 160) * 1024 - 2048) {
    rs = rs_will_run(p, BM, BN, WM * WN, NBUF, BUF);
    if constexpr (BM == 128 && BN == 128 && WM == 2 && WN == 2 && TERMS == 3) {
      const int mask = rs ? mixed_small_mask(p, p.tiles_n) : 0;
      if (mask) {   // members with their own tile shape: the small ones as 64 x 128 per-tap tiles (see the kernel's comment)
        p.small_mask = mask;
        grid_x = 0;
        for (int g = 0; g < p.groups; ++g) {
          p.gnblocks[g] = hn::cdiv(p.gM[g], (mask >> g) & 1 ? 64 : 128) * p.tiles_n;
          grid_x = grid_x > p.gnblocks[g] ? grid_x : p.gnblocks[g];
        }
        constexpr int LDS_MIXED = RS_LDS_BYTES > 3 * (64 + 128) * ROWH * 2 ? RS_LDS_BYTES : 3 * (64 + 128) * ROWH * 2;
        static bool mixed_attr[64] = {};
        int dev = 0;
        HN_CHECK_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !mixed_attr[dev]) {
          HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_f16x3_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_MIXED));
          if (dev >= 0 && dev < 64) mixed_attr[dev] = true;
        }
        hipLaunchKernelGGL(conv_igemm_f16x3_mixed_kernel, dim3(grid_x, 1, p.groups), dim3(256), LDS_MIXED, st, p);
        HN_CHECK_LAUNCH("conv_igemm_f16x3_mixed_kernel");
        return HN_OK;
      }
    }
    if (rs) {
      constexpr int LDS_BYTES = RS_LDS_BYTES;
      static bool attr_set[64] = {};  // per device: the attribute belongs to the function's image on the current device
      int dev = 0;
      HN_CHECK_HIP(hipGetDevice(&dev));
      if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, true, true, false, TERMS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
      }
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, true, true, false, TERMS>),
                         dim3(grid_x, 1, p.groups > 1 ? p.groups : 1), dim3(WM * WN * 64), LDS_BYTES, st, p);
    }
  }
  if (!rs)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, BUF, false, false, TERMS>),
                       dim3(grid_x, p.splits, p.groups > 1 ? p.groups : 1), dim3(WM * WN * 64), 0, st, p);
  HN_CHECK_LAUNCH("conv_igemm_f16x3_kernel");
  if (p.splits > 1) {
    const long total = (long)p.M * (p.Cout >> 3);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, p);
    HN_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return HN_OK;
}

// mg / sh such that floor(n / d) == umulhi(n, mg) >> sh for every 0 <= n < 2^31 (d >= 2); d == 1 -> mg = 0 (identity).
// With k = ceil(log2 d) and mg = ceil(2^(31+k) / d): mg * d - 2^(31+k) < d <= 2^k, and n * that < 2^(31+k), so the
// quotient is exact.
static void magic_u31(unsigned d, unsigned& mg, unsigned& sh) {
  if (d <= 1) { mg = 0; sh = 0; return; }
  unsigned k = 0;
  while ((1ull << k) < d) ++k;
  const unsigned long long pw = 1ull << (31 + k);
  mg = (unsigned)((pw + d - 1) / d);
  sh = k - 1;   // umulhi drops 32 bits: total shift 31 + k
}

// Operand extents for the buffer descriptors of the v6 addressing.  Falls back to the pointer-form kernel (one
// instantiation, 128x128) when an operand spans 2 GB or more (bit 31 of an offset must stay out of range) or the
// filter has more than 32 taps.
// Magic numbers of the index divisions and the extents of the buffer descriptors (v6 addressing).  Returns false when an operand
// spans 2 GB or more (bit 31 of an offset must stay out of range) or the filter has more than 32 taps: the pointer-form kernel.
static bool finish_params16(ConvParams16& p) {
  magic_u31((unsigned)(p.OH * p.OW), p.mg_ohow, p.sh_ohow);
  magic_u31((unsigned)p.OW, p.mg_ow, p.sh_ow);
  magic_u31((unsigned)p.W + 1u, p.mg_w1, p.sh_w1);
  magic_u31((unsigned)p.H, p.mg_h, p.sh_h);
  magic_u31(p.pool_ty > 0 ? (unsigned)(p.pool_ty * p.pool_tx) : 1u, p.mg_pt, p.sh_pt);
  magic_u31(p.pool_ty > 0 ? (unsigned)p.pool_tx : 1u, p.mg_ptx, p.sh_ptx);
  for (int g = 0; g < HN_CONV_MAX_GROUP; ++g) {
    const bool on = p.groups > 1 && g < p.groups;
    magic_u31(on ? (unsigned)(p.gOH[g] * p.gOW[g]) : 1u, p.gmg_ohow[g], p.gsh_ohow[g]);
    magic_u31(on ? (unsigned)p.gOW[g] : 1u, p.gmg_ow[g], p.gsh_ow[g]);
    magic_u31(on ? (unsigned)p.gW[g] + 1u : 1u, p.gmg_w1[g], p.gsh_w1[g]);
    magic_u31(on ? (unsigned)p.gH[g] : 1u, p.gmg_h[g], p.gsh_h[g]);
  }
  // (the padded index of the last slot stays far below 2^31: M < 2^31 / (1 + 1/W) is implied by the extent check below
  // for every xs >= 2)
  p.rs_ok = p.rs_ok && rs_geometry(p);
  const int64_t lim = (int64_t)1 << 31;
  auto extent = [&](int h, int pitch) {
    // bytes the A descriptor covers: it starts pad rows + pad columns before x (kernel: a_shift) and ends at the last
    // pixel's lo run (an upper bound for channel slices: the range check only has to keep bit 31 outside)
    return ((int64_t)p.N * h * pitch * p.xs + (p.lo_off > 32 ? p.lo_off : 0)) * 2 +
           ((int64_t)p.pad * pitch + p.pad) * p.xs * 2;
  };
  bool ok = p.R * p.S <= 32 && (int64_t)p.Cout * p.Ktot * 4 < lim;
  int64_t ea = extent(p.H, p.pitch);
  ok = ok && ea < lim;
  p.a_records = (unsigned)(ok ? ea : 0);
  p.b_records = (unsigned)(ok ? (int64_t)p.Cout * p.Ktot * 4 : 0);
  for (int g = 0; g < HN_CONV_MAX_GROUP; ++g) p.ga_records[g] = 0;
  if (p.groups > 1)
    for (int g = 0; g < p.groups; ++g) {
      ea = extent(p.gH[g], p.gW[g]);
      ok = ok && ea < lim;
      p.ga_records[g] = (unsigned)(ea < lim ? ea : 0);
    }
  return ok;
}

template <int BM, int BN, int WM, int WN, int NBUF, bool ALLOW_F16X1 = true>
int launch16(const ConvParams16& p0, hipStream_t st) {
  ConvParams16 p = p0;
  const bool ok = finish_params16(p);
  if (p.terms == 1) {   // throughput mode: descriptor-form kernels of the tiles the engines use
    if constexpr (ALLOW_F16X1) {
      if (ok) return launch16_impl<BM, BN, WM, WN, NBUF, true, 1>(p, st);
    }
    return hn::fail(HN_ERR_ARG, "the f16x1 mode exists for the descriptor-form kernels of the engine tiles only");
  }
  if (ok) return launch16_impl<BM, BN, WM, WN, NBUF, true>(p, st);
  return launch16_impl<128, 128, 2, 2, 2, false>(p, st);
}

int64_t nblocks16(const hn_conv_desc* d, int bm, int bn) {
  const int64_t M = (int64_t)d->n * d->oh * d->ow;
  return (int64_t)hn::cdiv(M, bm) * hn::cdiv(d->cout, bn);
}

}  // namespace

extern "C" int hn_conv2d_f16x3_pick_tile(const hn_conv_desc* d) {
  if (!d) return HN_TILE_64x64;
  if (d->tile != HN_TILE_AUTO) return d->tile;
  // From tools/tile_sweep.py (every distinct conv shape of the pipeline, batch 1 and 32, clocks kept hot):
  // within 0.3 % (batch 32) / 2 % (batch 1) of the best tile per shape.
  if (d->cout <= 32) return nblocks16(d, 128, 32) < 64 ? HN_TILE_32x64 : HN_TILE_128x32;
  const int64_t want = 256;
  // the largest tile wins as soon as it yields one workgroup per CU
  if (d->cout > 64 && nblocks16(d, 128, 128) >= want) return HN_TILE_128x128;
  // Cout <= 64 with many rows: four waves stacked along M keep the 64x64 wave tile of the big kernel (+3 %)
  if (d->cout <= 64 && nblocks16(d, 256, 64) >= 2 * want) return HN_TILE_256x64;
  // mid-size grids: 128 columns per workgroup when there are that many (fewer weight re-reads), else 128 rows
  if (d->cout >= 128 && nblocks16(d, 64, 128) >= 192) return HN_TILE_64x128;
  if (nblocks16(d, 128, 64) >= want) return HN_TILE_128x64;
  // tiny grids (11x11 maps, batch 1): 2-wave workgroups double the number of CUs that have work.  (Round 4: in isolation
  // tools/tile_sweep.py prefers 64x64 for the long-k layers here -- 512 -> 512 3x3 on 11 x 11 17.9 -> 15.1 us -- but inside the
  // batch-1 frame the rule costs 15 us, 2.327 -> 2.342 ms over three same-box pairs: not taken.)
  if (nblocks16(d, 64, 64) < 128) return HN_TILE_32x64;
  return HN_TILE_64x64;
}

// The tile forms behind the HN_TILE_* ids: {BM, BN, waves, LDS stages}; conv16_run's switch instantiates exactly these.
struct TileForm { int bm, bn, waves, nbuf; };
static bool rs32_preferred(const hn_conv_desc* d) {
  return d->r == 3 && d->s == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && !hn::env_flags().no_rs && !hn::env_flags().no_rs32;
}
static TileForm tile_form(int tile, bool rs32) {
  switch (tile) {
    case HN_TILE_128x128: return {128, 128, 4, 2};
    case HN_TILE_128x64: return {128, 64, 4, 2};
    case HN_TILE_64x64: return {64, 64, 4, 3};
    case HN_TILE_128x32: return {128, 32, 4, rs32 ? 2 : 3};   // 2 stages only when the row-shared form will run
    case HN_TILE_64x128: return {64, 128, 4, 3};
    case HN_TILE_256x128: return {256, 128, 4, 2};
    case HN_TILE_32x64: return {32, 64, 2, 4};
    case HN_TILE_256x64: return {256, 64, 4, 2};
    case HN_TILE_256x128_W8: return {256, 128, 8, 2};
    case HN_TILE_256x64_W8: return {256, 64, 8, 2};
    case HN_TILE_128x256_W8: return {128, 256, 8, 2};
    default: return {0, 0, 0, 0};
  }
}

// Does a launch of this descriptor run the row-shared-A kernel?  Evaluates the SAME functions as the launcher (tile pick,
// plan_splits, rs_will_run) for a call through hn_conv2d_nhwc_f16x3_ws with the engines' 32 MiB workspace (d->splitk < 0:
// no workspace), so a layer that goes split-K -- which uses the per-tap form -- answers 0.  For a grouped launch pass the
// smallest member width in d->w, all members' rows as n * oh * ow and the picked tile in d->tile.
extern "C" int hn_conv2d_f16x3_uses_halo(const hn_conv_desc* d, int has_residual) {
  return d && hn::conv3x3_halo_applies(d, false, false, has_residual ? (const void*)d : nullptr) ? 1 : 0;
}

extern "C" int hn_conv2d_f16x3_uses_stream(const hn_conv_desc* d) {
  return d && hn::conv1x1_stream_applies(d, false, false) ? 1 : 0;
}

extern "C" int hn_conv2d_f16x3_uses_rs(const hn_conv_desc* d) {
  if (!d || d->w <= 0 || d->cin <= 0) return 0;
  if (hn::conv3x3_halo_applies(d, false, false, d->res_mode ? (const void*)d : nullptr)) return 0;
  const int tile = hn_conv2d_f16x3_pick_tile(d);
  ConvParams16 p;
  p.R = d->r; p.S = d->s; p.stride = d->stride; p.dil = d->dil; p.pad = d->pad;
  // output size from the geometry (for a grouped launch the caller passes the narrowest member's width in d->w: its
  // d->oh / d->ow still describe the first member)
  p.H = d->h; p.W = d->w; p.pitch = d->w;
  p.OH = (d->h + 2 * d->pad - d->dil * (d->r - 1) - 1) / d->stride + 1;
  p.OW = (d->w + 2 * d->pad - d->dil * (d->s - 1) - 1) / d->stride + 1;
  p.Cout = d->cout;
  p.M = d->n * d->oh * d->ow;   // rows of the launch (grouped: all members' rows, as the tile heuristic sees them)
  p.ktiles = d->r * d->s * d->cin / BK;
  p.groups = 1;
  p.rs_ok = 1;
  p.gn_partial = nullptr;
  p.vec_epi = d->cout % 8 == 0;
  p.splitk_mode = d->splitk;
  p.split_ws = d->splitk >= 0 ? reinterpret_cast<float*>(16) : nullptr;   // "a workspace is given" (never dereferenced here)
  p.split_ws_bytes = (int64_t)32 << 20;
  TileForm f = tile_form(tile, tile == HN_TILE_128x32 && rs32_preferred(d));
  if (f.bm == 0) return 0;
  p.nblocks = hn::cdiv(p.M, f.bm) * hn::cdiv(p.Cout, f.bn);
  plan_splits(p, f.bm, f.bn);
  if (tile == HN_TILE_128x32 && f.nbuf == 2 && !rs_will_run(p, f.bm, f.bn, f.waves, 2, true)) return 0;
  return rs_will_run(p, f.bm, f.bn, f.waves, f.nbuf, true) ? 1 : 0;
}

static int conv16_run(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                      void* y, float* gn_partial, void* workspace, int64_t workspace_bytes, void* stream,
                      const hn_conv_group* group = nullptr);

extern "C" int hn_conv2d_nhwc_f16x3(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias,
                                    const void* residual, void* y, void* stream) {
  return conv16_run(d, x16, w16, bias, residual, y, nullptr, nullptr, 0, stream);
}

// The same convolution with a caller-provided fp32 workspace, which lets small grids use split-K
// (deterministic: partial tiles are summed in a fixed order by a second kernel).  The workspace is only
// touched between this call's two launches, so one buffer per stream serves every convolution.
extern "C" int hn_conv2d_nhwc_f16x3_ws(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias,
                                       const void* residual, void* y, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
  HN_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace % 16 == 0 && workspace_bytes >= 0), "bad workspace");
  return conv16_run(d, x16, w16, bias, residual, y, nullptr, workspace, workspace_bytes, stream);
}

// The same convolution, additionally emitting the GroupNorm partial sums of its fp32 output from the
// epilogue (fcos.py:232-239: conv -> GroupNorm): gn_partial [ceil(M/32)][Cout/8][4] = per 32-row group and
// 8-channel unit {sum, sumsq of the rows of the group's first image, sum, sumsq of the rows of the next
// image}; hn_groupnorm_finalize_rows32 turns them into the scale / shift tables.  Saves re-reading the
// output (hn_groupnorm_affine_f32's first pass).
extern "C" int hn_conv2d_nhwc_f16x3_gn(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias,
                                       void* y, float* gn_partial, void* stream) {
  HN_CHECK_ARG(gn_partial, "hn_conv2d_nhwc_f16x3_gn: null gn_partial");
  HN_CHECK_ARG(d && d->cout % 8 == 0, "GroupNorm statistics need cout %% 8 == 0");
  return conv16_run(d, x16, w16, bias, nullptr, y, gn_partial, nullptr, 0, stream);
}

// Same-shape convolutions as ONE launch (gridDim.z = count): the cls / reg tower layers of one FPN level, or the
// three A2J head layers, are independent and identical in shape; on their own each leaves CUs idle at small
// batch and costs a launch.  (Side streams do not help on this platform: tools/probes/exp/streams.sh.)
extern "C" int hn_conv2d_nhwc_f16x3_grouped(const hn_conv_desc* d, const hn_conv_group* group, void* stream) {
  HN_CHECK_ARG(d && group, "hn_conv2d_nhwc_f16x3_grouped: null pointer");
  HN_CHECK_ARG(group->count >= 1 && group->count <= HN_CONV_MAX_GROUP, "group count must be 1..%d", HN_CONV_MAX_GROUP);
  HN_CHECK_ARG(d->res_mode == 0, "grouped convolutions take no residual");
  bool any_gn = false, all_gn = true;
  for (int g = 0; g < group->count; ++g) {
    HN_CHECK_ARG(group->x16[g] && group->w16[g] && group->y[g], "group member %d has a null tensor", g);
    any_gn = any_gn || group->gn_partial[g];
    all_gn = all_gn && group->gn_partial[g];
  }
  HN_CHECK_ARG(any_gn == all_gn, "gn_partial must be given for every group member or for none");
  HN_CHECK_ARG(group->gn_units == 0 || group->gn_units >= d->cout / 8, "gn_units smaller than cout/8");
  return conv16_run(d, group->x16[0], group->w16[0], group->bias[0], nullptr, group->y[0], nullptr, nullptr, 0, stream,
                    group);
}

static inline bool gn_of_group_needs_32(const hn_conv_group* group, int g) { return group->gn_partial[g] != nullptr; }

// Argument checks + the parameter block of ONE plain convolution (no group): shared by conv16_run and the heterogeneous
// launch hn_conv2d_nhwc_f16x3_multi, so that a member of a multi launch is set up exactly like the same convolution alone.
static int fill_params16(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                         void* y, float* gn_partial, void* workspace, int64_t workspace_bytes, ConvParams16& p) {
  HN_CHECK_ARG(d && x16 && w16 && y, "hn_conv2d_nhwc_f16x3: null pointer");
  HN_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad tensor dims");
  HN_CHECK_ARG(d->cin % 32 == 0, "f16x3 conv needs cin %% 32 == 0 (got %d); use hn_conv2d_nhwc_f32", d->cin);
  HN_CHECK_ARG(d->r > 0 && d->s > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "bad filter geometry");
  const int oh = (d->h + 2 * d->pad - d->dil * (d->r - 1) - 1) / d->stride + 1;
  const int ow = (d->w + 2 * d->pad - d->dil * (d->s - 1) - 1) / d->stride + 1;
  HN_CHECK_ARG(oh == d->oh && ow == d->ow, "output size mismatch: desc %dx%d, computed %dx%d", d->oh, d->ow, oh, ow);
  HN_CHECK_ARG(d->res_mode >= 0 && d->res_mode <= 2, "bad res_mode %d", d->res_mode);
  HN_CHECK_ARG(d->terms == 0 || d->terms == 1 || d->terms == 3, "terms must be 0 / 3 (f16x3) or 1 (f16x1), got %d", d->terms);
  HN_CHECK_ARG(d->res_mode == 0 || residual, "res_mode set but residual is null");
  HN_CHECK_ARG(d->res_mode != 2 || (d->res_h > 0 && d->res_w > 0), "res_mode 2 needs res_h/res_w");
  HN_CHECK_ARG(!d->in_affine, "f16x3 conv takes pre-split input; apply GroupNorm with hn_affine_split_f32 first");
  HN_CHECK_ARG(!d->out_split || d->cout % 32 == 0, "S32 output needs cout %% 32 == 0 (got %d)", d->cout);
  HN_CHECK_ARG(d->in_pix_stride == 0 || (d->in_pix_stride >= 2 * d->cin && d->in_pix_stride % 64 == 0), "bad in_pix_stride");
  HN_CHECK_ARG(d->out_pix_stride == 0 || d->out_pix_stride >= (d->out_split ? 2 : 1) * d->cout, "bad out_pix_stride");
  HN_CHECK_ARG((int64_t)d->n * d->oh * d->ow < (int64_t)1 << 31, "too many output pixels");
  p.x = (const _Float16*)x16; p.w = (const _Float16*)w16; p.bias = bias; p.res = residual; p.y = y;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Cin = d->cin; p.Cout = d->cout; p.R = d->r; p.S = d->s;
  p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.OH = d->oh; p.OW = d->ow;
  p.M = d->n * d->oh * d->ow;
  p.Ktot = d->r * d->s * d->cin;
  p.ktiles = p.Ktot / BK;
  p.relu_cols = d->relu_cols; p.res_mode = d->res_mode; p.res_h = d->res_h; p.res_w = d->res_w;
  p.out_split = d->out_split; p.res_split = d->res_split;
  p.xs = d->in_pix_stride ? d->in_pix_stride : 2 * d->cin;
  p.pitch = d->w;
  p.lo_off = 32;
  p.ys = d->out_pix_stride ? d->out_pix_stride : (d->out_split ? 2 : 1) * d->cout;
  p.rs = d->res_pix_stride ? d->res_pix_stride : (d->res_split ? 2 : 1) * d->cout;
  p.vec_epi = (d->cout % 8 == 0) && (p.out_split || p.ys % 4 == 0) &&
              (p.res_mode == 0 || p.res_split || p.rs % 4 == 0) &&
              ((uintptr_t)y % 16 == 0) && (bias == nullptr || (uintptr_t)bias % 16 == 0) &&
              (residual == nullptr || (uintptr_t)residual % 16 == 0);
  p.tiles_m = p.tiles_n = p.nblocks = 0;
  p.small_mask = 0;
  p.range_flag = hn::range_flag_ptr();
  p.gn_partial = gn_partial;
  p.split_ws = (float*)workspace;
  p.split_ws_bytes = workspace ? workspace_bytes : 0;
  p.splitk_mode = d->splitk;
  p.groups = 1;
  p.rs_ok = 1;
  p.terms = d->terms == 1 ? 1 : 3;
  p.pool_ty = p.pool_tx = p.pool_oh = p.pool_ow = 0;
  p.gn_units = d->cout >> 3;
  return HN_OK;
}

static int conv16_run(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                      void* y, float* gn_partial, void* workspace, int64_t workspace_bytes, void* stream,
                      const hn_conv_group* group) {
  ConvParams16 p;
  HN_TRY16(fill_params16(d, x16, w16, bias, residual, y, gn_partial, workspace, workspace_bytes, p));
  // 64-output-channel 3x3 / stride-1 layers with many tiles (ResNet-34 layer1): direct convolution from an LDS halo patch
  // (conv3x3_halo.hip; same k order, bit-identical results; HN_CONV_NO_HALO=1 keeps them on this kernel)
  if (d->terms != 1 && hn::conv3x3_halo_applies(d, gn_partial != nullptr, group != nullptr, residual) &&
      hn::conv3x3_halo_operands_ok(d, x16, w16, bias, residual, y))
    return hn::conv3x3_halo(d, x16, w16, bias, residual, y, (hipStream_t)stream);
  // short-k 1x1 layers with 256 k output channels on many pixels (the FPN P3 lateral, the A2J 64 -> 256 / 128 -> 512 expansions
  // at batch >= ~32): filter bank in registers, activations streamed once (conv1x1_stream.hip; same k order, bit-identical)
  if (hn::conv1x1_stream_applies(d, gn_partial != nullptr, group != nullptr) &&
      hn::conv1x1_stream_operands_ok(d, x16, w16, bias, residual, y))
    return hn::conv1x1_stream(d, x16, w16, bias, residual, y, (hipStream_t)stream);

  hn_conv_desc tile_desc = *d;  // what the tile heuristic sees: for a group, all members' rows together
  if (group) {
    p.groups = group->count;
    if (group->gn_units > 0) p.gn_units = group->gn_units;
    int64_t total_m = 0;
    for (int g = 0; g < group->count; ++g) {
      const int gh = group->h[g] > 0 ? group->h[g] : d->h, gw = group->w[g] > 0 ? group->w[g] : d->w;
      p.gH[g] = gh; p.gW[g] = gw;
      p.gOH[g] = (gh + 2 * d->pad - d->dil * (d->r - 1) - 1) / d->stride + 1;
      p.gOW[g] = (gw + 2 * d->pad - d->dil * (d->s - 1) - 1) / d->stride + 1;
      HN_CHECK_ARG(p.gOH[g] > 0 && p.gOW[g] > 0, "group member %d has an empty output", g);
      HN_CHECK_ARG((int64_t)d->n * p.gOH[g] * p.gOW[g] < (int64_t)1 << 31, "too many output pixels");
      p.gM[g] = d->n * p.gOH[g] * p.gOW[g];
      total_m += p.gM[g];
      if (gn_of_group_needs_32(group, g)) HN_CHECK_ARG(p.gOH[g] * p.gOW[g] >= 32, "GroupNorm statistics need OH*OW >= 32");
      p.gx[g] = (const _Float16*)group->x16[g]; p.gw[g] = (const _Float16*)group->w16[g];
      p.gbias[g] = group->bias[g]; p.gy[g] = group->y[g]; p.ggn[g] = group->gn_partial[g];
      p.vec_epi = p.vec_epi && ((uintptr_t)group->y[g] % 16 == 0) &&
                  (group->bias[g] == nullptr || (uintptr_t)group->bias[g] % 16 == 0);
    }
    p.gn_partial = group->gn_partial[0];
    p.split_ws = nullptr;
    if (group->gn_partial[0])
      HN_CHECK_ARG(p.vec_epi && !d->out_split && d->relu_cols == 0,
                   "GroupNorm statistics need the vector epilogue, fp32 output and no ReLU");
    tile_desc.n = 1;
    tile_desc.oh = (int)(total_m < ((int64_t)1 << 30) ? total_m : ((int64_t)1 << 30));
    tile_desc.ow = 1;
  }
  p.splits = 1;
  p.kt_per = p.ktiles;
  if (gn_partial) {
    HN_CHECK_ARG(p.vec_epi && !d->out_split && d->res_mode == 0 && d->relu_cols == 0,
                 "GroupNorm statistics need the vector epilogue, fp32 output, no residual and no ReLU");
    HN_CHECK_ARG(d->oh * d->ow >= 32, "GroupNorm statistics in the epilogue need OH*OW >= 32");
  }
  hipStream_t st = (hipStream_t)stream;
  switch (hn_conv2d_f16x3_pick_tile(&tile_desc)) {
    case HN_TILE_128x128: return launch16<128, 128, 2, 2, 2>(p, st);
    // LDS stage counts from an in-pipeline sweep (round-1 commit 5dc48dd): extra stages only pay
    // where they do not cost occupancy
    case HN_TILE_128x64: return launch16<128, 64, 2, 2, 2>(p, st);
    case HN_TILE_64x64: return launch16<64, 64, 2, 2, 3>(p, st);
    case HN_TILE_128x32: {
      // few output columns: the A operand is nearly all of the traffic, so the row-shared form (2 stages) is preferred --
      // but only when it will really run (no split-K, rows long enough); otherwise the tuned 3-stage per-tap form
      bool rs2 = rs32_preferred(d);
      if (rs2) {
        ConvParams16 q = p;
        q.nblocks = hn::cdiv(q.M, 128) * hn::cdiv(q.Cout, 32);
        if (group) {
          q.nblocks = 0;
          for (int g = 0; g < q.groups; ++g) q.nblocks += hn::cdiv(q.gM[g], 128) * hn::cdiv(q.Cout, 32);
        }
        plan_splits(q, 128, 32);
        rs2 = rs_will_run(q, 128, 32, 4, 2, true);
      }
      if (rs2) return launch16<128, 32, 4, 1, 2>(p, st);
      return launch16<128, 32, 4, 1, 3>(p, st);
    }
    case HN_TILE_64x128: return launch16<64, 128, 2, 2, 3>(p, st);
    case HN_TILE_256x128: return launch16<256, 128, 2, 2, 2, false>(p, st);
    case HN_TILE_32x64: return launch16<32, 64, 1, 2, 4>(p, st);
    // (the row-shared A form does not fit two 256-row workgroups on a CU, and 128x64 with it -- 442 us on ResNet-34 layer1 --
    // loses to 256x64 without: 421 us)
    case HN_TILE_256x64: return launch16<256, 64, 4, 1, 2>(p, st);
    case HN_TILE_256x128_W8: return launch16<256, 128, 4, 2, 2, false>(p, st);
    case HN_TILE_256x64_W8: return launch16<256, 64, 4, 2, 2, false>(p, st);
    case HN_TILE_128x256_W8: return launch16<128, 256, 2, 4, 2, false>(p, st);
    default: return hn::fail(HN_ERR_ARG, "unknown tile id %d", d->tile);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Heterogeneous launch (include/handnet_hip.h: hn_conv2d_nhwc_f16x3_multi)
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NBUF>
static int launch_multi(MultiParams16& mp, hipStream_t st) {
  int total = 0, any_split = 0, red_grid = 1;
  for (int g = 0; g < mp.count; ++g) {
    ConvParams16& p = mp.m[g];
    mp.start[g] = total;
    total += p.nblocks * p.splits;
    if (p.splits > 1) {
      any_split = 1;
      const long units = (long)p.M * (p.Cout >> 3);
      const int grid = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
      red_grid = red_grid > grid ? red_grid : grid;
    }
  }
  for (int g = mp.count; g <= kMultiMax; ++g) mp.start[g] = total;
  hipLaunchKernelGGL((conv_igemm_f16x3_multi_kernel<BM, BN, WM, WN, NBUF>), dim3(total), dim3(WM * WN * 64), 0, st, mp);
  HN_CHECK_LAUNCH("conv_igemm_f16x3_multi_kernel");
  if (any_split) {
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(red_grid, mp.count), dim3(256), 0, st, mp);
    HN_CHECK_LAUNCH("splitk_reduce_multi_kernel");
  }
  return HN_OK;
}

// Would these convolutions run as ONE multi launch?  They do when every member, launched alone through
// hn_conv2d_nhwc_f16x3_ws, would take the implicit-GEMM kernel of one and the same tile form in its descriptor form; the
// members then keep their own split-K plans (the plan a member would get alone: its result does not depend on the grouping).
static bool multi_plan(const hn_conv_multi* mm, void* workspace, int64_t workspace_bytes, MultiParams16& mp, int& tile) {
  tile = -1;
  int64_t ws_off = 0;
  for (int g = 0; g < mm->count; ++g) {
    const hn_conv_desc* d = &mm->desc[g];
    ConvParams16& p = mp.m[g];
    if (fill_params16(d, mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g], nullptr, workspace, workspace_bytes, p) !=
        HN_OK)
      return false;
    if (p.terms != 3 || (hn::conv3x3_halo_applies(d, false, false, mm->residual[g]) &&
                         hn::conv3x3_halo_operands_ok(d, mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g])))
      return false;
    if (hn::conv1x1_stream_applies(d, false, false) &&
        hn::conv1x1_stream_operands_ok(d, mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g]))
      return false;
    const int t = hn_conv2d_f16x3_pick_tile(d);
    if (tile >= 0 && t != tile) return false;
    tile = t;
    TileForm f = tile_form(t, false);
    if (t == HN_TILE_128x32 || f.bm == 0) return false;   // (its stage count depends on the row-shared form: not worth a table)
    if (!finish_params16(p)) return false;
    p.tiles_m = hn::cdiv(p.M, f.bm);
    p.tiles_n = hn::cdiv(p.Cout, f.bn);
    p.nblocks = p.tiles_m * p.tiles_n;
    plan_splits(p, f.bm, f.bn);
    p.rs_ok = 0;
    if (p.splits > 1) {   // the member's partial planes: its own slice of the workspace
      const int64_t bytes = (int64_t)p.splits * p.M * p.Cout * 4;
      if (ws_off + bytes > workspace_bytes) return false;
      p.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ws_off);
      ws_off += (bytes + 255) & ~(int64_t)255;
    }
  }
  return true;
}

extern "C" int hn_conv2d_nhwc_f16x3_multi(const hn_conv_multi* mm, void* workspace, int64_t workspace_bytes, void* stream) {
  HN_CHECK_ARG(mm && mm->count >= 1 && mm->count <= HN_CONV_MULTI_MAX, "hn_conv2d_nhwc_f16x3_multi: count must be 1..%d",
               HN_CONV_MULTI_MAX);
  HN_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace % 16 == 0 && workspace_bytes >= 0), "bad workspace");
  for (int g = 0; g < mm->count; ++g)
    for (int h = 0; h < mm->count; ++h)
      HN_CHECK_ARG(g == h || mm->y[g] != mm->y[h], "members %d and %d write the same output", g, h);
  MultiParams16 mp;
  mp.count = mm->count;
  int tile = -1;
  const bool together = mm->count > 1 && !hn::env_flags().no_multi && multi_plan(mm, workspace, workspace ? workspace_bytes : 0, mp, tile);
  if (!together) {   // one after the other: the same results by the members' own launches
    for (int g = 0; g < mm->count; ++g)
      HN_TRY16(conv16_run(&mm->desc[g], mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g], nullptr, workspace,
                          workspace_bytes, stream));
    return HN_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (tile) {
    case HN_TILE_128x128: return launch_multi<128, 128, 2, 2, 2>(mp, st);
    case HN_TILE_128x64: return launch_multi<128, 64, 2, 2, 2>(mp, st);
    case HN_TILE_64x64: return launch_multi<64, 64, 2, 2, 3>(mp, st);
    case HN_TILE_64x128: return launch_multi<64, 128, 2, 2, 3>(mp, st);
    case HN_TILE_32x64: return launch_multi<32, 64, 1, 2, 4>(mp, st);
    default: break;
  }
  for (int g = 0; g < mm->count; ++g)
    HN_TRY16(conv16_run(&mm->desc[g], mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g], nullptr, workspace,
                        workspace_bytes, stream));
  return HN_OK;
}

// 1 when hn_conv2d_nhwc_f16x3_multi would run these members as one launch (host-only; tests / planning)
extern "C" int hn_conv2d_f16x3_multi_fuses(const hn_conv_multi* mm, int64_t workspace_bytes) {
  if (!mm || mm->count < 2 || mm->count > HN_CONV_MULTI_MAX || hn::env_flags().no_multi) return 0;
  MultiParams16 mp;
  mp.count = mm->count;
  int tile = -1;
  if (!multi_plan(mm, reinterpret_cast<void*>(256), workspace_bytes, mp, tile)) return 0;
  return tile == HN_TILE_128x128 || tile == HN_TILE_128x64 || tile == HN_TILE_64x64 || tile == HN_TILE_64x128 || tile == HN_TILE_32x64;
}

static int stem16_run(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout, const void* w16,
                      const float* bias, int relu, void* y, int out_split, bool pool, void* stream, int terms = 3) {
  HN_CHECK_ARG(x16 && w16 && y, "hn_conv_stem_f16x3: null pointer");
  HN_CHECK_ARG(n > 0 && ph > 0 && pw > 0 && cout > 0 && stride > 0, "bad dims");
  HN_CHECK_ARG(r >= 1 && r <= 8 && pad == r / 2, "stem filter must be R x R with R <= 8 and pad = R/2");
  HN_CHECK_ARG(!out_split || cout % 32 == 0, "S32 output needs cout %% 32 == 0 (got %d)", cout);
  const int hb = ph + 2 * pad, wb = pw + 2 * pad;
  const int oh = (hb - r) / stride + 1, ow = (wb - r) / stride + 1;
  HN_CHECK_ARG((ow - 1) * stride + 8 <= wb, "the 8-pixel k run of the last output column leaves the bordered row");
  HN_CHECK_ARG((int64_t)n * oh * ow < (int64_t)1 << 31, "too many output pixels");
  HN_CHECK_ARG((uintptr_t)x16 % 16 == 0 && (stride * 4 * 2) % 16 == 0 && (wb * 4 * 2) % 16 == 0,
               "LDS-DMA needs 16-byte aligned rows (stride and bordered width must be even)");
  ConvParams16 p;
  p.x = (const _Float16*)x16; p.w = (const _Float16*)w16; p.bias = bias; p.res = nullptr; p.y = y;
  p.N = n; p.H = hb; p.W = wb; p.Cin = 32; p.Cout = cout; p.R = r; p.S = 1;
  p.stride = stride; p.pad = 0; p.dil = 1; p.OH = oh; p.OW = ow;
  p.M = n * oh * ow;
  p.Ktot = r * 32;
  p.ktiles = r;
  p.relu_cols = relu ? cout : 0; p.res_mode = 0; p.res_h = p.res_w = 0;
  p.out_split = out_split; p.res_split = 0;
  p.xs = 4; p.pitch = wb; p.lo_off = (long)n * hb * wb * 4;
  p.gn_partial = nullptr;
  p.split_ws = nullptr; p.split_ws_bytes = 0; p.splits = 1; p.kt_per = p.ktiles; p.splitk_mode = -1; p.groups = 1; p.gn_units = cout >> 3;
  p.ys = (out_split ? 2 : 1) * cout;
  p.rs = 0;
  p.vec_epi = (cout % 8 == 0) && ((uintptr_t)y % 16 == 0) && (bias == nullptr || (uintptr_t)bias % 16 == 0);
  p.tiles_m = p.tiles_n = p.nblocks = 0;
  p.small_mask = 0;
  p.rs_ok = 0;
  p.terms = terms == 1 ? 1 : 3;
  p.pool_ty = p.pool_tx = p.pool_oh = p.pool_ow = 0;
  p.range_flag = hn::range_flag_ptr();
  hipStream_t st = (hipStream_t)stream;
  if (pool) {
    HN_CHECK_ARG(cout == 64 && relu && out_split && bias, "the fused stem + max-pool kernel needs cout = 64, bias, ReLU and an S32 output");
    HN_CHECK_ARG((uintptr_t)y % 16 == 0 && (uintptr_t)bias % 16 == 0, "unaligned output / bias");
    // the ResNet stem shape: direct convolution from an LDS-resident image patch (conv_stem_direct.hip); other filter sizes,
    // and HN_STEM_POOL_GENERIC=1, take the implicit-GEMM form below (same results bit for bit)
    if (r == 7 && stride == 2 && pad == 3 && !hn::env_flags().stem_generic && terms != 1 && (uintptr_t)w16 % 16 == 0)
      return hn::stem_pool_direct(x16, n, ph, pw, w16, bias, y, st);
    p.pool_oh = (oh + 2 - 3) / 2 + 1;
    p.pool_ow = (ow + 2 - 3) / 2 + 1;
    p.pool_ty = hn::cdiv(p.pool_oh, kPoolPR);
    p.pool_tx = hn::cdiv(p.pool_ow, kPoolPC);
    HN_CHECK_ARG((int64_t)n * p.pool_ty * p.pool_tx < (int64_t)1 << 31, "too many patches");
    return launch16<256, 64, 4, 1, 2>(p, st);
  }
  if (cout <= 32) return launch16<128, 32, 4, 1, 3>(p, st);
  // 128x64 measured against 256x64 (+2 %) and 64x64 / 3 stages (+21 %) on the 800x1088 canvas (tools/probes/exp/stem.py)
  if (cout <= 64) return launch16<128, 64, 2, 2, 2>(p, st);
  return launch16<128, 128, 2, 2, 2>(p, st);
}

// Stem convolution (R x R, 4-channel pixels, R <= 8) on the f16x3 kernel.  The image is stored as two fp16
// planes (hi, lo) of [n][ph + 2*pad][pw + 2*pad][4] with a physically zero border, so the R*4 <= 32
// values one filter ROW touches are 64 contiguous bytes per plane: filter row ky is one 32-deep k tile
// (k = kx*4 + c, zero weights for k >= R*4), the im2col row of output pixel (oy, ox) starts at bordered
// pixel (oy*stride + ky, ox*stride), and no tap is ever out of bounds.  Same kernel, same DMA path.
extern "C" int hn_conv_stem_f16x3(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                                  const void* w16, const float* bias, int relu, void* y, int out_split, void* stream) {
  return stem16_run(x16, n, ph, pw, pad, r, stride, cout, w16, bias, relu, y, out_split, false, stream);
}

// The same stem with the 3x3 / stride-2 / pad-1 max pooling that follows it in a ResNet (conv1 -> bn1 -> relu -> maxpool,
// torchvision resnet34 at fcos_utils/fcos.py:737) fused into the epilogue: y is the POOLED S32 map
// [n][(oh+1)/2][(ow+1)/2][64]; bit-identical to hn_conv_stem_f16x3 + hn_maxpool3x3s2_s32.
extern "C" int hn_conv_stem_pool_f16x3(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                                       const void* w16, const float* bias, void* y, void* stream) {
  return stem16_run(x16, n, ph, pw, pad, r, stride, cout, w16, bias, 1, y, 1, true, stream);
}

extern "C" int hn_conv_stem_pool_f16x3_terms(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                                             const void* w16, const float* bias, void* y, int terms, void* stream) {
  HN_CHECK_ARG(terms == 0 || terms == 1 || terms == 3, "terms must be 0 / 3 or 1");
  return stem16_run(x16, n, ph, pw, pad, r, stride, cout, w16, bias, 1, y, 1, true, stream, terms);
}
