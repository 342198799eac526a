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
// (round 5: this header holds the DEVICE side -- parameter blocks, the k loop and the epilogue, the kernel templates; launch
// planning is conv_igemm_f16x3_plan.h, the launchers and C entry points conv_igemm_f16x3.hip and conv_igemm_f16x3_multi.hip.
// bench.py / tools stamp the HBM-traffic and PMC files with the hash of THIS file.)
#pragma once
#include "hn_common.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// 256 bytes of zeros: source of every out-of-image tap
__device__ __attribute__((aligned(256))) _Float16 g_zero_page16[128];

// Split-K tickets (round 5): one counter per (output tile, wave) of a split launch whose LAST workgroup does the reduction itself
// (ConvParams16::ticket_base >= 0).  Zero at rest: the reducing workgroup puts its tile's counter back.  A launch owns the
// kTicketsPerSlot counters of the slot its workspace is registered under (split_ticket_slot in the plan header): two launches
// that are in flight together use different workspaces (they would race on the partial planes otherwise), hence different
// counters.  (One array per translation unit that includes this header: launches of both use the same slot numbers on their own
// arrays, which is fine for the same reason.)
constexpr int kTicketSlots = 128, kTicketsPerSlot = 4096;
// tile forms whose kernels hold the in-kernel reduction: the 32- and 64-row tiles, i.e. every split layer of the batch-1 frame.
// (The 128x128 / 256x64 forms sit at 128 + 128 registers and spill VGPRs with it, the 128x64 form two SGPRs; they split at
// batch 32 only, where a reduction launch is 1 % of the layer.)
constexpr bool fused_reduce_form(int bm, int bn) { return bm <= 64 && bn <= 128; }
__device__ int g_split_tickets[kTicketSlots * kTicketsPerSlot];

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
  int ticket_base;    // >= 0: the last workgroup of a tile reduces (index of the launch's first counter in g_split_tickets); -1: splitk_reduce_kernel does
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

#ifdef HN_IGEMM_MAIN_TU   // kernels that are not templates are compiled in the translation unit that launches them
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams16 p) { splitk_reduce_body(p); }
#endif

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
// TERMS = 3: the split-precision product (lo*hi + hi*lo + hi*hi, fp32-grade).  TERMS = 1 ("f16x1", the THROUGHPUT mode SURVEY D6
// plans beside the parity mode; never the default): only hi*hi is issued -- one MFMA per MAC on plain fp16 operands, identical
// data movement -- so that "what does the 1e-3 contract cost" has a measured answer (bench.py --precision f16x1).
// The kernel's body as a device function of (parameter block, workgroup coordinates): conv_igemm_f16x3_kernel passes its own
// kernel argument and blockIdx; conv_igemm_f16x3_multi_kernel (heterogeneous launches, below) the member's block and the
// member-local coordinates.  Always inlined: the single-problem kernel compiles to what it was.
// KK > 1 ("deep k", round 5; the small grids of batch 1): a stage of the ring holds KK consecutive k tiles and the loop has ONE
// barrier per stage instead of one per tile.  A lone 4-wave workgroup per CU spends ~500 cycles per 32-deep step for ~190
// cycles of MFMA issue (profiles/NOTEBOOK.md, round 4 stamps): the rest is the per-step rendezvous -- counted DMA wait, LDS read
// latency in front of the barrier, the barrier itself -- which no number of stages in flight shortens, but fewer rendezvous
// per MFMA do.  The deep loop is compiler-scheduled (plain LDS reads and builtin MFMAs between the barriers: both tiles of a
// stage have landed, so the next tile's fragment reads overlap this tile's MFMAs without any hand-counted wait); the DMA
// geometry, the k order, the term order and the epilogue are the pinned loop's: results are BIT-IDENTICAL to KK = 1.
template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, bool RS = false, int TERMS = 3, bool DYN = false, int KK = 1>
__device__ __forceinline__ void conv_igemm_f16x3_body(const ConvParams16& p, const int blk_x, const int blk_y, const int blk_z,
                                                      const unsigned karg_off = 0 /* byte offset of `p` in the kernel-argument segment */) {
  static_assert(TERMS == 3 || TERMS == 1, "three terms (fp32-grade) or the hi*hi term alone");
  static_assert(NBUF >= 2 && NBUF <= 6, "2..6 LDS stages");
  static_assert(!RS || (BUF && NBUF == 2), "row-shared A needs the descriptor form and the 2-stage pipeline");
  static_assert(KK == 1 || (BUF && !RS && DYN && TERMS == 3), "the deep-k loop: descriptor form, per-tap, dynamic LDS, three terms");
  constexpr int NSLOT = NBUF * KK;   // k-tile slots in LDS (KK per stage)
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
  __shared__ __attribute__((aligned(1024))) _Float16 smem_static[(RS || DYN) ? 8 : NSLOT * (A_BUF + B_BUF)];
  _Float16* smem = (RS || DYN) ? smem_dyn : smem_static;
  _Float16* As = smem;                 // [stage][A_ROWS][64]
  _Float16* Bs = smem + NSLOT * A_BUF;  // [slot][BN][64]

  int lid;
  {
    const int bid = blk_x, nb = o.nblocks;
    const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
  }
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
      // (deep-k form: with its scalar registers used up the compiler keeps the walk's counters in VECTOR registers and would
      // wrap every DMA in a waterfall loop over the "divergent" offset; the value is wave-uniform by construction)
      const int so = KK > 1 ? __builtin_amdgcn_readfirstlane((int)uoff) : (int)uoff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst, 16, (int)voff, so, 0, 0);
    } else {
      const bool ok = (unsigned)(a_ih0[it] + dr) < (unsigned)o.H && (unsigned)(a_iw0[it] + ds) < (unsigned)o.W;
      const _Float16* src = ok ? a_row[it] + uoff : g_zero_page16 + a_cc[it];  // padding taps read zeros
      __builtin_amdgcn_global_load_lds((gbl_void*)src, dst, 16, 0, 0);
    }
  };
  auto dma_b_piece = [&](int it, _Float16* Bd, long boff) {
    lds_void* dst = (lds_void*)(Bd + (it * ROWS_PASS + grp_row) * ROWH);
    if constexpr (BUF)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, dst, 16, (int)b_off[it],
                                               KK > 1 ? __builtin_amdgcn_readfirstlane((int)boff) : (int)boff, 0, 0);
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
  } else if constexpr (KK > 1) {
    // ---- deep-k loop: stage g = slots g*KK .. g*KK+KK-1; stages 0 .. NBUF-2 are put in flight, then per stage:
    //   counted wait (the NBUF-2 younger stages stay in flight) -> barrier (stage st has landed for every wave, and nobody reads
    //   stage st-1 any more) -> refill stage st-1's slots with the tiles of stage st+NBUF-1 -> the MFMAs of stage st's tiles.
    // Past the end of k the loader re-loads the last tile (saturating cursor) into slots whose MFMAs are skipped.
    static_assert((NBUF - 2) * KK * (A_IT + B_IT) <= 63, "vmcnt is a 6-bit counter");
    for (int s0 = 0; s0 < (NBUF - 1) * KK; ++s0) dma_tile(s0);
    const int nstages = (T + KK - 1) / KK;
    int grp = 0;
    // Order inside a stage (round 5, second half; PMC of the first version: the matrix pipe busy 22 % of the time, 4 scalar
    // instructions per MFMA): the fragment reads of the stage's FIRST tile go out right after the barrier, the refill's address
    // arithmetic + DMA issue (~75 scalar instructions that used to sit between the barrier and the first MFMA with the matrix
    // pipe idle) runs under their latency, the second tile's reads follow, and only then the MFMAs -- tile 0's with tile 1's
    // fragments already on their way.  Same tiles, same k order, same term order per accumulator: bit-identical.
    static_assert(KK == 2, "the stage schedule below is written for two tiles per stage");
    auto read_frags = [&](int slot, f16x8 (&fah)[TM], f16x8 (&fal)[TM], f16x8 (&fbh)[TN], f16x8 (&fbl)[TN]) {
      const _Float16* Ab = As + slot * A_BUF;
      const _Float16* Bb = Bs + slot * B_BUF;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fah[i] = *reinterpret_cast<const f16x8*>(Ab + a_rd[i][0]);
        fal[i] = *reinterpret_cast<const f16x8*>(Ab + a_rd[i][1]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fbh[j] = *reinterpret_cast<const f16x8*>(Bb + b_rd[j][0]);
        fbl[j] = *reinterpret_cast<const f16x8*>(Bb + b_rd[j][1]);
      }
    };
    auto mfma_tile = [&](const f16x8 (&fah)[TM], const f16x8 (&fal)[TM], const f16x8 (&fbh)[TN], const f16x8 (&fbl)[TN]) {
      // the pinned loop's term order per accumulator: lo*hi, hi*lo, hi*hi (W fragment = srcA)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fal[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[j], fah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fah[i], acc[i][j], 0, 0, 0);
    };
    (void)nstages;
    const int full = T / KK;   // stages with both tiles: ONE basic block each (no branch between the two tiles' MFMAs)
    for (int st = 0; st < full; ++st) {
      // lgkmcnt(0): this wave's fragment reads of the stage consumed last (compiler-scheduled ds_reads) have RETURNED before
      // it passes the barrier behind which that stage is refilled -- stated, not left to the MFMAs' implicit waits
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NBUF - 2) * KK * DPT) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      f16x8 a0h[TM], a0l[TM], b0h[TN], b0l[TN], a1h[TM], a1l[TM], b1h[TN], b1l[TN];
      read_frags(grp * KK, a0h, a0l, b0h, b0l);
      const int refill = grp == 0 ? NBUF - 1 : grp - 1;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) dma_tile(refill * KK + kk);
      read_frags(grp * KK + 1, a1h, a1l, b1h, b1l);
      mfma_tile(a0h, a0l, b0h, b0l);
      mfma_tile(a1h, a1l, b1h, b1l);
      grp = grp + 1 == NBUF ? 0 : grp + 1;
    }
    if (T & 1) {   // the ragged last stage: one tile, nothing left to refill
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * KK * DPT) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      f16x8 a0h[TM], a0l[TM], b0h[TN], b0l[TN];
      read_frags(grp * KK, a0h, a0l, b0h, b0l);
      mfma_tile(a0h, a0l, b0h, b0l);
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
  // split-K whose last workgroup reduces (below): the partial planes are addressed through a buffer descriptor with the
  // device-coherent cache policy
  constexpr bool FUSED_REDUCE = !RS && TERMS == 3 && fused_reduce_form(BM, BN);
  const bool tickets = FUSED_REDUCE && p.splits > 1 && p.ticket_base >= 0;
  __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.split_ws, 0, 0, 0x00020000);
  int plane_off = 0;
  if constexpr (FUSED_REDUCE) {
    if (tickets) {
      const int plane_bytes = o.M * p.Cout * 4;   // (splits * plane < 2 GB: assign_tickets)
      rsrc_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.split_ws, 0, plane_bytes * p.splits, 0x00020000);
      plane_off = blk_y * plane_bytes;
    }
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
          if constexpr (FUSED_REDUCE) {
            if (tickets) {   // (wave-uniform) the raw partial tile, written THROUGH this XCD's L2: see the reduction below
              const u32x4 d0 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
              const u32x4 d1 = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
              const int voff = (m * p.Cout + n) * 4;
              __builtin_amdgcn_raw_buffer_store_b128(d0, rsrc_ws, voff, plane_off, 16 /* sc1 */);
              __builtin_amdgcn_raw_buffer_store_b128(d1, rsrc_ws, voff + 16, plane_off, 16);
              continue;
            }
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
    // Split-K without the second launch (round 5; batch 1 spends 33 launches of ~5 us each on reductions that move a few hundred
    // KB): every wave takes a ticket once its part of the partial tile has been written; the one that draws the last ticket adds
    // the planes IN z ORDER -- the arithmetic of splitk_reduce_body, whichever workgroup arrives last: bit-identical to the
    // separate reduction -- and finishes its part of the tile.  The XCDs' L2s are not coherent with each other, and a device-scope fence
    // (write back this L2, invalidate it) costs ~30 us per launch (measured: the batch-1 frame 2.23 -> 3.00 ms): the partial tiles
    // are written and read with the device-coherent cache policy (sc1) instead, so the only ordering needed is "my stores have
    // been acknowledged" (vmcnt(0)) before the ticket is drawn.  (The row-shared form never splits; the f16x1 throughput mode
    // and the 128-row tiles keep the separate launch.)
    if constexpr (FUSED_REDUCE) {
      if (tickets) {
        // The real epilogue's parameters are read from the kernel-argument segment HERE (under the wait for the stores),
        // through an address the compiler cannot identify with the one it loaded `p` from: kept live from the prologue they
        // cost every split-capable kernel ~11 SGPRs it does not have (spills in the 128x64 form).
        typedef __attribute__((address_space(4))) const unsigned KW;
        KW* kw = (KW*)((__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr() + karg_off);
        asm volatile("" : "+s"(kw));
        ConvParams16 f;
        {
          unsigned* dst = reinterpret_cast<unsigned*>(&f);
#pragma unroll
          for (int w = 0; w < (int)(sizeof(ConvParams16) / 4); ++w) dst[w] = kw[w];
        }
        // One ticket per (tile, wave): wave w of every workgroup of a tile owns the same rows and columns of it, so the waves
        // need no rendezvous with their siblings -- each draws on its own counter as soon as ITS stores are in memory.
        int* ticket = g_split_tickets + p.ticket_base + lid * (WM * WN) + __builtin_amdgcn_readfirstlane(wave);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of the partial tile is in memory
        int drawn = 0;
        if (lane == 0) drawn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drawn = __builtin_amdgcn_readfirstlane(drawn);
        if (drawn != f.splits - 1) return;
        if (lane == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero at rest
        const int plane_bytes = f.M * f.Cout * 4;
        const int f_ohow = f.OH * f.OW;
        // all of this lane's 8-channel units at once, the planes ZC at a time: the loads of a chunk are in flight together
        // (a rolled loop waits a memory latency per plane); each element still adds its planes in z order
        constexpr int NU = TM * NP, ZC = NU <= 2 ? 4 : 2;
        float v[NU][8];
        int voff[NU];
        bool live[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int m = m0 + wm * (BM / WM) + (u / NP) * 16 + px, n = n_wave + (u % NP) * 32 + nsub;
          live[u] = m < f.M && n < f.Cout;
          voff[u] = live[u] ? (m * f.Cout + n) * 4 : 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
        }
        for (int z0 = 0; z0 < f.splits; z0 += ZC) {
          u32x4 ld[ZC][NU][2];
#pragma unroll
          for (int zz = 0; zz < ZC; ++zz) {
            const int z = z0 + zz < f.splits ? z0 + zz : f.splits - 1;   // (a clamped plane is read and not added)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              ld[zz][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, voff[u], z * plane_bytes, 16 /* sc1 */);
              ld[zz][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, voff[u] + 16, z * plane_bytes, 16);
            }
          }
#pragma unroll
          for (int zz = 0; zz < ZC; ++zz) {
            if (z0 + zz < f.splits) {
#pragma unroll
              for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[u][e] += __uint_as_float(ld[zz][u][0][e]);
                  v[u][4 + e] += __uint_as_float(ld[zz][u][1][e]);
                }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (!live[u]) continue;
          const int m = m0 + wm * (BM / WM) + (u / NP) * 16 + px, n = n_wave + (u % NP) * 32 + nsub;
          if (f.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(f.bias + n), b1 = *reinterpret_cast<const f32x4*>(f.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[u][e] += b0[e];
              v[u][4 + e] += b1[e];
            }
          }
          epi_finish8(f, m, n, v[u], f_ohow);
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

template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, bool RS = false, int TERMS = 3>
__global__ __launch_bounds__(WM* WN * 64, (BM * BN / (WM * WN) > 64 * 64 ? 1 : 2))
void conv_igemm_f16x3_kernel(const ConvParams16 p) {
  conv_igemm_f16x3_body<BM, BN, WM, WN, NBUF, BUF, RS, TERMS>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// the deep-k forms (KK k tiles per stage, NST stages; dynamic LDS, one workgroup per CU at most two)
template <int BM, int BN, int WM, int WN, int NST, int KK>
__global__ __launch_bounds__(WM* WN * 64, 2)
void conv_igemm_f16x3_deepk_kernel(const ConvParams16 p) {
  conv_igemm_f16x3_body<BM, BN, WM, WN, NST, true, false, 3, true, KK>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

#ifdef HN_IGEMM_MAIN_TU
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
    conv_igemm_f16x3_body<64, 128, 2, 2, 3, true, false, 3, true>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
  else
    conv_igemm_f16x3_body<128, 128, 2, 2, 2, true, true, 3>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

#endif

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

#ifdef HN_IGEMM_MULTI_TU
template <int BM, int BN, int WM, int WN, int NBUF, int KK = 1>
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
  conv_igemm_f16x3_body<BM, BN, WM, WN, NBUF, true, false, 3, (KK > 1), KK>(
      p, local, by, 0, (unsigned)(offsetof(MultiParams16, m) + (size_t)g * sizeof(ConvParams16)));
}

// the reductions of a multi launch's split-K members as ONE launch: gridDim.y = member
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const MultiParams16 mp) {
  ConvParams16 p;
  load_member16(mp, (int)blockIdx.y, p);
  if (p.splits <= 1 || p.ticket_base >= 0) return;   // (members with tickets reduced in their own last workgroups)
  splitk_reduce_body(p);
}
#endif

}  // namespace
