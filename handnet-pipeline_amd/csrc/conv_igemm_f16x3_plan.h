// Launch planning of the f16x3 implicit-GEMM convolution (host side): the split-K cost model, the row-shared-A and mixed-tile
// decisions, the descriptor extents / magic numbers, the tile forms behind the HN_TILE_* ids and the argument checks that fill a
// parameter block -- ONE definition of each decision, shared by the launchers (conv_igemm_f16x3.hip), the heterogeneous launch
// (conv_igemm_f16x3_multi.hip) and the host-only queries (hn_conv2d_f16x3_uses_rs, ...).
#pragma once
#include <mutex>
#include "conv_igemm_f16x3_kernel.h"

namespace {

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
static SplitModel split_model(int bm, int bn, int kk = 1) {
  const int cus = 256;
  if (kk > 1) return {1 * cus, 12.0, 0.10};   // the 64x64 deep-k form: one workgroup per CU (96 KB of LDS), a shorter time per k tile
  if (bm == 32 && bn == 64) return {3 * cus, 3.0, 0.27};
  if (bm == 64 && bn == 64) return {3 * cus, 12.0, 0.12};
  if (bm == 64 && bn == 128) return {2 * cus, 16.0, 0.12};
  if (bm == 128 && bn == 64) return {2 * cus, 16.0, 0.20};
  if (bm == 128 && bn == 32) return {2 * cus, 12.0, 0.15};
  return {2 * cus, 20.0, 0.40};   // 128x128 and the 256-row tiles
}
static void plan_splits(ConvParams16& p, int bm, int bn, int kk = 1) {   // needs p.nblocks; sets p.splits / p.kt_per
  p.splits = 1;
  p.kt_per = p.ktiles;
  p.ticket_base = -1;   // (assign_tickets, once the plan stands)
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
    SplitModel sm = split_model(bm, bn, kk);
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

// The ticket slot of a workspace (g_split_tickets in the kernel header): the first kTicketSlots distinct workspace addresses a
// process splits with get one each, for good -- a captured graph keeps the slot number of its launches, so a slot is never
// handed to another address; later workspaces take the separate reduction launch (same results).  -1: none.
static int split_ticket_slot(const void* workspace) {
  static std::mutex mu;
  static const void* keys[kTicketSlots];
  static int used = 0;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < used; ++i)
    if (keys[i] == workspace) return i;
  if (used == kTicketSlots) return -1;
  keys[used] = workspace;
  return used++;
}
// hn_debug_tickets_nonzero: counters of THIS translation unit's g_split_tickets that are not zero (synchronises the device).
// "Zero at rest" is the invariant the in-kernel reduction rests on: a launch that died between drawing and putting back would
// leave a counter behind and the next launch on that slot would never see "last".
static int tickets_nonzero_here(int64_t* count) {
  static int host[kTicketSlots * kTicketsPerSlot];
  HN_CHECK_HIP(hipDeviceSynchronize());
  HN_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_split_tickets), sizeof(host)));
  int64_t c = 0;
  for (int i = 0; i < kTicketSlots * kTicketsPerSlot; ++i) c += host[i] != 0;
  *count += c;
  return HN_OK;
}

// After plan_splits: the split launch reduces in its own last workgroups when it can have counters -- `taken` of the slot's
// counters already belong to earlier members of the same launch (heterogeneous launches; 0 otherwise).
static void assign_tickets(ConvParams16& p, int bm, int bn, int waves, const void* workspace, int& taken) {
  p.ticket_base = -1;
  if (p.splits <= 1 || !fused_reduce_form(bm, bn) || p.terms != 3 || !workspace || hn::env_flags().no_fused_reduce || taken + p.nblocks * waves > kTicketsPerSlot ||
      (int64_t)p.splits * p.M * p.Cout * 4 >= ((int64_t)1 << 31)) return;   // (the planes sit behind one buffer descriptor)
  const int slot = split_ticket_slot(workspace);
  if (slot < 0) return;
  p.ticket_base = slot * kTicketsPerSlot + taken;
  taken += p.nblocks * waves;   // one counter per (tile, wave)
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

int64_t nblocks16(const hn_conv_desc* d, int bm, int bn) {
  const int64_t M = (int64_t)d->n * d->oh * d->ow;
  return (int64_t)hn::cdiv(M, bm) * hn::cdiv(d->cout, bn);
}

// The tile forms behind the HN_TILE_* ids: {BM, BN, waves, LDS stages}; conv16_run's switch instantiates exactly these.
struct TileForm { int bm, bn, waves, nbuf, kk; };   // kk: k tiles per ring stage (1 = the pinned loop)
static bool rs32_preferred(const hn_conv_desc* d) {
  return d->r == 3 && d->s == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && !hn::env_flags().no_rs && !hn::env_flags().no_rs32;
}
static TileForm tile_form(int tile, bool rs32) {
  switch (tile) {
    case HN_TILE_128x128: return {128, 128, 4, 2, 1};
    case HN_TILE_128x64: return {128, 64, 4, 2, 1};
    case HN_TILE_64x64: return {64, 64, 4, 3, 1};
    case HN_TILE_128x32: return {128, 32, 4, rs32 ? 2 : 3, 1};   // 2 stages only when the row-shared form will run
    case HN_TILE_64x128: return {64, 128, 4, 3, 1};
    case HN_TILE_32x64: return {32, 64, 2, 4, 1};
    case HN_TILE_256x64: return {256, 64, 4, 2, 1};
    case HN_TILE_64x64_K2: return {64, 64, 4, 3, 2};
    default: return {0, 0, 0, 0, 1};
  }
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
  p.gn_units = d->cout >> 3;
  return HN_OK;
}

}  // namespace

// one plain or grouped convolution through the tile switch (conv_igemm_f16x3.hip); also the members of a multi launch that
// cannot share a grid
int hn_igemm_conv16_run(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                        float* gn_partial, void* workspace, int64_t workspace_bytes, void* stream,
                        const hn_conv_group* group = nullptr);
