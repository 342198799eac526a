// f16x3 implicit-GEMM convolution: launchers and C entry points (single, grouped, GroupNorm-sum and stem forms).
// Device code: conv_igemm_f16x3_kernel.h; launch planning: conv_igemm_f16x3_plan.h; heterogeneous launches:
// conv_igemm_f16x3_multi.hip.
#define HN_IGEMM_MAIN_TU 1
#include "conv_igemm_f16x3_plan.h"

namespace {

template <int BM, int BN, int WM, int WN, int NBUF, bool BUF, int TERMS = 3>
int launch16_impl(const ConvParams16& p0, hipStream_t st) {
  ConvParams16 p = p0;
  p.tiles_m = hn::cdiv(p.M, BM);
  p.tiles_n = hn::cdiv(p.Cout, BN);
  p.nblocks = p.tiles_m * p.tiles_n;
  plan_splits(p, BM, BN);
  int taken = 0;
  assign_tickets(p, BM, BN, WM * WN, p.split_ws, taken);
  int grid_x = p.nblocks;
  if (p.groups > 1) {
    grid_x = 0;
    for (int g = 0; g < p.groups; ++g) {
      p.gnblocks[g] = hn::cdiv(p.gM[g], BM) * p.tiles_n;
      grid_x = grid_x > p.gnblocks[g] ? grid_x : p.gnblocks[g];
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
        HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, true, true, TERMS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
      }
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, true, true, TERMS>),
                         dim3(grid_x, 1, p.groups > 1 ? p.groups : 1), dim3(WM * WN * 64), LDS_BYTES, st, p);
    }
  }
  if (!rs)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, NBUF, BUF, false, TERMS>),
                       dim3(grid_x, p.splits, p.groups > 1 ? p.groups : 1), dim3(WM * WN * 64), 0, st, p);
  HN_CHECK_LAUNCH("conv_igemm_f16x3_kernel");
  if (p.splits > 1 && p.ticket_base < 0) {
    const long total = (long)p.M * (p.Cout >> 3);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, p);
    HN_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return HN_OK;
}

// deep-k forms (conv_igemm_f16x3_deepk_kernel): three terms, descriptor form only -- anything else takes the parent tile
template <int BM, int BN, int WM, int WN, int NST, int KK>
int launch16_deepk(const ConvParams16& p0, hipStream_t st) {
  ConvParams16 p = p0;
  p.tiles_m = hn::cdiv(p.M, BM);
  p.tiles_n = hn::cdiv(p.Cout, BN);
  p.nblocks = p.tiles_m * p.tiles_n;
  plan_splits(p, BM, BN, KK);
  int taken = 0;
  assign_tickets(p, BM, BN, WM * WN, p.split_ws, taken);
  int grid_x = p.nblocks;
  if (p.groups > 1) {
    grid_x = 0;
    for (int g = 0; g < p.groups; ++g) {
      p.gnblocks[g] = hn::cdiv(p.gM[g], BM) * p.tiles_n;
      grid_x = grid_x > p.gnblocks[g] ? grid_x : p.gnblocks[g];
    }
  }
  constexpr int LDS_BYTES = NST * KK * (BM + BN) * ROWH * 2;
  static_assert(LDS_BYTES <= 160 * 1024, "deep-k ring exceeds the LDS");
  static bool attr_set[64] = {};
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_f16x3_deepk_kernel<BM, BN, WM, WN, NST, KK>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_igemm_f16x3_deepk_kernel<BM, BN, WM, WN, NST, KK>), dim3(grid_x, p.splits, p.groups > 1 ? p.groups : 1),
                     dim3(WM * WN * 64), LDS_BYTES, st, p);
  HN_CHECK_LAUNCH("conv_igemm_f16x3_deepk_kernel");
  if (p.splits > 1 && p.ticket_base < 0) {
    const long total = (long)p.M * (p.Cout >> 3);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, p);
    HN_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return HN_OK;
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

}  // namespace

static int deepk_of(const hn_conv_desc* d, int tile);
static int pick_tile_base(const hn_conv_desc* d);
extern "C" int hn_conv2d_f16x3_pick_tile(const hn_conv_desc* d) {
  if (!d) return HN_TILE_64x64;
  if (d->tile != HN_TILE_AUTO) return d->tile;
  return deepk_of(d, pick_tile_base(d));
}

static int pick_tile_base(const hn_conv_desc* d) {
  // From tools/tile_sweep.py (every distinct conv shape of the pipeline, batch 1 and 32, clocks kept hot):
  // within 0.3 % (batch 32) / 2 % (batch 1) of the best tile per shape.
  if (d->cout <= 32) return nblocks16(d, 128, 32) < 64 ? HN_TILE_32x64 : HN_TILE_128x32;
  const int64_t want = 256;
  // the largest tile wins as soon as it yields one workgroup per CU
  if (d->cout > 64 && nblocks16(d, 128, 128) >= want) return HN_TILE_128x128;
  // ... and a little earlier for very long k loops (round 6, tools/tile_sweep.py at batch 64 = BASELINE config 2): A2J's 2048 ->
  // 512 3 x 3 head layer on 64 crops is 61 x 4 = 244 tiles of 128 x 128 with 576 k tiles each -- 328 us against 360 us on the
  // 484 64 x 128 tiles the next rule picks; no shape of batch 1 or 32 comes near this corner
  if (d->cout > 64 && d->r * d->s * d->cin / 32 >= 288 && nblocks16(d, 128, 128) >= 224) return HN_TILE_128x128;
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

// the deep-k form of the 64x64 tile where the grid leaves every CU at most one workgroup and the k loop is long enough
// (ResNet-34 layer3 at batch 1: 28.9 -> 25.8 us per layer, 2.243 -> 2.215 ms per frame)
static int deepk_of(const hn_conv_desc* d, int tile) {
  if (hn::env_flags().no_deepk || d->terms == 1) return tile;
  const int ktiles = d->r * d->s * d->cin / 32;
  if (ktiles < 8) return tile;
  // (the 64x128 and 32x64 tiles were measured with the same loop and LOSE in the batch-1 frame -- ResNet-34 layer2 22.5 -> 23.2 us,
  // layer4's split-K members 26.5 -> 29.1 us --, and so do deeper or wider rings for this one: profiles/r05_deepk_ab.txt,
  // r05_deepk_variants.txt)
  if (tile == HN_TILE_64x64 && nblocks16(d, 64, 64) <= 256) return HN_TILE_64x64_K2;
  return tile;
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
  plan_splits(p, f.bm, f.bn, f.kk);
  if (tile == HN_TILE_128x32 && f.nbuf == 2 && !rs_will_run(p, f.bm, f.bn, f.waves, 2, true)) return 0;
  return rs_will_run(p, f.bm, f.bn, f.waves, f.nbuf, true) ? 1 : 0;
}


extern "C" int hn_conv2d_nhwc_f16x3(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias,
                                    const void* residual, void* y, void* stream) {
  return hn_igemm_conv16_run(d, x16, w16, bias, residual, y, nullptr, nullptr, 0, stream);
}

// The same convolution with a caller-provided fp32 workspace, which lets small grids use split-K
// (deterministic: partial tiles are summed in a fixed order, by the last workgroup of a tile or by a second kernel -- see
// assign_tickets).  The workspace is only touched between this call's launches, so one buffer per stream serves every convolution.
extern "C" int hn_conv2d_nhwc_f16x3_ws(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias,
                                       const void* residual, void* y, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
  HN_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace % 16 == 0 && workspace_bytes >= 0), "bad workspace");
  return hn_igemm_conv16_run(d, x16, w16, bias, residual, y, nullptr, workspace, workspace_bytes, stream);
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
  return hn_igemm_conv16_run(d, x16, w16, bias, nullptr, y, gn_partial, nullptr, 0, stream);
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
  return hn_igemm_conv16_run(d, group->x16[0], group->w16[0], group->bias[0], nullptr, group->y[0], nullptr, nullptr, 0, stream,
                    group);
}

int hn_igemm_conv16_run(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                        float* gn_partial, void* workspace, int64_t workspace_bytes, void* stream, const hn_conv_group* group) {
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
  int picked = hn_conv2d_f16x3_pick_tile(&tile_desc);
  switch (picked) {
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
    case HN_TILE_32x64: return launch16<32, 64, 1, 2, 4>(p, st);
    // deep-k forms: only the three-term descriptor form exists; f16x1 / oversized operands take the parent tile
    case HN_TILE_64x64_K2: {
      ConvParams16 q = p;
      if (p.terms == 3 && finish_params16(q)) return launch16_deepk<64, 64, 2, 2, 3, 2>(q, st);
      return launch16<64, 64, 2, 2, 3>(p, st);
    }
    // (the row-shared A form does not fit two 256-row workgroups on a CU, and 128x64 with it -- 442 us on ResNet-34 layer1 --
    // loses to 256x64 without: 421 us)
    case HN_TILE_256x64: return launch16<256, 64, 4, 1, 2>(p, st);
    default: return hn::fail(HN_ERR_ARG, "unknown or retired tile id %d", d->tile);
  }
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
  p.range_flag = hn::range_flag_ptr();
  hipStream_t st = (hipStream_t)stream;
  if (pool) {
    HN_CHECK_ARG(cout == 64 && relu && out_split && bias, "the fused stem + max-pool kernel needs cout = 64, bias, ReLU and an S32 output");
    HN_CHECK_ARG((uintptr_t)y % 16 == 0 && (uintptr_t)bias % 16 == 0, "unaligned output / bias");
    // the ResNet stem shape, the only one this path has: direct convolution from an LDS-resident image patch
    // (conv_stem_direct.hip; bit-identical to hn_conv_stem_f16x3 + hn_maxpool3x3s2_s32)
    HN_CHECK_ARG(r == 7 && stride == 2 && pad == 3 && (uintptr_t)w16 % 16 == 0,
                 "the fused stem + max-pool kernel is written for the 7x7 / stride-2 / pad-3 ResNet stem");
    return hn::stem_pool_direct(x16, n, ph, pw, w16, bias, y, terms == 1 ? 1 : 3, st);
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

int hn::tickets_nonzero_main(int64_t* count) { return tickets_nonzero_here(count); }
