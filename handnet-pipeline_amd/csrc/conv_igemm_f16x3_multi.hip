// f16x3 implicit-GEMM convolution: heterogeneous launches (hn_conv2d_nhwc_f16x3_multi) -- planning, launcher, C entry points.
// Device code: conv_igemm_f16x3_kernel.h (conv_igemm_f16x3_multi_kernel, splitk_reduce_multi_kernel).
#define HN_IGEMM_MULTI_TU 1
#include "conv_igemm_f16x3_plan.h"

// ---------------------------------------------------------------------------------------------------------------------
// Heterogeneous launch (include/handnet_hip.h: hn_conv2d_nhwc_f16x3_multi)
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NBUF, int KK = 1>
static int launch_multi(MultiParams16& mp, hipStream_t st) {
  int total = 0, any_split = 0, red_grid = 1;
  for (int g = 0; g < mp.count; ++g) {
    ConvParams16& p = mp.m[g];
    mp.start[g] = total;
    total += p.nblocks * p.splits;
    if (p.splits > 1 && p.ticket_base < 0) {
      any_split = 1;
      const long units = (long)p.M * (p.Cout >> 3);
      const int grid = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
      red_grid = red_grid > grid ? red_grid : grid;
    }
  }
  for (int g = mp.count; g <= kMultiMax; ++g) mp.start[g] = total;
  constexpr int LDS_BYTES = KK > 1 ? NBUF * KK * (BM + BN) * ROWH * 2 : 0;   // deep-k members: the ring lives in dynamic LDS
  if constexpr (KK > 1) {
    static bool attr_set[64] = {};
    int dev = 0;
    HN_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      HN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_f16x3_multi_kernel<BM, BN, WM, WN, NBUF, KK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL((conv_igemm_f16x3_multi_kernel<BM, BN, WM, WN, NBUF, KK>), dim3(total), dim3(WM * WN * 64), LDS_BYTES, st, mp);
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
static bool multi_plan(const hn_conv_multi* mm, void* workspace, int64_t workspace_bytes, MultiParams16& mp, int& tile, bool query = false) {
  tile = -1;
  int64_t ws_off = 0;
  int taken = 0;
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
    // members of one PARENT tile share a grid: a deep-k member (HN_TILE_*_K2) and a member of the same tile whose k loop is too
    // short for the deep form on its own run together in the deep-k kernel (it handles any tile count, bit-identically);
    // each member keeps the split-K plan of the form it would take alone
    const int t = hn_conv2d_f16x3_pick_tile(d);
    auto parent = [](int id) { return id == HN_TILE_64x64_K2 ? HN_TILE_64x64 : id; };
    if (tile >= 0 && parent(t) != parent(tile)) return false;
    tile = (tile >= 0 && tile != parent(tile)) ? tile : t;   // once a deep-k member is seen, the launch is the deep-k form
    TileForm f = tile_form(t, false);
    if (t == HN_TILE_128x32 || f.bm == 0) return false;   // (its stage count depends on the row-shared form: not worth a table)
    if (!finish_params16(p)) return false;
    p.tiles_m = hn::cdiv(p.M, f.bm);
    p.tiles_n = hn::cdiv(p.Cout, f.bn);
    p.nblocks = p.tiles_m * p.tiles_n;
    plan_splits(p, f.bm, f.bn, f.kk);
    p.rs_ok = 0;
    if (p.splits > 1) {   // the member's partial planes: its own slice of the workspace
      const int64_t bytes = (int64_t)p.splits * p.M * p.Cout * 4;
      if (ws_off + bytes > workspace_bytes) return false;
      p.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ws_off);
      ws_off += (bytes + 255) & ~(int64_t)255;
      if (!query) assign_tickets(p, f.bm, f.bn, f.waves, workspace, taken);   // the members' counters: consecutive ranges of the workspace's slot
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
      HN_TRY16(hn_igemm_conv16_run(&mm->desc[g], mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g], nullptr, workspace,
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
    case HN_TILE_64x64_K2: return launch_multi<64, 64, 2, 2, 3, 2>(mp, st);
    default: break;
  }
  for (int g = 0; g < mm->count; ++g)
    HN_TRY16(hn_igemm_conv16_run(&mm->desc[g], mm->x16[g], mm->w16[g], mm->bias[g], mm->residual[g], mm->y[g], nullptr, workspace,
                        workspace_bytes, stream));
  return HN_OK;
}

// 1 when hn_conv2d_nhwc_f16x3_multi would run these members as one launch (host-only; tests / planning)
extern "C" int hn_conv2d_f16x3_multi_fuses(const hn_conv_multi* mm, int64_t workspace_bytes) {
  if (!mm || mm->count < 2 || mm->count > HN_CONV_MULTI_MAX || hn::env_flags().no_multi) return 0;
  MultiParams16 mp;
  mp.count = mm->count;
  int tile = -1;
  if (!multi_plan(mm, reinterpret_cast<void*>(256), workspace_bytes, mp, tile, true)) return 0;
  return tile == HN_TILE_128x128 || tile == HN_TILE_128x64 || tile == HN_TILE_64x64 || tile == HN_TILE_64x128 || tile == HN_TILE_32x64 ||
         tile == HN_TILE_64x64_K2;
}

int hn::tickets_nonzero_multi(int64_t* count) { return tickets_nonzero_here(count); }
