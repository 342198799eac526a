/*
 * handnet_hip.h -- C ABI of libhandnet_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the FCOS -> crop -> A2J inference hot path of
 * IRVLUTD/handnet-pipeline.  The reference has NO FFI / plugin layer of its own
 * (its only seam is the Python nn.Module callable handnet_pipeline.HandNet,
 * handnet_pipeline/handnet_pipeline.py:38-116), so each entry point below names the
 * reference Python call site whose arithmetic it replaces.  Host code
 * (handnet-pipeline_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; hn_last_error()
 *    returns a thread-local description of the last failure.
 *  - all pointers are DEVICE pointers borrowed from the caller unless noted; no
 *    function allocates, frees or synchronises; all work is enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = default stream), so calls can be
 *    captured into a hipGraph.
 *  - activations are NHWC fp32; conv weights are [Cout][R][S][Cin] fp32
 *    (Cin % 4 == 0; pad with zeros), already folded with their BatchNorm.
 *  - "fp32" here is exact: f32-input MFMA accumulates as a k-ordered fmaf chain.
 */
#ifndef HANDNET_HIP_H
#define HANDNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HN_OK 0
#define HN_ERR_ARG 1
#define HN_ERR_HIP 2

/* ABI version; bumped whenever a struct below changes. */
#define HN_ABI_VERSION 35
int hn_abi_version(void);
const char* hn_last_error(void);

/* Device facts used by bench.py (no allocation; queries the current device). */
int hn_device_info(int* cu_count, int* clock_khz, char* arch_name, int arch_name_len);
/* PCI bus id ("0000:c1:00.0") of the current device: what a rank of an N > 1 run reports so that N ranks on fewer than N
 * devices cannot pass as N GPUs (bench.py config.devices). */
int hn_device_pci_bus_id(char* out /* host */, int out_len /* >= 16 */);

/* ---- timing helper: HIP events on the caller's stream (bench.py roofline leg) ---- */
int hn_event_create(void** ev);
int hn_event_destroy(void* ev);
int hn_event_record(void* ev, void* stream);
int hn_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */

/* Shader clock UNDER LOAD: one wave compares s_memtime with s_memrealtime for `micros` microseconds and writes
 * the MHz figure to *mhz (device float).  Launch it on a side stream while the kernels of interest run
 * (bench.py roofline leg: roofline.clock_mhz); the sampling loop is bounded by real time. */
int hn_clock_sample(int micros, float* mhz, void* stream);

/* ---- f16x3 range contract ---------------------------------------------------------------------------------
 * A value that is stored in the split format must be finite and |v| <= 65504 (hi = fp16(v) would be +-inf).
 * While hn_range_check_enable(1) is in force (the engines' default; HN_CHECK_RANGE=0 turns it off for A/B timing) every
 * launch of a split PRODUCER (f16x3 conv epilogue / split-K tail, hn_affine_split_f32, hn_fcos_preprocess_split / _list,
 * hn_stem_image_nhwc4) sets a sticky per-device flag word when it meets such a value:
 *   HN_RANGE_ACTIVATION        an intermediate activation (results would be inf / NaN or -- ReLU maps NaN to 0 -- silently
 *                              wrong: this model needs the exact f32 mode)
 *   HN_RANGE_INPUT             a finite INPUT value beyond the range (preprocessed RGB; a depth crop: depth is metres)
 *   HN_RANGE_INPUT_NONFINITE   a NaN / inf input value (invalid pixels of a 32FC1 depth image, which the reference passes
 *                              through, ros_demo.py:227-231; see hn_stem_image_nhwc4_valid)
 * The words live in a block of 4 device int32 (activation, input, input-non-finite, 0): the library's own, or -- after
 * hn_range_check_bind(block), which like the switch is read on the host at LAUNCH time -- the caller's (NULL unbinds).
 * The switch and the bound block are state of the calling HOST THREAD.  An engine brackets the launches of one step with
 * hn_range_scope_begin(block, on) / hn_range_scope_end(): inside the scope this thread's producers note into `block` (on = 0:
 * nowhere), afterwards the thread's previous switch and block are back -- so engines driven from different threads of one
 * process never see, redirect or switch off each other's flags, and scopes nest (at most 8 deep).
 * hn_range_check_collect enqueues ONE tiny kernel on `stream` that copies the words of `block` (NULL: the library's) to
 * dst[0..3] and clears them -- no synchronisation; the host reads dst with the copy of the results it makes anyway (the
 * drop-in HandNet.forward does).  hn_range_check_fetch is the synchronous form: *flag (host)
 * = OR of the HN_RANGE_* bits, optionally clearing them.  Weight banks are checked on the host when they are split
 * (hn_amd.weights.split_f16x3 raises; hn_finalize fails). */
#define HN_RANGE_ACTIVATION 1
#define HN_RANGE_INPUT 2
#define HN_RANGE_INPUT_NONFINITE 4
int hn_range_check_enable(int on);
int hn_range_check_enabled(void);
int hn_range_check_fetch(int* flag /* host */, int reset, void* stream);
int hn_range_check_bind(int32_t* block /* device, 4 words, zeroed by the caller; or NULL */);
int hn_range_scope_begin(int32_t* block /* device, 4 words; NULL with on = 1: the library's block */, int on);
int hn_range_scope_end(void);
int hn_range_check_collect(int32_t* block /* or NULL */, int32_t* dst /* device, 4 words */, void* stream);

/* Kernel-form switches.  Older forms of some kernels stay in the library as bit-identity references for the tests and for
 * same-box A/B timing ("conv_no_rs", "conv_no_rs32", "split_generic", "conv_no_halo", "preprocess_generic",
 * "conv_no_multi", "no_fuse_last_gn", "no_thin_outputs", "thin_form_tap", "thin_form_flat", "halo_stamps", "splitk_fill512",
 * "conv_no_stream", "conv_no_mixed", "conv_no_deepk", "conv_no_fused_reduce"; results unchanged
 * unless a test says otherwise).  The library NEVER reads them from the environment: a development host sets them by name
 * (bench.py and tools/ translate their HN_* variables through hn_amd/forms.py); a product process leaves them alone. */
int hn_set_form(const char* name, int value);
/* Debug / test: the number of split-K ticket counters of the CURRENT device that are not zero.  SYNCHRONISES the device.  The
 * in-kernel split-K reduction (hn_conv2d_nhwc_f16x3_ws, _multi) draws tickets from library-owned counters that are zero at rest:
 * after any sequence of completed launches the answer is 0; anything else means a launch died between drawing a ticket and
 * putting it back (the next split launch on that workspace would then never elect a reducing workgroup). */
int hn_debug_tickets_nonzero(int64_t* count /* host */);
/* Development: scale factor (default 1.0) of a constant of the split-K cost model -- "splitk_fix", "splitk_tk", "splitk_red0",
 * "splitk_plane" (csrc/conv_igemm_f16x3.hip: plan_splits) -- for in-frame scans of the model; never set by the product. */
int hn_set_tuning(const char* name, double value);

/* ------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on f32 / f16 MFMA), fused epilogue.
 * Replaces: torch.nn.Conv2d + (Frozen)BatchNorm2d + ReLU + residual add at
 *   a2j/resnet.py:78-96 (Bottleneck), a2j/a2j.py:70-89,116-135,162-181 (heads),
 *   tv resnet34 BasicBlock / FPN called at fcos_utils/fcos.py:737,
 *   fcos_utils/fcos.py:276-289,377-380 (FCOS towers / outputs; GroupNorm of the
 *   PREVIOUS layer is applied on load via in_scale/in_shift).
 * ------------------------------------------------------------------------------------ */
typedef struct hn_conv_desc {
  int32_t n, h, w, cin;      /* input  [n][h][w][cin], cin % 4 == 0                      */
  int32_t cout;              /* output channels (any >= 1)                               */
  int32_t r, s;              /* filter taps                                              */
  int32_t stride, pad, dil;
  int32_t oh, ow;            /* output spatial size (caller computes; checked)           */
  int32_t relu_cols;         /* ReLU on output channels [0, relu_cols); 0 = none         */
  int32_t res_mode;          /* 0 none | 1 residual[n][oh][ow][cout] |
                                2 residual[n][res_h][res_w][cout] read at (oh*res_h/oh.., nearest) */
  int32_t res_h, res_w;      /* residual spatial size for res_mode 2                     */
  int32_t in_affine;         /* 1: x <- relu(x*in_scale[img][c] + in_shift[img][c]) on load
                                 (f32 kernel only)                                        */
  int32_t tile;              /* 0 = auto; else one of HN_TILE_*                          */
  int32_t out_split;         /* 1: write y in the S32 split format (cout % 32 == 0)      */
  int32_t res_split;         /* 1: residual is an S32 tensor (f16x3 kernel only)         */
  int32_t res_pix_stride;    /* elements between residual pixels; 0 = dense              */
  int32_t in_pix_stride;     /* elements between consecutive input pixels; 0 = dense.  Lets a
                                 conv read a channel slice of a wider tensor (pass the slice's
                                 first element).  fp32: floats (multiple of 4); S32: halfs
                                 (2 x parent channels)                                     */
  int32_t out_pix_stride;    /* elements between consecutive output pixels; 0 = dense    */
  int32_t in_affine_stride;  /* floats between rows of in_scale / in_shift; 0 = cin     */
  int32_t splitk;            /* f16x3 + workspace: 0 = split-K only for long k loops (an extra launch
                              * per conv costs eager callers more than it saves), 1 = also for short
                              * ones (launch cost hidden, e.g. under hipGraph replay), -1 = never; >= 2 = exactly that many
                              * splits, whatever the grid (sweeps: tools/splitk_sweep.py) */
  int32_t terms;             /* f16x3 kernels: 0 / 3 = the three-term split product (fp32-grade, the default);
                              * 1 = hi*hi only ("f16x1": plain fp16 operands, ONE MFMA per MAC, same data movement) --
                              * the throughput mode SURVEY D6 plans beside the parity mode; misses the 1e-3 keypoint
                              * contract by two orders and is reported beside the headline, never as it */
} hn_conv_desc;

#define HN_TILE_AUTO 0
#define HN_TILE_128x128 1
#define HN_TILE_128x64 2
#define HN_TILE_64x64 3
#define HN_TILE_128x32 4
/* 5 (256x128), 9 (256x128 with 8 waves), 10 (256x64 with 8 waves), 11 (128x256 with 8 waves): sweep-only forms of rounds 1-4,
 * retired in round 5 (measured and not taken: profiles/r04_tile_128x256.txt, profiles/NOTEBOOK.md); the ids stay reserved */
#define HN_TILE_64x128 6
#define HN_TILE_32x64 7   /* f16x3 only: 2-wave workgroups for small-M layers */
#define HN_TILE_256x64 8  /* f16x3 only: Cout <= 64 layers with 64x64 wave tiles (4 waves stacked along M) */
/* f16x3 only, round 5: the "deep k" form of the 64x64 tile for grids of at most one workgroup per CU (ResNet-34 layer3 at batch
 * 1): a ring stage holds TWO k tiles and the loop has one barrier per stage instead of one per tile; bit-identical to
 * HN_TILE_64x64; the auto heuristic picks it for such grids with at least 8 k tiles ("conv_no_deepk" turns that off).  Members of
 * a heterogeneous launch whose tile is 64x64 share its kernel. */
#define HN_TILE_64x64_K2 12

int hn_conv2d_nhwc_f32(const hn_conv_desc* d, const float* x, const float* w,
                       const float* bias /* [cout] or NULL */,
                       const float* residual /* or NULL */,
                       const float* in_scale, const float* in_shift /* [n][cin] or NULL */,
                       float* y, void* stream);

/* Tile id (HN_TILE_*) the auto heuristic picks for this shape (host-only; for tests/plans). */
int hn_conv2d_pick_tile(const hn_conv_desc* d);

/* Same convolution on the f16 MFMA with SPLIT operands: every fp32 value v = hi + lo,
 * hi = fp16(v), lo = fp16(v - hi); a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in
 * fp32 (fp32-grade result, 16/3 x the f32-MFMA rate).  The input is an S32 split tensor
 * (below), written once by its producer; w16 is the filter bank split on the host the same
 * way: fp16 [cout][(cin/32)*r*s][2][32] (k tiles: 32-channel block outer, tap (r,s) inner;
 * hi run | lo run per tile).  Requires cin % 32 == 0.
 * GroupNorm-on-load (in_affine) is not available here: run hn_affine_split_f32 in between. */
int hn_conv2d_nhwc_f16x3(const hn_conv_desc* d, const void* x16 /* S32 */, const void* w16,
                         const float* bias, const void* residual /* fp32 or S32 */,
                         void* y /* fp32 or S32 (d->out_split) */, void* stream);
int hn_conv2d_f16x3_pick_tile(const hn_conv_desc* d);
/* 1 when a single-pass launch of this descriptor runs the row-shared-A kernels (3x3 / stride 1 / pad 1 / dilation 1 on the
 * 128x128, 128x64 or 128x32 tile, image rows long enough for the gap slots; HN_CONV_NO_RS=1 turns them off): the A
 * operand of a filter row is staged once and the three taps read it at slot offsets.  Same results bit for bit; the
 * profilers' kernel names differ (template argument RS), which is what bench.py asks this for.  For a grouped launch pass
 * the smallest member width in d->w and the picked tile in d->tile. */
int hn_conv2d_f16x3_uses_rs(const hn_conv_desc* d);
/* 1 when hn_conv2d_nhwc_f16x3(_ws) routes this descriptor to the halo-patch kernel (3x3 / stride 1 / pad 1, 64 output channels,
 * S32 in and out, ReLU, optional S32 residual of the output's shape, at least 512 tiles of 16 x 16 pixels; HN_CONV_NO_HALO=1
 * turns it off): a workgroup stages the 18 x 18 input patch of a 32-channel block once and reads all nine taps from it.
 * Same k order as the implicit-GEMM kernel: bit-identical results (ResNet-34 layer1, fcos_utils/fcos.py:737). */
int hn_conv2d_f16x3_uses_halo(const hn_conv_desc* d, int has_residual);
/* 1 when hn_conv2d_nhwc_f16x3(_ws) routes this descriptor to the streaming kernel for short-k 1x1 layers (1x1 / stride 1 / pad 0,
 * cin 64 or 128, cout % 256 == 0, dense S32 input, ReLU on all columns or none, at least 65536 output pixels; "conv_no_stream"
 * turns it off): a wave keeps the filter fragments of its 32 output channels in registers for the whole launch and the
 * activations are streamed through LDS once.  Same k order and epilogue arithmetic as the implicit-GEMM kernel: bit-identical
 * results (the FPN P3 lateral via fcos_utils/fcos.py:737, the bottleneck expansions of a2j/resnet.py:78-96 at batch >= ~32). */
int hn_conv2d_f16x3_uses_stream(const hn_conv_desc* d);

/* ---- S32 split activation format: fp16 [N][H][W][C/32][2][32] (hi[32] | lo[32] per block) ----
 * hn_affine_split_f32: fp32 NHWC -> S32; with scale/shift [n][c] it first applies
 *   y = x*scale[img][c] + shift[img][c] (the GroupNorm(32,256) of fcos_utils/fcos.py:232-239,
 *   352-359 expressed as a per-(image, channel) affine) and, if relu, max(y, 0).
 * hn_unsplit_f32: S32 -> fp32 (exact: hi + lo).   hn_maxpool3x3s2_s32: pooling on S32. */
int hn_affine_split_f32(const float* x, const float* scale, const float* shift, int relu,
                        int n, int hw, int c, int in_pix_stride, int affine_stride,
                        void* y16, int out_pix_stride, void* stream);
int hn_unsplit_f32(const void* x16, int n, int hw, int c, int in_pix_stride,
                   float* y, int out_pix_stride, void* stream);
int hn_maxpool3x3s2_s32(const void* x16, void* y16, int n, int h, int w, int c,
                        int oh, int ow, void* stream);

/* 3x3 / stride-2 / pad-1 max pooling, NHWC fp32 (c % 4 == 0).
 * Replaces nn.MaxPool2d at a2j/resnet.py:107,158 and tv resnet34 stem. */
int hn_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n, int h, int w, int c,
                             int oh, int ow, void* stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm statistics -> per-(image, channel) affine, consumed by the next conv's
 * in_scale / in_shift.  Replaces nn.GroupNorm(32, 256) + ReLU at
 * fcos_utils/fcos.py:232-239,352-359 (eps 1e-5, biased variance over group x H x W).
 *   scale[img][c] = gamma[c] * rstd[img][g],  shift[img][c] = beta[c] - mean*scale
 * `partial` is caller scratch of hn_groupnorm_scratch_floats(...) floats.
 * ------------------------------------------------------------------------------------ */
int64_t hn_groupnorm_scratch_floats(int n, int hw, int c, int groups);
int hn_groupnorm_affine_f32(const float* x /* [n][hw][c] */, const float* gamma,
                            const float* beta, int n, int hw, int c, int groups, float eps,
                            float* partial, float* scale, float* shift, void* stream);

/* hn_conv2d_nhwc_f16x3 with a caller-provided workspace (fp32 scratch, 16-byte aligned, any size; only used
 * between this call's own launches, so one buffer per stream serves all convolutions).  With it, layers
 * whose output grid would leave most CUs idle (n*oh*ow small: the 11x11 A2J maps, everything at batch 1)
 * run split-K: up to 16 workgroups share an output tile, partial tiles are summed in a fixed order -- for the 32- / 64-row
 * tiles by the workgroup that finishes a tile last (a ticket per tile in library-owned device memory, keyed by the workspace
 * ADDRESS: as before, launches that may be in flight together must not share a workspace), otherwise by a second launch;
 * the results are bit-identical either way ("conv_no_fused_reduce" forces the second launch). */
int hn_conv2d_nhwc_f16x3_ws(const hn_conv_desc* desc, const void* x16, const void* w16,
                            const float* bias, const void* residual, void* y,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* Up to HN_CONV_MAX_GROUP convolutions that share ONE descriptor (identical channels, filter, batch, strides and epilogue
 * flags -- only the spatial size may differ per member; no residual, no split-K) as a single launch: member g reads x16[g] / w16[g] / bias[g] (bias may be NULL) and writes
 * y[g]; gn_partial[g] is either given for every member (semantics of hn_conv2d_nhwc_f16x3_gn) or NULL for all.
 * Used for the cls / reg tower layers of an FPN level (fcos.py:276,377) and the three A2J head layers
 * (a2j/a2j.py:70-181), which are independent and identical in shape. */
#define HN_CONV_MAX_GROUP 6
typedef struct hn_conv_group {
  int32_t count;
  const void* x16[HN_CONV_MAX_GROUP];
  const void* w16[HN_CONV_MAX_GROUP];
  const float* bias[HN_CONV_MAX_GROUP];
  void* y[HN_CONV_MAX_GROUP];
  float* gn_partial[HN_CONV_MAX_GROUP];
  int32_t h[HN_CONV_MAX_GROUP], w[HN_CONV_MAX_GROUP];  /* per-member input size; 0 = the descriptor's (members may be the
                      * FPN levels of one layer: same channels / filter / batch, different maps) */
  int32_t gn_units;  /* 8-channel units per 32-row group in the GroupNorm slab; 0 = cout/8.  With y[g] / gn_partial[g]
                      * pointing at channel offset g*cout of ONE [rows][count*cout] tensor / slab (desc.out_pix_stride =
                      * count*cout, gn_units = count*cout/8) the members' outputs stay stacked, so one
                      * hn_groupnorm_finalize_rows32 and one hn_affine_split_f32 serve all of them. */
} hn_conv_group;
int hn_conv2d_nhwc_f16x3_grouped(const hn_conv_desc* desc, const hn_conv_group* group, void* stream);

/* Up to HN_CONV_MULTI_MAX INDEPENDENT convolutions of DIFFERENT shapes as one launch (plus one reduction launch for all their
 * split-K members): each member has its own descriptor (channels, filter, stride, dilation, residual, ReLU, output format,
 * splitk mode) and tensors, exactly as for hn_conv2d_nhwc_f16x3_ws, and computes exactly what that call would -- same
 * kernel body, same k order, the split-K plan it would get alone -- so results are bit-identical to separate calls.
 * The members run together when each of them alone would take the implicit-GEMM kernel of one common tile form
 * (hn_conv2d_f16x3_pick_tile); otherwise, or with HN_CONV_NO_MULTI=1, the call issues them one after the other.
 * Used where the layer graph has independent small grids: the 1x1 downsample beside conv1 of a residual block
 * (a2j/resnet.py:78-96, tv BasicBlock), the classification head beside layer4 of the A2J trunk (a2j/a2j.py:194-210,
 * 226-250).  No member may read another member's output. */
#define HN_CONV_MULTI_MAX 4
typedef struct hn_conv_multi {
  int32_t count;
  hn_conv_desc desc[HN_CONV_MULTI_MAX];
  const void* x16[HN_CONV_MULTI_MAX];
  const void* w16[HN_CONV_MULTI_MAX];
  const float* bias[HN_CONV_MULTI_MAX];      /* or NULL */
  const void* residual[HN_CONV_MULTI_MAX];   /* or NULL */
  void* y[HN_CONV_MULTI_MAX];
} hn_conv_multi;
int hn_conv2d_nhwc_f16x3_multi(const hn_conv_multi* mm, void* workspace, int64_t workspace_bytes, void* stream);
/* 1 when the call above would run these members as ONE launch (host-only). */
int hn_conv2d_f16x3_multi_fuses(const hn_conv_multi* mm, int64_t workspace_bytes);

/* Fused variant for the f16x3 path: hn_conv2d_nhwc_f16x3_gn is hn_conv2d_nhwc_f16x3 (fp32 output, no
 * residual / ReLU, cout % 8 == 0, oh*ow >= 32) whose epilogue also writes GroupNorm partial sums
 * gn_partial [ceil(n*oh*ow / 32)][cout/8][4] (hn_groupnorm_rows32_scratch_floats floats);
 * hn_groupnorm_finalize_rows32 reduces them to the same scale / shift tables as hn_groupnorm_affine_f32
 * without re-reading the conv output. */
int hn_conv2d_nhwc_f16x3_gn(const hn_conv_desc* desc, const void* x16, const void* w16,
                            const float* bias, void* y, float* gn_partial, void* stream);
int64_t hn_groupnorm_rows32_scratch_floats(int64_t rows, int c);
int hn_groupnorm_finalize_rows32(const float* partial, const float* gamma, const float* beta,
                                 int n, int hw, int c, int groups, float eps,
                                 float* scale, float* shift, void* stream);

/* The GroupNorm finalize / apply passes of ONE tower layer for all FPN levels at once (the tower weights and the
 * GroupNorm affine are shared across levels, fcos_utils/fcos.py:276-289,377-380; the levels differ in map size only).
 * Same arithmetic and bits as hn_groupnorm_finalize_rows32 / hn_affine_split_f32 per level; one launch instead of
 * `count` (at batch 1 every launch of this size is pure latency).  hn_affine_split_f32_levels falls back to the
 * per-level launches when a level is too large for any cache (>= 128 MiB), where the streaming-store form wins. */
#define HN_FCOS_MAX_LEVELS 5
typedef struct hn_gn_levels {
  int32_t count;
  int32_t hw[HN_FCOS_MAX_LEVELS];            /* pixels per image of each level (>= 32)        */
  const float* partial[HN_FCOS_MAX_LEVELS];  /* conv-epilogue partial sums of each level       */
  float* scale[HN_FCOS_MAX_LEVELS];          /* out: [n][c] per level                          */
  float* shift[HN_FCOS_MAX_LEVELS];
} hn_gn_levels;
int hn_groupnorm_finalize_rows32_levels(const hn_gn_levels* lv, const float* gamma, const float* beta,
                                        int n, int c, int groups, float eps, void* stream);
typedef struct hn_split_levels {
  int32_t count;
  int32_t hw[HN_FCOS_MAX_LEVELS];
  const float* x[HN_FCOS_MAX_LEVELS];        /* fp32 [n][hw][c] (pixel stride in_pix_stride)   */
  const float* scale[HN_FCOS_MAX_LEVELS];    /* [n][c] tables (row stride affine_stride)       */
  const float* shift[HN_FCOS_MAX_LEVELS];
  void* y16[HN_FCOS_MAX_LEVELS];             /* out: S32 [n][hw][c/32][2][32]                  */
} hn_split_levels;
int hn_affine_split_f32_levels(const hn_split_levels* lv, int relu, int n, int c, int in_pix_stride,
                               int affine_stride, int out_pix_stride, void* stream);

/* The FCOS head OUTPUT convolutions (cls_logits + hand_lr, bbox_reg + bbox_ctrness, the ext heads: fcos_utils/fcos.py:
 * 247-264,299-320,362-363): 3x3 / stride 1 / pad 1, cin % 32 == 0, 1 <= cout <= 16, the same filter on every FPN level,
 * all levels in ONE launch.  x16[l] = S32 input of level l (a channel slice of a wider tensor: in_pix_stride in halfs,
 * 0 = dense), y[l] = fp32 [n][h_l][w_l][cout] dense, ReLU on channels [0, relu_cols).  Two kernels (csrc/conv3x3_thin.hip):
 *   - cout <= 5, cin % 128 == 0, level width <= ~170, >= 65536 pixels in all (hn_conv3x3_thin_uses_flat() == 1): the P form -- one GEMM
 *     P[(tap, oc)][pixel] over the flat pixel list, every input pixel read from HBM once, then nine shifted adds per output
 *     from an LDS ring.  Same products as the implicit GEMM, summed per tap first: equal to fp32 rounding, not bit for bit;
 *   - otherwise the tap kernel: 16 x 16 pixel tiles, the implicit GEMM's k / term order, bit-identical to
 *     hn_conv2d_nhwc_f16x3_grouped on the same operands. */
typedef struct hn_thin_levels {
  int32_t count;
  const void* x16[HN_FCOS_MAX_LEVELS];
  float* y[HN_FCOS_MAX_LEVELS];
  int32_t h[HN_FCOS_MAX_LEVELS], w[HN_FCOS_MAX_LEVELS];
} hn_thin_levels;
int hn_conv3x3_thin_f16x3_levels(const hn_thin_levels* lv, int n, int cin, int cout, const void* w16, const float* bias,
                                 int relu_cols, int in_pix_stride, void* stream);
int hn_conv3x3_thin_uses_flat(const hn_thin_levels* lv, int n, int cin, int cout);   /* only count, h[] and w[] are read */

/* Up to three such convolutions -- different filter banks on their own inputs: the FCOS head outputs cls_logits + hand_lr, the
 * ext heads, bbox_reg + ctrness (fcos.py:255-264,299-320) -- in ONE launch where they run the tap kernel (a single frame: 89
 * workgroups each on 256 CUs, latency-bound at 17-18 us per launch; together one round of the chip).  Every workgroup runs
 * exactly what its member's own launch would have run: bit-identical to `count` calls of hn_conv3x3_thin_f16x3_levels, which is
 * also what happens where a member takes the P form (large pixel counts: HBM-bound streams) or for count == 1. */
typedef struct hn_thin_member {
  hn_thin_levels lv;
  int32_t cout, relu_cols;
  const void* w16;
  const float* bias;
} hn_thin_member;
int hn_conv3x3_thin_f16x3_levels_group(const hn_thin_member* members, int count /* 1..3 */, int n, int cin, int in_pix_stride,
                                       void* stream);

/* The P form with the LAST GroupNorm apply pass of the towers fused in (fcos.py:232-239 conv -> GroupNorm -> ReLU, then the
 * output conv): x[l] = RAW fp32 conv output [n][h_l][w_l][in_pix_stride] at the head's first channel, scale[l] / shift[l] =
 * the tables of hn_groupnorm_finalize_rows32 [n][affine_stride] at the same channel; the kernel computes relu(x * scale +
 * shift), splits it into fp16 hi + lo in registers (hn_affine_split_f32's arithmetic) and convolves: bit-identical to
 * hn_affine_split_f32_levels + hn_conv3x3_thin_f16x3_levels, without the 8 bytes per element the separate pass moves.
 * lv->x16 is ignored; lv->y / h / w as above.  Only where hn_conv3x3_thin_affine_applies() == 1. */
typedef struct hn_thin_affine {
  const float* x[HN_FCOS_MAX_LEVELS];
  const float* scale[HN_FCOS_MAX_LEVELS];
  const float* shift[HN_FCOS_MAX_LEVELS];
  int32_t in_pix_stride;   /* floats per pixel of x */
  int32_t affine_stride;   /* floats per image row of scale / shift */
} hn_thin_affine;
int hn_conv3x3_thin_affine_applies(const hn_thin_levels* lv, int n, int cin, int cout);
int hn_conv3x3_thin_affine_f16x3_levels(const hn_thin_levels* lv, const hn_thin_affine* aff, int n, int cin, int cout,
                                        const void* w16, const float* bias, int relu_cols, void* stream);

/* ------------------------------------------------------------------------------------
 * Caller-side ingest, fused: what ros_demo.py:227-231,266-269 does on the host before it calls the network --
 *   bgr8 HWC uint8 -> cv2.COLOR_BGR2RGB -> transpose(2,0,1) -> float32 / 255.0        -> rgb    [n][3][h][w]
 *   16UC1 depth in millimetres -> float32 / 1000.0 (32FC1 metres: passed through)      -> depth_m [n][1][h][w]
 *   RGB-D model: torch.cat([rgb, depth], dim=1)                                        -> rgbd   [n][4][h][w]
 * bgr [n][h][w][3] uint8 and depth [n][h][w] (depth_kind 1: uint16 mm, 2: float32 m, 0: none / NULL) only have to be
 * READABLE by the device: device memory, or pinned host memory read over PCIe by the kernel itself (1.5 MB per 640x480
 * frame instead of the 4.9 MB of the fp32 feed).  rgb / depth_m / rgbd are device tensors; any of them may be NULL if
 * not wanted (rgbd needs a depth input).  IEEE float32 divisions: bit-identical to the reference's numpy arithmetic.
 * ------------------------------------------------------------------------------------ */
int hn_ingest_u8bgr_u16mm(const uint8_t* bgr, const void* depth, int depth_kind, float* rgb, float* depth_m,
                          float* rgbd, int n, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------
 * FCOS pre-processing: normalize + bilinear resize (align_corners=False, scale =
 * in/out as with recompute_scale_factor=True) + zero pad, NCHW fp32 in -> NHWC(4) out.
 * Replaces torchvision GeneralizedRCNNTransform called at fcos_utils/fcos.py:709.
 * src [n][3][h][w], dst [n][ph][pw][4] (channel 3 = 0, rows >= oh / cols >= ow = 0).
 * ------------------------------------------------------------------------------------ */
int hn_fcos_preprocess_f32(const float* src, float* dst, int n, int h, int w,
                           int oh, int ow, int ph, int pw,
                           const float mean[3], const float stdv[3], void* stream);

/* Split-precision stem path.  hn_fcos_preprocess_split is hn_fcos_preprocess_f32 with the output stored as
 * the stem image of hn_conv_stem_f16x3: two fp16 planes (hi then lo, hi + lo = the fp32 value to 2^-22) of
 * [n][ph + 2*border][pw + 2*border][4], zero outside the resized image (border, padding, channel 3).
 * hn_conv_stem_f16x3 runs the R x R / stride convolution (pad = border = R/2, R <= 8; torchvision resnet
 * conv1 at fcos_utils/fcos.py:737) on the f16x3 kernel: w16 is fp16 [cout][R][2][32] with k = kx*4 + c inside
 * a filter row (hn_amd.weights.pack_stem_split); y is [n][oh][ow][cout] fp32 or S32 (out_split). */
int hn_fcos_preprocess_split(const float* src, void* dst16, int n, int h, int w, int oh, int ow,
                             int ph, int pw, int border, const float mean[3], const float stdv[3],
                             void* stream);
/* Batches of differently sized images (torchvision batch_images, fcos_utils/fcos.py:702-709): srcs = DEVICE
 * array of n device pointers to [3][h_i][w_i] images, geom = DEVICE int32 [n][4] = {h_i, w_i, oh_i, ow_i};
 * each image is resized on its own into the top-left corner of the common zero canvas.
 * split = 0: dst = fp32 [n][ph][pw][4]; split = 1: dst = the stem image (border as above). */
int hn_fcos_preprocess_list(const float* const* srcs, const int32_t* geom, void* dst, int split, int n,
                            int ph, int pw, int border, const float mean[3], const float stdv[3],
                            void* stream);
int hn_conv_stem_f16x3(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                       const void* w16, const float* bias, int relu, void* y, int out_split,
                       void* stream);
/* The stem with the ResNet's 3x3 / stride-2 / pad-1 max pooling fused into its epilogue (conv1 -> bn1 -> relu -> maxpool of
 * torchvision resnet34, fcos_utils/fcos.py:737): cout = 64, ReLU on, y = the POOLED S32 map [n][(oh+1)/2][(ow+1)/2][64].
 * Bit-identical to hn_conv_stem_f16x3 followed by hn_maxpool3x3s2_s32; the half-resolution conv map (1.8 GB at batch
 * 32) never reaches HBM.  Written for the 7x7 / stride-2 / pad-3 ResNet stem (r = 7, stride = 2, pad = 3), the only stem of this
 * path: other filter geometries are refused (run hn_conv_stem_f16x3 + hn_maxpool3x3s2_s32). */
int hn_conv_stem_pool_f16x3(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                            const void* w16, const float* bias, void* y, void* stream);
/* hn_conv_stem_pool_f16x3 with the term count of hn_conv_desc.terms (1 = the f16x1 throughput mode). */
int hn_conv_stem_pool_f16x3_terms(const void* x16, int n, int ph, int pw, int pad, int r, int stride, int cout,
                                  const void* w16, const float* bias, void* y, int terms, void* stream);

/* ------------------------------------------------------------------------------------
 * FCOS post-processing.  Replaces fcos_utils/fcos.py:572-659 (postprocess_detections),
 * det_utils.py:266-294 (BoxLinearCoder.decode_single), anchor_utils.py:82-132,
 * torchvision.ops.batched_nms (call site fcos.py:635) and resize_boxes fcos.py:770-783.
 *
 * Head tensors are the raw conv outputs per level, NHWC:
 *   cls_lr[l]  [n][h_l][w_l][num_classes + 2]   (cls_logits then hand_lr)
 *   reg_ctr[l] [n][h_l][w_l][5]                 (relu(bbox_reg)[4] then ctrness)
 * Step 1 hn_fcos_candidates: score = sqrt(sigmoid(cls)*sigmoid(ctr)); max/argmax over
 *   classes (ties -> lowest class); keep score > thresh; decode box; compact in anchor
 *   order.  Output per image (capacity `cap` = total points): cand_* arrays + count.
 * Step 2 hn_fcos_nms: sort candidates by (score desc, index asc), greedy NMS exactly as
 *   torchvision 0.11.3 batched_nms (coordinate trick when 4*K <= 4000, per-class
 *   otherwise; CPU-kernel rule: suppress iff (double)iou > iou_thresh), write kept
 *   detections in score order, boxes rescaled by (ratio_h, ratio_w).
 * ------------------------------------------------------------------------------------ */
typedef struct hn_fcos_levels {
  int32_t num_levels;
  int32_t h[HN_FCOS_MAX_LEVELS], w[HN_FCOS_MAX_LEVELS], stride[HN_FCOS_MAX_LEVELS];
  const float* cls_lr[HN_FCOS_MAX_LEVELS];
  const float* reg_ctr[HN_FCOS_MAX_LEVELS];
} hn_fcos_levels;

int hn_fcos_candidates(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh,
                       float* cand_boxes /* [n][cap][4] */, float* cand_scores /* [n][cap] */,
                       int32_t* cand_labels, int32_t* cand_sides, int32_t* cand_level /* [n][cap] */,
                       int32_t* cand_point /* [n][cap] anchor-point index, may be NULL */,
                       int32_t* cand_count /* [n] */, int cap, void* stream);
/* The same result from ceil(points / 1024) workgroups per image instead of one (two short launches; the single workgroup
 * costs 60 us per call at batch 1 whatever the number of candidates): workspace = hn_fcos_candidates_ws_bytes(n, total
 * points) bytes of device memory, contents irrelevant before and after. */
int64_t hn_fcos_candidates_ws_bytes(int n, int total_points);
int hn_fcos_candidates_ws(const hn_fcos_levels* lv, int n, int num_classes, float score_thresh, float* cand_boxes,
                          float* cand_scores, int32_t* cand_labels, int32_t* cand_sides, int32_t* cand_level,
                          int32_t* cand_point, int32_t* cand_count, int cap, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* ext=True detector outputs (fcos_utils/fcos.py:255-264 layers, :299-320 head maths, :605-607 argmax,
 * :631-647 gather) for the detections hn_fcos_nms kept.  ext[l] is the raw NHWC conv output
 * [n][h_l][w_l][8] = relu(hand_dydx_layer)[3] then hand_contact_state_layer[5] (host array of
 * lv->num_levels device pointers; lv supplies h/w only).  Row d of image i (d < det_count[i]):
 *   det_dxdymags[i][d] = (mag, 0.1*dx/max(|(dx,dy)|,1e-12), 0.1*dy/...), det_contacts[i][d] = argmax. */
int hn_fcos_ext_gather(const hn_fcos_levels* lv, const float* const* ext, const int32_t* det_keep,
                       const int32_t* cand_point, const int32_t* det_count, int n, int cap,
                       int32_t* det_contacts /* [n][cap] */, float* det_dxdymags /* [n][cap][3] */,
                       void* stream);

int64_t hn_fcos_nms_scratch_bytes(int n, int cap);
int hn_fcos_nms(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                const int32_t* cand_sides, const int32_t* cand_level, const int32_t* cand_count,
                int n, int cap, double iou_thresh, float ratio_h, float ratio_w,
                void* scratch,
                float* det_boxes /* [n][cap][4] */, float* det_scores, int32_t* det_labels,
                int32_t* det_sides, int32_t* det_level, int32_t* det_keep /* cand index */,
                int32_t* det_count /* [n] */, void* stream);

/* hn_fcos_nms with one (ratio_h, ratio_w) pair per image: ratios = DEVICE fp32 [n][2] (resize_boxes of a
 * batch whose images differ in size, fcos_utils/fcos.py:661-669). */
int hn_fcos_nms_ratios(const float* cand_boxes, const float* cand_scores, const int32_t* cand_labels,
                       const int32_t* cand_sides, const int32_t* cand_level, const int32_t* cand_count,
                       int n, int cap, double iou_thresh, const float* ratios, void* scratch,
                       float* det_boxes, float* det_scores, int32_t* det_labels, int32_t* det_sides,
                       int32_t* det_level, int32_t* det_keep, int32_t* det_count, void* stream);

/* Stand-alone NMS with torchvision.ops.nms semantics (tests / callers with own boxes). */
int hn_nms(const float* boxes, const float* scores, int k, double iou_thresh,
           void* scratch /* hn_fcos_nms_scratch_bytes(1,k) */, int32_t* keep, int32_t* num_keep,
           void* stream);

/* ------------------------------------------------------------------------------------
 * Crop: top-1 hand box -> int box -> pad 40 % -> clamp -> nearest resize to out x out.
 * Replaces handnet_pipeline/handnet_pipeline.py:74-105.
 * For each image: first detection (score order) with label == hand_label.  Writes
 *   crop_box [n][4] int64 (x1,y1,x2,y2 after padding; zeros if none), has_hand [n] int32 (0 / 1; the A2J stage of
 *   the pipeline raises it to 2 for a frame whose depth crop holds non-finite pixels: NaN keypoints, like the reference),
 *   crops [n][out][out][cpad] fp32 NHWC with depth in channel 0, other channels 0.
 * depth is [n][in_ch][h][w] fp32, in_ch = 1 (depth) or 4 (RGB-D; reorder_bgr = 1 applies the
 * reference's channel permutation [2,1,0,3], handnet_pipeline.py:102); crops channel c = image channel.
 * ------------------------------------------------------------------------------------ */
int hn_crop_resize(const float* det_boxes, const int32_t* det_labels, const int32_t* det_count,
                   int cap, int hand_label, const float* depth, int n, int in_ch, int reorder_bgr,
                   int h, int w, int out, int cpad, int64_t* crop_box, int32_t* has_hand, float* crops,
                   void* stream);

/* fp32 NHWC(4) image [n][h][w][4] -> the stem image of hn_conv_stem_f16x3 / hn_conv_stem_pool_f16x3 (two fp16 planes hi, lo of
 * [n][h + 2*border][w + 2*border][4], zero border): the A2J crops on their way to the split-precision stem
 * (a2j/resnet.py:155-158 conv1 -> bn1 -> relu -> maxpool). */
int hn_stem_image_nhwc4(const float* x, int n, int h, int w, int border, void* dst16, void* stream);
/* The same with the per-image flags of the aggregation (valid [n] int32 on the device, or NULL): an image with
 * valid[i] == 1 that holds a non-finite pixel gets valid[i] = 2.  The reference's network returns NaN for EVERY joint of
 * such a crop (each cell of its 11 x 11 maps sees every pixel of the crop; ReLU and max pooling propagate NaN in torch),
 * and hn_a2j_aggregate_f32 writes exactly that for valid[i] == 2 -- the split-precision convolutions in between need not
 * (and do not) propagate NaN. */
int hn_stem_image_nhwc4_valid(const float* x, int n, int h, int w, int border, void* dst16, int32_t* valid, void* stream);

/* Pack [n][1][h][w] depth crops into NHWC(cpad) for the A2J stem (A2J-only entry). */
int hn_pack_depth_nhwc(const float* src, float* dst, int n, int hw, int cpad, void* stream);

/* ------------------------------------------------------------------------------------
 * A2J anchor aggregation.  Replaces a2j/anchor.py:57-82 (post_process.forward).
 * Inputs are the raw head conv outputs, NHWC over the 11x11 anchor grid:
 *   cls [k][fh][fw][A*J], reg [k][fh][fw][A*J*2], dep [k][fh][fw][A*J], A = 16 anchors.
 * out [k][J][3] = (sum w*(anchor_0 + reg_0), sum w*(anchor_1 + reg_1), sum w*depth),
 * w = softmax over all fh*fw*A anchors per joint.  Anchor coords follow
 * a2j/anchor.py:7-42: coordinate 0 = h*stride + P[a/4], coordinate 1 = w*stride + P[a%4].
 * With valid != NULL: rows with valid[k] == 0 are written as zeros (no crop), rows with valid[k] == 2 as NaN (a crop
 * with non-finite pixels, see hn_stem_image_nhwc4_valid), rows with valid[k] == 1 are computed.
 * ------------------------------------------------------------------------------------ */
int hn_a2j_aggregate_f32(const float* cls, const float* reg, const float* dep,
                         const int32_t* valid, int k, int fh, int fw, int joints, int stride,
                         float* out, void* stream);

/* crop-(u,v,d) keypoints -> image (u,v,d) or, with paras = host array (fx, fy, cx, cy), camera
 * xyz in millimetres.  Replaces convert_joints (a2j/a2j.py:17-34) + uvd2xyz
 * (datasets3d/a2jdataset.py:31-38), the step every caller runs right after the path
 * (ros_demo.py:289,329-330).  kp [n][joints][3], crop_box [n][4] int64 (x1,y1,x2,y2),
 * valid [n] or NULL (rows with valid == 0 are written as zeros). */
int hn_convert_joints_f32(const float* kp, const int64_t* crop_box, const int32_t* valid,
                          int n, int joints, float crop_w, float crop_h,
                          const float* paras /* host, 4 floats, or NULL */, float* out, void* stream);

/* The evaluation caller's form of the same conversion (A2JModelLightning.test_step, a2j/a2j.py:333-348 -- on the prediction
 * and on the ground truth): box_f32 [n][4] fp32 = the dataset's crop box with fractional corners, sample_paras [n][4] fp32 =
 * each sample's (fx, fy, cx, cy), both on the DEVICE (datasets3d/a2jdataset.py:262-265,279,293).  Writes the image (u,v,d)
 * (out_image_uvd, or NULL) and / or the camera xyz in millimetres (out_xyz_mm, or NULL; needs sample_paras), [n][joints][3]
 * each.  All fp32 in the reference's order: bit-identical to numpy on float32 operands.  valid as above. */
int hn_convert_joints_samples_f32(const float* kp, const float* box_f32, const float* sample_paras /* or NULL */,
                                  const int32_t* valid /* or NULL */, int n, int joints, float crop_w, float crop_h,
                                  float* out_image_uvd, float* out_xyz_mm, void* stream);

/* The aggregation with convert_joints + uvd2xyz in its EPILOGUE: what hn_a2j_aggregate_f32 writes to out_uvd,
 * plus -- from the registers that hold each joint, no second launch -- its image (u,v,d) (out_image_uvd, or NULL) and its
 * camera xyz in millimetres (out_xyz_mm, or NULL; needs paras = host array (fx, fy, cx, cy)), each [k][J][3]: bit-identical
 * to hn_convert_joints_f32 on out_uvd (one device function serves both).  crop_box [k][4] int64 = the padded crop boxes the
 * crops were cut with (hn_crop_resize).  Rows with valid[k] == 0 are zeros in all outputs, rows with valid[k] == 2 NaN.
 * opts (host, or NULL = none): the live caller's two clamps BEFORE the conversion, which it applies to the values it converts
 * (ros_demo.py:279-283) -- they touch the converted outputs only, out_uvd stays what the network returned:
 *   clamp_keypoints  1: the crop-(u,v,d) is clamped to [0, crop_w] (torch.clamp(keypoint_pred, 0, 176), all three columns)
 *   clamp_box_h / _w > 0: box x1,y1 are clamped to [0, clamp_box_h] and x2,y2 to [0, clamp_box_w] (as written there:
 *                    detection[:2] against the image HEIGHT, detection[2:] against the WIDTH) */
typedef struct hn_convert_opts {
  int32_t clamp_keypoints;
  int32_t clamp_box_h, clamp_box_w;
  int32_t reserved;           /* 0 */
  /* The evaluation caller's operands (A2JModelLightning.test_step, a2j/a2j.py:333-348: the DATASET's box and intrinsics per
   * sample, datasets3d/a2jdataset.py:262-265,279,293), device pointers or NULL:
   *   sample_box    [k][4] fp32 (x1,y1,x2,y2), fractional corners: used INSTEAD of crop_box (which may then be NULL); every
   *                 operation of the conversion stays in fp32 like numpy's on a float32 box (bit-identical to the reference)
   *   sample_paras  [k][4] fp32 (fx, fy, cx, cy) per sample: used INSTEAD of `paras` */
  const float* sample_box;
  const float* sample_paras;
} hn_convert_opts;
int hn_a2j_aggregate_convert_f32(const float* cls, const float* reg, const float* dep, const int32_t* valid, int k, int fh,
                                 int fw, int joints, int stride, const int64_t* crop_box, float crop_w, float crop_h,
                                 const float* paras /* host, 4 floats, or NULL */, const hn_convert_opts* opts /* host or NULL */,
                                 float* out_uvd, float* out_image_uvd, float* out_xyz_mm, void* stream);

/* The lifter's input from a step's image-(u,v) joints (the live caller's glue between the path and Pose2Mesh,
 * ros_demo.py:148-157: get_bbox -> process_bbox -> j2d_processing (rot 0, no flip) -> / input_shape -> (x - mean) / std):
 * with no rotation that chain is a per-axis POSITIVE affine map followed by a per-axis standardisation over the J joints,
 * which cancels the map -- so the result is (x - mean_J(x)) / std_J(x) per frame and axis (population std, numpy's default).
 * image_uvd [n][J][3] (columns 0, 1 are read) -> out [n][J][2] fp32; rows with valid[i] != 1 are zeros. */
int hn_joints2d_standardize_f32(const float* image_uvd, const int32_t* valid /* or NULL */, int n, int joints, float* out,
                                void* stream);

/* Per-frame result records for the N > 1 all-gather (ONE collective of fixed-size records per step).
 * Record layout (rec_bytes >= 40 + 12*joints, multiple of 8): bytes 0..31 crop box 4 x int64, 32..35 has_hand,
 * 36..39 row-is-a-real-frame, 40.. keypoints joints*3 x fp32.  hn_pack_records writes rows [0, n) from the three
 * result tensors and zero rows [n, rows) (shard padding); hn_unpack_records is the inverse over `rows` records
 * (valid[r] = the row flag).  One launch each, so a step adds pack + collective + unpack to the engine's launches. */
int hn_pack_records(const float* keypoints, const int64_t* crop_box, const int32_t* has_hand, int n, int rows,
                    int joints, int rec_bytes, void* records, void* stream);
/* Wide records: up to two more [n][joints][3] fp32 fields behind the keypoints (the step's image (u,v,d) and camera xyz in mm,
 * hn_a2j_aggregate_convert_f32): bytes 40 + 12*joints .. and 40 + 24*joints ..; rec_bytes >= 40 + 12*joints per field.
 * hn_unpack_records reads the first field of records of any width. */
int hn_pack_records_ex(const float* keypoints, const int64_t* crop_box, const int32_t* has_hand, int n, int rows, int joints,
                       int rec_bytes, const float* extra0 /* or NULL */, const float* extra1 /* or NULL */, void* records,
                       void* stream);
int hn_unpack_records(const void* records, int rows, int joints, int rec_bytes, float* keypoints,
                      int64_t* crop_box, int32_t* has_hand, int32_t* valid, void* stream);

/* Always-on safety net of the f16x3 path (range contract): counts non-finite values of x[0..count) into
 * *flag (device int32; the caller zeroes it and reads it with the copy of the results it makes anyway). */
int hn_nonfinite_count_f32(const float* x, int64_t count, int32_t* flag, void* stream);

/* ------------------------------------------------------------------------------------
 * Pose2Mesh lifter: Chebyshev graph convolution helpers.  Replaces
 * pose2mesh/lib/models/backbones/cheby_graph_conv.py:5-42 (torch.sparse.mm recursion + concat / permute)
 * and the block residual of meshnet.py:105-113; the Linear + BatchNorm1d + ReLU of a graph conv run on
 * hn_conv2d_nhwc_f16x3_ws as a 1x1 convolution over [batch][vertex] rows (BatchNorm folded into the weights).
 * L is a CSR matrix (int32 indptr [v+1], indices, fp32 values; indices ascending per row); activations are
 * fp32 [batch][v][f], f % 4 == 0.
 *   hn_spmm_csr_f32        y = L x
 *   hn_cheby3_basis_split  x2 = 2 L x1 - x0; out16 = S32 rows [x0 | x1 | x2 | 0-pad] of cpad channels
 *                          (order k-major: channel k*f + i; the reference's (i, k) order is absorbed by
 *                          permuting the Linear's columns at load time, hn_amd/pose2mesh_engine.py)
 *   hn_feat_interp_add_f32 out[r*up + u][:] = y[r][:] + interp_linear(xin[r][:], fi -> fo), u < up
 *                          (F.interpolate(mode='linear') along the feature axis + nn.Upsample(2) on vertices)
 * ------------------------------------------------------------------------------------ */
int hn_spmm_csr_f32(const int32_t* indptr, const int32_t* indices, const float* values, int v,
                    const float* x, float* y, int batch, int f, void* stream);
/* One whole graph convolution of the 'mano' mesh net (K = 3; meshnet.py:95-100 = cheby_graph_conv.py:5-42 + F.relu) as ONE
 * launch, optionally with the block's residual and vertex up-sampling (meshnet.py:105-113) in its epilogue -- the lifter is
 * launch-bound at the live caller's batch:
 *   y[r*up + u][:] = act(W [x0 | L x0 | L2 x0 | 0][r] + bias) (+ interp_linear(xin[r][:], fi -> fout)),  u < up
 * L2 = 2 L L - I, precomputed by the host (the same second-order Chebyshev polynomial, one rounding per coefficient instead of
 * a second dependent sparse product); x fp32 [batch][v][fin] (fin % 4 == 0, <= 256); w16 = the split filter bank
 * [fout][K/32][2][32] over K = pad32(3*fin) k-major channels (as hn_cheby3_basis_split's operand) -- or, with w_frag = 1, the
 * same values in MFMA fragment order [ceil(fout/16)][K/32][2 (hi, lo)][64 lanes][8]: element i of lane l = column 16*nt +
 * (l & 15), channel 32*kt + 8*(l >> 4) + i, zero behind fout (a wave's load is then one contiguous KB); xin fp32 [batch][v][fi]
 * or NULL; y fp32 [batch][v*up][fout], or S32 rows when out_split (fout % 32 == 0).  fout <= 256. */
typedef struct hn_graph_csr {
  const int32_t* indptr;   /* [v + 1] */
  const int32_t* indices;  /* ascending per row */
  const float* values;
  int32_t v;
} hn_graph_csr;
int hn_graph_conv_cheby3_f16x3(const hn_graph_csr* L, const hn_graph_csr* L2, const float* x, int batch, int fin,
                               const void* w16, int w_frag, const float* bias, int fout, int relu, const float* xin, int fi,
                               int up, void* y, int out_split, void* stream);
/* Glue of the lifter as single launches: fp32 rows [rows][f] -> S32 rows of cpad channels, zero padded (the operand of
 * PoseNet's first Linear from the [batch][2J] joints); pose_combine (pose2mesh_net.py:20) = [pose2d | pose3d / 1000 | 0] per
 * joint as fp32 [batch*J][fpad] (the padded input of the first graph convolution; pose3d rows may be padded: PoseNet's last
 * Linear is run with 64 output columns so that it takes the vectorised / split-K form of the convolution kernel). */
int hn_pad_split_rows_f32(const float* x, int64_t rows, int f, int cpad, void* out16, void* stream);
/* The last lines of the live caller's mesh path (ros_demo.py:162,332-337): out[i][k][:] = ((mesh[i][perm[k]][:] * 1000 +
 * xyz_mm[i][0][:]) / 1000) * (1, -1, -1) -- the real mesh's vertices in original order (perm = graph_perm_reverse[:v], int64
 * on the device, entries < v0), moved by the first joint's camera position, y and z negated; numpy's float32 arithmetic, one
 * rounding per operation (bit-identical).  mesh [n][v0][3], xyz_mm [n][joints][3], out [n][v][3]; rows with valid != 1 are zeros. */
int hn_mesh_finish_f32(const float* mesh, const int64_t* perm, const float* xyz_mm, const int32_t* valid /* or NULL */, int n,
                       int v0, int v, int joints, float* out, void* stream);
/* A Linear layer on 1..4 rows as a matrix-vector product on the vector ALU (PoseNet at the live caller's batch, posenet.py:24-41,
 * 78-88: 67 MB of filter bank per 17 M MACs): y[m][:] = act(W (pre(x[m])) + bias (+ residual[m])), pre = relu(x * scale + shift)
 * when scale / shift ([k_real] fp32: the pre-activation BatchNorm) are given.  x fp32 [batch][x_stride] (k_real columns used), w16 =
 * the split bank [n][k/32][2][32] (k % 32 == 0, columns >= k_real zero), y / residual fp32.  w = hi + lo exactly and fp32
 * activations: at least as precise as the three-term MFMA form. */
int hn_linear_rows_f16x3(const float* x, int batch, int x_stride, int k_real, const float* scale, const float* shift,
                         const void* w16, int k, int n, const float* bias, const float* residual, int res_stride, int relu,
                         float* y, int y_stride, void* stream);
int hn_lifter_combine_f32(const float* pose2d /* [batch][joints][2] */, const float* pose3d /* [batch][pose3d_stride], 3*joints used */,
                          int batch, int joints, int pose3d_stride, int fpad, float* out, void* stream);
int hn_cheby3_basis_split(const int32_t* indptr, const int32_t* indices, const float* values, int v,
                          const float* x0, const float* x1, void* out16, int batch, int f, int cpad,
                          void* stream);
int hn_feat_interp_add_f32(const float* xin, const float* y, float* out, int64_t rows, int fi, int fo,
                           int up, void* stream);

/* ------------------------------------------------------------------------------------
 * Model-level entry points: the layer graphs of the detector, the pose network and the HandNet glue in C++
 * (csrc/model.hip), for hosts that are not Python.  They issue exactly the launches the Python engines issue
 * (hn_amd/fcos_engine.py, a2j_engine.py, pipeline.py), so both hosts give bit-identical results.
 *   hn_create          configuration -> handle
 *   hn_load_weight     one entry of a REFERENCE-layout state_dict ("backbone.body.conv1.weight",
 *                      "head.classification_head.conv.0.weight", "Backbone.model.layer1.0.bn1.running_var", ...;
 *                      a leading "a2j." is stripped, a2j/a2j.py:277), fp32 HOST data in torch layout; unknown names
 *                      are kept and ignored (strict=False, handnet_pipeline.py:20,33)
 *   hn_finalize        folds (Frozen)BatchNorm in fp64, repacks to [Cout][R][S][Cin], splits into fp16 hi/lo banks
 *                      (fails if a folded weight leaves the fp16 range), uploads.  Allocates and synchronises.
 *   hn_fcos_forward    fcos_utils/fcos.py:675-767 (eval; hn_fcos_forward_ext adds the ext=True outputs): rgb [n][3][h][w] fp32 0..1 on the device ->
 *                      score-ordered detections, fixed capacity cap = hn_fcos_capacity(m, h, w) rows per image:
 *                      det_boxes [n][cap][4] (original-image pixels), det_scores / det_labels / det_sides /
 *                      det_level [n][cap], det_count [n]
 *   hn_a2j_forward     a2j/a2j.py:243-250: crops [k][1][h][w] fp32 metres -> keypoints [k][J][3] (u, v, d) on the
 *                      device (the reference's .cpu() is the caller's copy); rows with valid[i] == 0 are zeros
 *   hn_handnet_forward(_xyz) handnet_pipeline/handnet_pipeline.py:58-116: rgb [n][3][h][w] + depth [n][1][h][w] (RGB-D
 *                      model: [n][4][h][w]) -> keypoints [n][J][3], crop_box [n][4] int64 (padded, clamped
 *                      x1,y1,x2,y2), has_hand [n]; frames without a hand give zero rows (deviations)
 * All tensors are device pointers borrowed from the caller; work is enqueued on `stream`.  Activations live in one
 * arena per model, sized by a dry pass over the graph: a forward allocates (and synchronises the device) only when
 * it needs more memory than any earlier call of this model; steady-state calls do neither.  One model must not be
 * used from two host threads at once (the reference callable is not re-entered either, ros_demo.py:240-257).
 * ------------------------------------------------------------------------------------ */
#define HN_MODEL_FCOS 1
#define HN_MODEL_A2J 2
#define HN_PRECISION_SPLIT 0
#define HN_PRECISION_F32 1
typedef struct hn_model_config {
  int32_t parts;        /* HN_MODEL_FCOS | HN_MODEL_A2J */
  int32_t num_classes;  /* FCOS classes (ros_demo.py:374 uses 3); hand class = num_classes - 1 */
  int32_t num_joints;   /* 21 */
  int32_t rgbd;         /* 1: 4-channel A2J stem + RGB-D crops with the [2,1,0,3] permutation */
  int32_t min_size, max_size; /* GeneralizedRCNNTransform sizes; 0 = 800 / 1333 (fcos.py:460-461) */
  int32_t ext;          /* 1: also load the ext=True heads (hand_dydx_layer, hand_contact_state_layer; the FCOS class
                           default, fcos.py:255-264) for hn_fcos_forward_ext */
  int32_t f16_terms;    /* 0 or 3: f16x3, the parity-grade default.  1: the f16x1 THROUGHPUT mode -- every convolution of
                           the layer graphs issues the hi*hi term only (hn_conv_desc.terms = 1): 1.4x faster, keypoints
                           ~0.08 px from the reference, outside the 1e-3 contract */
  int32_t precision;    /* HN_PRECISION_SPLIT (0): the split-fp16 kernels above.  HN_PRECISION_F32 (1): the reference's own
                           arithmetic -- every convolution on the exact f32-MFMA kernel (hn_conv2d_nhwc_f32), fp32 activations,
                           GroupNorm applied while the next convolution stages its input; ~4x slower, no fp16-range contract
                           (the mode to run a checkpoint in that trips hn_finalize's / the engines' range check).  Same
                           launches as FCOSEngine / A2JEngine(precision="f32"): bit-identical results */
  float image_mean[3], image_std[3]; /* the transform's normalisation (FCOS ctor arguments, fcos.py:501-505); image_std all
                           zero = the reference's default (ImageNet's 0.485 / 0.456 / 0.406, 0.229 / 0.224 / 0.225) */
} hn_model_config;
typedef struct hn_model hn_model;

int hn_create(const hn_model_config* cfg, hn_model** out);
int hn_load_weight(hn_model* m, const char* name, const float* host_data, const int64_t* shape, int ndim);
int hn_finalize(hn_model* m);
int64_t hn_fcos_capacity(const hn_model* m, int h, int w);
int hn_fcos_forward(hn_model* m, const float* rgb, int n, int h, int w, float* det_boxes, float* det_scores,
                    int32_t* det_labels, int32_t* det_sides, int32_t* det_level, int32_t* det_count, int cap,
                    void* stream);
/* A list of differently sized images (torchvision batch_images, fcos.py:702-709, as FCOS.forward takes it): images = HOST
 * array of n DEVICE pointers to fp32 [3][hs[i]][ws[i]], hs / ws HOST arrays.  Each image is resized on its own into the
 * common canvas and its boxes are rescaled by its own ratios; cap = hn_fcos_capacity_list(m, hs, ws, n). */
int64_t hn_fcos_capacity_list(const hn_model* m, const int32_t* hs, const int32_t* ws, int n);
int hn_fcos_forward_list(hn_model* m, const float* const* images, const int32_t* hs, const int32_t* ws, int n,
                         float* det_boxes, float* det_scores, int32_t* det_labels, int32_t* det_sides, int32_t* det_level,
                         int32_t* det_count, int cap, void* stream);
/* ext=True detections (fcos.py:637-647): additionally det_contacts [n][cap] (argmax of the contact state) and
 * det_dxdymags [n][cap][3] (magnitude, 0.1 * unit dx, 0.1 * unit dy) per kept detection. */
int hn_fcos_forward_ext(hn_model* m, const float* rgb, int n, int h, int w, float* det_boxes, float* det_scores,
                        int32_t* det_labels, int32_t* det_sides, int32_t* det_level, int32_t* det_count,
                        int32_t* det_contacts, float* det_dxdymags, int cap, void* stream);
int hn_a2j_forward(hn_model* m, const float* crops, int k, int h, int w, const int32_t* valid /* or NULL */,
                   float* keypoints, void* stream);
int hn_handnet_forward(hn_model* m, const float* rgb, const float* depth, int n, int h, int w, float* keypoints,
                       int64_t* crop_box, int32_t* has_hand, void* stream);
/* hn_handnet_forward + what the reference's caller computes from its results right away (ros_demo.py:289,329-330;
 * a2j/a2j.py:17-43): image (u,v,d) and / or camera xyz in mm per joint, written by the aggregation's epilogue
 * (hn_a2j_aggregate_convert_f32) -- same launches as hn_handnet_forward.  image_uvd / xyz_mm [n][J][3] device, either may be
 * NULL (not both); paras (host: fx, fy, cx, cy) is needed for xyz_mm; opts as in hn_a2j_aggregate_convert_f32. */
int hn_handnet_forward_xyz(hn_model* m, const float* rgb, const float* depth, int n, int h, int w, const float* paras,
                           const hn_convert_opts* opts, float* keypoints, float* image_uvd, float* xyz_mm, int64_t* crop_box,
                           int32_t* has_hand, void* stream);
int hn_destroy(hn_model* m);

#ifdef __cplusplus
}
#endif
#endif /* HANDNET_HIP_H */
