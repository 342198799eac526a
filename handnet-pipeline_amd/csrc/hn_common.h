// Shared helpers for libhandnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/handnet_hip.h"

namespace hn {

// thread-local error text returned by hn_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);
// device address of the sticky f16x3 range flag of the current device, or nullptr while the check is off
int* range_flag_ptr();
// Kernel-form switches (hn_set_form; all false in a product process: the library never reads the environment)
struct EnvFlags {
  bool no_rs, no_rs32, split_generic, no_halo, no_thin, thin_tap, thin_flat, no_fuse_last_gn, pre_generic, no_multi,
      halo_stamps, splitk_fill512, no_stream, no_mixed, no_deepk, no_fused_reduce, thin_nogroup;
};
const EnvFlags& env_flags();
// development: scale factors of planning constants (hn_set_tuning; 1.0 in a product process)
struct Tuning {
  double splitk_fix = 1.0, splitk_tk = 1.0, splitk_red0 = 1.0, splitk_plane = 1.0;
};
const Tuning& tuning();
// conv_stem_direct.hip: the 7x7 / stride-2 / 64-channel stem + ReLU + max pooling as a direct convolution from an LDS patch
// conv3x3_halo.hip: 3x3 / stride 1 / pad 1, 64 output channels, as a direct convolution from an LDS-resident halo patch
bool conv3x3_halo_applies(const hn_conv_desc* d, bool has_gn, bool has_group, const void* residual);
bool conv3x3_halo_operands_ok(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                              const void* y);
int conv3x3_halo(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                 hipStream_t st);
int stem_pool_direct(const void* x16, int n, int ph, int pw, const void* w16, const float* bias, void* y, int terms, hipStream_t st);
// conv1x1_stream.hip: 1x1 / stride 1, Cin 64 or 128, Cout % 256 == 0 on many pixels: filter bank in registers, activations streamed
bool conv1x1_stream_applies(const hn_conv_desc* d, bool has_gn, bool has_group);
bool conv1x1_stream_operands_ok(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual,
                                const void* y);
int conv1x1_stream(const hn_conv_desc* d, const void* x16, const void* w16, const float* bias, const void* residual, void* y,
                   hipStream_t st);

// hn_debug_tickets_nonzero: the split-K ticket arrays of the two translation units that hold one
int tickets_nonzero_main(int64_t* count);
int tickets_nonzero_multi(int64_t* count);

#define HN_CHECK_ARG(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return hn::fail(HN_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define HN_CHECK_HIP(expr)                                                         \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return hn::fail(HN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// kernel launches report configuration errors through hipGetLastError
#define HN_CHECK_LAUNCH(name)                                                          \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess)                                                              \
      return hn::fail(HN_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// f16x3 range contract: a value that is to be split into fp16 hi + lo must be finite and within the fp16 range
// (hi = fp16(v) is +-inf beyond 65504).  `flag` is null unless hn_range_check_enable(1) was called.
__device__ __forceinline__ void range_note(int* flag, float v) {
  if (!(fabsf(v) <= 65504.f)) *flag = 1;
}
// NaN-PROPAGATING maximum (IEEE 754-2019 "maximum": v_maximum3_f32 on gfx950, same issue cost as v_max3_f32, which returns the
// non-NaN operand).  ReLU and max pooling are built on it, so a NaN activation stays NaN -- as torch.relu / max_pool2d keep it --
// instead of being laundered to 0, and the range notes below see it.
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float relu(float v) { return max_nan(v, 0.f); }
// ACTIVATIONS, N values at once: one NaN-propagating maximum of the magnitudes (v_maximum3_f32 with |.| modifiers) and ONE
// comparison -- a third of the instructions of N separate range_note calls, which cost the short-k convolutions' epilogues
// +0.5 % of the batch-32 step.  !(m <= 65504) is true for a magnitude beyond the fp16 range, for +-inf AND for NaN (an fp32
// residual or bias that carries one, inf - inf inside an accumulator).
__device__ __forceinline__ float range_mag(float a, float b) { return max_nan(fabsf(a), fabsf(b)); }
__device__ __forceinline__ bool range_bad(float v) { return !(fabsf(v) <= 65504.f); }
template <int N>
__device__ __forceinline__ void range_note_n(int* flag, const float (&v)[N]) {
  float m = fabsf(v[0]);
#pragma unroll
  for (int e = 1; e < N; ++e) m = max_nan(m, fabsf(v[e]));
  if (!(m <= 65504.f)) *flag = 1;
}
// the same for a value that comes straight from the caller's INPUT (preprocessed RGB, depth crops): word 1 of the flag
// block for a finite value beyond the fp16 range, word 2 for a non-finite one (NaN / inf pixels of a depth camera, which
// the reference passes through: ros_demo.py:227-231).  `flag` = range_flag_ptr().
__device__ __forceinline__ void range_note_input(int* flag, float v) {
  if (!(fabsf(v) <= 65504.f)) flag[fabsf(v) <= 3.402823466e38f ? 1 : 2] = 1;
}

}  // namespace hn
