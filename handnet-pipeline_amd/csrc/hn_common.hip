// Error reporting, ABI version, device facts, HIP-event timing helpers, the f16x3 range-contract flag and the
// in-kernel shader-clock sampler.
#include "hn_common.h"

#include <stdlib.h>

#include <string.h>

// Sticky per-device flags of the f16x3 range contract (hn_range_check_enable): set by any producer of split
// (hi + lo fp16) data that meets a value outside the fp16 range or a non-finite one.  Word 0: an ACTIVATION (conv epilogue,
// GroupNorm apply); word 1: a finite INPUT value beyond +-65504 (preprocessed RGB, depth crop); word 2: a non-finite input
// value (a depth camera's NaN / inf pixels); word 3 unused.
__device__ int g_range_flag[4];
// The switch and the bound block are state of the calling HOST THREAD, read at LAUNCH time: an engine that opens a scope
// (hn_range_scope_begin) around its launches never changes what another thread's launches note into -- two engines driven
// from two threads of one process (a ROS node's callbacks) keep their flags apart, and an engine that runs with noting off
// does not turn it off for anybody else.
static thread_local int g_range_check_on = 0;
// hn_range_check_bind / hn_range_scope_begin: the caller's own flag block (4 device words) instead of the library's
static thread_local int* g_range_bound = nullptr;
// saved (switch, block) pairs of the open scopes of this thread
struct RangeScope { int on; int* bound; };
static thread_local RangeScope g_range_stack[8];
static thread_local int g_range_depth = 0;

namespace hn {

int* range_flag_ptr() {
  if (!g_range_check_on) return nullptr;
  if (g_range_bound) return g_range_bound;
  static int* ptrs[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!ptrs[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_range_flag)) != hipSuccess) return nullptr;
    ptrs[dev] = (int*)p;
  }
  return ptrs[dev];
}

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace hn

extern "C" int hn_abi_version(void) { return HN_ABI_VERSION; }

extern "C" int hn_debug_tickets_nonzero(int64_t* count) {
  HN_CHECK_ARG(count, "hn_debug_tickets_nonzero: null pointer");
  *count = 0;
  if (int rc = hn::tickets_nonzero_main(count)) return rc;
  return hn::tickets_nonzero_multi(count);
}

namespace hn {
// Kernel-form switches: older forms of some kernels stay in the library as bit-identity references for the tests and for
// same-box A/B timing.  The library never reads them from the environment: a development host sets them by name
// (hn_set_form; bench.py / tools translate their HN_* variables, hn_amd/forms.py).
static EnvFlags g_env = {};
const EnvFlags& env_flags() { return g_env; }
static Tuning g_tuning;
const Tuning& tuning() { return g_tuning; }
}  // namespace hn

extern "C" int hn_set_tuning(const char* name, double value) {
  HN_CHECK_ARG(name && value > 0.0, "hn_set_tuning: null name or non-positive value");
  struct Entry { const char* name; double hn::Tuning::*field; };
  static const Entry table[] = {{"splitk_fix", &hn::Tuning::splitk_fix}, {"splitk_tk", &hn::Tuning::splitk_tk},
                                {"splitk_red0", &hn::Tuning::splitk_red0}, {"splitk_plane", &hn::Tuning::splitk_plane}};
  for (const Entry& e : table)
    if (strcmp(e.name, name) == 0) {
      hn::g_tuning.*(e.field) = value;
      return HN_OK;
    }
  return hn::fail(HN_ERR_ARG, "hn_set_tuning: unknown constant '%s'", name);
}

extern "C" int hn_set_form(const char* name, int value) {
  HN_CHECK_ARG(name, "hn_set_form: null name");
  struct Entry { const char* name; bool hn::EnvFlags::*field; };
  static const Entry table[] = {
      {"conv_no_rs", &hn::EnvFlags::no_rs},                 // per-tap form of the 3x3 / stride-1 convolutions
      {"conv_no_rs32", &hn::EnvFlags::no_rs32},             // ... of the 128x32 tile only
      {"split_generic", &hn::EnvFlags::split_generic},      // generic GroupNorm-apply kernel
      {"conv_no_halo", &hn::EnvFlags::no_halo},             // implicit-GEMM form of the 64-channel 3x3 layers
      {"preprocess_generic", &hn::EnvFlags::pre_generic},   // per-pixel preprocess kernel
      {"conv_no_multi", &hn::EnvFlags::no_multi},           // hn_conv2d_nhwc_f16x3_multi: members one after the other
      {"no_fuse_last_gn", &hn::EnvFlags::no_fuse_last_gn},  // model.hip: separate last GroupNorm apply pass
      {"no_thin_outputs", &hn::EnvFlags::no_thin},          // model.hip: grouped implicit GEMM for the head outputs
      {"thin_form_tap", &hn::EnvFlags::thin_tap},           // thin kernel: never the P form
      {"thin_form_flat", &hn::EnvFlags::thin_flat},         // ... the P form at any size
      {"thin_no_group", &hn::EnvFlags::thin_nogroup},       // hn_conv3x3_thin_f16x3_levels_group: members one after the other
      {"halo_stamps", &hn::EnvFlags::halo_stamps},          // diagnostics: s_memtime stamps of the halo kernel
      {"splitk_fill512", &hn::EnvFlags::splitk_fill512},    // split-K plan of rounds 1-3 (fill 512 slots below 256 workgroups)
      {"conv_no_stream", &hn::EnvFlags::no_stream},         // implicit-GEMM form of the short-k 1x1 layers (conv1x1_stream.hip)
      {"conv_no_mixed", &hn::EnvFlags::no_mixed},           // grouped launches: one tile shape for every member
      {"conv_no_deepk", &hn::EnvFlags::no_deepk},           // small grids: the pinned one-tile-per-barrier loop
      {"conv_no_fused_reduce", &hn::EnvFlags::no_fused_reduce},   // split-K: the separate reduction launch
  };
  for (const Entry& e : table)
    if (strcmp(e.name, name) == 0) {
      hn::g_env.*(e.field) = value != 0;
      return HN_OK;
    }
  return hn::fail(HN_ERR_ARG, "hn_set_form: unknown form '%s'", name);
}

extern "C" const char* hn_last_error(void) { return hn::err_buf(); }

extern "C" int hn_device_info(int* cu_count, int* clock_khz, char* arch_name, int arch_name_len) {
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return HN_OK;
}

extern "C" int hn_device_pci_bus_id(char* out, int out_len) {
  HN_CHECK_ARG(out && out_len >= 16, "hn_device_pci_bus_id: buffer of at least 16 bytes");
  int dev = 0;
  HN_CHECK_HIP(hipGetDevice(&dev));
  HN_CHECK_HIP(hipDeviceGetPCIBusId(out, out_len, dev));
  return HN_OK;
}

extern "C" int hn_event_create(void** ev) {
  HN_CHECK_ARG(ev, "hn_event_create: null");
  hipEvent_t e;
  HN_CHECK_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return HN_OK;
}

extern "C" int hn_event_destroy(void* ev) {
  HN_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
  return HN_OK;
}

extern "C" int hn_event_record(void* ev, void* stream) {
  HN_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return HN_OK;
}

extern "C" int hn_event_elapsed_ms(void* start, void* stop, float* ms) {
  HN_CHECK_ARG(ms, "hn_event_elapsed_ms: null");
  HN_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
  HN_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return HN_OK;
}

// ---- f16x3 range contract ----
extern "C" int hn_range_check_enable(int on) {
  g_range_check_on = on ? 1 : 0;
  return HN_OK;
}

extern "C" int hn_range_check_enabled(void) { return g_range_check_on; }

extern "C" int hn_range_check_bind(int32_t* block) {
  g_range_bound = block;
  return HN_OK;
}

extern "C" int hn_range_scope_begin(int32_t* block, int on) {
  HN_CHECK_ARG(g_range_depth < 8, "hn_range_scope_begin: more than 8 nested scopes on this thread");
  g_range_stack[g_range_depth++] = RangeScope{g_range_check_on, g_range_bound};
  g_range_check_on = on ? 1 : 0;
  g_range_bound = on ? block : nullptr;
  return HN_OK;
}

extern "C" int hn_range_scope_end(void) {
  HN_CHECK_ARG(g_range_depth > 0, "hn_range_scope_end: no open scope on this thread");
  const RangeScope s = g_range_stack[--g_range_depth];
  g_range_check_on = s.on;
  g_range_bound = s.bound;
  return HN_OK;
}

extern "C" int hn_range_check_fetch(int* flag, int reset, void* stream) {
  HN_CHECK_ARG(flag, "hn_range_check_fetch: null");
  void* p = nullptr;
  int words[4] = {0, 0, 0, 0};
  HN_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_range_flag)));
  HN_CHECK_HIP(hipMemcpyAsync(words, p, sizeof(words), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HN_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  *flag = (words[0] ? HN_RANGE_ACTIVATION : 0) | (words[1] ? HN_RANGE_INPUT : 0) | (words[2] ? HN_RANGE_INPUT_NONFINITE : 0);
  if (reset) {
    HN_CHECK_HIP(hipMemsetAsync(p, 0, sizeof(words), (hipStream_t)stream));
    HN_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  return HN_OK;
}

namespace {
__global__ void range_collect_kernel(int* __restrict__ flags, int* __restrict__ dst) {
  const int t = threadIdx.x;
  if (t < 4) {
    dst[t] = flags[t];
    flags[t] = 0;
  }
}
}  // namespace

extern "C" int hn_range_check_collect(int32_t* block, int32_t* dst, void* stream) {
  HN_CHECK_ARG(dst, "hn_range_check_collect: null");
  void* p = block;
  if (!p) HN_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_range_flag)));
  hipLaunchKernelGGL(range_collect_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)p, dst);
  HN_CHECK_LAUNCH("range_collect_kernel");
  return HN_OK;
}

// ---- shader clock under load ----
// One wave compares s_memtime (shader cycles) with s_memrealtime (100 MHz) over `micros` microseconds.  Launched
// on a side stream while the kernels of interest run, it reads the clock the chip actually holds under that load
// (MI355X_MICROARCH.md, DVFS give-back (6)).  The loop ends after a bounded number of real-time ticks.
namespace {
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long ticks, float* mhz) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  for (long guard = 0; guard < (1L << 26) && r1 - r0 < ticks; ++guard) {
    __builtin_amdgcn_s_sleep(64);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) mhz[0] = (float)((double)(c1 - c0) / ((double)(r1 - r0) * 0.01));
}
}  // namespace

extern "C" int hn_clock_sample(int micros, float* mhz, void* stream) {
  HN_CHECK_ARG(mhz, "hn_clock_sample: null");
  HN_CHECK_ARG(micros > 0 && micros <= 2000000, "hn_clock_sample: micros must be in 1..2e6");
  hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)micros * 100ull, mhz);
  HN_CHECK_LAUNCH("clock_sample_kernel");
  return HN_OK;
}
