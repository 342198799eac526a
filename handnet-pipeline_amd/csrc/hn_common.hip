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
static int g_range_check_on = 0;
// hn_range_check_bind: the caller's own flag block (4 device words) instead of the library's, so that two engines of one
// process never see each other's flags; read at LAUNCH time on the host, like the switch itself
static int* g_range_bound = nullptr;

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

namespace hn {
static EnvFlags g_env;
static bool g_env_read = false;
static void read_env() {
  g_env.no_rs = getenv("HN_CONV_NO_RS") != nullptr;
  g_env.no_rs32 = getenv("HN_CONV_NO_RS32") != nullptr;
  g_env.split_generic = getenv("HN_SPLIT_GENERIC") != nullptr;
  g_env.stem_generic = getenv("HN_STEM_POOL_GENERIC") != nullptr;
  g_env.no_halo = getenv("HN_CONV_NO_HALO") != nullptr;
  g_env.pre_generic = getenv("HN_PREPROCESS_GENERIC") != nullptr;
  g_env.no_multi = getenv("HN_CONV_NO_MULTI") != nullptr;   // hn_conv2d_nhwc_f16x3_multi: members one after the other (A/B)
  g_env.no_fuse_last_gn = getenv("HN_FUSE_LAST_GN") != nullptr && getenv("HN_FUSE_LAST_GN")[0] == '0';
  g_env.no_thin = getenv("HN_THIN_OUTPUTS") != nullptr && getenv("HN_THIN_OUTPUTS")[0] == '0';   // model.hip: grouped implicit GEMM
  g_env.thin_tap = getenv("HN_THIN_FORM") != nullptr && getenv("HN_THIN_FORM")[0] == 't';          // thin kernel: never the P form
  g_env.thin_flat = getenv("HN_THIN_FORM") != nullptr && getenv("HN_THIN_FORM")[0] == 'f';         // ... the P form at any size
  g_env_read = true;
}
const EnvFlags& env_flags() {
  if (!g_env_read) read_env();
  return g_env;
}
}  // namespace hn

extern "C" int hn_reread_env(void) {
  hn::read_env();
  return HN_OK;
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
