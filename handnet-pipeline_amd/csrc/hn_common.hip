// Error reporting, ABI version, device facts and HIP-event timing helpers.
#include "hn_common.h"

#include <string.h>

namespace hn {

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
