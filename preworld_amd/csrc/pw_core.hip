// Error state, version and device query for libpreworld_hip.so.
#include "pw_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void pw_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[128] = "";

void pw_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

PW_API const char* pw_last_kernel(void) { return g_kernel; }

PW_API int pw_version(void) { return 300; }

// sha256 (first 16 hex digits) of the sources this binary was built from -- written by preworld_amd/build.py just before
// compiling; __graft_entry__.smoke() recomputes it from the tree next to the library and refuses a mismatch
#include "pw_build_id.inc"
PW_API const char* pw_build_id(void) { return PW_BUILD_ID; }

PW_API const char* pw_last_error(void) { return g_err; }

PW_API int pw_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len) {
  int dev = 0;
  PW_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PW_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return PW_OK;
}
