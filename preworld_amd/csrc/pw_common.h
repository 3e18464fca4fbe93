// Shared helpers for libpreworld_hip.so (gfx950 only -- no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/preworld_hip.h"

#define PW_API extern "C" __attribute__((visibility("default")))
#define PW_WAVE 64

void pw_set_error(const char* fmt, ...);
// records the name (as rocprofv3 prints it) of the dominant kernel an entry point launched; read by pw_last_kernel()
void pw_note_kernel(const char* fmt, ...);

#define PW_CHECK_ARG(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      pw_set_error(__VA_ARGS__);               \
      return PW_EINVAL;                        \
    }                                          \
  } while (0)

#define PW_CHECK_HIP(expr)                                                           \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      pw_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                   __LINE__);                                                        \
      return PW_EHIP;                                                                \
    }                                                                                \
  } while (0)

#define PW_CHECK_LAUNCH() PW_CHECK_HIP(hipGetLastError())

static inline hipStream_t pw_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t pw_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

static inline size_t pw_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// exclusive int32 scan (pw_lss.hip), out may alias in; sums: pw_scan_ws_bytes(n) bytes of scratch
size_t pw_scan_ws_bytes(int64_t n);
int pw_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* sums, hipStream_t st);
