// Error state, version and device query for libpreworld_hip.so.
#include "pw_common.h"

#include <stdlib.h>
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

// ---- measurement aid: what the socket's power cap leaves of the fp16 matrix pipe on THIS box.
// A bare v_mfma_f32_32x32x16_f16 stream -- operands in registers, 8 x 8 different register pairs of random fp16 values, no memory
// traffic in the loop -- for `seconds` of wall time at one wave per SIMD on every CU (tools/probes/mfma_power.hip is the
// stand-alone version with the constant-operand and two-waves-per-SIMD cases).  With real operand data an MI355X sustains
// ~1.7 PFLOP/s of it, not the 2.5 PFLOP/s of the data sheet (profiles/r03_power_wall.txt); bench.py prints the number next to
// `roofline.peak` so that a reader can tell a scheduling gap from the power wall.
typedef float pw_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pw_h8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) k_probe_mfma_f16(float* __restrict__ out, const pw_h8* __restrict__ src, int iters) {
  pw_f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  pw_h8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = src[i * 64 + (threadIdx.x & 63)];
    b[i] = src[(8 + i) * 64 + (threadIdx.x & 63)];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u)
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u >> 3) & 7], acc[u & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

PW_API int pw_probe_mfma_f16(double seconds, double* tflops) {
  PW_CHECK_ARG(tflops && seconds > 0 && seconds <= 10, "pw_probe_mfma_f16: bad arguments");
  int dev = 0;
  PW_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PW_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  const int nblk = prop.multiProcessorCount, iters = 2000;
  const size_t n_src = 16 * 64 * 8;
  _Float16* hsrc = (_Float16*)malloc(n_src * 2);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n_src; ++i) {                     // sum of 12 uniforms: random signs, exponents and mantissas
    float s = 0.f;
    for (int k = 0; k < 12; ++k) { st = st * 6364136223846793005ull + 1442695040888963407ull; s += (float)(st >> 40) * (1.f / 16777216.f); }
    hsrc[i] = (_Float16)((s - 6.f) * 0.125f);
  }
  pw_h8* src = nullptr; float* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PW_OK;
  do {
    if (hipMalloc(&src, n_src * 2) != hipSuccess || hipMalloc(&out, (size_t)nblk * 256 * 4) != hipSuccess ||
        hipMemcpy(src, hsrc, n_src * 2, hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess) { pw_set_error("pw_probe_mfma_f16: HIP allocation failed"); rc = PW_EHIP; break; }
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_probe_mfma_f16, dim3(nblk), dim3(256), 0, 0, out, src, iters);
    (void)hipDeviceSynchronize();
    double total_ms = 0.0; long long launches = 0;
    while (total_ms < seconds * 1e3) {
      (void)hipEventRecord(e0, 0);
      for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_probe_mfma_f16, dim3(nblk), dim3(256), 0, 0, out, src, iters);
      (void)hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) { pw_set_error("pw_probe_mfma_f16: kernel failed"); rc = PW_EHIP; break; }
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      total_ms += ms; launches += 10;
    }
    if (rc == PW_OK) *tflops = (double)launches * iters * 64.0 * 4.0 * nblk * 32768.0 / (total_ms * 1e-3) * 1e-12;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (out) (void)hipFree(out);
  free(hsrc);
  return rc;
}

// ------------------------------------------------------------------------------------
// Many small device-to-device copies in ONE launch: a sample's lifted inputs (per frame: depth, context features, five small camera
// tensors; the ego state) go into the static buffers of a captured step as 15 separate copy kernels otherwise -- most of them launch
// latency.  The segment table travels by value in the kernel arguments.
// ------------------------------------------------------------------------------------
namespace {
constexpr int CM_MAX = 32;
struct CopyMany { const char* src[CM_MAX]; char* dst[CM_MAX]; unsigned long long bytes[CM_MAX]; };

__global__ void __launch_bounds__(256) k_copy_many(CopyMany t) {
  const int seg = blockIdx.y;
  const char* s = t.src[seg];
  char* d = t.dst[seg];
  const unsigned long long n = t.bytes[seg];
  const unsigned long long n16 = ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) ? n / 16 : 0;     // 16-byte body when both are aligned
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  for (unsigned long long i = n16 * 16 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
    d[i] = s[i];
}
}  // namespace

PW_API int pw_copy_many(const void* const* src, void* const* dst, const size_t* bytes, int n, void* stream) {
  PW_CHECK_ARG(src && dst && bytes && n >= 0, "pw_copy_many: bad arguments");
  for (int i0 = 0; i0 < n; i0 += CM_MAX) {
    CopyMany t = {};
    const int m = n - i0 < CM_MAX ? n - i0 : CM_MAX;
    size_t biggest = 0;
    for (int i = 0; i < m; ++i) {
      PW_CHECK_ARG(bytes[i0 + i] == 0 || (src[i0 + i] && dst[i0 + i]), "pw_copy_many: null pointer");
      t.src[i] = (const char*)src[i0 + i]; t.dst[i] = (char*)dst[i0 + i]; t.bytes[i] = bytes[i0 + i];
      biggest = bytes[i0 + i] > biggest ? bytes[i0 + i] : biggest;
    }
    if (biggest == 0) continue;
    const int64_t want = pw_cdiv((int64_t)biggest, 256 * 16 * 4);                 // ~4 x 16 bytes per thread of the largest segment
    hipLaunchKernelGGL(k_copy_many, dim3((unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want)), (unsigned)m), dim3(256), 0, pw_stream(stream), t);
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
