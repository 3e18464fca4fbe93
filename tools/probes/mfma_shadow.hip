// Probe: what issues "for free" in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?
// One or two waves per SIMD stream MFMAs (2 independent chains); between consecutive MFMAs the
// wave issues K instructions of another class (VALU fma, ds_read_b128, buffer_load_dwordx4).
// Reported: s_memtime ticks per MFMA per SIMD (64.0 = the matrix pipe never waits).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { NONE = 0, VALU = 1, DSR = 2, VMEM = 3 };

template <int KIND, int K>
__global__ void __launch_bounds__(512) k(float* out, const float* src, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  v4f dsum = {0.f, 0.f, 0.f, 0.f};
  const v4f* lp = reinterpret_cast<const v4f*>(lds) + (threadIdx.x & 63);
  const v4f* gp = reinterpret_cast<const v4f*>(src) + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    v4f ld[8][K > 0 ? K : 1];             // loads land here; consumed once per 16 MFMAs
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      if (KIND == VALU) {
#pragma unroll
        for (int j = 0; j < K; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], b, a);
      } else if (KIND == DSR) {
#pragma unroll
        for (int j = 0; j < K; ++j) ld[u][j] = lp[((u * K + j) & 15) * 64];
      } else if (KIND == VMEM) {
#pragma unroll
        for (int j = 0; j < K; ++j) ld[u][j] = gp[((u * K + j) & 15) * 512];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == DSR || KIND == VMEM) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < K; ++j) dsum += ld[u][j];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = dsum[0] + dsum[1] + dsum[2] + dsum[3];
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int K>
void run(const char* name, int threads, float* out, float* src, long long* ticks) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(threads), 0, 0, out, src, ticks, iters);
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(threads), 0, 0, out, src, ticks, iters);
  (void)hipDeviceSynchronize();
  long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * 16;
  const double wps = threads / 64 / 4.0;
  long long mx = 0;
  for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("%-5s x%d per 2 MFMA, %1.0f wave(s)/SIMD: %6.1f ticks per MFMA per SIMD (slowest wave)\n", name, K, wps, mx / n_mfma / wps);
}

int main() {
  float *out, *src; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, 1 << 22); (void)hipMalloc(&ticks, 256 * 8 * 8);
  (void)hipMemset(src, 0, 1 << 22);
  for (int th = 256; th <= 512; th += 256) {
    run<NONE, 0>("none", th, out, src, ticks);
    run<VALU, 1>("valu", th, out, src, ticks); run<VALU, 2>("valu", th, out, src, ticks);
    run<VALU, 4>("valu", th, out, src, ticks); run<VALU, 8>("valu", th, out, src, ticks);
    run<VALU, 16>("valu", th, out, src, ticks);
    run<DSR, 1>("dsr", th, out, src, ticks); run<DSR, 2>("dsr", th, out, src, ticks); run<DSR, 4>("dsr", th, out, src, ticks);
    run<VMEM, 1>("vmem", th, out, src, ticks); run<VMEM, 2>("vmem", th, out, src, ticks);
  }
  return 0;
}
