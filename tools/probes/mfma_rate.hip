// Probe: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 as a function of waves per SIMD and of
// the number of independent accumulator chains, in s_memtime ticks and in wall time.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(512) k(float* out, long long* ticks, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS>
void run(int threads, int iters) {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 8 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, out, ticks, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * 16 * CHAINS;             // per wave
  const double waves_per_simd = threads / 64 / 4.0;
  const double flops = n_mfma * (threads / 64) * 256 * 4096.0;
  printf("waves/SIMD %.0f chains %d: %.1f ticks/MFMA/wave -> %.1f ticks per MFMA per SIMD; wall %.3f ms, %.1f TFLOP/s, "
         "implied tick rate %.3f GHz\n", waves_per_simd, CHAINS, h[0] / n_mfma, h[0] / n_mfma / waves_per_simd,
         ms, flops / ms * 1e-9, h[0] / (ms * 1e6));
  (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
  const int iters = 4000;
  run<1>(256, iters); run<2>(256, iters); run<4>(256, iters);
  run<1>(512, iters); run<2>(512, iters); run<4>(512, iters);
  return 0;
}
