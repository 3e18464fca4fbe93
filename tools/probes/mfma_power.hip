// Probe: what the 1400 W socket cap leaves of the fp16 matrix pipe when the operands are REAL data.
// A bare v_mfma_f32_32x32x16_f16 stream (operands in registers, no memory traffic inside the loop) with
//   mode 0: the same two operand registers every time (what mfma_f16_chain.hip measures: nothing toggles)
//   mode 1: 8 x 8 different operand registers holding random fp16 values, a different pair every MFMA
// at one and two waves per SIMD.  Prints wall TFLOP/s and the shader clock implied by the cycle counter; run it under
// tools/power_probe-style rocm-smi sampling for the socket power.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* ticks, const h8* src, int iters) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  h8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(MODE ? i : 0) * 64 + (threadIdx.x & 63)];
    b[i] = src[(8 + (MODE ? i : 0)) * 64 + (threadIdx.x & 63)];
  }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u)
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 7], b[(u >> 3) & 7], acc[u & 3], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const h8* src, int threads, double secs) {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 8 * 8);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, ticks, src, iters);
  (void)hipDeviceSynchronize();
  int reps = 0; float total = 0.f;
  while (total < secs * 1e3f) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, ticks, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); total += ms; reps += 20;
  }
  long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double ms = total / reps;
  printf("%s operands, %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD, wall %.0f TFLOP/s, implied clock %.2f GHz\n",
         MODE ? "random, a different pair every MFMA" : "constant", threads / 256, h[0] / (iters * 64.0) / (threads / 256.0),
         iters * 64.0 * (threads / 64) * 256 * 32768.0 / ms * 1e-9, h[0] / (ms * 1e6));
  fflush(stdout);
  (void)hipFree(out); (void)hipFree(ticks);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  std::vector<_Float16> hsrc(16 * 64 * 8);
  srand(1);
  for (auto& v : hsrc) {                          // ~N(0,1) / 8: random signs, exponents and mantissas
    float s = 0.f; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; 
    v = (_Float16)((s - 6.f) * 0.125f);
  }
  h8* src; (void)hipMalloc(&src, hsrc.size() * 2);
  (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 2, hipMemcpyHostToDevice);
  run<0>(src, 256, secs); run<1>(src, 256, secs); run<0>(src, 512, secs); run<1>(src, 512, secs);
  return 0;
}
