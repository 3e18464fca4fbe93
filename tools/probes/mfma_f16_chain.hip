// Probe: v_mfma_f32_32x32x16_f16 issue interval vs number of independent accumulator chains (1 wave per SIMD),
// and the cost of ds_read_b128 / buffer_load_dwordx4 / SALU fillers issued between the MFMAs by the same wave.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_chain mfma_f16_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int CHAINS, int LDS_PER_12, int VMEM_PER_12, int VALU_PER_12 = 0>
__global__ void __launch_bounds__(512) k(float* out, long long* ticks, const float* gsrc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 1e-3f;
  __syncthreads();
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(1.f + e * 0.5f); }
  v4f keep = {0, 0, 0, 0};
  float vf[4] = {1.f, 2.f, 3.f, 4.f};
  const float vk = 0.999f + threadIdx.x * 1e-9f;
  const v4f* lp = reinterpret_cast<const v4f*>(lds) + (threadIdx.x & 63);
  const v4f* gp = reinterpret_cast<const v4f*>(gsrc) + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    v4f l[LDS_PER_12 > 0 ? LDS_PER_12 : 1], g[VMEM_PER_12 > 0 ? VMEM_PER_12 : 1];
#pragma unroll
    for (int q = 0; q < LDS_PER_12; ++q) l[q] = lp[q * 64 + (it & 3) * 512];
#pragma unroll
    for (int q = 0; q < VMEM_PER_12; ++q) g[q] = gp[q * 256 + (it & 7) * 2048];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % CHAINS], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < VALU_PER_12 / 12; ++q) {       // independent fp32 FMAs (4 chains) between the MFMAs
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(vf[(u * (VALU_PER_12 / 12) + q) & 3]) : "v"(vk));
      }
    }
#pragma unroll
    for (int q = 0; q < LDS_PER_12; ++q) keep += l[q];
#pragma unroll
    for (int q = 0; q < VMEM_PER_12; ++q) keep += g[q];
  }
  long long t1 = __builtin_readcyclecounter();
  float s = keep[0] + keep[1] + keep[2] + keep[3] + vf[0] + vf[1] + vf[2] + vf[3];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS, int L, int V, int A = 0>
void run(const float* gsrc, int threads = 256) {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 8 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, L, V, A>), dim3(256), dim3(threads), 0, 0, out, ticks, gsrc, iters);
  hipLaunchKernelGGL((k<CHAINS, L, V, A>), dim3(256), dim3(threads), 0, 0, out, ticks, gsrc, iters);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<CHAINS, L, V, A>), dim3(256), dim3(threads), 0, 0, out, ticks, gsrc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  (void)hipDeviceSynchronize();
  long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double wps = threads / 256.0;
  printf("%g wave(s)/SIMD, chains %d, per 12 MFMAs of a wave: %2d ds_read_b128 + %2d global dwordx4 + %2d v_fma: %.1f ticks per MFMA per SIMD, wall %.0f TFLOP/s, implied clock %.2f GHz\n",
         wps, CHAINS, L, V, A, h[0] / (iters * 12.0) / wps, iters * 12.0 * (threads / 64) * 256 * 32768.0 / ms * 1e-9, h[0] / (ms * 1e6));
  (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
  float* gsrc; (void)hipMalloc(&gsrc, 1 << 22); (void)hipMemset(gsrc, 0, 1 << 22);
  run<1, 0, 0>(gsrc); run<2, 0, 0>(gsrc); run<3, 0, 0>(gsrc); run<4, 0, 0>(gsrc);
  run<2, 8, 0>(gsrc); run<2, 8, 4>(gsrc); run<2, 0, 4>(gsrc); run<4, 8, 8>(gsrc); run<4, 8, 4>(gsrc); run<2, 16, 0>(gsrc);
  run<4, 16, 0>(gsrc); run<4, 4, 0>(gsrc); run<2, 4, 0>(gsrc);
  run<2, 0, 0>(gsrc, 512); run<2, 8, 0>(gsrc, 512); run<1, 8, 0>(gsrc, 512); run<2, 12, 0>(gsrc, 512); run<2, 16, 0>(gsrc, 512); run<2, 8, 4>(gsrc, 512); run<2, 0, 4>(gsrc, 512);
  run<1, 16, 0>(gsrc, 512); run<1, 24, 0>(gsrc, 512);
  // VALU fillers: does a wave's own VALU work hide under its MFMAs (1 wave), or only under a sibling wave's (2 waves)?
  run<4, 0, 0, 12>(gsrc); run<4, 0, 0, 36>(gsrc); run<4, 0, 0, 72>(gsrc); run<4, 8, 0, 36>(gsrc);
  run<2, 0, 0, 12>(gsrc, 512); run<2, 0, 0, 36>(gsrc, 512); run<2, 0, 0, 72>(gsrc, 512); run<2, 8, 0, 36>(gsrc, 512); run<2, 8, 4, 36>(gsrc, 512);
  return 0;
}
