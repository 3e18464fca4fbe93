// Which SIMD does each wave of a 512-thread block land on?  (hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip)
// Result on MI355X: waves 0-3 on four distinct SIMDs, waves 4-7 pair up with them in the same order -- what the
// wave-specialised Winograd kernels (4 GEMM + 4 transform waves) rely on.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
  extern __shared__ float lds[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  if (threadIdx.x == 9999) lds[0] = 1.f;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 150000, 0, d);
  unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d[simd %u wave %u cu %u se %u]", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15, (h[b*8+w] >> 13) & 7); printf("\n"); }
  return 0;
}
