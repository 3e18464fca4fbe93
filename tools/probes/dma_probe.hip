// Feasibility probe: buffer_load_dwordx4 ... lds (global -> LDS without VGPRs) on gfx950.
//   * lane-permuted sources (the XOR swizzle moves to the global side), OOB lanes must land as 0
//   * partial EXEC (16 lanes) writes only its 256 bytes
// build: hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void k(const float* src, unsigned bytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  // instr 0: all 64 lanes, lane L reads chunk (L ^ 5); lanes 8..15 forced OOB
  unsigned voff = (unsigned)((lane ^ 5) * 16);
  if (lane >= 8 && lane < 16) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds), 16, voff, 0, 0, 0);
  // instr 1: lanes 0..15 only, soffset 1024, into lds + 1024 bytes
  if (lane < 16)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 256), 16, (unsigned)(lane * 16), 1024, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  (void)hipMalloc(&d, N * 4); (void)hipMalloc(&o, 1024 * 4);
  (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, (unsigned)(N * 4), o);
  std::vector<float> r(1024);
  (void)hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int L = 0; L < 64; ++L)
    for (int e = 0; e < 4; ++e) {
      float want = (L >= 8 && L < 16) ? 0.f : (float)((L ^ 5) * 4 + e);
      if (r[L * 4 + e] != want) { if (bad < 8) printf("i0 lane %d e %d got %g want %g\n", L, e, r[L*4+e], want); ++bad; }
    }
  for (int L = 0; L < 64; ++L)
    for (int e = 0; e < 4; ++e) {
      float want = L < 16 ? (float)(256 + L * 4 + e) : -1.f;
      if (r[256 + L * 4 + e] != want) { if (bad < 16) printf("i1 lane %d e %d got %g want %g\n", L, e, r[256+L*4+e], want); ++bad; }
    }
  printf("dma_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
