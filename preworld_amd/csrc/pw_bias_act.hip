// y = act(x + bias[c]) on channels-last rows x (N, C) and its backward, for the per-voxel layers of the training path whose
// Linear / conv bias and activation torch ran as separate passes: the attribute MLPs' Linear -> Softplus (mmdet3d/models/detectors/
// preworld.py:101-110), final_conv's bias + ReLU (preworld.py:72-79), the trajectory branch's biased 2x2x2 convs.  Per step of the
// pre-train configuration that was `y + bias` (a pass), softplus (a pass), softplus_backward (a pass) and a column sum for d bias
// (a pass) per layer over up to 41 M elements.  Any C <= 256: a block covers floor(256 / C) whole rows per pass, so a thread keeps ONE
// channel (bias / partial sums in registers) and consecutive threads touch consecutive addresses.
//   act: 0 none, 1 ReLU, 2 Softplus(beta = 1, threshold = 20) -- torch.nn.Softplus' defaults
#include "pw_common.h"

namespace {
constexpr int BA_BLOCKS = 1024;

__device__ __forceinline__ float ba_act(float z, int act) {
  if (act == 1) return fmaxf(z, 0.f);
  if (act == 2) return z > 20.f ? z : log1pf(expf(z));
  return z;
}
// d act / d z
__device__ __forceinline__ float ba_dact(float z, int act) {
  if (act == 1) return z > 0.f ? 1.f : 0.f;
  if (act == 2) {
    if (z > 20.f) return 1.f;
    const float e = expf(z);
    return e / (e + 1.f);
  }
  return 1.f;
}

__global__ void __launch_bounds__(256) k_bias_act(const float* __restrict__ x, const float* __restrict__ bias, int64_t N, int C, int act,
                                                  float* __restrict__ y) {
  const int rows = 256 / C;
  if ((int)threadIdx.x >= rows * C) return;
  const int c = threadIdx.x % C, rsub = threadIdx.x / C;
  const float b = bias ? bias[c] : 0.f;
  for (int64_t r = (int64_t)blockIdx.x * rows + rsub; r < N; r += (int64_t)gridDim.x * rows) y[r * C + c] = ba_act(x[r * C + c] + b, act);
}

// dx = dy * act'(x + bias); partial[block][c] = this block's sum of dx over its rows (double)
__global__ void __launch_bounds__(256) k_bias_act_bwd(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ dy,
                                                      int64_t N, int C, int act, float* __restrict__ dx, double* __restrict__ partial) {
  __shared__ double red[256];
  const int rows = 256 / C;
  const bool on = (int)threadIdx.x < rows * C;
  const int c = threadIdx.x % C, rsub = threadIdx.x / C;
  double s = 0.0;
  if (on) {
    const float b = bias ? bias[c] : 0.f;
    for (int64_t r = (int64_t)blockIdx.x * rows + rsub; r < N; r += (int64_t)gridDim.x * rows) {
      const float g = dy[r * C + c] * ba_dact(x[r * C + c] + b, act);
      dx[r * C + c] = g;
      s += (double)g;
    }
  }
  red[threadIdx.x] = on ? s : 0.0;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double a = 0.0;
    for (int q = 0; q < rows; ++q) a += red[q * C + threadIdx.x];
    partial[(size_t)blockIdx.x * C + threadIdx.x] = a;
  }
}

__global__ void __launch_bounds__(64) k_bias_act_finish(const double* __restrict__ partial, int blocks, int C, float* __restrict__ dbias) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double a = 0.0;
  for (int b = lane; b < blocks; b += 64) a += partial[(size_t)b * C + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if (lane == 0) dbias[c] = (float)a;
}

int ba_blocks(int64_t N, int C) {
  const int64_t want = pw_cdiv(N, (int64_t)(256 / C) * 8);
  return (int)(want < 1 ? 1 : (want > BA_BLOCKS ? BA_BLOCKS : want));
}
}  // namespace

PW_API size_t pw_bias_act_workspace_bytes(int C) { return (size_t)BA_BLOCKS * (C > 0 ? C : 1) * sizeof(double) + 256; }

PW_API int pw_bias_act(const float* x, const float* bias, int64_t N, int C, int act, float* y, void* stream) {
  PW_CHECK_ARG(x && y && N > 0 && C > 0 && C <= 256 && act >= 0 && act <= 2, "pw_bias_act: bad arguments (C <= 256, act 0 | 1 | 2)");
  hipLaunchKernelGGL(k_bias_act, dim3((unsigned)ba_blocks(N, C) * 4u), dim3(256), 0, pw_stream(stream), x, bias, N, C, act, y);
  pw_note_kernel("k_bias_act");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_bias_act_backward(const float* x, const float* bias, const float* dy, int64_t N, int C, int act, float* dx, float* dbias,
                                void* workspace, size_t workspace_bytes, void* stream) {
  PW_CHECK_ARG(x && dy && dx && N > 0 && C > 0 && C <= 256 && act >= 0 && act <= 2, "pw_bias_act_backward: bad arguments");
  PW_CHECK_ARG(workspace && workspace_bytes >= pw_bias_act_workspace_bytes(C) && ((uintptr_t)workspace & 7) == 0,
               "pw_bias_act_backward: workspace too small");
  const int blocks = ba_blocks(N, C);
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_bias_act_bwd, dim3((unsigned)blocks), dim3(256), 0, st, x, bias, dy, N, C, act, dx, (double*)workspace);
  if (dbias) hipLaunchKernelGGL(k_bias_act_finish, dim3((unsigned)C), dim3(64), 0, st, (const double*)workspace, blocks, C, dbias);
  pw_note_kernel("k_bias_act_bwd");
  PW_CHECK_LAUNCH();
  return PW_OK;
}
