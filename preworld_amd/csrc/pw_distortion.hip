// Distortion loss of the render head (mmdet3d/models/nerf/nerf_head.py:316-327 calls torch_efficient_distloss'
// flatten_eff_distloss -- third-party, absent from the reference tree; its published algorithm, Sun et al. DVGOv2 / mip-NeRF 360
// eq. 15, restated on the dense (R, S) weights of the fused render kernel, preworld_amd.modules.NerfHead.compute_loss):
//     per ray:  uni = (1/3) interval sum_i w_i^2,   bi = 2 sum_i w_i (s_i Wpre_i - WMpre_i),
//               Wpre_i = sum_{j<i} w_j,  WMpre_i = sum_{j<i} w_j s_j
//     loss = sum over rays (uni + bi) / n_rays,   interval = 1 / n_kept (kept = samples with w > 0 over the whole batch),
//     n_rays = 1 + the last ray index that kept a sample
//     d loss / d w_k = [ (2/3) interval w_k + 2 ( s_k (Wpre_k - Wsuf_k) - (WMpre_k - WMsuf_k) ) ] / n_rays     (suf: j > k)
// As torch ops this is a dozen elementwise / cumsum passes over the 38 400 x 417 weights forward and as many backward (about 1 ms of
// the pre-train step); here one pass each way: a wave walks a ray in chunks of 64 samples with a running carry (wave-level scans),
// per-block partial sums go to a workspace and one small kernel folds them in double (deterministic, no atomics).
#include "pw_common.h"

namespace {
__device__ __forceinline__ float wave_incl_scan(float v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// partial[block] = {sum w^2, sum bi, kept samples, 1 + last kept ray of the block (0: none) AS INT BITS -- a float holds a ray index
// exactly only below 2^24}
__global__ void __launch_bounds__(256) k_distortion_fwd(const float* __restrict__ w, const float* __restrict__ s, int R, int S,
                                                        float4* __restrict__ partial) {
  __shared__ float red[4][3];
  __shared__ int last[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wave;
  float a = 0.f, b = 0.f, n = 0.f;
  if (ray < R) {
    const float* wr = w + (size_t)ray * S;
    float cw = 0.f, cm = 0.f;                       // carries: sum of w / w s over the chunks before this one
    for (int c0 = 0; c0 < S; c0 += 64) {
      const int i = c0 + lane;
      const float wi = i < S ? wr[i] : 0.f, si = i < S ? s[i] : 0.f;
      const float mi = wi * si;
      const float iw = wave_incl_scan(wi), im = wave_incl_scan(mi);
      const float wpre = cw + (iw - wi), mpre = cm + (im - mi);
      a = fmaf(wi, wi, a);
      b = fmaf(wi, si * wpre - mpre, b);
      n += wi > 0.f ? 1.f : 0.f;
      cw += __shfl(iw, 63, 64);
      cm += __shfl(im, 63, 64);
    }
  }
  a = wave_sum(a); b = wave_sum(b); n = wave_sum(n);
  if (lane == 0) { red[wave][0] = a; red[wave][1] = b; red[wave][2] = n; last[wave] = (ray < R && n > 0.f) ? ray + 1 : 0; }
  __syncthreads();
  if (threadIdx.x == 0)
    partial[blockIdx.x] = make_float4(red[0][0] + red[1][0] + red[2][0] + red[3][0], red[0][1] + red[1][1] + red[2][1] + red[3][1],
                                      red[0][2] + red[1][2] + red[2][2] + red[3][2],
                                      __int_as_float(max(max(last[0], last[1]), max(last[2], last[3]))));
}

// loss[0] = ((1/3) A / n_kept + 2 B) / n_rays;  scal = {(2/3) / n_kept / n_rays, 2 / n_rays}: the two coefficients of the gradient
__global__ void __launch_bounds__(256) k_distortion_finish(const float4* __restrict__ partial, int nb, float* __restrict__ loss,
                                                           float* __restrict__ scal) {
  __shared__ double red[3][256];
  __shared__ int rmax[256];
  double a = 0.0, b = 0.0, n = 0.0;
  int m = 0;
  for (int i = threadIdx.x; i < nb; i += 256) {
    const float4 p = partial[i];
    a += (double)p.x; b += (double)p.y; n += (double)p.z; m = max(m, __float_as_int(p.w));
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = n; rmax[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st];
      red[2][threadIdx.x] += red[2][threadIdx.x + st]; rmax[threadIdx.x] = max(rmax[threadIdx.x], rmax[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nk = red[2][0] < 1.0 ? 1.0 : red[2][0];             // kept.sum().clamp_min(1)
    const double nr = rmax[0] > 0 ? (double)rmax[0] : 1.0;         // no ray kept anything: 1
    loss[0] = (float)(((1.0 / 3.0) * red[0][0] / nk + 2.0 * red[1][0]) / nr);
    scal[0] = (float)((2.0 / 3.0) / nk / nr);
    scal[1] = (float)(2.0 / nr);
  }
}

// gw[ray][k] = gout * (scal0 w_k + scal1 (s_k (Wpre_k - Wsuf_k) - (WMpre_k - WMsuf_k))); the ray's totals come from a first sweep
__global__ void __launch_bounds__(256) k_distortion_bwd(const float* __restrict__ w, const float* __restrict__ s, int R, int S,
                                                        const float* __restrict__ scal, const float* __restrict__ gout,
                                                        float* __restrict__ gw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= R) return;
  const float* wr = w + (size_t)ray * S;
  float* gr = gw + (size_t)ray * S;
  const float c0 = scal[0] * gout[0], c1 = scal[1] * gout[0];
  float tw = 0.f, tm = 0.f;
  for (int i = lane; i < S; i += 64) { const float wi = wr[i]; tw += wi; tm = fmaf(wi, s[i], tm); }
  tw = wave_sum(tw); tm = wave_sum(tm);
  float cw = 0.f, cm = 0.f;
  for (int b0 = 0; b0 < S; b0 += 64) {
    const int i = b0 + lane;
    const float wi = i < S ? wr[i] : 0.f, si = i < S ? s[i] : 0.f;
    const float mi = wi * si;
    const float iw = wave_incl_scan(wi), im = wave_incl_scan(mi);
    const float wpre = cw + (iw - wi), mpre = cm + (im - mi);
    const float wsuf = tw - wpre - wi, msuf = tm - mpre - mi;
    if (i < S) gr[i] = fmaf(c0, wi, c1 * (si * (wpre - wsuf) - (mpre - msuf)));
    cw += __shfl(iw, 63, 64);
    cm += __shfl(im, 63, 64);
  }
}
}  // namespace

PW_API size_t pw_distortion_workspace_bytes(int n_rays) { return (size_t)pw_cdiv(n_rays > 0 ? n_rays : 1, 4) * sizeof(float4) + 256; }

PW_API int pw_distortion_loss(const float* weights, const float* s, int n_rays, int n_samples, void* workspace, size_t workspace_bytes,
                              float* loss, float* scal, void* stream) {
  PW_CHECK_ARG(weights && s && workspace && loss && scal && n_rays > 0 && n_samples > 0, "pw_distortion_loss: bad arguments");
  PW_CHECK_ARG(workspace_bytes >= pw_distortion_workspace_bytes(n_rays) && ((uintptr_t)workspace & 15) == 0,
               "pw_distortion_loss: workspace too small or misaligned");
  const int nb = (int)pw_cdiv(n_rays, 4);
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_distortion_fwd, dim3((unsigned)nb), dim3(256), 0, st, weights, s, n_rays, n_samples, (float4*)workspace);
  hipLaunchKernelGGL(k_distortion_finish, dim3(1), dim3(256), 0, st, (const float4*)workspace, nb, loss, scal);
  pw_note_kernel("k_distortion_fwd");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_distortion_loss_backward(const float* weights, const float* s, int n_rays, int n_samples, const float* scal,
                                       const float* grad_loss, float* grad_weights, void* stream) {
  PW_CHECK_ARG(weights && s && scal && grad_loss && grad_weights && n_rays > 0 && n_samples > 0, "pw_distortion_loss_backward: bad arguments");
  hipLaunchKernelGGL(k_distortion_bwd, dim3((unsigned)pw_cdiv(n_rays, 4)), dim3(256), 0, pw_stream(stream), weights, s, n_rays, n_samples,
                     scal, grad_loss, grad_weights);
  pw_note_kernel("k_distortion_bwd");
  PW_CHECK_LAUNCH();
  return PW_OK;
}
