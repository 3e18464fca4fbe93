// Volume-rendering attribute-projection head for gfx950.  Replaces (reference paths):
//   mmdet3d/models/nerf/cuda/render_utils_kernel.cu:431-443,507-517   raw2alpha (+backward)
//   mmdet3d/models/nerf/cuda/render_utils_kernel.cu:577-677           alpha2weight (+backward)
//   mmdet3d/models/nerf/cuda/ub360_utils_kernel.cu:13-47              cumdist_thres
//   mmdet3d/models/nerf/nerf_head.py:32-55,165-269,331-353            sample_ray, render_one_scene,
//                                                                     render_depth/semantic/color
// Two layers:
//  (1) the five ops with the reference's semantics on compacted point arrays, so that the
//      reference's autograd wrappers (mmdet3d/models/nerf/utils.py:26-68) can bind to them;
//  (2) pw_render_rays: the whole forward of render_one_scene + render_* fused, ONE WAVEFRONT PER
//      RAY.  The 417 candidate samples of a ray live in registers (7 per lane); the three
//      data-dependent compactions of the reference (inner|cumdist mask, alpha > 1e-7,
//      weight > 1e-7) become predicates, the two inherently sequential recurrences (cumulative
//      distance with reset, transmittance with early stop) run as wave-uniform scans over
//      v_readlane, and only surviving samples gather the 8 x 24-channel corners of the packed
//      attribute grid.  Nothing of size (rays x samples) ever reaches HBM unless asked for.
// Compiled with -ffp-contract=off: sample positions / masks follow the oracle's op order.
#include "pw_common.h"

// ------------------------------------------------------------------------------------
// (1) reference-ABI ops
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raw2alpha(const float* __restrict__ density, float shift, float interval, int64_t n,
            float* __restrict__ exp_d, float* __restrict__ alpha) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = expf(density[i] + shift);      // can be inf
  exp_d[i] = e;
  alpha[i] = 1.f - powf(1.f + e, -interval);
}

__global__ void __launch_bounds__(256)
k_raw2alpha_bwd(const float* __restrict__ exp_d, const float* __restrict__ grad_back,
                float interval, int64_t n, float* __restrict__ grad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // min(exp_d, 1e10) * pow(1+exp_d, -interval-1) * interval * grad_back, the 1e10 literal makes
  // the product double in the reference (render_utils_kernel.cu:515)
  const double m = (double)exp_d[i] < 1e10 ? (double)exp_d[i] : 1e10;
  grad[i] = (float)(m * (double)powf(1.f + exp_d[i], -interval - 1.f) * (double)interval *
                    (double)grad_back[i]);
}

// __set_i_for_segment_start_end (render_utils_kernel.cu:607-617); the host-side
// `i_end[ray_id[n-1]] = n` fix-up (:635) is the index == n_pts case here.
__global__ void __launch_bounds__(256)
k_segment_bounds(const int64_t* __restrict__ ray_id, int64_t n_pts, int64_t* __restrict__ i_start,
                 int64_t* __restrict__ i_end) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == n_pts && n_pts > 0) { i_end[ray_id[n_pts - 1]] = n_pts; return; }
  if (0 < idx && idx < n_pts && ray_id[idx] != ray_id[idx - 1]) {
    i_start[ray_id[idx]] = idx;
    i_end[ray_id[idx - 1]] = idx;
  }
}

__global__ void __launch_bounds__(256)
k_alpha2weight(const float* __restrict__ alpha, int n_rays, float* __restrict__ weight,
               float* __restrict__ T, float* __restrict__ alphainv_last,
               const int64_t* __restrict__ i_start, int64_t* __restrict__ i_end) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int i_s = (int)i_start[r], i_e_max = (int)i_end[r];
  float T_cum = 1.f;
  int i;
  for (i = i_s; i < i_e_max; ++i) {
    T[i] = T_cum;
    weight[i] = T_cum * alpha[i];
    T_cum = (float)((double)T_cum * (1. - (double)alpha[i]));   // `1. - alpha` is double (:596)
    if ((double)T_cum < 1e-3) { i += 1; break; }
  }
  i_end[r] = i;
  alphainv_last[r] = T_cum;
}

__global__ void __launch_bounds__(256)
k_alpha2weight_bwd(const float* __restrict__ alpha, const float* __restrict__ weight,
                   const float* __restrict__ T, const float* __restrict__ alphainv_last,
                   const int64_t* __restrict__ i_start, const int64_t* __restrict__ i_end,
                   int n_rays, const float* __restrict__ grad_weights,
                   const float* __restrict__ grad_last, float* __restrict__ grad) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int i_s = (int)i_start[r], i_e = (int)i_end[r];
  float back_cum = grad_last[r] * alphainv_last[r];
  for (int i = i_e - 1; i >= i_s; --i) {
    grad[i] = (float)((double)(grad_weights[i] * T[i]) -
                      (double)back_cum / (1. - (double)alpha[i] + 1e-10));
    back_cum += grad_weights[i] * weight[i];
  }
}

__global__ void __launch_bounds__(256)
k_cumdist_thres(const float* __restrict__ dist, float thres, int n_rays, int n_pts,
                uint8_t* __restrict__ mask) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  float cum = 0.f;
  const int64_t base = (int64_t)r * n_pts;
  for (int i = 0; i < n_pts; ++i) {
    cum += dist[base + i];
    const bool over = cum > thres;
    cum *= (float)(!over);
    mask[base + i] = (uint8_t)over;
  }
}

__global__ void __launch_bounds__(256) k_zero_f32(float* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

PW_API int pw_raw2alpha(const float* density, float shift, float interval, int64_t n, float* exp_d,
                        float* alpha, void* stream) {
  if (n == 0) return PW_OK;
  PW_CHECK_ARG(density && exp_d && alpha && n > 0, "pw_raw2alpha: bad arguments");
  hipLaunchKernelGGL(k_raw2alpha, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream),
                     density, shift, interval, n, exp_d, alpha);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval,
                                 int64_t n, float* grad, void* stream) {
  if (n == 0) return PW_OK;
  PW_CHECK_ARG(exp_d && grad_back && grad && n > 0, "pw_raw2alpha_backward: bad arguments");
  hipLaunchKernelGGL(k_raw2alpha_bwd, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0,
                     pw_stream(stream), exp_d, grad_back, interval, n, grad);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// weight/T/alphainv_last/i_start/i_end are initialised here (zeros/ones/ones/zeros/zeros) exactly as
// alpha2weight_cuda does (render_utils_kernel.cu:624-631)
__global__ void __launch_bounds__(256)
k_a2w_init(int64_t n_pts, int n_rays, float* weight, float* T, float* last, int64_t* i_start,
           int64_t* i_end) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pts) { weight[i] = 0.f; T[i] = 1.f; }
  if (i < n_rays) { last[i] = 1.f; i_start[i] = 0; i_end[i] = 0; }
}

PW_API int pw_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int n_rays,
                           float* weight, float* T, float* alphainv_last, int64_t* i_start,
                           int64_t* i_end, void* stream) {
  PW_CHECK_ARG(n_pts >= 0 && n_rays >= 0, "pw_alpha2weight: bad sizes");
  PW_CHECK_ARG((n_pts == 0 || (weight && T)) && (n_rays == 0 || (alphainv_last && i_start && i_end)),
               "pw_alpha2weight: null output");
  hipStream_t st = pw_stream(stream);
  int64_t m = n_pts > n_rays ? n_pts : n_rays;
  if (m == 0) return PW_OK;
  hipLaunchKernelGGL(k_a2w_init, dim3((unsigned)pw_cdiv(m, 256)), dim3(256), 0, st, n_pts, n_rays,
                     weight, T, alphainv_last, i_start, i_end);
  if (n_pts > 0) {
    PW_CHECK_ARG(alpha && ray_id, "pw_alpha2weight: null input");
    hipLaunchKernelGGL(k_segment_bounds, dim3((unsigned)pw_cdiv(n_pts + 1, 256)), dim3(256), 0, st,
                       ray_id, n_pts, i_start, i_end);
    hipLaunchKernelGGL(k_alpha2weight, dim3((unsigned)pw_cdiv(n_rays, 256)), dim3(256), 0, st, alpha,
                       n_rays, weight, T, alphainv_last, i_start, i_end);
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_alpha2weight_backward(const float* alpha, const float* weight, const float* T,
                                    const float* alphainv_last, const int64_t* i_start,
                                    const int64_t* i_end, int n_rays, const float* grad_weights,
                                    const float* grad_last, int64_t n_pts, float* grad,
                                    void* stream) {
  PW_CHECK_ARG(n_pts >= 0 && n_rays >= 0, "pw_alpha2weight_backward: bad sizes");
  if (n_pts == 0) return PW_OK;
  PW_CHECK_ARG(alpha && weight && T && alphainv_last && i_start && i_end && grad_weights &&
                   grad_last && grad,
               "pw_alpha2weight_backward: null pointer");
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)pw_cdiv(n_pts, 256)), dim3(256), 0, st, grad, (int64_t)n_pts);   // not hipMemsetAsync: see pw_lss.hip k_zero_i32
  if (n_rays > 0)
    hipLaunchKernelGGL(k_alpha2weight_bwd, dim3((unsigned)pw_cdiv(n_rays, 256)), dim3(256), 0, st,
                       alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights,
                       grad_last, grad);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_cumdist_thres(const float* dist, float thres, int n_rays, int n_pts, uint8_t* mask,
                            void* stream) {
  if (n_rays == 0 || n_pts == 0) return PW_OK;
  PW_CHECK_ARG(dist && mask && n_rays > 0 && n_pts > 0, "pw_cumdist_thres: bad arguments");
  hipLaunchKernelGGL(k_cumdist_thres, dim3((unsigned)pw_cdiv(n_rays, 256)), dim3(256), 0,
                     pw_stream(stream), dist, thres, n_rays, n_pts, mask);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// (2) fused forward: one wavefront per ray
// ------------------------------------------------------------------------------------
struct RenderArgs {
  const float* rays_o;     // (R,3)
  const float* rays_d;     // (R,3)
  const float* t;          // (S) sample distances (nerf_head.py:35-43)
  const float* grid;       // packed attribute grid (Z,Y,X,GC) channels-last
  float center[3], radius[3], bda[9], xyz_min[3], xyz_max[3];
  float bg_len, act_shift, interval, dist_thres, fast_thres, depth_scale;
  int R, S, X, Y, Z, GC, c_sigma, c_sem, n_sem, c_rgb;
  float* out_depth;        // (R)
  float* out_sem;          // (R, n_sem)
  float* out_rgb;          // (R, 3)
  float* out_last;         // (R)   alphainv_last
  int* out_counts;         // (R, 3): #masked, #alpha>thr, #weight>thr  (or null)
  float* out_weights;      // (R, S) dense per-sample weights, 0 where culled (or null)
  uint8_t* out_mask;       // (R, S) inner|cumdist sample mask (or null)
  // backward (k_render_rays<NP, true>): upstream gradients and the gradient of the packed grid
  const float* g_depth;    // (R)
  const float* g_sem;      // (R, n_sem)
  const float* g_rgb;      // (R, 3)
  const float* g_last;     // (R)   d loss / d alphainv_last
  const float* g_w;        // (R, S) d loss / d dense weights, or null
  float* grad_grid;        // (Z,Y,X,GC), ACCUMULATED into (zero it first)
  // backward, deterministic form (k_render_rays<NP, 2>): instead of scatter-adding, the march EMITS one entry per (visited
  // sample, in-bounds trilinear corner) -- see "sorted backward" below
  int2* e_kr;              // (voxel (z*Y + y)*X + x, arrival rank inside the voxel's segment = the returning histogram atomic)
  float4* e_pay;           // (ray as int bits, a = w_i * corner weight (0 for culled samples), b = d loss / d sigma_i * corner weight, 0)
  int* e_head;             // [0] number of entries, [1] bits of max |e_b|, [4] set when a reservation passed e_cap (see below)
  int e_cap;               // entries the arrays hold
  int* e_count;            // per-voxel histogram (zeroed by the host side)
};

constexpr int RPASS = 7;   // 7 x 64 = 448 >= 417 samples per ray

// trilinear corner set-up following ATen grid_sampler_3d (align_corners=True, zeros padding)
// with the reference's axis flip (nerf_head.py:211): grid axes (X,Y,Z) <-> torch (D,H,W).
struct Tri {
  int x0, y0, z0;
  float wx0, wx1, wy0, wy1, wz0, wz1;
};

__device__ __forceinline__ Tri tri_setup(const RenderArgs& a, float px, float py, float pz) {
  Tri t;
  const float gx = ((px - a.xyz_min[0]) / (a.xyz_max[0] - a.xyz_min[0])) * 2.f - 1.f;
  const float gy = ((py - a.xyz_min[1]) / (a.xyz_max[1] - a.xyz_min[1])) * 2.f - 1.f;
  const float gz = ((pz - a.xyz_min[2]) / (a.xyz_max[2] - a.xyz_min[2])) * 2.f - 1.f;
  const float fx = ((gx + 1.f) / 2.f) * (float)(a.X - 1);
  const float fy = ((gy + 1.f) / 2.f) * (float)(a.Y - 1);
  const float fz = ((gz + 1.f) / 2.f) * (float)(a.Z - 1);
  const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
  // clamp the integer base far outside the grid so the bounds test below cannot overflow
  t.x0 = (int)fminf(fmaxf(x0f, -2.f), (float)a.X + 1.f);
  t.y0 = (int)fminf(fmaxf(y0f, -2.f), (float)a.Y + 1.f);
  t.z0 = (int)fminf(fmaxf(z0f, -2.f), (float)a.Z + 1.f);
  t.wx1 = fx - x0f; t.wx0 = (x0f + 1.f) - fx;
  t.wy1 = fy - y0f; t.wy0 = (y0f + 1.f) - fy;
  t.wz1 = fz - z0f; t.wz0 = (z0f + 1.f) - fz;
  return t;
}

// iterate the 8 corners in ATen's accumulation order (torch D=our X outermost, W=our Z innermost)
#define PW_FOR_CORNERS(t, BODY)                                                        \
  _Pragma("unroll") for (int cx = 0; cx < 2; ++cx)                                     \
  _Pragma("unroll") for (int cy = 0; cy < 2; ++cy)                                     \
  _Pragma("unroll") for (int cz = 0; cz < 2; ++cz) {                                   \
    const int xi = t.x0 + cx, yi = t.y0 + cy, zi = t.z0 + cz;                          \
    const float wgt = ((cz ? t.wz1 : t.wz0) * (cy ? t.wy1 : t.wy0)) * (cx ? t.wx1 : t.wx0); \
    const bool inb = (unsigned)xi < (unsigned)a.X && (unsigned)yi < (unsigned)a.Y &&  \
                     (unsigned)zi < (unsigned)a.Z;                                     \
    const size_t cbase = (((size_t)zi * a.Y + yi) * a.X + xi) * a.GC;                  \
    BODY                                                                               \
  }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// NP = number of 64-sample passes a lane makes over its ray (2 for <= 128 samples, 4, or 7 = NP).
// Where the time goes (38 400 rays x 417 samples: 1.36 ms): the two order-dependent scans (cumdist
// reset scan, transmittance product with early stop) run as wave-uniform VALU chains, ~12 k
// instructions per ray; hoisting the corner gathers out of their bounds tests changed nothing.
// BWD = true: the same march (identical masks, compactions and early stop by construction), then the reverse pass of
// mmdet3d/models/nerf/utils.py:37-68 + render_utils_kernel.cu:507-517,654-677 + grid_sample's backward in the same wave:
//   d loss / d w_i      = g_depth s_i radius + g_sem . sem_i + g_rgb . rgb_i (+ g_w[i])        for samples with w_i > thres
//   d loss / d alpha_i  = gw_i T_i - back_cum / (1 - alpha_i + 1e-10),  back_cum += gw_i w_i   (reverse over the scanned samples)
//   d loss / d sigma_i  = min(e, 1e10) (1 + e)^(-interval-1) interval d alpha_i
// and the trilinear corner scatter-adds of (w_i g_sem, w_i g_rgb, d sigma_i) into the packed (Z,Y,X,24) gradient grid with
// hardware float atomics (summation order across rays is not deterministic, like the reference's grid_sample backward).
// BF16 = true: the packed grid is stored as bfloat16 (48 B per trilinear corner instead of 96: BASELINE.json configs[4] /
// north_star "bf16 storage, fp32 accumulate"); every value is widened to fp32 on load, all arithmetic stays fp32.
template <bool BF16>
__device__ __forceinline__ float grid_at(const float* grid, size_t idx) {
  if constexpr (BF16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(grid)[idx] << 16);
  else return grid[idx];
}

template <int NP, int BWD, bool BF16 = false>      // BWD: 0 forward, 1 backward with float atomics, 2 backward emitting entries
__global__ void __launch_bounds__(256) k_render_rays(RenderArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.R) return;
  const int S = a.S;

  // ---- A13 sample_ray (nerf_head.py:32-55): normalise, march, contract, undo bda
  float o[3], d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = (a.rays_o[ray * 3 + i] - a.center[i]) / a.radius[i];
  {
    float nn = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) nn += a.rays_d[ray * 3 + i] * a.rays_d[ray * 3 + i];
    nn = sqrtf(nn);
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = a.rays_d[ray * 3 + i] / nn;
  }
  float px[NP], py[NP], pz[NP], tt[NP], dq[NP];
  unsigned long long innerbits[NP], maskbits[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int s = p * 64 + lane;
    const bool valid = s < S;
    const float ts = a.t[valid ? s : S - 1];
    float q0 = o[0] + d[0] * ts, q1 = o[1] + d[1] * ts, q2 = o[2] + d[2] * ts;
    const float norm = sqrtf((q0 * q0 + q1 * q1) + q2 * q2);
    const bool inner = norm <= 1.f;
    if (!inner) {
      const float sc = (1.f + a.bg_len) - a.bg_len / norm;
      q0 = q0 / norm * sc; q1 = q1 / norm * sc; q2 = q2 / norm * sc;
    }
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float acc = 0.f;
      acc += a.bda[i * 3 + 0] * q0;
      acc += a.bda[i * 3 + 1] * q1;
      acc += a.bda[i * 3 + 2] * q2;
      r[i] = acc;
    }
    px[p] = r[0]; py[p] = r[1]; pz[p] = r[2]; tt[p] = ts;
    innerbits[p] = __ballot(valid && inner);
  }
  // distance to the previous sample (nerf_head.py:198): dq[s] = |p[s] - p[s-1]|, s >= 1
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float ux = __shfl_up(px[p], 1, 64), uy = __shfl_up(py[p], 1, 64), uz = __shfl_up(pz[p], 1, 64);
    // lane 0 pairs with lane 63 of the previous pass (wave-wide shuffles stay unconditional)
    const float vx = p > 0 ? __shfl(px[p > 0 ? p - 1 : 0], 63, 64) : 0.f;
    const float vy = p > 0 ? __shfl(py[p > 0 ? p - 1 : 0], 63, 64) : 0.f;
    const float vz = p > 0 ? __shfl(pz[p > 0 ? p - 1 : 0], 63, 64) : 0.f;
    if (lane == 0) { ux = vx; uy = vy; uz = vz; }
    const float ex = px[p] - ux, ey = py[p] - uy, ez = pz[p] - uz;
    dq[p] = sqrtf((ex * ex + ey * ey) + ez * ez);
  }
  // ---- A14 cumdist_thres (ub360_utils_kernel.cu:13-32): cum += d; over = cum > thres; cum *= !over.
  // The scan is order-dependent, but a sample whose own step already exceeds the threshold is
  // "over" whatever came before (cum >= 0) and leaves cum = 0 -- true for every sample of the
  // uniformly spaced inner region.  Those are found with one ballot per pass; only the runs of
  // short steps (the contracted outer shell) are walked sequentially, each run starting from the
  // exact 0 the reference would hold there: same mask, ~30 instead of 417 serial trips.
  {
    float cum = 0.f;
    bool prev_hard = true;                           // "cum == 0 on entry" (also true at sample 0)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int lim = min(64, S - p * 64);
      const unsigned long long live = lim >= 64 ? ~0ull : (lim <= 0 ? 0ull : ((1ull << lim) - 1ull));
      const unsigned long long scan = live & (p == 0 ? ~1ull : ~0ull);       // sample 0 has no step
      const unsigned long long hard = __ballot(dq[p] > a.dist_thres) & scan;
      unsigned long long over_bits = hard;
      unsigned long long soft = scan & ~hard;
      while (soft) {
        const int l = __builtin_ctzll(soft);
        soft &= soft - 1;
        // the previous scanned sample was hard (or this is the first one): cum is exactly 0
        const bool after_hard = l == 0 ? prev_hard : (((hard >> (l - 1)) & 1ull) != 0ull || (p == 0 && l == 1));
        if (after_hard) cum = 0.f;
        const float dv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dq[p]), l));
        cum += dv;
        const bool over = cum > a.dist_thres;
        cum *= (float)(!over);
        over_bits |= (unsigned long long)over << l;
      }
      if (lim > 0) prev_hard = ((hard >> (lim - 1)) & 1ull) != 0ull;
      maskbits[p] = innerbits[p] | over_bits;      // mask[:,1:] |= cumdist (nerf_head.py:199)
    }
  }
  // ---- A15/A16 density gather + raw2alpha on masked samples
  float alpha[NP], eq[NP];
  int n_mask = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool m = (maskbits[p] >> lane) & 1ull;
    n_mask += __popcll(maskbits[p]);
    float al = 0.f, ee = 0.f;
    if (m) {
      const Tri t3 = tri_setup(a, px[p], py[p], pz[p]);
      float sig = 0.f;
      PW_FOR_CORNERS(t3, { if (inb) sig += grid_at<BF16>(a.grid, cbase + a.c_sigma) * wgt; })
      const float e = expf(sig + a.act_shift);
      al = 1.f - powf(1.f + e, -a.interval);
      ee = e;
    }
    alpha[p] = al;
    eq[p] = ee;
  }
  // ---- A17 alpha2weight (render_utils_kernel.cu:577-605) over samples with alpha > thres
  float w[NP], Tq[NP];
  unsigned long long proc[NP];                 // samples the transmittance scan visited (up to and including the stopper)
#pragma unroll
  for (int p = 0; p < NP; ++p) { w[p] = 0.f; Tq[p] = 1.f; proc[p] = 0ull; }
  float T_cum = 1.f;
  int n_alpha = 0;
  bool stopped = false;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned long long abits = __ballot(((maskbits[p] >> lane) & 1ull) && alpha[p] > a.fast_thres);
    n_alpha += __popcll(abits);
    unsigned long long rem = abits;
    while (rem && !stopped) {
      const int l = __builtin_ctzll(rem);
      rem &= rem - 1;
      const float al = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha[p]), l));
      const float wq = T_cum * al;
      if (lane == l) { w[p] = wq; Tq[p] = T_cum; }
      proc[p] |= 1ull << l;
      T_cum = (float)((double)T_cum * (1. - (double)al));
      if ((double)T_cum < 1e-3) stopped = true;
    }
  }
  if constexpr (BWD != 0) {
    // ---- reverse pass
    float gsem[17], grgb[3];
#pragma unroll
    for (int k = 0; k < 17; ++k) gsem[k] = a.g_sem[(size_t)ray * a.n_sem + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) grgb[k] = a.g_rgb[(size_t)ray * 3 + k];
    const float gdep = a.g_depth[ray] * a.depth_scale;
    float gw[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      gw[p] = 0.f;
      const bool keep = w[p] > a.fast_thres;
      if (keep) {
        const float sdist = 1.f - 1.f / (1.f + tt[p]);
        const Tri t3 = tri_setup(a, px[p], py[p], pz[p]);
        float acc = gdep * sdist;
        if (a.g_w) acc += a.g_w[(size_t)ray * S + p * 64 + lane];
        PW_FOR_CORNERS(t3, {
          if (inb) {
            const float* g = a.grid + cbase;
            float* gg = a.grad_grid + cbase;
            const float ww = w[p] * wgt;
            for (int k = 0; k < 17; ++k) { acc += gsem[k] * (g[a.c_sem + k] * wgt); if constexpr (BWD == 1) unsafeAtomicAdd(gg + a.c_sem + k, ww * gsem[k]); }
            for (int k = 0; k < 3; ++k) { acc += grgb[k] * (g[a.c_rgb + k] * wgt); if constexpr (BWD == 1) unsafeAtomicAdd(gg + a.c_rgb + k, ww * grgb[k]); }
            (void)gg; (void)ww;
          }
        })
        gw[p] = acc;
      }
    }
    float back_cum = a.g_last[ray] * T_cum;               // grad_last * alphainv_last (render_utils_kernel.cu:671)
    float galpha[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) galpha[p] = 0.f;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const int p = NP - 1 - pp;
      unsigned long long rem = proc[p];
      while (rem) {
        const int l = 63 - __builtin_clzll(rem);
        rem &= ~(1ull << l);
        const float gwl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gw[p]), l));
        const float Tl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tq[p]), l));
        const float al = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha[p]), l));
        const float wl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[p]), l));
        const float g = (float)((double)(gwl * Tl) - (double)back_cum / (1. - (double)al + 1e-10));
        if (lane == l) galpha[p] = g;
        back_cum += gwl * wl;
      }
    }
    if constexpr (BWD == 1) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if ((proc[p] >> lane) & 1ull) {
          const double m = (double)eq[p] < 1e10 ? (double)eq[p] : 1e10;
          const float gsig = (float)(m * (double)powf(1.f + eq[p], -a.interval - 1.f) * (double)a.interval * (double)galpha[p]);
          const Tri t3 = tri_setup(a, px[p], py[p], pz[p]);
          PW_FOR_CORNERS(t3, { if (inb) unsafeAtomicAdd(a.grad_grid + cbase + a.c_sigma, gsig * wgt); })
        }
      }
      return;
    }
    // ---- BWD == 2: emit (voxel, ray, w * corner weight, d sigma * corner weight) for every visited sample and in-bounds corner.
    // Space in the entry arrays is reserved once per wave and pass (one atomic), the per-voxel histogram atomic returns the
    // entry's rank inside its voxel's segment, so the sort below is a plain scatter.
    unsigned bmax = 0u;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool pr = (proc[p] >> lane) & 1ull;
      float gsig = 0.f;
      if (pr) {
        const double m = (double)eq[p] < 1e10 ? (double)eq[p] : 1e10;
        gsig = (float)(m * (double)powf(1.f + eq[p], -a.interval - 1.f) * (double)a.interval * (double)galpha[p]);
      }
      const float wk = w[p] > a.fast_thres ? w[p] : 0.f;
      const Tri t3 = tri_setup(a, px[p], py[p], pz[p]);
      int cnt = 0;
      PW_FOR_CORNERS(t3, { cnt += (pr && inb) ? 1 : 0; (void)wgt; (void)cbase; })
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      const int total = __builtin_amdgcn_readlane(incl, 63);
      if (total == 0) continue;                                     // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(a.e_head, total);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + total > a.e_cap) {            // cannot happen with the caller's bound (8 corners x samples above the alpha threshold);
        if (lane == 0) a.e_head[4] = 1;        // if it does: nothing is written, the later kernels stand down and the result is poisoned
        continue;
      }
      int pos = base + incl - cnt;
      PW_FOR_CORNERS(t3, {
        // histogram atomic, one per RUN of lanes (consecutive samples of the ray, half a voxel apart) whose corner is the same
        // voxel: the run's first lane adds the run length, the others take base + offset (cf. k_hist in pw_lss.hip)
        const int v = (pr && inb) ? (zi * a.Y + yi) * a.X + xi : -1;
        const int vprev = __shfl_up(v, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || v != vprev);
        const int start = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
        const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
        const int end = above ? __builtin_ctzll(above) : 64;
        int rbase = 0;
        if (v >= 0 && lane == start) rbase = atomicAdd(a.e_count + v, end - start);
        rbase = __shfl(rbase, start, 64);
        if (v >= 0) {
          const float bb = gsig * wgt;
          a.e_kr[pos] = make_int2(v, rbase + lane - start);
          a.e_pay[pos] = make_float4(__int_as_float(ray), wk * wgt, bb, 0.f);
          bmax = max(bmax, __float_as_uint(bb) & 0x7fffffffu);
          ++pos;
        }
        (void)cbase;
      })
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) bmax = max(bmax, (unsigned)__shfl_xor((int)bmax, off, 64));
    if (lane == 0 && bmax) atomicMax(reinterpret_cast<unsigned*>(a.e_head) + 1, bmax);
    return;
  }
  // ---- A18 render_depth/semantic/color over samples with weight > thres
  float acc_d = 0.f, acc_rgb[3] = {0.f, 0.f, 0.f};
  float acc_sem[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) acc_sem[k] = 0.f;
  int n_w = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool keep = w[p] > a.fast_thres;
    n_w += __popcll(__ballot(keep));
    if (a.out_weights && p * 64 + lane < S) a.out_weights[(size_t)ray * S + p * 64 + lane] = keep ? w[p] : 0.f;
    if (a.out_mask && p * 64 + lane < S) a.out_mask[(size_t)ray * S + p * 64 + lane] = (maskbits[p] >> lane) & 1ull;
    if (keep) {
      const float sdist = 1.f - 1.f / (1.f + tt[p]);            // s = 1 - 1/(1+t) (nerf_head.py:256)
      acc_d += w[p] * sdist;
      const Tri t3 = tri_setup(a, px[p], py[p], pz[p]);
      float sem[17], rgb[3];
#pragma unroll
      for (int k = 0; k < 17; ++k) sem[k] = 0.f;
      rgb[0] = rgb[1] = rgb[2] = 0.f;
      PW_FOR_CORNERS(t3, {
        if (inb) {
          for (int k = 0; k < 17; ++k) sem[k] += grid_at<BF16>(a.grid, cbase + a.c_sem + k) * wgt;
          for (int k = 0; k < 3; ++k) rgb[k] += grid_at<BF16>(a.grid, cbase + a.c_rgb + k) * wgt;
        }
      })
#pragma unroll
      for (int k = 0; k < 17; ++k) acc_sem[k] += w[p] * sem[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc_rgb[k] += w[p] * rgb[k];
    }
  }
  acc_d = wave_sum(acc_d);
#pragma unroll
  for (int k = 0; k < 17; ++k) acc_sem[k] = wave_sum(acc_sem[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) acc_rgb[k] = wave_sum(acc_rgb[k]);
  if (lane == 0) {
    a.out_depth[ray] = (acc_d + 1e-7f) * a.depth_scale;         // (+1e-7) * radius (nerf_head.py:337-338)
    a.out_last[ray] = T_cum;
    if (a.out_counts) {
      a.out_counts[ray * 3 + 0] = n_mask;
      a.out_counts[ray * 3 + 1] = n_alpha;
      a.out_counts[ray * 3 + 2] = n_w;
    }
  }
  if (lane < 17 && lane < a.n_sem) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 17; ++k) v = (lane == k) ? acc_sem[k] : v;
    a.out_sem[(size_t)ray * a.n_sem + lane] = v;
  }
  if (lane < 3) {
    float v = lane == 0 ? acc_rgb[0] : (lane == 1 ? acc_rgb[1] : acc_rgb[2]);
    a.out_rgb[(size_t)ray * 3 + lane] = v;
  }
}

PW_API int pw_render_rays(const float* rays_o, const float* rays_d, int n_rays, const float* t,
                          int n_samples, const float* grid, int X, int Y, int Z, int grid_channels,
                          int c_sigma, int c_sem, int n_sem, int c_rgb,
                          const float* consts_host /* 3 center, 3 radius, 9 bda, 3 xyz_min, 3 xyz_max,
                          bg_len, act_shift, interval, dist_thres, fast_thres, depth_scale = 27 floats */,
                          float* out_depth, float* out_sem, float* out_rgb, float* out_last,
                          int32_t* out_counts, float* out_weights, uint8_t* out_mask, int grid_bf16, void* stream) {
  if (n_rays == 0) return PW_OK;
  PW_CHECK_ARG(rays_o && rays_d && t && grid && consts_host && out_depth && out_sem && out_rgb &&
                   out_last,
               "pw_render_rays: null pointer");
  PW_CHECK_ARG(n_rays > 0 && n_samples > 1 && n_samples <= RPASS * 64,
               "pw_render_rays: n_samples must be in [2, %d]", RPASS * 64);
  PW_CHECK_ARG(X > 1 && Y > 1 && Z > 1 && grid_channels > 0, "pw_render_rays: bad grid");
  PW_CHECK_ARG(n_sem == 17, "pw_render_rays: built for 17 semantic classes (got %d)", n_sem);
  PW_CHECK_ARG(c_sigma >= 0 && c_sigma < grid_channels && c_sem >= 0 && c_sem + n_sem <= grid_channels &&
                   c_rgb >= 0 && c_rgb + 3 <= grid_channels,
               "pw_render_rays: channel offsets outside the packed grid");
  RenderArgs a;
  a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.grid = grid;
  const float* c = consts_host;
  for (int i = 0; i < 3; ++i) { a.center[i] = c[i]; a.radius[i] = c[3 + i]; a.xyz_min[i] = c[15 + i]; a.xyz_max[i] = c[18 + i]; }
  for (int i = 0; i < 9; ++i) a.bda[i] = c[6 + i];
  a.bg_len = c[21]; a.act_shift = c[22]; a.interval = c[23]; a.dist_thres = c[24]; a.fast_thres = c[25];
  a.depth_scale = c[26];
  a.R = n_rays; a.S = n_samples; a.X = X; a.Y = Y; a.Z = Z; a.GC = grid_channels;
  a.c_sigma = c_sigma; a.c_sem = c_sem; a.n_sem = n_sem; a.c_rgb = c_rgb;
  a.out_depth = out_depth; a.out_sem = out_sem; a.out_rgb = out_rgb; a.out_last = out_last;
  a.out_counts = out_counts; a.out_weights = out_weights; a.out_mask = out_mask;
  const dim3 grid_dim((unsigned)pw_cdiv(n_rays, 4));
  if (grid_bf16) {
    if (n_samples <= 128) hipLaunchKernelGGL((k_render_rays<2, 0, true>), grid_dim, dim3(256), 0, pw_stream(stream), a);
    else if (n_samples <= 256) hipLaunchKernelGGL((k_render_rays<4, 0, true>), grid_dim, dim3(256), 0, pw_stream(stream), a);
    else hipLaunchKernelGGL((k_render_rays<RPASS, 0, true>), grid_dim, dim3(256), 0, pw_stream(stream), a);
  } else {
    if (n_samples <= 128) hipLaunchKernelGGL((k_render_rays<2, 0>), grid_dim, dim3(256), 0, pw_stream(stream), a);
    else if (n_samples <= 256) hipLaunchKernelGGL((k_render_rays<4, 0>), grid_dim, dim3(256), 0, pw_stream(stream), a);
    else hipLaunchKernelGGL((k_render_rays<RPASS, 0>), grid_dim, dim3(256), 0, pw_stream(stream), a);
  }
  pw_note_kernel("k_render_rays<%d, 0, %s>", n_samples <= 128 ? 2 : (n_samples <= 256 ? 4 : RPASS), grid_bf16 ? "true" : "false");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_render_rays_backward(const float* rays_o, const float* rays_d, int n_rays, const float* t, int n_samples,
                                   const float* grid, int X, int Y, int Z, int grid_channels, int c_sigma, int c_sem,
                                   int n_sem, int c_rgb, const float* consts_host, const float* g_depth, const float* g_sem,
                                   const float* g_rgb, const float* g_last, const float* g_weights, float* grad_grid,
                                   void* stream) {
  if (n_rays == 0) return PW_OK;
  PW_CHECK_ARG(rays_o && rays_d && t && grid && consts_host && g_depth && g_sem && g_rgb && g_last && grad_grid,
               "pw_render_rays_backward: null pointer");
  PW_CHECK_ARG(n_rays > 0 && n_samples > 1 && n_samples <= RPASS * 64, "pw_render_rays_backward: n_samples must be in [2, %d]", RPASS * 64);
  PW_CHECK_ARG(X > 1 && Y > 1 && Z > 1 && grid_channels > 0 && n_sem == 17, "pw_render_rays_backward: bad grid / n_sem");
  PW_CHECK_ARG(c_sigma >= 0 && c_sigma < grid_channels && c_sem >= 0 && c_sem + n_sem <= grid_channels && c_rgb >= 0 &&
                   c_rgb + 3 <= grid_channels, "pw_render_rays_backward: channel offsets outside the packed grid");
  RenderArgs a = {};
  a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.grid = grid;
  const float* c = consts_host;
  for (int i = 0; i < 3; ++i) { a.center[i] = c[i]; a.radius[i] = c[3 + i]; a.xyz_min[i] = c[15 + i]; a.xyz_max[i] = c[18 + i]; }
  for (int i = 0; i < 9; ++i) a.bda[i] = c[6 + i];
  a.bg_len = c[21]; a.act_shift = c[22]; a.interval = c[23]; a.dist_thres = c[24]; a.fast_thres = c[25];
  a.depth_scale = c[26];
  a.R = n_rays; a.S = n_samples; a.X = X; a.Y = Y; a.Z = Z; a.GC = grid_channels;
  a.c_sigma = c_sigma; a.c_sem = c_sem; a.n_sem = n_sem; a.c_rgb = c_rgb;
  a.g_depth = g_depth; a.g_sem = g_sem; a.g_rgb = g_rgb; a.g_last = g_last; a.g_w = g_weights; a.grad_grid = grad_grid;
  const dim3 grid_dim((unsigned)pw_cdiv(n_rays, 4));
  if (n_samples <= 128) hipLaunchKernelGGL((k_render_rays<2, 1>), grid_dim, dim3(256), 0, pw_stream(stream), a);
  else if (n_samples <= 256) hipLaunchKernelGGL((k_render_rays<4, 1>), grid_dim, dim3(256), 0, pw_stream(stream), a);
  else hipLaunchKernelGGL((k_render_rays<RPASS, 1>), grid_dim, dim3(256), 0, pw_stream(stream), a);
  pw_note_kernel("k_render_rays<%d, 1, false>", n_samples <= 128 ? 2 : (n_samples <= 256 ? 4 : RPASS));
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// (2b) sorted backward: deterministic, no float atomics  -- entry point pw_render_rays_backward_sorted
// The gradient of the packed grid is  grad[v, :] = sum over (sample i of ray r, corner hitting v) of
//     [ d sigma_i * wgt ,  (w_i * wgt) * g_sem[r, 0..16] ,  (w_i * wgt) * g_rgb[r, 0..2] ]
// -- per (ray, voxel) only TWO scalars; the 20 semantic / colour channels are their outer product with the ray's upstream
// gradient.  The scatter form (k_render_rays<NP, 1>) issues 168 global float atomics per kept sample, 1.3 G of them at
// 38 400 x 417 on a dense grid (54 ms, arrival order decides the last bits).  Here
//   1. the march (k_render_rays<NP, 2>) emits (voxel, ray, a = w wgt, b = d sigma wgt) per visited sample and in-bounds corner,
//      one int histogram atomic each -- its return value is the entry's rank inside the voxel's segment;
//   2. an exclusive scan of the histogram gives the segment starts;
//   3. k_rb_scatter moves every entry to start[voxel] + rank  (counting sort by voxel, one pass, no comparison);
//   4. k_rb_gather: one wave per voxel, lane = (channel 0..20, entry j mod 3): term = b  or  a * g[ray][channel], converted
//      to 64-bit FIXED POINT and summed in integers -- exact, hence independent of the order the atomics produced -- and
//      written once by the voxel's only writer.  Voxels with more than RB_LONG entries (the cells around the cameras, 10^4..10^5
//      entries each) are split over many waves whose integer partial sums meet through int64 atomics (still order-free).
// Fixed-point units: 2^-40 of a power of two above the largest possible term of the channel group (max |g| for the semantic /
// colour channels since a <= 1, the recorded max |b| for sigma): 2^22 terms fit, and a term 2^-16 of the largest still carries
// a full fp32 significand.
// ------------------------------------------------------------------------------------
namespace {
constexpr int RB_LONG = 6144;          // entries per wave-chunk of a long voxel
constexpr int RB_CH = 21;              // sigma + 17 semantic + 3 colour

struct RbArgs {
  const int2* kr; const float4* pay;                           // unsorted entries (see RenderArgs::e_kr / e_pay)
  float4* sorted;                                              // payloads sorted by voxel
  const int* head;            // [0] n entries, [1] bits of max |b|, [2] n_long (written by k_rb_scatter)
  int* head_w;
  const int* seg_start;       // [n_vox + 1]
  int* long_list;
  int* touched;               // voxels with 1 .. RB_LONG entries (head[3] of them): the gather's work list
  long long* long_acc;        // [max_long][RB_CH] int64, zeroed
  const float* g;             // (R, 20) = [g_sem | g_rgb]
  const float* gmax;          // device scalar: max |g|
  float* grad_grid;
  int n_vox, GC, c_sigma, c_sem, c_rgb, max_long;
};

__device__ __forceinline__ int rb_channel_index(const RbArgs& a, int ch) {
  return ch == 0 ? a.c_sigma : (ch <= 17 ? a.c_sem + ch - 1 : a.c_rgb + ch - 18);
}
// power-of-two scale that maps |x| <= m to < 2^40
__device__ __forceinline__ double rb_scale(float m) {
  if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0;
  int ex;
  (void)frexpf(m, &ex);
  return ldexp(1.0, 40 - ex);
}
}  // namespace

__global__ void __launch_bounds__(256) k_rb_scatter(RbArgs a) {
  if (a.head[4]) return;                                       // entry arrays overflowed: stand down (k_rb_finish_long poisons the result)
  const int n = a.head[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int2 kr = a.kr[i];
    const int v = kr.x, r = kr.y;
    const int s = a.seg_start[v];
    a.sorted[s + r] = a.pay[i];                                // one 16-byte store per entry
    if (r == 0) {                                              // the first arrival registers its voxel
      if (a.seg_start[v + 1] - s > RB_LONG) {
        const int li = atomicAdd(a.head_w + 2, 1);
        if (li < a.max_long) a.long_list[li] = v;
      } else {
        a.touched[atomicAdd(a.head_w + 3, 1)] = v;
      }
    }
  }
}

// sum of the entries [lo, hi) of one voxel for this lane's (channel, sub-entry): integer units.  Four entries' loads are issued
// before the first is used (the chain entry -> g[ray] is two dependent round trips; one entry at a time was latency-bound)
__device__ __forceinline__ long long rb_term(const RbArgs& a, const float4& e, int ch, double sc_g, double sc_b) {
  double term;
  if (ch == 0) term = (double)e.z * sc_b;
  else term = (double)(e.y * a.g[(size_t)__float_as_int(e.x) * 20 + ch - 1]) * sc_g;
  return __double2ll_rn(term);
}
__device__ __forceinline__ long long rb_partial(const RbArgs& a, int lo, int hi, int ch, int sub, double sc_g, double sc_b) {
  long long acc = 0;
  int i = lo + sub;
  for (; i + 9 < hi; i += 12) {
    const float4 e0 = a.sorted[i], e1 = a.sorted[i + 3], e2 = a.sorted[i + 6], e3 = a.sorted[i + 9];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (ch) {
      g0 = a.g[(size_t)__float_as_int(e0.x) * 20 + ch - 1]; g1 = a.g[(size_t)__float_as_int(e1.x) * 20 + ch - 1];
      g2 = a.g[(size_t)__float_as_int(e2.x) * 20 + ch - 1]; g3 = a.g[(size_t)__float_as_int(e3.x) * 20 + ch - 1];
    }
    acc += __double2ll_rn(ch ? (double)(e0.y * g0) * sc_g : (double)e0.z * sc_b);
    acc += __double2ll_rn(ch ? (double)(e1.y * g1) * sc_g : (double)e1.z * sc_b);
    acc += __double2ll_rn(ch ? (double)(e2.y * g2) * sc_g : (double)e2.z * sc_b);
    acc += __double2ll_rn(ch ? (double)(e3.y * g3) * sc_g : (double)e3.z * sc_b);
  }
  for (; i < hi; i += 3) acc += rb_term(a, a.sorted[i], ch, sc_g, sc_b);
  return acc;
}

__device__ __forceinline__ long long rb_shfl_down(long long v, int d) {
  const int lo = __shfl_down((int)(v & 0xffffffffll), d, 64), hi = __shfl_down((int)(v >> 32), d, 64);
  return ((long long)hi << 32) | (unsigned)lo;
}

__global__ void __launch_bounds__(256) k_rb_gather(RbArgs a) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / RB_CH, ch = lane - sub * RB_CH;      // lane 63: sub = 3, idle
  const double sc_g = rb_scale(*a.gmax), sc_b = rb_scale(__uint_as_float((unsigned)a.head[1]));
  const int nt = a.head[3];
  // one wave per touched voxel (the list's order is arbitrary; every voxel has exactly one writer and an order-free sum)
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nt; k += gridDim.x * 4) {
    const int v = a.touched[k];
    const int s = a.seg_start[v], e = a.seg_start[v + 1];
    long long acc = sub < 3 ? rb_partial(a, s, e, ch, sub, sc_g, sc_b) : 0;
    acc += rb_shfl_down(acc, RB_CH) + rb_shfl_down(acc, 2 * RB_CH);
    if (lane < RB_CH) {
      float* dst = a.grad_grid + (size_t)v * a.GC + rb_channel_index(a, ch);
      *dst += (float)((double)acc / (ch == 0 ? sc_b : sc_g));
    }
  }
}

__global__ void __launch_bounds__(256) k_rb_gather_long(RbArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const int nl = min(a.head[2], a.max_long);
  const int sub = lane / RB_CH, ch = lane - sub * RB_CH;
  const double sc_g = rb_scale(*a.gmax), sc_b = rb_scale(__uint_as_float((unsigned)a.head[1]));
  for (int li = 0; li < nl; ++li) {
    const int v = a.long_list[li];
    const int s = a.seg_start[v], e = a.seg_start[v + 1];
    // chunk c of long voxel li belongs to wave (c + 61 li) mod nw: most long voxels have a handful of chunks, so without the
    // rotation the first few waves would own all of them (measured: 185 ms instead of 4)
    const int first = (int)(((long long)wid - 61ll * li) % nw + nw) % nw;
    for (int c0 = s + first * RB_LONG; c0 < e; c0 += nw * RB_LONG) {
      long long acc = sub < 3 ? rb_partial(a, c0, min(c0 + RB_LONG, e), ch, sub, sc_g, sc_b) : 0;
      acc += rb_shfl_down(acc, RB_CH) + rb_shfl_down(acc, 2 * RB_CH);
      if (lane < RB_CH) atomicAdd(reinterpret_cast<unsigned long long*>(a.long_acc + (size_t)li * RB_CH + ch), (unsigned long long)acc);
    }
  }
}

__global__ void __launch_bounds__(256) k_rb_finish_long(RbArgs a) {
  const int nl = min(a.head[2], a.max_long);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the caller's entry bound was too small (a bug, not a data condition): no partial gradient passes for a result
  if (i == 0 && a.head[4]) a.grad_grid[0] = __uint_as_float(0x7fc00000u);
  if (i >= nl * RB_CH) return;
  const int li = i / RB_CH, ch = i - li * RB_CH;
  const double sc = ch == 0 ? rb_scale(__uint_as_float((unsigned)a.head[1])) : rb_scale(*a.gmax);
  a.grad_grid[(size_t)a.long_list[li] * a.GC + rb_channel_index(a, ch)] += (float)((double)a.long_acc[i] / sc);
}

// entries the arrays are laid out for: the caller's bound when it gives one, else every sample of every ray with 8 in-bounds corners
static size_t rb_capacity(int n_rays, int n_samples, int64_t max_entries) {
  const size_t worst = (size_t)n_rays * n_samples * 8;
  return max_entries > 0 && (size_t)max_entries < worst ? (size_t)max_entries : worst;
}

PW_API size_t pw_render_backward_workspace_bytes(int n_rays, int n_samples, int X, int Y, int Z, int64_t max_entries) {
  const size_t cap = rb_capacity(n_rays, n_samples, max_entries), nv = (size_t)X * Y * Z;
  const size_t max_long = cap / RB_LONG + 1;
  return 256 + 3 * pw_align_up((nv + 1) * 4, 256) + pw_scan_ws_bytes((int64_t)nv + 1) + 10 * pw_align_up(cap * 4, 256) +
         pw_align_up(max_long * 4, 256) + pw_align_up(max_long * RB_CH * 8, 256);
}

// Same contract as pw_render_rays_backward (grad_grid is ACCUMULATED into), plus: g_semrgb = the (R, 20) row-wise
// concatenation [g_sem | g_rgb], g_absmax = device scalar max |g_semrgb|, workspace of pw_render_backward_workspace_bytes.
// Bit-reproducible from run to run.
PW_API int pw_render_rays_backward_sorted(const float* rays_o, const float* rays_d, int n_rays, const float* t, int n_samples,
                                          const float* grid, int X, int Y, int Z, int grid_channels, int c_sigma, int c_sem,
                                          int n_sem, int c_rgb, const float* consts_host, const float* g_depth, const float* g_sem,
                                          const float* g_rgb, const float* g_last, const float* g_weights,
                                          const float* g_semrgb, const float* g_absmax, void* workspace, size_t workspace_bytes,
                                          int64_t max_entries, float* grad_grid, void* stream) {
  if (n_rays == 0) return PW_OK;
  PW_CHECK_ARG(rays_o && rays_d && t && grid && consts_host && g_depth && g_sem && g_rgb && g_last && grad_grid && g_semrgb &&
                   g_absmax && workspace, "pw_render_rays_backward_sorted: null pointer");
  PW_CHECK_ARG(n_rays > 0 && n_samples > 1 && n_samples <= RPASS * 64, "pw_render_rays_backward_sorted: n_samples must be in [2, %d]", RPASS * 64);
  PW_CHECK_ARG(X > 1 && Y > 1 && Z > 1 && grid_channels > 0 && n_sem == 17, "pw_render_rays_backward_sorted: bad grid / n_sem");
  PW_CHECK_ARG(c_sigma >= 0 && c_sigma < grid_channels && c_sem >= 0 && c_sem + n_sem <= grid_channels && c_rgb >= 0 &&
                   c_rgb + 3 <= grid_channels, "pw_render_rays_backward_sorted: channel offsets outside the packed grid");
  PW_CHECK_ARG((size_t)n_rays * n_samples * 8 < (1ull << 31) && (size_t)X * Y * Z < (1ull << 31), "pw_render_rays_backward_sorted: too many entries for int32 indices");
  PW_CHECK_ARG(workspace_bytes >= pw_render_backward_workspace_bytes(n_rays, n_samples, X, Y, Z, max_entries) && ((uintptr_t)workspace & 255) == 0,
               "pw_render_rays_backward_sorted: workspace too small or not 256-byte aligned");
  hipStream_t st = pw_stream(stream);
  const size_t cap = rb_capacity(n_rays, n_samples, max_entries), nv = (size_t)X * Y * Z;
  const size_t max_long = cap / RB_LONG + 1;
  char* ws = (char*)workspace;
  int* head = (int*)ws; ws += 256;
  int* count = (int*)ws; ws += pw_align_up((nv + 1) * 4, 256);
  int* seg = (int*)ws; ws += pw_align_up((nv + 1) * 4, 256);
  int* touched = (int*)ws; ws += pw_align_up((nv + 1) * 4, 256);
  int* sums = (int*)ws; ws += pw_scan_ws_bytes((int64_t)nv + 1);
  int2* e_kr = (int2*)ws; ws += 2 * pw_align_up(cap * 4, 256);
  float4* e_pay = (float4*)ws; ws += 4 * pw_align_up(cap * 4, 256);
  float4* sorted = (float4*)ws; ws += 4 * pw_align_up(cap * 4, 256);
  int* long_list = (int*)ws; ws += pw_align_up(max_long * 4, 256);
  long long* long_acc = (long long*)ws;
  // zero: head, histogram, the long voxels' accumulators (kernels, not memset nodes: DESIGN.md 4.6)
  hipLaunchKernelGGL(k_zero_f32, dim3(1), dim3(64), 0, st, (float*)head, (int64_t)64);
  hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)pw_cdiv((int64_t)nv + 1, 256)), dim3(256), 0, st, (float*)count, (int64_t)nv + 1);
  hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)pw_cdiv((int64_t)max_long * RB_CH * 2, 256)), dim3(256), 0, st, (float*)long_acc,
                     (int64_t)max_long * RB_CH * 2);
  RenderArgs a = {};
  a.rays_o = rays_o; a.rays_d = rays_d; a.t = t; a.grid = grid;
  const float* c = consts_host;
  for (int i = 0; i < 3; ++i) { a.center[i] = c[i]; a.radius[i] = c[3 + i]; a.xyz_min[i] = c[15 + i]; a.xyz_max[i] = c[18 + i]; }
  for (int i = 0; i < 9; ++i) a.bda[i] = c[6 + i];
  a.bg_len = c[21]; a.act_shift = c[22]; a.interval = c[23]; a.dist_thres = c[24]; a.fast_thres = c[25];
  a.depth_scale = c[26];
  a.R = n_rays; a.S = n_samples; a.X = X; a.Y = Y; a.Z = Z; a.GC = grid_channels;
  a.c_sigma = c_sigma; a.c_sem = c_sem; a.n_sem = n_sem; a.c_rgb = c_rgb;
  a.g_depth = g_depth; a.g_sem = g_sem; a.g_rgb = g_rgb; a.g_last = g_last; a.g_w = g_weights; a.grad_grid = grad_grid;
  a.e_kr = e_kr; a.e_pay = e_pay;
  a.e_head = head; a.e_count = count; a.e_cap = (int)cap;
  const dim3 grid_dim((unsigned)pw_cdiv(n_rays, 4));
  if (n_samples <= 128) hipLaunchKernelGGL((k_render_rays<2, 2>), grid_dim, dim3(256), 0, st, a);
  else if (n_samples <= 256) hipLaunchKernelGGL((k_render_rays<4, 2>), grid_dim, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_render_rays<RPASS, 2>), grid_dim, dim3(256), 0, st, a);
  pw_note_kernel("k_render_rays<%d, 2, false>", n_samples <= 128 ? 2 : (n_samples <= 256 ? 4 : RPASS));
  if (int rc = pw_scan_exclusive_i32(count, seg, (int64_t)nv + 1, sums, st)) return rc;
  RbArgs r = {};
  r.kr = e_kr; r.pay = e_pay; r.sorted = sorted;
  r.head = head; r.head_w = head; r.seg_start = seg; r.long_list = long_list; r.touched = touched; r.long_acc = long_acc;
  r.g = g_semrgb; r.gmax = g_absmax; r.grad_grid = grad_grid;
  r.n_vox = (int)nv; r.GC = grid_channels; r.c_sigma = c_sigma; r.c_sem = c_sem; r.c_rgb = c_rgb; r.max_long = (int)max_long;
  hipLaunchKernelGGL(k_rb_scatter, dim3(4096), dim3(256), 0, st, r);
  hipLaunchKernelGGL(k_rb_gather, dim3((unsigned)std::min<int64_t>(pw_cdiv((int64_t)nv, 4), 8192)), dim3(256), 0, st, r);
  hipLaunchKernelGGL(k_rb_gather_long, dim3(1024), dim3(256), 0, st, r);
  hipLaunchKernelGGL(k_rb_finish_long, dim3((unsigned)pw_cdiv((int64_t)max_long * RB_CH, 256)), dim3(256), 0, st, r);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// Ray table + WRS weights of the pre-train dataloader (mmdet3d/datasets/ray.py:34-119).
// One thread per labelled pixel; the 64-byte ray row leaves as four float4 stores.
// (This file is compiled with -ffp-contract=off: products and sums round like the torch ops.)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pts2ray(const float* __restrict__ coor, const float* __restrict__ depth, const float* __restrict__ seg,
          const float* __restrict__ img, const float* __restrict__ c2w, const float* __restrict__ K,
          int64_t n, float4* __restrict__ rays) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = coor[i * 2], y = coor[i * 2 + 1];
  const float d0 = ((x + 0.5f) - K[2]) / K[0];
  const float d1 = ((y + 0.5f) - K[5]) / K[4];
  const float d2 = 1.f;
  float rd[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) rd[k] = (d0 * c2w[k * 4 + 0] + d1 * c2w[k * 4 + 1]) + d2 * c2w[k * 4 + 2];
  const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
  rays[i * 4 + 0] = make_float4(x, y, depth[i], seg[i]);
  rays[i * 4 + 1] = make_float4(c2w[3], c2w[7], c2w[11], rd[0]);
  rays[i * 4 + 2] = make_float4(rd[1], rd[2], rd[0] / nrm, rd[1] / nrm);
  rays[i * 4 + 3] = make_float4(rd[2] / nrm, img[i * 3], img[i * 3 + 1], img[i * 3 + 2]);
}

__global__ void __launch_bounds__(256)
k_class_count(const float* __restrict__ rays, int64_t n, int n_cls, unsigned long long* __restrict__ counts) {
  __shared__ unsigned int local[64];
  if (threadIdx.x < 64) local[threadIdx.x] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float c = rays[i * 16 + 3];
    const int ci = (int)c;
    if (ci >= 0 && ci < n_cls && (float)ci == c) atomicAdd(&local[ci], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < n_cls && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
k_wrs_weights(const float* __restrict__ rays, int64_t n, int frame_id, const float* __restrict__ bw, int n_cls,
              const int32_t* __restrict__ dyn, int n_dyn, float w_adj, float w_dyn, float* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c = rays[i * 16 + 3];
  float wt = 1.f;
  if (frame_id != 0) {
    wt = w_adj;
    for (int k = 0; k < n_dyn; ++k)
      if ((float)dyn[k] == c) wt = w_dyn;
  }
  const int ci = min(max((int)c, 0), n_cls - 1);          // .long() truncation (ray.py:108)
  w[i] = bw[ci] * wt;
}

PW_API int pw_pts2ray(const float* coor, const float* depth, const float* seg, const float* img,
                      const float* c2w, const float* K, int64_t n, float* rays, void* stream) {
  if (n == 0) return PW_OK;
  PW_CHECK_ARG(coor && depth && seg && img && c2w && K && rays && n > 0, "pw_pts2ray: bad arguments");
  PW_CHECK_ARG(((uintptr_t)rays & 15) == 0, "pw_pts2ray: rays must be 16-B aligned");
  hipLaunchKernelGGL(k_pts2ray, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream), coor, depth,
                     seg, img, c2w, K, n, (float4*)rays);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_class_count(const float* rays, int64_t n, int n_cls, int64_t* counts, void* stream) {
  if (n == 0) return PW_OK;
  PW_CHECK_ARG(rays && counts && n > 0 && n_cls > 0 && n_cls <= 64, "pw_class_count: bad arguments (n_cls <= 64)");
  int64_t want = pw_cdiv(n, 256);
  hipLaunchKernelGGL(k_class_count, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, pw_stream(stream),
                     rays, n, n_cls, (unsigned long long*)counts);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_wrs_weights(const float* rays, int64_t n, int frame_id, const float* balance_weight, int n_cls,
                          const int32_t* dynamic_class, int n_dyn, float weight_adj, float weight_dyn,
                          float* weights, void* stream) {
  if (n == 0) return PW_OK;
  PW_CHECK_ARG(rays && balance_weight && weights && n > 0 && n_cls > 0 && (n_dyn == 0 || dynamic_class),
               "pw_wrs_weights: bad arguments");
  hipLaunchKernelGGL(k_wrs_weights, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream), rays, n,
                     frame_id, balance_weight, n_cls, dynamic_class, n_dyn, weight_adj, weight_dyn, weights);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
