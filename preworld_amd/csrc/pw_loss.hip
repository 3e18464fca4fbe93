// Voxel-grid training losses: CE_ssc_loss, sem_scal_loss, geo_scal_loss of
// mmdet3d/models/detectors/loss.py:20-113 as composed by loss_voxel
// (mmdet3d/models/detectors/preworld_temporal_traj.py:176-199).
//
// The reference makes ~18 x 10 full passes over the 46 MB probability tensor (one masked gather +
// several reductions per class).  Here: ONE pass computes the softmax of every voxel in registers and
// accumulates the 104 sums all three losses are functions of (HBM-bound: logits read once); the
// scalar loss algebra runs on those sums; the backward is a second pass that recomputes the softmax
// and writes d(loss)/d(logits) directly (per-class coefficient vectors come from the sums).
#include "pw_common.h"
#include "pw_vox.h"

namespace {
constexpr int MAXC = 32;
constexpr int NSTATS = 104;

struct LossArgs {
  const float* logits;
  const uint8_t* target;
  const uint8_t* cam;
  const float* cw;
  int B, C, X, Y, Z;
  long long sb, sc, sx, sy, sz;
  int ignore, empty;
  VoxWalk walk;
};

template <int C>
__device__ __forceinline__ void softmax_c(const LossArgs& a, long long off, int ncls, float (&p)[C], float& lse) {
  float m = -3.402823466e38f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    p[c] = c < ncls ? a.logits[off + c * a.sc] : -3.402823466e38f;
    m = fmaxf(m, p[c]);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    p[c] = c < ncls ? __expf(p[c] - m) : 0.f;
    s += p[c];
  }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
  lse = m + __logf(s);
}
}  // namespace

// one thread per voxel, per-thread sums -> wave shuffle -> LDS -> one fp64 atomic per block and sum
template <int C>
__global__ void __launch_bounds__(256) k_voxel_loss_stats(LossArgs a, long long n_vox, double* __restrict__ stats) {
  __shared__ float red[4][NSTATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ce = 0.f, cew = 0.f, nM = 0.f, gI = 0.f, gSne = 0.f, gSnt = 0.f, gSe0 = 0.f, gS0 = 0.f;
  float sp[C], st[C], spt[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { sp[c] = 0.f; st[c] = 0.f; spt[c] = 0.f; }
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_vox; u += (long long)gridDim.x * blockDim.x) {
    int b, x, y, z;
    long long v;
    const long long off = vox_walk(a, u, b, x, y, z, v);
    float p[C], lse;
    softmax_c<C>(a, off, a.C, p, lse);
    const int t = a.target[v];
    const bool cam = a.cam ? a.cam[v] != 0 : true;
    if (t != a.ignore && t < a.C) {
      const float w = a.cw ? a.cw[t] : 1.f;
      ce += w * (lse - a.logits[off + t * a.sc]);
      cew += w;
    }
    const bool inM = t != a.ignore && cam;
    if (inM) {
      nM += 1.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        sp[c] += p[c];
        if (t == c) { st[c] += 1.f; spt[c] += p[c]; }
      }
    }
    float pe = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) pe = (c == a.empty) ? p[c] : pe;
    const float nt = (t != a.empty && cam) ? 1.f : 0.f;
    gI += nt * (1.f - pe); gSne += 1.f - pe; gSnt += nt; gSe0 += (1.f - nt) * pe; gS0 += 1.f - nt;
  }
  auto wsum = [](float x) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  };
  float vals[8] = {ce, cew, nM, gI, gSne, gSnt, gSe0, gS0};
#pragma unroll
  for (int k = 0; k < 8; ++k) vals[k] = wsum(vals[k]);
#pragma unroll
  for (int c = 0; c < C; ++c) { sp[c] = wsum(sp[c]); st[c] = wsum(st[c]); spt[c] = wsum(spt[c]); }
  if (lane == 0) {
    red[wave][0] = vals[0]; red[wave][1] = vals[1]; red[wave][2] = vals[2];
    for (int c = 0; c < MAXC; ++c) {
      red[wave][3 + c] = c < C ? sp[c < C ? c : 0] : 0.f;
      red[wave][35 + c] = c < C ? st[c < C ? c : 0] : 0.f;
      red[wave][67 + c] = c < C ? spt[c < C ? c : 0] : 0.f;
    }
    for (int k = 0; k < 5; ++k) red[wave][99 + k] = vals[3 + k];
  }
  __syncthreads();
  if (threadIdx.x < NSTATS) {
    const double s = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] +
                     (double)red[3][threadIdx.x];
    if (s != 0.0) atomicAdd(&stats[threadIdx.x], s);
  }
}

template <int C>
__global__ void __launch_bounds__(256) k_voxel_loss_grad(LossArgs a, long long n_vox, const float* __restrict__ coef,
                                                        float* __restrict__ grad) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_vox) return;
  int b, x, y, z;
  long long v;
  const long long off = vox_walk(a, u, b, x, y, z, v);
  float p[C], lse;
  softmax_c<C>(a, off, a.C, p, lse);
  const int t = a.target[v];
  const bool cam = a.cam ? a.cam[v] != 0 : true;
  const bool inM = t != a.ignore && cam;
  const float nt = (t != a.empty && cam) ? 1.f : 0.f;
  const float gc0 = coef[2 * MAXC], gc1 = coef[2 * MAXC + 1], ce_scale = coef[2 * MAXC + 2];
  float g[C];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float gv = 0.f;
    if (c < a.C) {
      if (inM) gv = coef[c] + (t == c ? coef[MAXC + c] : 0.f);
      if (c == a.empty) gv += gc0 + gc1 * nt;
    }
    g[c] = gv;
    dot += p[c] * gv;
  }
  const float wce = (t != a.ignore && t < a.C) ? ce_scale * (a.cw ? a.cw[t] : 1.f) : 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c)
    if (c < a.C) grad[off + c * a.sc] = p[c] * (g[c] - dot) + wce * (p[c] - (t == c ? 1.f : 0.f));
}

static int fill_args(LossArgs& a, const float* logits, const uint8_t* target, const uint8_t* cam, const float* cw,
                     int B, int n_cls, int X, int Y, int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy,
                     int64_t sz, int ignore_index, int empty_idx) {
  PW_CHECK_ARG(logits && target, "pw_voxel_loss: null pointer");
  PW_CHECK_ARG(B > 0 && X > 0 && Y > 0 && Z > 0 && n_cls > 1 && n_cls <= MAXC, "pw_voxel_loss: bad shape (n_cls <= 32)");
  PW_CHECK_ARG(empty_idx >= 0 && empty_idx < n_cls, "pw_voxel_loss: empty_idx outside the classes");
  a.logits = logits; a.target = target; a.cam = cam; a.cw = cw;
  a.B = B; a.C = n_cls; a.X = X; a.Y = Y; a.Z = Z;
  a.sb = sb; a.sc = sc; a.sx = sx; a.sy = sy; a.sz = sz;
  a.ignore = ignore_index; a.empty = empty_idx;
  a.walk = vox_walk_order(sx, sy, sz);
  return PW_OK;
}

PW_API int pw_voxel_loss_stats(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                               const float* class_weights, int B, int n_cls, int X, int Y, int Z, int64_t sb,
                               int64_t sc, int64_t sx, int64_t sy, int64_t sz, int ignore_index, int empty_idx,
                               double* stats, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, logits, target, cam_mask, class_weights, B, n_cls, X, Y, Z, sb, sc, sx, sy, sz,
                         ignore_index, empty_idx)) return rc;
  PW_CHECK_ARG(stats, "pw_voxel_loss_stats: null stats");
  const long long n = (long long)B * X * Y * Z;
  const long long want = pw_cdiv(n, 256);
  // few, long-running blocks: the 104-value block reduction at the end is the expensive part
  const unsigned nb = (unsigned)(want < 512 ? want : 512);
  if (n_cls <= 18) hipLaunchKernelGGL(k_voxel_loss_stats<18>, dim3(nb), dim3(256), 0, pw_stream(stream), a, n, stats);
  else hipLaunchKernelGGL(k_voxel_loss_stats<32>, dim3(nb), dim3(256), 0, pw_stream(stream), a, n, stats);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_voxel_loss_grad(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                              const float* class_weights, const float* coef, int B, int n_cls, int X, int Y,
                              int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy, int64_t sz,
                              int ignore_index, int empty_idx, float* grad_logits, void* stream) {
  LossArgs a;
  if (int rc = fill_args(a, logits, target, cam_mask, class_weights, B, n_cls, X, Y, Z, sb, sc, sx, sy, sz,
                         ignore_index, empty_idx)) return rc;
  PW_CHECK_ARG(coef && grad_logits, "pw_voxel_loss_grad: null pointer");
  const long long n = (long long)B * X * Y * Z;
  const unsigned nb = (unsigned)pw_cdiv(n, 256);
  if (n_cls <= 18) hipLaunchKernelGGL(k_voxel_loss_grad<18>, dim3(nb), dim3(256), 0, pw_stream(stream), a, n, coef, grad_logits);
  else hipLaunchKernelGGL(k_voxel_loss_grad<32>, dim3(nb), dim3(256), 0, pw_stream(stream), a, n, coef, grad_logits);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ---- scalar algebra on the sums (loss.py:58-80, 104-113; F.binary_cross_entropy(x, 1) = -log x with
// the log clamped at -100), one thread, fp64
__device__ __forceinline__ double bce1(double x) { return -fmax(log(x), -100.0); }

__global__ void k_voxel_loss_finish(const double* __restrict__ s, int C, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double N = s[2];
  double sem = 0.0;
  int count = 0;
  for (int i = 0; i < C; ++i) {
    const double Sp = s[3 + i], St = s[35 + i], Spt = s[67 + i];
    if (St > 0) {
      ++count;
      double lc = 0.0;
      if (Sp > 0) lc += bce1(Spt / Sp);
      lc += bce1(Spt / St);
      if (N - St > 0) lc += bce1((N - Sp - St + Spt) / (N - St));
      sem += lc;
    }
  }
  out[0] = (float)(s[0] / s[1]);
  out[1] = (float)(sem / count);
  out[2] = (float)(bce1(s[99] / s[100]) + bce1(s[99] / s[101]) + bce1(s[102] / s[103]));
}

// coef = {ga[32], gb[32], gc0, gc1, ce_scale} for pw_voxel_loss_grad; gout = upstream grads of (ce, sem, geo)
__global__ void k_voxel_loss_coef(const double* __restrict__ s, int C, const float* __restrict__ gout,
                                  float* __restrict__ coef) {
  const int i = threadIdx.x;
  if (blockIdx.x != 0 || i >= 2 * MAXC + 3) return;
  const double N = s[2];
  int count = 0;
  for (int k = 0; k < C; ++k) count += s[35 + k] > 0;
  double v = 0.0;
  if (i < 2 * MAXC) {
    const int c = i < MAXC ? i : i - MAXC;
    if (c < C && s[35 + c] > 0) {
      const double Sp = s[3 + c], St = s[35 + c], Spt = s[67 + c];
      const double a_prec = Sp > 0 ? 1.0 / Sp : 0.0;
      const double a_spec = (N - St > 0) ? 1.0 / (N - Sp - St + Spt) : 0.0;
      const double b_nom = (Sp > 0 ? -1.0 / Spt : 0.0) - 1.0 / Spt;
      v = (i < MAXC ? (a_prec + a_spec) : (b_nom - a_spec)) / count * (double)gout[1];
    }
  } else if (i == 2 * MAXC) {
    v = (-1.0 / s[100] - 1.0 / s[102]) * (double)gout[2];
  } else if (i == 2 * MAXC + 1) {
    v = (2.0 / s[99] + 1.0 / s[102]) * (double)gout[2];
  } else {
    v = (double)gout[0] / s[1];
  }
  coef[i] = (float)v;
}

PW_API int pw_voxel_loss_finish(const double* stats, int n_cls, float* losses, void* stream) {
  PW_CHECK_ARG(stats && losses && n_cls > 1 && n_cls <= MAXC, "pw_voxel_loss_finish: bad arguments");
  hipLaunchKernelGGL(k_voxel_loss_finish, dim3(1), dim3(64), 0, pw_stream(stream), stats, n_cls, losses);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_voxel_loss_coef(const double* stats, int n_cls, const float* grad_losses, float* coef, void* stream) {
  PW_CHECK_ARG(stats && grad_losses && coef && n_cls > 1 && n_cls <= MAXC, "pw_voxel_loss_coef: bad arguments");
  hipLaunchKernelGGL(k_voxel_loss_coef, dim3(1), dim3(128), 0, pw_stream(stream), stats, n_cls, grad_losses, coef);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
