/*
 * pw_oracle.c -- CPU restatement of PreWorld's camera->voxel occupancy hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  The product
 * path (preworld_amd/) never imports it and fails loudly without its HIP library.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows.  All math is IEEE fp32 with contraction disabled
 * (build with -ffp-contract=off) so that the op order written here IS the result.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pooling fwd/bwd: pinned by the reference's only KAT (bev_pool.py:145-176)
 *   - geometry/ranks, conv stack, heads, render: pinned by fixtures generated from
 *     the imported reference Python (tools/gen_golden.py -> tests/golden/)
 *   - raw2alpha / alpha2weight / cumdist_thres kernels: restated from the .cu text
 *     (no reference CPU build exists) -- checked through the reference's Python
 *     wrappers only; "parity unpinned" for their CUDA libm (powf/expf) last bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PWO_API __attribute__((visibility("default")))

PWO_API int pwo_version(void) { return 1; }

PWO_API int pwo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

PWO_API void pwo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* A1  create_frustum                       view_transformer.py:84-112       */
/* ------------------------------------------------------------------------- */
/* torch.linspace(start,end,steps) for float32 (ATen RangeFactories linspace:
 * step=(end-start)/(steps-1); idx<steps/2 ? start+step*idx : end-step*(steps-idx-1)) */
static float pwo_linspace_f32(float start, float end, int steps, int idx) {
  if (steps == 1) return start;
  float step = (end - start) / (float)(steps - 1);
  int halfway = steps / 2;
  if (idx < halfway) return start + step * (float)idx;
  return end - step * (float)(steps - idx - 1);
}

/* frustum[d][h][w] = (x, y, depth); depth = d0 + d*dstep (torch.arange float) */
PWO_API void pwo_create_frustum(int D, int Hf, int Wf, float d0, float dstep,
                                int H_in, int W_in, float* frustum) {
  for (int d = 0; d < D; ++d)
    for (int h = 0; h < Hf; ++h)
      for (int w = 0; w < Wf; ++w) {
        float* p = frustum + (((size_t)d * Hf + h) * Wf + w) * 3;
        p[0] = pwo_linspace_f32(0.f, (float)(W_in - 1), Wf, w);
        p[1] = pwo_linspace_f32(0.f, (float)(H_in - 1), Hf, h);
        /* torch.arange(start,end,step,dtype=float): start + i*step computed in
         * double accumulation then cast (ATen arange: value = start + step*i) */
        p[2] = (float)((double)d0 + (double)dstep * (double)d);
      }
}

/* ------------------------------------------------------------------------- */
/* 3x3 helpers (closed form; the product's pw_lss_camera_matrices is the same) */
/* ------------------------------------------------------------------------- */
static void inv3x3_f32(const float* m, float* o) {
  /* adjugate / determinant, fp32, fixed op order */
  float a = m[0], b = m[1], c = m[2];
  float d = m[3], e = m[4], f = m[5];
  float g = m[6], h = m[7], i = m[8];
  float A = e * i - f * h;
  float B = c * h - b * i;
  float C = b * f - c * e;
  float D = f * g - d * i;
  float E = a * i - c * g;
  float F = c * d - a * f;
  float G = d * h - e * g;
  float H = b * g - a * h;
  float I = a * e - b * d;
  float det = (a * A + b * D) + c * G;
  float r = 1.0f / det;
  o[0] = A * r; o[1] = B * r; o[2] = C * r;
  o[3] = D * r; o[4] = E * r; o[5] = F * r;
  o[6] = G * r; o[7] = H * r; o[8] = I * r;
}

static void mat3_mul(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) acc += a[i * 3 + k] * b[k * 3 + j];
      o[i * 3 + j] = acc;
    }
}

/* view_transformer.py:139-150: inverse(post_rots); combine = R(sensor2ego) @ inverse(cam2imgs);
 * trans = sensor2ego[:3,3].  BN = B*N cameras. */
PWO_API void pwo_camera_matrices(int BN, const float* sensor2ego /*BN,4,4*/,
                                 const float* cam2imgs /*BN,3,3*/,
                                 const float* post_rots /*BN,3,3*/,
                                 float* inv_post_rot, float* combine, float* trans) {
  for (int c = 0; c < BN; ++c) {
    float R[9], Kinv[9];
    const float* S = sensor2ego + c * 16;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[i * 3 + j] = S[i * 4 + j];
    inv3x3_f32(post_rots + c * 9, inv_post_rot + c * 9);
    inv3x3_f32(cam2imgs + c * 9, Kinv);
    mat3_mul(R, Kinv, combine + c * 9);
    trans[c * 3 + 0] = S[3];
    trans[c * 3 + 1] = S[7];
    trans[c * 3 + 2] = S[11];
  }
}

/* ------------------------------------------------------------------------- */
/* A2  get_lidar_coor                       view_transformer.py:114-153      */
/* ------------------------------------------------------------------------- */
/* One frustum point through the chain.  matmul = ATen baddbmm_cpu_kernel order
 * (acc=0; acc+=a[k]*b[k] for k=0..2), no FMA. */
static inline void pwo_point_chain(const float* fr, const float* ipr, const float* pt,
                                   const float* comb, const float* tr, const float* bda,
                                   float* out3) {
  float p0 = fr[0] - pt[0], p1 = fr[1] - pt[1], p2 = fr[2] - pt[2];
  float q[3];
  for (int i = 0; i < 3; ++i) {
    float acc = 0.f;
    acc += ipr[i * 3 + 0] * p0;
    acc += ipr[i * 3 + 1] * p1;
    acc += ipr[i * 3 + 2] * p2;
    q[i] = acc;
  }
  float u0 = q[0] * q[2], u1 = q[1] * q[2], u2 = q[2];
  float r[3];
  for (int i = 0; i < 3; ++i) {
    float acc = 0.f;
    acc += comb[i * 3 + 0] * u0;
    acc += comb[i * 3 + 1] * u1;
    acc += comb[i * 3 + 2] * u2;
    r[i] = acc + tr[i];
  }
  for (int i = 0; i < 3; ++i) {
    float acc = 0.f;
    acc += bda[i * 3 + 0] * r[0];
    acc += bda[i * 3 + 1] * r[1];
    acc += bda[i * 3 + 2] * r[2];
    out3[i] = acc;
  }
}

PWO_API void pwo_lidar_coor(int B, int N, int D, int H, int W, const float* frustum,
                            const float* inv_post_rot, const float* post_trans,
                            const float* combine, const float* trans, const float* bda,
                            float* coor /*B,N,D,H,W,3*/) {
  size_t DHW = (size_t)D * H * W;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      int c = b * N + n;
      for (size_t i = 0; i < DHW; ++i)
        pwo_point_chain(frustum + i * 3, inv_post_rot + c * 9, post_trans + c * 3,
                        combine + c * 9, trans + c * 3, bda + b * 9,
                        coor + ((size_t)c * DHW + i) * 3);
    }
}

/* ------------------------------------------------------------------------- */
/* A3  voxel_pooling_prepare_v2             view_transformer.py:203-261      */
/* ------------------------------------------------------------------------- */
/* coor -> voxel id (or -1): ((coor-lower)/interval).long() truncates toward zero
 * (:228); in-box filter (:234-236); rank = b*XYZ + z*XY + y*X + x (:242-245). */
PWO_API void pwo_voxel_index(size_t npts_per_batch, int B, const float* coor,
                             const float* lower3, const float* interval3,
                             int gx, int gy, int gz, int32_t* vox) {
  size_t total = npts_per_batch * (size_t)B;
#pragma omp parallel for
  for (size_t i = 0; i < total; ++i) {
    const float* p = coor + i * 3;
    float fx = (p[0] - lower3[0]) / interval3[0];
    float fy = (p[1] - lower3[1]) / interval3[1];
    float fz = (p[2] - lower3[2]) / interval3[2];
    /* .long(): C cast truncates toward zero; guard NaN/huge like a saturating cast */
    int64_t ix = (fx == fx && fabsf(fx) < 9.0e18f) ? (int64_t)fx : INT64_MIN;
    int64_t iy = (fy == fy && fabsf(fy) < 9.0e18f) ? (int64_t)fy : INT64_MIN;
    int64_t iz = (fz == fz && fabsf(fz) < 9.0e18f) ? (int64_t)fz : INT64_MIN;
    int b = (int)(i / npts_per_batch);
    if (ix >= 0 && ix < gx && iy >= 0 && iy < gy && iz >= 0 && iz < gz)
      vox[i] = (int32_t)(((int64_t)b * gz + iz) * gy * gx + iy * gx + ix);
    else
      vox[i] = -1;
  }
}

/* Stable counting sort by voxel id (the reference's argsort is unstable, :246, so
 * any intra-voxel order is "reference behaviour"; we fix ascending point index).
 * Outputs sized for the worst case (n_total); returns kept count, *n_intervals set.
 * ranks_feat = point index with the depth axis removed (:219-224). */
PWO_API int64_t pwo_voxel_prepare(size_t n_total, int n_voxels, const int32_t* vox,
                                  int D, int HW, int32_t* ranks_bev, int32_t* ranks_depth,
                                  int32_t* ranks_feat, int32_t* interval_starts,
                                  int32_t* interval_lengths, int32_t* n_intervals) {
  int32_t* start = (int32_t*)calloc((size_t)n_voxels + 1, sizeof(int32_t));
  for (size_t i = 0; i < n_total; ++i)
    if (vox[i] >= 0) start[vox[i] + 1]++;
  for (int v = 0; v < n_voxels; ++v) start[v + 1] += start[v];
  int64_t kept = start[n_voxels];
  int32_t* cur = (int32_t*)malloc((size_t)n_voxels * sizeof(int32_t));
  memcpy(cur, start, (size_t)n_voxels * sizeof(int32_t));
  size_t DHW = (size_t)D * HW;
  for (size_t i = 0; i < n_total; ++i) {
    int v = vox[i];
    if (v < 0) continue;
    int pos = cur[v]++;
    ranks_bev[pos] = v;
    ranks_depth[pos] = (int32_t)i;
    size_t cam = i / DHW;            /* b*N + n */
    size_t hw = i % (size_t)HW;
    ranks_feat[pos] = (int32_t)(cam * HW + hw);
  }
  int ni = 0;
  for (int v = 0; v < n_voxels; ++v) {
    int len = start[v + 1] - start[v];
    if (len > 0) {
      interval_starts[ni] = start[v];
      interval_lengths[ni] = len;
      ++ni;
    }
  }
  *n_intervals = ni;
  free(cur);
  free(start);
  return kept;
}

/* ------------------------------------------------------------------------- */
/* A4  bev_pool_v2 forward                  bev_pool_cuda.cu:21-48           */
/* ------------------------------------------------------------------------- */
/* one (interval, channel) per "thread": serial psum over the interval, ASSIGN to
 * out[ranks_bev[start]*c + ch]; out is pre-zeroed by the caller (bev_pool.py:27). */
PWO_API void pwo_bev_pool_v2_forward(int c, int n_intervals, const float* depth,
                                     const float* feat, const int32_t* ranks_depth,
                                     const int32_t* ranks_feat, const int32_t* ranks_bev,
                                     const int32_t* interval_starts,
                                     const int32_t* interval_lengths, float* out) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int idx = 0; idx < n_intervals; ++idx) {
    int s = interval_starts[idx], len = interval_lengths[idx];
    float* o = out + (size_t)ranks_bev[s] * c;
    for (int ch = 0; ch < c; ++ch) {
      float psum = 0.f;
      for (int i = 0; i < len; ++i)
        psum += feat[(size_t)ranks_feat[s + i] * c + ch] * depth[ranks_depth[s + i]];
      o[ch] = psum;
    }
  }
}

/* A5  bev_pool_v2 backward                 bev_pool_cuda.cu:67-121          */
/* intervals here are per feat pixel (bev_pool.py:47-57 re-sorts by ranks_feat) */
PWO_API void pwo_bev_pool_v2_backward(int c, int n_intervals, const float* out_grad,
                                      const float* depth, const float* feat,
                                      const int32_t* ranks_depth, const int32_t* ranks_feat,
                                      const int32_t* ranks_bev, const int32_t* interval_starts,
                                      const int32_t* interval_lengths, float* depth_grad,
                                      float* feat_grad) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int idx = 0; idx < n_intervals; ++idx) {
    int s = interval_starts[idx], len = interval_lengths[idx];
    for (int i = 0; i < len; ++i) {
      const float* og = out_grad + (size_t)ranks_bev[s + i] * c;
      const float* f = feat + (size_t)ranks_feat[s + i] * c;
      float g = 0.f;
      for (int ch = 0; ch < c; ++ch) g += og[ch] * f[ch];
      depth_grad[ranks_depth[s + i]] = g;
    }
    float* fg = feat_grad + (size_t)ranks_feat[s] * c;
    for (int ch = 0; ch < c; ++ch) {
      float g = 0.f;
      for (int i = 0; i < len; ++i)
        g += out_grad[(size_t)ranks_bev[s + i] * c + ch] * depth[ranks_depth[s + i]];
      fg[ch] = g;
    }
  }
}

/* bev_pool.py:47-57: argsort by ranks_feat + per-pixel intervals (stable here) */
PWO_API int pwo_bev_pool_bp_prepare(int64_t n_pts, int n_feat_pix, const int32_t* ranks_bev,
                                    const int32_t* ranks_depth, const int32_t* ranks_feat,
                                    int32_t* o_bev, int32_t* o_depth, int32_t* o_feat,
                                    int32_t* interval_starts, int32_t* interval_lengths) {
  int32_t* start = (int32_t*)calloc((size_t)n_feat_pix + 1, sizeof(int32_t));
  for (int64_t i = 0; i < n_pts; ++i) start[ranks_feat[i] + 1]++;
  for (int v = 0; v < n_feat_pix; ++v) start[v + 1] += start[v];
  int32_t* cur = (int32_t*)malloc((size_t)n_feat_pix * sizeof(int32_t));
  memcpy(cur, start, (size_t)n_feat_pix * sizeof(int32_t));
  for (int64_t i = 0; i < n_pts; ++i) {
    int pos = cur[ranks_feat[i]]++;
    o_bev[pos] = ranks_bev[i];
    o_depth[pos] = ranks_depth[i];
    o_feat[pos] = ranks_feat[i];
  }
  int ni = 0;
  for (int v = 0; v < n_feat_pix; ++v) {
    int len = start[v + 1] - start[v];
    if (len > 0) { interval_starts[ni] = start[v]; interval_lengths[ni] = len; ++ni; }
  }
  free(cur);
  free(start);
  return ni;
}

/* ------------------------------------------------------------------------- */
/* A6-A9,A11  Conv3d / BN / ReLU / upsample (torch semantics, NCDHW)         */
/* resnet.py:88-184, lss_fpn.py:103-148, occupancy_head.py:80-105,           */
/* preworld.py:72-79                                                          */
/* ------------------------------------------------------------------------- */
/* Direct cross-correlation, weight (Cout,Cin,k,k,k), zero padding, optional bias.
 * Accumulation order: bias, then cin-major, taps (kd,kh,kw)-minor -- fp32, no FMA. */
PWO_API void pwo_conv3d(const float* x, const float* w, const float* bias, int N, int Cin,
                        int D, int H, int W, int Cout, int k, int stride, int pad,
                        float* y) {
  int Do = (D + 2 * pad - k) / stride + 1;
  int Ho = (H + 2 * pad - k) / stride + 1;
  int Wo = (W + 2 * pad - k) / stride + 1;
  size_t in_cs = (size_t)D * H * W, out_cs = (size_t)Do * Ho * Wo;
#pragma omp parallel for collapse(3) schedule(dynamic, 4)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int od = 0; od < Do; ++od) {
        float* yo = y + ((size_t)n * Cout + co) * out_cs + (size_t)od * Ho * Wo;
        float b0 = bias ? bias[co] : 0.f;
        for (size_t i = 0; i < (size_t)Ho * Wo; ++i) yo[i] = b0;
        for (int ci = 0; ci < Cin; ++ci) {
          const float* xi = x + ((size_t)n * Cin + ci) * in_cs;
          const float* wk = w + ((size_t)co * Cin + ci) * k * k * k;
          for (int kd = 0; kd < k; ++kd) {
            int id = od * stride - pad + kd;
            if (id < 0 || id >= D) continue;
            for (int kh = 0; kh < k; ++kh)
              for (int kw = 0; kw < k; ++kw) {
                float wv = wk[(kd * k + kh) * k + kw];
                for (int oh = 0; oh < Ho; ++oh) {
                  int ih = oh * stride - pad + kh;
                  if (ih < 0 || ih >= H) continue;
                  const float* xr = xi + ((size_t)id * H + ih) * W;
                  float* yr = yo + (size_t)oh * Wo;
                  /* valid ow range: 0 <= ow*stride - pad + kw < W */
                  int lo = 0;
                  while (lo < Wo && lo * stride - pad + kw < 0) ++lo;
                  int hi = Wo;
                  while (hi > lo && (hi - 1) * stride - pad + kw >= W) --hi;
                  if (stride == 1) {
                    const float* xs = xr + (lo - pad + kw);
                    for (int ow = lo; ow < hi; ++ow) yr[ow] += wv * xs[ow - lo];
                  } else {
                    for (int ow = lo; ow < hi; ++ow)
                      yr[ow] += wv * xr[ow * stride - pad + kw];
                  }
                }
              }
          }
        }
      }
}

/* BatchNorm3d eval: y = (x-mean)/sqrt(var+eps)*gamma + beta, fused optional
 * residual add and ReLU (BasicBlock3D.forward resnet.py:113-123). In place OK. */
PWO_API void pwo_bn_act(float* x, int N, int C, size_t spatial, const float* gamma,
                        const float* beta, const float* mean, const float* var, float eps,
                        const float* residual, int relu) {
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float inv = 1.0f / sqrtf(var[c] + eps);
      float a = gamma ? gamma[c] * inv : inv;
      float b = (beta ? beta[c] : 0.f) - mean[c] * a;
      float* p = x + ((size_t)n * C + c) * spatial;
      const float* r = residual ? residual + ((size_t)n * C + c) * spatial : NULL;
      for (size_t i = 0; i < spatial; ++i) {
        float v = p[i] * a + b;
        if (r) v += r[i];
        if (relu && v < 0.f) v = 0.f;
        p[i] = v;
      }
    }
}

PWO_API void pwo_relu(float* x, size_t n) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i)
    if (x[i] < 0.f) x[i] = 0.f;
}

/* nn.Upsample(scale_factor=s, mode='trilinear', align_corners=True) lss_fpn.py:111-114
 * ATen upsample_trilinear3d: src = dst * (in-1)/(out-1); lambda in fp32. */
PWO_API void pwo_upsample_trilinear_ac(const float* x, int NC, int D, int H, int W, int s,
                                       float* y) {
  int Do = D * s, Ho = H * s, Wo = W * s;
  float sd = Do > 1 ? (float)(D - 1) / (float)(Do - 1) : 0.f;
  float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
#pragma omp parallel for collapse(2)
  for (int c = 0; c < NC; ++c)
    for (int od = 0; od < Do; ++od) {
      const float* xc = x + (size_t)c * D * H * W;
      float fd = sd * (float)od;
      int d0 = (int)fd;
      int d1 = d0 + (d0 < D - 1 ? 1 : 0);
      float ld1 = fd - (float)d0, ld0 = 1.f - ld1;
      for (int oh = 0; oh < Ho; ++oh) {
        float fh = sh * (float)oh;
        int h0 = (int)fh;
        int h1 = h0 + (h0 < H - 1 ? 1 : 0);
        float lh1 = fh - (float)h0, lh0 = 1.f - lh1;
        float* yr = y + (((size_t)c * Do + od) * Ho + oh) * Wo;
        for (int ow = 0; ow < Wo; ++ow) {
          float fw = sw * (float)ow;
          int w0 = (int)fw;
          int w1 = w0 + (w0 < W - 1 ? 1 : 0);
          float lw1 = fw - (float)w0, lw0 = 1.f - lw1;
#define XV(d, h, w) xc[((size_t)(d) * H + (h)) * W + (w)]
          yr[ow] = ld0 * (lh0 * (lw0 * XV(d0, h0, w0) + lw1 * XV(d0, h0, w1)) +
                          lh1 * (lw0 * XV(d0, h1, w0) + lw1 * XV(d0, h1, w1))) +
                   ld1 * (lh0 * (lw0 * XV(d1, h0, w0) + lw1 * XV(d1, h0, w1)) +
                          lh1 * (lw0 * XV(d1, h1, w0) + lw1 * XV(d1, h1, w1)));
#undef XV
        }
      }
    }
}

/* ------------------------------------------------------------------------- */
/* A10/A12  per-voxel MLPs                 preworld_temporal_traj.py:81-150  */
/* ------------------------------------------------------------------------- */
/* nn.Softplus(beta=1, threshold=20) */
static inline float pwo_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

/* y[m][o] = act(b[o] + sum_i x[m][i]*w[o][i]); act: 0 none, 1 relu, 2 softplus */
PWO_API void pwo_linear(const float* x, size_t M, int In, const float* w, const float* b,
                        int Out, int act, float* y) {
#pragma omp parallel for
  for (size_t m = 0; m < M; ++m) {
    const float* xr = x + m * In;
    float* yr = y + m * Out;
    for (int o = 0; o < Out; ++o) {
      float acc = b ? b[o] : 0.f;
      const float* wr = w + (size_t)o * In;
      for (int i = 0; i < In; ++i) acc += xr[i] * wr[i];
      if (act == 1) acc = acc < 0.f ? 0.f : acc;
      else if (act == 2) acc = pwo_softplus(acc);
      yr[o] = acc;
    }
  }
}

/* One recursion step of the state-conditioned forecast (preworld_temporal_traj.py:
 * 335-342): v_out = v + fusion_head(cat(v, e)); fusion_head = Linear(2C,4C) Softplus
 * Linear(4C,C).  v:(M,C) channels-last voxel features, e:(C) ego feature (plan_head
 * output, voxel-invariant). */
PWO_API void pwo_forecast_step(const float* v, size_t M, int C, const float* e,
                               const float* w1 /*4C,2C*/, const float* b1, const float* w2
                               /*C,4C*/, const float* b2, float* v_out) {
  int Hd = 4 * C;
#pragma omp parallel
  {
    float* hid = (float*)malloc(sizeof(float) * Hd);
#pragma omp for
    for (size_t m = 0; m < M; ++m) {
      const float* vr = v + m * C;
      for (int h = 0; h < Hd; ++h) {
        float acc = b1[h];
        const float* wr = w1 + (size_t)h * 2 * C;
        for (int i = 0; i < C; ++i) acc += vr[i] * wr[i];
        for (int i = 0; i < C; ++i) acc += e[i] * wr[C + i];
        hid[h] = pwo_softplus(acc);
      }
      for (int o = 0; o < C; ++o) {
        float acc = b2[o];
        const float* wr = w2 + (size_t)o * Hd;
        for (int h = 0; h < Hd; ++h) acc += hid[h] * wr[h];
        v_out[m * C + o] = acc + vr[o];
      }
    }
    free(hid);
  }
}

/* argmax over the last dim (first max wins, torch.argmax CPU semantics) -> uint8 */
PWO_API void pwo_argmax_u8(const float* x, size_t M, int C, uint8_t* out) {
#pragma omp parallel for
  for (size_t m = 0; m < M; ++m) {
    const float* r = x + m * C;
    int best = 0;
    float bv = r[0];
    for (int c = 1; c < C; ++c)
      if (r[c] > bv) { bv = r[c]; best = c; }
    out[m] = (uint8_t)best;
  }
}

/* ------------------------------------------------------------------------- */
/* A13  sample_ray                          nerf_head.py:32-55               */
/* ------------------------------------------------------------------------- */
/* t table is passed in (built by the caller exactly as :38-43, torch.linspace).
 * rays_o/d:(R,3), out ray_pts:(R,S,3), inner_mask:(R,S) */
PWO_API void pwo_sample_ray(int R, int S, const float* rays_o, const float* rays_d,
                            const float* t, const float* center3, const float* radius3,
                            float bg_len, const float* bda /*3x3*/, float* ray_pts,
                            uint8_t* inner_mask) {
#pragma omp parallel for
  for (int r = 0; r < R; ++r) {
    float o[3], d[3];
    for (int i = 0; i < 3; ++i) o[i] = (rays_o[r * 3 + i] - center3[i]) / radius3[i];
    /* torch.norm(dim=-1): sqrt(sum of squares), fp32 sequential */
    float nn = 0.f;
    for (int i = 0; i < 3; ++i) nn += rays_d[r * 3 + i] * rays_d[r * 3 + i];
    nn = sqrtf(nn);
    for (int i = 0; i < 3; ++i) d[i] = rays_d[r * 3 + i] / nn;
    for (int s = 0; s < S; ++s) {
      float p[3];
      for (int i = 0; i < 3; ++i) p[i] = o[i] + d[i] * t[s];
      float norm = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
      int inner = norm <= 1.f;
      if (!inner) {
        float sc = (1.f + bg_len) - bg_len / norm;
        for (int i = 0; i < 3; ++i) p[i] = p[i] / norm * sc;
      }
      float* q = ray_pts + ((size_t)r * S + s) * 3;
      for (int i = 0; i < 3; ++i) {
        float acc = 0.f;
        acc += bda[i * 3 + 0] * p[0];
        acc += bda[i * 3 + 1] * p[1];
        acc += bda[i * 3 + 2] * p[2];
        q[i] = acc;
      }
      inner_mask[(size_t)r * S + s] = (uint8_t)inner;
    }
  }
}

/* A14  cumdist_thres                       ub360_utils_kernel.cu:13-32      */
PWO_API void pwo_cumdist_thres(int n_rays, int n_pts, const float* dist, float thres,
                               uint8_t* mask) {
#pragma omp parallel for
  for (int r = 0; r < n_rays; ++r) {
    float cum = 0.f;
    for (int i = 0; i < n_pts; ++i) {
      size_t k = (size_t)r * n_pts + i;
      cum += dist[k];
      int over = cum > thres;
      cum *= (float)(!over);
      mask[k] = (uint8_t)over;
    }
  }
}

/* A15  F.grid_sample 5-D, bilinear(=trilinear), align_corners=True, zeros padding
 * nerf_head.py:211-225.  grid (C,X,Y,Z) viewed by torch as (C,D=X,H=Y,W=Z); the
 * sample coordinate is ind_norm = ((xyz-min)/(max-min)).flip(-1)*2-1 so that
 * grid_sample's x<->W=Z, y<->H=Y, z<->D=X.  Here we take xyz (P,3) directly. */
PWO_API void pwo_grid_sample_xyz(const float* grid, int C, int X, int Y, int Z,
                                 const float* xyz, size_t P, const float* xyz_min,
                                 const float* xyz_max, float* out /*P,C*/) {
#pragma omp parallel for
  for (size_t p = 0; p < P; ++p) {
    float g[3];
    for (int i = 0; i < 3; ++i) {
      float nrm = (xyz[p * 3 + i] - xyz_min[i]) / (xyz_max[i] - xyz_min[i]);
      g[i] = nrm * 2.f - 1.f;
    }
    /* unnormalize, align_corners=True: ((coord+1)/2)*(size-1) */
    float fx = ((g[0] + 1.f) / 2.f) * (float)(X - 1);
    float fy = ((g[1] + 1.f) / 2.f) * (float)(Y - 1);
    float fz = ((g[2] + 1.f) / 2.f) * (float)(Z - 1);
    float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    /* ATen GridSampler.cpp weights: near corner (i0+1)-f, far corner f-i0 */
    float tx1 = fx - x0f, ty1 = fy - y0f, tz1 = fz - z0f;
    float tx0 = (x0f + 1.f) - fx, ty0 = (y0f + 1.f) - fy, tz0 = (z0f + 1.f) - fz;
    for (int c = 0; c < C; ++c) {
      const float* gc = grid + (size_t)c * X * Y * Z;
      float acc = 0.f;
      /* ATen grid_sampler_3d corner order: tnw,tne,tsw,tse,bnw,bne,bsw,bse with
       * (ix<->W, iy<->H, iz<->D): weights (ix_tse-ix)*(iy_tse-iy)*(iz_bse-iz) ... */
      for (int dzx = 0; dzx < 2; ++dzx)       /* our X axis == torch D ("z") */
        for (int dy = 0; dy < 2; ++dy)        /* Y == torch H ("y") */
          for (int dzz = 0; dzz < 2; ++dzz) { /* our Z axis == torch W ("x") */
            int xi = x0 + dzx, yi = y0 + dy, zi = z0 + dzz;
            float wx = dzx ? tx1 : tx0;
            float wy = dy ? ty1 : ty0;
            float wz = dzz ? tz1 : tz0;
            if (xi < 0 || xi >= X || yi < 0 || yi >= Y || zi < 0 || zi >= Z) continue;
            acc += gc[((size_t)xi * Y + yi) * Z + zi] * ((wz * wy) * wx);
          }
      out[p * C + c] = acc;
    }
  }
}

/* A16  raw2alpha / backward        render_utils_kernel.cu:431-443,507-517 */
PWO_API void pwo_raw2alpha(const float* density, float shift, float interval, size_t n,
                           float* exp_d, float* alpha) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) {
    float e = expf(density[i] + shift);
    exp_d[i] = e;
    alpha[i] = 1.f - powf(1.f + e, -interval);
  }
}

PWO_API void pwo_raw2alpha_backward(const float* exp_d, const float* grad_back,
                                    float interval, size_t n, float* grad) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) {
    /* min(exp_d, 1e10) is evaluated in double in the .cu (1e10 literal) */
    double m = (double)exp_d[i] < 1e10 ? (double)exp_d[i] : 1e10;
    grad[i] = (float)(m * (double)powf(1.f + exp_d[i], -interval - 1.f) *
                      (double)interval * (double)grad_back[i]);
  }
}

/* A17  alpha2weight                 render_utils_kernel.cu:577-651 */
/* ray_id sorted int64; weight zero-init, T ones-init, alphainv_last ones-init,
 * i_start/i_end zero-init then set by __set_i_for_segment_start_end + host fixup. */
PWO_API void pwo_alpha2weight(const float* alpha, const int64_t* ray_id, size_t n_pts,
                              int n_rays, float* weight, float* T, float* alphainv_last,
                              int64_t* i_start, int64_t* i_end) {
  for (size_t i = 0; i < n_pts; ++i) { weight[i] = 0.f; T[i] = 1.f; }
  for (int r = 0; r < n_rays; ++r) { alphainv_last[r] = 1.f; i_start[r] = 0; i_end[r] = 0; }
  if (n_pts == 0) return;
  for (size_t i = 1; i < n_pts; ++i)
    if (ray_id[i] != ray_id[i - 1]) {
      i_start[ray_id[i]] = (int64_t)i;
      i_end[ray_id[i - 1]] = (int64_t)i;
    }
  i_end[ray_id[n_pts - 1]] = (int64_t)n_pts;
#pragma omp parallel for
  for (int r = 0; r < n_rays; ++r) {
    int i_s = (int)i_start[r], i_e_max = (int)i_end[r];
    float T_cum = 1.f;
    int i;
    for (i = i_s; i < i_e_max; ++i) {
      T[i] = T_cum;
      weight[i] = T_cum * alpha[i];
      T_cum = (float)((double)T_cum * (1. - (double)alpha[i])); /* `1. - alpha` is double */
      if ((double)T_cum < 1e-3) { i += 1; break; }
    }
    i_end[r] = i;
    alphainv_last[r] = T_cum;
  }
}

/* alpha2weight backward             render_utils_kernel.cu:654-677 */
PWO_API void pwo_alpha2weight_backward(const float* alpha, const float* weight,
                                       const float* T, const float* alphainv_last,
                                       const int64_t* i_start, const int64_t* i_end,
                                       int n_rays, const float* grad_weights,
                                       const float* grad_last, size_t n_pts, float* grad) {
  for (size_t i = 0; i < n_pts; ++i) grad[i] = 0.f;
#pragma omp parallel for
  for (int r = 0; r < n_rays; ++r) {
    int i_s = (int)i_start[r], i_e = (int)i_end[r];
    float back_cum = grad_last[r] * alphainv_last[r];
    for (int i = i_e - 1; i >= i_s; --i) {
      /* (1-alpha+1e-10) is double in the .cu; back_cum / double -> double -> float */
      grad[i] = (float)((double)(grad_weights[i] * T[i]) -
                        (double)back_cum / (1. - (double)alpha[i] + 1e-10));
      back_cum += grad_weights[i] * weight[i];
    }
  }
}

/* A18  segment_coo(sum) over sorted ray_id   nerf_head.py:331-353 */
PWO_API void pwo_segment_sum(const float* src, const int64_t* index, size_t n, int C,
                             int n_seg, float* out /*n_seg,C zero-init by us*/) {
  memset(out, 0, sizeof(float) * (size_t)n_seg * C);
  for (size_t i = 0; i < n; ++i)
    for (int c = 0; c < C; ++c) out[(size_t)index[i] * C + c] += src[i * C + c];
}

/* ------------------------------------------------------------------------- */
/* A22  confusion histogram           occ_metrics.py:82-105                  */
/* ------------------------------------------------------------------------- */
PWO_API void pwo_confusion_hist(const uint8_t* pred, const uint8_t* gt, const uint8_t* mask,
                                size_t n, int n_cl, int64_t* hist /*n_cl*n_cl, accumulates*/) {
  for (size_t i = 0; i < n; ++i) {
    if (mask && !mask[i]) continue;
    if (gt[i] < n_cl) hist[(size_t)n_cl * gt[i] + pred[i]] += 1;
  }
}
