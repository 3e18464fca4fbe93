// Stereo cost volume of the DepthNet (second half):
// mmdet3d/models/necks/view_transformer.py:546-604 (DepthNet.gen_grid + calculate_cost_volumn).
//
// The reference builds a (B*N, D*H, W, 2) sampling grid, then loops 32 times over groups of 4
// channels: grid_sample of the previous frame's stereo feature at every frustum point, |curr - warped|
// summed over the group, accumulated; + bias where the warp fell outside; negate; softmax over D.
// Every iteration materialises a (B*N, 4, D*H, W) tensor (380 MB at the reference shape).
// Here one thread owns one frustum point (pixel, depth bin): it projects the point into the previous
// frame (same operation order as gen_grid), gathers the 4 bilinear corners of ALL channels
// (float4 = one channel group per load when the features are channels-last), accumulates the L1
// cost in the reference's group order, and the 88 threads of a pixel finish the softmax through LDS.
// Nothing but the (B*N, D, H, W) result is written.  Compiled with -ffp-contract=off.
#include "pw_common.h"

namespace {
constexpr int PIX = 8;                 // pixels (consecutive w) per block

struct StereoArgs {
  const float* prev;                   // element (bn, c, y, x) at bn*sbn + c*sc + y*sy + x*sx
  const float* curr;
  long long sbn, sc, sy, sx;
  int BN, C, H, W, D;                  // H, W: stereo feature map = frustum height/width
  const float* ds;                     // [D] depth bins, xs [W], ys [H] frustum pixel coordinates (input-image pixels)
  const float* xs;
  const float* ys;
  const float* ipr;                    // [BN][9] inverse(post_rots)
  const float* post_trans;             // [BN][3]
  const float* comb;                   // [BN][9] k2s_sensor[:3,:3] @ inverse(intrins)
  const float* trans;                  // [BN][3] k2s_sensor[:3,3]
  const float* intrins;                // [BN][9]
  const float* post_rots;              // [BN][9]
  float wi, hi;                        // input image size the frustum coordinates live in (4*W, 4*H)
  float bias;
  float* out;                          // [BN][D][H][W]
};

__device__ __forceinline__ void mat3v(const float* m, const float (&v)[3], float (&o)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = (m[r * 3] * v[0] + m[r * 3 + 1] * v[1]) + m[r * 3 + 2] * v[2];
}
}  // namespace

template <bool CL>    // CL: channels-last features (sc == 1): float4 channel-group loads
__global__ void __launch_bounds__(1024) k_stereo_cost_volume(StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_curr = sm;                              // [PIX][C]
  float* s_cost = sm + PIX * a.C;                  // [PIX][D]
  const int tid = threadIdx.x;
  const int pix = tid / a.D, d = tid - pix * a.D;
  const int w0 = blockIdx.x * PIX, h = blockIdx.y, bn = blockIdx.z;
  const int w = w0 + pix;
  const bool live = pix < PIX && w < a.W;
  // this block's pixels of the current frame's feature
  for (int k = tid; k < PIX * a.C; k += blockDim.x) {
    const int pp = k / a.C, c = k - pp * a.C;
    s_curr[k] = (w0 + pp < a.W) ? a.curr[bn * a.sbn + c * a.sc + h * a.sy + (long long)(w0 + pp) * a.sx] : 0.f;
  }
  __syncthreads();
  float cost = 0.f;
  if (live) {
    // ---- gen_grid (view_transformer.py:546-573), one point
    float p[3] = {a.xs[w] - a.post_trans[bn * 3], a.ys[h] - a.post_trans[bn * 3 + 1], a.ds[d] - a.post_trans[bn * 3 + 2]};
    float q[3];
    mat3v(a.ipr + bn * 9, p, q);
    float r[3] = {q[0] * q[2], q[1] * q[2], q[2]};
    mat3v(a.comb + bn * 9, r, q);
    q[0] += a.trans[bn * 3]; q[1] += a.trans[bn * 3 + 1]; q[2] += a.trans[bn * 3 + 2];
    const bool neg = q[2] < 1e-3f;
    mat3v(a.intrins + bn * 9, q, r);
    const float u = r[0] / r[2], v = r[1] / r[2];
    const float* pr = a.post_rots + bn * 9;
    const float x = (pr[0] * u + pr[1] * v) + a.post_trans[bn * 3];
    const float y = (pr[3] * u + pr[4] * v) + a.post_trans[bn * 3 + 1];
    float px = x / (a.wi - 1.0f) * 2.0f - 1.0f;
    float py = y / (a.hi - 1.0f) * 2.0f - 1.0f;
    if (neg) { px = -2.f; py = -2.f; }
    // ---- F.grid_sample(bilinear, align_corners=True, zeros): ATen grid_sampler_2d
    const float ix = ((px + 1.f) / 2.f) * (float)(a.W - 1);
    const float iy = ((py + 1.f) / 2.f) * (float)(a.H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx1 = ix - x0f, tx0 = (x0f + 1.f) - ix, ty1 = iy - y0f, ty0 = (y0f + 1.f) - iy;
    const float wnw = tx0 * ty0, wne = tx1 * ty0, wsw = tx0 * ty1, wse = tx1 * ty1;
    // clamp far-away bases so the int conversion cannot overflow; they stay out of range
    const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)a.W), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)a.H);
    const bool vx0 = (unsigned)x0 < (unsigned)a.W, vx1 = (unsigned)(x0 + 1) < (unsigned)a.W;
    const bool vy0 = (unsigned)y0 < (unsigned)a.H, vy1 = (unsigned)(y0 + 1) < (unsigned)a.H;
    const bool vnw = vx0 && vy0, vne = vx1 && vy0, vsw = vx0 && vy1, vse = vx1 && vy1;
    const int xc0 = min(max(x0, 0), a.W - 1), xc1 = min(max(x0 + 1, 0), a.W - 1);
    const int yc0 = min(max(y0, 0), a.H - 1), yc1 = min(max(y0 + 1, 0), a.H - 1);
    const float* base = a.prev + bn * a.sbn;
    const float* pnw = base + yc0 * a.sy + xc0 * a.sx;
    const float* pne = base + yc0 * a.sy + xc1 * a.sx;
    const float* psw = base + yc1 * a.sy + xc0 * a.sx;
    const float* pse = base + yc1 * a.sy + xc1 * a.sx;
    const float* cur = s_curr + pix * a.C;
    float first_of_last_group = 0.f;
    for (int c0 = 0; c0 < a.C; c0 += 4) {             // one channel group of the reference loop (:587-596)
      float s4[4];
      if (CL) {
        const float4 nw = *reinterpret_cast<const float4*>(pnw + c0), ne = *reinterpret_cast<const float4*>(pne + c0);
        const float4 sw = *reinterpret_cast<const float4*>(psw + c0), se = *reinterpret_cast<const float4*>(pse + c0);
        const float nwv[4] = {nw.x, nw.y, nw.z, nw.w}, nev[4] = {ne.x, ne.y, ne.z, ne.w};
        const float swv[4] = {sw.x, sw.y, sw.z, sw.w}, sev[4] = {se.x, se.y, se.z, se.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float acc = 0.f;                              // ATen order: nw, ne, sw, se
          if (vnw) acc += nwv[k] * wnw;
          if (vne) acc += nev[k] * wne;
          if (vsw) acc += swv[k] * wsw;
          if (vse) acc += sev[k] * wse;
          s4[k] = acc;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const long long co = (long long)(c0 + k) * a.sc;
          float acc = 0.f;
          if (vnw) acc += pnw[co] * wnw;
          if (vne) acc += pne[co] * wne;
          if (vsw) acc += psw[co] * wsw;
          if (vse) acc += pse[co] * wse;
          s4[k] = acc;
        }
      }
      const float g = ((fabsf(cur[c0] - s4[0]) + fabsf(cur[c0 + 1] - s4[1])) + fabsf(cur[c0 + 2] - s4[2])) +
                      fabsf(cur[c0 + 3] - s4[3]);
      cost += g;
      first_of_last_group = s4[0];
    }
    if (a.bias != 0.f && first_of_last_group == 0.f) cost += a.bias;     // :597-599
    cost = -cost;
    s_cost[pix * a.D + d] = cost;
  }
  __syncthreads();
  if (live) {
    const float* row = s_cost + pix * a.D;
    float m = -3.402823466e38f;
    for (int k = 0; k < a.D; ++k) m = fmaxf(m, row[k]);
    float s = 0.f;
    for (int k = 0; k < a.D; ++k) s += expf(row[k] - m);
    a.out[(((long long)bn * a.D + d) * a.H + h) * a.W + w] = expf(cost - m) / s;
  }
}

PW_API int pw_stereo_cost_volume(const float* prev, const float* curr, int BN, int C, int H, int W,
                                 int64_t s_bn, int64_t s_c, int64_t s_y, int64_t s_x, const float* ds, int D,
                                 const float* xs, const float* ys, const float* inv_post_rot,
                                 const float* post_trans, const float* combine, const float* trans,
                                 const float* intrins, const float* post_rots, float wi, float hi, float bias,
                                 float* out, void* stream) {
  PW_CHECK_ARG(prev && curr && ds && xs && ys && inv_post_rot && post_trans && combine && trans && intrins &&
                   post_rots && out,
               "pw_stereo_cost_volume: null pointer");
  PW_CHECK_ARG(BN > 0 && C > 0 && C % 4 == 0 && H > 1 && W > 1 && D > 0 && D * PIX <= 1024,
               "pw_stereo_cost_volume: need C %% 4 == 0 and D <= 128 (got C=%d D=%d)", C, D);
  StereoArgs a;
  a.prev = prev; a.curr = curr; a.sbn = s_bn; a.sc = s_c; a.sy = s_y; a.sx = s_x;
  a.BN = BN; a.C = C; a.H = H; a.W = W; a.D = D; a.ds = ds; a.xs = xs; a.ys = ys;
  a.ipr = inv_post_rot; a.post_trans = post_trans; a.comb = combine; a.trans = trans; a.intrins = intrins;
  a.post_rots = post_rots; a.wi = wi; a.hi = hi; a.bias = bias; a.out = out;
  const unsigned nthreads = (unsigned)((D * PIX + 63) / 64 * 64);
  const size_t lds = (size_t)(PIX * C + PIX * D) * 4;
  dim3 grid((unsigned)pw_cdiv(W, PIX), (unsigned)H, (unsigned)BN);
  const bool cl = s_c == 1 && (s_x % 4 == 0) && (s_y % 4 == 0) && (s_bn % 4 == 0) && (((uintptr_t)prev & 15) == 0);
  if (cl) hipLaunchKernelGGL(k_stereo_cost_volume<true>, grid, dim3(nthreads), lds, pw_stream(stream), a);
  else hipLaunchKernelGGL(k_stereo_cost_volume<false>, grid, dim3(nthreads), lds, pw_stream(stream), a);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
