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
  long long* probe;                    // PW_STEREO_PROBE: cycles per phase of block (0, 0, 0) (tools/bench_kernels.py)
};

__device__ __forceinline__ void mat3v(const float* m, const float (&v)[3], float (&o)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = (m[r * 3] * v[0] + m[r * 3 + 1] * v[1]) + m[r * 3 + 2] * v[2];
}
// One frustum point -> the four bilinear corners in the previous frame (clamped pixel coordinates, weights, validity) in the
// reference's operation order: gen_grid (:546-573), then ATen grid_sampler_2d with bilinear / zeros / align_corners=True.
struct StereoTaps { int xc0, xc1, yc0, yc1; float w[4]; bool v[4]; };      // corners: nw, ne, sw, se

__device__ __forceinline__ StereoTaps stereo_taps(const StereoArgs& a, int bn, int h, int w, int d) {
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
  const float ix = ((px + 1.f) / 2.f) * (float)(a.W - 1);
  const float iy = ((py + 1.f) / 2.f) * (float)(a.H - 1);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float tx1 = ix - x0f, tx0 = (x0f + 1.f) - ix, ty1 = iy - y0f, ty0 = (y0f + 1.f) - iy;
  StereoTaps t;
  t.w[0] = tx0 * ty0; t.w[1] = tx1 * ty0; t.w[2] = tx0 * ty1; t.w[3] = tx1 * ty1;
  // clamp far-away bases so the int conversion cannot overflow; they stay out of range
  const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)a.W), y0 = (int)fminf(fmaxf(y0f, -2.f), (float)a.H);
  const bool vx0 = (unsigned)x0 < (unsigned)a.W, vx1 = (unsigned)(x0 + 1) < (unsigned)a.W;
  const bool vy0 = (unsigned)y0 < (unsigned)a.H, vy1 = (unsigned)(y0 + 1) < (unsigned)a.H;
  t.v[0] = vx0 && vy0; t.v[1] = vx1 && vy0; t.v[2] = vx0 && vy1; t.v[3] = vx1 && vy1;
  t.xc0 = min(max(x0, 0), a.W - 1); t.xc1 = min(max(x0 + 1, 0), a.W - 1);
  t.yc0 = min(max(y0, 0), a.H - 1); t.yc1 = min(max(y0 + 1, 0), a.H - 1);
  return t;
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
    const StereoTaps tp = stereo_taps(a, bn, h, w, d);
    const bool vnw = tp.v[0], vne = tp.v[1], vsw = tp.v[2], vse = tp.v[3];
    const float wnw = tp.w[0], wne = tp.w[1], wsw = tp.w[2], wse = tp.w[3];
    const float* base = a.prev + bn * a.sbn;
    const float* pnw = base + tp.yc0 * a.sy + tp.xc0 * a.sx;
    const float* pne = base + tp.yc0 * a.sy + tp.xc1 * a.sx;
    const float* psw = base + tp.yc1 * a.sy + tp.xc0 * a.sx;
    const float* pse = base + tp.yc1 * a.sy + tp.xc1 * a.sx;
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

// ------------------------------------------------------------------------------------
// channels-last fast path: the previous frame's footprint of a pixel tile staged in LDS.
// PMC of the point-per-lane kernel above at the reference shape: 1.58 G cache-line lookups in the vector L1 (a load
// instruction touches up to 64 lines and uses 16 bytes of each) = 84 % of the kernel's cycles at one lookup per clock, and
// 31 GB from L2 on top of that: every prev pixel is fetched again by each of the ~4 points per depth bin that touch it,
// and again by the neighbouring bins.  Here a block owns an 8 x 8 pixel tile of one camera and walks the D bins:
//   * geometry of 8 bins x 64 pixels at a time (one point per lane, stereo_taps) -> LDS, with the bounding box of the
//     clamped corner pixels per bin;
//   * consecutive bins are grouped while the union of their boxes stays <= ST_CAP prev pixels (far bins move by a
//     fraction of a pixel: typically 3-8 bins share one box); the box is copied to LDS once, 512 contiguous bytes per pixel;
//   * a half-wave owns a point, lane g channel group g: the four corners are four ds_read_b128 of contiguous 512-byte
//     rows, the L1 cost of the group is summed over the half-wave by a DPP / bpermute tree (NOT the reference's serial
//     group order: differences at the 1e-7 level of the cost, the tests hold both kernels to the same tolerance);
//   * a bin whose box alone exceeds ST_CAP (near bins under strong parallax) gathers its corners from global memory instead;
//   * the negated costs go to `out`, and the wave that wrote a pixel's D values reads them back for the softmax.
// ------------------------------------------------------------------------------------
namespace {
constexpr int ST_TP = 8;             // tile edge (pixels); wave = tile row, lane pair = pixel
constexpr int ST_DC = 8;             // depth bins per geometry chunk
constexpr int ST_CAP = 120;          // prev pixels staged at once
constexpr int ST_THREADS = 512;
struct __attribute__((aligned(16))) StereoGeo { int xc0, xc1, yc0, yc1; float w[4]; };   // xc0 < 0: no corner inside the map
struct __attribute__((aligned(16))) StereoBox { int x0, y0, x1, y1; };                   // empty: x1 < x0

__device__ __forceinline__ StereoBox box_union(const StereoBox& a, const StereoBox& b) {
  return StereoBox{min(a.x0, b.x0), min(a.y0, b.y0), max(a.x1, b.x1), max(a.y1, b.y1)};
}
__device__ __forceinline__ int box_area(const StereoBox& b) { return b.x1 < b.x0 ? 0 : (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1); }

// sum over the 32 lanes of a half-wave; every lane ends up with a total
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm 1,0,3,2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm 2,3,0,1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __shfl_xor(v, 16);
  return v;
}

typedef float st_f2 __attribute__((ext_vector_type(2)));

// L1 cost of one channel group (4 channels) of one point; written on float pairs so that it compiles to v_pk_mul_f32 /
// v_pk_add_f32 (no contraction: the file is built with -ffp-contract=off)
__device__ __forceinline__ float stereo_group_cost(const float4& cu, const float4 (&c)[4], const float (&w)[4], float& first) {
  const st_f2 w0 = {w[0], w[0]}, w1 = {w[1], w[1]}, w2 = {w[2], w[2]}, w3 = {w[3], w[3]};
  const st_f2 sa = ((st_f2{c[0].x, c[0].y} * w0 + st_f2{c[1].x, c[1].y} * w1) + st_f2{c[2].x, c[2].y} * w2) + st_f2{c[3].x, c[3].y} * w3;
  const st_f2 sb = ((st_f2{c[0].z, c[0].w} * w0 + st_f2{c[1].z, c[1].w} * w1) + st_f2{c[2].z, c[2].w} * w2) + st_f2{c[3].z, c[3].w} * w3;
  const st_f2 da = st_f2{cu.x, cu.y} - sa, db = st_f2{cu.z, cu.w} - sb;       // ATen corner order nw, ne, sw, se above
  first = sa[0];
  return ((fabsf(da[0]) + fabsf(da[1])) + fabsf(db[0])) + fabsf(db[1]);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) v = fmaxf(v, __shfl_xor(v, sft));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) v += __shfl_xor(v, sft);
  return v;
}
typedef __amdgpu_buffer_rsrc_t st_rsrc;
constexpr unsigned ST_OOB = 0xfffffff0u;               // lane offset beyond num_records: the load returns 0, the store is dropped
__device__ __forceinline__ st_rsrc st_make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 st_buf_load4(st_rsrc r, unsigned voff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);   // keep `auto` (see pw_conv3d_common.h)
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

struct StereoRun {                                       // block-uniform state of one staging run
  const float* s_stage; const StereoGeo* geo;            // geo: this wave's [ST_DC][8] entries
  st_rsrc prevr, outr;                                   // this camera's prev features / the whole output
  unsigned out_lane;                                     // byte offset of (bn, d = 0, hrow, w0) in out; ST_OOB for lanes that never store
  unsigned sy4, sx4, HW4;                                // byte strides
  int bx0, by0, nx, d0, D, W, w0, G;
  float bias;
};

// bins [dl0, dl1) of a run; the two points a lane pair handles per step are two independent straight-line chains
// (geometry -> 4 corner loads -> group cost -> lane tree -> store through an out-of-range offset for non-writers), so that
// they interleave and nothing in the loop body branches: with a branch per point the chains ran one after the other
template <int CT, bool DIRECT>
__device__ __forceinline__ void stereo_run_bins(const StereoRun& r, const float4 (&cur)[4], int dl0, int dl1, int hl, int g) {
  const int C = CT ? CT : 4 * r.G;
  const int gi = min(g, r.G - 1);
  for (int dl = dl0; dl < dl1; ++dl) {
    const int d = r.d0 + dl;
    if (d >= r.D) break;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      StereoGeo ge[2];
      float4 c[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k) ge[k] = r.geo[dl * 8 + 2 * (2 * kk + k) + hl];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool ok = ge[k].xc0 >= 0;
        if constexpr (DIRECT) {
          const unsigned o0 = (unsigned)ge[k].yc0 * r.sy4 + 16u * (unsigned)gi, o1 = (unsigned)ge[k].yc1 * r.sy4 + 16u * (unsigned)gi;
          const unsigned x0 = (unsigned)ge[k].xc0 * r.sx4, x1 = (unsigned)ge[k].xc1 * r.sx4;
          c[k][0] = st_buf_load4(r.prevr, ok ? o0 + x0 : ST_OOB); c[k][1] = st_buf_load4(r.prevr, ok ? o0 + x1 : ST_OOB);
          c[k][2] = st_buf_load4(r.prevr, ok ? o1 + x0 : ST_OOB); c[k][3] = st_buf_load4(r.prevr, ok ? o1 + x1 : ST_OOB);
        } else {
          const int r0 = (ge[k].yc0 - r.by0) * r.nx - r.bx0, r1 = (ge[k].yc1 - r.by0) * r.nx - r.bx0;
          const float* b = r.s_stage + 4 * gi;
          c[k][0] = *reinterpret_cast<const float4*>(b + (ok ? r0 + ge[k].xc0 : ST_CAP) * C);      // row ST_CAP: zeros
          c[k][1] = *reinterpret_cast<const float4*>(b + (ok ? r0 + ge[k].xc1 : ST_CAP) * C);
          c[k][2] = *reinterpret_cast<const float4*>(b + (ok ? r1 + ge[k].xc0 : ST_CAP) * C);
          c[k][3] = *reinterpret_cast<const float4*>(b + (ok ? r1 + ge[k].xc1 : ST_CAP) * C);
        }
      }
      float cost[2], first[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float gc = stereo_group_cost(cur[2 * kk + k], c[k], ge[k].w, first[k]);
        cost[k] = g < r.G ? gc : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) cost[k] = half_wave_sum(cost[k]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = 2 * (2 * kk + k) + hl;
        const float cst = cost[k] + ((r.bias != 0.f && first[k] == 0.f) ? r.bias : 0.f);      // :597-599 (x + 0 = x)
        const unsigned off = (r.out_lane != ST_OOB && r.w0 + px < r.W) ? r.out_lane + (unsigned)d * r.HW4 + 4u * (unsigned)px : ST_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-cst), r.outr, off, 0, 0);
      }
    }
  }
}
}  // namespace

template <int CT>     // CT: the channel count when it is the reference's 128 (row addresses become shifts), 0 = any C <= 128
__global__ void __launch_bounds__(ST_THREADS, 4) k_stereo_cost_volume_tile(StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = CT ? CT : a.C;
  float* s_stage = sm;                                                             // [ST_CAP + 1][C], last row zeros
  StereoGeo* s_geo = reinterpret_cast<StereoGeo*>(sm + (ST_CAP + 1) * C);              // [8 waves][ST_DC][8 pixels]
  StereoBox* s_box = reinterpret_cast<StereoBox*>(s_geo + 8 * 64);                 // [8 waves][ST_DC]
  int* s_plan = reinterpret_cast<int*>(s_box + 64);                                // count, then {dl0, dl1, x0, y0, nx, ny} x <= 8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hl = lane >> 5, g = lane & 31, G = C >> 2;
  for (int k = tid; k < C; k += ST_THREADS) s_stage[ST_CAP * C + k] = 0.f;
  const int w0 = blockIdx.x * ST_TP, hrow = blockIdx.y * ST_TP + wave, bn = blockIdx.z;
  const bool row_ok = hrow < a.H;
  const float* prevb = a.prev + bn * a.sbn;
  const long long HW = (long long)a.H * a.W;
  float* outb = a.out + (long long)bn * a.D * HW + (long long)hrow * a.W;          // + d * HW + w

  float4 cur[4];                                  // this lane's channel group of pixels 2 k + hl of the wave's row
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int w = w0 + 2 * k + hl;
    cur[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok && w < a.W && g < G)
      cur[k] = *reinterpret_cast<const float4*>(a.curr + bn * a.sbn + hrow * a.sy + (long long)w * a.sx + 4 * g);
  }

  StereoRun run;
  run.s_stage = s_stage; run.geo = s_geo + wave * 64;
  run.prevr = st_make_rsrc(prevb, (unsigned)(((long long)(a.H - 1) * a.sy + (long long)(a.W - 1) * a.sx + C) * 4));
  run.outr = st_make_rsrc(a.out, (unsigned)((long long)a.BN * a.D * HW * 4));
  run.out_lane = (g == G - 1 && row_ok) ? (unsigned)((((long long)bn * a.D * a.H + hrow) * a.W + w0) * 4) : ST_OOB;
  run.sy4 = (unsigned)(a.sy * 4); run.sx4 = (unsigned)(a.sx * 4); run.HW4 = (unsigned)(HW * 4);
  run.D = a.D; run.W = a.W; run.w0 = w0; run.G = G; run.bias = a.bias;
  long long tq[6] = {0, 0, 0, 0, 0, 0}, tk = a.probe ? __builtin_readcyclecounter() : 0;
#define ST_TICK(i) if (a.probe) { const long long now_ = __builtin_readcyclecounter(); tq[i] += now_ - tk; tk = now_; }
  for (int d0 = 0; d0 < a.D; d0 += ST_DC) {
    {   // geometry of this wave's 8 pixels x 8 bins, one point per lane
      const int dl = lane >> 3, px = lane & 7, d = d0 + dl, w = w0 + px;
      StereoGeo ge;
      ge.xc0 = -1; ge.xc1 = ge.yc0 = ge.yc1 = 0; ge.w[0] = ge.w[1] = ge.w[2] = ge.w[3] = 0.f;
      StereoBox bb = {0x7fffffff, 0x7fffffff, -1, -1};
      if (d < a.D && row_ok && w < a.W) {
        const StereoTaps tp = stereo_taps(a, bn, hrow, w, d);
        if (tp.v[0] || tp.v[1] || tp.v[2] || tp.v[3]) {
          ge.xc0 = tp.xc0; ge.xc1 = tp.xc1; ge.yc0 = tp.yc0; ge.yc1 = tp.yc1;
#pragma unroll
          for (int k = 0; k < 4; ++k) ge.w[k] = tp.v[k] ? tp.w[k] : 0.f;      // an invalid corner adds value * 0
          bb = StereoBox{tp.xc0, tp.yc0, tp.xc1, tp.yc1};
        }
      }
      s_geo[wave * 64 + lane] = ge;
#pragma unroll
      for (int sft = 1; sft < 8; sft <<= 1)
        bb = box_union(bb, StereoBox{__shfl_xor(bb.x0, sft), __shfl_xor(bb.y0, sft), __shfl_xor(bb.x1, sft), __shfl_xor(bb.y1, sft)});
      if (px == 0) s_box[wave * ST_DC + dl] = bb;
    }
    __syncthreads();
    ST_TICK(0)
    if (wave == 0) {   // plan: greedy runs of bins whose boxes share one staging buffer (scalar code on readlane values)
      StereoBox u = {0x7fffffff, 0x7fffffff, -1, -1};
      if (lane < ST_DC)
        for (int wv = 0; wv < 8; ++wv) u = box_union(u, s_box[wv * ST_DC + lane]);
      StereoBox cu = {0x7fffffff, 0x7fffffff, -1, -1};
      int n = 0, dl0 = 0;
#pragma unroll
      for (int k = 0; k <= ST_DC; ++k) {
        StereoBox uk = cu;
        if (k < ST_DC)
          uk = StereoBox{__builtin_amdgcn_readlane(u.x0, k & 7), __builtin_amdgcn_readlane(u.y0, k & 7),
                         __builtin_amdgcn_readlane(u.x1, k & 7), __builtin_amdgcn_readlane(u.y1, k & 7)};
        const StereoBox nu = box_union(cu, uk);
        if (k == ST_DC || (k > dl0 && box_area(nu) > ST_CAP)) {
          if (lane == 0) {
            int* e = s_plan + 1 + 6 * n;
            const int ar = box_area(cu);
            e[0] = dl0; e[1] = k; e[2] = cu.x0; e[3] = cu.y0;
            e[4] = ar == 0 ? 0 : (ar > ST_CAP ? -1 : cu.x1 - cu.x0 + 1);
            e[5] = ar == 0 ? 0 : cu.y1 - cu.y0 + 1;
          }
          ++n; dl0 = k; cu = uk;
        } else {
          cu = nu;
        }
      }
      if (lane == 0) s_plan[0] = n;
    }
    __syncthreads();
    ST_TICK(1)
    const int nplan = s_plan[0];
    for (int sidx = 0; sidx < nplan; ++sidx) {
      const int* e = s_plan + 1 + 6 * sidx;
      const int dl0 = e[0], dl1 = e[1], bx0 = e[2], by0 = e[3], nx = e[4], ny = e[5];
      const bool direct = nx < 0;                   // block-uniform
      if (nx > 0) {
        const int total = nx * ny * G;
        const float rnx = 1.0f / (float)nx;
        for (int idx = tid; idx < total; idx += ST_THREADS) {
          const int slot = CT ? idx >> 5 : idx / G, gg = CT ? idx & 31 : idx - slot * G;
          const int yy = (int)(((float)slot + 0.5f) * rnx), xx = slot - yy * nx;      // exact: slot, nx <= ST_CAP
          *reinterpret_cast<float4*>(s_stage + slot * C + 4 * gg) =
              *reinterpret_cast<const float4*>(prevb + (by0 + yy) * a.sy + (long long)(bx0 + xx) * a.sx + 4 * gg);
        }
      }
      __syncthreads();
      ST_TICK(2)
      run.bx0 = bx0; run.by0 = by0; run.nx = nx; run.d0 = d0;
      if (direct) stereo_run_bins<CT, true>(run, cur, dl0, dl1, hl, g);
      else stereo_run_bins<CT, false>(run, cur, dl0, dl1, hl, g);
      ST_TICK(3)
      __syncthreads();                               // the staging buffer and the plan are reused
      ST_TICK(4)
    }
  }

  // softmax over D: this wave wrote its row's costs itself
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // same wave, other lanes: no L2 write-back needed
  if (row_ok) {
    float v[ST_TP][2];
#pragma unroll
    for (int px = 0; px < ST_TP; ++px) {               // all 16 loads of the lane go out before the first reduction
      const bool in = w0 + px < a.W;
      v[px][0] = (in && lane < a.D) ? outb[(long long)lane * HW + w0 + px] : -3.402823466e38f;
      v[px][1] = (in && lane + 64 < a.D) ? outb[(long long)(lane + 64) * HW + w0 + px] : -3.402823466e38f;
    }
#pragma unroll
    for (int px = 0; px < ST_TP; ++px) {
      const bool in = w0 + px < a.W;
      const float m = wave_max(fmaxf(v[px][0], v[px][1]));
      const float e0 = lane < a.D ? expf(v[px][0] - m) : 0.f, e1 = lane + 64 < a.D ? expf(v[px][1] - m) : 0.f;
      const float sum = wave_sum(e0 + e1);
      if (in && lane < a.D) outb[(long long)lane * HW + w0 + px] = e0 / sum;
      if (in && lane + 64 < a.D) outb[(long long)(lane + 64) * HW + w0 + px] = e1 / sum;
    }
  }
  ST_TICK(5)
#undef ST_TICK
  if (a.probe && lane == 0 && blockIdx.x == 3 && blockIdx.y == 3 && blockIdx.z == 0)
  {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.probe[wave * 6 + i] = tq[i];
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
  PW_CHECK_ARG(s_bn >= 0 && s_c >= 0 && s_y >= 0 && s_x >= 0, "pw_stereo_cost_volume: strides must be non-negative");
  StereoArgs a;
  a.prev = prev; a.curr = curr; a.sbn = s_bn; a.sc = s_c; a.sy = s_y; a.sx = s_x;
  a.BN = BN; a.C = C; a.H = H; a.W = W; a.D = D; a.ds = ds; a.xs = xs; a.ys = ys;
  a.ipr = inv_post_rot; a.post_trans = post_trans; a.comb = combine; a.trans = trans; a.intrins = intrins;
  a.post_rots = post_rots; a.wi = wi; a.hi = hi; a.bias = bias; a.out = out;
  a.probe = nullptr;
  if (const char* e = getenv("PW_STEREO_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  const bool cl = s_c == 1 && (s_x % 4 == 0) && (s_y % 4 == 0) && (s_bn % 4 == 0) && (((uintptr_t)prev & 15) == 0) &&
                  (((uintptr_t)curr & 15) == 0);
  const bool fits32 = (long long)BN * D * H * W * 4 < (1ll << 32) &&            // the tiled kernel addresses through buffer descriptors
                      ((long long)(H - 1) * s_y + (long long)(W - 1) * s_x + C) * 4 < (1ll << 32);
  if (cl && fits32 && C <= 128) {
    const size_t lds = (size_t)(ST_CAP + 1) * C * 4 + 8 * 64 * sizeof(StereoGeo) + 64 * sizeof(StereoBox) + 64 * 4;
    constexpr int lds_max = (ST_CAP + 1) * 128 * 4 + 8 * 64 * 32 + 64 * 16 + 64 * 4;
    static int once = [] {
      int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_cost_volume_tile<128>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
      if (e == 0)
        e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_cost_volume_tile<0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
      return e;
    }();
    PW_CHECK_HIP((hipError_t)once);
    dim3 grid((unsigned)pw_cdiv(W, ST_TP), (unsigned)pw_cdiv(H, ST_TP), (unsigned)BN);
    if (C == 128) hipLaunchKernelGGL(k_stereo_cost_volume_tile<128>, grid, dim3(ST_THREADS), lds, pw_stream(stream), a);
    else hipLaunchKernelGGL(k_stereo_cost_volume_tile<0>, grid, dim3(ST_THREADS), lds, pw_stream(stream), a);
    pw_note_kernel("k_stereo_cost_volume_tile");
  } else {
    const unsigned nthreads = (unsigned)((D * PIX + 63) / 64 * 64);
    const size_t lds = (size_t)(PIX * C + PIX * D) * 4;
    dim3 grid((unsigned)pw_cdiv(W, PIX), (unsigned)H, (unsigned)BN);
    if (cl) hipLaunchKernelGGL(k_stereo_cost_volume<true>, grid, dim3(nthreads), lds, pw_stream(stream), a);
    else hipLaunchKernelGGL(k_stereo_cost_volume<false>, grid, dim3(nthreads), lds, pw_stream(stream), a);
    pw_note_kernel("k_stereo_cost_volume<%s>", cl ? "true" : "false");
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
