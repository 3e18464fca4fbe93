// One frame's lift + voxel pooling as a VOXEL-DRIVEN GATHER (round 5): no atomics, no per-voxel slots, no sort, three launches, every
// row of the pooled grid written exactly once.  Same results, bit for bit, as pw_lss_lift_pool (pw_lss_fused.hip) and the sort form
// (pw_lss.hip); replaces, like them,
//   mmdet3d/models/necks/view_transformer.py:114-153  get_lidar_coor
//   mmdet3d/models/necks/view_transformer.py:203-261  voxel_pooling_prepare_v2
//   mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48  bev_pool_v2 forward
//
// The reference (and rounds 1-4 here) goes point -> voxel: every frustum point computes its voxel and the points of a voxel are then
// brought together -- by an argsort there, by returning atomics + id slots in pw_lss_fused.hip (180 MB of HBM traffic for 90 MB of
// algorithmic bytes: slots, ranks, counters, overflow lists).  This file goes voxel -> points: a voxel projects its box into every
// camera, which bounds the depth planes and the feature pixels whose frustum points CAN fall into it; each candidate (camera, depth
// plane, pixel) is then put through the reference's own forward arithmetic (lss_voxel_of_fr, the oracle's op order, fp-contract off)
// and counts iff it lands in this voxel.  The bounds only have to be a superset (they are interval bounds with margins: 2e-3 of a
// voxel, 0.02 of a pixel / depth step, against fp32 errors of 1e-5); membership is decided by the exact test, so the point sets are
// identical to the reference's.  Candidates are enumerated in ascending point id (camera, depth, row, column), so a voxel's sum runs in
// ascending point order with one non-contracted multiply-add per point -- the order that makes the pooled fp32 sums bit-identical to
// the oracle -- without any sort.  At the C3 shape: 1.54 M candidate tests for 880 k kept points (of 1.49 M frustum points).
//
//   k_lssg_prologue  per camera: forward matrices (as pw_lss_fused.hip) + the inverse ones of the bounds; checks the preconditions
//   k_lssg_main      lane = voxel: candidate count over all cameras; voxels with <= 24 candidates are enumerated at once (hits to LDS),
//                    then eight lanes per voxel gather the rows and add them in order; empty voxels get their zero row here; voxels
//                    with more candidates (or more than 8 hits) go on one of two lists (one atomic per wave and list)
//   k_lssg_heavy     listed voxels: a wave (<= 160 candidates) or a block each; candidates tested 64 / 256 at a time, hits compacted
//                    in order into LDS, rows requested together, added in order
// Preconditions (checked on the device by the prologue; a violation POISONS the output with NaN instead of producing wrong sums):
// cam2img is a pinhole matrix [[fx, s, cx], [0, fy, cy], [0, 0, 1]] and post_rot / post_trans are an image-plane augmentation
// ([[a, b, 0], [c, d, 0], [0, 0, 1]], z translation 0) -- what datasets/pipelines/loading.py:988-1000 produces.  The frustum must be
// separable (x depends on the column, y on the row, depth on the plane only: create_frustum, view_transformer.py:84-112); the host
// wrapper checks that once per frustum tensor and takes pw_lss_lift_pool otherwise.
// HBM-bound integer/byte work: no MFMA.  Compiled with -ffp-contract=off (see pw_lss_common.h).
#include "pw_lss_common.h"

namespace {

constexpr int LPV = 8;                // lanes per voxel (float4 of channels per lane): C = 32
constexpr int GROUPS = 64 / LPV;
constexpr int GL_SLOTS = 16;          // hits a light voxel may hold (two rounds of the eight-lane sweep)
constexpr int GL_TL = 32;             // candidates up to which a voxel is enumerated by its own lane
constexpr int GL_TC = 160;            // candidates above which a voxel gets a block instead of a wave
constexpr int GB_LIST = 256;          // wave-per-voxel: hits held in LDS between two flushes
constexpr int G_SEGS = 64;            // (camera, depth plane) candidate segments of a listed voxel held at a time
constexpr int GC_LIST = 2048;         // block-per-voxel: likewise
constexpr int GC_BLOCKS = 256, GB_BLOCKS = 768;
constexpr int HC_ROWS = 128;          // rows per LDS chunk of the block-pooled voxels (16 KB)
constexpr float G_EPS = 2e-3f;        // box margin, voxels
constexpr float G_DEL = 0.02f;        // index margin, pixels / depth steps
enum { H_BAD = 0, H_NB = 1, H_NC = 2, H_WORDS = 8 };

struct CamG {                         // 32 floats per camera
  float Wc[9];                        // (bda R)^-1: ego (post-bda) offsets -> camera axes
  float Oc[3];                        // bda T: camera centre in ego coordinates
  float au[3], av[3];                 // (Xc / Zc, Yc / Zc, 1) -> feature column / row INDEX: post_rot K and the frustum's pixel grid folded
  float pad[14];
};

struct FrScale { float xstep, ystep, d0, dstep, x0, y0; };

struct FeatIdx {
  int DHW, HW;
  float inv_dhw, inv_hw;
};

__device__ __forceinline__ int feat_index(int id, const FeatIdx& fi) {
  int cam = (int)((float)id * fi.inv_dhw);
  int p = id - cam * fi.DHW;
  if (p < 0) { --cam; p += fi.DHW; } else if (p >= fi.DHW) { ++cam; p -= fi.DHW; }
  const int d = (int)((float)p * fi.inv_hw);
  int hw = p - d * fi.HW;
  if (hw < 0) hw += fi.HW; else if (hw >= fi.HW) hw -= fi.HW;
  return cam * fi.HW + hw;
}

__device__ __forceinline__ float4 half_to_quads(float acc, int lane) {
  const int src = (lane & 32) + 4 * (lane & 7);
  float4 f;
  f.x = __shfl(acc, src, 64);
  f.y = __shfl(acc, src + 1, 64);
  f.z = __shfl(acc, src + 2, 64);
  f.w = __shfl(acc, src + 3, 64);
  return f;
}

// 3x3 inverse in double (prologue only)
__device__ void inv3x3d(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
  const double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
  const double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
  const double r = 1.0 / (a * A + b * D + c * G);
  o[0] = A * r; o[1] = B * r; o[2] = C * r; o[3] = D * r; o[4] = E * r; o[5] = F * r; o[6] = G * r; o[7] = H * r; o[8] = I * r;
}

struct Geo {                          // everything a kernel needs to bound and test candidates
  const CamG* cams;
  const float *ipr, *ptr, *comb, *trn, *bda;
  GridParams gp;
  FrScale fs;
  int N, D, H, W;
};

// voxel v -> batch element and its box (centre / half extents in ego coordinates, margins included).  Cell 0 of an axis is two cells
// wide: .long() truncates toward zero, so (-1, 0) lands in cell 0 too (view_transformer.py:228)
struct VoxBox { int b; float oc[3], hf[3]; };
__device__ __forceinline__ VoxBox vox_box(int v, const GridParams& gp) {
  const unsigned uv = (unsigned)v;
  const unsigned t1 = uv / (unsigned)gp.gx;
  const int x = (int)(uv - t1 * (unsigned)gp.gx);
  const unsigned t2 = t1 / (unsigned)gp.gy;
  const int y = (int)(t1 - t2 * (unsigned)gp.gy);
  const unsigned t3 = t2 / (unsigned)gp.gz;
  const int z = (int)(t2 - t3 * (unsigned)gp.gz);
  VoxBox q;
  q.b = (int)t3;
  const int idx[3] = {x, y, z};
  const float lw[3] = {gp.lx, gp.ly, gp.lz}, iv[3] = {gp.ix, gp.iy, gp.iz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = lw[a] + (float)(idx[a] - (idx[a] == 0 ? 1 : 0)) * iv[a], hi = lw[a] + (float)(idx[a] + 1) * iv[a];
    q.oc[a] = 0.5f * (lo + hi);
    q.hf[a] = 0.5f * (hi - lo) + G_EPS * fabsf(iv[a]);
  }
  return q;
}

// the box in camera axes: centre pc, half extents hc (L1 bound of a linear map); depth planes k0 .. k1 it can reach
struct CamBox { float X0, X1, Y0, Y1; int k0, k1; };
__device__ __forceinline__ CamBox cam_box(const CamG& c, const VoxBox& q, const Geo& g) {
  const float dx = q.oc[0] - c.Oc[0], dy = q.oc[1] - c.Oc[1], dz = q.oc[2] - c.Oc[2];
  float pc[3], hc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pc[r] = c.Wc[3 * r] * dx + c.Wc[3 * r + 1] * dy + c.Wc[3 * r + 2] * dz;
    hc[r] = fabsf(c.Wc[3 * r]) * q.hf[0] + fabsf(c.Wc[3 * r + 1]) * q.hf[1] + fabsf(c.Wc[3 * r + 2]) * q.hf[2];
  }
  CamBox o;
  o.X0 = pc[0] - hc[0]; o.X1 = pc[0] + hc[0]; o.Y0 = pc[1] - hc[1]; o.Y1 = pc[1] + hc[1];
  const float inv = __builtin_amdgcn_rcpf(g.fs.dstep);
  const float a0 = (pc[2] - hc[2] - g.fs.d0) * inv, a1 = (pc[2] + hc[2] - g.fs.d0) * inv;
  const float f0 = fminf(a0, a1) - G_DEL, f1 = fmaxf(a0, a1) + G_DEL;
  // (clamped in float first: the conversions below must not see values beyond the int range)
  o.k0 = (int)ceilf(fminf(fmaxf(f0, 0.f), (float)g.D));
  o.k1 = (int)floorf(fmaxf(fminf(f1, (float)(g.D - 1)), -1.f));
  return o;
}

// feature pixels of depth plane k the box can project to: [w0, w1] x [h0, h1] (empty when w0 > w1 or h0 > h1).  Interval bounds
// of the affine map (Xc, Yc) / D_k -> (column, row) index that the prologue folded per camera; v_rcp_f32's 1 ulp is far inside G_DEL.
struct PlaneBox { int w0, w1, h0, h1; };
__device__ __forceinline__ PlaneBox plane_box(const CamG& c, const CamBox& cb, int k, const Geo& g) {
  const float Dk = g.fs.d0 + (float)k * g.fs.dstep;
  PlaneBox o;
  if (!(Dk > 1e-3f)) {                // no pinhole projection of this plane: the whole image is the candidate set
    o.w0 = 0; o.w1 = g.W - 1; o.h0 = 0; o.h1 = g.H - 1;
    return o;
  }
  const float r = __builtin_amdgcn_rcpf(Dk);
  const float rx0 = cb.X0 * r, rx1 = cb.X1 * r, ry0 = cb.Y0 * r, ry1 = cb.Y1 * r;
  const float ua = c.au[0] * rx0, ub = c.au[0] * rx1, uc = c.au[1] * ry0, ud = c.au[1] * ry1;
  const float va = c.av[0] * rx0, vb = c.av[0] * rx1, vc = c.av[1] * ry0, vd = c.av[1] * ry1;
  const float ul = c.au[2] + fminf(ua, ub) + fminf(uc, ud) - G_DEL, uh = c.au[2] + fmaxf(ua, ub) + fmaxf(uc, ud) + G_DEL;
  const float vl = c.av[2] + fminf(va, vb) + fminf(vc, vd) - G_DEL, vh = c.av[2] + fmaxf(va, vb) + fmaxf(vc, vd) + G_DEL;
  o.w0 = (int)ceilf(fminf(fmaxf(ul, 0.f), (float)g.W));
  o.w1 = (int)floorf(fmaxf(fminf(uh, (float)(g.W - 1)), -1.f));
  o.h0 = (int)ceilf(fminf(fmaxf(vl, 0.f), (float)g.H));
  o.h1 = (int)floorf(fmaxf(fminf(vh, (float)(g.H - 1)), -1.f));
  return o;
}

__device__ __forceinline__ int plane_count(const PlaneBox& p) {
  return (p.w0 <= p.w1 && p.h0 <= p.h1) ? (p.w1 - p.w0 + 1) * (p.h1 - p.h0 + 1) : 0;
}

// the reference's forward arithmetic for frustum entry (cam, k, h, w): does it land in voxel v?
__device__ __forceinline__ bool lands_in(int v, int cam, int b, int k, int h, int w, const float* __restrict__ tabs, const Geo& g) {
  const float fr0 = tabs[w], fr1 = tabs[g.W + h], fr2 = tabs[g.W + g.H + k];
  return lss_voxel_of_fr(fr0, fr1, fr2, cam, b, g.ipr, g.ptr, g.comb, g.trn, g.bda, g.gp, nullptr) == v;
}

__device__ __forceinline__ void load_tables(float* tabs, const float* __restrict__ frustum, const Geo& g) {
  for (int i = threadIdx.x; i < g.W + g.H + g.D; i += blockDim.x) {
    float v;
    if (i < g.W) v = frustum[(size_t)i * 3];
    else if (i < g.W + g.H) v = frustum[(size_t)(i - g.W) * g.W * 3 + 1];
    else v = frustum[(size_t)(i - g.W - g.H) * g.H * g.W * 3 + 2];
    tabs[i] = v;
  }
}

// Per-camera constants in LDS (one copy per block): read through s_load they cost one memory latency per camera visit and matrix --
// 30-50 us of a kernel that has two to four waves per SIMD to hide them.  Layout as in global memory, so lss_voxel_of_fr indexes them
// the same way (the host entry point limits B * N to G_STAGE_MAX_BN).
constexpr int G_STAGE_MAX_BN = 32;
__host__ __device__ __forceinline__ int stage_floats(int B, int BN) { return BN * (32 + 9 + 9 + 3 + 3) + B * 9; }
__device__ __forceinline__ void stage_consts(float* dst, int B, int BN, Geo& g) {
  float* cams = dst;
  float* ipr = cams + BN * 32;
  float* comb = ipr + BN * 9;
  float* trn = comb + BN * 9;
  float* ptr = trn + BN * 3;
  float* bda = ptr + BN * 3;
  const float* gc = reinterpret_cast<const float*>(g.cams);
  for (int i = threadIdx.x; i < BN * 32; i += blockDim.x) cams[i] = gc[i];
  for (int i = threadIdx.x; i < BN * 9; i += blockDim.x) { ipr[i] = g.ipr[i]; comb[i] = g.comb[i]; }
  for (int i = threadIdx.x; i < BN * 3; i += blockDim.x) { trn[i] = g.trn[i]; ptr[i] = g.ptr[i]; }
  for (int i = threadIdx.x; i < B * 9; i += blockDim.x) bda[i] = g.bda[i];
  g.cams = reinterpret_cast<const CamG*>(cams);
  g.ipr = ipr; g.comb = comb; g.trn = trn; g.ptr = ptr; g.bda = bda;
}

}  // namespace

__global__ void __launch_bounds__(256)
k_lssg_prologue(int B, int N, int D, int H, int W, const float* __restrict__ frustum, const float* __restrict__ s2e,
                const float* __restrict__ K, const float* __restrict__ pr, const float* __restrict__ pt, const float* __restrict__ bda,
                float* __restrict__ ipr, float* __restrict__ comb, float* __restrict__ tr, CamG* __restrict__ cams,
                float* __restrict__ fscale, int32_t* __restrict__ hdr) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const int BN = B * N;
  for (int c = threadIdx.x; c < BN; c += blockDim.x) {
    lss_camera_matrix_one(c, s2e, K, pr, ipr, comb, tr);
    const float* S = s2e + c * 16;
    const float* A = bda + (c / N) * 9;
    double AR[9], Wd[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += (double)A[i * 3 + k] * (double)S[k * 4 + j];
        AR[i * 3 + j] = a;
      }
    inv3x3d(AR, Wd);
    CamG g;
    for (int i = 0; i < 9; ++i) g.Wc[i] = (float)Wd[i];
    for (int i = 0; i < 3; ++i)
      g.Oc[i] = (float)((double)A[i * 3] * S[3] + (double)A[i * 3 + 1] * S[7] + (double)A[i * 3 + 2] * S[11]);
    const float* Kc = K + c * 9;
    const float* P = pr + c * 9;
    const float* T = pt + c * 3;
    {  // column index = ((P00 (fx rx + s ry + cx) + P01 (fy ry + cy) + PT0) - x0) / xstep, likewise the row index
      const double fx = Kc[0], sk = Kc[1], cx = Kc[2], fy = Kc[4], cy = Kc[5];
      const double xstep = W > 1 ? (double)frustum[3] - (double)frustum[0] : 1.0, ystep = H > 1 ? (double)frustum[(size_t)W * 3 + 1] - (double)frustum[1] : 1.0;
      const double x0 = frustum[0], y0 = frustum[1];
      g.au[0] = (float)(P[0] * fx / xstep); g.au[1] = (float)((P[0] * sk + P[1] * fy) / xstep);
      g.au[2] = (float)((P[0] * cx + P[1] * cy + T[0] - x0) / xstep);
      g.av[0] = (float)(P[3] * fx / ystep); g.av[1] = (float)((P[3] * sk + P[4] * fy) / ystep);
      g.av[2] = (float)((P[3] * cx + P[4] * cy + T[1] - y0) / ystep);
    }
    for (int i = 0; i < 14; ++i) g.pad[i] = 0.f;
    cams[c] = g;
    bool ok = Kc[3] == 0.f && Kc[6] == 0.f && Kc[7] == 0.f && Kc[8] == 1.f;
    ok = ok && P[2] == 0.f && P[5] == 0.f && P[6] == 0.f && P[7] == 0.f && P[8] == 1.f && T[2] == 0.f;
    for (int i = 0; i < 9; ++i) ok = ok && isfinite(g.Wc[i]);
    for (int i = 0; i < 3; ++i) ok = ok && isfinite(g.au[i]) && isfinite(g.av[i]);
    if (!ok) atomicOr(&bad, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float xstep = W > 1 ? frustum[3] - frustum[0] : 1.f;
    const float ystep = H > 1 ? frustum[(size_t)W * 3 + 1] - frustum[1] : 1.f;
    const float d0 = frustum[2];
    const float dstep = D > 1 ? frustum[(size_t)H * W * 3 + 2] - d0 : 1.f;
    fscale[0] = xstep; fscale[1] = ystep; fscale[2] = d0; fscale[3] = dstep; fscale[4] = frustum[0]; fscale[5] = frustum[1];
    const bool fok = xstep > 0.f && ystep > 0.f && dstep > 0.f && isfinite(xstep) && isfinite(ystep) && isfinite(dstep) && isfinite(d0);
    hdr[H_BAD] = (bad || !fok) ? 1 : 0;
    hdr[H_NB] = 0;
    hdr[H_NC] = 0;
  }
}

struct GatherArgs {
  const float* frustum;
  const CamG* cams;
  const float *ipr, *ptr, *comb, *trn, *bda;
  const float* fscale;
  int32_t* hdr;
  int32_t *list_b, *list_c;
  const float* depth;
  const float4* feat;
  float4* out;
  int* out_rng;
  GridParams gp;
  int B, N, D, H, W, out_h2;
  int nbx, nby, nbz;       // 4 x 4 x 4 voxel blocks per axis
  int64_t n_blk;
  FeatIdx fi;
};

__device__ __forceinline__ Geo make_geo(const GatherArgs& a) {
  Geo g;
  g.cams = a.cams; g.ipr = a.ipr; g.ptr = a.ptr; g.comb = a.comb; g.trn = a.trn; g.bda = a.bda; g.gp = a.gp;
  g.fs.xstep = a.fscale[0]; g.fs.ystep = a.fscale[1]; g.fs.d0 = a.fscale[2]; g.fs.dstep = a.fscale[3]; g.fs.x0 = a.fscale[4]; g.fs.y0 = a.fscale[5];
  g.N = a.N; g.D = a.D; g.H = a.H; g.W = a.W;
  return g;
}

// A wave takes a 4 x 4 x 4 block of voxels (lane = voxel): its 64 boxes see the same cameras, reach the same depth planes and project
// to bounds of similar size, so the per-lane loops below run in step (with 64 voxels of one x-row -- 25 m at the C3 grid -- a quarter
// of the lanes worked; round-5 measurement: 75 us -> see profiles/r05_lss_kernels_pmc.md).
__global__ void __launch_bounds__(256) k_lssg_main(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tabs = smem;                                               // x[W], y[H], depth[D] of the separable frustum
  Geo g = make_geo(a);
  const int ntab = (g.W + g.H + g.D + 3) & ~3;
  const int nst = (stage_floats(a.B, a.B * a.N) + 3) & ~3;
  int32_t* slots = reinterpret_cast<int32_t*>(smem + ntab + nst);   // [wave][slot][lane]
  load_tables(tabs, a.frustum, g);
  stage_consts(smem + ntab, a.B, a.B * a.N, g);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPV, grp = lane / LPV, gbase = lane - sub;
  int32_t* myslots = slots + wave * (GL_SLOTS * 64);
  const bool bad = a.hdr[H_BAD] != 0;
  const int e_out = a.out_h2 ? rng_exp(a.out_rng) : 0;
  const float omul = rng_pow2(-e_out);
  unsigned amax = 0u;
  for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < a.n_blk; blk += (int64_t)gridDim.x * 4) {
  VoxBox q;
  int v;
  bool inr;
  {
    const unsigned ub = (unsigned)blk;
    const unsigned t1 = ub / (unsigned)a.nbx;
    const int bx = (int)(ub - t1 * (unsigned)a.nbx);
    const unsigned t2 = t1 / (unsigned)a.nby;
    const int by = (int)(t1 - t2 * (unsigned)a.nby);
    const unsigned t3 = t2 / (unsigned)a.nbz;
    const int bz = (int)(t2 - t3 * (unsigned)a.nbz);
    q.b = (int)t3;
    int idx[3] = {bx * 4 + (lane & 3), by * 4 + ((lane >> 2) & 3), bz * 4 + (lane >> 4)};
    inr = idx[0] < g.gp.gx && idx[1] < g.gp.gy && idx[2] < g.gp.gz;
    idx[0] = min(idx[0], g.gp.gx - 1); idx[1] = min(idx[1], g.gp.gy - 1); idx[2] = min(idx[2], g.gp.gz - 1);
    v = ((q.b * g.gp.gz + idx[2]) * g.gp.gy + idx[1]) * g.gp.gx + idx[0];
    const float lw[3] = {g.gp.lx, g.gp.ly, g.gp.lz}, iv[3] = {g.gp.ix, g.gp.iy, g.gp.iz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const float lo = lw[ax] + (float)(idx[ax] - (idx[ax] == 0 ? 1 : 0)) * iv[ax], hi = lw[ax] + (float)(idx[ax] + 1) * iv[ax];
      q.oc[ax] = 0.5f * (lo + hi);
      q.hf[ax] = 0.5f * (hi - lo) + G_EPS * fabsf(iv[ax]);
    }
  }
  // ---- candidates of this lane's voxel over the cameras of its batch element; enumerated right away while they stay few
  int est = 0, nh = 0;
  if (!bad) {
    for (int n = 0; n < g.N; ++n) {
      const int cam = q.b * g.N + n;
      const CamG c = g.cams[cam];
      CamBox cb = cam_box(c, q, g);
      if (!inr) cb.k1 = cb.k0 - 1;
      for (int dk = 0; __ballot(cb.k0 + dk <= cb.k1) != 0ull; ++dk) {
        const int k = cb.k0 + dk;
        const bool on = k <= cb.k1;
        const PlaneBox pb = plane_box(c, cb, on ? k : 0, g);
        const int cnt = on ? plane_count(pb) : 0;
        est += cnt;
        int w = pb.w0, h = pb.h0;
        bool live = cnt > 0 && est <= GL_TL;
        while (__ballot(live) != 0ull) {
          if (live) {
            if (lands_in(v, cam, q.b, k, h, w, tabs, g)) {
              if (nh < GL_SLOTS) myslots[nh * 64 + lane] = ((cam * g.D + k) * g.H + h) * g.W + w;
              ++nh;
            }
            if (++w > pb.w1) { w = pb.w0; if (++h > pb.h1) live = false; }
          }
        }
      }
    }
  }
  const bool heavy = inr && (est > GL_TL || nh > GL_SLOTS);
  {  // listed for k_lssg_heavy: one atomic per wave and list
    const bool toc = heavy && est > GL_TC, tob = heavy && !toc;
    const unsigned long long mb = __ballot(tob), mc = __ballot(toc);
    if (mb) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&a.hdr[H_NB], __builtin_popcountll(mb));
      base = __shfl(base, 0, 64);
      if (tob) a.list_b[base + __builtin_popcountll(mb & ((1ull << lane) - 1ull))] = v;
    }
    if (mc) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&a.hdr[H_NC], __builtin_popcountll(mc));
      base = __shfl(base, 0, 64);
      if (toc) a.list_c[base + __builtin_popcountll(mc & ((1ull << lane) - 1ull))] = v;
    }
  }
  // ---- rows: the empty voxels' zeros (NaN when a precondition failed), the light voxels' sums -- eight lanes per row, eight rows
  // per instruction (a block's rows are 16 runs of 4 consecutive voxels = 512 contiguous bytes each)
  const int c = !inr ? -1 : (heavy ? -2 : nh);
  {
    const unsigned long long empty = __ballot(c == 0);
    if (empty) {
      const float z = bad ? __uint_as_float(0x7fc00000u) : 0.f;
#pragma unroll
      for (int j = 0; j < GROUPS; ++j) {
        const int L = j * GROUPS + grp;
        const int64_t vv = __shfl(v, L, 64);
        if ((empty >> L) & 1ull) a.out[vv * LPV + sub] = make_float4(z, z, z, z);
      }
    }
    if (bad) amax = 0x7fc00000u;
  }
  const bool light = c > 0;
  const unsigned long long m = __ballot(light);
  const int nl = __builtin_popcountll(m);
  const int below = __builtin_popcountll(m & ((1ull << lane) - 1ull));
  const int dst = light ? below : nl + (lane - below);
  const int packed = __builtin_amdgcn_ds_permute(dst << 2, lane | ((c > 0 ? c : 0) << 8));
  __builtin_amdgcn_wave_barrier();
  for (int it = 0; it * GROUPS < nl; ++it) {
    const int qi = it * GROUPS + grp;
    const int pk = __shfl(packed, qi, 64);
    const bool on = qi < nl;
    const int cnt = on ? pk >> 8 : 0;
    const int L = pk & 63;
    const int64_t vv = __shfl(v, L, 64);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < GL_SLOTS; r0 += LPV) {
      if (__ballot(cnt > r0) == 0ull) break;
      const int cr = cnt - r0;                                      // rows of this round: min(cr, 8)
      int my_pf = 0;
      float my_d = 0.f;
      if (sub < cr) {
        const int id = myslots[(r0 + sub) * 64 + L];
        my_pf = feat_index(id, a.fi);
        my_d = a.depth[id];
      }
      float4 f[LPV];
#pragma unroll
      for (int u = 0; u < LPV; ++u) f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < LPV; ++u) {
        if (__ballot(u < cr) == 0ull) break;
        const int pf = __shfl(my_pf, gbase + u, 64);
        if (u < cr) f[u] = a.feat[(int64_t)pf * LPV + sub];
      }
#pragma unroll
      for (int u = 0; u < LPV; ++u) {
        if (__ballot(u < cr) == 0ull) break;
        const float d = __shfl(my_d, gbase + u, 64);
        if (u < cr) fma4_nc(acc, f[u], d);
      }
    }
    if (on) pool_store<LPV>(a.out, vv, sub, acc, a.out_h2, omul, amax);
  }
  __builtin_amdgcn_wave_barrier();                                  // the slots are reused by the next block of voxels
  }
  if (a.out_h2) rng_note(a.out_rng, amax, e_out);
}

namespace {
// (camera, depth plane) segments of a listed voxel: candidates [start, start + nw * nh) are the pixels of its plane box, row-major
struct SegTab { int* start; int4* info; };       // info = {cam * D + k, w0, nw, h0}

// candidate j of the table -> frustum entry; returns false past the end
__device__ __forceinline__ bool seg_candidate(const SegTab& t, int nseg, int total, int j, int& camk, int& h, int& w) {
  if (j >= total) return false;
  int sidx = 0;
  for (int s2 = 1; s2 < nseg; ++s2) sidx += (j >= t.start[s2]) ? 1 : 0;       // starts ascend: the last one not above j
  const int4 e = t.info[sidx];
  const int o = j - t.start[sidx];
  int r = (int)(((float)o + 0.5f) * __builtin_amdgcn_rcpf((float)e.z));
  int cw = o - r * e.z;
  if (cw < 0) { --r; cw += e.z; } else if (cw >= e.z) { ++r; cw -= e.z; }
  camk = e.x; h = e.w + r; w = e.y + cw;
  return true;
}
}  // namespace

// The listed voxels.  Blocks [0, GC_BLOCKS): one voxel of list_c per trip, 256 candidates a round; the other blocks: one voxel of
// list_b per wave and trip, 64 candidates a round.  A voxel's (camera, depth plane) boxes are laid end to end (SegTab), so a round is
// full whatever the size of a single box.  Hits are compacted IN ORDER (ballot rank, wave by wave) into an LDS list of point ids; when
// the list fills (and at the end) it is flushed: depths and rows are requested together, lane c adds channel c in list order -- the
// sum of a voxel is sequential in its points, that is what makes it bit-exact.  (No global load sits inside the candidate rounds: a
// first version fetched depth[id] per hit there and paid one memory latency per round, 77 us for the kernel.)
__global__ void __launch_bounds__(256) k_lssg_heavy(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tabs = smem;
  Geo g = make_geo(a);
  const int ntab = (g.W + g.H + g.D + 3) & ~3;
  const int nst = (stage_floats(a.B, a.B * a.N) + 3) & ~3;
  float* work = smem + ntab + nst;
  load_tables(tabs, a.frustum, g);
  stage_consts(smem + ntab, a.B, a.B * a.N, g);
  __syncthreads();
  if (a.hdr[H_BAD] != 0) return;                                   // (k_lssg_main wrote NaN rows everywhere; nothing is listed)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPV;
  const int e_out = a.out_h2 ? rng_exp(a.out_rng) : 0;
  const float omul = rng_pow2(-e_out);
  unsigned amax = 0u;
  const float* __restrict__ featf = reinterpret_cast<const float*>(a.feat);
  if ((int)blockIdx.x < GC_BLOCKS) {
    int32_t* lid = reinterpret_cast<int32_t*>(work);                // [GC_LIST] hit ids
    float* rowsf = work + GC_LIST;                                  // [HC_ROWS][32]
    float4* rows4 = reinterpret_cast<float4*>(rowsf);
    SegTab st;
    st.info = reinterpret_cast<int4*>(work + GC_LIST + HC_ROWS * 32);
    st.start = reinterpret_cast<int*>(st.info + G_SEGS);
    __shared__ int wcnt[4];
    const int nc = a.hdr[H_NC];
    const int rs = threadIdx.x >> 3, qd = threadIdx.x & 7;
    for (int li = blockIdx.x; li < nc; li += GC_BLOCKS) {
      const int v = a.list_c[li];
      const VoxBox q = vox_box(v, g.gp);
      int n = 0;
      float acc1 = 0.f;
      // add the n listed rows to acc1 (threads 0..31: channel = thread), HC_ROWS at a time through LDS
      auto flush = [&]() {
        __syncthreads();
        for (int base = 0; base < n; base += HC_ROWS) {
          float4 reg[4];
          float dc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = base + u * 32 + rs;
            const int id = j < n ? lid[j] : -1;
            dc[u] = id >= 0 ? a.depth[id] : 0.f;
            reg[u] = id >= 0 ? a.feat[(int64_t)feat_index(id, a.fi) * LPV + qd] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          __syncthreads();                                          // the previous chunk has been added
#pragma unroll
          for (int u = 0; u < 4; ++u)
            rows4[(u * 32 + rs) * LPV + qd] = make_float4(reg[u].x * dc[u], reg[u].y * dc[u], reg[u].z * dc[u], reg[u].w * dc[u]);
          __syncthreads();
          if (threadIdx.x < 32) {
            const int mrows = min(HC_ROWS, n - base);
            for (int j = 0; j < mrows; j += 16) {
              float x[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) x[u] = rowsf[(j + u) * 32 + threadIdx.x];
#pragma unroll
              for (int u = 0; u < 16; ++u)
                if (j + u < mrows) acc1 = acc1 + x[u];
            }
          }
        }
        __syncthreads();
        n = 0;
      };
      int nseg = 0, total = 0;
      auto run = [&]() {
        __syncthreads();                                            // the table is complete
        for (int base = 0; base < total; base += 256) {
          if (n + 256 > GC_LIST) flush();
          int camk, h, w;
          bool hit = false;
          int id = 0;
          if (seg_candidate(st, nseg, total, base + (int)threadIdx.x, camk, h, w)) {
            const int cam = camk / g.D, k = camk - cam * g.D;
            hit = lands_in(v, cam, q.b, k, h, w, tabs, g);
            id = (camk * g.H + h) * g.W + w;
          }
          const unsigned long long mb = __ballot(hit);
          if (lane == 0) wcnt[wave] = __builtin_popcountll(mb);
          __syncthreads();
          int pos = n + __builtin_popcountll(mb & ((1ull << lane) - 1ull));
          for (int w2 = 0; w2 < wave; ++w2) pos += wcnt[w2];
          if (hit) lid[pos] = id;
          n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
          __syncthreads();
        }
        nseg = 0; total = 0;
      };
      for (int cn = 0; cn < g.N; ++cn) {
        const int cam = q.b * g.N + cn;
        const CamG c = g.cams[cam];
        const CamBox cb = cam_box(c, q, g);
        for (int k = cb.k0; k <= cb.k1; ++k) {
          const PlaneBox pb = plane_box(c, cb, k, g);
          const int cnt = plane_count(pb);
          if (cnt == 0) continue;
          if (nseg == G_SEGS) run();
          if (threadIdx.x == 0) {
            st.start[nseg] = total;
            st.info[nseg] = make_int4(cam * g.D + k, pb.w0, pb.w1 - pb.w0 + 1, pb.h0);
          }
          ++nseg; total += cnt;
        }
      }
      run();
      flush();
      if (wave == 0) {
        const float4 acc = half_to_quads(acc1, lane);
        if (lane < LPV) pool_store<LPV>(a.out, v, sub, acc, a.out_h2, omul, amax);
      }
    }
  } else {
    // per wave: hit ids [GB_LIST], (feat pixel, depth) of a batch of 64 [64 + 64], segment table [G_SEGS x (4 + 1)]
    constexpr int PER_WAVE = GB_LIST + 128 + 5 * G_SEGS;
    int32_t* lid = reinterpret_cast<int32_t*>(work) + wave * PER_WAVE;
    int32_t* lpf = lid + GB_LIST;
    float* ld = reinterpret_cast<float*>(lpf + 64);
    SegTab st;
    st.info = reinterpret_cast<int4*>(lpf + 128);
    st.start = reinterpret_cast<int*>(st.info + G_SEGS);
    const int nb = a.hdr[H_NB];
    const int cch = lane & 31;
    const int nwaves = ((int)gridDim.x - GC_BLOCKS) * 4;
    for (int li = ((int)blockIdx.x - GC_BLOCKS) * 4 + wave; li < nb; li += nwaves) {
      const int v = a.list_b[li];
      const VoxBox q = vox_box(v, g.gp);
      int n = 0;
      float acc1 = 0.f;
      auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int b0 = 0; b0 < n; b0 += 64) {
          const int mb = min(64, n - b0);
          float dv = 0.f;
          if (lane < mb) {
            const int id = lid[b0 + lane];
            dv = a.depth[id];
            lpf[lane] = feat_index(id, a.fi);
          }
          __builtin_amdgcn_wave_barrier();
          float x0[32];
#pragma unroll
          for (int p = 0; p < 32; ++p) {
            x0[p] = 0.f;
            if (p < mb) x0[p] = featf[(unsigned)(lpf[p] * 32 + cch)];
          }
          if (lane < mb) ld[lane] = dv;
          __builtin_amdgcn_wave_barrier();
          if (mb > 32) {
            float x1[32];
#pragma unroll
            for (int p = 0; p < 32; ++p) {
              x1[p] = 0.f;
              if (32 + p < mb) x1[p] = featf[(unsigned)(lpf[32 + p] * 32 + cch)];
            }
#pragma unroll
            for (int p = 0; p < 32; ++p) acc1 = acc1 + x0[p] * ld[p];
#pragma unroll
            for (int p = 0; p < 32; ++p)
              if (32 + p < mb) acc1 = acc1 + x1[p] * ld[32 + p];
          } else {
#pragma unroll
            for (int p = 0; p < 32; ++p)
              if (p < mb) acc1 = acc1 + x0[p] * ld[p];
          }
          __builtin_amdgcn_wave_barrier();
        }
        n = 0;
      };
      int nseg = 0, total = 0;
      auto run = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < total; base += 64) {
          if (n + 64 > GB_LIST) flush();
          int camk, h, w;
          bool hit = false;
          int id = 0;
          if (seg_candidate(st, nseg, total, base + lane, camk, h, w)) {
            const int cam = camk / g.D, k = camk - cam * g.D;
            hit = lands_in(v, cam, q.b, k, h, w, tabs, g);
            id = (camk * g.H + h) * g.W + w;
          }
          const unsigned long long mb = __ballot(hit);
          if (hit) lid[n + __builtin_popcountll(mb & ((1ull << lane) - 1ull))] = id;
          n += __builtin_popcountll(mb);
        }
        __builtin_amdgcn_wave_barrier();
        nseg = 0; total = 0;
      };
      for (int cn = 0; cn < g.N; ++cn) {
        const int cam = q.b * g.N + cn;
        const CamG c = g.cams[cam];
        const CamBox cb = cam_box(c, q, g);
        for (int k = cb.k0; k <= cb.k1; ++k) {
          const PlaneBox pb = plane_box(c, cb, k, g);
          const int cnt = plane_count(pb);
          if (cnt == 0) continue;
          if (nseg == G_SEGS) run();
          if (lane == 0) {
            st.start[nseg] = total;
            st.info[nseg] = make_int4(cam * g.D + k, pb.w0, pb.w1 - pb.w0 + 1, pb.h0);
          }
          ++nseg; total += cnt;
        }
      }
      run();
      flush();
      const float4 acc = half_to_quads(acc1, lane);
      if (lane < LPV) pool_store<LPV>(a.out, v, sub, acc, a.out_h2, omul, amax);
    }
  }
  if (a.out_h2) rng_note(a.out_rng, amax, e_out);
}

namespace {
struct GatherWs {
  int32_t* hdr;
  float* fscale;
  float *ipr, *comb, *tr;
  CamG* cams;
  int32_t *list_b, *list_c;
  size_t bytes;
};

GatherWs gather_ws(char* base, int64_t n_vox, int BN) {
  GatherWs w;
  char* p = base;
  auto take = [&](size_t bytes) { char* q = p; p += pw_align_up(bytes, 256); return q; };
  w.hdr = (int32_t*)take(H_WORDS * 4);
  w.fscale = (float*)take(8 * 4);
  w.ipr = (float*)take((size_t)BN * 9 * 4);
  w.comb = (float*)take((size_t)BN * 9 * 4);
  w.tr = (float*)take((size_t)BN * 3 * 4);
  w.cams = (CamG*)take((size_t)BN * sizeof(CamG));
  w.list_b = (int32_t*)take((size_t)n_vox * 4);
  w.list_c = (int32_t*)take((size_t)n_vox * 4);
  w.bytes = (size_t)(p - base);
  return w;
}
}  // namespace

PW_API size_t pw_lss_lift_gather_workspace_bytes(int64_t n_voxels, int BN) { return gather_ws(nullptr, n_voxels, BN).bytes; }

PW_API int pw_lss_lift_gather(int B, int N, int D, int H, int W, const float* frustum, const float* sensor2ego,
                              const float* cam2imgs, const float* post_rots, const float* post_trans, const float* bda,
                              const float* lower3_host, const float* interval3_host, int gx, int gy, int gz, const float* depth,
                              const float* feat, int c, void* workspace, size_t workspace_bytes, float* out, int out_h2,
                              int32_t* out_rng, void* stream) {
  PW_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && W > 0 && gx > 0 && gy > 0 && gz > 0, "pw_lss_lift_gather: bad shape");
  PW_CHECK_ARG(frustum && sensor2ego && cam2imgs && post_rots && post_trans && bda && lower3_host && interval3_host && depth &&
                   feat && workspace && out,
               "pw_lss_lift_gather: null pointer");
  PW_CHECK_ARG(c == 4 * LPV, "pw_lss_lift_gather: C must be 32 (use pw_segment_sort + pw_bev_pool_dense for other widths)");
  PW_CHECK_ARG(B * N <= G_STAGE_MAX_BN, "pw_lss_lift_gather: at most %d cameras in the batch (B * N)", G_STAGE_MAX_BN);
  PW_CHECK_ARG(((uintptr_t)feat & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               "pw_lss_lift_gather: feat / out must be 16-byte aligned, the workspace 256-byte aligned");
  PW_CHECK_ARG(interval3_host[0] > 0.f && interval3_host[1] > 0.f && interval3_host[2] > 0.f, "pw_lss_lift_gather: grid intervals must be positive");
  const int64_t n_vox = (int64_t)B * gx * gy * gz;
  const int64_t DHW = (int64_t)D * H * W, total = DHW * B * N;
  PW_CHECK_ARG(n_vox < (int64_t)1 << 31 && total < (int64_t)1 << 31, "pw_lss_lift_gather: sizes must fit int32");
  PW_CHECK_ARG(DHW < (1 << 24) && (int64_t)B * N < (1 << 20), "pw_lss_lift_gather: D*H*W must be below 2^24, B*N below 2^20");
  PW_CHECK_ARG(W + H + D <= 4096, "pw_lss_lift_gather: W + H + D must not exceed 4096");
  const GatherWs w = gather_ws((char*)workspace, n_vox, B * N);
  if (workspace_bytes < w.bytes) {
    pw_set_error("pw_lss_lift_gather: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    return PW_ENOSPC;
  }
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_lssg_prologue, dim3(1), dim3(256), 0, st, B, N, D, H, W, frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda,
                     w.ipr, w.comb, w.tr, w.cams, w.fscale, w.hdr);
  GatherArgs a;
  a.frustum = frustum; a.cams = w.cams; a.ipr = w.ipr; a.ptr = post_trans; a.comb = w.comb; a.trn = w.tr; a.bda = bda;
  a.fscale = w.fscale; a.hdr = w.hdr; a.list_b = w.list_b; a.list_c = w.list_c; a.depth = depth; a.feat = (const float4*)feat;
  a.out = (float4*)out; a.out_rng = out_rng;
  a.gp = GridParams{lower3_host[0], lower3_host[1], lower3_host[2], interval3_host[0], interval3_host[1], interval3_host[2], gx, gy, gz};
  a.B = B; a.N = N; a.D = D; a.H = H; a.W = W; a.out_h2 = out_h2;
  a.nbx = (gx + 3) / 4; a.nby = (gy + 3) / 4; a.nbz = (gz + 3) / 4;
  a.n_blk = (int64_t)B * a.nbx * a.nby * a.nbz;
  a.fi = FeatIdx{(int)DHW, H * W, 1.0f / (float)DHW, 1.0f / (float)(H * W)};
  const int ntab = ((W + H + D + 3) & ~3) + ((stage_floats(B, B * N) + 3) & ~3);
  const size_t lds_main = (size_t)ntab * 4 + 4 * GL_SLOTS * 64 * 4;
  {
    const int64_t want = pw_cdiv(a.n_blk, 4), cap = 4 * 256;
    hipLaunchKernelGGL(k_lssg_main, dim3((unsigned)(want < cap ? want : cap)), dim3(256), lds_main, st, a);
  }
  const size_t lds_c = (size_t)(GC_LIST + HC_ROWS * 32 + 5 * G_SEGS) * 4, lds_b = (size_t)4 * (GB_LIST + 128 + 5 * G_SEGS) * 4;
  const size_t lds_heavy = (size_t)ntab * 4 + (lds_c > lds_b ? lds_c : lds_b);
  hipLaunchKernelGGL(k_lssg_heavy, dim3(GC_BLOCKS + GB_BLOCKS), dim3(256), lds_heavy, st, a);
  pw_note_kernel("k_lssg_main");
  PW_CHECK_LAUNCH();
  return PW_OK;
}
