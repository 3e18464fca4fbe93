// 3x3x3 stride-1 convolution by Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores.
//
// Why: the direct kernels (pw_conv3d.hip) sit on the ceiling of v_mfma_f32_32x32x2_f32 (81-90 % of
// 157 TFLOP/s); the only way past it in exact-fp32 arithmetic is fewer multiplies.
// F(2,3) per axis turns the 27 multiply-accumulates of an output voxel into 64 / 8 = 8 (3.375x fewer):
//   U = G g G^T  (weights, once on the host)      V = B^T d B  (4x4x4 input tile -> 64 points)
//   M[xi] = sum_c V[xi][c] U[xi][c][k]            Y = A^T M A  (64 points -> 2x2x2 outputs)
// The per-point channel contraction is a batch of 64 GEMMs (M = Winograd tiles, N = couts, K = cins)
// -> MFMA; the two transforms are adds/subs only (B and A have entries in {0, +-1}) -> VALU.
// (cuDNN picks the same algorithm family for fp32 3x3 convolutions; rounding differs from the direct sum
// at the 1e-6 level, far inside the stated tolerance, see tests.)
//
// One 256-thread block computes a 4x8x8 output tile (= 2x4x4 = 32 Winograd tiles) x 32 output channels:
//   * the 6x10x10x32ch halo is written to LDS by buffer_load ... lds (R, 76.8 KB, unswizzled);
//   * for each d-transform row i_d (4 chunks of 16 points): thread (tile, channel quad) combines its two
//     d-planes, transforms along h and w in registers (48 float4 add/sub) and writes its 16 points to V
//     (64 KB: [point][tile][32 ch], row order and 16-byte slots chosen so the MFMA A reads are
//     bank-conflict free);
//   * wave w owns 16 tiles (one d-pair) x 16 couts: per point 2 ds_read_b128 (A) + 2 buffer loads (U) +
//     8 v_mfma_f32_16x16x4_f32 into a fresh accumulator, which is then added/subtracted into the (at most
//     8) outputs it contributes to -- 8 x 4 accumulator registers per lane hold the whole output tile;
//   * epilogue = the direct kernels' (scale/bias, residual, ReLU, two destinations, row stride).
#include "pw_conv3d_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int WINO_R_BYTES = TV * KC * 4;            // 76800: raw halo
constexpr int WINO_V_BYTES = 16 * 32 * KC * 4;       // 65536: 16 points x 32 tiles x 32 ch
constexpr int WINO_LDS = WINO_R_BYTES + WINO_V_BYTES;

// LDS row of tile t16 (0..15) inside a 16-tile half: bit0 = t2 ^ t3 so that the two ds_read_b128 lane
// groups {0-3,12-15,20-27} / {4-11,16-19,28-31} each touch 16 distinct 16-byte slots of a 256-byte bank row
__device__ __forceinline__ int wino_row16(int t) {
  return (t & 8) | ((t & 2) << 1) | ((t & 1) << 1) | (((t >> 2) ^ (t >> 3)) & 1);
}

// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] applied to (x0..x3)
__device__ __forceinline__ float4 f4sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ void bt4(const float4& x0, const float4& x1, const float4& x2, const float4& x3,
                                    float4& y0, float4& y1, float4& y2, float4& y3) {
  y0 = f4sub(x0, x2); y1 = f4add(x1, x2); y2 = f4sub(x2, x1); y3 = f4sub(x1, x3);
}
}  // namespace

// A^T = [[1,1,1,0],[0,1,-1,-1]]: sign of point index i (0..3) in output o (0..1), 0 = no contribution
__device__ __forceinline__ constexpr int at_sign(int o, int i) {
  return o == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : (i == 1 ? 1 : -1));
}

template <int ID, int XI>     // chunk ID = i_d, XI = i_h*4 + i_w: add the fresh product M into the outputs it feeds
__device__ __forceinline__ void wino_scatter(const f32x4& M, f32x4 (&Y)[8]) {
  constexpr int ih = XI >> 2, iw = XI & 3;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int sd = at_sign(o >> 2, ID), sh = at_sign((o >> 1) & 1, ih), sw = at_sign(o & 1, iw);
    const int sg = sd * sh * sw;
    if (sg > 0) Y[o] += M;
    else if (sg < 0) Y[o] -= M;
  }
}

struct WinoCtx {
  lds3_t lds3;
  rsrc_t wr;
  unsigned a_addr[2];        // LDS byte address of this lane's A fragment in point 0 of V, q = 0, 1
  unsigned lane_off;         // lane * 32
  unsigned ustep;            // bytes between the weights of consecutive points
};

// Points are processed in PAIRS (two independent MFMA chains interleaved: a single 16x16x4 chain is
// latency-bound), operands of the next pair are requested before the MFMAs, and the add/sub of a pair's
// products into the outputs is deferred until the next pair's MFMAs have been issued.
template <int XI>
__device__ __forceinline__ void wino_load_pair(const WinoCtx& c, unsigned usoff, float4 (&aq)[2][2], float4 (&bq)[2][2]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const v4f v = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(
          c.lds3 + c.a_addr[q] + (XI + k) * 4096);
      aq[k][q] = make_float4(v[0], v[1], v[2], v[3]);
      bq[k][q] = buf_load4(c.wr, c.lane_off + (unsigned)(q * 16), usoff + (unsigned)(XI + k) * c.ustep);
    }
}

template <int ID, int XI>     // XI even: points XI, XI+1
__device__ __forceinline__ void wino_pair(const WinoCtx& c, unsigned usoff, float4 (&ac)[2][2], float4 (&bc)[2][2],
                                          float4 (&an)[2][2], float4 (&bn)[2][2], f32x4 (&Mp)[2], f32x4 (&Y)[8]) {
  if constexpr (XI + 2 < 16) wino_load_pair<XI + 2>(c, usoff, an, bn);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 M0 = {0.f, 0.f, 0.f, 0.f}, M1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float a0[4] = {ac[0][q].x, ac[0][q].y, ac[0][q].z, ac[0][q].w};
    const float b0[4] = {bc[0][q].x, bc[0][q].y, bc[0][q].z, bc[0][q].w};
    const float a1[4] = {ac[1][q].x, ac[1][q].y, ac[1][q].z, ac[1][q].w};
    const float b1[4] = {bc[1][q].x, bc[1][q].y, bc[1][q].z, bc[1][q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      M0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], M0, 0, 0, 0);
      M1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], M1, 0, 0, 0);
    }
  }
  if constexpr (XI >= 2) {                 // the previous pair's products, under this pair's MFMAs
    wino_scatter<ID, XI - 2>(Mp[0], Y);
    wino_scatter<ID, XI - 1>(Mp[1], Y);
  }
  __builtin_amdgcn_sched_barrier(0);
  Mp[0] = M0; Mp[1] = M1;
  if constexpr (XI + 2 < 16) wino_pair<ID, XI + 2>(c, usoff, an, bn, ac, bc, Mp, Y);
}

// d-transform row ID of this thread's tile: 16 (h, w) positions x one channel quad, then h and w transforms, -> V
template <int ID>
__device__ __forceinline__ void wino_transform_chunk(const char* ldsb, char* vb, unsigned r_base, unsigned v_base) {
  // B^T row ID: (plane A, plane B, sign of B): 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
  constexpr int pa = ID == 0 ? 0 : (ID == 2 ? 2 : 1);
  constexpr int pb = ID == 0 ? 2 : (ID == 1 ? 2 : (ID == 2 ? 1 : 3));
  constexpr bool plus = ID == 1;
  // one w-column at a time (8 reads -> 4 combined values -> h-transform): keeps the live set at the 16
  // h-transformed values + one column instead of two full 4x4 arrays
  float4 y[4][4];
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    float4 x[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      const float4 A = *reinterpret_cast<const float4*>(ldsb + r_base + (unsigned)(((pa * TH + hh) * TW + ww) * 128));
      const float4 B = *reinterpret_cast<const float4*>(ldsb + r_base + (unsigned)(((pb * TH + hh) * TW + ww) * 128));
      x[hh] = plus ? f4add(A, B) : f4sub(A, B);
    }
    bt4(x[0], x[1], x[2], x[3], y[0][ww], y[1][ww], y[2][ww], y[3][ww]);
  }
#pragma unroll
  for (int ih = 0; ih < 4; ++ih) {
    float4 z0, z1, z2, z3;
    bt4(y[ih][0], y[ih][1], y[ih][2], y[ih][3], z0, z1, z2, z3);
    *reinterpret_cast<float4*>(vb + v_base + (unsigned)((ih * 4 + 0) * 4096)) = z0;
    *reinterpret_cast<float4*>(vb + v_base + (unsigned)((ih * 4 + 1) * 4096)) = z1;
    *reinterpret_cast<float4*>(vb + v_base + (unsigned)((ih * 4 + 2) * 4096)) = z2;
    *reinterpret_cast<float4*>(vb + v_base + (unsigned)((ih * 4 + 3) * 4096)) = z3;
  }
}

template <int ID, int NG>
__device__ __forceinline__ void wino_chunk(const WinoCtx& c, const char* ldsb, char* vb, unsigned r_base, unsigned v_base,
                                           unsigned usoff, f32x4 (&Y)[NG][8]) {
  wino_transform_chunk<ID>(ldsb, vb, r_base, v_base);
  __syncthreads();                                   // V of this chunk complete
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {                  // the transformed tile serves every 32-cout group
    float4 a0[2][2], a1[2][2], b0[2][2], b1[2][2];
    f32x4 Mp[2];
    const unsigned us = usoff + (unsigned)ng * 4096u;            // 2 n16 blocks of 2048 B per 32-cout group
    wino_load_pair<0>(c, us, a0, b0);
    wino_pair<ID, 0>(c, us, a0, b0, a1, b1, Mp, Y[ng]);
    wino_scatter<ID, 14>(Mp[0], Y[ng]);
    wino_scatter<ID, 15>(Mp[1], Y[ng]);
  }
  __syncthreads();                                   // everyone done reading V before the next chunk overwrites it
}

// unswizzled halo offsets of a lane for a tile column position (the transform reads whole 128-byte rows)
__device__ __forceinline__ void wino_lane_offsets(const ConvArgs& a, int w0, int lane, PipeDma& dm) {
  const int shift = w0 == 0 ? 1 : 0;
  {
    const int ww = lane >> 3, slot = lane & 7;
    dm.voff[0][0] = dm.voff[1][0] = (unsigned)(w0 - 1 + ww) < (unsigned)a.W ? (unsigned)((ww - shift) * a.Cin + slot * 4) * 4u : PIPE_OOB;
  }
  {
    const int ww = 8 + (lane >> 5), dw = lane & 31;
    dm.voff[0][1] = dm.voff[1][1] = (unsigned)(w0 - 1 + ww) < (unsigned)a.W ? (unsigned)((ww - shift) * a.Cin + dw) * 4u : PIPE_OOB;
  }
}

// One block per output tile, all 32-cout groups of the layer inside the block (the transformed tile is
// reused by every group).  A persistent variant with the next tile's halo DMA issued early was tried
// (dedicated DMA wave: starved by the MFMA waves; early issue from the compute waves: needs the chunk's 16
// weight taps preloaded past the in-order vmcnt, 128 more live registers -> spills): 195 us for 32->32 but
// far slower for two cout groups, so the halo load stays exposed (~15 % of a tile).
template <int NG>    // 32-cout groups (cout_total = 32 NG)
__global__ void __launch_bounds__(256, 1) k_conv3d_wino(ConvArgs a, int n16_total) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int mh = wave & 1, nh = wave >> 1;            // tile half (d-pair) and cout half of this wave
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int tw_ = bid % a.tiles_w; bid /= a.tiles_w;
  const int th_ = bid % a.tiles_h; bid /= a.tiles_h;
  const int td_ = bid % a.tiles_d;
  const int b = bid / a.tiles_d;
  const int d0 = td_ * BD, h0 = th_ * BH, w0 = tw_ * BW;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  char* vb = reinterpret_cast<char*>(lds) + WINO_R_BYTES;

  // transform role: thread = (tile 0..31, channel quad 0..7)
  const int tile = tid >> 3, quad = tid & 7;
  const int ttd = tile >> 4, tth = (tile >> 2) & 3, ttw = tile & 3;
  const unsigned r_base = (unsigned)((((2 * ttd) * TH + 2 * tth) * TW + 2 * ttw) * 128 + quad * 16);
  const int t16 = tile & 15;
  const unsigned v_base = (unsigned)(((ttd * 16 + wino_row16(t16)) * 8 + (quad ^ (t16 & 7))) * 16);

  // GEMM role: lane = (tile l&15 of the wave's half, k-group l>>4)
  WinoCtx c;
  c.lds3 = (lds3_t)lds;
  {
    const int lt = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      c.a_addr[q] = (unsigned)WINO_R_BYTES + (unsigned)(((mh * 16 + wino_row16(lt)) * 8 + ((g * 2 + q) ^ (lt & 7))) * 16);
  }
  const int nchunk = a.Cin / KC;
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 64 * n16_total * 2048));
  c.lane_off = (unsigned)lane * 32u;
  c.ustep = (unsigned)n16_total * 2048u;               // bytes between the weights of consecutive points
  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  f32x4 Y[NG][8];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng)
#pragma unroll
    for (int o = 0; o < 8; ++o) Y[ng][o] = f32x4{0.f, 0.f, 0.f, 0.f};

  PipeDma dm;
  wino_lane_offsets(a, w0, lane, dm);
  dm.b = b; dm.d0 = d0; dm.h0 = h0; dm.wbase = w0 > 0 ? w0 - 1 : 0; dm.ldsbuf = 0; dm.live = true;
  for (int ch = 0; ch < nchunk; ++ch) {
    const unsigned ubase = (unsigned)(((ch * 64) * n16_total + nh) * 2048);
    const unsigned ustep = c.ustep;
    if (ch > 0) __syncthreads();
    dm.ch = ch;
    pipe_dma_row<0>(a, xr, c.lds3, dm, wave); pipe_dma_row<1>(a, xr, c.lds3, dm, wave); pipe_dma_row<2>(a, xr, c.lds3, dm, wave);
    pipe_dma_row<3>(a, xr, c.lds3, dm, wave); pipe_dma_row<4>(a, xr, c.lds3, dm, wave); pipe_dma_row<5>(a, xr, c.lds3, dm, wave);
    pipe_dma_row<6>(a, xr, c.lds3, dm, wave); pipe_dma_row<7>(a, xr, c.lds3, dm, wave); pipe_dma_row<8>(a, xr, c.lds3, dm, wave);
    pipe_dma_row<9>(a, xr, c.lds3, dm, wave); pipe_dma_row<10>(a, xr, c.lds3, dm, wave); pipe_dma_row<11>(a, xr, c.lds3, dm, wave);
    pipe_dma_row<12>(a, xr, c.lds3, dm, wave); pipe_dma_row<13>(a, xr, c.lds3, dm, wave); pipe_dma_row<14>(a, xr, c.lds3, dm, wave);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // point index = i_d*16 + i_h*4 + i_w; weights of point p at ubase + p * n16_total * 2048
    wino_chunk<0, NG>(c, ldsb, vb, r_base, v_base, ubase + 0u * 16u * ustep, Y);
    wino_chunk<1, NG>(c, ldsb, vb, r_base, v_base, ubase + 16u * ustep, Y);
    wino_chunk<2, NG>(c, ldsb, vb, r_base, v_base, ubase + 32u * ustep, Y);
    wino_chunk<3, NG>(c, ldsb, vb, r_base, v_base, ubase + 48u * ustep, Y);
  }
  // ---- epilogue: lane holds cout l&15 for tiles (l>>4)*4 + r of its half; output o = (od, oh, ow)
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int n = ng * 32 + nh * 16 + (lane & 15);
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float bi = a.bias ? a.bias[n] : 0.f;
    const int tth_e = lane >> 4;
    const int n0 = ng * 32 + nh * 16;                   // first packed column of this wave (uniform)
    const bool to_y0 = n0 < a.cout0;
    float* dst = to_y0 ? a.y0 : a.y1;
    const int ncols = to_y0 ? a.cout0 : a.cout1, ld = to_y0 ? a.ld0 : a.ld1;
    const int col0 = to_y0 ? n0 : n0 - a.n1_start;
    const bool interior = d0 + BD <= a.Do && h0 + BH <= a.Ho && w0 + BW <= a.Wo && dst != nullptr && col0 >= 0 &&
                          col0 + 16 <= ncols;
    if (interior) {
      const bool has_res = to_y0 && a.residual != nullptr;
      const float lo = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
      const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
      const rsrc_t yr = make_rsrc(dst, out_vox * (unsigned)ld * 4u);
      const rsrc_t rr = make_rsrc(has_res ? a.residual : dst, out_vox * (unsigned)ld * 4u);
      // lane offset: its h-pair row and column; (od, oh, ow, r) steps are scalar
      const unsigned lane_o = (unsigned)((2 * tth_e * a.Wo) * ld + col0 + (lane & 15)) * 4u;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int od = d0 + 2 * mh + (o >> 2), oh = (o >> 1) & 1, ow = o & 1;
        const unsigned so = (unsigned)((((b * a.Do + od) * a.Ho + h0 + oh) * a.Wo + w0 + ow) * ld) * 4u;
        float rv[4];
        if (has_res) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rv[r] = buf_load1(rr, lane_o, so + (unsigned)(2 * r * ld) * 4u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = Y[ng][o][r] * sc + bi;
          if (has_res) v += rv[r];
          buf_store1(yr, lane_o, so + (unsigned)(2 * r * ld) * 4u, fmaxf(v, lo));
        }
      }
    } else {
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int od = d0 + 2 * mh + (o >> 2), oh = h0 + 2 * tth_e + ((o >> 1) & 1), ow = w0 + 2 * r + (o & 1);
          if (od < a.Do && oh < a.Ho && ow < a.Wo) {
            const size_t vox = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
            store_out(a, n, vox, Y[ng][o][r] * sc + bi);
          }
        }
    }
  }   // ng
}

PW_API int pw_conv3d_wino(const float* x, const float* uwpk, const float* scale, const float* bias,
                          const float* residual, float* y0, float* y1, int B, int D, int H, int W, int Cin,
                          int cout_total, int cout0, int cout1, int ld_y0, int ld_y1, int relu0, int relu1,
                          void* stream) {
  PW_CHECK_ARG(x && uwpk && y0, "pw_conv3d_wino: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cin % KC == 0, "pw_conv3d_wino: bad shape");
  PW_CHECK_ARG(cout_total > 0 && cout_total % 32 == 0 && cout0 > 0 && cout0 <= cout_total && cout1 >= 0,
               "pw_conv3d_wino: bad cout split");
  PW_CHECK_ARG(!(cout1 > 0 && !y1), "pw_conv3d_wino: cout1 > 0 needs y1");
  ConvArgs a = {};
  a.x = x; a.wpk = uwpk; a.scale = scale; a.bias = bias; a.residual = residual; a.y0 = y0; a.y1 = y1;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.ld0 = ld_y0 > 0 ? ld_y0 : cout0; a.ld1 = ld_y1 > 0 ? ld_y1 : cout1;
  a.n1_start = (cout0 + 31) / 32 * 32;
  a.relu0 = relu0; a.relu1 = relu1;
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  PW_CHECK_ARG((size_t)B * D * H * W * Cin * 4 < (1ull << 32) &&
                   (size_t)B * D * H * W * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) * 4 < (1ull << 32),
               "pw_conv3d_wino: tensors must be < 4 GiB (32-bit buffer addressing)");
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  PW_CHECK_ARG(nblk < (1ll << 20), "pw_conv3d_wino: too many tiles");
  const int NG = cout_total / 32;
  PW_CHECK_ARG(NG == 1 || NG == 2, "pw_conv3d_wino: cout_total must be 32 or 64 (got %d)", cout_total);
#define PW_WINO(NGv)                                                                                      \
  do {                                                                                                    \
    static int once = set_lds_limit(k_conv3d_wino<NGv>, WINO_LDS);                                         \
    if (once) return once;                                                                                \
    hipLaunchKernelGGL(k_conv3d_wino<NGv>, dim3((unsigned)nblk), dim3(256), WINO_LDS, pw_stream(stream), a, \
                       cout_total / 16);                                                                  \
  } while (0)
  if (NG == 1) PW_WINO(1); else PW_WINO(2);
#undef PW_WINO
  PW_CHECK_LAUNCH();
  return PW_OK;
}
