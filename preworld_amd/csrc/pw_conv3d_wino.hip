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
#include "pw_wino_common.h"

// Points are processed a ROW (4 points = 4 independent MFMA chains, interleaved) at a time; the operands
// of the next row are requested before this row's MFMAs, and the output transform of a row's products is
// deferred until the next row's MFMAs have been issued (it runs in their shadow).
template <int IH>
__device__ __forceinline__ void wino_load_row(const WinoCtx& c, unsigned usoff, f32x4 (&aq)[4][2], f32x4 (&bq)[4][2]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      aq[k][q] = lds_read4(c.lds3, c.a_addr[q] + (unsigned)(IH * 4 + k) * 4096u);
      const float4 w = buf_load4(c.wr, c.lane_off + (unsigned)(q * 16), usoff + (unsigned)(IH * 4 + k) * c.ustep);
      bq[k][q] = f32x4{w.x, w.y, w.z, w.w};
    }
}

template <int ID, int IH>
__device__ __forceinline__ void wino_row(const WinoCtx& c, unsigned usoff, f32x4 (&ac)[4][2], f32x4 (&bc)[4][2],
                                         f32x4 (&an)[4][2], f32x4 (&bn)[4][2], f32x4 (&Mp)[4], f32x4 (&Y)[8]) {
  if constexpr (IH + 1 < 4) wino_load_row<IH + 1>(c, usoff, an, bn);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 M[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) M[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        M[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k][q][e], bc[k][q][e], M[k], 0, 0, 0);
  if constexpr (IH >= 1) wino_scatter_row<ID, IH - 1>(Mp, Y);      // the previous row's products, under this row's MFMAs
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) Mp[k] = M[k];
  if constexpr (IH + 1 < 4) wino_row<ID, IH + 1>(c, usoff, an, bn, ac, bc, Mp, Y);
}

// d-transform row ID of this thread's tile: 16 (h, w) positions x one channel quad, then h and w transforms, -> V
template <int ID>
__device__ __forceinline__ void wino_transform_chunk(lds3_t lds3, unsigned r_base, unsigned v_base) {
  // B^T row ID: (plane A, plane B, sign of B): 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
  constexpr int pa = ID == 0 ? 0 : (ID == 2 ? 2 : 1);
  constexpr int pb = ID == 0 ? 2 : (ID == 1 ? 2 : (ID == 2 ? 1 : 3));
  constexpr bool plus = ID == 1;
  // one w-column at a time (8 reads -> 4 combined values -> h-transform): keeps the live set at the 16
  // h-transformed values + one column instead of two full 4x4 arrays
  f32x4 y[4][4];
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    f32x4 x[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      const f32x4 A = lds_read4(lds3, r_base + (unsigned)(((pa * TH + hh) * TW + ww) * 128));
      const f32x4 B = lds_read4(lds3, r_base + (unsigned)(((pb * TH + hh) * TW + ww) * 128));
      x[hh] = plus ? A + B : sub4(A, B);
    }
    bt4(x[0], x[1], x[2], x[3], y[0][ww], y[1][ww], y[2][ww], y[3][ww]);
  }
#pragma unroll
  for (int ih = 0; ih < 4; ++ih) {
    f32x4 z0, z1, z2, z3;
    bt4(y[ih][0], y[ih][1], y[ih][2], y[ih][3], z0, z1, z2, z3);
    lds_write4(lds3, (unsigned)WINO_R_BYTES + v_base + (unsigned)((ih * 4 + 0) * 4096), z0);
    lds_write4(lds3, (unsigned)WINO_R_BYTES + v_base + (unsigned)((ih * 4 + 1) * 4096), z1);
    lds_write4(lds3, (unsigned)WINO_R_BYTES + v_base + (unsigned)((ih * 4 + 2) * 4096), z2);
    lds_write4(lds3, (unsigned)WINO_R_BYTES + v_base + (unsigned)((ih * 4 + 3) * 4096), z3);
  }
}

template <int ID, int NG>
__device__ __forceinline__ void wino_chunk(const WinoCtx& c, unsigned r_base, unsigned v_base, unsigned usoff,
                                           f32x4 (&Y)[NG][8]) {
  wino_transform_chunk<ID>(c.lds3, r_base, v_base);
  __syncthreads();                                   // V of this chunk complete
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {                  // the transformed tile serves every 32-cout group
    f32x4 a0[4][2], a1[4][2], b0[4][2], b1[4][2];
    f32x4 Mp[4];
    const unsigned us = usoff + (unsigned)ng * 4096u;            // 2 n16 blocks of 2048 B per 32-cout group
    wino_load_row<0>(c, us, a0, b0);
    wino_row<ID, 0>(c, us, a0, b0, a1, b1, Mp, Y[ng]);
    wino_scatter_row<ID, 3>(Mp, Y[ng]);
  }
  __syncthreads();                                   // everyone done reading V before the next chunk overwrites it
}

// scale/bias, residual, ReLU and the two destinations for the wave's 16 tiles (d-pair mh) x 16 couts (half nh)
template <int NG>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& a, const f32x4 (&Y)[NG][8], int b, int d0, int h0, int w0,
                                              int mh, int nh, int lane, int ng0 = 0, const float* scb = nullptr) {
  // ---- epilogue: lane holds cout l&15 for tiles (l>>4)*4 + r of its half; output o = (od, oh, ow)
  // scb: {scale, bias} per column group already in registers (the persistent kernel requests them at the start of the
  // tile, a global-load latency before they are needed)
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int n = (ng0 + ng) * 32 + nh * 16 + (lane & 15);
    const float sc = scb ? scb[2 * ng] : (a.scale ? a.scale[n] : 1.f);
    const float bi = scb ? scb[2 * ng + 1] : (a.bias ? a.bias[n] : 0.f);
    const int tth_e = lane >> 4;
    const int n0 = (ng0 + ng) * 32 + nh * 16;           // first packed column of this wave (uniform)
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
      // all 32 residual values of the column group are requested BEFORE the first store: with an in-place residual
      // (y0 == residual, the BasicBlock3D case) the compiler must keep every later load behind the earlier stores, which
      // made this loop eight dependent global round trips (+9-15 % kernel time, tools/bench_layers.py EPI=1)
      unsigned so[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int od = d0 + 2 * mh + (o >> 2), oh = (o >> 1) & 1, ow = o & 1;
        so[o] = (unsigned)((((b * a.Do + od) * a.Ho + h0 + oh) * a.Wo + w0 + ow) * ld) * 4u;
      }
      float rv[8][4];
      if (has_res) {
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int r = 0; r < 4; ++r) rv[o][r] = buf_load1(rr, lane_o, so[o] + (unsigned)(2 * r * ld) * 4u);
      }
#pragma unroll
      for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = Y[ng][o][r] * sc + bi;
          if (has_res) v += rv[o][r];
          buf_store1(yr, lane_o, so[o] + (unsigned)(2 * r * ld) * 4u, fmaxf(v, lo));
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
    wino_chunk<0, NG>(c, r_base, v_base, ubase + 0u * 16u * ustep, Y);
    wino_chunk<1, NG>(c, r_base, v_base, ubase + 16u * ustep, Y);
    wino_chunk<2, NG>(c, r_base, v_base, ubase + 32u * ustep, Y);
    wino_chunk<3, NG>(c, r_base, v_base, ubase + 48u * ustep, Y);
  }
  wino_epilogue<NG>(a, Y, b, d0, h0, w0, mh, nh, lane);
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised persistent variant: 512 threads = 4 GEMM waves + 4 transform waves, one per SIMD each.
// The kernel above runs its phases back to back in every wave (halo DMA -> transform -> MFMAs -> store)
// with one wave per SIMD, so every LDS / L2 latency and every barrier skew is exposed: the matrix pipe is
// busy 31-42 % of the time and 45 % of the wave cycles are s_waitcnt (profiles/r01_pmc_wino_v6.md).  Here
//   * the transform waves run ONE HALF-STEP AHEAD of the GEMM waves: V is two 32 KB buffers of 8 points
//     (two i_h rows); while the GEMM waves do the MFMAs of half-step g from V[g & 1], the transform waves
//     write half-step g + 1 into V[(g + 1) & 1]; one workgroup barrier per half-step;
//   * the raw halo R is read only by the even half-steps (d-combine + h-transform of all four i_h rows, the
//     second pair stays in registers), the last time in step 3 of a 32-channel chunk (plane d3 is prefetched
//     there): the DMA of the next chunk's / next tile's halo is issued by the transform waves in step 5 and
//     awaited at the end of step 6, so the halo load overlaps the MFMAs too, and the kernel is persistent;
//   * the GEMM waves stream weight rows one row (4 points) ahead across half-steps, chunks and tiles.
// Measured (16x200x200, sustained): 32->32 141 us (tile-per-block kernel 201), 32->64 232 (332), 64->64 431 (623).
// In the first version of this kernel (158 us) the GEMM waves alone took 129 us for 68 us of MFMA time, the transform
// waves alone 76 us, neither 25 us (first DMA, epilogue, barriers): the two instruction streams of a SIMD add up
// (DESIGN.md section 4), so WHERE the transform role's instructions sit between the barriers decides who waits --
// see the schedule in ws_transform_role (pw_wino_common.h), tuned from per-barrier arrival times of all eight waves.
// s_setprio on either role changes nothing or costs 4 % (raised transform waves), so none is set.

// flat row index R of a 32-channel chunk: R = ((ID * 2 + HH) * 2 + r) * NG + ng, i_h = 2 HH + r: the NG
// cout groups of a row of points follow each other and share the row's A operands
template <int R, int NG> struct WsRow {
  static constexpr int H = R / (2 * NG), ID = H >> 1, HH = H & 1, ng = R % NG, r = (R / NG) & 1, IH = 2 * HH + r;
  static constexpr bool first = R % (2 * NG) == 0, last = R % (2 * NG) == 2 * NG - 1;
  static constexpr unsigned point = (unsigned)(ID * 16 + IH * 4);     // first point of the row
  static constexpr unsigned a_off = (unsigned)HH * 32768u + (unsigned)(r * 4) * 4096u;
};

template <int R, int NG>
__device__ __forceinline__ void ws_load_a(const WinoCtx& c, f32x4 (&aq)[4][2]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) aq[k][q] = lds_read4(c.lds3, c.a_addr[q] + WsRow<R, NG>::a_off + (unsigned)k * 4096u);
}

template <int R, int NG>
__device__ __forceinline__ void ws_load_b(const WinoCtx& c, unsigned ubase, f32x4 (&bq)[4][2]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 w = buf_load4(c.wr, c.lane_off + (unsigned)(q * 16),
                                 ubase + (WsRow<R, NG>::point + (unsigned)k) * c.ustep + (unsigned)WsRow<R, NG>::ng * 4096u);
      bq[k][q] = f32x4{w.x, w.y, w.z, w.w};
    }
}

// rows R .. 16 NG - 1 of one chunk; bc holds row R's weights on entry; on exit of the last row bc/ac of the
// CALLER hold the first row of the next chunk (ubase_next) again -- the row count is even, so the ping-pong
// ends where it started
template <int R, int NG>
__device__ __forceinline__ void ws_rows(const WinoCtx& c, unsigned ubase, unsigned ubase_next, f32x4 (&ac)[4][2],
                                        f32x4 (&bc)[4][2], f32x4 (&an)[4][2], f32x4 (&bn)[4][2], f32x4 (&Mp)[4],
                                        f32x4 (&Y)[NG][8]) {
  typedef WsRow<R, NG> W;
  constexpr int TOTAL = 16 * NG;
  // A operands: NG == 1: the first row of a half-step loads its own (the V buffer became valid at the barrier),
  // the second row's are prefetched with its weights; NG == 2: loaded once per row of points (every second R)
  // into a0 and used by both cout groups -- no prefetch, the registers go to the second accumulator set
  f32x4 (&acur)[4][2] = NG == 1 ? ac : (W::ng == 0 ? ac : an);
  if constexpr (NG == 1 ? W::first : W::ng == 0) ws_load_a<R, NG>(c, acur);
  if constexpr (R + 1 < TOTAL) {
    ws_load_b<R + 1, NG>(c, ubase, bn);
    if constexpr (NG == 1 && !W::last) ws_load_a<R + 1, NG>(c, an);
  } else {
    ws_load_b<0, NG>(c, ubase_next, bn);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 M[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) M[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        M[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[k][q][e], bc[k][q][e], M[k], 0, 0, 0);
  if constexpr (R >= 1) {
    typedef WsRow<R - 1, NG> P;
    wino_scatter_row<P::ID, P::IH>(Mp, Y[P::ng]);               // the previous row's products, under this row's MFMAs
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) Mp[k] = M[k];
  if constexpr (W::last) __syncthreads();                       // end of the half-step
  if constexpr (R + 1 < TOTAL) ws_rows<R + 1, NG>(c, ubase, ubase_next, an, bn, ac, bc, Mp, Y);
}

template <int NG>
__global__ void __launch_bounds__(512, 1) k_conv3d_wino_ws(ConvArgs a, PipeArgs p, int n16_total) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;
  const int nchunk = a.Cin / KC;
  const lds3_t lds3 = (lds3_t)lds;

  if (wave >= 4) {
    ws_transform_role(a, p, lds3, item, it_end, nslots, nchunk, wave - 4, tid - 256, lane);
    return;
  }

  // -------------------------------------------------------------------- GEMM + output transform role
  const int mh = wave & 1, nh = wave >> 1;            // tile half (d-pair) and cout half of this wave
  WinoCtx c;
  c.lds3 = lds3;
  {
    const int lt = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      c.a_addr[q] = (unsigned)WINO_R_BYTES + (unsigned)(((mh * 16 + wino_row16(lt)) * 8 + ((g * 2 + q) ^ (lt & 7))) * 16);
  }
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 64 * n16_total * 2048));
  c.lane_off = (unsigned)lane * 32u;
  c.ustep = (unsigned)n16_total * 2048u;
  const unsigned chunk_bytes = 64u * c.ustep;
  f32x4 a0[4][2], a1[4][2], b0[4][2], b1[4][2], Mp[4];
  // work item = (tile, group of NG x 32 couts); the group's weights start (group * 2 NG + nh) n16-blocks in
  PipeTile t = pipe_decode(a, p, item);
  unsigned gbase = (unsigned)((t.ng * 2 * NG + nh) * 2048);
  ws_load_b<0, NG>(c, gbase, b0);
  __syncthreads();                                              // barrier A
  __syncthreads();                                              // barrier B
  for (; item < it_end; item += nslots) {
    f32x4 Y[NG][8];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
      for (int o = 0; o < 8; ++o) Y[ng][o] = f32x4{0.f, 0.f, 0.f, 0.f};
    float scb[2 * NG];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
      const int n = (t.ng * NG + ng) * 32 + nh * 16 + (lane & 15);
      scb[2 * ng] = a.scale ? a.scale[n] : 1.f;
      scb[2 * ng + 1] = a.bias ? a.bias[n] : 0.f;
    }
    const PipeTile tn = pipe_decode(a, p, item + nslots < it_end ? item + nslots : item);
    const unsigned gnext = (unsigned)((tn.ng * 2 * NG + nh) * 2048);
    for (int ch = 0; ch < nchunk; ++ch) {
      const unsigned ubase = (unsigned)ch * chunk_bytes + gbase;
      const unsigned unext = ch + 1 < nchunk ? (unsigned)(ch + 1) * chunk_bytes + gbase : gnext;
      ws_rows<0, NG>(c, ubase, unext, a0, b0, a1, b1, Mp, Y);
      wino_scatter_row<3, 3>(Mp, Y[NG - 1]);
    }
    wino_epilogue<NG>(a, Y, t.b, t.d0, t.h0, t.w0, mh, nh, lane, t.ng * NG, scb);
    t = tn; gbase = gnext;
  }
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
  {
    // wave-specialised persistent kernel (default); PW_WINO_WS=0 keeps the tile-per-block kernel (<= 64 columns).
    // Work item = (tile, group of NG x 32 output columns).  NG = 2 halves the input-transform work per output but
    // halves the item count: taken when that still leaves >= 2 items per CU (PW_WINO_NG overrides).
    const char* e = getenv("PW_WINO_WS");
    if (!e || atoi(e)) {
      const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
      int NG = (cout_total % 64 == 0 && nblk * (cout_total / 64) >= 2ll * nb) ? 2 : 1;
      if (const char* g = getenv("PW_WINO_NG")) NG = (atoi(g) == 2 && cout_total % 64 == 0) ? 2 : 1;
      PipeArgs p = {};
      p.ngroups = cout_total / (32 * NG);
      PW_CHECK_ARG(nblk * p.ngroups < (1ll << 20), "pw_conv3d_wino: too many work items");
      p.n_items = (int)nblk * p.ngroups;
      p.m_ng = magic_of(p.ngroups); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
#define PW_WINO_WS(NGv)                                                                                     \
  do {                                                                                                      \
    static int once = set_lds_limit(k_conv3d_wino_ws<NGv>, WINO_LDS);                                        \
    if (once) return once;                                                                                  \
    hipLaunchKernelGGL(k_conv3d_wino_ws<NGv>, dim3(nb), dim3(512), WINO_LDS, pw_stream(stream), a, p,         \
                       cout_total / 16);                                                                    \
    pw_note_kernel("k_conv3d_wino_ws<%d>", NGv);                                                            \
  } while (0)
      if (NG == 1) PW_WINO_WS(1); else PW_WINO_WS(2);
#undef PW_WINO_WS
      PW_CHECK_LAUNCH();
      return PW_OK;
    }
  }
  const int NG = cout_total / 32;
  PW_CHECK_ARG(NG == 1 || NG == 2, "pw_conv3d_wino: the tile-per-block kernel takes cout_total 32 or 64 (got %d)", cout_total);
#define PW_WINO(NGv)                                                                                      \
  do {                                                                                                    \
    static int once = set_lds_limit(k_conv3d_wino<NGv>, WINO_LDS);                                         \
    if (once) return once;                                                                                \
    hipLaunchKernelGGL(k_conv3d_wino<NGv>, dim3((unsigned)nblk), dim3(256), WINO_LDS, pw_stream(stream), a, \
                       cout_total / 16);                                                                  \
    pw_note_kernel("k_conv3d_wino<%d>", NGv);                                                             \
  } while (0)
  if (NG == 1) PW_WINO(1); else PW_WINO(2);
#undef PW_WINO
  PW_CHECK_LAUNCH();
  return PW_OK;
}
