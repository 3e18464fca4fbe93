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
// Work item = a 4x8x8 output tile (= 2x4x4 = 32 Winograd tiles) x a group of 32 or 64 output channels; the 6x10x10x32ch halo lands
// in LDS by buffer_load ... lds (R, 76.8 KB); thread (tile, channel quad) of the transform role combines its two d-planes, transforms
// along h and w in registers and writes V ([point][tile][32 ch], slots swizzled for conflict-free A reads); a GEMM wave owns 16
// tiles x 16 couts: per point 2 ds_read_b128 (A) + 2 buffer loads (U) + 8 v_mfma_f32_16x16x4_f32 into a fresh accumulator, which is
// then added / subtracted into the (at most 8) outputs it contributes to; epilogue = the direct kernels' (scale / bias, residual,
// ReLU, two destinations, row stride).  The kernel below is the wave-specialised persistent form (the first, tile-per-block form --
// every wave doing every phase in turn, 201 vs 141 us at 32 -> 32 -- was removed in round 4; it is in the history).
#include "pw_wino_common.h"

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
  // Work item = (tile, group of NG x 32 output columns).  NG = 2 halves the input-transform work per output but halves the item
  // count: taken when that still leaves >= 2 items per CU.
  const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
  const int NG = (cout_total % 64 == 0 && nblk * (cout_total / 64) >= 2ll * nb) ? 2 : 1;
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
