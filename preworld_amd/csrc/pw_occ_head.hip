// Fused OccHead kernels (A11) -- entry point pw_occ_head_fused in include/preworld_hip.h.
#include "pw_wino_common.h"
#include "pw_occ_tail.h"

// ------------------------------------------------------------------------------------
// OccHead on v_mfma_f32_16x16x4_f32: the head's 3x3x3 conv has only 16 output channels, which
// would leave half of a 32-wide N tile empty; the 16x16x4 shape (same FLOP rate) has no waste.
//   lane l: A[voxel = l&15][k = l>>4], B[k = l>>4][cout = l&15]; D: col l&15, rows (l>>4)*4 + reg.
// A wave owns the 4 M-tiles (2x8 voxels each) of one d-slice of the 4x8x8 block tile; the four
// accumulators are independent, which covers the 40-cycle dependent latency at 32-cycle issue.
// K order inside a (chunk, tap): k-group g = l>>4 walks channels g*8 .. g*8+7 (2 x ds_read_b128,
// 2 x global_load_dwordx4 of packed weights [chunk][tap][lane][8]).
// Fused tail (mmdet3d/models/heads/occupancy_head.py:92-99,124-161): after conv + BN + ReLU each
// voxel runs 1x1x1 16->8 + BN + ReLU, 1x1x1 8->18 and argmax -> uint8 inside the epilogue; the
// 46 MB logits tensor is only written on request.
// ------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_b16(rsrc_t wr, unsigned wsoff, unsigned lane_off, float4 (&b)[2]) {
  b[0] = buf_load4(wr, lane_off, wsoff);
  b[1] = buf_load4(wr, lane_off + 16u, wsoff);
}

template <int TAP>
__device__ __forceinline__ void tap_mfma16(const float* lds, const unsigned (&aaddr)[2][3][2],
                                           const float4 (&b)[2], f32x4 (&acc)[4]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  float4 aq[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const unsigned imm = (unsigned)(((kd * TH + mt * 2 + kh) * TW) * 128);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      aq[mt][q] = *reinterpret_cast<const float4*>(ldsb + (aaddr[kh & 1][kw][q] + imm));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float bv[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float av[4] = {aq[mt][q].x, aq[mt][q].y, aq[mt][q].z, aq[mt][q].w};
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[mt], 0, 0, 0);
      }
    }
  }
}

template <int TAP>
__device__ __forceinline__ void tap_pair16(const float* lds, const unsigned (&aaddr)[2][3][2],
                                           rsrc_t wr, unsigned wsoff, unsigned lane_off,
                                           float4 (&b0)[2], float4 (&b1)[2], f32x4 (&acc)[4]) {
  if constexpr (TAP + 1 < 27) {
    load_b16(wr, wsoff + (unsigned)(TAP + 1) * 2048u, lane_off, b1);
    __builtin_amdgcn_sched_barrier(0);
    tap_mfma16<TAP>(lds, aaddr, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_b16(wr, wsoff + (unsigned)(TAP + 2) * 2048u, lane_off, b0);
    __builtin_amdgcn_sched_barrier(0);
    tap_mfma16<TAP + 1>(lds, aaddr, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    tap_pair16<TAP + 2>(lds, aaddr, wr, wsoff, lane_off, b0, b1, acc);
  } else {
    tap_mfma16<TAP>(lds, aaddr, b0, acc);
  }
}

template <int WD>
__global__ void __launch_bounds__(256 * WD, 2) k_occ_head16(ConvArgs a, OccTail tail) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int tw = bid % a.tiles_w; bid /= a.tiles_w;
  const int th = bid % a.tiles_h; bid /= a.tiles_h;
  const int td = bid % a.tiles_d;
  const int b = bid / a.tiles_d;
  const int d0 = td * (BD * WD), h0 = th * BH, w0 = tw * BW;
  unsigned aaddr[2][3][2];
#pragma unroll
  for (int khp = 0; khp < 2; ++khp)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = (i & 7) + kw, hr = i >> 3;
      const int f = ((ww >> 1) & 3) | (((hr + khp) & 1) << 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        aaddr[khp][kw][q] = (unsigned)((((wave * TH + hr) * TW + ww) * 8 + ((g * 2 + q) ^ f)) * 16);
    }
  const unsigned lane_off = (unsigned)lane * 32u;
  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  const rsrc_t wr = make_rsrc(a.wpk, (unsigned)((size_t)(a.Cin / KC) * 27 * 2048));
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;
  const int nchunk = a.Cin / KC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const unsigned wsoff = (unsigned)(ch * 27 * 2048);
    float4 b0[2], b1[2];
    load_b16(wr, wsoff, lane_off, b0);
    stage_halo_chunk_dma(a, xr, lds, b, d0, h0, w0, ch, wave, lane);
    tap_pair16<0>(lds, aaddr, wr, wsoff, lane_off, b0, b1, acc);
  }
  // ---- tail: BN+ReLU, transpose 64 voxels x 16 channels through LDS, per-voxel MLP + argmax
  constexpr int MS = 17;
  const int od = d0 + wave;
  __syncthreads();
  float* sm = lds + wave * (64 * MS);
  {
    const float sc = a.scale ? a.scale[i] : 1.f;
    const float bi = a.bias ? a.bias[i] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaxf(acc[mt][r] * sc + bi, 0.f);
        sm[(mt * 16 + g * 4 + r) * MS + i] = v;
      }
  }
  __syncthreads();
  const int mt = lane >> 4, row = lane & 15;
  const int oh = h0 + mt * 2 + (row >> 3), ow = w0 + (row & 7);
  if (od < a.Do && oh < a.Ho && ow < a.Wo) {
    const size_t vox = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
    float mid[16], hid[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) mid[k] = sm[lane * MS + k];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float s_ = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s_ += mid[k] * tail.w1[o * 16 + k];
      hid[o] = fmaxf(s_ * tail.s1[o] + tail.b1[o], 0.f);
    }
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < 18; ++c) {
      float s_ = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) s_ += hid[o] * tail.w2[c * 8 + o];
      if (tail.logits) tail.logits[vox * 18 + c] = s_;
      if (c == 0 || s_ > best) { best = s_; arg = c; }
    }
    tail.occ[vox] = (uint8_t)arg;
    if (tail.geo) tail.geo[vox] = arg != tail.empty_idx ? (uint8_t)0 : (uint8_t)(tail.n_cls - 1);
  }
}

// A11  fused OccHead: conv3x3x3 (Cin->16, BN, ReLU) + 1x1x1 16->8 (BN, ReLU) + 1x1x1 8->18 + argmax
// ------------------------------------------------------------------------------------
// OccHead on the Winograd machinery (pw_wino_common.h, pw_conv3d_wino.hip): the 3x3x3 32->16 conv as
// F(2x2x2,3x3x3) in the wave-specialised persistent form -- 4 transform + DMA waves one half-step ahead of
// 4 GEMM waves -- with the 16->8->18 + argmax tail in the GEMM waves' epilogue.
// 16 output columns are one MFMA N-tile, so the four GEMM waves split (tile half mh) x (row of points RSEL
// of each half-step) instead of (tile half) x (column half): every wave accumulates a partial sum over its
// rows for all 8 outputs of its 16 tiles; at the end of a tile the two row-partners swap halves through LDS
// (wave RSEL keeps o_d = RSEL) under the half-step's own barrier, and each wave finishes 64 voxels x 16
// channels -- exactly the shape of k_occ_head16's tail (transpose through LDS, one voxel per lane).
// LDS: R 76.8 KB + V 2 x 32 KB + 4 x 4352 B exchange / transpose areas + 1 152 B of tail weights = 160 896 B.
// The 16->8->18 weights sit in LDS for the life of the (persistent) block and are read as broadcasts: fetching them per
// tile with 72 wave-wide global loads made the tail as long as all of the tile's MFMAs (7.9 k of 25 k cycles, probe).
constexpr int OCCW_XCH = 64 * 17 * 4;                            // per GEMM wave
constexpr int OCCW_TAILW = WINO_LDS + 4 * OCCW_XCH;              // 288 floats: w1 [8][16], s1 [8], b1 [8], w2 [18][8]
constexpr int OCCW_LDS = OCCW_TAILW + 288 * 4;

template <int H, int RSEL> struct OccRow {                       // the wave's row of half-step H
  static constexpr int ID = H >> 1, HH = H & 1, IH = 2 * HH + RSEL;
  static constexpr unsigned point = (unsigned)(ID * 16 + IH * 4);
  static constexpr unsigned a_off = (unsigned)HH * 32768u + (unsigned)(RSEL * 4) * 4096u;
};

template <int H, int RSEL>
__device__ __forceinline__ void occw_load_b(const WinoCtx& c, f32x4 (&bq)[4][2]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 w = buf_load4(c.wr, c.lane_off + (unsigned)(q * 16), (OccRow<H, RSEL>::point + (unsigned)k) * 2048u);
      bq[k][q] = f32x4{w.x, w.y, w.z, w.w};
    }
}

// half-steps H..7 of one tile.  Entry: the barrier that made V[H & 1] valid has been passed, bc = this row's
// weights.  Order inside a half-step: A loads, next row's weights, the PREVIOUS row's output transform (fills
// the LDS latency), 32 MFMAs, barrier.
template <int H, int RSEL>
__device__ __forceinline__ void occw_rows(const WinoCtx& c, f32x4 (&bc)[4][2], f32x4 (&bn)[4][2], f32x4 (&Mp)[4],
                                          f32x4 (&Y)[8], unsigned xch_mine) {
  typedef OccRow<H, RSEL> W;
  f32x4 aq[4][2];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) aq[k][q] = lds_read4(c.lds3, c.a_addr[q] + W::a_off + (unsigned)k * 4096u);
  occw_load_b<(H + 1) & 7, RSEL>(c, bn);
  if constexpr (H >= 1) wino_scatter_row<OccRow<H - 1, RSEL>::ID, OccRow<H - 1, RSEL>::IH>(Mp, Y);
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
        M[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[k][q][e], bc[k][q][e], M[k], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (H < 7) {
#pragma unroll
    for (int k = 0; k < 4; ++k) Mp[k] = M[k];
    __syncthreads();
    occw_rows<H + 1, RSEL>(c, bn, bc, Mp, Y, xch_mine);
  } else {
    wino_scatter_row<W::ID, W::IH>(M, Y);
    // the partner finishes o_d = 1 - RSEL: hand it those four outputs
    const unsigned lane16 = (unsigned)(threadIdx.x & 63) * 16u;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds_write4(c.lds3, xch_mine + (unsigned)j * 1024u + lane16, Y[(1 - RSEL) * 4 + j]);
    __syncthreads();
  }
}

template <int RSEL>
__device__ __forceinline__ void occw_gemm_role(const ConvArgs& a, const PipeArgs& p, const OccTail& tail, lds3_t lds3,
                                               int item, int it_end, int nslots, int mh, int lane) {
  WinoCtx c;
  c.lds3 = lds3;
  {
    const int lt = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      c.a_addr[q] = (unsigned)WINO_R_BYTES + (unsigned)(((mh * 16 + wino_row16(lt)) * 8 + ((g * 2 + q) ^ (lt & 7))) * 16);
  }
  c.wr = make_rsrc(a.wpk, 64u * 2048u);
  c.lane_off = (unsigned)lane * 32u;
  c.ustep = 2048u;
  const unsigned xch_mine = (unsigned)WINO_LDS + (unsigned)((mh * 2 + RSEL) * OCCW_XCH);
  const unsigned xch_peer = (unsigned)WINO_LDS + (unsigned)((mh * 2 + 1 - RSEL) * OCCW_XCH);
  const int i = lane & 15, g = lane >> 4;
  const float sc = a.scale ? a.scale[i] : 1.f;
  const float bi = a.bias ? a.bias[i] : 0.f;
  f32x4 b0[4][2], b1[4][2], Mp[4];
  occw_load_b<0, RSEL>(c, b0);
  {
    __attribute__((address_space(3))) float* wl = reinterpret_cast<__attribute__((address_space(3))) float*>(lds3 + OCCW_TAILW);
    const int tid = threadIdx.x;                                // 0..255: the four GEMM waves
    if (tid < 128) wl[tid] = tail.w1[tid];
    else if (tid < 136) wl[tid] = tail.s1[tid - 128];
    else if (tid < 144) wl[tid] = tail.b1[tid - 136];
    if (tid < 144) wl[144 + tid] = tail.w2[tid];
  }
  __syncthreads();                                              // barrier A (halo of the first tile; tail weights in LDS)
  __syncthreads();                                              // barrier B (half-step 0 in V[0])
  for (; item < it_end; item += nslots) {
    f32x4 Y[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) Y[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    occw_rows<0, RSEL>(c, b0, b1, Mp, Y, xch_mine);             // 8 rows: b0 holds row 0 of the next tile again
    // ---- tail: add the partner's half, BN + ReLU, transpose 64 voxels x 16 channels through the partner's area
    // (it is ours once read), then one voxel per lane: 16->8 (+BN+ReLU), 8->18, argmax
    constexpr int MS = 17;
    f32x4 Z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Z[j] = Y[RSEL * 4 + j] + lds_read4(lds3, xch_peer + (unsigned)j * 1024u + (unsigned)lane * 16u);
    __attribute__((address_space(3))) float* sm = reinterpret_cast<__attribute__((address_space(3))) float*>(lds3 + xch_peer);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // tile (t_h = g, t_w = r) of the wave's d-pair, output (o_h, o_w) = (j >> 1, j & 1) -> voxel (h, w) of the 8x8 plane
        const int vh = 2 * g + (j >> 1), vw = 2 * r + (j & 1);
        sm[(vh * 8 + vw) * MS + i] = fmaxf(Z[j][r] * sc + bi, 0.f);
      }
    const PipeTile t = pipe_decode(a, p, item);
    const int od = t.d0 + 2 * mh + RSEL, oh = t.h0 + (lane >> 3), ow = t.w0 + (lane & 7);
    if (od < a.Do && oh < a.Ho && ow < a.Wo) {
      const size_t vox = (((size_t)t.b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
      float mid[16], hid[8];
#pragma unroll
      for (int k = 0; k < 16; ++k) mid[k] = sm[lane * MS + k];
      const unsigned wq = (unsigned)OCCW_TAILW;                   // uniform addresses: LDS broadcasts
      // every layer's weights are requested in one go and consumed afterwards: read one by one next to their FMAs the 72
      // LDS round trips were the tail (6.5 k cycles of a 25 k-cycle tile)
      f32x4 W1[36];
#pragma unroll
      for (int q = 0; q < 36; ++q) W1[q] = lds_read4(lds3, wq + (unsigned)q * 16u);            // w1 (32), s1 (2), b1 (2)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float s_ = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s_ += mid[k] * W1[o * 4 + (k >> 2)][k & 3];
        hid[o] = fmaxf(s_ * W1[32 + (o >> 2)][o & 3] + W1[34 + (o >> 2)][o & 3], 0.f);
      }
      f32x4 W2[36];
#pragma unroll
      for (int q = 0; q < 36; ++q) W2[q] = lds_read4(lds3, wq + 576u + (unsigned)q * 16u);     // w2 [18][8]
      __builtin_amdgcn_sched_barrier(0);
      float best = 0.f;
      int arg = 0;
#pragma unroll
      for (int cc = 0; cc < 18; ++cc) {
        float s_ = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) s_ += hid[o] * W2[cc * 2 + (o >> 2)][o & 3];
        if (tail.logits) tail.logits[vox * 18 + cc] = s_;
        if (cc == 0 || s_ > best) { best = s_; arg = cc; }
      }
      tail.occ[vox] = (uint8_t)arg;
      if (tail.geo) tail.geo[vox] = arg != tail.empty_idx ? (uint8_t)0 : (uint8_t)(tail.n_cls - 1);
    }
  }
}

__global__ void __launch_bounds__(512, 1) k_occ_head_wino(ConvArgs a, PipeArgs p, OccTail tail) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;
  const lds3_t lds3 = (lds3_t)lds;
  if (wave >= 4) {
    ws_transform_role(a, p, lds3, item, it_end, nslots, 1, wave - 4, tid - 256, lane);
    return;
  }
  if (wave >> 1) occw_gemm_role<1>(a, p, tail, lds3, item, it_end, nslots, wave & 1, lane);
  else occw_gemm_role<0>(a, p, tail, lds3, item, it_end, nslots, wave & 1, lane);
}

PW_API int pw_occ_head_fused(const float* x, const float* wpk, const float* scale, const float* bias,
                             const float* w1, const float* s1, const float* b1, const float* w2,
                             uint8_t* occ, float* logits, uint8_t* geo, int empty_idx, int B, int D,
                             int H, int W, int Cin, int n_mid, int n_hid, int n_cls, int wpk_layout,
                             void* stream) {
  PW_CHECK_ARG(x && wpk && w1 && s1 && b1 && w2 && occ, "pw_occ_head_fused: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cin % KC == 0, "pw_occ_head_fused: bad shape");
  if (n_mid != 16 || n_hid != 8 || n_cls != 18) {
    pw_set_error("pw_occ_head_fused: only the PreWorld head shape 16/8/18 is built (got %d/%d/%d)",
                 n_mid, n_hid, n_cls);
    return PW_EUNSUP;
  }
  ConvArgs a = {};
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = 32; a.cout0 = n_mid; a.relu0 = 1;
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  OccTail t = {w1, s1, b1, w2, occ, logits, geo, empty_idx, n_mid, n_hid, n_cls};
  long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (wpk_layout == 64) {
    // Winograd-domain weights (pack_conv_weight_wino with 16 columns): the wave-specialised persistent kernel
    PW_CHECK_ARG(Cin == KC, "pw_occ_head_fused: the Winograd kernel takes 32 input channels");
    PW_CHECK_ARG(nblk < (1ll << 20), "pw_occ_head_fused: too many tiles");
    PipeArgs p = {};
    p.ngroups = 1; p.n_items = (int)nblk;
    p.m_ng = magic_of(1); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
    const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
    static int once = set_lds_limit(k_occ_head_wino, OCCW_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head_wino, dim3(nb), dim3(512), OCCW_LDS, pw_stream(stream), a, p, t);
    pw_note_kernel("k_occ_head_wino");
    PW_CHECK_LAUNCH();
    return PW_OK;
  }
  PW_CHECK_ARG(wpk_layout == 16, "pw_occ_head_fused: wpk_layout must be 16 (direct 16x16x4 MFMA packing) or 64 (Winograd)");
  {
    static int once = set_lds_limit(k_occ_head16<1>, TileGeom<1>::LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head16<1>, dim3((unsigned)nblk, 1), dim3(256), TileGeom<1>::LDS,
                       pw_stream(stream), a, t);
    pw_note_kernel("k_occ_head16<1>");
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}

