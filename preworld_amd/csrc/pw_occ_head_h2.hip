// Fused OccHead (A11) on the fp16 matrix cores with split-fp16 operands -- entry point pw_occ_head_h2.
//   mmdet3d/models/heads/occupancy_head.py:92-99,124-177: conv3x3x3 32->16 (no bias) + BN + ReLU, 1x1x1 16->8 + BN + ReLU,
//   1x1x1 8->18, argmax -> uint8; geo_occ of detectors/preworld_temporal_traj.py:313-319 from the same kernel.
//
// v_mfma_f32_16x16x32_f16 has M = 16 = the conv's output channels, so nothing of the tile is wasted (a 32-wide tile would be
// half empty), and with Cin = 32 one instruction covers a whole tap.  The GEMM is transposed like k_conv3d_h2:
//     D[cout][voxel] += W[cout][32 ch] X[32 ch][voxel]        3 instructions per tap: hi.hi, lo_w.hi_x, hi_w.lo_x (pw_h2.h)
//   * A (weights): ALL 27 taps x {hi, lo} stay in registers for the life of the persistent block (216 VGPRs; one wave per
//     SIMD has 512), so the tap loop issues no weight loads at all;
//   * B (activations): lane (voxel i = l & 15, k-group g = l >> 4) reads the 8 channels 16 (g >> 1) + 8 (g & 1) + 0..7 of its
//     voxel: hi and lo are the two adjacent 16-byte slots 4 (g & 1) + 2 (g >> 1) + {0, 1} of the voxel's 128-byte chunk
//     in h2 storage -- 2 ds_read_b128 per fragment;
//   * a wave owns one d-slice of the 4x8x8 tile as 4 groups of 16 voxels; group m = output rows {m, m + 4} x 8 columns.
//     Tap (kd, kh, kw) of group m reads halo rows {m + kh, m + kh + 4}: the same fragment as tap (kd, 0, kw) of group
//     m + kh.  So per (kd, kw) the 6 fragments R(j) = halo rows {j, j + 4}, j = 0..5, feed all 4 groups x 3 kh = 12
//     (group, tap) pairs: 12 ds_read_b128 per 36 MFMAs (24 without the sharing).  Rows j and j + 4 must not collide in the
//     LDS banks, so this kernel's halo swizzle takes bit 2 of the halo row (pw_conv3d_common.h uses bit 0): the DMA staging
//     applies it on the global side as usual.
// Stage = one 4x8x8 tile: the 6x10x10 halo of the NEXT tile lands in the second LDS buffer by `buffer_load ... lds` while
// this one computes (15 rows per wave, issued between the MFMA groups); one barrier per tile.
// Tail, also on the matrix cores and entirely in registers: BN + ReLU on the accumulators gives, per lane, 4 consecutive mid
// channels (4 g + r) of one voxel -- exactly the B operand of v_mfma_f32_16x16x16_f16 (k = 4 g + e).  So 16->8 (+BN+ReLU) and
// 8->18 chain as split-fp16 products (3 + 2 x 3 small MFMAs per 16 voxels) on weight fragments that live in 12 VGPRs, with no
// LDS transpose and no weight traffic; the 18 logits of a voxel end up spread over its 4 lanes (classes 4 g + r, and 16 / 17
// in lane g = 0 of the second M-tile), argmax = local scan + two v_permlane{16,32}_swap exchanges, ties -> lowest class
// like torch.argmax.  The tail of a tile is a phase of its own behind the stage's barrier (2.7 k of the 10.6 k cycles of a stage,
// profiles/r06_stage_probes.txt): with one wave per SIMD a 4-pass MFMA blocks the wave's in-order issue for its 16 cycles, so tail
// instructions moved into the next tile's tap loop cost there what they save here (round 2 measured both interleavings: slower /
// equal; DESIGN_HISTORY.md 5.2).  Stores are bounds-checked by out-of-range buffer offsets, so the tail is one branch-free region.
#include "pw_h2.h"
#include "pw_occ_tail.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int OH_LDS = 2 * PIPE_BUF_BYTES;                       // 153 600
constexpr int OH_ROW_BYTES = TW * 128;                           // 1 280: one halo row = 16 parked voxels x 80 B
constexpr int OH_WPK_BYTES = 27 * 2 * 64 * 16;                   // packed conv weights [tap][plane][lane][8 halfs]
constexpr int OH_TAILPK_BYTES = 800 * 4;                         // tail operands (pack_occ_tail_h2)
}  // namespace

// halo row `wave + 4 K` of the stage described by dm (pipe_dma_row with this kernel's swizzle bit)
template <int K>
__device__ __forceinline__ void oh_dma_row(const ConvArgs& a, rsrc_t xr, lds3_t lds3, const PipeDma& dm, int wave) {
  const int row = wave + 4 * K;                    // wave-uniform, < 60
  const int dd = row / TH, hh = row - dd * TH;
  const int gd = dm.d0 + dd - 1, gh = dm.h0 + hh - 1;
  const bool rok = dm.live && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H;
  // branch-free on purpose (a branch would split the scheduling region the MFMAs are interleaved in): an out-of-volume row
  // keeps a harmless scalar offset and sends every lane out of range instead
  const unsigned soff = (unsigned)(((((dm.b * a.D + gd) * a.H + gh) * a.W + dm.wbase) * KC) * 4) & (rok ? 0xffffffffu : 0u);
  const int par = (hh >> 2) & 1;
  const unsigned v0 = rok ? (par ? dm.voff[1][0] : dm.voff[0][0]) : PIPE_OOB;
  const unsigned v1 = rok ? (par ? dm.voff[1][1] : dm.voff[0][1]) : PIPE_OOB;
  lds3_t dst = lds3 + (dm.ldsbuf + (unsigned)row * OH_ROW_BYTES);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v0, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 4, v1, soff, 0, 0);
}

// the same row inside the tap loop, with the per-stage part of the address arithmetic hoisted (OhCtx::dm_base / dm_pitch) and the row's
// (d, h) split done on compile-time constants + one carry -- ~10 instead of ~22 scalar instructions per row; a single wave per SIMD pays
// 4 cycles of issue for each of them (k_conv3d_h2's h2_dma_row does the same)
template <int K>
__device__ __forceinline__ void oh_dma_row_fast(const ConvArgs& a, rsrc_t xr, lds3_t lds3, const PipeDma& dm, unsigned base, unsigned pitch,
                                                int wave) {
  constexpr int c1 = (4 * K) / TH, c2 = (4 * K) % TH;
  const int t = c2 + wave;                                   // wave-uniform
  const int carry = t >= TH ? 1 : 0;
  const int dd = c1 + carry, hh = t - TH * carry;
  const bool rok = (int)dm.live & (int)((unsigned)(dm.d0 - 1 + dd) < (unsigned)a.D) & (int)((unsigned)(dm.h0 - 1 + hh) < (unsigned)a.H);
  const unsigned soff = base + (unsigned)(dd * a.H + hh) * pitch;      // rows outside the volume: every lane is out of range
  const int par = (hh >> 2) & 1;
  const unsigned v0 = rok ? (par ? dm.voff[1][0] : dm.voff[0][0]) : PIPE_OOB;
  const unsigned v1 = rok ? (par ? dm.voff[1][1] : dm.voff[0][1]) : PIPE_OOB;
  lds3_t dst = lds3 + (dm.ldsbuf + (unsigned)(c1 * TH + c2) * OH_ROW_BYTES + (unsigned)wave * OH_ROW_BYTES);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v0, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 4, v1, soff, 0, 0);
}

// the 6 x {hi, lo} activation fragments of (kd, kw): R[j][p] = plane p of halo rows {j, j + 4} of d-slice wave + kd
template <int KD, int KW>
__device__ __forceinline__ void oh_read_group(lds3_t lds3, const unsigned (&ad)[3][2][2], h8 (&R)[6][2]) {
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned imm = (unsigned)(((KD * TH + j) * TW) * 128);
      const v4f v = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(lds3 + ad[KW][j >> 2][p] + imm);
      R[j][p] = __builtin_bit_cast(h8, v);
    }
}

template <int KD, int KW>
__device__ __forceinline__ void oh_mfma_group(const h8 (&wh)[27], const h8 (&wl)[27], const h8 (&R)[6][2], f32x4 (&acc)[4]) {
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    constexpr int tap0 = KD * 9 + KW;
    const int tap = tap0 + 3 * kh;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], R[m + kh][0], acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[tap], R[m + kh][0], acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], R[m + kh][1], acc[m], 0, 0, 0);
  }
}

struct OhCtx {
  lds3_t lds3;
  rsrc_t xr, occr, geor, lgr;
  PipeDma dm;
  unsigned dm_base, dm_pitch;    // next stage's halo: byte offset of row (d0 - 1, h0 - 1) at wbase; bytes per h row
  int wave, g;
  float inv2;
};

struct OhTailW {                 // per-lane resident operands of the 16 -> 8 -> 18 tail (pack_occ_tail_h2)
  h4 w1h, w1l, w2h[2], w2l[2];
  float s1[4], b1[4];            // folded BN of hid channels 4 g + r, divided by W1's pre-scale; zero for g >= 2
};

struct OhPend {                  // the tile whose tail is in flight
  f32x4 mid[4];                  // relu(BN(conv)): channels 4 g + r of voxel i of group m
  unsigned vox[4];               // voxel index of (group m, i), PIPE_OOB when outside the volume / nothing pending
  unsigned ovx[4];               // byte offset of that voxel in the occ / geo grids (== vox unless the caller gave strides)
};

__device__ __forceinline__ void oh_split(const f32x4& v, h4& hi, h4& lo) {
  const float t[4] = {v[0], v[1], v[2], v[3]};
  u2 a, b;
  h2_split4(t, a, b);
  hi = __builtin_bit_cast(h4, a);
  lo = __builtin_bit_cast(h4, b);
}

// mid -> hid -> logits of the four groups, written stage by stage ACROSS the groups so that the four dependent chains
// interleave (L[m][0]: classes 4 g + r, L[m][1]: classes 16 + 4 g + r of group m's voxel i; all x W2's pre-scale)
__device__ __forceinline__ void oh_tail_a(const OhTailW& tw, const OhPend& pe, f32x4 (&L)[4][2]) {
  h4 mh[4], ml[4];
  f32x4 ha[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) oh_split(pe.mid[m], mh[m], ml[m]);
#pragma unroll
  for (int m = 0; m < 4; ++m) ha[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w1h, mh[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) ha[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w1l, mh[m], ha[m], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 4; ++m) ha[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w1h, ml[m], ha[m], 0, 0, 0);
  h4 hh[4], hl[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 hid;
#pragma unroll
    for (int r = 0; r < 4; ++r) hid[r] = fmaxf(fmaf(ha[m][r], tw.s1[r], tw.b1[r]), 0.f);
    oh_split(hid, hh[m], hl[m]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) L[m][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w2h[t], hh[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) L[m][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w2l[t], hh[m], L[m][t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) L[m][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(tw.w2h[t], hl[m], L[m][t], 0, 0, 0);
}

// (best, idx) of the partner half / row against mine: the larger logit, on a tie the lower class (torch.argmax)
__device__ __forceinline__ void oh_pick(float& best, int& idx, float ob, int oi) {
  const bool take = (ob > best) | ((ob == best) & (oi < idx));       // no short-circuit: stays one basic block
  best = take ? ob : best;
  idx = take ? oi : idx;
}
__device__ __forceinline__ void oh_scan(float& best, int& idx, float v, int cls) {   // classes visited in ascending order
  const bool take = v > best;
  best = take ? v : best;
  idx = take ? cls : idx;
}

// piece B of group M: argmax over the voxel's 18 classes, stores
template <int M, bool LOGITS>
__device__ __forceinline__ void oh_tail_b(const OhCtx& c, const OccTail& tail, const OhPend& pe, const f32x4 (&L)[2]) {
  const int g = c.g;
  float best = L[0][0];
  int idx = 4 * g;
#pragma unroll
  for (int r = 1; r < 4; ++r) oh_scan(best, idx, L[0][r], 4 * g + r);
  // classes 16, 17 live in lane g = 0 of the second M-tile
  oh_scan(best, idx, g == 0 ? L[1][0] : -INFINITY, 16);
  oh_scan(best, idx, g == 0 ? L[1][1] : -INFINITY, 17);
  {
    const auto vb = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto vi = __builtin_amdgcn_permlane16_swap((unsigned)idx, (unsigned)idx, false, false);
    best = __uint_as_float(vb[0]); idx = (int)vi[0];
    oh_pick(best, idx, __uint_as_float(vb[1]), (int)vi[1]);
  }
  {
    const auto vb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto vi = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
    best = __uint_as_float(vb[0]); idx = (int)vi[0];
    oh_pick(best, idx, __uint_as_float(vb[1]), (int)vi[1]);
  }
  const unsigned vox = pe.vox[M];
  const unsigned vb = g == 0 ? pe.ovx[M] : PIPE_OOB;                // one of the voxel's four lanes writes the bytes
  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)idx, c.occr, vb, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b8(idx != tail.empty_idx ? (unsigned char)0 : (unsigned char)(tail.n_cls - 1), c.geor, vb, 0, 0);
  if constexpr (LOGITS) {
    const unsigned lo = vox == PIPE_OOB ? PIPE_OOB : vox * 72u;
    const float t0[4] = {L[0][0] * c.inv2, L[0][1] * c.inv2, L[0][2] * c.inv2, L[0][3] * c.inv2};
    buf_store4(c.lgr, lo == PIPE_OOB ? PIPE_OOB : lo + 16u * (unsigned)g, 0, t0);
    u2 t1;
    t1[0] = __float_as_uint(L[1][0] * c.inv2); t1[1] = __float_as_uint(L[1][1] * c.inv2);
    buf_store2(c.lgr, (g == 0 && lo != PIPE_OOB) ? lo + 64u : PIPE_OOB, 0, t1);
  }
}

// group G = KD * 3 + KW of a stage: fragments of group G + 1 are requested, up to two halo rows of the next tile are DMA'd, and
// the 36 MFMAs of group G run on the fragments read one group earlier
template <int G>
__device__ __forceinline__ void oh_step(const ConvArgs& a, const OhCtx& c, const unsigned (&ad)[3][2][2], const h8 (&wh)[27],
                                        const h8 (&wl)[27], h8 (&Rc)[6][2], h8 (&Rn)[6][2], f32x4 (&acc)[4]) {
  if constexpr (G + 1 < 9) oh_read_group<(G + 1) / 3, (G + 1) % 3>(c.lds3, ad, Rn);
#ifdef PW_X_OH_OLD_DMA          // (A/B build: the row addressing of rounds 2-4)
  if constexpr (2 * G < PIPE_ROWS_PER_WAVE) oh_dma_row<2 * G>(a, c.xr, c.lds3, c.dm, c.wave);
  if constexpr (2 * G + 1 < PIPE_ROWS_PER_WAVE) oh_dma_row<2 * G + 1>(a, c.xr, c.lds3, c.dm, c.wave);
#else
  if constexpr (2 * G < PIPE_ROWS_PER_WAVE) oh_dma_row_fast<2 * G>(a, c.xr, c.lds3, c.dm, c.dm_base, c.dm_pitch, c.wave);
  if constexpr (2 * G + 1 < PIPE_ROWS_PER_WAVE) oh_dma_row_fast<2 * G + 1>(a, c.xr, c.lds3, c.dm, c.dm_base, c.dm_pitch, c.wave);
#endif
  oh_mfma_group<G / 3, G % 3>(wh, wl, Rc, acc);
  // one scheduling region: a fragment read after every third MFMA
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    if (G + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (G + 1 < 9) oh_step<G + 1>(a, c, ad, wh, wl, Rn, Rc, acc);
}

// a-priori magnitude bounds of the two hidden layers (host: preworld_amd.ops.pack_occ_*_h2), see the range note in the kernel
struct OhBounds { float mid_a, mid_b, hid_a, hid_b; };

template <bool LOGITS>
__global__ void __launch_bounds__(256, 1) k_occ_head_h2(ConvArgs a, PipeArgs p, OccTail tail, const float* tailpk, float inv2,
                                                        OhBounds bd) {
#ifdef PW_X_SKIP_OCC            // ablation builds only
  return;
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  // MFMA column c = l & 15 <-> voxel i of a group: any bijection works for the matrix cores, this one makes the fragment reads
  // conflict-free under ds_read_b128's lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with i = c the two lanes of a
  // group that share (w & 1, (w >> 1) & 3) also shared the row bit -- 2-way conflicts on every read (PMC: 4 extra LDS cycles
  // per instruction)
  const int g = lane >> 4, cidx = lane & 15;
  const int i = (cidx & 1) | (((cidx >> 2) & 1) << 1) | (((cidx >> 3) & 1) << 2) | (((cidx >> 1) & 1) << 3);
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;

  OhCtx c;
  c.lds3 = (lds3_t)lds;
  const unsigned nvox = (unsigned)((size_t)a.B * a.D * a.H * a.W);
  c.xr = make_rsrc(a.x, nvox * (unsigned)(KC * 4));
  const unsigned obytes = tail.span;
  c.occr = make_rsrc(tail.occ, obytes);
  c.geor = make_rsrc(tail.geo, tail.geo ? obytes : 0u);
  c.lgr = make_rsrc(tail.logits, tail.logits ? nvox * 72u : 0u);
  c.wave = wave; c.g = g;
  // this lane's share of a uint8 grid offset (the host always passes strides: contiguous (B, D, H, W) when the caller gave none)
  const int lane_ovx = wave * tail.sd + (i & 7) * tail.sw + 4 * (i >> 3) * tail.sh;
  const int e_in = rng_exp(a.x_rng);        // (the load goes out here; first use behind the prologue's halo DMA)

  // fragment addresses: [kw][j >= 4][plane] for halo rows {j, j + 4} (row-pair base at j = 0 / immediates add (kd, j))
  unsigned ad0[3][2][2];
  {
    const int w = i & 7, hr = i >> 3;
    const int slot = 4 * (g & 1) + 2 * (g >> 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        const int ww = w + kw;
        const int f = ((ww >> 1) & 3) | ((hr ^ jh) << 2);
        const unsigned base = (unsigned)((((wave * TH + 4 * hr) * TW + ww) * 128) + ((slot ^ f) * 16));
        ad0[kw][jh][0] = base;
        ad0[kw][jh][1] = base ^ 16u;
      }
  }

  // conv weights: 27 taps x {hi, lo}, resident
  h8 wh[27], wl[27];
  {
    const rsrc_t wr = make_rsrc(a.wpk, (unsigned)OH_WPK_BYTES);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const auto vh = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)lane * 16u, (unsigned)((2 * t) * 1024), 0);
      const auto vl = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)lane * 16u, (unsigned)((2 * t + 1) * 1024), 0);
      typedef unsigned bu4 __attribute__((ext_vector_type(4)));
      bu4 th, tl;
#pragma unroll
      for (int e = 0; e < 4; ++e) { th[e] = vh[e]; tl[e] = vl[e]; }
      wh[t] = __builtin_bit_cast(h8, th);
      wl[t] = __builtin_bit_cast(h8, tl);
    }
  }
  OhTailW tw;
  {
    const rsrc_t tr = make_rsrc(tailpk, (unsigned)OH_TAILPK_BYTES);
    const u2 f0 = buf_load2(tr, (unsigned)lane * 8u, 0u), f1 = buf_load2(tr, (unsigned)lane * 8u, 512u);
    const u2 f2 = buf_load2(tr, (unsigned)lane * 8u, 1024u), f3 = buf_load2(tr, (unsigned)lane * 8u, 1536u);
    const u2 f4 = buf_load2(tr, (unsigned)lane * 8u, 2048u), f5 = buf_load2(tr, (unsigned)lane * 8u, 2560u);
    tw.w1h = __builtin_bit_cast(h4, f0); tw.w1l = __builtin_bit_cast(h4, f1);
    tw.w2h[0] = __builtin_bit_cast(h4, f2); tw.w2l[0] = __builtin_bit_cast(h4, f3);
    tw.w2h[1] = __builtin_bit_cast(h4, f4); tw.w2l[1] = __builtin_bit_cast(h4, f5);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tw.s1[r] = tailpk[768 + 4 * g + r];
      tw.b1[r] = tailpk[784 + 4 * g + r];
    }
  }
  // folded BN of the conv for this lane's 4 channels (4 g + r); the weights' power-of-two pre-scale is folded into scale
  float sc[4], bi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = a.scale[4 * g + r];
    bi[r] = a.bias[4 * g + r];
  }

  PipeTile t = pipe_decode(a, p, item);
  {
    PipeDma dm;
    pipe_lane_offsets(a, t.w0, lane, dm.voff);
    dm.b = t.b; dm.d0 = t.d0; dm.h0 = t.h0; dm.wbase = t.w0 > 0 ? t.w0 - 1 : 0; dm.ch = 0; dm.ldsbuf = 0; dm.live = true;
    oh_dma_row<0>(a, c.xr, c.lds3, dm, wave); oh_dma_row<1>(a, c.xr, c.lds3, dm, wave); oh_dma_row<2>(a, c.xr, c.lds3, dm, wave);
    oh_dma_row<3>(a, c.xr, c.lds3, dm, wave); oh_dma_row<4>(a, c.xr, c.lds3, dm, wave); oh_dma_row<5>(a, c.xr, c.lds3, dm, wave);
    oh_dma_row<6>(a, c.xr, c.lds3, dm, wave); oh_dma_row<7>(a, c.xr, c.lds3, dm, wave); oh_dma_row<8>(a, c.xr, c.lds3, dm, wave);
    oh_dma_row<9>(a, c.xr, c.lds3, dm, wave); oh_dma_row<10>(a, c.xr, c.lds3, dm, wave); oh_dma_row<11>(a, c.xr, c.lds3, dm, wave);
    oh_dma_row<12>(a, c.xr, c.lds3, dm, wave); oh_dma_row<13>(a, c.xr, c.lds3, dm, wave); oh_dma_row<14>(a, c.xr, c.lds3, dm, wave);
    // Range (pw_h2.h "Range").  x is read under its slot's exponent e_in.  The two hidden layers are split in registers, so their
    // units are chosen from A-PRIORI bounds: |mid| <= mid_a 2^(16 + e_in) + mid_b (mid_a = max_c |scale_c| ||S w_c||_1: every
    // stored input is below 2^16) and |hid| <= hid_a |mid|_max + hid_b; each layer is computed directly in units that put its
    // bound at 2^15.  Powers of two, folded into the BN constants below; the logits come back in true units through inv2.
    int e_mid, e_hid;
    {
      const float midb = fmaf(bd.mid_a, rng_pow2(16 + e_in), bd.mid_b);
      const float hidb = fmaf(bd.hid_a, midb, bd.hid_b);
      int ex;
      (void)frexpf(midb, &ex);
      e_mid = midb > 0.f ? __builtin_amdgcn_readfirstlane(ex) - 15 : 0;
      (void)frexpf(hidb, &ex);
      e_hid = hidb > 0.f ? __builtin_amdgcn_readfirstlane(ex) - 15 : 0;
      e_mid = e_mid < -100 ? -100 : (e_mid > 100 ? 100 : e_mid);
      e_hid = e_hid < -100 ? -100 : (e_hid > 100 ? 100 : e_hid);
    }
    c.inv2 = inv2 * rng_pow2(e_hid);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] *= rng_pow2(e_in - e_mid); bi[r] *= rng_pow2(-e_mid);
      tw.s1[r] *= rng_pow2(e_mid - e_hid); tw.b1[r] *= rng_pow2(-e_hid);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }

  h8 R0[6][2], R1[6][2];
  f32x4 acc[4];
  OhPend pe;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    pe.mid[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    pe.vox[m] = PIPE_OOB;
    pe.ovx[m] = PIPE_OOB;
  }
  for (int stage = 0;; ++stage) {
    const unsigned bufoff = (stage & 1) ? (unsigned)PIPE_BUF_BYTES : 0u;
    unsigned ad[3][2][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          ad[kw][jh][q] = ad0[kw][jh][q] + bufoff;
          asm volatile("" : "+v"(ad[kw][jh][q]));
        }
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.probe) ts0 = __builtin_readcyclecounter();
    oh_read_group<0, 0>(c.lds3, ad, R0);
    const int itemn = item + nslots;
    const bool has_next = itemn < it_end;
    PipeTile tn = t;
    if (has_next) tn = pipe_decode(a, p, itemn);
    pipe_lane_offsets(a, tn.w0, lane, c.dm.voff);
    c.dm.b = tn.b; c.dm.d0 = tn.d0; c.dm.h0 = tn.h0; c.dm.wbase = tn.w0 > 0 ? tn.w0 - 1 : 0;
    c.dm.ch = 0; c.dm.ldsbuf = (unsigned)PIPE_BUF_BYTES - bufoff; c.dm.live = has_next;
    c.dm_pitch = (unsigned)(a.W * KC) * 4u;
    c.dm_base = (unsigned)(((((tn.b * a.D + tn.d0 - 1) * a.H + tn.h0 - 1) * a.W + c.dm.wbase) * KC) * 4);

    oh_step<0>(a, c, ad, wh, wl, R0, R1, acc);

    if (a.probe) ts1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();                   // next tile's halo landed; this stage's buffer is free
    if (a.probe) ts2 = __builtin_readcyclecounter();

    // tail of this tile: the four groups' chains are independent and interleave in one scheduling region
    {
      const int od = t.d0 + wave, ow = t.w0 + (i & 7);
      const bool okdw = od < a.D && ow < a.W;
      const int obase = t.b * tail.sb + t.d0 * tail.sd + t.h0 * tail.sh + t.w0 * tail.sw;      // wave-uniform part of the grid offsets
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pe.mid[m][r] = fmaxf(fmaf(acc[m][r], sc[r], bi[r]), 0.f);
        const int oh = t.h0 + m + 4 * (i >> 3);                       // group m = output rows {m, m + 4}
        const unsigned vx = (unsigned)(((t.b * a.D + od) * a.H + oh) * a.W + ow);
        const bool okv = okdw & (oh < a.H);
        pe.vox[m] = okv ? vx : PIPE_OOB;
        pe.ovx[m] = okv ? (unsigned)(obase + lane_ovx + m * tail.sh) : PIPE_OOB;
      }
      f32x4 L[4][2];
      oh_tail_a(tw, pe, L);
      oh_tail_b<0, LOGITS>(c, tail, pe, L[0]); oh_tail_b<1, LOGITS>(c, tail, pe, L[1]);
      oh_tail_b<2, LOGITS>(c, tail, pe, L[2]); oh_tail_b<3, LOGITS>(c, tail, pe, L[3]);
    }
    if (a.probe && lane == 0 && stage < 16) {   // {stage start, taps done, barrier passed, tail set up} (tools/probe_conv_pipe.py)
      long long* pp = a.probe + (((size_t)blockIdx.x * 8 + wave) * 16 + stage) * 4;
      pp[0] = ts0; pp[1] = ts1; pp[2] = ts2; pp[3] = __builtin_readcyclecounter();
    }
    if (!has_next) break;
    t = tn; item = itemn;
  }
}

// x: (B, D, H, W, 32) channels-last in h2 storage; wpk: preworld_amd.ops.pack_occ_weight_h2 (55 296 bytes); scale / bias [16]:
// folded BN of the conv with the weights' per-channel pre-scale divided out; tailpk: preworld_amd.ops.pack_occ_tail_h2 (800
// floats: six 64-lane fp16x4 fragments of the two 1x1x1 layers, then s1 / b1 of the 8 hidden channels padded to 16), inv2 = 1 /
// the pre-scale of the last layer (applied to the logits output only; argmax does not need it).  Outputs as pw_occ_head_fused.
PW_API int pw_occ_head_h2(const float* x, const float* wpk, const float* scale, const float* bias, const float* tailpk,
                          float inv2, uint8_t* occ, float* logits, uint8_t* geo, int empty_idx, int B, int D, int H, int W,
                          int Cin, int n_mid, int n_hid, int n_cls, const int32_t* x_rng, float mid_a, float mid_b, float hid_a,
                          float hid_b, void* stream) {
  return pw_occ_head_h2_strided(x, wpk, scale, bias, tailpk, inv2, occ, logits, geo, nullptr, 0, empty_idx, B, D, H, W, Cin, n_mid, n_hid,
                                n_cls, x_rng, mid_a, mid_b, hid_a, hid_b, stream);
}

// out_strides4_host: byte strides of occ AND geo along (b, d, h, w) -- e.g. (2 XYZ, 1, Z, Y Z) writes state b's (Z, Y, X) result as
// the (X, Y, Z)-contiguous grid the reference hands out, straight into row 2 b of a (n, 2, X, Y, Z) payload buffer -- or NULL =
// contiguous (B, D, H, W); out_span_bytes: bytes from occ / geo to the end of the buffer they point into (bounds of the stores).
PW_API int pw_occ_head_h2_strided(const float* x, const float* wpk, const float* scale, const float* bias, const float* tailpk,
                                  float inv2, uint8_t* occ, float* logits, uint8_t* geo, const int64_t* out_strides4_host,
                                  int64_t out_span_bytes, int empty_idx, int B, int D, int H, int W, int Cin, int n_mid, int n_hid,
                                  int n_cls, const int32_t* x_rng, float mid_a, float mid_b, float hid_a, float hid_b, void* stream) {
  PW_CHECK_ARG(x && wpk && scale && bias && tailpk && occ, "pw_occ_head_h2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pw_occ_head_h2: bad shape");
  if (Cin != KC || n_mid != 16 || n_hid != 8 || n_cls != 18) {
    pw_set_error("pw_occ_head_h2: only the PreWorld head shape 32 -> 16/8/18 is built (got %d -> %d/%d/%d)", Cin, n_mid, n_hid,
                 n_cls);
    return PW_EUNSUP;
  }
  PW_CHECK_ARG((size_t)B * D * H * W * KC * 4 < (1ull << 32), "pw_occ_head_h2: input must stay below 4 GiB");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)logits) & 15) == 0 && ((uintptr_t)tailpk & 7) == 0,
               "pw_occ_head_h2: x / wpk / logits must be 16-byte aligned, tailpk 8-byte aligned");
  ConvArgs a = {};
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = 16; a.cout0 = n_mid; a.relu0 = 1;
  a.x_rng = x_rng;
  PW_CHECK_ARG(mid_a >= 0.f && mid_b >= 0.f && hid_a >= 0.f && hid_b >= 0.f, "pw_occ_head_h2: magnitude bounds must be >= 0");
  const OhBounds bd = {mid_a, mid_b, hid_a, hid_b};
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  PW_CHECK_ARG(nblk < (1ll << 20), "pw_occ_head_h2: too many tiles");
  OccTail t = {nullptr, nullptr, nullptr, nullptr, occ, logits, geo, empty_idx, n_mid, n_hid, n_cls, D * H * W, H * W, W, 1,
               (unsigned)((size_t)B * D * H * W)};
  if (out_strides4_host) {
    const int64_t* q = out_strides4_host;
    const int64_t last = (B - 1) * q[0] + (D - 1) * q[1] + (H - 1) * q[2] + (W - 1) * q[3];
    PW_CHECK_ARG(q[0] >= 0 && q[1] > 0 && q[2] > 0 && q[3] > 0 && last < out_span_bytes && out_span_bytes < (1ll << 31),
                 "pw_occ_head_h2_strided: strides must be positive and stay inside out_span_bytes (< 2 GiB)");
    // no two voxels may share a byte: sorted by stride, every axis must step over the whole extent of the axes below it
    // (batch stride 0 with B > 1, or any interleaving that overlaps, would be a write race)
    {
      int64_t st[4] = {q[0], q[1], q[2], q[3]};
      int64_t ex[4] = {B, D, H, W};
      for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
          if (st[j] < st[i]) { int64_t s_ = st[i]; st[i] = st[j]; st[j] = s_; s_ = ex[i]; ex[i] = ex[j]; ex[j] = s_; }
      int64_t reach = 0;                                   // last byte offset the smaller-stride axes can reach
      for (int i = 0; i < 4; ++i) {
        if (ex[i] == 1) continue;                          // a singleton axis never steps: its stride is irrelevant
        PW_CHECK_ARG(st[i] > reach, "pw_occ_head_h2_strided: output strides overlap (two voxels would store to the same byte)");
        reach += (ex[i] - 1) * st[i];
      }
    }
    t.sb = (int)q[0]; t.sd = (int)q[1]; t.sh = (int)q[2]; t.sw = (int)q[3]; t.span = (unsigned)out_span_bytes;
  }
  PipeArgs p = {};
  p.ngroups = 1; p.n_items = (int)nblk;
  p.m_ng = magic_of(1); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
  const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
  if (logits) {
    PW_CHECK_ARG((size_t)B * D * H * W * 72 < (1ull << 32), "pw_occ_head_h2: logits must stay below 4 GiB");
    static int once = set_lds_limit(k_occ_head_h2<true>, OH_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head_h2<true>, dim3(nb), dim3(256), OH_LDS, pw_stream(stream), a, p, t, tailpk, inv2, bd);
    pw_note_kernel("k_occ_head_h2<true>");
  } else {
    static int once = set_lds_limit(k_occ_head_h2<false>, OH_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head_h2<false>, dim3(nb), dim3(256), OH_LDS, pw_stream(stream), a, p, t, tailpk, inv2, bd);
    pw_note_kernel("k_occ_head_h2<false>");
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
