// Fused OccHead (A11) on the fp16 matrix cores, "stacked" form -- entry point pw_occ_head_s (round 6).
//   mmdet3d/models/heads/occupancy_head.py:92-99,124-177: conv3x3x3 32->16 (no bias) + BN + ReLU, 1x1x1 16->8 + BN + ReLU,
//   1x1x1 8->18, argmax -> uint8; geo_occ of detectors/preworld_temporal_traj.py:313-319 from the same kernel.
//
// Why a second form.  k_occ_head_h2 (pw_occ_head_h2.hip) uses v_mfma_f32_16x16x32_f16 (M = 16 = the conv's output channels, three
// instructions per tap and 16 voxels).  Its stage is 10.6 k cycles for 5.2 k of MFMA (profiles/r06_stage_probes.txt): a 4-pass MFMA
// blocks the single wave's in-order issue for its 16 cycles, so every fragment read, DMA instruction and the whole 16->8->18 tail
// (2.7 k cycles, a phase of its own) ADD to the matrix time; two waves per SIMD would hide them but do not fit the register file.
// This kernel keeps one wave per SIMD and changes the instruction instead: v_mfma_f32_32x32x16_f16 (8 passes, 32 cycles -- its second
// half leaves issue slots, which is how k_conv3d_h2 hides its side work) with the hi and the lo plane of the weights STACKED in M:
//     A[m][k] = hi(S_c w[c][k]) for m = c < 16,  lo(S_c w[c][k]) for m = 16 + c          (one 1 KB piece per tap and k-step)
//     D[m][voxel] += A . X_hi  ;  D[m][voxel] += A . X_lo                                  (2 instructions per k-step)
// rows c and 16 + c of D then hold w_hi.(x_hi + x_lo) and w_lo.(x_hi + x_lo): their sum is the full product including the lo.lo term the
// three-product form drops.  4 instead of 3 matrix cycles per voxel and tap (6.9 k per stage), but the cycles overlap the side work.
// In the 32x32 accumulator layout rows c and 16 + c are registers i and i + 8 of the SAME lane, so the sum is 8 adds per voxel tile.
//
// Structure = k_conv3d_h2's: one persistent 4-wave block per CU over an XCD-local range of 4x8x8 tiles, the 6x10x10 halo of the next
// tile landing in the second LDS buffer by `buffer_load ... lds` during this tile's taps, a lane = one voxel (l & 31 -> 4x8 patch
// position, conflict-free fragment reads as in pw_conv3d_common.h) and a k-half (l >> 5); a wave = one d-slice = two 32-voxel tiles.
// All 27 x 2 weight pieces stay in registers (216 VGPRs).  Tail on the matrix cores in registers: after BN + ReLU a lane holds mid
// channels {4h..4h+3, 8+4h..8+4h+3} (h = k-half) of its voxel = its 8 k-values of a B operand under a permuted k order, so
//     hid = [W1_hi; W1_lo] . mid     (2 instructions, rows r and 8 + r summed)      logits = W2 . hid  (3 instructions, k padded to 16)
// and the 18 logits of a voxel end up in its two lanes (classes 4h+{0..3}, 8+4h+{0..3}, and 16, 17 in h = 0): argmax = local scan +
// one v_permlane32_swap, ties -> lowest class like torch.argmax.
#include "pw_h2.h"
#include "pw_occ_tail.h"

namespace {
constexpr int OS_LDS = 2 * PIPE_BUF_BYTES;                       // 153 600
constexpr int OS_WPK_BYTES = 27 * 2 * 1024;                      // packed conv weights [tap][k-step][lane][8 halfs]
constexpr int OS_TAILPK_BYTES = 800 * 4;                         // tail operands (pack_occ_tail_s)
constexpr int OS_DMA_TAP0 = 2;                                   // halo row K of the next tile is issued in tap OS_DMA_TAP0 + K
}  // namespace

struct OsDmaView { rsrc_t xr; lds3_t lds3; unsigned base, pitch; int wave; };

// ---- halo DMA by table (round 6).  h2_dma_row / os_dma_row spend ~26 scalar + 5 vector instructions per halo row on (d, h) of the row,
// its bounds, its scalar offset and the lane-offset selects -- 15 rows per wave and stage, 1.9 k cycles of a 9.4 k-cycle tap loop here
// (ablation build OS_X_NODMA).  The geometry of a halo tile relative to its origin never changes, so:
//   * OsDmaTab::loff[I] (per kernel, 30 VGPRs): the byte offset of DMA instruction I's lane relative to voxel (d0 - 1, h0 - 1, w0 - 1) of the
//     tile -- row offset + swizzled slot, everything pipe_lane_offsets + the row arithmetic produced;
//   * per stage: ONE buffer descriptor whose base is that voxel of the next tile (64-bit pointer arithmetic, may point in front of the
//     tensor: such lanes are masked), a 15-bit mask of this wave's rows inside the volume (one ballot over lanes = rows), two lane masks
//     of the columns outside it (w boundary tiles), the LDS base of this wave's rows;
//   * per instruction: scalar pick of the lane mask (row outside -> all lanes), one v_cndmask to PIPE_OOB, the DMA: ~4 scalar + 1 vector.
struct OsDmaTab {
  unsigned loff[2 * PIPE_ROWS_PER_WAVE];
  unsigned oob;                 // PIPE_OOB in a VGPR (v_cndmask operand)
  int rowinfo;                  // lane K < 15: dd | hh << 8 of row wave + 4 K; other lanes: never valid
};
struct OsDmaStage {
  rsrc_t xr;                    // base = voxel (d0 - 1, h0 - 1, w0 - 1) of the next tile
  unsigned long long rowmask;   // bit K: halo row wave + 4 K lies inside the volume (0 without a next tile)
  unsigned long long wbad[2];   // lanes of pass 0 / 1 whose column lies outside the volume
  lds3_t dst;                   // LDS address of halo row `wave` in the buffer being filled
};

__device__ __forceinline__ void os_dma_tab_init(const ConvArgs& a, int wave, int lane, OsDmaTab& tb) {
  const unsigned pitch = (unsigned)(a.W * KC) * 4u;
#pragma unroll
  for (int K = 0; K < PIPE_ROWS_PER_WAVE; ++K) {
    const int row = wave + 4 * K;
    const int dd = (row * 205) >> 11, hh = row - dd * TH;          // row / 10 for row < 60
    const int par = hh & 1;
    const unsigned rowoff = (unsigned)(dd * a.H + hh) * pitch;
    {
      const int ww = lane >> 3, slot = lane & 7;
      const int f = ((ww >> 1) & 3) | (par << 2);
      tb.loff[2 * K] = rowoff + (unsigned)(ww * KC + (slot ^ f) * 4) * 4u;
    }
    {
      const int ww = 8 + (lane >> 5), dw = lane & 31, slot = dw >> 2;
      const int f = ((ww >> 1) & 3) | (par << 2);
      tb.loff[2 * K + 1] = rowoff + (unsigned)(ww * KC + (slot ^ f) * 4 + (dw & 3)) * 4u;
    }
  }
  tb.oob = PIPE_OOB;
  asm volatile("" : "+v"(tb.oob));
  const int row = wave + 4 * lane;
  const int dd = (row * 205) >> 11, hh = row - dd * TH;
  tb.rowinfo = lane < PIPE_ROWS_PER_WAVE ? (dd | (hh << 8)) : 0x7f7f;
}

__device__ __forceinline__ void os_dma_stage(const ConvArgs& a, const OsDmaTab& tb, const PipeTile& tn, bool live, lds3_t lds3, unsigned ldsbuf,
                                             int wave, int lane, OsDmaStage& st) {
  const long long toff = ((((long long)tn.b * a.D + (tn.d0 - 1)) * a.H + (tn.h0 - 1)) * a.W + (tn.w0 - 1)) * (long long)(KC * 4);
  st.xr = make_rsrc(reinterpret_cast<const char*>(a.x) + toff, 0x80000000u);
  const int dd = tb.rowinfo & 0xff, hh = tb.rowinfo >> 8;
  const bool rv = (int)live & (int)((unsigned)(tn.d0 - 1 + dd) < (unsigned)a.D) & (int)((unsigned)(tn.h0 - 1 + hh) < (unsigned)a.H);
  st.rowmask = __builtin_amdgcn_ballot_w64(rv);
  st.wbad[0] = __builtin_amdgcn_ballot_w64(!((unsigned)(tn.w0 - 1 + (lane >> 3)) < (unsigned)a.W));
  st.wbad[1] = __builtin_amdgcn_ballot_w64(!((unsigned)(tn.w0 - 1 + 8 + (lane >> 5)) < (unsigned)a.W));
  st.dst = lds3 + (ldsbuf + (unsigned)wave * (TW * 128));
}

template <int K>
__device__ __forceinline__ void os_dma_row_tab(const OsDmaTab& tb, const OsDmaStage& st) {
  constexpr unsigned imm = (unsigned)((4 * K) * (TW * 128));         // halo row wave + 4 K
  const bool rv = (st.rowmask >> K) & 1ull;
  const unsigned long long bad0 = rv ? st.wbad[0] : ~0ull, bad1 = rv ? st.wbad[1] : ~0ull;
  unsigned v0, v1;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v0) : "v"(tb.loff[2 * K]), "v"(tb.oob), "s"(bad0));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v1) : "v"(tb.loff[2 * K + 1]), "v"(tb.oob), "s"(bad1));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xr, st.dst + imm, 16, v0, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xr, st.dst + imm + 1024, 4, v1, 0, 0, 0);
}

// halo row `wave + 4 K` of the next tile (h2_dma_row of pw_conv3d_h2.hip: per-stage address arithmetic hoisted, branch-free)
template <int K>
__device__ __forceinline__ void os_dma_row(const ConvArgs& a, const PipeDma& dm, const OsDmaView& c) {
  constexpr int c1 = (4 * K) / TH, c2 = (4 * K) % TH;
  const int t = c2 + c.wave;                               // wave-uniform
  const int carry = t >= TH ? 1 : 0;
  const int dd = c1 + carry, hh = t - TH * carry;
  const bool rok = (int)dm.live & (int)((unsigned)(dm.d0 - 1 + dd) < (unsigned)a.D) & (int)((unsigned)(dm.h0 - 1 + hh) < (unsigned)a.H);
  const unsigned soff = c.base + (unsigned)(dd * a.H + hh) * c.pitch;      // rows outside the volume: every lane is out of range
  const unsigned v0 = rok ? ((hh & 1) ? dm.voff[1][0] : dm.voff[0][0]) : PIPE_OOB;
  const unsigned v1 = rok ? ((hh & 1) ? dm.voff[1][1] : dm.voff[0][1]) : PIPE_OOB;
  lds3_t dst = c.lds3 + (dm.ldsbuf + (unsigned)(c1 * TH + c2) * (TW * 128) + (unsigned)c.wave * (TW * 128));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, dst, 16, v0, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, dst + 1024, 4, v1, soff, 0, 0);
}

// activation fragments of one tap: aq[mt][2 ks + p] (k-step, plane) of the lane's voxel in tile mt (rows 0-3 / 4-7 of the patch)
template <int TAP>
__device__ __forceinline__ void os_read_tap(lds3_t lds3, const unsigned (&aaddr)[2][3][4], v4f (&aq)[2][4]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    constexpr unsigned imm0 = (unsigned)(((kd * TH + kh) * TW) * 128);
    const unsigned imm = imm0 + (unsigned)(mt * 4 * TW * 128);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      aq[mt][q] = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(lds3 + aaddr[kh & 1][kw][q] + imm);
  }
}

__device__ __forceinline__ constexpr int os_share(int i, int slots, int n) { return ((i + 1) * n) / slots - (i * n) / slots; }
template <int I, int S, int NVM, int NDS, int NSA, int NVA>
__device__ __forceinline__ void os_pipeline() {
  if constexpr (I < S) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (os_share(I, S, NDS) > 0) __builtin_amdgcn_sched_group_barrier(0x100, os_share(I, S, NDS), 0);
    if constexpr (os_share(I, S, NVM) > 0) __builtin_amdgcn_sched_group_barrier(0x020, os_share(I, S, NVM), 0);
    if constexpr (os_share(I, S, NSA) > 0) __builtin_amdgcn_sched_group_barrier(0x004, os_share(I, S, NSA), 0);
    if constexpr (os_share(I, S, NVA) > 0) __builtin_amdgcn_sched_group_barrier(0x002, os_share(I, S, NVA), 0);
    os_pipeline<I + 1, S, NVM, NDS, NSA, NVA>();
  }
}

struct OsCtx {
  lds3_t lds3;
  rsrc_t xr, occr, geor, lgr;
  OsDmaStage ds;
  int wave, half;
  float inv2;
};

struct OsTailW {                 // per-lane resident operands of the tail (ops.pack_occ_tail_s)
  h8 a1, a2h, a2l;               // [W1_hi; W1_lo] (rows 0-7 / 8-15), W2_hi, W2_lo (rows 0-17) as 32x16 A fragments
  float sc[8], bi[8];            // folded BN of the conv for this lane's mid channels ch(i) = 8 (i >> 2) + 4 h + (i & 3)
  float s1[4], b1[4];            // folded BN of hid channels 4 h + r (divided by W1's pre-scale)
};

// (best, idx) against the partner lane's: the larger logit, on a tie the lower class (torch.argmax)
__device__ __forceinline__ void os_pick(float& best, int& idx, float ob, int oi) {
  const bool take = (ob > best) | ((ob == best) & (oi < idx));
  best = take ? ob : best;
  idx = take ? oi : idx;
}
__device__ __forceinline__ void os_scan(float& best, int& idx, float v, int cls) {   // classes visited in ascending order
  const bool take = v > best;
  best = take ? v : best;
  idx = take ? cls : idx;
}

// the tail of one 32-voxel tile: raw conv sums (rows c and 16 + c already added) -> occ / geo bytes (and logits)
template <bool LOGITS>
__device__ __forceinline__ void os_tail(const OsCtx& c, const OccTail& tail, const OsTailW& tw, const float (&raw)[8], unsigned vox,
                                        unsigned ovx) {
  float mid[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) mid[i] = fmaxf(fmaf(raw[i], tw.sc[i], tw.bi[i]), 0.f);
  h8 mh, ml;
  h2_split8(mid, mh, ml);
  f32x16 ha = {};
  ha = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw.a1, mh, ha, 0, 0, 0);
  ha = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw.a1, ml, ha, 0, 0, 0);
  float hid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) hid[r] = fmaxf(fmaf(ha[r] + ha[4 + r], tw.s1[r], tw.b1[r]), 0.f);
  u2 hh2, hl2;
  h2_split4(hid, hh2, hl2);
  v4f hhv = {__uint_as_float(hh2[0]), __uint_as_float(hh2[1]), 0.f, 0.f}, hlv = {__uint_as_float(hl2[0]), __uint_as_float(hl2[1]), 0.f, 0.f};
  const h8 hh = __builtin_bit_cast(h8, hhv), hl = __builtin_bit_cast(h8, hlv);
  f32x16 L = {};
  L = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw.a2h, hh, L, 0, 0, 0);
  L = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw.a2l, hh, L, 0, 0, 0);
  L = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw.a2h, hl, L, 0, 0, 0);
  const int h = c.half;
  float best = L[0];
  int idx = 4 * h;
#pragma unroll
  for (int r = 1; r < 4; ++r) os_scan(best, idx, L[r], 4 * h + r);
#pragma unroll
  for (int r = 0; r < 4; ++r) os_scan(best, idx, L[4 + r], 8 + 4 * h + r);
  os_scan(best, idx, h == 0 ? L[8] : -INFINITY, 16);
  os_scan(best, idx, h == 0 ? L[9] : -INFINITY, 17);
  {
    const auto vb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto vi = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
    best = __uint_as_float(vb[0]); idx = (int)vi[0];
    os_pick(best, idx, __uint_as_float(vb[1]), (int)vi[1]);
  }
  const unsigned vb = h == 0 ? ovx : PIPE_OOB;                     // one of the voxel's two lanes writes the bytes
  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)idx, c.occr, vb, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b8(idx != tail.empty_idx ? (unsigned char)0 : (unsigned char)(tail.n_cls - 1), c.geor, vb, 0, 0);
  if constexpr (LOGITS) {
    const unsigned lo = vox == PIPE_OOB ? PIPE_OOB : vox * 72u;
    const float t0[4] = {L[0] * c.inv2, L[1] * c.inv2, L[2] * c.inv2, L[3] * c.inv2};
    const float t1[4] = {L[4] * c.inv2, L[5] * c.inv2, L[6] * c.inv2, L[7] * c.inv2};
    buf_store4(c.lgr, lo == PIPE_OOB ? PIPE_OOB : lo + 16u * (unsigned)h, 0, t0);
    buf_store4(c.lgr, lo == PIPE_OOB ? PIPE_OOB : lo + 32u + 16u * (unsigned)h, 0, t1);
    u2 t2;
    t2[0] = __float_as_uint(L[8] * c.inv2); t2[1] = __float_as_uint(L[9] * c.inv2);
    buf_store2(c.lgr, (h == 0 && lo != PIPE_OOB) ? lo + 64u : PIPE_OOB, 0, t2);
  }
}

// tap TAP of a stage: fragments of tap TAP + 1 are requested, one halo row of the next tile is DMA'd, the 8 MFMAs of this tap run
template <int TAP>
__device__ __forceinline__ void os_step(const ConvArgs& a, const OsCtx& c, const OsDmaTab& tb, const unsigned (&aaddr)[2][3][4], v4f (&ac)[2][4],
                                        v4f (&an)[2][4], const v4f (&wres)[27][2], f32x16 (&acc)[2]) {
  __builtin_amdgcn_sched_barrier(0);
  constexpr int K = TAP - OS_DMA_TAP0;
  constexpr bool dma = K >= 0 && K < PIPE_ROWS_PER_WAVE;
  // (OS_X_*: timing-only ablation builds via tools/build_variant.py -- wrong results, never in libpreworld_hip.so)
#ifndef OS_X_NODMA
  if constexpr (dma) os_dma_row_tab<K>(tb, c.ds);
#endif
#ifndef OS_X_NOREAD
  if constexpr (TAP < 26) os_read_tap<TAP + 1>(c.lds3, aaddr, an);
#else
  if constexpr (TAP < 26) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) an[mt][q] = ac[mt][q];
  }
#endif
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wres[TAP][ks]), __builtin_bit_cast(h8, ac[mt][2 * ks + px]),
                                                        acc[mt], 0, 0, 0);
  os_pipeline<0, 8, dma ? 2 : 0, TAP < 26 ? 8 : 0, dma ? 6 : 2, dma ? 2 : 0>();
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP < 26) os_step<TAP + 1>(a, c, tb, aaddr, an, ac, wres, acc);
}

struct OsBounds { float mid_a, mid_b, hid_a, hid_b; };

template <bool LOGITS>
__global__ void __launch_bounds__(256, 1) k_occ_head_s(ConvArgs a, PipeArgs p, OccTail tail, const float* tailpk, float inv2, OsBounds bd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int pj = patch_of_row(j), pr = pj >> 3, pc = pj & 7;
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;

  OsCtx c;
  c.lds3 = (lds3_t)lds;
  const unsigned nvox = (unsigned)((size_t)a.B * a.D * a.H * a.W);
  c.xr = make_rsrc(a.x, nvox * (unsigned)(KC * 4));
  c.occr = make_rsrc(tail.occ, tail.span);
  c.geor = make_rsrc(tail.geo, tail.geo ? tail.span : 0u);
  c.lgr = make_rsrc(tail.logits, tail.logits ? nvox * 72u : 0u);
  c.wave = wave; c.half = half;
  const int e_in = rng_exp(a.x_rng);

  unsigned aaddr0[2][3][4];
#pragma unroll
  for (int khp = 0; khp < 2; ++khp)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = pc + kw;
      const int f = ((ww >> 1) & 3) | (((pr + khp) & 1) << 2);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        aaddr0[khp][kw][q] = (unsigned)((((wave * TH + pr) * TW + ww) * 8 + ((half * 4 + q) ^ f)) * 16);
    }

  // conv weights: 27 taps x 2 k-steps of [hi rows; lo rows], resident
  v4f wres[27][2];
  {
    const rsrc_t wr = make_rsrc(a.wpk, (unsigned)OS_WPK_BYTES);
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)lane * 16u, (unsigned)((2 * t + ks) * 1024), 0);
        v4f o;
        o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
        wres[t][ks] = o;
      }
  }
  OsTailW tw;
  {
    const rsrc_t tr = make_rsrc(tailpk, (unsigned)OS_TAILPK_BYTES);
    typedef unsigned bu4 __attribute__((ext_vector_type(4)));
    const auto f0 = __builtin_amdgcn_raw_buffer_load_b128(tr, (unsigned)lane * 16u, 0u, 0);
    const auto f1 = __builtin_amdgcn_raw_buffer_load_b128(tr, (unsigned)lane * 16u, 1024u, 0);
    const auto f2 = __builtin_amdgcn_raw_buffer_load_b128(tr, (unsigned)lane * 16u, 2048u, 0);
    bu4 u0, u1, u2_;
#pragma unroll
    for (int e = 0; e < 4; ++e) { u0[e] = f0[e]; u1[e] = f1[e]; u2_[e] = f2[e]; }
    tw.a1 = __builtin_bit_cast(h8, u0); tw.a2h = __builtin_bit_cast(h8, u1); tw.a2l = __builtin_bit_cast(h8, u2_);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tw.s1[r] = tailpk[768 + 4 * half + r];
      tw.b1[r] = tailpk[784 + 4 * half + r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
      tw.sc[i] = a.scale[ch];
      tw.bi[i] = a.bias[ch];
    }
  }

  OsDmaTab tb;
  os_dma_tab_init(a, wave, lane, tb);
  PipeTile t = pipe_decode(a, p, item);
  {
    PipeDma dm;
    pipe_lane_offsets(a, t.w0, lane, dm.voff);
    dm.b = t.b; dm.d0 = t.d0; dm.h0 = t.h0; dm.wbase = t.w0 > 0 ? t.w0 - 1 : 0; dm.ch = 0; dm.ldsbuf = 0; dm.live = true;
    pipe_dma_row<0>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<1>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<2>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<3>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<4>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<5>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<6>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<7>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<8>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<9>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<10>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<11>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<12>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<13>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<14>(a, c.xr, c.lds3, dm, wave);
    // Range (pw_h2.h "Range"; same scheme as k_occ_head_h2): x is read under its slot's exponent; the two hidden layers are split
    // in registers in units chosen from a-priori bounds (each bound at 2^15), powers of two folded into the BN constants
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
    for (int i = 0; i < 8; ++i) { tw.sc[i] *= rng_pow2(e_in - e_mid); tw.bi[i] *= rng_pow2(-e_mid); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { tw.s1[r] *= rng_pow2(e_mid - e_hid); tw.b1[r] *= rng_pow2(-e_hid); }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }

  v4f a0[2][4], a1[2][4];
  f32x16 acc[2];
  for (int stage = 0;; ++stage) {
    const unsigned bufoff = (stage & 1) ? (unsigned)PIPE_BUF_BYTES : 0u;
    unsigned aaddr[2][3][4];
#pragma unroll
    for (int khp = 0; khp < 2; ++khp)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          aaddr[khp][kw][q] = aaddr0[khp][kw][q] + bufoff;
          asm volatile("" : "+v"(aaddr[khp][kw][q]));
        }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.probe) ts0 = __builtin_readcyclecounter();
    os_read_tap<0>(c.lds3, aaddr, a0);
    const int itemn = item + nslots;
    const bool has_next = itemn < it_end;
    PipeTile tn = t;
    if (has_next) tn = pipe_decode(a, p, itemn);
    os_dma_stage(a, tb, tn, has_next, c.lds3, (unsigned)PIPE_BUF_BYTES - bufoff, wave, lane, c.ds);

    os_step<0>(a, c, tb, aaddr, a0, a1, wres, acc);

    if (a.probe) ts1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();                   // next tile's halo landed; this stage's buffer is free
    if (a.probe) ts2 = __builtin_readcyclecounter();

    {
      const int od = t.d0 + wave, ow = t.w0 + pc;
      const bool okdw = od < a.D && ow < a.W;
      const int obase = t.b * tail.sb + od * tail.sd + ow * tail.sw;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int oh = t.h0 + pr + 4 * mt;
        const bool okv = okdw & (oh < a.H);
        const unsigned vx = okv ? (unsigned)(((t.b * a.D + od) * a.H + oh) * a.W + ow) : PIPE_OOB;
        const unsigned ovx = okv ? (unsigned)(obase + oh * tail.sh) : PIPE_OOB;
        float raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = acc[mt][i] + acc[mt][i + 8];
#ifndef OS_X_NOTAIL
        os_tail<LOGITS>(c, tail, tw, raw, vx, ovx);
#else
        if (raw[0] == 1234.5f) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)1, c.occr, ovx, 0, 0);
#endif
      }
    }
    if (a.probe && lane == 0 && stage < 16) {   // {stage start, taps done, barrier passed, tail done} (tools/probe_conv_pipe.py)
      long long* pp = a.probe + (((size_t)blockIdx.x * 8 + wave) * 16 + stage) * 4;
      pp[0] = ts0; pp[1] = ts1; pp[2] = ts2; pp[3] = __builtin_readcyclecounter();
    }
    if (!has_next) break;
    t = tn; item = itemn;
  }
}

// x: (B, D, H, W, 32) channels-last in h2 storage; wpk: preworld_amd.ops.pack_occ_weight_s (55 296 bytes); scale / bias [16]: folded BN of
// the conv with the weights' per-channel pre-scale divided out; tailpk: preworld_amd.ops.pack_occ_tail_s (800 floats); inv2 = 1 / the
// pre-scale of the last layer (logits output only).  out_strides4_host / out_span_bytes as pw_occ_head_h2_strided (NULL = contiguous).
PW_API int pw_occ_head_s(const float* x, const float* wpk, const float* scale, const float* bias, const float* tailpk, float inv2,
                         uint8_t* occ, float* logits, uint8_t* geo, const int64_t* out_strides4_host, int64_t out_span_bytes, int empty_idx,
                         int B, int D, int H, int W, int Cin, int n_mid, int n_hid, int n_cls, const int32_t* x_rng, float mid_a, float mid_b,
                         float hid_a, float hid_b, void* stream) {
  PW_CHECK_ARG(x && wpk && scale && bias && tailpk && occ, "pw_occ_head_s: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pw_occ_head_s: bad shape");
  if (Cin != KC || n_mid != 16 || n_hid != 8 || n_cls != 18) {
    pw_set_error("pw_occ_head_s: only the PreWorld head shape 32 -> 16/8/18 is built (got %d -> %d/%d/%d)", Cin, n_mid, n_hid, n_cls);
    return PW_EUNSUP;
  }
  PW_CHECK_ARG((size_t)B * D * H * W * KC * 4 < (1ull << 32), "pw_occ_head_s: input must stay below 4 GiB");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)logits | (uintptr_t)tailpk) & 15) == 0,
               "pw_occ_head_s: x / wpk / tailpk / logits must be 16-byte aligned");
  ConvArgs a = {};
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = 32; a.cout0 = n_mid; a.relu0 = 1;
  a.x_rng = x_rng;
  PW_CHECK_ARG(mid_a >= 0.f && mid_b >= 0.f && hid_a >= 0.f && hid_b >= 0.f, "pw_occ_head_s: magnitude bounds must be >= 0");
  const OsBounds bd = {mid_a, mid_b, hid_a, hid_b};
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  PW_CHECK_ARG(nblk < (1ll << 20), "pw_occ_head_s: too many tiles");
  OccTail t = {nullptr, nullptr, nullptr, nullptr, occ, logits, geo, empty_idx, n_mid, n_hid, n_cls, D * H * W, H * W, W, 1,
               (unsigned)((size_t)B * D * H * W)};
  if (out_strides4_host) {
    const int64_t* q = out_strides4_host;
    const int64_t last = (B - 1) * q[0] + (D - 1) * q[1] + (H - 1) * q[2] + (W - 1) * q[3];
    PW_CHECK_ARG(q[0] >= 0 && q[1] > 0 && q[2] > 0 && q[3] > 0 && last < out_span_bytes && out_span_bytes < (1ll << 31),
                 "pw_occ_head_s: strides must be positive and stay inside out_span_bytes (< 2 GiB)");
    {
      int64_t st[4] = {q[0], q[1], q[2], q[3]};
      int64_t ex[4] = {B, D, H, W};
      for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j)
          if (st[j] < st[i]) { int64_t s_ = st[i]; st[i] = st[j]; st[j] = s_; s_ = ex[i]; ex[i] = ex[j]; ex[j] = s_; }
      int64_t reach = 0;
      for (int i = 0; i < 4; ++i) {
        if (ex[i] == 1) continue;
        PW_CHECK_ARG(st[i] > reach, "pw_occ_head_s: output strides overlap (two voxels would store to the same byte)");
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
    PW_CHECK_ARG((size_t)B * D * H * W * 72 < (1ull << 32), "pw_occ_head_s: logits must stay below 4 GiB");
    static int once = set_lds_limit(k_occ_head_s<true>, OS_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head_s<true>, dim3(nb), dim3(256), OS_LDS, pw_stream(stream), a, p, t, tailpk, inv2, bd);
    pw_note_kernel("k_occ_head_s<true>");
  } else {
    static int once = set_lds_limit(k_occ_head_s<false>, OS_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head_s<false>, dim3(nb), dim3(256), OS_LDS, pw_stream(stream), a, p, t, tailpk, inv2, bd);
    pw_note_kernel("k_occ_head_s<false>");
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
