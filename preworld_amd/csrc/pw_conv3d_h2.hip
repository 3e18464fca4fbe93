// 3x3x3 stride-1 convolution of the voxel encoder on the fp16 matrix cores with split-fp16 operands
// (pw_h2.h: x = hi + lo, three v_mfma_f32_32x32x16_f16 per product block, fp32 accumulate -> fp32-level accuracy at
// 5.3x the throughput of v_mfma_f32_32x32x2_f32).  Same role as pw_conv3d_ndhwc / pw_conv3d_wino:
//   mmdet3d/models/backbones/resnet.py:88-184 (BasicBlock3D / CustomResNet3D), detectors/preworld.py:72-79 (final_conv).
//
// Structure = the persistent DMA-pipelined kernel of pw_conv3d.hip (one 4-wave block per CU walking an XCD-local
// range of (4x8x8 tile, N-group, 32-channel chunk) stages; the 6x10x10 halo of stage s+1 lands in the second LDS
// buffer by `buffer_load ... lds` while stage s computes; weights two taps ahead, A fragments one tap ahead), with
//   * input in h2 storage: the 128 bytes of a voxel chunk are staged as they are; a lane's four ds_read_b128 of a tap
//     are {hi, lo} x {k-step 0, 1} of its voxel;
//   * the GEMM TRANSPOSED: D[cout][voxel] = W[cout][k] X[k][voxel] (A operand = packed weights, B = activations), so
//     a lane ends up with 16 output channels of ONE voxel in groups of 4 consecutive channels: the epilogue stores
//     16-byte (fp32) or 8-byte hi + 8-byte lo (h2) pieces instead of 16 scattered dwords, and its bounds test is one
//     predicate per lane;
//   * per tap and (M-tile, N-tile): 6 MFMAs of 32 cycles (hi.hi, lo_w.hi_x, hi_w.lo_x for both k-steps);
//   * folded-BN scale / bias of all output columns parked in LDS once per block.
// Weights: preworld_amd.ops.pack_conv_weight_h2 (per-output-channel power-of-two pre-scale, undone through `scale`).
#include "pw_h2.h"

namespace {
constexpr int H2_SB_OFF = 2 * PIPE_BUF_BYTES;        // scale/bias area behind the two halo buffers
constexpr int H2_MAX_COUT = 256;
constexpr int H2_LDS = H2_SB_OFF + 2 * H2_MAX_COUT * 4;
}  // namespace

// ---- software-pipelined epilogue of the persistent kernel (EPI > 0).
// The tile that finished in stage s is written out DURING stage s+1.  At the end of stage s every wave only parks its raw
// accumulators in LDS (h2_epi_park: 8 ds_write_b128 per N-tile).  The 4 NT passes that turn them into
// scale / bias / residual / ReLU / fp16 split / coalesced stores are BRANCH-FREE straight-line code placed in the same
// scheduling region as the 12 NT MFMAs of taps 2 .. 4 NT + 1 of the next stage (their LDS / residual loads one tap earlier),
// so that they issue in the ~5 free slots a single wave has per 32-cycle MFMA: a separate epilogue phase cost 4.9 k of a
// 22.9 k-cycle stage, and the same passes placed before the MFMAs (branchy version) simply moved those cycles into the taps.
// Being branch-free means the variant is a template parameter:
//   EPI 1: y0 (and y1) in h2 storage, no residual      EPI 2: y0 h2 + h2 residual (no y1)      EPI 3: fp32, no residual
//   EPI 0: anything else -- the generic epilogue phase (h2_epilogue) at the end of the stage.
// A stage with nothing parked runs the same code with every lane's offset out of range (loads return 0, stores are dropped).
// Parking area = this wave's own halo rows (wave + 4 K, K < 7 / 13 for NT 1 / 2) of the buffer the stage just consumed: those
// rows are refilled only by this wave's own DMA instructions, which are issued after the passes (other rows first).
struct EpiTile { int b, d0, h0, w0, ng; unsigned bufoff; bool pending; };
struct EpiRegs {
  v4f x0, x1, s0, s1, b0, b1; float4 r0, r1; unsigned vbase, soff;
  float amax0, amax1;        // largest |stored value| written to y0 / y1 (tiles whose passes have completed)
  float amax_nt[2];          // ... by the passes of the tile in flight, per N-tile of the wave: which destination an N-tile
                             // feeds depends on the tile's N-group (wave-uniform), so it is resolved once per tile (epi_fold_amax)
};

// fold the in-flight tile's per-N-tile maxima into the per-destination ones; ng = that tile's N-group
template <int NT>
__device__ __forceinline__ void epi_fold_amax(const ConvArgs& a, int ng, EpiRegs& r) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const bool to_y0 = (ng * NT + nt) * 32 < a.cout0;
    r.amax0 = fmaxf(r.amax0, to_y0 ? r.amax_nt[nt] : 0.f);
    r.amax1 = fmaxf(r.amax1, to_y0 ? 0.f : r.amax_nt[nt]);
    r.amax_nt[nt] = 0.f;
  }
}

__device__ __forceinline__ unsigned epi_slot_off(int u) {          // parked voxel row u (128 B): 10 per halo row
  const int row = (int)(((unsigned)u * 205u) >> 11);               // u / 10 for u < 1024
  return (unsigned)(row * (4 * TW * 128) + (u - row * 10) * 128);
}

template <int NT>
__device__ __forceinline__ void h2_epi_park(const f32x16 (&acc)[2][NT], char* stg, int lane) {
  const int half = lane >> 5, pj = patch_of_row(lane & 31);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int u = nt * 64 + mt * 32 + pj;
      char* dst = stg + epi_slot_off(u);
      const int sw = (u >> 1) & 7;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v4f v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * g + e];
        *reinterpret_cast<v4f*>(dst + (((2 * g + half) ^ sw) * 16)) = v;
      }
    }
}

// pass P of 4 NT covers 16 parked voxel rows; lane = (row u = 16 P + (lane >> 2), channel octet o = lane & 3)
template <int NT, int EPI, int P>
__device__ __forceinline__ void h2_epi_load(const ConvArgs& a, const EpiTile& t, const char* stg, const float* sb, int wave,
                                            int lane, EpiRegs& r) {
  constexpr int nt = P >> 2;
  const int n0 = (t.ng * NT + nt) * 32;
  const bool to_y0 = n0 < a.cout0;
  const int ld = to_y0 ? a.ld0 : a.ld1;
  const int col0 = to_y0 ? n0 : n0 - a.n1_start;
  const int od = t.d0 + wave;
  const int o = lane & 3;
  const int u = 16 * P + (lane >> 2), sv = u & 63;
  const int mt = sv >> 5, vr = (sv >> 3) & 3, vc = sv & 7;
  // (no short-circuit: the pass lives inside a tap's scheduling region, a branch would cut that region in two)
  const bool ok = (int)t.pending & (int)(od < a.Do) & (int)((t.h0 + mt * 4 + vr) < a.Ho) & (int)((t.w0 + vc) < a.Wo);
  r.soff = (unsigned)(((((t.b * a.Do + od) * a.Ho + t.h0) * a.Wo + t.w0) * ld) * 4);
  const unsigned pos = (EPI == 3 ? (unsigned)(32 * o) : (unsigned)((4 * (o & 1) + 2 * (o >> 1)) * 16));
  const unsigned vin = (unsigned)(((mt * 4 + vr) * a.Wo + vc) * ld) * 4u + (unsigned)col0 * 4u + pos;
  r.vbase = ok ? vin : PIPE_OOB;
  const char* src = stg + epi_slot_off(u);
  const int sw = (u >> 1) & 7;
  r.x0 = *reinterpret_cast<const v4f*>(src + (((2 * o) ^ sw) * 16));
  r.x1 = *reinterpret_cast<const v4f*>(src + (((2 * o + 1) ^ sw) * 16));
  r.s0 = *reinterpret_cast<const v4f*>(sb + n0 + 8 * o);
  r.s1 = *reinterpret_cast<const v4f*>(sb + n0 + 8 * o + 4);
  r.b0 = *reinterpret_cast<const v4f*>(sb + H2_MAX_COUT + n0 + 8 * o);
  r.b1 = *reinterpret_cast<const v4f*>(sb + H2_MAX_COUT + n0 + 8 * o + 4);
  if constexpr (EPI == 2) {
    const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
    const rsrc_t rr = make_rsrc(a.residual, out_vox * (unsigned)ld * 4u);
    r.r0 = buf_load4(rr, r.vbase, r.soff);
    r.r1 = buf_load4(rr, r.vbase == PIPE_OOB ? PIPE_OOB : r.vbase + 16u, r.soff);
  }
}

template <int NT, int EPI, int P>
__device__ __forceinline__ void h2_epi_compute(const ConvArgs& a, const EpiTile& t, EpiRegs& r, float res_mul) {
  constexpr int nt = P >> 2;
  const int n0 = (t.ng * NT + nt) * 32;
  const bool to_y0 = n0 < a.cout0;
  float* dst = to_y0 ? a.y0 : a.y1;
  const int ld = to_y0 ? a.ld0 : a.ld1;
  const float lo_clamp = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
  const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
  const rsrc_t yr = make_rsrc(dst, out_vox * (unsigned)ld * 4u);
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = fmaf(r.x0[e], r.s0[e], r.b0[e]); v[4 + e] = fmaf(r.x1[e], r.s1[e], r.b1[e]); }
  if constexpr (EPI == 2) {                         // r0 = 8 hi halves, r1 = 8 lo halves of channels 8o .. 8o+7
    const h8 rh = __builtin_bit_cast(h8, r.r0), rl = __builtin_bit_cast(h8, r.r1);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf((float)rh[e] + (float)rl[e], res_mul, v[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], lo_clamp);
  {                                                 // range bookkeeping: rows that are not stored (nothing parked, outside the volume) do not count
    float m = fabsf(v[0]);
#pragma unroll
    for (int e = 1; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    r.amax_nt[nt] = fmaxf(r.amax_nt[nt], r.vbase == PIPE_OOB ? 0.f : m);       // a select, not a branch: this runs inside the MFMA scheduling region
  }
  const unsigned second = r.vbase == PIPE_OOB ? PIPE_OOB : r.vbase + 16u;
  if constexpr (EPI == 3) {
    const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
    buf_store4(yr, r.vbase, r.soff, va);
    buf_store4(yr, second, r.soff, vb);
  } else {
    h8 oh, ol;
    h2_split8(v, oh, ol);
    const v4f wh = __builtin_bit_cast(v4f, oh), wl = __builtin_bit_cast(v4f, ol);
    const float va[4] = {wh[0], wh[1], wh[2], wh[3]}, vb[4] = {wl[0], wl[1], wl[2], wl[3]};
    buf_store4(yr, r.vbase, r.soff, va);
    buf_store4(yr, second, r.soff, vb);
  }
}

template <int NT, int EPI, int P>
__device__ __forceinline__ void h2_epi_rest(const ConvArgs& a, const EpiTile& t, const char* stg, const float* sb, int wave,
                                            int lane, EpiRegs& r, float res_mul) {
  if constexpr (P < 4 * NT) {
    h2_epi_load<NT, EPI, P>(a, t, stg, sb, wave, lane, r);
    h2_epi_compute<NT, EPI, P>(a, t, r, res_mul);
    h2_epi_rest<NT, EPI, P + 1>(a, t, stg, sb, wave, lane, r, res_mul);
  }
}

// halo rows (K = first, count) this wave's DMA issues at tap TAP of a stage: a few rows per tap right after the epilogue
// passes (taps 2 .. 4 NT + 1, which read the parked rows K < 7 / 13), so that the parked rows are never refilled under a pass
// and the whole halo has the rest of the stage to land.  (One row per tap over 13-15 taps, 5 or 15 per tap: the same time within
// noise, profiles/r03_conv_h2_ablation.txt section 3.  That file's "DMA with every lane out of range" variant is 23 % faster
// only because an all-zero halo stops the operand bits toggling on a power-capped chip -- profiles/r03_power_wall.txt.)
#ifndef PW_DMA_ROWS_PER_TAP_NT1
#define PW_DMA_ROWS_PER_TAP_NT1 3
#endif
#ifndef PW_DMA_ROWS_PER_TAP_NT2
#define PW_DMA_ROWS_PER_TAP_NT2 5
#endif
template <int NT, int TAP>
__device__ __forceinline__ constexpr int h2_dma_first_row() {
  constexpr int per = NT == 1 ? PW_DMA_ROWS_PER_TAP_NT1 : PW_DMA_ROWS_PER_TAP_NT2, t0 = 4 * NT + 2;
  return TAP < t0 ? PIPE_ROWS_PER_WAVE : (TAP - t0) * per;
}
template <int NT, int TAP>
__device__ __forceinline__ constexpr int h2_dma_row_count() {
  constexpr int per = NT == 1 ? PW_DMA_ROWS_PER_TAP_NT1 : PW_DMA_ROWS_PER_TAP_NT2;
  constexpr int k0 = h2_dma_first_row<NT, TAP>();
  return k0 >= PIPE_ROWS_PER_WAVE ? 0 : (k0 + per <= PIPE_ROWS_PER_WAVE ? per : PIPE_ROWS_PER_WAVE - k0);
}

template <int NT>
struct H2Ctx {
  lds3_t lds3;
  rsrc_t xr, wr;
  unsigned lane_off, wstride;
  unsigned wsoff, wsoff_next;
  bool has_next;
  PipeDma dm;
  unsigned dm_base, dm_pitch;  // next stage's halo: byte offset of row (d0-1, h0-1) at wbase / chunk ch; bytes per h row
  int wave, lane;
  long long* tap_probe;
  EpiTile epi;                 // tile parked by the previous stage, written out during this one
  const char* ldsg;            // generic pointer to the LDS base (the parked rows are read through it)
  float res_mul;               // residual as stored -> units of y0 as stored (RngScale::res)
};

template <int TAP>
__device__ __forceinline__ void h2_read_a_tap(lds3_t lds3, const unsigned (&aaddr)[2][3][4], v4f (&aq)[2][4]) {
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

// WR (resident hi planes, see k_conv3d_h2): only the lo pieces q = 1, 3 are fetched per tap
template <int NT, bool WR = false>
__device__ __forceinline__ void h2_load_b(rsrc_t wr, unsigned wsoff, unsigned lane_off, v4f (&b)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = WR ? 1 : 0; q < 4; q += WR ? 2 : 1) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(wr, lane_off + (unsigned)q * H2W_PIECE, wsoff + (unsigned)(nt * 4096), 0);
      v4f o;
      o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
      b[nt][q] = o;
    }
}

// slots of a lane: q = 2*ks + p (p = 0 hi, 1 lo) for activations (aq) and weights (b) alike
template <int NT, bool WR = false, int TAP = 0, int NW = 1>
__device__ __forceinline__ void h2_mfma(const v4f (&aq)[2][4], const v4f (&b)[NT][4], f32x16 (&acc)[2][NT], const v4f (&wres)[NW][2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int prod = 0; prod < 3; ++prod) {          // hi_w.hi_x, lo_w.hi_x, hi_w.lo_x
      const int pw = prod == 1 ? 1 : 0, px = prod == 2 ? 1 : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          v4f w = b[nt][2 * ks + pw];
          if constexpr (WR) { if (pw == 0) w = wres[TAP][ks]; }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w), __builtin_bit_cast(h8, aq[mt][2 * ks + px]),
                                                               acc[mt][nt], 0, 0, 0);
        }
    }
  }
}
template <int NT>
__device__ __forceinline__ void h2_mfma(const v4f (&aq)[2][4], const v4f (&b)[NT][4], f32x16 (&acc)[2][NT]) {
  const v4f none[1][2] = {};
  h2_mfma<NT, false, 0, 1>(aq, b, acc, none);
}

// ---- issue order inside a tap.  One wave per SIMD means nothing else fills the matrix pipe while this wave issues its own
// loads and address arithmetic: with the tap's side work (weight loads two taps ahead, next tap's A fragments, halo row DMA with its
// ~25 scalar instructions, an epilogue pass) in front of the MFMAs, a 768-cycle tap (NT 2) took 930 .. 1600 cycles
// (profiles/r03_conv_h2_stage_probe_64to64.txt).  So the whole tap is ONE scheduling region, branch-free, and the pipeline below
// spreads the side work over the shadows of the 12 NT MFMAs (32 cycles each = ~7 issue slots): slot i gets its share of the LDS
// reads, VMEM reads, SALU and VALU instructions.  64 -> 64 stage: 35.4 k -> 28.4 k cycles, barrier wait 1.2 k -> 0.2 k
// (profiles/r03_conv_h2_stage_probe_64to64_v2.txt).  The WALL time moved by 5 % only: with real operand data the socket is at its
// 1400 W cap and fewer idle cycles are answered with a lower clock (profiles/r03_power_wall.txt).
__device__ __forceinline__ constexpr int h2_share(int i, int slots, int n) { return ((i + 1) * n) / slots - (i * n) / slots; }

template <int I, int S, int NVM, int NDS, int NSA, int NVA>
__device__ __forceinline__ void h2_pipeline() {
  if constexpr (I < S) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (h2_share(I, S, NDS) > 0) __builtin_amdgcn_sched_group_barrier(0x100, h2_share(I, S, NDS), 0);
    if constexpr (h2_share(I, S, NVM) > 0) __builtin_amdgcn_sched_group_barrier(0x020, h2_share(I, S, NVM), 0);
    if constexpr (h2_share(I, S, NSA) > 0) __builtin_amdgcn_sched_group_barrier(0x004, h2_share(I, S, NSA), 0);
    if constexpr (h2_share(I, S, NVA) > 0) __builtin_amdgcn_sched_group_barrier(0x002, h2_share(I, S, NVA), 0);
    h2_pipeline<I + 1, S, NVM, NDS, NSA, NVA>();
  }
}

// halo row `wave + 4 K` of the next stage (pipe_dma_row of pw_conv3d_common.h with the per-stage part of the address
// arithmetic hoisted into H2Ctx and no branch: the row's scalar code has to sit inside the tap's scheduling region)
struct H2DmaView { rsrc_t xr; lds3_t lds3; unsigned base, pitch; int wave; };
template <int K>
__device__ __forceinline__ void h2_dma_row(const ConvArgs& a, const PipeDma& dm, const H2DmaView& c) {
  constexpr int c1 = (4 * K) / TH, c2 = (4 * K) % TH;
  const int t = c2 + c.wave;                               // wave-uniform
  const int carry = t >= TH ? 1 : 0;
  const int dd = c1 + carry, hh = t - TH * carry;
  const bool rok = (int)dm.live & (int)((unsigned)(dm.d0 - 1 + dd) < (unsigned)a.D) & (int)((unsigned)(dm.h0 - 1 + hh) < (unsigned)a.H);
  const unsigned soff = c.base + (unsigned)(dd * a.H + hh) * c.pitch;      // rows outside the volume: every lane is out of range
#ifdef PW_X_DMAOOB
  const unsigned v0 = rok ? PIPE_OOB : PIPE_OOB - 16u, v1 = v0;
#else
  const unsigned v0 = rok ? ((hh & 1) ? dm.voff[1][0] : dm.voff[0][0]) : PIPE_OOB;
  const unsigned v1 = rok ? ((hh & 1) ? dm.voff[1][1] : dm.voff[0][1]) : PIPE_OOB;
#endif
  lds3_t dst = c.lds3 + (dm.ldsbuf + (unsigned)(c1 * TH + c2) * (TW * 128) + (unsigned)c.wave * (TW * 128));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, dst, 16, v0, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, dst + 1024, 4, v1, soff, 0, 0);
}

template <int K0, int N>
__device__ __forceinline__ void h2_dma_rows(const ConvArgs& a, const PipeDma& dm, const H2DmaView& c) {
  if constexpr (N > 0) {
    h2_dma_row<K0>(a, dm, c);
    h2_dma_rows<K0 + 1, N - 1>(a, dm, c);
  }
}

// (PW_X_*: timing-only ablation switches -- each leaves one piece of a tap's side work out, or makes it trivial, and computes
// WRONG results; built by tools/build_variant.py into variant libraries, never into libpreworld_hip.so.  Their table, and why half
// of it measures the power cap rather than the piece left out: profiles/r03_conv_h2_ablation.txt.)
template <int NT, int EPI, int TAP, bool WR = false, int NW = 1>
__device__ __forceinline__ void h2_step(const ConvArgs& a, const H2Ctx<NT>& c, const unsigned (&aaddr)[2][3][4],
                                        v4f (&ac)[2][4], v4f (&an)[2][4], v4f (&b0)[NT][4], v4f (&b1)[NT][4],
                                        v4f (&b2)[NT][4], f32x16 (&acc)[2][NT], EpiRegs& er, const v4f (&wres)[NW][2]) {
#ifdef PW_CONV_TAP_PROBE                   // development aid: cycle counter at every tap of one stage
  if (c.tap_probe) {
    if (c.lane == 0) c.tap_probe[c.wave * 27 + TAP] = __builtin_readcyclecounter();
  }
#endif
  __builtin_amdgcn_sched_barrier(0);
  constexpr int dma_k0 = h2_dma_first_row<NT, TAP>(), dma_n = h2_dma_row_count<NT, TAP>();
#ifdef PW_X_NOEPI
  constexpr bool epi_tap = false, epi_ld = false;
#else
  constexpr bool epi_tap = EPI > 0 && TAP >= 2 && TAP <= 4 * NT + 1;      // pass TAP-2 of the previous tile (loaded last tap)
  constexpr bool epi_ld = EPI > 0 && TAP >= 1 && TAP <= 4 * NT;           // loads of pass TAP-1
#endif
#ifdef PW_X_BSAME
  h2_load_b<NT, WR>(c.wr, c.wsoff, c.lane_off, b2);
#elif !defined(PW_X_NOB)
  if constexpr (TAP + 2 < 27) {
    h2_load_b<NT, WR>(c.wr, c.wsoff + (unsigned)(TAP + 2) * c.wstride, c.lane_off, b2);
  } else {
    h2_load_b<NT, WR>(c.wr, c.wsoff_next + (unsigned)(TAP + 2 - 27) * c.wstride, c.lane_off, b2);
  }
#endif
#ifndef PW_X_NODMA
  h2_dma_rows<dma_k0, dma_n>(a, c.dm, H2DmaView{c.xr, c.lds3, c.dm_base, c.dm_pitch, c.wave});
#endif
#ifndef PW_X_NOA
  if constexpr (TAP < 26) h2_read_a_tap<TAP + 1>(c.lds3, aaddr, an);
#endif
  if constexpr (epi_tap) h2_epi_compute<NT, EPI, TAP - 2>(a, c.epi, er, c.res_mul);
  if constexpr (epi_ld)
    h2_epi_load<NT, EPI, TAP - 1>(a, c.epi, c.ldsg + c.epi.bufoff + (unsigned)c.wave * (TW * 128),
                                  reinterpret_cast<const float*>(c.ldsg + H2_SB_OFF), c.wave, c.lane, er);
  h2_mfma<NT, WR, TAP, NW>(ac, b0, acc, wres);
  {
    constexpr int n_vm = (WR ? 2 : 4) * NT + 2 * dma_n + ((epi_ld && EPI == 2) ? 2 : 0);
    constexpr int n_ds = (TAP < 26 ? 8 : 0) + (epi_ld ? 6 : 0);
    constexpr int n_sa = 4 + 26 * dma_n + (epi_ld ? 12 : 0);
    constexpr int n_va = 5 * dma_n + (epi_tap ? (EPI == 2 ? 100 : 76) : 0) + (epi_ld ? 36 : 0);
    h2_pipeline<0, 12 * NT, n_vm, n_ds, n_sa, n_va>();
    if constexpr (epi_tap) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP < 26) h2_step<NT, EPI, TAP + 1, WR, NW>(a, c, aaddr, an, ac, b1, b2, b0, acc, er, wres);
}

// ---- epilogue shared by the kernels below: y = acc*scale + bias (+residual) (ReLU) through an LDS staging area.
// The transposed accumulators (lane = voxel, 16 channels in groups of 4) are written as fp32 rows into `stg` (rows of
// 8 voxels x 128 B, pitch 4 halo rows = 5120 B: the halo rows wave, wave+4, .. of a buffer no other wave touches), then read
// back with 4 lanes per voxel (8 channels each): every global access is a 32-byte piece of a voxel row and a wave
// instruction covers 16 complete 128-byte rows -- residual loads and stores fully coalesced (storing 8-byte pieces straight
// from the accumulator layout cost 6.3 k cycles per stage, store-issue bound).
template <int NT>
__device__ __forceinline__ void h2_epilogue(const ConvArgs& a, const f32x16 (&acc)[2][NT], const float* sb, char* stg,
                                            int b, int d0, int h0, int w0, int ng, int wave, int lane, float res_mul,
                                            float& amax0, float& amax1) {
  const int od = d0 + wave;
  const int half = lane >> 5, pj = patch_of_row(lane & 31);
  const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
  constexpr unsigned STG_ROW = 4 * TW * 128;                 // rows wave, wave+4, ...: 8 voxels x 128 B used of each
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = (ng * NT + nt) * 32;
    const bool to_y0 = n0 < a.cout0;
    float* dst = to_y0 ? a.y0 : a.y1;
    if (dst == nullptr) continue;
    const int stride = to_y0 ? a.cout0 : a.cout1;
    const int ld = to_y0 ? a.ld0 : a.ld1;
    const int col0 = to_y0 ? n0 : n0 - a.n1_start;
    if (col0 < 0 || col0 >= stride) continue;
    const int fmt = to_y0 ? a.fmt_y0 : a.fmt_y1;
    const float lo_clamp = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
    const bool has_res = to_y0 && a.residual != nullptr;
    const rsrc_t yr = make_rsrc(dst, out_vox * (unsigned)ld * 4u);
    const rsrc_t rr = make_rsrc(has_res ? a.residual : dst, out_vox * (unsigned)ld * 4u);
    // phase A: scale / bias, fp32 rows to LDS (16-byte channel quads XOR-swizzled by the voxel index)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4f sc = *reinterpret_cast<const v4f*>(sb + n0 + 8 * g + 4 * half);
      const v4f bi = *reinterpret_cast<const v4f*>(sb + H2_MAX_COUT + n0 + 8 * g + 4 * half);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int sv = mt * 32 + pj;
        v4f v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[mt][nt][4 * g + e], sc[e], bi[e]);
        *reinterpret_cast<v4f*>(stg + (sv >> 3) * STG_ROW + (sv & 7) * 128 + (((2 * g + half) ^ ((sv >> 1) & 7)) * 16)) = v;
      }
    }
    // phase B: 4 passes of 16 voxels; lane = (voxel sv = 16 q + (lane >> 2), channel octet o = lane & 3)
    const int o = lane & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sv = 16 * q + (lane >> 2);
      const int mt = sv >> 5, vr = (sv >> 3) & 3, vc = sv & 7;
      const bool ok = od < a.Do && (h0 + mt * 4 + vr) < a.Ho && (w0 + vc) < a.Wo;
      const unsigned soff = (unsigned)(((((b * a.Do + od) * a.Ho + h0) * a.Wo + w0) * ld) * 4);
      const unsigned vox = (unsigned)(((mt * 4 + vr) * a.Wo + vc) * ld) * 4u + (unsigned)col0 * 4u;
      const char* src = stg + (sv >> 3) * STG_ROW + (sv & 7) * 128;
      const int sw = (sv >> 1) & 7;
      const v4f lo4 = *reinterpret_cast<const v4f*>(src + (((2 * o) ^ sw) * 16));
      const v4f hi4 = *reinterpret_cast<const v4f*>(src + (((2 * o + 1) ^ sw) * 16));
      float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      const unsigned pos = (fmt == 0 ? (unsigned)(32 * o) : (unsigned)((4 * (o & 1) + 2 * (o >> 1)) * 16));
      const unsigned rpos = (a.fmt_res == 0 ? (unsigned)(32 * o) : (unsigned)((4 * (o & 1) + 2 * (o >> 1)) * 16));
      if (has_res) {
        const unsigned base = ok ? vox + rpos : PIPE_OOB;
        const float4 r0 = buf_load4(rr, base, soff), r1 = buf_load4(rr, ok ? base + 16u : PIPE_OOB, soff);
        if (a.fmt_res == 0) {
          const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(rv[e], res_mul, v[e]);
        } else {                                    // r0 = 8 hi halves, r1 = 8 lo halves of channels 8o .. 8o+7
          const h8 rh = __builtin_bit_cast(h8, r0), rl = __builtin_bit_cast(h8, r1);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf((float)rh[e] + (float)rl[e], res_mul, v[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], lo_clamp);
      if (ok) {
        float m = fabsf(v[0]);
#pragma unroll
        for (int e = 1; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
        if (to_y0) amax0 = fmaxf(amax0, m); else amax1 = fmaxf(amax1, m);
      }
      const unsigned obase = ok ? vox + pos : PIPE_OOB;
      if (fmt == 0) {
        const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
        buf_store4(yr, obase, soff, va);
        buf_store4(yr, ok ? obase + 16u : PIPE_OOB, soff, vb);
      } else {
        h8 oh, ol;
        h2_split8(v, oh, ol);
        const v4f wh = __builtin_bit_cast(v4f, oh), wl = __builtin_bit_cast(v4f, ol);
        const float va[4] = {wh[0], wh[1], wh[2], wh[3]}, vb[4] = {wl[0], wl[1], wl[2], wl[3]};
        buf_store4(yr, obase, soff, va);
        buf_store4(yr, ok ? obase + 16u : PIPE_OOB, soff, vb);
      }
    }
  }
}

// WR (NT = 1, one input chunk, one cout tile: the 32 -> 32 layers of pre_process_net / final_conv): every stage uses the same
// 27 x 4 weight pieces, so the hi planes (27 taps x 2 k-steps = 216 VGPRs; one wave per SIMD has 512) stay in registers and a
// tap fetches only its two lo pieces -- 8 LDS + 2 global operand loads per 12 MFMAs instead of 8 + 4 (profiles/r02_hw_probes.md).
template <int NT, int EPI, bool WR = false>
__global__ void __launch_bounds__(256, 1) k_conv3d_h2(ConvArgs a, PipeArgs p) {
#ifdef PW_X_SKIP_SMALLCONV      // ablation builds only (tools/ablate_step.sh): what would the step gain if the small-grid layers were free?
  if ((long long)a.D * a.H * a.W <= 80000) return;
#endif
#ifdef PW_X_SKIP_BIGCONV
  if ((long long)a.D * a.H * a.W > 80000) return;
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);                  // = d-slice of the tile
  const int half = lane >> 5, j = lane & 31;
  const int pj = patch_of_row(j), pr = pj >> 3, pc = pj & 7;
  const int ntiles_total = a.cout_total >> 5;
  const int nchunk = a.Cin / KC;

  // the operands' range exponents (pw_h2.h "Range"): the loads go out here, their first use is behind the prologue's halo DMA
  const RngScale rs = rng_scales(a);

  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;

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

  H2Ctx<NT> c;
  c.lds3 = (lds3_t)lds;
  c.xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 27 * ntiles_total * 4096));
  c.lane_off = (unsigned)lane * 16u;
  c.wstride = (unsigned)ntiles_total * 4096u;
  c.wave = wave; c.lane = lane;
  c.ldsg = reinterpret_cast<const char*>(lds);
  c.epi.pending = false; c.epi.b = c.epi.d0 = c.epi.h0 = c.epi.w0 = c.epi.ng = 0; c.epi.bufoff = 0;

  constexpr int NW = WR ? 27 : 1;
  v4f wres[NW][2] = {};
  if constexpr (WR) {
#pragma unroll
    for (int tp = 0; tp < 27; ++tp)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(c.wr, c.lane_off + (unsigned)(2 * ks) * H2W_PIECE, (unsigned)tp * c.wstride, 0);
        v4f o;
        o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
        wres[tp][ks] = o;
      }
  }
  PipeTile t = pipe_decode(a, p, item);
  int ch = 0;
  v4f a0[2][4], a1[2][4], b0[NT][4], b1[NT][4], b2[NT][4];
  EpiRegs er = {};
  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  {  // prologue: the first stage's halo goes out in one burst
    PipeDma dm;
    pipe_lane_offsets(a, t.w0, lane, dm.voff);
    dm.b = t.b; dm.d0 = t.d0; dm.h0 = t.h0; dm.wbase = t.w0 > 0 ? t.w0 - 1 : 0; dm.ch = 0; dm.ldsbuf = 0;
    dm.live = true;
    h2_load_b<NT, WR>(c.wr, (unsigned)((t.ng * NT) * 4096), c.lane_off, b0);
    h2_load_b<NT, WR>(c.wr, (unsigned)((t.ng * NT) * 4096) + c.wstride, c.lane_off, b1);
    pipe_dma_row<0>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<1>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<2>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<3>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<4>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<5>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<6>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<7>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<8>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<9>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<10>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<11>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<12>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<13>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<14>(a, c.xr, c.lds3, dm, wave);
    // folded-BN scale / bias of every packed column -> LDS (read back as float4 per channel group in the epilogue), with the
    // range exponents folded in (powers of two, exact) -- in the shadow of the halo loads just issued
    float* sb = lds + H2_SB_OFF / 4;
    for (int n = tid; n < a.cout_total; n += 256) {
      const bool to_y0 = n < a.cout0;
      sb[n] = (a.scale ? a.scale[n] : 1.f) * (to_y0 ? rs.s0 : rs.s1);
      sb[H2_MAX_COUT + n] = (a.bias ? a.bias[n] : 0.f) * (to_y0 ? rs.b0 : rs.b1);
    }
    c.res_mul = rs.res;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }

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
          asm volatile("" : "+v"(aaddr[khp][kw][q]));       // one address register per variant, tap offset = immediate
        }
    long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.probe) ts0 = __builtin_readcyclecounter();
    h2_read_a_tap<0>(c.lds3, aaddr, a0);

    // next stage: next chunk of this tile, else chunk 0 of the block's next item
    PipeTile tn = t;
    int chn = ch + 1, itemn = item;
    if (chn == nchunk) { chn = 0; itemn = item + nslots; }
    c.has_next = itemn < it_end;
    if (c.has_next && chn == 0) tn = pipe_decode(a, p, itemn);
    if (!c.has_next) chn = 0;
    c.wsoff = (unsigned)((ch * 27 * ntiles_total + t.ng * NT) * 4096);
    c.wsoff_next = (unsigned)((chn * 27 * ntiles_total + tn.ng * NT) * 4096);
    pipe_lane_offsets(a, tn.w0, lane, c.dm.voff);
    c.dm.b = tn.b; c.dm.d0 = tn.d0; c.dm.h0 = tn.h0; c.dm.wbase = tn.w0 > 0 ? tn.w0 - 1 : 0;
    c.dm.ch = chn; c.dm.ldsbuf = (unsigned)PIPE_BUF_BYTES - bufoff; c.dm.live = c.has_next;
    c.dm_pitch = (unsigned)(a.W * a.Cin) * 4u;
    c.dm_base = (unsigned)(((((tn.b * a.D + tn.d0 - 1) * a.H + tn.h0 - 1) * a.W + c.dm.wbase) * a.Cin + chn * KC) * 4);

    c.tap_probe = (a.probe && blockIdx.x == 17 && stage == 3) ? a.probe + 256 * 8 * 16 * 4 : nullptr;
    h2_step<NT, EPI, 0, WR, NW>(a, c, aaddr, a0, a1, b0, b1, b2, acc, er, wres);
    if (a.probe) ts1 = __builtin_readcyclecounter();

    __builtin_amdgcn_s_waitcnt(0);     // my DMA rows of the next stage have landed
    __syncthreads();                   // everyone's have; everyone is done with this buffer
    if (a.probe) ts2 = __builtin_readcyclecounter();

    if constexpr (EPI > 0) { if (c.epi.pending) epi_fold_amax<NT>(a, c.epi.ng, er); }
    c.epi.pending = false;               // its passes ran inside this stage's taps
    if (ch == nchunk - 1) {
      char* stg = reinterpret_cast<char*>(lds) + bufoff + (unsigned)wave * (TW * 128);
      if constexpr (EPI == 0) {
        // generic epilogue phase (any mix of formats / residual): parks, reads back and stores before the next stage
        h2_epilogue<NT>(a, acc, lds + H2_SB_OFF / 4, stg, t.b, t.d0, t.h0, t.w0, t.ng, wave, lane, rs.res, er.amax0, er.amax1);
      } else {
        // first half of the pipelined epilogue: park the raw accumulators in this wave's rows of the consumed buffer
        h2_epi_park<NT>(acc, stg, lane);
        c.epi.b = t.b; c.epi.d0 = t.d0; c.epi.h0 = t.h0; c.epi.w0 = t.w0; c.epi.ng = t.ng; c.epi.bufoff = bufoff;
        c.epi.pending = true;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    }
    if (a.probe && lane == 0 && stage < 16) {   // {stage start, taps done, barrier passed, epilogue done}
      long long* pp = a.probe + (((size_t)blockIdx.x * 8 + wave) * 16 + stage) * 4;
      pp[0] = ts0; pp[1] = ts1; pp[2] = ts2; pp[3] = __builtin_readcyclecounter();
    }
    if (!c.has_next) break;
    t = tn; ch = chn; item = itemn;
  }
  // the last tile of this block has no next stage to ride on
  if constexpr (EPI > 0) {
    if (c.epi.pending)
      h2_epi_rest<NT, EPI, 0>(a, c.epi, reinterpret_cast<const char*>(lds) + c.epi.bufoff + (unsigned)wave * (TW * 128),
                              lds + H2_SB_OFF / 4, wave, lane, er, rs.res);
    if (c.epi.pending) epi_fold_amax<NT>(a, c.epi.ng, er);
  }
  if (a.fmt_y0) rng_note(a.y0_rng, __float_as_uint(er.amax0), rs.e0);
  if (a.fmt_y1 && a.y1) rng_note(a.y1_rng, __float_as_uint(er.amax1), rs.e1);
}

// (A paired-wave variant of this kernel -- 8 waves per block, two per SIMD, wave pairs splitting the two k-steps of a chunk and
// meeting in LDS when the tile is parked -- was built and measured in round 3, commit 578d615: 29.4 k instead of 35.4 k cycles
// per 64 -> 64 stage, and the SAME wall time: with real data the socket sits at its 1400 W cap and the shader clock drops to
// match, 1.87 -> 1.69 GHz.  profiles/r03_power_wall.txt; DESIGN.md 4.13.  It was removed again.  Round 4 put it back for the SMALL
// grids, which run far below the cap: 8x100x100 64->64 58.0 -> 56.2 us, 4x50x50 128->128 37.2 -> 35.3 us, no difference in the step
// (389.6 vs 389.0 samples/s over three alternating runs on one box) -- removed again; profiles/r04_small_grid.txt has the table
// and the ablation that shows why: no single piece of a tap's side work costs more than 10 % there.)

// ------------------------------------------------------------------------------------ fp32 <-> h2
// one thread per (voxel, 4-channel group); ld_* = floats between consecutive voxels (channel slices of wider buffers).
// rng: the destination's / source's range slot (pw_h2.h "Range") or null.  auto_exp (host side): the slot's exponent is first
// derived from the largest FINITE |x| (k_rng_clear, k_absmax, k_rng_pick) instead of being taken as it is.
// one block per slot: rng[1] = max(rng[1], partial maxima), partials cleared; optional compact (n, 2) copy
__global__ void __launch_bounds__(256) k_rng_fold(int* tab, int* compact) {
  int* r = tab + (size_t)blockIdx.x * PW_RNG_ROW;
  __shared__ unsigned sm[4];
  unsigned m = 0u;
  for (int k = threadIdx.x; k < PW_RNG_WORDS; k += 256) {
    m = max(m, (unsigned)r[PW_RNG_SCRATCH + k]);
    r[PW_RNG_SCRATCH + k] = 0;
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    m = max(m, (unsigned)r[1]);
    r[1] = (int)m;
    if (compact) { compact[2 * blockIdx.x] = r[0]; compact[2 * blockIdx.x + 1] = (int)m; }
  }
}
// auto_exp of pw_f32_to_h2: fold k_absmax's partials, then pick the exponent (and start recording afresh)
__global__ void k_rng_clear(int* rng) { rng[1] = 0; }
__global__ void k_rng_pick(int* rng) { rng[0] = rng_ideal_exp((unsigned)rng[1]); rng[1] = 0; }

PW_API int pw_rng_fold(int32_t* tab, int n_slots, int32_t* compact, void* stream) {
  PW_CHECK_ARG(tab && n_slots > 0, "pw_rng_fold: bad arguments");
  hipLaunchKernelGGL(k_rng_fold, dim3((unsigned)n_slots), dim3(256), 0, pw_stream(stream), tab, compact);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// the hard window of ops.RangeCtx.check on the device: one thread per slot of a folded (n, 2) table
__global__ void __launch_bounds__(256) k_rng_audit(const int* __restrict__ compact, int n_slots, int* sticky) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_slots; i += 256) {
    const unsigned bits = (unsigned)compact[2 * i + 1];
    if (bits == 0u) continue;
    const float stored = __builtin_ldexpf(__uint_as_float(bits), -compact[2 * i]);
    if (!(stored >= 64.f && stored <= H2_MAX)) atomicAdd(&bad, 1);            // NaN / Inf fail the comparison
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sticky[0] += bad;
    sticky[1] += bad ? 1 : 0;
    sticky[2] += 1;
  }
}
PW_API int pw_rng_audit(const int32_t* compact, int n_slots, int32_t* sticky, void* stream) {
  PW_CHECK_ARG(compact && sticky && n_slots > 0, "pw_rng_audit: bad arguments");
  hipLaunchKernelGGL(k_rng_audit, dim3(1), dim3(256), 0, pw_stream(stream), compact, n_slots, sticky);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

__global__ void k_absmax(const float* __restrict__ x, long long n_vox, int C, int ld_x, int* rng) {
  const int groups = C >> 2;
  const long long n = n_vox * groups;
  unsigned m = 0u;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const long long v = idx / groups;
    const int c = (int)(idx - v * groups) * 4;
    const float4 f = *reinterpret_cast<const float4*>(x + v * ld_x + c);
    const unsigned b[4] = {rng_absbits(f.x), rng_absbits(f.y), rng_absbits(f.z), rng_absbits(f.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) m = max(m, b[k] < 0x7f800000u ? b[k] : 0u);       // Inf / NaN do not choose the exponent
  }
  rng_note(rng, m, 0);
}

__global__ void k_f32_to_h2(const float* __restrict__ x, float* __restrict__ y, long long n_vox, int C, int ld_x, int ld_y,
                            int* rng) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = C >> 2;
  const int e = rng ? rng[0] : 0;
  unsigned m = 0u;
  if (idx < n_vox * groups) {
    const float mul = rng_pow2(-e);
    const long long v = idx / groups;
    const int c = (int)(idx - v * groups) * 4;
    const float4 f = *reinterpret_cast<const float4*>(x + v * ld_x + c);
    const float in[4] = {f.x * mul, f.y * mul, f.z * mul, f.w * mul};
#pragma unroll
    for (int k = 0; k < 4; ++k) m = max(m, rng_absbits(in[k]));
    u2 hi, lo;
    h2_split4(in, hi, lo);
    char* dst = reinterpret_cast<char*>(y + v * ld_y + (c & ~31));
    *reinterpret_cast<u2*>(dst + h2_group_off(c & 31, 0)) = hi;
    *reinterpret_cast<u2*>(dst + h2_group_off(c & 31, 1)) = lo;
  }
  rng_note(rng, m, e);
}

__global__ void k_h2_to_f32(const float* __restrict__ x, float* __restrict__ y, long long n_vox, int C, int ld_x, int ld_y,
                            const int* rng) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = C >> 2;
  if (idx >= n_vox * groups) return;
  const float mul = rng_pow2(rng ? rng[0] : 0);
  const long long v = idx / groups;
  const int c = (int)(idx - v * groups) * 4;
  const char* src = reinterpret_cast<const char*>(x + v * ld_x + (c & ~31));
  const u2 hi = *reinterpret_cast<const u2*>(src + h2_group_off(c & 31, 0));
  const u2 lo = *reinterpret_cast<const u2*>(src + h2_group_off(c & 31, 1));
  float out[4];
  h2_join4(hi, lo, out);
  *reinterpret_cast<float4*>(y + v * ld_y + c) = make_float4(out[0] * mul, out[1] * mul, out[2] * mul, out[3] * mul);
}

PW_API int pw_f32_to_h2(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, int32_t* rng, int auto_exp,
                        void* stream) {
  PW_CHECK_ARG(x && y && n_vox > 0 && C > 0 && C % 32 == 0, "pw_f32_to_h2: C must be a positive multiple of 32");
  if (ld_x <= 0) ld_x = C;
  if (ld_y <= 0) ld_y = C;
  PW_CHECK_ARG(ld_x >= C && ld_y >= C && ld_x % 4 == 0 && ld_y % 32 == 0, "pw_f32_to_h2: bad row strides");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_f32_to_h2: pointers must be 16-B aligned");
  PW_CHECK_ARG(!(auto_exp && !rng), "pw_f32_to_h2: auto_exp needs a range slot");
  const long long n = n_vox * (C / 4);
  hipStream_t st = pw_stream(stream);
  if (auto_exp) {
    hipLaunchKernelGGL(k_rng_clear, dim3(1), dim3(1), 0, st, rng);
    const long long want = pw_cdiv(n, 256);
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, st, x, (long long)n_vox, C, ld_x, rng);
    hipLaunchKernelGGL(k_rng_fold, dim3(1), dim3(256), 0, st, rng, (int*)nullptr);
    hipLaunchKernelGGL(k_rng_pick, dim3(1), dim3(1), 0, st, rng);
  }
  hipLaunchKernelGGL(k_f32_to_h2, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, st, x, y, (long long)n_vox, C, ld_x, ld_y, rng);
  pw_note_kernel("k_f32_to_h2");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_h2_to_f32(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, const int32_t* rng, void* stream) {
  PW_CHECK_ARG(x && y && n_vox > 0 && C > 0 && C % 32 == 0, "pw_h2_to_f32: C must be a positive multiple of 32");
  if (ld_x <= 0) ld_x = C;
  if (ld_y <= 0) ld_y = C;
  PW_CHECK_ARG(ld_x >= C && ld_y >= C && ld_x % 32 == 0 && ld_y % 4 == 0, "pw_h2_to_f32: bad row strides");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_h2_to_f32: pointers must be 16-B aligned");
  const long long n = n_vox * (C / 4);
  hipLaunchKernelGGL(k_h2_to_f32, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream), x, y, (long long)n_vox, C, ld_x,
                     ld_y, rng);
  pw_note_kernel("k_h2_to_f32");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------ host entry
PW_API int pw_conv3d_h2(const float* x, const float* wpk, const float* scale, const float* bias, const float* residual,
                        float* y0, float* y1, int B, int D, int H, int W, int Cin, int cout_total, int cout0, int cout1,
                        int ld_y0, int ld_y1, int ksize, int stride, int relu0, int relu1, int algo, int fmt_y0, int fmt_y1,
                        int fmt_res, const int32_t* x_rng, const int32_t* res_rng, int32_t* y0_rng, int32_t* y1_rng, void* stream) {
  PW_CHECK_ARG(x && wpk && y0, "pw_conv3d_h2: null pointer");
  PW_CHECK_ARG((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1),
               "pw_conv3d_h2: 3x3x3 stride 1 / 2 and 1x1x1 stride 1 are built");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pw_conv3d_h2: bad shape");
  PW_CHECK_ARG(Cin > 0 && Cin % KC == 0, "pw_conv3d_h2: Cin must be a multiple of 32 (got %d)", Cin);
  PW_CHECK_ARG(cout_total > 0 && cout_total % 32 == 0 && cout_total <= H2_MAX_COUT,
               "pw_conv3d_h2: cout_total must be a multiple of 32, at most %d", H2_MAX_COUT);
  PW_CHECK_ARG(cout0 > 0 && cout0 % 32 == 0 && cout1 >= 0 && cout1 % 32 == 0 && cout0 + cout1 <= cout_total,
               "pw_conv3d_h2: cout0 / cout1 must be multiples of 32 within cout_total");
  PW_CHECK_ARG(!(cout1 > 0 && !y1), "pw_conv3d_h2: cout1 > 0 needs y1");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)wpk | (uintptr_t)y0 | (uintptr_t)y1 | (uintptr_t)residual) & 15) == 0,
               "pw_conv3d_h2: pointers must be 16-B aligned");
  PW_CHECK_ARG((unsigned)fmt_y0 < 2 && (unsigned)fmt_y1 < 2 && (unsigned)fmt_res < 2, "pw_conv3d_h2: formats are 0 (fp32) or 1 (h2)");
  ConvArgs a = {};
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias; a.residual = residual; a.y0 = y0; a.y1 = y1;
  const int pad = (ksize - 1) / 2;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
  a.Do = (D + 2 * pad - ksize) / stride + 1; a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.ld0 = ld_y0 > 0 ? ld_y0 : cout0; a.ld1 = ld_y1 > 0 ? ld_y1 : cout1;
  PW_CHECK_ARG(a.ld0 >= cout0 && a.ld1 >= cout1 && a.ld0 % 32 == 0 && a.ld1 % 32 == 0,
               "pw_conv3d_h2: ld_y0 / ld_y1 must be multiples of 32 >= the channel counts");
  a.n1_start = cout0;
  a.relu0 = relu0; a.relu1 = relu1;
  a.fmt_y0 = fmt_y0; a.fmt_y1 = fmt_y1; a.fmt_res = fmt_res;
  a.x_rng = x_rng; a.res_rng = res_rng; a.y0_rng = y0_rng; a.y1_rng = y1_rng;
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  a.tiles_d = (a.Do + BD - 1) / BD; a.tiles_h = (a.Ho + BH - 1) / BH; a.tiles_w = (a.Wo + BW - 1) / BW;
  PW_CHECK_ARG((size_t)B * D * H * W * Cin * 4 < (1ull << 32) &&
                   (size_t)B * a.Do * a.Ho * a.Wo * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) * 4 < (1ull << 32),
               "pw_conv3d_h2: tensors must be < 4 GiB (32-bit buffer addressing)");
  const int ntiles = cout_total / 32;
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (!(ksize == 3 && stride == 1) || algo == 2 || algo == 3) {
    // stride-2 / 1x1x1 (and, on request, tiny 3x3x3 grids): the gather kernel with split-fp16 operands
    if (ksize == 3 && stride == 2 && algo == 0) {
      // LDS-tiled stride-2 kernel (pw_conv3d_h2_s2.hip) for the shapes it is built for (algo 2 keeps the gather kernel)
      const int rc = pw_launch_conv3d_h2_s2(a, pw_stream(stream));
      if (rc == PW_OK) { PW_CHECK_LAUNCH(); return PW_OK; }
      if (rc != PW_EUNSUP) return rc;
    }
    const int NTg = (ntiles % 2 == 0) ? 2 : 1;
    const long long n_out = (long long)B * a.Do * a.Ho * a.Wo;
    if (int rc = pw_launch_conv3d_gather(a, NTg, ntiles / NTg, ksize, stride, algo, Cin, n_out, pw_stream(stream), true)) return rc;
    PW_CHECK_LAUNCH();
    return PW_OK;
  }
  // two N-tiles per wave (A fragments shared) when that still leaves every CU two or more work items
  int NT = (ntiles % 2 == 0 && nblk * (ntiles / 2) >= 2 * pw_num_cus()) ? 2 : 1;
  PipeArgs p;
  p.ngroups = ntiles / NT;
  PW_CHECK_ARG(nblk * p.ngroups < (1ll << 20), "pw_conv3d_h2: too many work items");
  p.n_items = (int)(nblk * p.ngroups);
  p.m_ng = magic_of(p.ngroups); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
  const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
  hipStream_t st = pw_stream(stream);
  // epilogue variant (see "software-pipelined epilogue"): the branch-free in-tap passes need uniform formats
  int epi = 0;
  const bool y1_h2 = cout1 == 0 || fmt_y1 == 1, y1_f32 = cout1 == 0 || fmt_y1 == 0;
  if (fmt_y0 == 1 && y1_h2 && !residual) epi = 1;
  else if (fmt_y0 == 1 && cout1 == 0 && residual && fmt_res == 1) epi = 2;
  else if (fmt_y0 == 0 && y1_f32 && !residual) epi = 3;
#define PW_H2_LAUNCH(NTv, EPIv)                                                          \
  do {                                                                                   \
    static int once = set_lds_limit(k_conv3d_h2<NTv, EPIv>, H2_LDS);                      \
    if (once) return once;                                                               \
    hipLaunchKernelGGL((k_conv3d_h2<NTv, EPIv>), dim3(nb), dim3(256), H2_LDS, st, a, p);  \
    pw_note_kernel("k_conv3d_h2<%d, %d, false>", NTv, EPIv);                                     \
  } while (0)
  // resident hi planes: one chunk, one cout tile
  const bool wres = NT == 1 && ntiles == 1 && Cin == KC && epi > 0;
#define PW_H2_LAUNCH_WR(EPIv)                                                                  \
  do {                                                                                         \
    static int once = set_lds_limit(k_conv3d_h2<1, EPIv, true>, H2_LDS);                        \
    if (once) return once;                                                                     \
    hipLaunchKernelGGL((k_conv3d_h2<1, EPIv, true>), dim3(nb), dim3(256), H2_LDS, st, a, p);    \
    pw_note_kernel("k_conv3d_h2<1, %d, true>", EPIv);                                             \
  } while (0)
  if (wres) {
    if (epi == 1) PW_H2_LAUNCH_WR(1); else if (epi == 2) PW_H2_LAUNCH_WR(2); else PW_H2_LAUNCH_WR(3);
  } else if (NT == 2) {
    if (epi == 1) PW_H2_LAUNCH(2, 1); else if (epi == 2) PW_H2_LAUNCH(2, 2); else if (epi == 3) PW_H2_LAUNCH(2, 3); else PW_H2_LAUNCH(2, 0);
  } else {
    if (epi == 1) PW_H2_LAUNCH(1, 1); else if (epi == 2) PW_H2_LAUNCH(1, 2); else if (epi == 3) PW_H2_LAUNCH(1, 3); else PW_H2_LAUNCH(1, 0);
  }
#undef PW_H2_LAUNCH
#undef PW_H2_LAUNCH_WR
  PW_CHECK_LAUNCH();
  return PW_OK;
}
