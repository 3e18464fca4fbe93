// 3x3x3 stride-1 convolution by Winograd F(2x2x2, 3x3x3) with SPLIT-FP16 operands in the transform domain, on
// v_mfma_f32_32x32x16_f16 tiles (round 4).  Same role as pw_conv3d_h2's 3x3x3 stride-1 case:
//   mmdet3d/models/backbones/resnet.py:88-184 (BasicBlock3D / CustomResNet3D), detectors/preworld.py:72-79 (final_conv).
//
// Why.  The direct split-fp16 kernel executes 3 MFMA-FLOPs per direct-form FLOP and sits at the socket's power cap
// (DESIGN.md 4.13): only executing fewer MFMAs shortens it.  F(2,3) per axis turns the 27 multiply-accumulates of an output voxel
// into 64 / 8 = 8, so the split product executes 3 x 8 / 27 = 0.89.  Round 3's first version of this idea (the wave-specialised fp32
// Winograd kernel with 16x16x32 fp16 MFMAs: a GEMM wave = 16 tiles x 16 columns) was operand-bound: a 4-pass MFMA blocks the
// wave's issue for its 16 cycles, the weights of a point were fetched by two waves, and it tied with the direct kernel.  Here:
//   * a block's 32 Winograd tiles (one 4x8x8 output tile) are ONE 32-column MFMA tile and the output channels come in groups of
//     32 rows: per point, k-step and column group 3 MFMAs of 8 passes (hi.hi, lo_u.hi_v, hi_u.lo_v), whose second half leaves
//     issue slots for the operand loads and the output transform;
//   * the four GEMM waves split (column group g, point subset ph) -- every weight piece is fetched by exactly one wave of the CU
//     (NT = 2: ph = row of the half-step's two i_h rows; NT = 1: ph = quarter of the half-step's 8 points);
//   * each wave accumulates a PARTIAL sum of all 8 outputs of its (tiles x 32 channels); at the end of a tile the partners
//     exchange halves through the V buffers (the transform role does not run ahead across a tile boundary), after which a wave
//     owns 4 (NT = 2) or 2 (NT = 1) complete outputs;
//   * epilogue straight from registers: one v_permlane32_swap per register pair gives a lane whole 8-channel octets of one voxel
//     = whole 16-byte h2 slots; scale / bias / residual / ReLU / split / range maximum, four 16-byte stores per output;
//   * V rows are padded to 144 bytes (hi 16 B | lo 16 B per channel octet) instead of XOR-swizzled: the transform thread's hi and
//     lo halves go out as one ds_write2_b64 and every ds_read_b128 lane group of the GEMM role hits 16 distinct bank quads.
// Transform + DMA role: ws_transform_read of pw_wino_common.h (thread = tile x channel quad, one half-step ahead).
// Weights: preworld_amd.ops.pack_conv_weight_wino_h2 (transform in float64, per-output-channel power-of-two pre-scale).
#include "pw_wino_common.h"
#include "pw_h2.h"

namespace {
constexpr int WX_ROW = 144;                          // bytes of one (point, tile) row of V: 4 octets x (hi | lo) + 16 pad
constexpr int WX_VPT = 32 * WX_ROW;                  // 4608 bytes per point
constexpr int WX_VBUF = 8 * WX_VPT;                  // 36864 bytes per half-step buffer (8 points)
constexpr int WX_SB_OFF = WINO_R_BYTES + 2 * WX_VBUF;   // folded scale / bias table: 64 + 64 floats
constexpr int WX_LDS = WX_SB_OFF + 512;              // 151040 bytes
constexpr unsigned WX_OOB = 0xffffff00u;             // lane offset beyond any num_records (+ <= 255 bytes of slot offsets)

typedef _Float16 wh8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wx_barrier() {
  // LDS operations of this wave done, then the workgroup barrier -- WITHOUT draining vmcnt: the weight prefetch of the GEMM role
  // stays in flight across half-steps (__syncthreads() would wait for it at every barrier)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// development aid (PW_CONV_PROBE=<device pointer>, tools/probe_wino_h2.py): cycle counter of every wave of block 17 when it ARRIVES at
// and LEAVES each workgroup barrier of the block's second tile -> pp[2 k], pp[2 k + 1] for the k-th barrier of the tile
struct WxProbe { long long* pp; int k; };
__device__ __forceinline__ void wx_probe_arrive(WxProbe& q) { if (q.pp) q.pp[2 * q.k] = __builtin_readcyclecounter(); }
__device__ __forceinline__ void wx_probe_leave(WxProbe& q) { if (q.pp) q.pp[2 * q.k + 1] = __builtin_readcyclecounter(); ++q.k; }
__device__ __forceinline__ WxProbe wx_probe_for(const ConvArgs& a, int wave8, int lane, int tile_no) {
  WxProbe q;
  q.pp = (a.probe && blockIdx.x == 17 && tile_no == 1 && lane == 0) ? a.probe + wave8 * 64 : nullptr;
  q.k = 0;
  return q;
}

__device__ __forceinline__ f32x4 wx_buf_load4(rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// ------------------------------------------------------------------------------------------------ transform + DMA role
__device__ __forceinline__ void wx_write_slot(lds3_t lds3, unsigned o, const f32x4& z) {
  const f32x4 s = wino_split_slot(z);                // {4 hi halves (8 bytes), 4 lo halves (8 bytes)} of z / 8
  typedef __attribute__((address_space(3))) f32x2* p2;
  *reinterpret_cast<p2>(lds3 + o) = f32x2{s.x, s.y};
  *reinterpret_cast<p2>(lds3 + o + 16u) = f32x2{s.z, s.w};
}

template <int HH>       // w transform of rows i_h = 2 HH, 2 HH + 1 -> V[HH] (8 points)
__device__ __forceinline__ void wx_transform_write(lds3_t lds3, unsigned v_base, const f32x4 (&y)[4][4]) {
#ifdef WX_X_NO_TWRITE
  return;
#endif
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    f32x4 z0, z1, z2, z3;
    bt4(y[2 * HH + r][0], y[2 * HH + r][1], y[2 * HH + r][2], y[2 * HH + r][3], z0, z1, z2, z3);
    const unsigned o = (unsigned)WINO_R_BYTES + (unsigned)HH * (unsigned)WX_VBUF + v_base + (unsigned)(r * 4) * (unsigned)WX_VPT;
    wx_write_slot(lds3, o, z0);
    wx_write_slot(lds3, o + (unsigned)WX_VPT, z1);
    wx_write_slot(lds3, o + 2u * (unsigned)WX_VPT, z2);
    wx_write_slot(lds3, o + 3u * (unsigned)WX_VPT, z3);
  }
}

#ifdef WX_X_NO_TREAD
#define WX_TREAD(ID) do { } while (0)
#else
#define WX_TREAD(ID) ws_transform_read<ID, H2IN>(lds3, r_base, y, P1, P2)
#endif
// ws_transform_role of pw_wino_common.h with this kernel's V layout and tile-end protocol: on the last chunk of a tile the GEMM
// waves exchange their partial sums through both V buffers between barriers E0 .. E2 (NT = 1: two exchange phases), so the first
// half-step of the next tile is transformed into registers before E0 and written after E2.
template <bool H2IN, int NT>
__device__ __forceinline__ void wx_transform_role(const ConvArgs& a, const PipeArgs& p, lds3_t lds3, int item, int it_end,
                                                  int nslots, int nchunk, int tw, int tt, int lane) {
  const int tile = tt >> 3, quad = tt & 7;
  const int ttd = tile >> 4, tth = (tile >> 2) & 3, ttw = tile & 3;
  const unsigned r_base = (unsigned)((((2 * ttd) * TH + 2 * tth) * TW + 2 * ttw) * 128) + wino_quad_off<H2IN>(quad);
  const unsigned v_base = (unsigned)(tile * WX_ROW + (quad >> 1) * 32 + (quad & 1) * 8);
  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  // halo DMA: this wave moves halo rows tw + 4 K, K = 0..14; the row-only terms live in lane K (see ws_transform_role)
  const int krow = tw + 4 * (lane & 15);
  const int kdd = krow / TH, khh = krow - kdd * TH;
  const unsigned v_rowoff = (unsigned)((kdd * a.H + khh) * a.W) * (unsigned)a.Cin * 4u;
  unsigned dma_base = 0, voff0 = 0, voff1 = 0;
  unsigned long long dma_ok = 0;
  auto aim = [&](int it, int ch) {
    const PipeTile t = pipe_decode(a, p, it);
    PipeDma dm;
    wino_lane_offsets(a, t.w0, lane, dm);
    voff0 = dm.voff[0][0]; voff1 = dm.voff[0][1];
    const int wbase = t.w0 > 0 ? t.w0 - 1 : 0;
    dma_base = (unsigned)(((((t.b * a.D + t.d0 - 1) * a.H + t.h0 - 1) * a.W + wbase) * a.Cin + ch * KC) * 4);
    const int gd = t.d0 + kdd - 1, gh = t.h0 + khh - 1;
    dma_ok = __ballot((unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H);
  };
  auto dma_row = [&](int K) {                            // K is a compile-time constant at every call site
    const unsigned soff = dma_base + (unsigned)__builtin_amdgcn_readlane((int)v_rowoff, K);
    const bool ok = (dma_ok >> K) & 1ull;
    const unsigned v0 = ok ? voff0 : PIPE_OOB, v1 = ok ? voff1 : PIPE_OOB;
    lds3_t dst = lds3 + (unsigned)(tw + 4 * K) * (TW * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v0, ok ? soff : 0u, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 4, v1, ok ? soff : 0u, 0, 0);
  };
  auto dma = [&]() {
    dma_row(0); dma_row(1); dma_row(2); dma_row(3); dma_row(4); dma_row(5); dma_row(6); dma_row(7);
    dma_row(8); dma_row(9); dma_row(10); dma_row(11); dma_row(12); dma_row(13); dma_row(14);
  };
  f32x4 y[4][4] = {}, P1[4][4] = {}, P2[4][4] = {};
  aim(item, 0);
  dma();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                                            // barrier A: R of the first chunk
  WX_TREAD(0);
  wx_transform_write<0>(lds3, v_base, y);
  __syncthreads();                                            // barrier B: half-step 0 in V[0]
  int tile_no = 0;
  for (; item < it_end; item += nslots, ++tile_no) {
    WxProbe q = wx_probe_for(a, 4 + tw, lane, tile_no);
    for (int ch = 0; ch < nchunk; ++ch) {
      const bool more_ch = ch + 1 < nchunk;
      const bool has_next = more_ch || item + nslots < it_end;
      wx_transform_write<1>(lds3, v_base, y); { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }                                                        // step 0
      WX_TREAD(1); wx_transform_write<0>(lds3, v_base, y); { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }   // step 1
      wx_transform_write<1>(lds3, v_base, y); { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }                                                        // step 2
      WX_TREAD(2); wx_transform_write<0>(lds3, v_base, y); { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }   // step 3
      wx_transform_write<1>(lds3, v_base, y);                                                                         // step 4
      if (has_next) aim(more_ch ? item : item + nslots, more_ch ? ch + 1 : 0);
      { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }
      if (has_next) dma();                                                                                            // step 5
      WX_TREAD(3); wx_transform_write<0>(lds3, v_base, y);
      { wx_probe_arrive(q); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); wx_probe_leave(q); }      // (not __syncthreads(): the DMA stays in flight)
      wx_transform_write<1>(lds3, v_base, y);                                                                         // step 6
      __builtin_amdgcn_s_waitcnt(0);
      { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }
      if (has_next) WX_TREAD(0);                                                                                      // step 7
      if (!more_ch) {
        { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }                                        // E0: the GEMM waves are done with V
        { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }                                        // E1: partial sums written
        if constexpr (NT == 1) { { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); } { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); } }
        { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }                                        // E2: ... and read; V is free again
      }
      if (has_next) wx_transform_write<0>(lds3, v_base, y);
      { wx_probe_arrive(q); __syncthreads(); wx_probe_leave(q); }
    }
  }
}

// ------------------------------------------------------------------------------------------------ GEMM role
struct WxCtx {
  lds3_t lds3;
  rsrc_t wr;
  unsigned v_addr[2];        // LDS byte address of this lane's row in V[0] / V[1], first point of the wave's share
  unsigned lane_off;         // lane * 16
};

// stream position S of half-step (ID, HH) of a wave: point kp = S >> 1 of its share, k-step ks = S & 1; S >= NS runs on into the
// following half-steps (weight prefetch), H >= 8 = the next chunk
template <int NT, int ID, int HH, int S> struct WxPos {
  static constexpr int NS = 4 * NT;
  static constexpr int H = ID * 2 + HH + S / NS;
  static constexpr bool next_chunk = H >= 8;
  static constexpr int Hm = H % 8, s = S % NS, kp = s >> 1, ks = s & 1;
  // bytes from the wave's weight base of the chunk: point Hm * 8 + (ph * PPW) + kp, 4096 NT bytes per point, k-step ks = pieces 2 ks, 2 ks + 1
  static constexpr unsigned uoff = (unsigned)(((Hm * 8 + kp) * NT) * 4096 + ks * 2048);
};

// the point (i_d, i_h, i_w) wave PH handles as kp-th of half-step H (0..7)
template <int NT, int PH, int H, int KP> struct WxPoint {
  static constexpr int ID = H >> 1, HH = H & 1;
  static constexpr int IH = 2 * HH + (NT == 2 ? PH : (PH >> 1));
  static constexpr int IW = NT == 2 ? KP : 2 * (PH & 1) + KP;
};

// half HALF (accumulator registers 8 HALF .. 8 HALF + 7) of the output transform of one point's products
template <int ID, int IH, int IW, int HALF>
__device__ __forceinline__ void wx_scatter(const f32x16& M, f32x2 (&Y)[8][8]) {
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int sg = at_sign(o >> 2, ID) * at_sign((o >> 1) & 1, IH) * at_sign(o & 1, IW);
    if (sg == 0) continue;
#pragma unroll
    for (int e = 4 * HALF; e < 4 * HALF + 4; ++e) {
      const f32x2 m = {M[2 * e], M[2 * e + 1]};
      Y[o][e] = sg > 0 ? pk_add(Y[o][e], m) : pk_sub(Y[o][e], m);
    }
  }
}

template <int NT, int PH, int ID, int HH, int S>
__device__ __forceinline__ void wx_step(const WxCtx& c, unsigned ub_cur, unsigned ub_next, f32x4 (&U)[4][2], f32x4 (&V)[2][2],
                                        f32x16 (&M)[2], f32x2 (&Y)[8][8]) {
  constexpr int NS = 4 * NT, PPW = 2 * NT;
  typedef WxPos<NT, ID, HH, S> P;
  // (WX_X_*: timing-only ablation switches -- each leaves one piece of the role out and computes WRONG results; built by
  // tools/build_variant.py into variant libraries, never into libpreworld_hip.so)
#ifndef WX_X_NO_U
  {  // weights three k-steps ahead (hi, lo pieces of 1 KB = 64 lanes x 16 B)
    typedef WxPos<NT, ID, HH, S + 3> Q;
    const unsigned b = (Q::next_chunk ? ub_next : ub_cur) + Q::uoff;
    constexpr int r = (S + 3) & 3;
    U[r][0] = wx_buf_load4(c.wr, c.lane_off, b);
    U[r][1] = wx_buf_load4(c.wr, c.lane_off, b + 1024u);
  }
#endif
#ifndef WX_X_NO_V
  if constexpr (S + 1 < NS) {   // V fragments one k-step ahead
    typedef WxPos<NT, ID, HH, S + 1> Q;
    constexpr int r = (S + 1) & 1;
    const unsigned o = c.v_addr[HH] + (unsigned)(Q::kp * WX_VPT + Q::ks * 32);
    V[r][0] = lds_read4(c.lds3, o);
    V[r][1] = lds_read4(c.lds3, o + 16u);
  }
#endif
#ifndef WX_X_NO_MFMA
  {
    const wh8 uh = __builtin_bit_cast(wh8, U[S & 3][0]), ul = __builtin_bit_cast(wh8, U[S & 3][1]);
    const wh8 vh = __builtin_bit_cast(wh8, V[S & 1][0]), vl = __builtin_bit_cast(wh8, V[S & 1][1]);
    f32x16 acc;
    if constexpr (P::ks == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
      acc = M[P::kp & 1];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl, acc, 0, 0, 0);
    M[P::kp & 1] = acc;
  }
#endif
#ifndef WX_X_NO_SCATTER
  {  // output transform of the PREVIOUS point's products (the other accumulator), half of it per k-step, under these MFMAs
    constexpr int Hc = ID * 2 + HH;
    constexpr int Hp = P::kp > 0 ? Hc : (Hc + 7) % 8;
    constexpr int KPp = P::kp > 0 ? P::kp - 1 : PPW - 1;
    typedef WxPoint<NT, PH, Hp, KPp> Pt;
    wx_scatter<Pt::ID, Pt::IH, Pt::IW, P::ks>(M[(P::kp & 1) ^ 1], Y);
  }
#endif
  // one scheduling region per k-step: the loads and the transform's packed adds go into the MFMA shadows
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (S + 1 < NS) wx_step<NT, PH, ID, HH, S + 1>(c, ub_cur, ub_next, U, V, M, Y);
}

template <int NT, int PH, int ID, int HH>
__device__ __forceinline__ void wx_halfstep(const WxCtx& c, unsigned ub_cur, unsigned ub_next, f32x4 (&U)[4][2], f32x4 (&V)[2][2],
                                            f32x16 (&M)[2], f32x2 (&Y)[8][8]) {
  V[0][0] = lds_read4(c.lds3, c.v_addr[HH]);           // position 0 = point 0, k-step 0 (the buffer became valid at the barrier)
  V[0][1] = lds_read4(c.lds3, c.v_addr[HH] + 16u);
  wx_step<NT, PH, ID, HH, 0>(c, ub_cur, ub_next, U, V, M, Y);
}

// kept / sent outputs of the exchange
template <int NT, int PH, int I> struct WxKeep {
  // NT 2: keep od = PH (o = 4 PH + I, I = 0..3).  NT 1: od = PH & 1, oh = PH >> 1, ow = I (I = 0..1)
  static constexpr int o = NT == 2 ? 4 * PH + I : 4 * (PH & 1) + 2 * (PH >> 1) + I;
  static constexpr int od = o >> 2, oh = (o >> 1) & 1, ow = o & 1;
};

__device__ __forceinline__ void wx_send(lds3_t lds3, unsigned base, const f32x2 (&Yo)[8], int slot) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    lds_write4(lds3, base + (unsigned)(slot * 4 + q) * 1024u, f32x4{Yo[2 * q].x, Yo[2 * q].y, Yo[2 * q + 1].x, Yo[2 * q + 1].y});
}
__device__ __forceinline__ void wx_recv_add(lds3_t lds3, unsigned base, f32x2 (&Yo)[8], int slot) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = lds_read4(lds3, base + (unsigned)(slot * 4 + q) * 1024u);
    Yo[2 * q] = pk_add(Yo[2 * q], f32x2{v.x, v.y});
    Yo[2 * q + 1] = pk_add(Yo[2 * q + 1], f32x2{v.z, v.w});
  }
}

// what a GEMM wave needs to write its outputs (fixed for the life of the block: its column group goes to one destination)
struct WxEpi {
  rsrc_t yr, rr;
  unsigned lane_vo;          // byte offset of the lane's Winograd tile origin + its first slot inside the destination row
  unsigned offB;             // bytes from the lane's first octet to its second
  int ld, fmt, fmt_res;
  bool has_res;
  float relu_lo, res_mul;
  unsigned sb_off;           // LDS byte address of this lane's 16 scales (its 16 biases 256 bytes on)
  int ttd, tth, ttw;
};

// one complete output (od, oh, ow) of the wave's 32 tiles x 32 channels: F = 16 accumulator registers of this lane
template <int OD, int OH, int OW>
__device__ __forceinline__ void wx_store_output(const ConvArgs& a, const WxEpi& E, const PipeTile& t, lds3_t lds3, f32x2 (&F)[8],
                                                float& amax) {
  // v_permlane32_swap: registers (a, b) and (a + 2, b) trade their upper / lower lane halves -> lane half 0 holds channel octets
  // 0, 1 of its tile's voxel, lane half 1 octets 2, 3 (accumulator register r = 4 a + b = channel 8 a + 4 (lane >> 5) + b)
  float x[16];
#pragma unroll
  for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r0 = 4 * a2 + b, r1 = 4 * (a2 + 2) + b;
      const float v0 = F[r0 >> 1][r0 & 1], v1 = F[r1 >> 1][r1 & 1];
      const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
      x[8 * a2 + b] = __uint_as_float(s[0]);           // channels +0..3 of octet a2 (half 0) / a2 + 2 (half 1)
      x[8 * a2 + 4 + b] = __uint_as_float(s[1]);       // channels +4..7
    }
  const bool valid = t.d0 + 2 * E.ttd + OD < a.Do && t.h0 + 2 * E.tth + OH < a.Ho && t.w0 + 2 * E.ttw + OW < a.Wo;
  const unsigned vo = valid ? E.lane_vo : WX_OOB;
  const unsigned so = (unsigned)(((t.b * a.Do + t.d0 + OD) * a.Ho + t.h0 + OH) * a.Wo + t.w0 + OW) * (unsigned)E.ld * 4u;
  // residual first (its loads fly while the scale / bias table is read)
  float res[16];
  if (E.has_res) {
#pragma unroll
    for (int oc = 0; oc < 2; ++oc) {
      const unsigned ro = vo + (oc ? E.offB : 0u);
      const f32x4 r0 = wx_buf_load4(E.rr, ro, so), r1 = wx_buf_load4(E.rr, ro + 16u, so);
      if (E.fmt_res) {
        const wh8 hi = __builtin_bit_cast(wh8, r0), lo = __builtin_bit_cast(wh8, r1);
#pragma unroll
        for (int e = 0; e < 8; ++e) res[8 * oc + e] = (float)hi[e] + (float)lo[e];
      } else {
        res[8 * oc + 0] = r0.x; res[8 * oc + 1] = r0.y; res[8 * oc + 2] = r0.z; res[8 * oc + 3] = r0.w;
        res[8 * oc + 4] = r1.x; res[8 * oc + 5] = r1.y; res[8 * oc + 6] = r1.z; res[8 * oc + 7] = r1.w;
      }
    }
  }
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 sc = lds_read4(lds3, E.sb_off + (unsigned)q * 16u), bi = lds_read4(lds3, E.sb_off + 256u + (unsigned)q * 16u);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = x[4 * q + e] * sc[e] + bi[e];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (E.has_res) v[e] += res[e] * E.res_mul;
    v[e] = fmaxf(v[e], E.relu_lo);
    amax = fmaxf(amax, fabsf(v[e]));
  }
#pragma unroll
  for (int oc = 0; oc < 2; ++oc) {
    const unsigned wo = vo + (oc ? E.offB : 0u);
    if (E.fmt) {
      const float v8[8] = {v[8 * oc], v[8 * oc + 1], v[8 * oc + 2], v[8 * oc + 3], v[8 * oc + 4], v[8 * oc + 5], v[8 * oc + 6], v[8 * oc + 7]};
      h8 hi, lo;
      h2_split8(v8, hi, lo);
      const v4f wh = __builtin_bit_cast(v4f, hi), wl = __builtin_bit_cast(v4f, lo);
      const float va[4] = {wh[0], wh[1], wh[2], wh[3]}, vb[4] = {wl[0], wl[1], wl[2], wl[3]};
      buf_store4(E.yr, wo, so, va);
      buf_store4(E.yr, wo + 16u, so, vb);
    } else {
      const float va[4] = {v[8 * oc], v[8 * oc + 1], v[8 * oc + 2], v[8 * oc + 3]};
      const float vb[4] = {v[8 * oc + 4], v[8 * oc + 5], v[8 * oc + 6], v[8 * oc + 7]};
      buf_store4(E.yr, wo, so, va);
      buf_store4(E.yr, wo + 16u, so, vb);
    }
  }
}

// end of a tile: flush, exchange the partial sums with the partner wave(s), write the outputs this wave ends up owning
template <int NT, int PH>
__device__ __forceinline__ void wx_tile_end(const ConvArgs& a, const WxEpi& E, const PipeTile& t, lds3_t lds3, int wave, int lane,
                                            f32x2 (&Y)[8][8], float& amax, WxProbe& q) {
  const unsigned xbase = (unsigned)WINO_R_BYTES + (unsigned)lane * 16u;
  const unsigned mine = xbase + (unsigned)wave * 16384u;
  { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                                   // E0: every GEMM wave is done reading V
  if constexpr (NT == 2) {
    const unsigned theirs = xbase + (unsigned)(wave ^ 2) * 16384u;
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_send(lds3, mine, Y[4 * (1 - PH) + i], i);
    { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                                 // E1
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_recv_add(lds3, theirs, Y[4 * PH + i], i);
    { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                                 // E2
    wx_store_output<WxKeep<2, PH, 0>::od, WxKeep<2, PH, 0>::oh, WxKeep<2, PH, 0>::ow>(a, E, t, lds3, Y[WxKeep<2, PH, 0>::o], amax);
    wx_store_output<WxKeep<2, PH, 1>::od, WxKeep<2, PH, 1>::oh, WxKeep<2, PH, 1>::ow>(a, E, t, lds3, Y[WxKeep<2, PH, 1>::o], amax);
    wx_store_output<WxKeep<2, PH, 2>::od, WxKeep<2, PH, 2>::oh, WxKeep<2, PH, 2>::ow>(a, E, t, lds3, Y[WxKeep<2, PH, 2>::o], amax);
    wx_store_output<WxKeep<2, PH, 3>::od, WxKeep<2, PH, 3>::oh, WxKeep<2, PH, 3>::ow>(a, E, t, lds3, Y[WxKeep<2, PH, 3>::o], amax);
  } else {
    constexpr int od = PH & 1, oh = PH >> 1;
    {  // phase 1: partner wave ^ 1 owns the other output plane
      const unsigned theirs = xbase + (unsigned)(wave ^ 1) * 16384u;
#pragma unroll
      for (int i = 0; i < 4; ++i) wx_send(lds3, mine, Y[4 * (1 - od) + i], i);
      { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                               // E1
#pragma unroll
      for (int i = 0; i < 4; ++i) wx_recv_add(lds3, theirs, Y[4 * od + i], i);
      { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                               // E1b: phase-1 reads done before the regions are rewritten
    }
    {  // phase 2: partner wave ^ 2 owns the other output row of this plane
      const unsigned theirs = xbase + (unsigned)(wave ^ 2) * 16384u;
#pragma unroll
      for (int i = 0; i < 2; ++i) wx_send(lds3, mine, Y[4 * od + 2 * (1 - oh) + i], i);
      { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                               // E1c
#pragma unroll
      for (int i = 0; i < 2; ++i) wx_recv_add(lds3, theirs, Y[4 * od + 2 * oh + i], i);
      { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                               // E2
    }
    wx_store_output<od, oh, 0>(a, E, t, lds3, Y[4 * od + 2 * oh + 0], amax);
    wx_store_output<od, oh, 1>(a, E, t, lds3, Y[4 * od + 2 * oh + 1], amax);
  }
}

template <int NT, int PH>
__device__ __forceinline__ void wx_gemm_role(const ConvArgs& a, const PipeArgs& p, lds3_t lds3, int item, int it_end, int nslots,
                                             int nchunk, int wave, int lane, const RngScale& rs) {
  constexpr int PPW = 2 * NT;
  const int g = NT == 2 ? (wave & 1) : 0;
  const int j = lane & 31, kg = lane >> 5;
  WxCtx c;
  c.lds3 = lds3;
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 64 * NT * 4096));
  c.lane_off = (unsigned)lane * 16u;
#pragma unroll
  for (int b = 0; b < 2; ++b)
    c.v_addr[b] = (unsigned)(WINO_R_BYTES + b * WX_VBUF + (PH * PPW) * WX_VPT + j * WX_ROW + 64 * kg);
  const unsigned ubase_w = (unsigned)(((PH * PPW) * NT + g) * 4096);
  const unsigned chunk_bytes = (unsigned)(64 * NT * 4096);

  // destination of this wave's column group
  WxEpi E;
  {
    const int n0 = 32 * g;
    const bool to_y0 = n0 < a.cout0;
    float* dst = to_y0 ? a.y0 : a.y1;
    E.ld = to_y0 ? a.ld0 : a.ld1;
    E.fmt = to_y0 ? a.fmt_y0 : a.fmt_y1;
    E.fmt_res = a.fmt_res;
    E.has_res = to_y0 && a.residual != nullptr;
    E.relu_lo = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
    E.res_mul = rs.res;
    const int col0 = to_y0 ? n0 : n0 - a.n1_start;
    const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
    E.yr = make_rsrc(dst, out_vox * (unsigned)E.ld * 4u);
    E.rr = make_rsrc(E.has_res ? a.residual : dst, out_vox * (unsigned)E.ld * 4u);
    E.ttd = j >> 4; E.tth = (j >> 2) & 3; E.ttw = j & 3;
    // first octet of the lane: h2 storage: octet 2 kg = slot (half 0, k-step kg) at byte 32 kg, second octet (half 1) 64 bytes on;
    // fp32: channels 16 kg .. 16 kg + 7 at byte 64 kg, the second octet 32 bytes on.  (Formats of y and the residual agree or the
    // residual is fp32 next to an fp32 destination -- checked on the host.)
    const unsigned first = E.fmt ? 32u * (unsigned)kg : 64u * (unsigned)kg;
    E.offB = E.fmt ? 64u : 32u;
    E.lane_vo = (unsigned)(((2 * E.ttd * a.Ho + 2 * E.tth) * a.Wo + 2 * E.ttw) * E.ld) * 4u + (unsigned)col0 * 4u + first;
    E.sb_off = (unsigned)WX_SB_OFF + (unsigned)(n0 + 16 * kg) * 4u;
  }
  float amax = 0.f;

  f32x4 U[4][2], V[2][2];
  f32x16 M[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { M[0][r] = 0.f; M[1][r] = 0.f; }
  // weights of stream positions 0..2 of the first half-step
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const unsigned b = ubase_w + (unsigned)(((s >> 1) * NT) * 4096 + (s & 1) * 2048);
    U[s][0] = wx_buf_load4(c.wr, c.lane_off, b);
    U[s][1] = wx_buf_load4(c.wr, c.lane_off, b + 1024u);
  }
  PipeTile t = pipe_decode(a, p, item);
  wx_barrier();                                                 // barrier A
  wx_barrier();                                                 // barrier B
  int tile_no = 0;
  for (; item < it_end; item += nslots, ++tile_no) {
    WxProbe q = wx_probe_for(a, wave, lane, tile_no);
    f32x2 Y[8][8];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) Y[o][e] = f32x2{0.f, 0.f};
    for (int ch = 0; ch < nchunk; ++ch) {
      const unsigned ub_cur = (unsigned)ch * chunk_bytes + ubase_w;
      const unsigned ub_next = (ch + 1 < nchunk ? (unsigned)(ch + 1) * chunk_bytes : 0u) + ubase_w;
      wx_halfstep<NT, PH, 0, 0>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 0, 1>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 1, 0>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 1, 1>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 2, 0>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 2, 1>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 3, 0>(c, ub_cur, ub_next, U, V, M, Y); { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
      wx_halfstep<NT, PH, 3, 1>(c, ub_cur, ub_next, U, V, M, Y);
      if (ch + 1 < nchunk) { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }
    }
    {  // the last point's products (the half-step code transforms a point under the NEXT point's MFMAs)
      typedef WxPoint<NT, PH, 7, PPW - 1> Pt;
      wx_scatter<Pt::ID, Pt::IH, Pt::IW, 0>(M[1], Y);
      wx_scatter<Pt::ID, Pt::IH, Pt::IW, 1>(M[1], Y);
#pragma unroll
      for (int r = 0; r < 16; ++r) M[1][r] = 0.f;                // the next tile's first step adds it again: zeros
    }
    wx_tile_end<NT, PH>(a, E, t, lds3, wave, lane, Y, amax, q);
    if (item + nslots < it_end) t = pipe_decode(a, p, item + nslots);
    { wx_probe_arrive(q); wx_barrier(); wx_probe_leave(q); }                                               // B': half-step 0 of the next tile is in V[0]
  }
  {
    const bool to_y0 = 32 * g < a.cout0;
    if (E.fmt) rng_note(to_y0 ? a.y0_rng : a.y1_rng, __float_as_uint(amax), to_y0 ? rs.e0 : rs.e1);
  }
}
}  // namespace

template <int NT, bool H2IN>
__global__ void __launch_bounds__(512, 1) k_conv3d_wino_h2(ConvArgs a, PipeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;
  const int nchunk = a.Cin / KC;
  const lds3_t lds3 = (lds3_t)lds;
  const RngScale rs = rng_scales(a);
  // folded scale / bias of every packed column, range exponents folded in (powers of two, exact); visible after barrier A
  if (tid < a.cout_total) {
    const bool to_y0 = tid < a.cout0;
    float* sb = lds + WX_SB_OFF / 4;
    sb[tid] = (a.scale ? a.scale[tid] : 1.f) * (to_y0 ? rs.s0 : rs.s1);
    sb[64 + tid] = (a.bias ? a.bias[tid] : 0.f) * (to_y0 ? rs.b0 : rs.b1);
  }
  if (wave >= 4) {
    wx_transform_role<H2IN, NT>(a, p, lds3, item, it_end, nslots, nchunk, wave - 4, tid - 256, lane);
    return;
  }
  if constexpr (NT == 2) {
    if ((wave >> 1) == 0) wx_gemm_role<2, 0>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
    else wx_gemm_role<2, 1>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
  } else {
    if (wave == 0) wx_gemm_role<1, 0>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
    else if (wave == 1) wx_gemm_role<1, 1>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
    else if (wave == 2) wx_gemm_role<1, 2>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
    else wx_gemm_role<1, 3>(a, p, lds3, item, it_end, nslots, nchunk, wave, lane, rs);
  }
}

// Winograd F(2x2x2, 3x3x3) form of pw_conv3d_h2's 3x3x3 stride-1 case (include/preworld_hip.h).
PW_API int pw_conv3d_wino_h2(const float* x, int fmt_x, const float* uwpk, const float* scale, const float* bias,
                             const float* residual, float* y0, float* y1, int B, int D, int H, int W, int Cin, int cout_total,
                             int cout0, int cout1, int ld_y0, int ld_y1, int relu0, int relu1, int fmt_y0, int fmt_y1, int fmt_res,
                             const int32_t* x_rng, const int32_t* res_rng, int32_t* y0_rng, int32_t* y1_rng, void* stream) {
  PW_CHECK_ARG(x && uwpk && y0, "pw_conv3d_wino_h2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cin % KC == 0, "pw_conv3d_wino_h2: bad shape");
  PW_CHECK_ARG(cout_total == 32 || cout_total == 64, "pw_conv3d_wino_h2: cout_total must be 32 or 64 (got %d)", cout_total);
  PW_CHECK_ARG(cout0 > 0 && cout0 % 32 == 0 && cout1 >= 0 && cout1 % 32 == 0 && cout0 + cout1 == cout_total,
               "pw_conv3d_wino_h2: cout0 / cout1 must be multiples of 32 adding up to cout_total");
  PW_CHECK_ARG(!(cout1 > 0 && !y1), "pw_conv3d_wino_h2: cout1 > 0 needs y1");
  PW_CHECK_ARG((fmt_x == 0 || fmt_x == 1) && (fmt_y0 == 0 || fmt_y0 == 1) && (fmt_y1 == 0 || fmt_y1 == 1) &&
                   (fmt_res == 0 || fmt_res == 1),
               "pw_conv3d_wino_h2: formats are 0 (fp32) or 1 (h2)");
  PW_CHECK_ARG(!residual || fmt_res == fmt_y0, "pw_conv3d_wino_h2: the residual must be stored in y0's format");
  ConvArgs a = {};
  a.x = x; a.wpk = uwpk; a.scale = scale; a.bias = bias; a.residual = residual; a.y0 = y0; a.y1 = y1;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.ld0 = ld_y0 > 0 ? ld_y0 : cout0; a.ld1 = ld_y1 > 0 ? ld_y1 : cout1;
  a.n1_start = cout0;
  a.relu0 = relu0; a.relu1 = relu1;
  a.fmt_y0 = fmt_y0; a.fmt_y1 = fmt_y1; a.fmt_res = residual ? fmt_res : 0;
  a.x_rng = fmt_x ? x_rng : nullptr; a.res_rng = (residual && fmt_res) ? res_rng : nullptr; a.y0_rng = y0_rng; a.y1_rng = y1_rng;
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  PW_CHECK_ARG(a.ld0 % 32 == 0 && (cout1 == 0 || a.ld1 % 32 == 0), "pw_conv3d_wino_h2: row strides must be multiples of 32 channels");
  PW_CHECK_ARG((size_t)B * D * H * W * Cin * 4 < 0xff000000ull &&
                   (size_t)B * D * H * W * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) * 4 < 0xff000000ull,
               "pw_conv3d_wino_h2: tensors must be < 4 GiB (32-bit buffer addressing)");
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  PW_CHECK_ARG(nblk < (1ll << 20), "pw_conv3d_wino_h2: too many tiles");
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
  PipeArgs p = {};
  p.ngroups = 1;
  p.n_items = (int)nblk;
  p.m_ng = magic_of(1); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
#define PW_WINO_X(NTv, INv)                                                                                            \
  do {                                                                                                                 \
    PW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3d_wino_h2<NTv, INv>),                        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, WX_LDS));                             \
    hipLaunchKernelGGL((k_conv3d_wino_h2<NTv, INv>), dim3(nb), dim3(512), WX_LDS, pw_stream(stream), a, p);             \
    pw_note_kernel("k_conv3d_wino_h2<%d, %s>", NTv, INv ? "true" : "false");                                           \
  } while (0)
  if (cout_total == 64) { if (fmt_x) PW_WINO_X(2, true); else PW_WINO_X(2, false); }
  else { if (fmt_x) PW_WINO_X(1, true); else PW_WINO_X(1, false); }
#undef PW_WINO_X
  PW_CHECK_LAUNCH();
  return PW_OK;
}
