// Shared pieces of the Winograd F(2x2x2, 3x3x3) kernels (pw_conv3d_wino.hip, pw_occ_head.hip): LDS layout,
// packed-add helpers, input / output transforms, the transform + DMA role of the wave-specialised kernels.
#pragma once
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

// Packed fp32 adds: the transforms are pure add/sub streams, v_pk_add_f32 does two per instruction (the
// compiler packs fadd but not fsub, hence the neg_lo/neg_hi form by hand).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (the compiler packs only part of the plain vector adds -- 248 of the GEMM role's 384 output-transform adds per chunk
// stayed scalar v_add_f32 -- so additions are spelled out as well)
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) {
  const f32x2 lo = pk_add(a.xy, b.xy), hi = pk_add(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(const f32x4& a, const f32x4& b) {
  const f32x2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] applied to (x0..x3)
__device__ __forceinline__ void bt4(const f32x4& x0, const f32x4& x1, const f32x4& x2, const f32x4& x3,
                                    f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) {
  y0 = sub4(x0, x2); y1 = add4(x1, x2); y2 = sub4(x2, x1); y3 = sub4(x1, x3);
}
__device__ __forceinline__ f32x4 lds_read4(lds3_t base, unsigned off) {
  return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(base + off);
}
__device__ __forceinline__ void lds_write4(lds3_t base, unsigned off, const f32x4& v) {
  *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(base + off) = v;
}
}  // namespace

// A^T = [[1,1,1,0],[0,1,-1,-1]]: sign of point index i (0..3) in output o (0..1), 0 = no contribution
__device__ __forceinline__ constexpr int at_sign(int o, int i) {
  return o == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : (i == 1 ? 1 : -1));
}

// Output transform of one row of points (i_d = ID, i_h = IH, i_w = 0..3): along w first
// (t0 = M0 + M1 + M2, t1 = M1 - M2 - M3), then each t into the (o_d, o_h) outputs it feeds --
// 8.5 adds per accumulator register and row instead of 13.5 for point-by-point scattering.
template <int ID, int IH>
__device__ __forceinline__ void wino_scatter_row(const f32x4 (&M)[4], f32x4 (&Y)[8]) {
  const f32x4 s12 = add4(M[1], M[2]), d12 = sub4(M[1], M[2]);
  const f32x4 t[2] = {add4(M[0], s12), sub4(d12, M[3])};
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int sg = at_sign(o >> 2, ID) * at_sign((o >> 1) & 1, IH);
    if (sg > 0) Y[o] = add4(Y[o], t[o & 1]);
    else if (sg < 0) Y[o] = sub4(Y[o], t[o & 1]);
  }
}

struct WinoCtx {
  lds3_t lds3;
  rsrc_t wr;
  unsigned a_addr[2];        // LDS byte address of this lane's A fragment in point 0 of V, q = 0, 1
  unsigned lane_off;         // lane * 32
  unsigned ustep;            // bytes between the weights of consecutive points
};

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

// R -> y[i_h][w]: d-combine of the two planes of B^T row ID (0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), then
// the h transform.  The middle planes d1 and d2 are each needed by three of the four rows: they are read once (ID 1
// resp. ID 0) and kept in registers (P1, P2: 2 x 16 float4), so a tile's 4x4x4 patch is read from LDS exactly once
// per chunk (64 ds_read_b128 per thread instead of 128) -- LDS bandwidth, not the matrix pipe, is the busiest unit of
// these kernels (R reads + V writes + A reads + halo DMA: ~1 MB per tile and chunk at 128 B/clk).
template <int ID>
__device__ __forceinline__ void ws_transform_read(lds3_t lds3, unsigned r_base, f32x4 (&y)[4][4], f32x4 (&P1)[4][4],
                                                  f32x4 (&P2)[4][4]) {
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    f32x4 x[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      if constexpr (ID == 0) {
        const f32x4 p0 = lds_read4(lds3, r_base + (unsigned)(((0 * TH + hh) * TW + ww) * 128));
        P2[hh][ww] = lds_read4(lds3, r_base + (unsigned)(((2 * TH + hh) * TW + ww) * 128));
        x[hh] = sub4(p0, P2[hh][ww]);
      } else if constexpr (ID == 1) {
        P1[hh][ww] = lds_read4(lds3, r_base + (unsigned)(((1 * TH + hh) * TW + ww) * 128));
        x[hh] = add4(P1[hh][ww], P2[hh][ww]);
      } else if constexpr (ID == 2) {
        x[hh] = sub4(P2[hh][ww], P1[hh][ww]);
        // d2 is dead from here on: its registers take plane d3 now, in a step where this role has slack, so that the
        // last read of R happens two half-steps before the halo DMA of the next tile needs the buffer
        P2[hh][ww] = lds_read4(lds3, r_base + (unsigned)(((3 * TH + hh) * TW + ww) * 128));
      } else {
        x[hh] = sub4(P1[hh][ww], P2[hh][ww]);                 // P2 holds plane d3 (loaded by the ID 2 call)
      }
    }
    bt4(x[0], x[1], x[2], x[3], y[0][ww], y[1][ww], y[2][ww], y[3][ww]);
  }
}

template <int HH>       // w transform of rows i_h = 2 HH, 2 HH + 1 -> V[HH] (8 points)
__device__ __forceinline__ void ws_transform_write(lds3_t lds3, unsigned v_base, const f32x4 (&y)[4][4]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    f32x4 z0, z1, z2, z3;
    bt4(y[2 * HH + r][0], y[2 * HH + r][1], y[2 * HH + r][2], y[2 * HH + r][3], z0, z1, z2, z3);
    const unsigned o = (unsigned)WINO_R_BYTES + (unsigned)HH * 32768u + v_base + (unsigned)(r * 4) * 4096u;
    lds_write4(lds3, o, z0);
    lds_write4(lds3, o + 4096u, z1);
    lds_write4(lds3, o + 8192u, z2);
    lds_write4(lds3, o + 12288u, z3);
  }
}

// The transform + DMA role of the wave-specialised kernels (4 waves, tw = 0..3, tt = thread 0..255 = (tile, channel
// quad)): runs one half-step ahead of the GEMM waves, see pw_conv3d_wino.hip.  Two barriers of prologue, then 8 per
// (work item, 32-channel chunk), one per half-step; consecutive items of a block are item, item + nslots, ... < it_end.
__device__ __forceinline__ void ws_transform_role(const ConvArgs& a, const PipeArgs& p, lds3_t lds3, int item, int it_end,
                                                  int nslots, int nchunk, int tw, int tt, int lane) {
  const int tile = tt >> 3, quad = tt & 7;
  const int ttd = tile >> 4, tth = (tile >> 2) & 3, ttw = tile & 3;
  const unsigned r_base = (unsigned)((((2 * ttd) * TH + 2 * tth) * TW + 2 * ttw) * 128) + (unsigned)(quad * 16);
  const int t16 = tile & 15;
  const unsigned v_base = (unsigned)(((ttd * 16 + wino_row16(t16)) * 8 + (quad ^ (t16 & 7))) * 16);
  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  // Halo DMA of a tile and chunk: this wave moves halo rows tw + 4 K, K = 0..14 (row = d * 10 + h of the 6 x 10 x 10
  // halo), two buffer_load ... lds each (8 + 2 voxels).  What depends only on the row -- its byte offset inside the volume
  // and its (d, h) -- is computed ONCE, vectorially, lane K holding row K's values; per tile a row then costs a readlane, an
  // add and two selects next to its two DMA instructions.  (The generic pipe_dma_row recomputes ~30 scalar instructions
  // per row; in this role they sit between two barriers of a wave that shares its SIMD with an MFMA stream, and the
  // tile decode + 15 rows were the longest barrier wait of the GEMM waves: 4.7 k cycles per tile.)
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
  f32x4 y[4][4], P1[4][4], P2[4][4];
  aim(item, 0);
  dma();
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                                            // barrier A: R of the first chunk
  ws_transform_read<0>(lds3, r_base, y, P1, P2);
  ws_transform_write<0>(lds3, v_base, y);
  __syncthreads();                                            // barrier B: half-step 0 in V[0]
  for (; item < it_end; item += nslots) {
    for (int ch = 0; ch < nchunk; ++ch) {
      const bool more_ch = ch + 1 < nchunk;
      const bool has_next = more_ch || item + nslots < it_end;
      ws_transform_write<1>(lds3, v_base, y); __syncthreads();                                       // step 0
      ws_transform_read<1>(lds3, r_base, y, P1, P2); ws_transform_write<0>(lds3, v_base, y); __syncthreads();  // step 1
      ws_transform_write<1>(lds3, v_base, y); __syncthreads();                                       // step 2
      ws_transform_read<2>(lds3, r_base, y, P1, P2); ws_transform_write<0>(lds3, v_base, y); __syncthreads();  // step 3
      // step 4 (light): the next tile's decode is done here, off the DMA step's critical path -- a wave that shares its SIMD
      // with an MFMA stream issues roughly one instruction per 14 cycles, so WHERE its instructions sit decides who waits
      ws_transform_write<1>(lds3, v_base, y);
      if (has_next) aim(more_ch ? item : item + nslots, more_ch ? ch + 1 : 0);
      __syncthreads();
      // step 5: R was last read in step 3 (plane d3 is prefetched there), so the next chunk's / tile's halo DMA goes out
      // now and has steps 5 and 6 (~6 k cycles; it needs ~3.5 k) to land before the step-6 barrier publishes it
      if (has_next) dma();
      ws_transform_read<3>(lds3, r_base, y, P1, P2); ws_transform_write<0>(lds3, v_base, y);
      // end of step 5 WITHOUT draining the DMA: __syncthreads() would (correctly, for a release fence) wait for vmcnt
      // because the in-flight buffer_load ... lds are LDS writes; only this wave's ds_writes must have landed here
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ws_transform_write<1>(lds3, v_base, y);                                                        // step 6
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (has_next) {                                                                                // step 7
        ws_transform_read<0>(lds3, r_base, y, P1, P2);
        ws_transform_write<0>(lds3, v_base, y);
      }
      __syncthreads();
    }
  }
}
