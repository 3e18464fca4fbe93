// Shared pieces of the conv kernels (pw_conv3d.hip, pw_conv3d_gather.hip, pw_occ_head.hip, pw_fpn3d.hip):
// ConvArgs, buffer addressing, the LDS halo-tile layout and its two staging paths, the MFMA tap bodies,
// work-item decoding of the persistent kernels, host-side launch helpers.
#ifndef PW_CONV3D_COMMON_H_
#define PW_CONV3D_COMMON_H_
#include "pw_common.h"

#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BD = 4, BH = 8, BW = 8;
constexpr int TD = BD + 2, TH = BH + 2, TW = BW + 2;
constexpr int TV = TD * TH * TW;                 // 600 halo voxels
constexpr unsigned WPIECE = 1024u;               // bytes per piece of a packed weight tile (4 pieces x 64 lanes x 16 B)
constexpr int KC = 32;                           // input channels per LDS chunk
}  // namespace

struct ConvArgs {
  const float* x;
  const float* wpk;       // packed weights [Cin/32][taps][cout_total/32][64 lanes][16]
  const float* scale;     // [cout_total] or null (=1)
  const float* bias;      // [cout_total] or null (=0)
  const float* residual;  // same layout as y0, or null
  float* y0;
  float* y1;              // second destination (columns >= n1_start) or null
  int B, D, H, W, Cin;    // input dims
  int Do, Ho, Wo;         // output dims
  int cout_total;         // multiple of 32
  int cout0, cout1;       // real channel counts of y0 / y1
  int ld0, ld1;           // row stride (floats per voxel) of y0 (and residual) / y1: >= cout, a channel
                          // slice of a wider channels-last buffer when larger (no concat copy)
  int n1_start;           // first packed column that goes to y1
  int relu0, relu1;
  int tiles_d, tiles_h, tiles_w;
  long long* probe;       // development aid: per-wave phase timestamps (PW_CONV_PROBE) or null
  int dma_stage;          // pw_fpn3d.hip: input-format flag (x8 and wpk8 are split-fp16)
  int fmt_y0, fmt_y1, fmt_res;  // split-fp16 kernels (pw_h2.h): 0 = fp32, 1 = h2 storage of y0 / y1 / residual
  // range slots of the h2 operands (pw_h2.h "Range"; null = exponent 0, nothing recorded): x / residual are read under
  // x_rng[0] / res_rng[0], y0 / y1 are written under y*_rng[0] and their largest magnitude is recorded in y*_rng[1]
  const int* x_rng;
  const int* res_rng;
  int* y0_rng;
  int* y1_rng;
};

// per-lane epilogue state of the split-fp16 kernels that write through store_out
struct RngEpi {
  float res_mul;          // residual as stored -> units y0 is stored in
  float amax0, amax1;     // largest |stored value| this lane wrote to y0 / y1
};

// MFMA row (0..31) -> voxel of the 4x8 patch, chosen for conflict-free ds_read_b128 groups
__device__ __forceinline__ int patch_of_row(int i) {
  int g = i >> 2;
  int set = (0x96 >> g) & 1;
  return set * 16 + (g >> 1) * 4 + (i & 3);
}

// byte offset of the hi half of channel n inside a channels-last row in h2 storage (pw_h2.h); the lo half sits 16 bytes on
__device__ __forceinline__ size_t h2_elem_off(int n) {
  const int c = n & 31;
  return (size_t)(n >> 5) * 128 + (size_t)((4 * ((c >> 3) & 1) + 2 * (c >> 4)) * 16 + 2 * (c & 7));
}
__device__ __forceinline__ void h2_store_elem(float* row, int n, float v) {
  char* p = reinterpret_cast<char*>(row) + h2_elem_off(n);
  const _Float16 hi = (_Float16)v;                       // unsaturated: see h2_split1 (pw_h2.h)
  *reinterpret_cast<_Float16*>(p) = hi;
  *reinterpret_cast<_Float16*>(p + 16) = (_Float16)__builtin_amdgcn_fmed3f(v - (float)hi, -65504.f, 65504.f);
}
__device__ __forceinline__ float h2_load_elem(const float* row, int n) {
  const char* p = reinterpret_cast<const char*>(row) + h2_elem_off(n);
  return (float)*reinterpret_cast<const _Float16*>(p) + (float)*reinterpret_cast<const _Float16*>(p + 16);
}

__device__ __forceinline__ void store_out(const ConvArgs& a, int n, size_t vox, float v, RngEpi* re = nullptr) {
  // n = packed output column; fmt_* = 0 (fp32) in the fp32 kernels (ConvArgs zero-initialised there).  re (split-fp16 kernels):
  // v arrives in the units its destination is stored in; the residual is rescaled into them and the magnitude recorded.
  if (n < a.cout0) {
    if (a.residual) {
      const float r = a.fmt_res ? h2_load_elem(a.residual + vox * a.ld0, n) : a.residual[vox * a.ld0 + n];
      v += re ? r * re->res_mul : r;
    }
    if (a.relu0) v = fmaxf(v, 0.f);
    if (re) re->amax0 = fmaxf(re->amax0, fabsf(v));
    if (a.fmt_y0) h2_store_elem(a.y0 + vox * a.ld0, n, v);
    else a.y0[vox * a.ld0 + n] = v;
  } else {
    int n1 = n - a.n1_start;
    if (a.y1 && n1 >= 0 && n1 < a.cout1) {
      if (a.relu1) v = fmaxf(v, 0.f);
      if (re) re->amax1 = fmaxf(re->amax1, fabsf(v));
      if (a.fmt_y1) h2_store_elem(a.y1 + vox * a.ld1, n1, v);
      else a.y1[vox * a.ld1 + n1] = v;
    }
  }
}

// ------------------------------------------------------------------------------------
// VALU budget.  On gfx950 the fp32-input MFMA executes on the SIMD's fp32 vector datapath:
// every VALU instruction of ANY wave on the SIMD displaces matrix work (measured on the first
// version of this kernel: 2.3k VALU instructions per wave -> 31 % of the MFMA issue slots idle;
// phase timestamps showed the 19 staging loads taking 31k cycles just to ISSUE next to an
// MFMA-streaming partner wave, s_setprio made no difference).  So everything around the MFMAs
// is written to need (almost) no vector ALU:
//   * staging walks the halo tile by ROWS that are wave-uniform (wave w takes rows w, w+4, ..):
//     row decode, bounds tests and the 64-bit global address are scalar; a lane only adds a
//     precomputed 32-bit offset (saddr-form global_load) and one LDS address add;
//   * the 27 taps are fully unrolled and the swizzled LDS read addresses are precomputed per lane
//     for the 6 (tap-row parity, kw) variants, so a tap's ds_read_b128 is base + immediate;
//   * weights are read through a scalar base that the scalar ALU advances per tap;
//   * the epilogue uses a scalar destination base + one 32-bit mad per element, and skips all
//     bounds tests on interior tiles.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Buffer (SRD) addressing: descriptor + scalar byte offset + 32-bit lane offset -> the address
// arithmetic of every load/store is scalar; no 64-bit VALU adds (see "VALU budget" above).
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, unsigned voff, unsigned soff) {
  // NB: keep `auto` -- converting the builtin's vector to an ext_vector_type makes hipcc (ROCm 7.2)
  // load only the first dword
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                     __uint_as_float(v[3]));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// WD = number of 4-deep d-groups per block: WD=1 -> 4 waves, tile 4x8x8, 76.8 KB LDS, 2 blocks/CU;
// WD=2 -> 8 waves, tile 8x8x8, 128 KB LDS, ONE block per CU.  Measured with per-phase timestamps:
// next to a wave that streams fp32 MFMAs, every VGPR-reading instruction of the partner wave
// (VALU, VMEM, DS) is starved -- the 32-store epilogue takes 3k cycles alone but 51k beside an
// MFMA stream, staging 12k vs 60k.  With two independent blocks per CU the waves sharing a SIMD
// drift into anti-phase and the non-MFMA phases crawl; with one 8-wave block the block's own
// barriers keep both waves of every SIMD in the SAME phase: staging and epilogue run at full
// speed, and during the taps the two waves hide each other's LDS/weight-load latency.
template <int WD> struct TileGeom {
  static constexpr int BDt = 4 * WD, TDt = BDt + 2, ROWS = TDt * TH, NW = 4 * WD;
  static constexpr int ROWS_PER_WAVE = (ROWS + NW - 1) / NW;      // 15 (WD=1) / 13 (WD=2)
  static constexpr int LDS = TDt * TH * TW * KC * 4;              // 76800 / 128000 bytes
};

template <int NT>
__device__ __forceinline__ void load_b(rsrc_t wr, unsigned wsoff, unsigned lane_off, float4 (&b)[NT][4]) {
  // wsoff: wave-uniform byte offset of this (chunk, tap, N-group) tile; lane_off = lane*16 bytes.  A 4096-byte weight tile is
  // four 1024-byte pieces of 64 lanes x 16 B (WPIECE): one instruction reads 1 KB of consecutive bytes = 8 cache lines, where
  // the 64-bytes-per-lane order of rounds 1-2 touched 32 lines per instruction (the vector L1 looks up one line per cycle)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) b[nt][q] = buf_load4(wr, lane_off + (unsigned)q * WPIECE, wsoff + (unsigned)(nt * 4096));
}

// one tap: 2 M-tiles x NT N-tiles x 16 k-steps of v_mfma_f32_32x32x2_f32.
// aaddr[khp][kw][q]: precomputed swizzled LDS byte address of this lane's voxel for tap-row
// parity khp and column shift kw; the rest of the tap offset is a compile-time immediate.
template <int NT, int TAP>
__device__ __forceinline__ void tap_mfma(const float* lds, const unsigned (&aaddr)[2][3][4],
                                         const float4 (&b)[NT][4], f32x16 (&acc)[2][NT]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
  const char* ldsb = reinterpret_cast<const char*>(lds);
  // all 8 A reads of the tap go out first (pinned by the sched_barrier): the LDS latency is then
  // paid once per tap under the previous tap's trailing MFMAs instead of before every 4 MFMAs
  float4 aq[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const unsigned imm = (unsigned)(((kd * TH + mt * 4 + kh) * TW) * 128);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      aq[mt][q] = *reinterpret_cast<const float4*>(ldsb + (aaddr[kh & 1][kw][q] + imm));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[4] = {aq[mt][q].x, aq[mt][q].y, aq[mt][q].z, aq[mt][q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv[4] = {b[nt][q].x, b[nt][q].y, b[nt][q].z, b[nt][q].w};
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
}

// taps TAP, TAP+1 with the weight ping-pong; recursion unrolls all 27 taps at compile time
template <int NT, int TAP>
__device__ __forceinline__ void tap_pair(const float* lds, const unsigned (&aaddr)[2][3][4],
                                         rsrc_t wr, unsigned wsoff, unsigned lane_off, unsigned wstride,
                                         float4 (&b0)[NT][4], float4 (&b1)[NT][4],
                                         f32x16 (&acc)[2][NT]) {
  if constexpr (TAP + 1 < 27) {
    load_b<NT>(wr, wsoff + (unsigned)(TAP + 1) * wstride, lane_off, b1);
    __builtin_amdgcn_sched_barrier(0);
    tap_mfma<NT, TAP>(lds, aaddr, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_b<NT>(wr, wsoff + (unsigned)(TAP + 2) * wstride, lane_off, b0);
    __builtin_amdgcn_sched_barrier(0);
    tap_mfma<NT, TAP + 1>(lds, aaddr, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    tap_pair<NT, TAP + 2>(lds, aaddr, wr, wsoff, lane_off, wstride, b0, b1, acc);
  } else {
    tap_mfma<NT, TAP>(lds, aaddr, b0, acc);
  }
}

// (pr, pc) patch position of accumulator register r for lane half h -- see patch_of_row
__device__ __forceinline__ constexpr int acc_patch(int r, int h) {
  const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
  const int g = i >> 2;
  const int set = (0x96 >> g) & 1;
  return set * 16 + (g >> 1) * 4 + (i & 3);
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2).  Remap the linear
// block id so that XCD x works on a CONTIGUOUS range of tiles: neighbouring tiles share halo
// voxels, and with the plain order every halo line was fetched into several L2s (PMC FETCH_SIZE:
// 2.1x the input tensor per launch; 1.4x with contiguous ranges).
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk) {
  const int x = bid & 7, idx = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  return x * q + min(x, r) + idx;
}

// ---- halo staging by `buffer_load ... lds` (shared by the tile-per-block and the persistent kernels)
typedef __attribute__((address_space(3))) char* lds3_t;
constexpr unsigned PIPE_OOB = 0xfffffff0u;     // voffset beyond any num_records -> load returns 0
constexpr int PIPE_BUF_BYTES = TV * KC * 4;     // 76800
constexpr int PIPE_ROWS_PER_WAVE = TD * TH / 4; // 15 halo rows per wave and stage

struct PipeDma {                                 // what the DMA of one stage needs
  unsigned voff[2][2];                           // [halo-row parity][pass] lane offset or PIPE_OOB
  int b, d0, h0, wbase, ch;                      // scalars
  unsigned ldsbuf;                               // byte offset of the destination buffer
  bool live;                                     // false: no next stage, every lane goes OOB
};

// lane offsets for a tile column position w0 (see stage_lane_setup for the w0 == 0 shift);
// pass 0 = voxels 0..7 as 16-byte slots, pass 1 = voxels 8..9 as dwords
__device__ __forceinline__ void pipe_lane_offsets(const ConvArgs& a, int w0, int lane, unsigned (&voff)[2][2]) {
  const int shift = w0 == 0 ? 1 : 0;
  {
    const int ww = lane >> 3, slot = lane & 7;
    const bool wok = (unsigned)(w0 - 1 + ww) < (unsigned)a.W;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int f = ((ww >> 1) & 3) | (par << 2);
      voff[par][0] = wok ? (unsigned)((ww - shift) * a.Cin + (slot ^ f) * 4) * 4u : PIPE_OOB;
    }
  }
  {
    const int ww = 8 + (lane >> 5), dw = lane & 31, slot = dw >> 2;
    const bool wok = (unsigned)(w0 - 1 + ww) < (unsigned)a.W;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int f = ((ww >> 1) & 3) | (par << 2);
      voff[par][1] = wok ? (unsigned)((ww - shift) * a.Cin + (slot ^ f) * 4 + (dw & 3)) * 4u : PIPE_OOB;
    }
  }
}

// halo row `wave + 4 K` of the stage described by dm: two DMA instructions (8 + 2 voxels)
template <int K>
__device__ __forceinline__ void pipe_dma_row(const ConvArgs& a, rsrc_t xr, lds3_t lds3, const PipeDma& dm,
                                             int wave) {
  const int row = wave + 4 * K;                    // wave-uniform, < 60
  const int dd = row / TH, hh = row - dd * TH;
  const int gd = dm.d0 + dd - 1, gh = dm.h0 + hh - 1;
  const bool rok = dm.live && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H;
  const unsigned soff = rok ? (unsigned)(((((dm.b * a.D + gd) * a.H + gh) * a.W + dm.wbase) * a.Cin + dm.ch * KC) * 4) : 0u;
  const unsigned v0 = rok ? ((hh & 1) ? dm.voff[1][0] : dm.voff[0][0]) : PIPE_OOB;
  const unsigned v1 = rok ? ((hh & 1) ? dm.voff[1][1] : dm.voff[0][1]) : PIPE_OOB;
  lds3_t dst = lds3 + (dm.ldsbuf + (unsigned)row * (TW * 128));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, v0, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 4, v1, soff, 0, 0);
}


// tile-per-block kernels: the whole halo of one chunk as 30 DMA instructions per wave instead of the
// ~100-instruction global -> VGPR -> swizzled ds_write sequence (which crawls next to an MFMA-streaming
// sibling block, section "VALU budget")
__device__ __forceinline__ void stage_halo_chunk_dma(const ConvArgs& a, rsrc_t xr, float* lds, int b, int d0,
                                                     int h0, int w0, int ch, int wave, int lane) {
  if (ch > 0) __syncthreads();   // every wave finished reading the previous chunk
  PipeDma dm;
  pipe_lane_offsets(a, w0, lane, dm.voff);
  dm.b = b; dm.d0 = d0; dm.h0 = h0; dm.wbase = w0 > 0 ? w0 - 1 : 0; dm.ch = ch; dm.ldsbuf = 0; dm.live = true;
  const lds3_t lds3 = (lds3_t)lds;
  pipe_dma_row<0>(a, xr, lds3, dm, wave); pipe_dma_row<1>(a, xr, lds3, dm, wave); pipe_dma_row<2>(a, xr, lds3, dm, wave);
  pipe_dma_row<3>(a, xr, lds3, dm, wave); pipe_dma_row<4>(a, xr, lds3, dm, wave); pipe_dma_row<5>(a, xr, lds3, dm, wave);
  pipe_dma_row<6>(a, xr, lds3, dm, wave); pipe_dma_row<7>(a, xr, lds3, dm, wave); pipe_dma_row<8>(a, xr, lds3, dm, wave);
  pipe_dma_row<9>(a, xr, lds3, dm, wave); pipe_dma_row<10>(a, xr, lds3, dm, wave); pipe_dma_row<11>(a, xr, lds3, dm, wave);
  pipe_dma_row<12>(a, xr, lds3, dm, wave); pipe_dma_row<13>(a, xr, lds3, dm, wave); pipe_dma_row<14>(a, xr, lds3, dm, wave);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
}


// ---- persistent kernels: work items
struct PipeArgs {
  unsigned m_ng, m_tw, m_th, m_td;   // floor(2^32 / d) + 1 for exact x / d by mulhi (x * d < 2^32)
  int ngroups, n_items;
};


__device__ __forceinline__ int udiv_magic(int x, int d, unsigned magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

struct PipeTile { int b, d0, h0, w0, ng; };

__device__ __forceinline__ PipeTile pipe_decode(const ConvArgs& a, const PipeArgs& p, int item) {
  PipeTile t;
  int tile = udiv_magic(item, p.ngroups, p.m_ng);
  t.ng = item - tile * p.ngroups;
  int q = udiv_magic(tile, a.tiles_w, p.m_tw);
  t.w0 = (tile - q * a.tiles_w) * BW; tile = q;
  q = udiv_magic(tile, a.tiles_h, p.m_th);
  t.h0 = (tile - q * a.tiles_h) * BH; tile = q;
  q = udiv_magic(tile, a.tiles_d, p.m_td);
  t.d0 = (tile - q * a.tiles_d) * BD;
  t.b = q;
  return t;
}


// ---- host-side helpers
template <typename K>
static int set_lds_limit(K kernel, int bytes) {
  PW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PW_OK;
}

// exact x / d by one mulhi for x * d < 2^32 (tile counts): floor(2^32 / d) + 1
static inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

static inline int pw_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n < 8) n = 256;
  }
  return n;
}

// Persistent DMA-pipelined kernel or tile-per-block kernel (algo 4 / 1 of pw_conv3d_ndhwc force one).  Sustained timings (1 s
// loops, clocks settled at 2.39 GHz, tools/bench_layers.py), tile-per-block vs pipelined:
//   16x200x200 32->32  277.9 / 278.4 us    32->64  518.3 / 537.6    64->64 1002.8 / 1049.3
//   8x100x100  64->64  180.4 / 165.6 us    64->128 308.0 / 319.5    4x50x50 128->128 110.8 / 110.4
// Both designs sit on the same ceiling (operand loads cost matrix-pipe time, see the kernel comment);
// the pipelined one wins where a wave owns a single N-tile and the grid gives every CU 2+ items.
static inline bool use_pipe(long long n_items, int NT, int algo) {
  if (algo == 1 || algo == 4) return algo == 4 && n_items < (1ll << 20);
  return NT == 1 && n_items >= 512 && n_items < (1ll << 20);
}

// gather kernel launcher (pw_conv3d_gather.hip)
int pw_launch_conv3d_gather(const ConvArgs& a, int NT, int ngroups, int ksize, int stride, int algo, int Cin,
                             long long n_out, hipStream_t st, bool f16 = false);
// LDS-tiled split-fp16 stride-2 kernel (pw_conv3d_h2_s2.hip); PW_EUNSUP when the shape / formats are not built
int pw_launch_conv3d_h2_s2(const ConvArgs& a, hipStream_t st);

#endif  // PW_CONV3D_COMMON_H_
