// 3-D convolution for the voxel encoder / heads on gfx950: im2col-free implicit GEMM on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  Replaces the cuDNN Conv3d + BatchNorm3d
// + ReLU (+ residual) chains the reference builds in
//   mmdet3d/models/backbones/resnet.py:88-184   (BasicBlock3D / CustomResNet3D)
//   mmdet3d/models/necks/lss_fpn.py:103-148     (LSSFPN3D 1x1x1)
//   mmdet3d/models/detectors/preworld.py:72-79  (final_conv)
//   mmdet3d/models/heads/occupancy_head.py:80-105 (OccHead convs)
//
// Layout: activations are channels-last (B, D, H, W, C) fp32 in HBM; the module layer hands
// out (B, C, D, H, W) *views* of these buffers, so no permute copy ever runs.
//
// GEMM view: M = output voxels, N = output channels, K = taps x input channels.
//   MFMA 32x32x2: lane l supplies A[row=l&31][k=l>>5] and B[k=l>>5][col=l&31];
//   D: lane holds column l&31, rows (reg&3) + 8*(reg>>2) + 4*(l>>5).
// K is ordered so that within one (channel-chunk, tap) the lane half h = l>>5 walks input
// channels h*16 .. h*16+15: each lane then reads 16 CONTIGUOUS floats of its voxel
// (4 x ds_read_b128) and 16 contiguous packed weights (4 x global_load_dwordx4).
//
// k3 stride-1 kernel: a 256-thread block computes a 4(d) x 8(h) x 8(w) output tile
// (8 M-tiles of 4x8 voxels, 2 per wave) from a 6x10x10 halo tile of 32 input channels
// staged in LDS (76.8 KB -> 2 blocks per CU).  The 16-byte slots of a voxel are XOR-swizzled
// with ((w>>1)&3 | (h&1)<<2) and MFMA rows are mapped to patch voxels so that every
// ds_read_b128 lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} touches 16 distinct
// slots of the 256-B bank row: conflict-free A reads.
// Epilogue: y = acc * scale[n] + bias[n] (+ residual) (ReLU) with BatchNorm(eval) folded into
// scale/bias by the caller; two destination tensors are supported so that BasicBlock3D's
// conv1 and downsample (same input) run as ONE pass over the input with N = 2 x Cout.
#include "pw_common.h"

#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BD = 4, BH = 8, BW = 8;
constexpr int TD = BD + 2, TH = BH + 2, TW = BW + 2;
constexpr int TV = TD * TH * TW;                 // 600 halo voxels
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
  int dma_stage;          // tile-per-block kernels: stage the halo with buffer_load ... lds
};

// MFMA row (0..31) -> voxel of the 4x8 patch, chosen for conflict-free ds_read_b128 groups
__device__ __forceinline__ int patch_of_row(int i) {
  int g = i >> 2;
  int set = (0x96 >> g) & 1;
  return set * 16 + (g >> 1) * 4 + (i & 3);
}

__device__ __forceinline__ void store_out(const ConvArgs& a, int n, size_t vox, float v) {
  // n = packed output column
  if (n < a.cout0) {
    size_t o = vox * a.ld0 + n;
    if (a.residual) v += a.residual[o];
    if (a.relu0) v = fmaxf(v, 0.f);
    a.y0[o] = v;
  } else {
    int n1 = n - a.n1_start;
    if (a.y1 && n1 >= 0 && n1 < a.cout1) {
      if (a.relu1) v = fmaxf(v, 0.f);
      a.y1[vox * a.ld1 + n1] = v;
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

// per-lane constants of the staging pattern: a halo row is 10 voxels x 8 slots = 80 float4;
// pass 0 covers voxels 0..7 (64 lanes), pass 1 voxels 8..9 (lanes 0..15)
struct StageLane {
  unsigned goff[2];        // global BYTE offset inside a row: (ww*Cin + slot*4)*4
  unsigned loff[2][2];     // LDS byte offset inside a row, [hh parity][pass], swizzle applied
  bool wok[2];             // w0-1+ww inside [0,W)
};

__device__ __forceinline__ StageLane stage_lane_setup(const ConvArgs& a, int w0, int lane) {
  StageLane s;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int ww = ps * 8 + (lane >> 3), slot = lane & 7;
    // offsets are relative to voxel max(w0-1, 0) of the row: buffer soffset/voffset are UNSIGNED,
    // so the "-1 voxel" of the halo cannot be expressed as a negative scalar offset at w0 = 0
    // (there lane ww = 0 is masked by wok and its wrapped offset is never used)
    s.goff[ps] = (unsigned)((ww - (w0 == 0 ? 1 : 0)) * a.Cin + slot * 4) * 4u;      // bytes
    s.wok[ps] = (unsigned)(w0 - 1 + ww) < (unsigned)a.W && (ps == 0 || lane < 16);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int f = ((ww >> 1) & 3) | (par << 2);
      s.loff[par][ps] = (unsigned)((ww * 8 + (slot ^ f)) * 16);
    }
  }
  return s;
}

// stage the 6x10x10 halo tile of one 32-channel chunk: global -> registers -> swizzled LDS
template <int WD, int KB0>
__device__ __forceinline__ void stage_halo_chunk(const ConvArgs& a, rsrc_t xr, float* lds,
                                                 const StageLane& sl, int b, int d0, int h0, int w0,
                                                 int ch, int wave, int lane) {
  using G = TileGeom<WD>;
  char* ldsb = reinterpret_cast<char*>(lds);
  if (ch > 0) __syncthreads();   // every wave finished reading the previous chunk
  // KB0 rows per batch (all loads of a batch are issued before its LDS writes): 8 rows = 64 VGPRs
  // for the NT=1 kernels, 4 rows where the 2 x 32-wide accumulators leave fewer registers
  constexpr int NBATCH = (G::ROWS_PER_WAVE + KB0 - 1) / KB0;
#pragma unroll
  for (int batch = 0; batch < NBATCH; ++batch) {
    const int k0 = batch * KB0, k1 = (batch + 1) * KB0 < G::ROWS_PER_WAVE ? (batch + 1) * KB0 : G::ROWS_PER_WAVE;
    float4 tmp[KB0][2];
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int row = wave + G::NW * k;                 // wave-uniform (wave comes from readfirstlane)
      const int dd = row / TH, hh = row - dd * TH;
      const int gd = d0 + dd - 1, gh = h0 + hh - 1;
      const bool rok = row < G::ROWS && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H;
      // scalar byte offset of voxel max(w0-1, 0) of this row (see stage_lane_setup)
      const unsigned soff = (unsigned)((((((long long)b * a.D + gd) * a.H + gh) * a.W + (w0 > 0 ? w0 - 1 : 0)) * a.Cin + ch * KC) * 4);
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok && sl.wok[ps]) v = buf_load4(xr, sl.goff[ps], soff);
        tmp[k - k0][ps] = v;
      }
    }
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int row = wave + G::NW * k;
      const int dd = row / TH, hh = row - dd * TH;
      const unsigned rofs = (unsigned)row * (TW * 128);
      if (row < G::ROWS) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const unsigned lo = (hh & 1) ? sl.loff[1][ps] : sl.loff[0][ps];
          if (ps == 0 || lane < 16)
            *reinterpret_cast<float4*>(ldsb + rofs + lo) = tmp[k - k0][ps];
        }
      }
    }
  }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void load_b(rsrc_t wr, unsigned wsoff, unsigned lane_off, float4 (&b)[NT][4]) {
  // wsoff: wave-uniform byte offset of this (chunk, tap, N-group); lane_off = lane*64 bytes
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) b[nt][q] = buf_load4(wr, lane_off + (unsigned)(q * 16), wsoff + (unsigned)(nt * 4096));
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

// ------------------------------------------------------------------------------------
// 3x3x3, stride 1, pad 1, LDS halo tile
// ------------------------------------------------------------------------------------
template <int NT, int WD>
__global__ void __launch_bounds__(256 * WD, 2) k_conv3d_k3s1(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int half = lane >> 5, i = lane & 31;
  const int ng = blockIdx.y;
  const int pj = patch_of_row(i), pr = pj >> 3, pc = pj & 7;
  const int ntiles_total = a.cout_total >> 5;

  // swizzled LDS read addresses (bytes) of this lane's patch voxel, 6 variants x 4 slots
  unsigned aaddr[2][3][4];
#pragma unroll
  for (int khp = 0; khp < 2; ++khp)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = pc + kw;
      const int f = ((ww >> 1) & 3) | (((pr + khp) & 1) << 2);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        aaddr[khp][kw][q] = (unsigned)((((wave * TH + pr) * TW + ww) * 8 + ((half * 4 + q) ^ f)) * 16);
    }
  const unsigned lane_off = (unsigned)lane * 64u;
  const unsigned wstride = (unsigned)ntiles_total * 4096u;           // bytes per tap
  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  const rsrc_t wr = make_rsrc(a.wpk, (unsigned)((size_t)(a.Cin / KC) * 27 * ntiles_total * 4096));

  // (A persistent tile loop and a one-block-per-CU 8-wave variant were both measured: neither
  // moved the needle -- the remaining gap of the NT=1 kernel is the exposed halo-load latency.)
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int tw = bid % a.tiles_w; bid /= a.tiles_w;
  const int th = bid % a.tiles_h; bid /= a.tiles_h;
  const int td = bid % a.tiles_d;
  const int b = bid / a.tiles_d;
  const int d0 = td * (BD * WD), h0 = th * BH, w0 = tw * BW;
  const StageLane sl = stage_lane_setup(a, w0, lane);

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nchunk = a.Cin / KC;
  long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (a.probe) ts0 = __builtin_readcyclecounter();
  for (int ch = 0; ch < nchunk; ++ch) {
    // first tap's weights go out before the halo loads so they are not queued behind them
    const unsigned wsoff = (unsigned)((ch * 27 * ntiles_total + ng * NT) * 4096);
    float4 b0[NT][4], b1[NT][4];
    load_b<NT>(wr, wsoff, lane_off, b0);
    if (WD == 1 && a.dma_stage) stage_halo_chunk_dma(a, xr, lds, b, d0, h0, w0, ch, wave, lane);
    else stage_halo_chunk<WD, 8>(a, xr, lds, sl, b, d0, h0, w0, ch, wave, lane);
    if (a.probe && ch == 0) ts1 = __builtin_readcyclecounter();
    tap_pair<NT, 0>(lds, aaddr, wr, wsoff, lane_off, wstride, b0, b1, acc);
  }
  if (a.probe) ts2 = __builtin_readcyclecounter();

  // ---- epilogue: y = acc*scale + bias (+residual) (ReLU).  Destination, residual and the
  // M-tile origin are scalar (buffer descriptor + soffset); a lane adds one 32-bit offset per
  // element.  Interior tiles with full 32-column N-tiles take a branch-free path: all residual
  // loads first, then the math, then the stores.
  const int od = d0 + wave;
  if (od < a.Do) {
  const bool interior = h0 + BH <= a.Ho && w0 + BW <= a.Wo;
  const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = (ng * NT + nt) * 32;               // first packed column of this N-tile (uniform)
    const int n = n0 + i;
    const bool to_y0 = n0 < a.cout0;
    float* dst = to_y0 ? a.y0 : a.y1;
    if (dst == nullptr) continue;
    const int stride = to_y0 ? a.cout0 : a.cout1;      // valid columns
    const int ld = to_y0 ? a.ld0 : a.ld1;              // row stride
    const int col0 = to_y0 ? n0 : n0 - a.n1_start;    // uniform
    if (col0 < 0 || col0 >= stride) continue;
    const bool fullcols = col0 + 32 <= stride;
    const float lo = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
    const bool has_res = to_y0 && a.residual != nullptr;
    const rsrc_t yr = make_rsrc(dst, out_vox * (unsigned)ld * 4u);
    const rsrc_t rr = make_rsrc(has_res ? a.residual : dst, out_vox * (unsigned)ld * 4u);
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float bi = a.bias ? a.bias[n] : 0.f;
    const unsigned lanecol = (unsigned)(col0 + i) * 4u;
    // Accumulator register r = 4g+e of lane half h sits at patch (row pr(g,h), column pc0(g,h)+e):
    // one per-lane byte offset per group g; the +e column step and the M-tile origin are scalar
    // (buffer soffset), so no per-element address VALU at all.
    unsigned goff[4];
    int gpr[4], gpc[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int p0 = acc_patch(4 * g4, 0), p1 = acc_patch(4 * g4, 1);
      gpr[g4] = half ? (p1 >> 3) : (p0 >> 3);
      gpc[g4] = half ? (p1 & 7) : (p0 & 7);
      goff[g4] = (unsigned)((gpr[g4] * a.Wo + gpc[g4]) * ld) * 4u + lanecol;
    }
    const unsigned estep = (unsigned)ld * 4u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      // byte offset of patch position (0,0) of this M-tile: uniform
      const unsigned soff = (unsigned)((((((long long)b * a.Do + od) * a.Ho + (h0 + mt * 4)) * a.Wo + w0) * ld) * 4);
      if (interior && fullcols) {
        float rv[16];
        if (has_res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = buf_load1(rr, goff[r >> 2], soff + (unsigned)(r & 3) * estep);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[mt][nt][r] * sc + bi;
          if (has_res) v += rv[r];
          buf_store1(yr, goff[r >> 2], soff + (unsigned)(r & 3) * estep, fmaxf(v, lo));
        }
      } else {
        const bool colok = col0 + i < stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (colok && (h0 + mt * 4 + gpr[r >> 2] < a.Ho) && (w0 + gpc[r >> 2] + (r & 3) < a.Wo)) {
            float v = acc[mt][nt][r] * sc + bi;
            const unsigned so = soff + (unsigned)(r & 3) * estep;
            if (has_res) v += buf_load1(rr, goff[r >> 2], so);
            buf_store1(yr, goff[r >> 2], so, fmaxf(v, lo));
          }
        }
      }
    }
  }
  }   // od < Do
  if (a.probe && lane == 0) {
    long long* p = a.probe + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (4 * WD) + wave) * 4;
    p[0] = ts0; p[1] = ts1; p[2] = ts2; p[3] = __builtin_readcyclecounter();
  }
}

// ------------------------------------------------------------------------------------
// 3x3x3 stride 1, PERSISTENT + DMA-PIPELINED variant (the default for large grids).
//
// Measured on gfx950 (tools/probes/mfma_shadow.hip): nothing is free next to fp32 MFMAs.  Every
// ds_read_b128 / buffer_load_dwordx4 a SIMD issues costs ~15 ticks of matrix-pipe time and a VALU
// instruction ~3-6, whether it comes from the MFMA wave or from a sibling; and instructions of a
// sibling wave only issue when the MFMA wave stalls, so the tile-per-block kernel above runs its
// halo staging and its MFMA phases back to back (~15-20 % of the time not on the matrix pipe).
// Hence:
//   * the halo is staged by `buffer_load_dword(x4) ... lds`: the load unit writes LDS directly --
//     no data VGPRs, no ds_write, 2 instructions per halo row instead of ~10 -- and the MFMA wave
//     itself issues them, one halo row of the NEXT stage per tap;
//   * one 4-wave block per CU, persistent over a contiguous, XCD-local range of work items
//     (tile x N-group); LDS = two 76.8 KB halo buffers, stage s computes from buffer s&1 while
//     the DMA for stage s+1 fills the other; ONE barrier per stage;
//   * a wave owns 2 M-tiles x NT N-tiles (weights are reused across the M-tiles, A fragments
//     across the N-tiles: 0.375 / 0.25 operand loads per MFMA for NT = 1 / 2); A fragments of
//     tap t+1 and weights of tap t+2 are requested before the MFMAs of tap t;
//   * the XOR swizzle moves to the global side: the DMA writes lane L's bytes at LDS row base +
//     16 L (4 L for the dword form), so lane (ww, slot') fetches channel slot slot' ^ f(ww, hh)
//     of voxel ww -- the lanes of a voxel still cover its whole 128-byte line;
//   * a halo row is 8 + 2 voxels: one dwordx4 DMA (64 lanes x 16 B) and one dword DMA
//     (64 lanes x 4 B), every lane active, no EXEC games, no branches inside a tap;
//   * out-of-volume halo voxels (and "no next stage"): the lane's buffer offset is forced out of
//     range, the load unit returns 0 and the zero lands in LDS (tools/probes/dma_probe.hip).
// ------------------------------------------------------------------------------------
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

template <int NT>
struct PipeCtx {
  lds3_t lds3;
  rsrc_t xr, wr;
  unsigned lane_off, wstride;
  unsigned wsoff;              // this stage's (chunk, N-group) weight base
  unsigned wsoff_next;         // next stage's
  bool has_next;
  PipeDma dm;                  // next stage's DMA description
  int wave, lane;
  long long* tap_probe;
};

template <int TAP>
__device__ __forceinline__ void pipe_read_a_tap(lds3_t lds3, const unsigned (&aaddr)[2][3][4], float4 (&aq)[2][4]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
  typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    constexpr unsigned imm0 = (unsigned)(((kd * TH + kh) * TW) * 128);
    const unsigned imm = imm0 + (unsigned)(mt * 4 * TW * 128);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f v = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(lds3 + aaddr[kh & 1][kw][q] + imm);
      aq[mt][q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <int NT>
__device__ __forceinline__ void pipe_mfma(const float4 (&aq)[2][4], const float4 (&b)[NT][4], f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float av[4] = {aq[mt][q].x, aq[mt][q].y, aq[mt][q].z, aq[mt][q].w};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv[4] = {b[nt][q].x, b[nt][q].y, b[nt][q].z, b[nt][q].w};
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
}

// Tap TAP computes from (ac, b0).  Before its MFMAs it requests the weights of tap TAP+2 (-> b2),
// one halo row of the next stage (taps 1..15) and the A fragments of tap TAP+1 (-> an).
// vmcnt is an in-order counter: waiting for the weights of tap t also waits for every load issued
// before them, so the DMA rows go out AFTER a weight request and the weights run two taps ahead --
// a DMA row then has two taps (~1.7 us) to land before anything has to wait for it.
// 27 taps rotate the three weight buffers back to their starting roles; after tap 26 b0/b1 hold
// taps 0/1 of the NEXT stage.
template <int NT, int TAP>
__device__ __forceinline__ void pipe_step(const ConvArgs& a, const PipeCtx<NT>& c, const unsigned (&aaddr)[2][3][4],
                                          float4 (&ac)[2][4], float4 (&an)[2][4], float4 (&b0)[NT][4],
                                          float4 (&b1)[NT][4], float4 (&b2)[NT][4], f32x16 (&acc)[2][NT]) {
  if (c.tap_probe) {                       // development aid: cycle counter at every tap of one stage
    if (c.lane == 0) c.tap_probe[c.wave * 27 + TAP] = __builtin_readcyclecounter();
  }
  if constexpr (TAP + 2 < 27) {
    load_b<NT>(c.wr, c.wsoff + (unsigned)(TAP + 2) * c.wstride, c.lane_off, b2);
  } else {
    // past the last item this re-reads a valid (unused) weight block: no branch in the tap
    load_b<NT>(c.wr, c.wsoff_next + (unsigned)(TAP + 2 - 27) * c.wstride, c.lane_off, b2);
  }
  if constexpr (TAP >= 1 && TAP <= PIPE_ROWS_PER_WAVE) pipe_dma_row<TAP - 1>(a, c.xr, c.lds3, c.dm, c.wave);
  if constexpr (TAP < 26) pipe_read_a_tap<TAP + 1>(c.lds3, aaddr, an);
  __builtin_amdgcn_sched_barrier(0);
  pipe_mfma<NT>(ac, b0, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP < 26) pipe_step<NT, TAP + 1>(a, c, aaddr, an, ac, b1, b2, b0, acc);
}

template <int NT>
__global__ void __launch_bounds__(256, 1) k_conv3d_k3s1_pipe(ConvArgs a, PipeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);                  // = d-slice of the tile
  const int half = lane >> 5, i = lane & 31;
  const int pj = patch_of_row(i), pr = pj >> 3, pc = pj & 7;
  const int ntiles_total = a.cout_total >> 5;
  const int nchunk = a.Cin / KC;

  // work items of this block: XCD x (workgroups are dealt round-robin to the 8 XCDs) owns a
  // contiguous eighth of the items, so neighbouring tiles (shared halos, same weights) meet in
  // one L2; inside the XCD the blocks stride over that range
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

  PipeCtx<NT> c;
  c.lds3 = (lds3_t)lds;
  c.xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 27 * ntiles_total * 4096));
  c.lane_off = (unsigned)lane * 64u;
  c.wstride = (unsigned)ntiles_total * 4096u;
  c.wave = wave; c.lane = lane;

  PipeTile t = pipe_decode(a, p, item);
  int ch = 0;
  float4 a0[2][4], a1[2][4], b0[NT][4], b1[NT][4], b2[NT][4];
  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // prologue: the first stage's halo goes out in one burst
  {
    PipeDma dm;
    pipe_lane_offsets(a, t.w0, lane, dm.voff);
    dm.b = t.b; dm.d0 = t.d0; dm.h0 = t.h0; dm.wbase = t.w0 > 0 ? t.w0 - 1 : 0; dm.ch = 0; dm.ldsbuf = 0;
    dm.live = true;
    load_b<NT>(c.wr, (unsigned)((t.ng * NT) * 4096), c.lane_off, b0);
    load_b<NT>(c.wr, (unsigned)((t.ng * NT) * 4096) + c.wstride, c.lane_off, b1);
    pipe_dma_row<0>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<1>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<2>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<3>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<4>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<5>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<6>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<7>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<8>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<9>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<10>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<11>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<12>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<13>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<14>(a, c.xr, c.lds3, dm, wave);
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
          // opaque: keep ONE address register per variant and the tap offset as the ds_read
          // immediate (the compiler otherwise re-adds buffer + tap offset per read: 8 VALU per tap)
          asm volatile("" : "+v"(aaddr[khp][kw][q]));
        }
    long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.probe) ts0 = __builtin_readcyclecounter();
    pipe_read_a_tap<0>(c.lds3, aaddr, a0);
    // folded-BN scale/bias of this lane's output column: requested now, used in the epilogue
    // (all waves of the block reach the epilogue together, nothing would hide the latency there)
    float sc_r[NT], bi_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (t.ng * NT + nt) * 32 + i;
      sc_r[nt] = a.scale ? a.scale[n] : 1.f;
      bi_r[nt] = a.bias ? a.bias[n] : 0.f;
    }

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
    c.tap_probe = (a.probe && blockIdx.x == 17 && stage == 3) ? a.probe + 256 * 8 * 16 * 4 : nullptr;

    pipe_step<NT, 0>(a, c, aaddr, a0, a1, b0, b1, b2, acc);
    if (a.probe) ts1 = __builtin_readcyclecounter();

    __builtin_amdgcn_s_waitcnt(0);     // my DMA rows of the next stage have landed
    __syncthreads();                   // everyone's have; everyone is done with this buffer
    if (a.probe) ts2 = __builtin_readcyclecounter();

    if (ch == nchunk - 1) {
      // ---- epilogue: y = acc*scale + bias (+residual) (ReLU), see k_conv3d_k3s1
      const int od = t.d0 + wave;
      if (od < a.Do) {
        const bool interior = t.h0 + BH <= a.Ho && t.w0 + BW <= a.Wo;
        const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n0 = (t.ng * NT + nt) * 32;
          const bool to_y0 = n0 < a.cout0;
          float* dst = to_y0 ? a.y0 : a.y1;
          if (dst == nullptr) continue;
          const int stride = to_y0 ? a.cout0 : a.cout1;      // valid columns
    const int ld = to_y0 ? a.ld0 : a.ld1;              // row stride
          const int col0 = to_y0 ? n0 : n0 - a.n1_start;
          if (col0 < 0 || col0 >= stride) continue;
          const bool fullcols = col0 + 32 <= stride;
          const float lo = (to_y0 ? a.relu0 : a.relu1) ? 0.f : -3.402823466e38f;
          const bool has_res = to_y0 && a.residual != nullptr;
          const rsrc_t yr = make_rsrc(dst, out_vox * (unsigned)ld * 4u);
          const rsrc_t rr = make_rsrc(has_res ? a.residual : dst, out_vox * (unsigned)ld * 4u);
          const float sc = sc_r[nt];
          const float bi = bi_r[nt];
          const unsigned lanecol = (unsigned)(col0 + i) * 4u;
          unsigned goff[4];
          int gpr[4], gpc[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int p0 = acc_patch(4 * g4, 0), p1 = acc_patch(4 * g4, 1);
            gpr[g4] = half ? (p1 >> 3) : (p0 >> 3);
            gpc[g4] = half ? (p1 & 7) : (p0 & 7);
            goff[g4] = (unsigned)((gpr[g4] * a.Wo + gpc[g4]) * ld) * 4u + lanecol;
          }
          const unsigned estep = (unsigned)ld * 4u;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const unsigned soff = (unsigned)(((((t.b * a.Do + od) * a.Ho + (t.h0 + mt * 4)) * a.Wo + t.w0) * ld) * 4);
            if (interior && fullcols) {
              float rv[16];
              if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = buf_load1(rr, goff[r >> 2], soff + (unsigned)(r & 3) * estep);
              }
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] * sc + bi;
                if (has_res) v += rv[r];
                buf_store1(yr, goff[r >> 2], soff + (unsigned)(r & 3) * estep, fmaxf(v, lo));
              }
            } else {
              const bool colok = col0 + i < stride;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                if (colok && (t.h0 + mt * 4 + gpr[r >> 2] < a.Ho) && (t.w0 + gpc[r >> 2] + (r & 3) < a.Wo)) {
                  float v = acc[mt][nt][r] * sc + bi;
                  const unsigned so = soff + (unsigned)(r & 3) * estep;
                  if (has_res) v += buf_load1(rr, goff[r >> 2], so);
                  buf_store1(yr, goff[r >> 2], so, fmaxf(v, lo));
                }
              }
            }
          }
        }
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
}

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
struct OccTail {
  const float* w1;      // [8][16]  occ_pred_conv.0.weight
  const float* s1;      // [8]      folded BN scale
  const float* b1;      // [8]      folded BN bias
  const float* w2;      // [18][8]  occ_pred_conv.3.weight
  uint8_t* occ;         // [B*D*H*W] argmax class
  float* logits;        // [B*D*H*W][18] or null
  uint8_t* geo;         // [B*D*H*W] geo_occ or null
  int empty_idx;
  int n_mid, n_hid, n_cls;
};

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
  const StageLane sl = stage_lane_setup(a, w0, lane);
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
    if (WD == 1 && a.dma_stage) stage_halo_chunk_dma(a, xr, lds, b, d0, h0, w0, ch, wave, lane);
    else stage_halo_chunk<WD, 8>(a, xr, lds, sl, b, d0, h0, w0, ch, wave, lane);
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

// ------------------------------------------------------------------------------------
// OccHead, persistent DMA-pipelined variant: the stage machinery of k_conv3d_k3s1_pipe (two halo
// buffers, buffer_load ... lds issued by the MFMA wave, A of tap t+1 / weights of tap t+2 in
// flight, one item = one 4x8x8 tile) around the 16x16x4 tap body and the fused 16->8->18+argmax
// tail of k_occ_head16.  The tail transposes through the stage's own halo buffer once every wave
// is done reading it, hence a second barrier before the next stage's DMA may overwrite it.
// ------------------------------------------------------------------------------------
struct OccPipeCtx {
  lds3_t lds3;
  rsrc_t xr, wr;
  unsigned lane_off;
  unsigned wsoff, wsoff_next;
  PipeDma dm;
  int wave;
};

template <int TAP>
__device__ __forceinline__ void occ_read_a_tap(lds3_t lds3, const unsigned (&aaddr)[2][3][2], float4 (&aq)[4][2]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
  typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    constexpr unsigned imm0 = (unsigned)(((kd * TH + kh) * TW) * 128);
    const unsigned imm = imm0 + (unsigned)(mt * 2 * TW * 128);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const v4f v = *reinterpret_cast<const __attribute__((address_space(3))) v4f*>(lds3 + aaddr[kh & 1][kw][q] + imm);
      aq[mt][q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

__device__ __forceinline__ void occ_mfma(const float4 (&aq)[4][2], const float4 (&b)[2], f32x4 (&acc)[4]) {
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
__device__ __forceinline__ void occ_step(const ConvArgs& a, const OccPipeCtx& c, const unsigned (&aaddr)[2][3][2],
                                         float4 (&ac)[4][2], float4 (&an)[4][2], float4 (&b0)[2], float4 (&b1)[2],
                                         float4 (&b2)[2], f32x4 (&acc)[4]) {
  if constexpr (TAP + 2 < 27) load_b16(c.wr, c.wsoff + (unsigned)(TAP + 2) * 2048u, c.lane_off, b2);
  else load_b16(c.wr, c.wsoff_next + (unsigned)(TAP + 2 - 27) * 2048u, c.lane_off, b2);
  if constexpr (TAP >= 1 && TAP <= PIPE_ROWS_PER_WAVE) pipe_dma_row<TAP - 1>(a, c.xr, c.lds3, c.dm, c.wave);
  if constexpr (TAP < 26) occ_read_a_tap<TAP + 1>(c.lds3, aaddr, an);
  __builtin_amdgcn_sched_barrier(0);
  occ_mfma(ac, b0, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP < 26) occ_step<TAP + 1>(a, c, aaddr, an, ac, b1, b2, b0, acc);
}

__global__ void __launch_bounds__(256, 1) k_occ_head16_pipe(ConvArgs a, PipeArgs p, OccTail tail) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int nchunk = a.Cin / KC;
  const int nslots = (int)gridDim.x >> 3;
  const int per = (p.n_items + 7) >> 3;
  const int it_end = min(((int)blockIdx.x & 7) * per + per, p.n_items);
  int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (item >= it_end) return;

  unsigned aaddr0[2][3][2];
#pragma unroll
  for (int khp = 0; khp < 2; ++khp)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = (i & 7) + kw, hr = i >> 3;
      const int f = ((ww >> 1) & 3) | (((hr + khp) & 1) << 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        aaddr0[khp][kw][q] = (unsigned)((((wave * TH + hr) * TW + ww) * 8 + ((g * 2 + q) ^ f)) * 16);
    }
  OccPipeCtx c;
  c.lds3 = (lds3_t)lds;
  c.xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 27 * 2048));
  c.lane_off = (unsigned)lane * 32u;
  c.wave = wave;

  PipeTile t = pipe_decode(a, p, item);
  int ch = 0;
  float4 a0[4][2], a1[4][2], b0[2], b1[2], b2[2];
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;
  {
    PipeDma dm;
    pipe_lane_offsets(a, t.w0, lane, dm.voff);
    dm.b = t.b; dm.d0 = t.d0; dm.h0 = t.h0; dm.wbase = t.w0 > 0 ? t.w0 - 1 : 0; dm.ch = 0; dm.ldsbuf = 0;
    dm.live = true;
    load_b16(c.wr, 0u, c.lane_off, b0);
    load_b16(c.wr, 2048u, c.lane_off, b1);
    pipe_dma_row<0>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<1>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<2>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<3>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<4>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<5>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<6>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<7>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<8>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<9>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<10>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<11>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<12>(a, c.xr, c.lds3, dm, wave); pipe_dma_row<13>(a, c.xr, c.lds3, dm, wave);
    pipe_dma_row<14>(a, c.xr, c.lds3, dm, wave);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  const float sc = a.scale ? a.scale[i] : 1.f;
  const float bi = a.bias ? a.bias[i] : 0.f;

  for (int stage = 0;; ++stage) {
    const unsigned bufoff = (stage & 1) ? (unsigned)PIPE_BUF_BYTES : 0u;
    unsigned aaddr[2][3][2];
#pragma unroll
    for (int khp = 0; khp < 2; ++khp)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          aaddr[khp][kw][q] = aaddr0[khp][kw][q] + bufoff;
          asm volatile("" : "+v"(aaddr[khp][kw][q]));
        }
    occ_read_a_tap<0>(c.lds3, aaddr, a0);
    PipeTile tn = t;
    int chn = ch + 1, itemn = item;
    if (chn == nchunk) { chn = 0; itemn = item + nslots; }
    const bool has_next = itemn < it_end;
    if (has_next && chn == 0) tn = pipe_decode(a, p, itemn);
    if (!has_next) chn = 0;
    c.wsoff = (unsigned)(ch * 27 * 2048);
    c.wsoff_next = (unsigned)(chn * 27 * 2048);
    pipe_lane_offsets(a, tn.w0, lane, c.dm.voff);
    c.dm.b = tn.b; c.dm.d0 = tn.d0; c.dm.h0 = tn.h0; c.dm.wbase = tn.w0 > 0 ? tn.w0 - 1 : 0;
    c.dm.ch = chn; c.dm.ldsbuf = (unsigned)PIPE_BUF_BYTES - bufoff; c.dm.live = has_next;

    occ_step<0>(a, c, aaddr, a0, a1, b0, b1, b2, acc);

    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();                   // next stage's halo landed; this stage's buffer is free

    if (ch == nchunk - 1) {
      // ---- tail (see k_occ_head16): BN+ReLU, 64x16 transpose through this stage's halo buffer,
      // per-voxel 16->8->18 + argmax
      constexpr int MS = 17;
      float* sm = lds + (bufoff >> 2) + wave * (64 * MS);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[(mt * 16 + g * 4 + r) * MS + i] = fmaxf(acc[mt][r] * sc + bi, 0.f);
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's LDS writes are done (wave-private tile)
      const int od = t.d0 + wave;
      const int mt = lane >> 4, row = lane & 15;
      const int oh = t.h0 + mt * 2 + (row >> 3), ow = t.w0 + (row & 7);
      if (od < a.Do && oh < a.Ho && ow < a.Wo) {
        const size_t vox = (((size_t)t.b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
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
        for (int cc = 0; cc < 18; ++cc) {
          float s_ = 0.f;
#pragma unroll
          for (int o = 0; o < 8; ++o) s_ += hid[o] * tail.w2[cc * 8 + o];
          if (tail.logits) tail.logits[vox * 18 + cc] = s_;
          if (cc == 0 || s_ > best) { best = s_; arg = cc; }
        }
        tail.occ[vox] = (uint8_t)arg;
        if (tail.geo) tail.geo[vox] = arg != tail.empty_idx ? (uint8_t)0 : (uint8_t)(tail.n_cls - 1);
      }
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m4][r] = 0.f;
      __syncthreads();                 // transposes done before the next stage's DMA reuses this buffer
    }
    if (!has_next) break;
    t = tn; ch = chn; item = itemn;
  }
}

// ------------------------------------------------------------------------------------
// generic gather kernel: KS in {1,2,3}, STRIDE in {1,2}; A fragments straight from global/L2.
// One M-tile = 32 consecutive output voxels (linear index) per wave; used for the stride-2
// convs, the 1x1x1 convs, the 2x2x2 patchify convs (A20) and as the any-shape fallback.
//
// Branch-free and software-pipelined: a lane keeps ONE byte offset (its reference input voxel,
// always inside the volume); the tap displacement and the channel chunk move the scalar buffer
// base, the per-axis bounds tests are 3 x KS lane masks computed once, and a tap that falls
// outside the volume swaps the offset for an out-of-range one (the load unit returns 0).  With no
// branch around the loads, A and weights of tap t+1 are requested before the MFMAs of tap t.
// (First version: `if (inb)` around the A loads + 64-bit address math per tap: every tap paid an
// exposed L2 round trip -- 196 us for the 32->128 stride-2 layer.)
// ------------------------------------------------------------------------------------
constexpr unsigned GATHER_OOB = 0xfffffff0u;

template <int KS, int MT>
struct GatherCtx {
  const float* xbase;          // a.x (scalar)
  unsigned voff[MT];           // byte offset of this lane's reference voxel (+ its 64-byte half) per M-tile
  bool vd[MT][KS], vh[MT][KS], vw[MT][KS];  // per lane: tap plane/row/column inside the volume (lane masks in SGPRs)
  int H, W, Cin;
  rsrc_t wr;
  unsigned lane_off, wstride;
};

template <int NT, int KS, int MT, int TAP>
__device__ __forceinline__ void gather_load(const GatherCtx<KS, MT>& c, int ch, unsigned wsoff,
                                            float4 (&aq)[MT][4], float4 (&bq)[NT][4]) {
  constexpr int PAD = (KS - 1) / 2;
  constexpr int kd = TAP / (KS * KS), kh = (TAP / KS) % KS, kw = TAP % KS;
  // scalar: element displacement of this tap relative to the reference tap (PAD,PAD,PAD), plus the chunk
  const long long delta = ((long long)((kd - PAD) * c.H + (kh - PAD)) * c.W + (kw - PAD)) * c.Cin + ch * KC;
  const rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.xbase + delta), 0, 0xffffffe0u, 0x00020000);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const unsigned v = (c.vd[mt][kd] && c.vh[mt][kh] && c.vw[mt][kw]) ? c.voff[mt] : GATHER_OOB;
#pragma unroll
    for (int q = 0; q < 4; ++q) aq[mt][q] = buf_load4(xr, v, (unsigned)(q * 16));
  }
  load_b<NT>(c.wr, wsoff + (unsigned)TAP * c.wstride, c.lane_off, bq);
}

template <int NT, int MT>
__device__ __forceinline__ void gather_mfma(const float4 (&aq)[MT][4], const float4 (&bq)[NT][4], f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float av[4] = {aq[mt][q].x, aq[mt][q].y, aq[mt][q].z, aq[mt][q].w};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv[4] = {bq[nt][q].x, bq[nt][q].y, bq[nt][q].z, bq[nt][q].w};
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[mt][nt], 0, 0, 0);
        }
      }
  }
}

// tap TAP computes from (ac, bc) while (an, bn) receive tap TAP+1 (or tap 0 of the next chunk)
template <int NT, int KS, int MT, int TAP>
__device__ __forceinline__ void gather_step(const GatherCtx<KS, MT>& c, int ch, int ch_step, bool more_chunks, unsigned wsoff,
                                            unsigned wsoff_next, float4 (&ac)[MT][4], float4 (&bc)[NT][4],
                                            float4 (&an)[MT][4], float4 (&bn)[NT][4], f32x16 (&acc)[MT][NT]) {
  constexpr int TAPS = KS * KS * KS;
  if constexpr (TAP + 1 < TAPS) {
    gather_load<NT, KS, MT, TAP + 1>(c, ch, wsoff, an, bn);
  } else {
    if (more_chunks) gather_load<NT, KS, MT, 0>(c, ch + ch_step, wsoff_next, an, bn);
  }
  __builtin_amdgcn_sched_barrier(0);
  gather_mfma<NT, MT>(ac, bc, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP + 1 < TAPS)
    gather_step<NT, KS, MT, TAP + 1>(c, ch, ch_step, more_chunks, wsoff, wsoff_next, an, bn, ac, bc, acc);
}

// MT = M-tiles (32 output voxels each) per wave (default 1, see the dispatch)
// ksplit (1, 2 or 4): the block's 4 waves are 4/ksplit M-groups x ksplit partitions of the input-channel
// chunks (wave w: M-group w / ksplit, chunks w % ksplit, + ksplit, ...).  The partial accumulators
// meet in LDS and partition 0 adds them in a fixed order (deterministic) before the epilogue.  Small
// grids need this: with one (M-tile, N-group) per wave the 4x50x50 stage has ~1.2 waves of 1728-3456
// MFMAs per SIMD, i.e. the slowest SIMD does 2 of them; split by 4 it is ~5 waves of 432.
template <int NT, int KS, int STRIDE, int MT, int KSPL>
__global__ void __launch_bounds__(256) k_conv3d_gather(ConvArgs a, long long n_out_vox) {
  constexpr int ksplit = KSPL;          // compile-time: the unsplit kernel keeps its straight-line code
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, i = lane & 31;
  const int mslot = wave / ksplit, kpart = wave - mslot * ksplit;
  const long long m0 = ((long long)blockIdx.x * (4 / ksplit) + mslot) * (32 * MT);
  const bool active = m0 < n_out_vox;                 // wave-uniform; inactive waves still meet the barriers
  const int ng = blockIdx.y;
  const int ntiles_total = a.cout_total >> 5;
  constexpr int TAPS = KS * KS * KS;
  constexpr int PAD = (KS - 1) / 2;       // k3: 1, k2 (stride-2 patchify, A20): 0, k1: 0

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  if (active) {
    GatherCtx<KS, MT> c;
    c.xbase = a.x; c.H = a.H; c.W = a.W; c.Cin = a.Cin;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      long long m = m0 + mt * 32 + i;
      const bool mvalid = m < n_out_vox;
      if (!mvalid) m = n_out_vox - 1;
      const int ow = (int)(m % a.Wo); long long t = m / a.Wo;
      const int oh = (int)(t % a.Ho); t /= a.Ho;
      const int od = (int)(t % a.Do);
      const int b = (int)(t / a.Do);
      // reference tap (PAD,PAD,PAD) = input voxel (od*S, oh*S, ow*S): always inside the volume
      c.voff[mt] = (unsigned)((((((long long)b * a.D + od * STRIDE) * a.H + oh * STRIDE) * a.W + ow * STRIDE) * a.Cin + half * 16) * 4);
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        c.vd[mt][k] = mvalid && (unsigned)(od * STRIDE - PAD + k) < (unsigned)a.D;
        c.vh[mt][k] = (unsigned)(oh * STRIDE - PAD + k) < (unsigned)a.H;
        c.vw[mt][k] = (unsigned)(ow * STRIDE - PAD + k) < (unsigned)a.W;
      }
    }
    const int nchunk = a.Cin / KC;
    c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * TAPS * ntiles_total * 4096));
    c.lane_off = (unsigned)lane * 64u;
    c.wstride = (unsigned)ntiles_total * 4096u;

    float4 a0[MT][4], a1[MT][4], b0[NT][4], b1[NT][4];
    if (kpart < nchunk)
      gather_load<NT, KS, MT, 0>(c, kpart, (unsigned)((kpart * TAPS * ntiles_total + ng * NT) * 4096), a0, b0);
    for (int ch = kpart; ch < nchunk; ch += ksplit) {
      const unsigned wsoff = (unsigned)((ch * TAPS * ntiles_total + ng * NT) * 4096);
      const unsigned wsoff_next = (unsigned)(((ch + ksplit) * TAPS * ntiles_total + ng * NT) * 4096);
      gather_step<NT, KS, MT, 0>(c, ch, ksplit, ch + ksplit < nchunk, wsoff, wsoff_next, a0, b0, a1, b1, acc);
      if constexpr (TAPS & 1) {             // an odd tap count leaves the next chunk's tap 0 in (a1, b1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) a0[mt][q] = a1[mt][q];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) b0[nt][q] = b1[nt][q];
      }
    }
  }
  if (ksplit > 1) {
    // partial sums of partitions 1.. -> LDS [slot][mt][nt][r][lane]; partition 0 adds them in order
    if (kpart > 0) {
      float* dst = red + (size_t)((mslot * (ksplit - 1) + (kpart - 1)) * MT * NT) * 1024 + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((mt * NT + nt) * 16 + r) * 64] = acc[mt][nt][r];
    }
    __syncthreads();
    if (kpart > 0) return;
    for (int pp = 1; pp < ksplit; ++pp) {
      const float* src = red + (size_t)((mslot * (ksplit - 1) + (pp - 1)) * MT * NT) * 1024 + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += src[((mt * NT + nt) * 16 + r) * 64];
    }
  }
  if (!active) return;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (ng * NT + nt) * 32 + i;
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float bi = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const long long vox = m0 + mt * 32 + row;
        if (vox < n_out_vox) store_out(a, n, (size_t)vox, acc[mt][nt][r] * sc + bi);
      }
  }
}

// ------------------------------------------------------------------------------------
// LSSFPN3D fused (mmdet3d/models/necks/lss_fpn.py:132-148): the reference upsamples the 1/2
// and 1/4 resolution maps x2/x4 (trilinear, align_corners=True), concatenates 32+64+128
// channels (a 573 MB tensor) and runs a 1x1x1 conv 224->32 + BN + ReLU.  Trilinear
// interpolation and a 1x1x1 conv commute (both linear, no bias in between), so the conv is
// applied at the LOW resolution first (y16 = W[:,32:96] x16, y32 = W[:,96:224] x32 -- plain
// pw_conv3d_ndhwc 1x1x1 calls) and this kernel computes, at full resolution,
//   out = ReLU(BN(W[:,0:32] x8 + up2(y16) + up4(y32)))
// reading x8 once and writing out once: no concat tensor, no upsampled tensors.
// ------------------------------------------------------------------------------------
struct FpnArgs {
  const float* y16;   // (B, D2, H2, W2, 32)
  const float* y32;   // (B, D4, H4, W4, 32)
  int D2, H2, W2, D4, H4, W4;
};

__device__ __forceinline__ float trilerp_ac(const float* __restrict__ y, int b, int Dl, int Hl,
                                            int Wl, float sd, float sh, float sw, int od, int oh,
                                            int ow, int ch) {
  // ATen upsample_trilinear3d, align_corners=True: src = dst*(in-1)/(out-1)
  const float fd = sd * (float)od, fh = sh * (float)oh, fw = sw * (float)ow;
  const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
  const int d1 = d0 + (d0 < Dl - 1), h1 = h0 + (h0 < Hl - 1), w1 = w0 + (w0 < Wl - 1);
  const float ld1 = fd - (float)d0, ld0 = 1.f - ld1;
  const float lh1 = fh - (float)h0, lh0 = 1.f - lh1;
  const float lw1 = fw - (float)w0, lw0 = 1.f - lw1;
  const float* p = y + (size_t)b * Dl * Hl * Wl * 32 + ch;
#define YV(d, h, w) p[(unsigned)(((d) * Hl + (h)) * Wl + (w)) * 32u]
  const float v000 = YV(d0, h0, w0), v001 = YV(d0, h0, w1), v010 = YV(d0, h1, w0), v011 = YV(d0, h1, w1);
  const float v100 = YV(d1, h0, w0), v101 = YV(d1, h0, w1), v110 = YV(d1, h1, w0), v111 = YV(d1, h1, w1);
#undef YV
  return ld0 * (lh0 * (lw0 * v000 + lw1 * v001) + lh1 * (lw0 * v010 + lw1 * v011)) +
         ld1 * (lh0 * (lw0 * v100 + lw1 * v101) + lh1 * (lw0 * v110 + lw1 * v111));
}

__global__ void __launch_bounds__(256) k_fpn3d_fuse(ConvArgs a, FpnArgs f, long long n_vox) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, i = lane & 31;
  const long long m0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (m0 >= n_vox) return;
  long long m = m0 + i;
  if (m >= n_vox) m = n_vox - 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nchunk = a.Cin / KC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* src = a.x + (size_t)m * a.Cin + ch * KC + half * 16;
    const float* wt = a.wpk + (size_t)ch * 1024 + lane * 16;
    float4 aq[4], bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aq[q] = *reinterpret_cast<const float4*>(src + q * 4);
      bq[q] = *reinterpret_cast<const float4*>(wt + q * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[4] = {aq[q].x, aq[q].y, aq[q].z, aq[q].w};
      const float bv[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
  }
  const float sc = a.scale ? a.scale[i] : 1.f;
  const float bi = a.bias ? a.bias[i] : 0.f;
  const float sd2 = a.D > 1 ? (float)(f.D2 - 1) / (float)(a.D - 1) : 0.f;
  const float sh2 = a.H > 1 ? (float)(f.H2 - 1) / (float)(a.H - 1) : 0.f;
  const float sw2 = a.W > 1 ? (float)(f.W2 - 1) / (float)(a.W - 1) : 0.f;
  const float sd4 = a.D > 1 ? (float)(f.D4 - 1) / (float)(a.D - 1) : 0.f;
  const float sh4 = a.H > 1 ? (float)(f.H4 - 1) / (float)(a.H - 1) : 0.f;
  const float sw4 = a.W > 1 ? (float)(f.W4 - 1) / (float)(a.W - 1) : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    const long long vox = m0 + row;
    if (vox < n_vox) {
      // 32-bit index math (host guarantees n_vox < 2^31): 64-bit div/mod dominated this kernel
      const unsigned uv = (unsigned)vox;
      const unsigned t1 = uv / (unsigned)a.W;
      const int ow = (int)(uv - t1 * (unsigned)a.W);
      const unsigned t2 = t1 / (unsigned)a.H;
      const int oh = (int)(t1 - t2 * (unsigned)a.H);
      const int b = (int)(t2 / (unsigned)a.D);
      const int od = (int)(t2 - (unsigned)b * (unsigned)a.D);
      float v = acc[r];
      v += trilerp_ac(f.y16, b, f.D2, f.H2, f.W2, sd2, sh2, sw2, od, oh, ow, i);
      v += trilerp_ac(f.y32, b, f.D4, f.H4, f.W4, sd4, sh4, sw4, od, oh, ow, i);
      v = v * sc + bi;
      if (a.relu0) v = fmaxf(v, 0.f);
      a.y0[(size_t)vox * 32 + i] = v;
    }
  }
}

PW_API int pw_fpn3d_fuse(const float* x8, const float* wpk8, const float* y16, const float* y32,
                         const float* scale, const float* bias, float* out, int B, int D, int H,
                         int W, int Cin8, int D2, int H2, int W2, int D4, int H4, int W4, int relu,
                         void* stream) {
  PW_CHECK_ARG(x8 && wpk8 && y16 && y32 && out, "pw_fpn3d_fuse: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin8 > 0 && Cin8 % 32 == 0, "pw_fpn3d_fuse: bad shape");
  PW_CHECK_ARG(D2 > 0 && H2 > 0 && W2 > 0 && D4 > 0 && H4 > 0 && W4 > 0, "pw_fpn3d_fuse: bad level shape");
  ConvArgs a = {};
  a.x = x8; a.wpk = wpk8; a.scale = scale; a.bias = bias; a.y0 = out;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin8; a.relu0 = relu; a.cout_total = 32; a.cout0 = 32; a.ld0 = 32;
  FpnArgs f = {y16, y32, D2, H2, W2, D4, H4, W4};
  const long long n = (long long)B * D * H * W;
  PW_CHECK_ARG(n < (1ll << 31), "pw_fpn3d_fuse: more than 2^31 voxels");
  hipLaunchKernelGGL(k_fpn3d_fuse, dim3((unsigned)pw_cdiv(n, 128)), dim3(256), 0, pw_stream(stream),
                     a, f, n);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// host entry
// ------------------------------------------------------------------------------------
template <typename K>
static int set_lds_limit(K kernel, int bytes) {
  PW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PW_OK;
}

// halo staging of the tile-per-block kernels by buffer_load ... lds: on by default (A/B on one box, C3 step:
// 7.125 ms with the VGPR staging, 7.056 ms with DMA); PW_CONV_DMA_STAGE=0 selects the VGPR path
static int dma_stage_default() {
  const char* e = getenv("PW_CONV_DMA_STAGE");
  return e ? (atoi(e) ? 1 : 0) : 1;
}

// exact x / d by one mulhi for x * d < 2^32 (tile counts): floor(2^32 / d) + 1
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

static int pw_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n < 8) n = 256;
  }
  return n;
}

// Persistent DMA-pipelined kernel: PW_CONV_PIPE=0|1 forces it off/on.  Sustained timings (1 s loops,
// clocks settled at 2.39 GHz, tools/bench_layers.py), tile-per-block vs pipelined:
//   16x200x200 32->32  277.9 / 278.4 us    32->64  518.3 / 537.6    64->64 1002.8 / 1049.3
//   8x100x100  64->64  180.4 / 165.6 us    64->128 308.0 / 319.5    4x50x50 128->128 110.8 / 110.4
// Both designs sit on the same ceiling (operand loads cost matrix-pipe time, see the kernel comment);
// the pipelined one wins where a wave owns a single N-tile and the grid gives every CU 2+ items.
static bool use_pipe(long long n_items, int NT) {
  const char* e = getenv("PW_CONV_PIPE");            // read per call: tests flip it inside one process
  const int forced = e ? (atoi(e) ? 1 : 0) : 2;
  if (forced != 2) return forced == 1 && n_items < (1ll << 20);
  return NT == 1 && n_items >= 512 && n_items < (1ll << 20);
}

// 8-wave blocks (WD=2) when the grid has enough 8x8x8 tiles to fill the 256 CUs more than once;
// 4-wave blocks otherwise (small encoder stages).  PW_CONV_WD=1|2 forces a variant (A/B runs).
static int choose_wd(int B, int Do, int Ho, int Wo, int ngroups) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("PW_CONV_WD");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 1 || forced == 2) return forced;
  // measured equal within noise on the 16x200x200 grid (325 vs 326 us for 32->32); the 4-wave
  // variant is the default because it needs less LDS per block and tiles small grids better
  (void)B; (void)Do; (void)Ho; (void)Wo; (void)ngroups;
  return 1;
}

PW_API int pw_conv3d_ndhwc(const float* x, const float* wpk, const float* scale, const float* bias,
                           const float* residual, float* y0, float* y1, int B, int D, int H, int W,
                           int Cin, int cout_total, int cout0, int cout1, int ld_y0, int ld_y1, int ksize,
                           int stride, int relu0, int relu1, int algo, void* stream) {
  PW_CHECK_ARG(x && wpk && y0, "pw_conv3d_ndhwc: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pw_conv3d_ndhwc: bad shape");
  PW_CHECK_ARG(Cin > 0 && Cin % KC == 0, "pw_conv3d_ndhwc: Cin must be a multiple of 32 (got %d)", Cin);
  PW_CHECK_ARG(cout_total > 0 && cout_total % 32 == 0, "pw_conv3d_ndhwc: cout_total must be a multiple of 32");
  PW_CHECK_ARG(cout0 > 0 && cout0 <= cout_total && cout1 >= 0, "pw_conv3d_ndhwc: bad cout split");
  PW_CHECK_ARG(ksize >= 1 && ksize <= 3, "pw_conv3d_ndhwc: kernel size must be 1, 2 or 3");
  PW_CHECK_ARG(!(ksize == 2 && stride != 2), "pw_conv3d_ndhwc: 2x2x2 is built for stride 2 only");
  PW_CHECK_ARG(stride == 1 || stride == 2, "pw_conv3d_ndhwc: stride must be 1 or 2");
  PW_CHECK_ARG(!(ksize == 1 && stride != 1), "pw_conv3d_ndhwc: 1x1x1 stride 2 unsupported");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)wpk) & 15) == 0, "pw_conv3d_ndhwc: x/wpk must be 16-B aligned");
  const int pad = (ksize - 1) / 2;
  ConvArgs a;
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias; a.residual = residual; a.y0 = y0; a.y1 = y1;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
  a.Do = (D + 2 * pad - ksize) / stride + 1;
  a.Ho = (H + 2 * pad - ksize) / stride + 1;
  a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.ld0 = ld_y0 > 0 ? ld_y0 : cout0; a.ld1 = ld_y1 > 0 ? ld_y1 : cout1;
  PW_CHECK_ARG(a.ld0 >= cout0 && a.ld1 >= cout1, "pw_conv3d_ndhwc: ld_y0/ld_y1 must be >= the channel counts");
  a.n1_start = (cout0 + 31) / 32 * 32;
  a.relu0 = relu0; a.relu1 = relu1;
  a.tiles_d = (a.Do + BD - 1) / BD; a.tiles_h = (a.Ho + BH - 1) / BH; a.tiles_w = (a.Wo + BW - 1) / BW;
  PW_CHECK_ARG(!(cout1 > 0 && !y1), "pw_conv3d_ndhwc: cout1 > 0 needs y1");
  PW_CHECK_ARG(a.n1_start + cout1 <= cout_total || cout1 == 0, "pw_conv3d_ndhwc: cout split exceeds cout_total");
  PW_CHECK_ARG((size_t)B * D * H * W * Cin * 4 < (1ull << 32) &&
                   (size_t)B * a.Do * a.Ho * a.Wo * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) * 4 < (1ull << 32),
               "pw_conv3d_ndhwc: tensors must be < 4 GiB (32-bit buffer addressing)");
  hipStream_t st = pw_stream(stream);
  const int ntiles = cout_total / 32;
  int NT = (ntiles % 2 == 0) ? 2 : 1;
  if (ksize == 3 && stride == 1 && algo != 2 && NT == 2) {
    // small grids: 2 blocks x 256 CUs = 512 resident slots; prefer twice as many half-size
    // blocks when NT=2 cannot fill them (the 8x100x100 and 4x50x50 encoder stages)
    long long nblk2 = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w * (ntiles / 2);
    if (nblk2 < 512) NT = 1;
  }
  if (const char* e = getenv("PW_CONV_NT")) {          // experiments only
    const int f = atoi(e);
    if ((f == 1 || f == 2) && ntiles % f == 0) NT = f;
  }
  const int ngroups = ntiles / NT;
  const long long n_out = (long long)B * a.Do * a.Ho * a.Wo;
  // algo: 0 = auto, 1 = force LDS-tiled (k3 s1 only), 2 = force gather, 3 = gather with the input
  // channels split over the 4 waves of a block (small grids, see k_conv3d_gather)
  // auto: a grid with fewer (tile, N-group) items than CUs and >= 4 input chunks (the 4x50x50 128->128
  // stage: 196 items) runs ~5 % faster on the channel-split gather kernel (106 vs 112 us)
  if (algo == 0 && ksize == 3 && stride == 1 && NT == 1 && (Cin / KC) % 4 == 0 &&
      (long long)B * a.tiles_d * a.tiles_h * a.tiles_w * ngroups < pw_num_cus())
    algo = 3;
  const bool tiled = (ksize == 3 && stride == 1 && algo != 2 && algo != 3);
  PW_CHECK_ARG(!(algo == 1 && !tiled), "pw_conv3d_ndhwc: algo=1 needs ksize 3 stride 1");
  a.probe = nullptr;
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  a.dma_stage = dma_stage_default();
  if (tiled && use_pipe((long long)B * a.tiles_d * a.tiles_h * a.tiles_w * ngroups, NT)) {
    PipeArgs p;
    p.ngroups = ngroups;
    p.n_items = (int)((long long)B * a.tiles_d * a.tiles_h * a.tiles_w * ngroups);
    p.m_ng = magic_of(ngroups); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h);
    p.m_td = magic_of(a.tiles_d);
    const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
    constexpr int PLDS = 2 * PIPE_BUF_BYTES;
    if (NT == 2) {
      static int once = set_lds_limit(k_conv3d_k3s1_pipe<2>, PLDS);
      if (once) return once;
      hipLaunchKernelGGL((k_conv3d_k3s1_pipe<2>), dim3(nb), dim3(256), PLDS, st, a, p);
    } else {
      static int once = set_lds_limit(k_conv3d_k3s1_pipe<1>, PLDS);
      if (once) return once;
      hipLaunchKernelGGL((k_conv3d_k3s1_pipe<1>), dim3(nb), dim3(256), PLDS, st, a, p);
    }
  } else if (tiled) {
    const int WD = choose_wd(B, a.Do, a.Ho, a.Wo, ngroups);
    a.tiles_d = (a.Do + BD * WD - 1) / (BD * WD);
    long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
    PW_CHECK_ARG(nblk < (1ll << 31), "pw_conv3d_ndhwc: grid too large");
    dim3 grid((unsigned)nblk, (unsigned)ngroups);
#define PW_LAUNCH_TILED(NTv, WDv)                                                              \
  do {                                                                                         \
    static int once = set_lds_limit(k_conv3d_k3s1<NTv, WDv>, TileGeom<WDv>::LDS);               \
    if (once) return once;                                                                     \
    hipLaunchKernelGGL((k_conv3d_k3s1<NTv, WDv>), grid, dim3(256 * WDv), TileGeom<WDv>::LDS, st, a); \
  } while (0)
    if (NT == 2 && WD == 2) PW_LAUNCH_TILED(2, 2);
    else if (NT == 2) PW_LAUNCH_TILED(2, 1);
    else if (WD == 2) PW_LAUNCH_TILED(1, 2);
    else PW_LAUNCH_TILED(1, 1);
#undef PW_LAUNCH_TILED
  } else {
    // M-tiles per wave: 2 halves the weight loads per MFMA but measured slower (32->128 stride 2:
    // 199 us vs 180 us -- fewer waves to hide the L2 gather latency); PW_GATHER_MT=2 selects it
    const char* mte = getenv("PW_GATHER_MT");
    const int MT = (mte && atoi(mte) == 2) ? 2 : 1;
    const int nchunk = Cin / KC;
    int ksplit = 1;
    if (algo == 3) ksplit = (nchunk % 4 == 0) ? 4 : (nchunk % 2 == 0 ? 2 : 1);
    if (const char* e = getenv("PW_GATHER_KSPLIT")) {
      const int f = atoi(e);
      if ((f == 1 || f == 2 || f == 4) && nchunk % f == 0) ksplit = f;
    }
    if (ksplit > 1 && MT != 1) ksplit = 1;              // the split variants are built for MT = 1
    const int mgroups = 4 / ksplit;                     // M-groups (32*MT voxels each) per block
    dim3 grid((unsigned)pw_cdiv(n_out, 32 * MT * mgroups), (unsigned)ngroups);
    const size_t red_bytes = ksplit > 1 ? (size_t)mgroups * (ksplit - 1) * MT * NT * 4096 : 0;
#define PW_GATHER_L(NTv, KSv, STv, MTv, KSPv) \
  hipLaunchKernelGGL((k_conv3d_gather<NTv, KSv, STv, MTv, KSPv>), grid, dim3(256), red_bytes, st, a, n_out)
#define PW_GATHER(NTv, KSv, STv)                                        \
  do {                                                                  \
    if (ksplit == 4) PW_GATHER_L(NTv, KSv, STv, 1, 4);                  \
    else if (ksplit == 2) PW_GATHER_L(NTv, KSv, STv, 1, 2);             \
    else if (MT == 2) PW_GATHER_L(NTv, KSv, STv, 2, 1);                 \
    else PW_GATHER_L(NTv, KSv, STv, 1, 1);                              \
  } while (0)
    if (ksize == 1) {
      if (NT == 2) PW_GATHER(2, 1, 1); else PW_GATHER(1, 1, 1);
    } else if (ksize == 2) {
      if (NT == 2) PW_GATHER(2, 2, 2); else PW_GATHER(1, 2, 2);
    } else if (stride == 1) {
      if (NT == 2) PW_GATHER(2, 3, 1); else PW_GATHER(1, 3, 1);
    } else {
      if (NT == 2) PW_GATHER(2, 3, 2); else PW_GATHER(1, 3, 2);
    }
#undef PW_GATHER
#undef PW_GATHER_L
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// A11  fused OccHead: conv3x3x3 (Cin->16, BN, ReLU) + 1x1x1 16->8 (BN, ReLU) + 1x1x1 8->18 + argmax
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
  a.dma_stage = dma_stage_default();
  long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  PW_CHECK_ARG(wpk_layout == 16, "pw_occ_head_fused: wpk_layout must be 16 (the 16x16x4 MFMA packing)");
  // persistent DMA-pipelined variant: opt-in with PW_OCC_PIPE=1.  Measured at 16x200x200: 176 us vs
  // 160 us for the tile-per-block kernel -- with one block per CU nothing overlaps the fused tail
  // (two LDS transposes, ~330 VALU, the 16->8->18 weights) that the second resident block hides there.
  {
    const char* e = getenv("PW_OCC_PIPE");
    const int forced = e ? (atoi(e) ? 1 : 0) : 2;
    const bool pipe = forced == 1;
    if (pipe && nblk < (1ll << 20)) {
      PipeArgs p;
      p.ngroups = 1; p.n_items = (int)nblk;
      p.m_ng = magic_of(1); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
      const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
      constexpr int PLDS = 2 * PIPE_BUF_BYTES;
      static int once = set_lds_limit(k_occ_head16_pipe, PLDS);
      if (once) return once;
      hipLaunchKernelGGL(k_occ_head16_pipe, dim3(nb), dim3(256), PLDS, pw_stream(stream), a, p, t);
      PW_CHECK_LAUNCH();
      return PW_OK;
    }
  }
  const int WD = choose_wd(B, D, H, W, 1);
  a.tiles_d = (D + BD * WD - 1) / (BD * WD);
  nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (WD == 2) {
    static int once = set_lds_limit(k_occ_head16<2>, TileGeom<2>::LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head16<2>, dim3((unsigned)nblk, 1), dim3(512), TileGeom<2>::LDS,
                       pw_stream(stream), a, t);
  } else {
    static int once = set_lds_limit(k_occ_head16<1>, TileGeom<1>::LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_occ_head16<1>, dim3((unsigned)nblk, 1), dim3(256), TileGeom<1>::LDS,
                       pw_stream(stream), a, t);
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
