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
#include "pw_conv3d_common.h"

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
  const unsigned lane_off = (unsigned)lane * 16u;
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
    stage_halo_chunk_dma(a, xr, lds, b, d0, h0, w0, ch, wave, lane);
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
  c.lane_off = (unsigned)lane * 16u;
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
// host entry
// ------------------------------------------------------------------------------------
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
  ConvArgs a = {};
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
  const int ngroups = ntiles / NT;
  const long long n_out = (long long)B * a.Do * a.Ho * a.Wo;
  // algo: 0 = auto, 1 = the tile-per-block LDS kernel (k3 s1 only), 2 = the gather kernel, 3 = gather with the input channels
  // split over the 4 waves of a block (small grids, see k_conv3d_gather), 4 = the persistent DMA-pipelined kernel (k3 s1 only)
  // auto: a grid with fewer (tile, N-group) items than CUs and >= 4 input chunks (the 4x50x50 128->128
  // stage: 196 items) runs ~5 % faster on the channel-split gather kernel (106 vs 112 us)
  if (algo == 0 && ksize == 3 && stride == 1 && NT == 1 && (Cin / KC) % 4 == 0 &&
      (long long)B * a.tiles_d * a.tiles_h * a.tiles_w * ngroups < pw_num_cus())
    algo = 3;
  const bool tiled = (ksize == 3 && stride == 1 && algo != 2 && algo != 3);
  PW_CHECK_ARG(!((algo == 1 || algo == 4) && !tiled), "pw_conv3d_ndhwc: algo 1 / 4 need ksize 3 stride 1");
  a.probe = nullptr;
  if (const char* e = getenv("PW_CONV_PROBE")) a.probe = (long long*)strtoull(e, nullptr, 0);
  if (tiled && use_pipe((long long)B * a.tiles_d * a.tiles_h * a.tiles_w * ngroups, NT, algo)) {
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
      pw_note_kernel("k_conv3d_k3s1_pipe<2>");
    } else {
      static int once = set_lds_limit(k_conv3d_k3s1_pipe<1>, PLDS);
      if (once) return once;
      hipLaunchKernelGGL((k_conv3d_k3s1_pipe<1>), dim3(nb), dim3(256), PLDS, st, a, p);
      pw_note_kernel("k_conv3d_k3s1_pipe<1>");
    }
  } else if (tiled) {
    long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
    PW_CHECK_ARG(nblk < (1ll << 31), "pw_conv3d_ndhwc: grid too large");
    dim3 grid((unsigned)nblk, (unsigned)ngroups);
#define PW_LAUNCH_TILED(NTv)                                                                   \
  do {                                                                                         \
    static int once = set_lds_limit(k_conv3d_k3s1<NTv, 1>, TileGeom<1>::LDS);                   \
    if (once) return once;                                                                     \
    hipLaunchKernelGGL((k_conv3d_k3s1<NTv, 1>), grid, dim3(256), TileGeom<1>::LDS, st, a);      \
    pw_note_kernel("k_conv3d_k3s1<%d, 1>", NTv);                                                \
  } while (0)
    if (NT == 2) PW_LAUNCH_TILED(2); else PW_LAUNCH_TILED(1);
#undef PW_LAUNCH_TILED
  } else {
    if (int rc = pw_launch_conv3d_gather(a, NT, ngroups, ksize, stride, algo, Cin, n_out, st)) return rc;
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}

