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
constexpr int LDS_BYTES = TV * KC * 4;           // 76800
constexpr int STAGE_ITEMS = TV * (KC / 4);       // float4 items per chunk (4800)
constexpr int STAGE_ITERS = (STAGE_ITEMS + 255) / 256;
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
  int n1_start;           // first packed column that goes to y1
  int relu0, relu1;
  int tiles_d, tiles_h, tiles_w;
  long long* probe;       // optional per-block phase timestamps (development aid) or null
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
    size_t o = vox * a.cout0 + n;
    if (a.residual) v += a.residual[o];
    if (a.relu0) v = fmaxf(v, 0.f);
    a.y0[o] = v;
  } else {
    int n1 = n - a.n1_start;
    if (a.y1 && n1 >= 0 && n1 < a.cout1) {
      if (a.relu1) v = fmaxf(v, 0.f);
      a.y1[vox * a.cout1 + n1] = v;
    }
  }
}

// stage the 6x10x10 halo tile of one 32-channel chunk: global -> registers -> swizzled LDS.
// All 19 loads are issued before the first LDS write; zero padding comes from the bounds test.
__device__ __forceinline__ void stage_halo_chunk(const ConvArgs& a, float* lds, int b, int d0, int h0,
                                                 int w0, int ch, int tid, long long* ts = nullptr) {
  // Two waves share each SIMD and instruction issue is arbitrated by priority, then age: next to
  // a partner that streams MFMAs this ~1000-instruction address/load/ds_write phase was measured
  // at ~37 cycles per instruction (31k cycles, vs 1.4k actually waiting for the loads).  Run the
  // non-MFMA phases at high priority; they are short, so the partner's MFMA stream barely moves.
  __builtin_amdgcn_s_setprio(3);
  float4 tmp[STAGE_ITERS];
#pragma unroll
  for (int k = 0; k < STAGE_ITERS; ++k) {
    int it = tid + k * 256;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (it < STAGE_ITEMS) {
      int vox = it >> 3, slot = it & 7;
      int dd = vox / (TH * TW);
      int rem = vox - dd * (TH * TW);
      int hh = rem / TW, ww = rem - hh * TW;
      int gd = d0 + dd - 1, gh = h0 + hh - 1, gw = w0 + ww - 1;
      if ((unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W) {
        size_t g = ((((size_t)b * a.D + gd) * a.H + gh) * a.W + gw) * a.Cin + ch * KC + slot * 4;
        v = *reinterpret_cast<const float4*>(a.x + g);
      }
    }
    tmp[k] = v;
  }
  if (ts) { ts[0] = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ts[1] = __builtin_readcyclecounter(); }
  if (ch > 0) __syncthreads();   // every wave finished reading the previous chunk
#pragma unroll
  for (int k = 0; k < STAGE_ITERS; ++k) {
    int it = tid + k * 256;
    if (it < STAGE_ITEMS) {
      int vox = it >> 3, slot = it & 7;
      int dd = vox / (TH * TW);
      int rem = vox - dd * (TH * TW);
      int hh = rem / TW, ww = rem - hh * TW;
      int f = ((ww >> 1) & 3) | ((hh & 1) << 2);
      *reinterpret_cast<float4*>(lds + ((vox << 3) + (slot ^ f)) * 4) = tmp[k];
    }
  }
  if (ts) ts[2] = __builtin_readcyclecounter();
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
}

template <int NT>
__device__ __forceinline__ void load_b(const float* w, float4 (&b)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) b[nt][q] = *reinterpret_cast<const float4*>(w + nt * 1024 + q * 4);
}

// one tap: 2 M-tiles x NT N-tiles x 16 k-steps of v_mfma_f32_32x32x2_f32
template <int NT>
__device__ __forceinline__ void tap_mfma(const float* lds, int tap, int wave, int half, int pr,
                                         int pc, const float4 (&b)[NT][4], f32x16 (&acc)[2][NT]) {
  const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int hh = mt * 4 + pr + kh, ww = pc + kw, dd = wave + kd;
    const int vl = (dd * TH + hh) * TW + ww;
    const int f = ((ww >> 1) & 3) | ((hh & 1) << 2);
    float4 aq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      aq[q] = *reinterpret_cast<const float4*>(lds + ((vl << 3) + ((half * 4 + q) ^ f)) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[4] = {aq[q].x, aq[q].y, aq[q].z, aq[q].w};
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

// ------------------------------------------------------------------------------------
// 3x3x3, stride 1, pad 1, LDS halo tile
// ------------------------------------------------------------------------------------
// Fused OccHead tail (mmdet3d/models/heads/occupancy_head.py:92-99,124-161): after the 3x3x3
// conv + BN + ReLU (16 mid channels) each voxel runs 1x1x1 16->8 + BN + ReLU, 1x1x1 8->18 and
// argmax -> uint8 inside the conv epilogue; the 46 MB logits tensor is only written on request.
struct OccTail {
  const float* w1;      // [8][16]  occ_pred_conv.0.weight
  const float* s1;      // [8]      folded BN scale
  const float* b1;      // [8]      folded BN bias
  const float* w2;      // [18][8]  occ_pred_conv.3.weight
  uint8_t* occ;         // [B*D*H*W] argmax class
  float* logits;        // [B*D*H*W][18] or null
  int n_mid, n_hid, n_cls;
};

template <int NT, int EPI>
__global__ void __launch_bounds__(256, 2) k_conv3d_k3s1(ConvArgs a, OccTail tail) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, i = lane & 31;
  int bid = blockIdx.x;
  const int tw = bid % a.tiles_w; bid /= a.tiles_w;
  const int th = bid % a.tiles_h; bid /= a.tiles_h;
  const int td = bid % a.tiles_d;
  const int b = bid / a.tiles_d;
  const int d0 = td * BD, h0 = th * BH, w0 = tw * BW;
  const int ng = blockIdx.y;
  const int pj = patch_of_row(i), pr = pj >> 3, pc = pj & 7;
  const int ntiles_total = a.cout_total >> 5;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nchunk = a.Cin / KC;
  long long t_start = 0, t_staged = 0, t_taps = 0, t_s0 = 0, t_s1 = 0, t_s2 = 0;
  if (a.probe) t_start = __builtin_readcyclecounter();
  for (int ch = 0; ch < nchunk; ++ch) {
    long long ts[3] = {0, 0, 0};
    stage_halo_chunk(a, lds, b, d0, h0, w0, ch, tid, (a.probe && ch == 0) ? ts : nullptr);
    if (a.probe && ch == 0) { t_staged = __builtin_readcyclecounter(); t_s0 = ts[0]; t_s1 = ts[1]; t_s2 = ts[2]; }

    // ---- 27 taps x 16 k-steps
    const float* wch = a.wpk + ((size_t)ch * 27 * ntiles_total + (size_t)ng * NT) * 1024 + lane * 16;
    // weights are double-buffered in registers (ping-pong, taps two at a time) so that the
    // loads of tap t+1 stay in flight under the 32*NT MFMAs of tap t
    const size_t wstride = (size_t)ntiles_total * 1024;
    float4 b0[NT][4], b1[NT][4];
    load_b<NT>(wch, b0);
#pragma unroll 1
    for (int tap = 0; tap < 26; tap += 2) {
      // sched_barrier pins "issue next tap's loads, THEN compute": the waitcnt pass can then
      // use a counted vmcnt that leaves the 4*NT prefetch loads in flight under the MFMAs
      load_b<NT>(wch + (size_t)(tap + 1) * wstride, b1);
      __builtin_amdgcn_sched_barrier(0);
      tap_mfma<NT>(lds, tap, wave, half, pr, pc, b0, acc);
      __builtin_amdgcn_sched_barrier(0);
      load_b<NT>(wch + (size_t)(tap + 2) * wstride, b0);
      __builtin_amdgcn_sched_barrier(0);
      tap_mfma<NT>(lds, tap + 1, wave, half, pr, pc, b1, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    tap_mfma<NT>(lds, 26, wave, half, pr, pc, b0, acc);
  }

  // ---- epilogue (high issue priority, see stage_halo_chunk)
  const int od = d0 + wave;
  if (a.probe) t_taps = __builtin_readcyclecounter();
  __builtin_amdgcn_s_setprio(3);
  if constexpr (EPI == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (ng * NT + nt) * 32 + i;
      const float sc = a.scale ? a.scale[n] : 1.f;
      const float bi = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int pjr = patch_of_row(row);
          const int oh = h0 + mt * 4 + (pjr >> 3), ow = w0 + (pjr & 7);
          if (od < a.Do && oh < a.Ho && ow < a.Wo) {
            size_t vox = (((size_t)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
            store_out(a, n, vox, acc[mt][nt][r] * sc + bi);
          }
        }
      }
    }
    if (a.probe) {
      const long long t_issued = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t_drained = __builtin_readcyclecounter();
      if (lane == 0) {
        long long* p = a.probe + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        p[0] = t_start; p[1] = t_s0; p[2] = t_s1; p[3] = t_s2; p[4] = t_staged; p[5] = t_taps;
        p[6] = t_issued; p[7] = t_drained;
      }
    }
  } else {
    // transpose the wave's 64 voxels x 16 mid channels through LDS (reusing the halo tile),
    // then one lane per voxel runs the tiny MLP + argmax.
    static_assert(EPI == 0 || NT == 1, "OccHead tail expects a single N-tile");
    constexpr int MS = 17;                         // padded row stride (bank spread)
    __syncthreads();                               // all waves are done with the halo tile
    float* sm = lds + wave * (64 * MS);
    {
      const float sc = a.scale ? a.scale[i] : 1.f;
      const float bi = a.bias ? a.bias[i] : 0.f;
      if (i < tail.n_mid) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[mt][0][r] * sc + bi;
            if (a.relu0) v = fmaxf(v, 0.f);
            sm[(mt * 32 + row) * MS + i] = v;
          }
      }
    }
    __syncthreads();
    const int mt = lane >> 5, row = lane & 31;
    const int pjr = patch_of_row(row);
    const int oh = h0 + mt * 4 + (pjr >> 3), ow = w0 + (pjr & 7);
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
    }
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
// ------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_b16(const float* w, float4 (&b)[2]) {
  b[0] = *reinterpret_cast<const float4*>(w);
  b[1] = *reinterpret_cast<const float4*>(w + 4);
}

__device__ __forceinline__ void tap_mfma16(const float* lds, int tap, int wave, int g, int i,
                                           const float4 (&b)[2], f32x4 (&acc)[4]) {
  const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
  float4 aq[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int hh = mt * 2 + (i >> 3) + kh, ww = (i & 7) + kw, dd = wave + kd;
    const int vl = (dd * TH + hh) * TW + ww;
    const int f = ((ww >> 1) & 3) | ((hh & 1) << 2);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      aq[mt][q] = *reinterpret_cast<const float4*>(lds + ((vl << 3) + ((g * 2 + q) ^ f)) * 4);
  }
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

__global__ void __launch_bounds__(256, 2) k_occ_head16(ConvArgs a, OccTail tail) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i = lane & 15;
  int bid = blockIdx.x;
  const int tw = bid % a.tiles_w; bid /= a.tiles_w;
  const int th = bid % a.tiles_h; bid /= a.tiles_h;
  const int td = bid % a.tiles_d;
  const int b = bid / a.tiles_d;
  const int d0 = td * BD, h0 = th * BH, w0 = tw * BW;
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;
  const int nchunk = a.Cin / KC;
  for (int ch = 0; ch < nchunk; ++ch) {
    stage_halo_chunk(a, lds, b, d0, h0, w0, ch, tid);
    const float* wch = a.wpk + (size_t)ch * 27 * 512 + lane * 8;
    float4 b0[2], b1[2];
    load_b16(wch, b0);
#pragma unroll 1
    for (int tap = 0; tap < 26; tap += 2) {
      load_b16(wch + (size_t)(tap + 1) * 512, b1);
      __builtin_amdgcn_sched_barrier(0);
      tap_mfma16(lds, tap, wave, g, i, b0, acc);
      __builtin_amdgcn_sched_barrier(0);
      load_b16(wch + (size_t)(tap + 2) * 512, b0);
      __builtin_amdgcn_sched_barrier(0);
      tap_mfma16(lds, tap + 1, wave, g, i, b1, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    tap_mfma16(lds, 26, wave, g, i, b0, acc);
  }
  // ---- tail: BN+ReLU, transpose 64 voxels x 16 channels through LDS, per-voxel MLP + argmax
  constexpr int MS = 17;
  const int od = d0 + wave;
  __builtin_amdgcn_s_setprio(3);
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
  }
}

// ------------------------------------------------------------------------------------
// generic gather kernel: KS in {1,3}, STRIDE in {1,2}; A fragments straight from global/L2.
// One M-tile = 32 consecutive output voxels (linear index) per wave; used for the stride-2
// convs, the 1x1x1 convs and as the any-shape fallback.
// ------------------------------------------------------------------------------------
template <int NT, int KS, int STRIDE>
__global__ void __launch_bounds__(256) k_conv3d_gather(ConvArgs a, long long n_out_vox) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, i = lane & 31;
  const long long m0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (m0 >= n_out_vox) return;
  const int ng = blockIdx.y;
  const int ntiles_total = a.cout_total >> 5;
  constexpr int TAPS = KS * KS * KS;
  constexpr int PAD = KS / 2;
  long long m = m0 + i;
  const bool mvalid = m < n_out_vox;
  if (!mvalid) m = n_out_vox - 1;
  int ow = (int)(m % a.Wo); long long t = m / a.Wo;
  int oh = (int)(t % a.Ho); t /= a.Ho;
  int od = (int)(t % a.Do);
  int b = (int)(t / a.Do);

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const int nchunk = a.Cin / KC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* wch = a.wpk + ((size_t)ch * TAPS * ntiles_total + (size_t)ng * NT) * 1024 + lane * 16;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
      const int id = od * STRIDE - PAD + kd, ih = oh * STRIDE - PAD + kh, iw = ow * STRIDE - PAD + kw;
      const bool inb = mvalid && (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H &&
                       (unsigned)iw < (unsigned)a.W;
      float4 aq[4];
      if (inb) {
        const float* src = a.x + ((((size_t)b * a.D + id) * a.H + ih) * a.W + iw) * a.Cin + ch * KC + half * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q] = *reinterpret_cast<const float4*>(src + q * 4);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* wt = wch + (size_t)tap * ntiles_total * 1024;
      float4 bq[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bq[nt][q] = *reinterpret_cast<const float4*>(wt + nt * 1024 + q * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float av[4] = {aq[q].x, aq[q].y, aq[q].z, aq[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float bv[4] = {bq[nt][q].x, bq[nt][q].y, bq[nt][q].z, bq[nt][q].w};
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[nt], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (ng * NT + nt) * 32 + i;
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float bi = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const long long vox = m0 + row;
      if (vox < n_out_vox) store_out(a, n, (size_t)vox, acc[nt][r] * sc + bi);
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
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin8; a.relu0 = relu; a.cout_total = 32; a.cout0 = 32;
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
static int set_lds_limit(K kernel) {
  PW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  return PW_OK;
}

PW_API int pw_conv3d_ndhwc(const float* x, const float* wpk, const float* scale, const float* bias,
                           const float* residual, float* y0, float* y1, int B, int D, int H, int W,
                           int Cin, int cout_total, int cout0, int cout1, int ksize, int stride,
                           int relu0, int relu1, int algo, void* stream) {
  PW_CHECK_ARG(x && wpk && y0, "pw_conv3d_ndhwc: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pw_conv3d_ndhwc: bad shape");
  PW_CHECK_ARG(Cin > 0 && Cin % KC == 0, "pw_conv3d_ndhwc: Cin must be a multiple of 32 (got %d)", Cin);
  PW_CHECK_ARG(cout_total > 0 && cout_total % 32 == 0, "pw_conv3d_ndhwc: cout_total must be a multiple of 32");
  PW_CHECK_ARG(cout0 > 0 && cout0 <= cout_total && cout1 >= 0, "pw_conv3d_ndhwc: bad cout split");
  PW_CHECK_ARG(ksize == 1 || ksize == 3, "pw_conv3d_ndhwc: kernel size must be 1 or 3");
  PW_CHECK_ARG(stride == 1 || stride == 2, "pw_conv3d_ndhwc: stride must be 1 or 2");
  PW_CHECK_ARG(!(ksize == 1 && stride != 1), "pw_conv3d_ndhwc: 1x1x1 stride 2 unsupported");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)wpk) & 15) == 0, "pw_conv3d_ndhwc: x/wpk must be 16-B aligned");
  const int pad = ksize / 2;
  ConvArgs a;
  a.x = x; a.wpk = wpk; a.scale = scale; a.bias = bias; a.residual = residual; a.y0 = y0; a.y1 = y1;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin;
  a.Do = (D + 2 * pad - ksize) / stride + 1;
  a.Ho = (H + 2 * pad - ksize) / stride + 1;
  a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.n1_start = (cout0 + 31) / 32 * 32;
  a.relu0 = relu0; a.relu1 = relu1;
  a.tiles_d = (a.Do + BD - 1) / BD; a.tiles_h = (a.Ho + BH - 1) / BH; a.tiles_w = (a.Wo + BW - 1) / BW;
  PW_CHECK_ARG(!(cout1 > 0 && !y1), "pw_conv3d_ndhwc: cout1 > 0 needs y1");
  PW_CHECK_ARG(a.n1_start + cout1 <= cout_total || cout1 == 0, "pw_conv3d_ndhwc: cout split exceeds cout_total");
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
  // algo: 0 = auto, 1 = force LDS-tiled (k3 s1 only), 2 = force gather
  const bool tiled = (ksize == 3 && stride == 1 && algo != 2);
  PW_CHECK_ARG(!(algo == 1 && !tiled), "pw_conv3d_ndhwc: algo=1 needs ksize 3 stride 1");
  a.probe = nullptr;
  {
    const char* e = getenv("PW_CONV_PROBE");     // development aid: address of a device buffer
    if (e) a.probe = (long long*)strtoull(e, nullptr, 0);
  }
  if (tiled) {
    long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
    PW_CHECK_ARG(nblk < (1ll << 31), "pw_conv3d_ndhwc: grid too large");
    dim3 grid((unsigned)nblk, (unsigned)ngroups);
    OccTail none = {};
    if (NT == 2) {
      static int once = set_lds_limit(k_conv3d_k3s1<2, 0>);
      if (once) return once;
      hipLaunchKernelGGL((k_conv3d_k3s1<2, 0>), grid, dim3(256), LDS_BYTES, st, a, none);
    } else {
      static int once = set_lds_limit(k_conv3d_k3s1<1, 0>);
      if (once) return once;
      hipLaunchKernelGGL((k_conv3d_k3s1<1, 0>), grid, dim3(256), LDS_BYTES, st, a, none);
    }
  } else {
    dim3 grid((unsigned)pw_cdiv(n_out, 128), (unsigned)ngroups);
#define PW_GATHER(NTv, KSv, STv) \
  hipLaunchKernelGGL((k_conv3d_gather<NTv, KSv, STv>), grid, dim3(256), 0, st, a, n_out)
    if (ksize == 1) {
      if (NT == 2) PW_GATHER(2, 1, 1); else PW_GATHER(1, 1, 1);
    } else if (stride == 1) {
      if (NT == 2) PW_GATHER(2, 3, 1); else PW_GATHER(1, 3, 1);
    } else {
      if (NT == 2) PW_GATHER(2, 3, 2); else PW_GATHER(1, 3, 2);
    }
#undef PW_GATHER
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// A11  fused OccHead: conv3x3x3 (Cin->16, BN, ReLU) + 1x1x1 16->8 (BN, ReLU) + 1x1x1 8->18 + argmax
PW_API int pw_occ_head_fused(const float* x, const float* wpk, const float* scale, const float* bias,
                             const float* w1, const float* s1, const float* b1, const float* w2,
                             uint8_t* occ, float* logits, int B, int D, int H, int W, int Cin,
                             int n_mid, int n_hid, int n_cls, int wpk_layout, void* stream) {
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
  OccTail t = {w1, s1, b1, w2, occ, logits, n_mid, n_hid, n_cls};
  long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (wpk_layout == 16) {
    static int once16 = set_lds_limit(k_occ_head16);
    if (once16) return once16;
    hipLaunchKernelGGL(k_occ_head16, dim3((unsigned)nblk, 1), dim3(256), LDS_BYTES, pw_stream(stream),
                       a, t);
  } else {
    PW_CHECK_ARG(wpk_layout == 32, "pw_occ_head_fused: wpk_layout must be 16 or 32");
    static int once = set_lds_limit(k_conv3d_k3s1<1, 1>);
    if (once) return once;
    hipLaunchKernelGGL((k_conv3d_k3s1<1, 1>), dim3((unsigned)nblk, 1), dim3(256), LDS_BYTES,
                       pw_stream(stream), a, t);
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
