// Gather (no LDS tile) convolution kernel: stride-2, 1x1x1, 2x2x2 and tiny grids -- see pw_conv3d.hip for
// the GEMM view and include/preworld_hip.h (pw_conv3d_ndhwc) for the entry point.
#include "pw_h2.h"

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

typedef _Float16 gh8 __attribute__((ext_vector_type(8)));

// F16: x and weights in split-fp16 (h2) storage -- a lane's four 16-byte pieces are {hi, lo} x {k-step 0, 1} -- and the
// product block is three v_mfma_f32_32x32x16_f16 (pw_h2.h); otherwise the exact-fp32 v_mfma_f32_32x32x2_f32
template <int NT, int MT, bool F16>
__device__ __forceinline__ void gather_mfma(const float4 (&aq)[MT][4], const float4 (&bq)[NT][4], f32x16 (&acc)[MT][NT]) {
  if constexpr (F16) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int prod = 0; prod < 3; ++prod) {          // hi_x.hi_w, hi_x.lo_w, lo_x.hi_w
        const int pw = prod == 1 ? 1 : 0, px = prod == 2 ? 1 : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            // TRANSPOSED product D[cout][voxel] (weights as the A operand): a lane ends up with 4 consecutive output channels
            // of ONE voxel per register group, which the h2 epilogue stores as 8-byte pieces
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gh8, bq[nt][2 * ks + pw]),
                                                                 __builtin_bit_cast(gh8, aq[mt][2 * ks + px]), acc[mt][nt], 0, 0, 0);
      }
    return;
  }
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
template <int NT, int KS, int MT, int TAP, bool F16>
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
  gather_mfma<NT, MT, F16>(ac, bc, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP + 1 < TAPS)
    gather_step<NT, KS, MT, TAP + 1, F16>(c, ch, ch_step, more_chunks, wsoff, wsoff_next, an, bn, ac, bc, acc);
}

// MT = M-tiles (32 output voxels each) per wave (default 1, see the dispatch)
// ksplit (1, 2 or 4): the block's 4 waves are 4/ksplit M-groups x ksplit partitions of the input-channel
// chunks (wave w: M-group w / ksplit, chunks w % ksplit, + ksplit, ...).  The partial accumulators
// meet in LDS and partition 0 adds them in a fixed order (deterministic) before the epilogue.  Small
// grids need this: with one (M-tile, N-group) per wave the 4x50x50 stage has ~1.2 waves of 1728-3456
// MFMAs per SIMD, i.e. the slowest SIMD does 2 of them; split by 4 it is ~5 waves of 432.
// H2EPI (F16 only): every destination in h2 storage and no residual -> the vector epilogue (chosen by the launcher)
template <int NT, int KS, int STRIDE, int MT, int KSPL, bool F16 = false, bool H2EPI = false>
__global__ void __launch_bounds__(256) k_conv3d_gather(ConvArgs a, long long n_out_vox) {
  constexpr int ksplit = KSPL;          // compile-time: the unsplit kernel keeps its straight-line code
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int tid = threadIdx.x, lane = tid & 63;
  // scalar wave index: the chunk partition, hence the scalar offsets / descriptors of every A and weight load, derive from it;
  // left in a VGPR, hipcc wraps each of those loads in a waterfall loop (readfirstlane + branch)
  const int wave = uni(tid >> 6);
  const int half = lane >> 5, i = lane & 31;
  const int mslot = wave / ksplit, kpart = wave - mslot * ksplit;
  const long long m0 = ((long long)blockIdx.x * (4 / ksplit) + mslot) * (32 * MT);
  const bool active = m0 < n_out_vox;                 // wave-uniform; inactive waves still meet the barriers
  const int ng = blockIdx.y;
  const int ntiles_total = a.cout_total >> 5;
  constexpr int TAPS = KS * KS * KS;
  constexpr int PAD = (KS - 1) / 2;       // k3: 1, k2 (stride-2 patchify, A20): 0, k1: 0

  RngScale rs = {};                                  // split-fp16 kernels: loaded here, first used in the epilogue
  if constexpr (F16) rs = rng_scales(a);
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
    c.lane_off = (unsigned)lane * 16u;
    c.wstride = (unsigned)ntiles_total * 4096u;

    float4 a0[MT][4], a1[MT][4], b0[NT][4], b1[NT][4];
    if (kpart < nchunk)
      gather_load<NT, KS, MT, 0>(c, kpart, (unsigned)((kpart * TAPS * ntiles_total + ng * NT) * 4096), a0, b0);
    for (int ch = kpart; ch < nchunk; ch += ksplit) {
      const unsigned wsoff = (unsigned)((ch * TAPS * ntiles_total + ng * NT) * 4096);
      const unsigned wsoff_next = (unsigned)(((ch + ksplit) * TAPS * ntiles_total + ng * NT) * 4096);
      gather_step<NT, KS, MT, 0, F16>(c, ch, ksplit, ch + ksplit < nchunk, wsoff, wsoff_next, a0, b0, a1, b1, acc);
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
  if constexpr (F16) {
    // transposed accumulators: column = voxel i of the M-tile, register r = output channel (r & 3) + 8 (r >> 2) + 4 half of
    // the N-tile.  All-h2 destinations without a residual (every stride-2 / 1x1x1 layer of the encoder and the neck) get the
    // vector epilogue: float4 scale / bias, 8-byte hi + 8-byte lo stores per 4 channels.  (The first version kept the
    // voxel-major product and stored element by element: 2 x 2-byte stores and ~10 address instructions per output, about as
    // many instructions as the whole tap loop.)
    // Range exponents of the operands (pw_h2.h "Range") are folded into scale / bias; the magnitudes written are recorded.
    RngEpi re = {rs.res, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long long vox = m0 + mt * 32 + i;
      if (vox >= n_out_vox) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (ng * NT + nt) * 32;                 // a 32-column tile lies in one destination (cout0 % 32 == 0)
        const bool first = n0 < a.cout0;
        const int nn0 = first ? n0 : n0 - a.n1_start;
        if (!first && !(a.y1 && nn0 >= 0 && nn0 < a.cout1)) continue;
        float* row = first ? a.y0 + vox * a.ld0 : a.y1 + vox * a.ld1;
        const bool relu = first ? a.relu0 != 0 : a.relu1 != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 8 * q + 4 * half;                   // first of this lane's 4 channels inside the tile
          float sc[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
          if (a.scale) { const float4 t = *reinterpret_cast<const float4*>(a.scale + n0 + c); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
          if (a.bias) { const float4 t = *reinterpret_cast<const float4*>(a.bias + n0 + c); bi[0] = t.x; bi[1] = t.y; bi[2] = t.z; bi[3] = t.w; }
          float v[4];
          const float sm = first ? rs.s0 : rs.s1, bm = first ? rs.b0 : rs.b1;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * q + e] * (sc[e] * sm) + bi[e] * bm;
          if constexpr (H2EPI) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
            const float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            if (first) re.amax0 = fmaxf(re.amax0, m4); else re.amax1 = fmaxf(re.amax1, m4);
            u2 hi, lo;
            h2_split4(v, hi, lo);
            char* chunk = reinterpret_cast<char*>(row + ((nn0 + c) & ~31));
            *reinterpret_cast<u2*>(chunk + h2_group_off((nn0 + c) & 31, 0)) = hi;
            *reinterpret_cast<u2*>(chunk + h2_group_off((nn0 + c) & 31, 1)) = lo;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_out(a, n0 + c + e, (size_t)vox, v[e], &re);
          }
        }
      }
    }
    if (a.fmt_y0) rng_note(a.y0_rng, __float_as_uint(re.amax0), rs.e0);
    if (a.fmt_y1 && a.y1) rng_note(a.y1_rng, __float_as_uint(re.amax1), rs.e1);
    return;
  }
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

int pw_launch_conv3d_gather(const ConvArgs& a, int NT, int ngroups, int ksize, int stride, int algo, int Cin,
                             long long n_out, hipStream_t st, bool f16) {
    // M-tiles per wave: 2 halves the weight loads per MFMA but measured slower on the fp32 pipe (32->128 stride 2:
    // 199 us vs 180 us -- fewer waves to hide the L2 gather latency)
    int MT = 1;
    if (f16 && ksize == 3 && stride == 2) {
      // split-fp16 stride-2 layers: the 32-cycle MFMA makes the weight loads weigh 4x more; two voxel tiles per wave win once the
      // launch has few blocks anyway (8x100x100 64->2x128: 107 -> 88 us) and lose while there are enough (16x200x200 32->2x64:
      // 135 -> 152 us)
      MT = (pw_cdiv(n_out, 128) * ngroups < 512) ? 2 : 1;
    }
    const int nchunk = Cin / KC;
    int ksplit = 1;
    if (algo == 3) ksplit = (nchunk % 4 == 0) ? 4 : (nchunk % 2 == 0 ? 2 : 1);
    if (ksplit > 1 && MT != 1) ksplit = 1;              // the split variants are built for MT = 1
    const int mgroups = 4 / ksplit;                     // M-groups (32*MT voxels each) per block
    dim3 grid((unsigned)pw_cdiv(n_out, 32 * MT * mgroups), (unsigned)ngroups);
    const size_t red_bytes = ksplit > 1 ? (size_t)mgroups * (ksplit - 1) * MT * NT * 4096 : 0;
#define PW_GATHER_L(NTv, KSv, STv, MTv, KSPv)                                                                    \
  do {                                                                                                           \
    hipLaunchKernelGGL((k_conv3d_gather<NTv, KSv, STv, MTv, KSPv>), grid, dim3(256), red_bytes, st, a, n_out);   \
    pw_note_kernel("k_conv3d_gather<%d, %d, %d, %d, %d>", NTv, KSv, STv, MTv, KSPv);                             \
  } while (0)
  if (f16) {          // split-fp16 operands (pw_conv3d_h2): k3 s2, k1 s1 and k3 s1 (tiny grids), one M-tile per wave
    const bool h2epi = a.fmt_y0 == 1 && (a.cout1 == 0 || a.fmt_y1 == 1) && !a.residual;
#define PW_GATHER_F(NTv, KSv, STv, KSPv)                                                                                        \
  do {                                                                                                                          \
    if (h2epi) hipLaunchKernelGGL((k_conv3d_gather<NTv, KSv, STv, 1, KSPv, true, true>), grid, dim3(256), red_bytes, st, a, n_out); \
    else hipLaunchKernelGGL((k_conv3d_gather<NTv, KSv, STv, 1, KSPv, true, false>), grid, dim3(256), red_bytes, st, a, n_out);  \
    pw_note_kernel("k_conv3d_gather<%d, %d, %d, 1, %d, true, %s>", NTv, KSv, STv, KSPv, h2epi ? "true" : "false");                                            \
  } while (0)
#define PW_GATHER_FK(NTv, KSv, STv)                      \
  do {                                                   \
    if (ksplit == 4) PW_GATHER_F(NTv, KSv, STv, 4);      \
    else if (ksplit == 2) PW_GATHER_F(NTv, KSv, STv, 2); \
    else PW_GATHER_F(NTv, KSv, STv, 1);                  \
  } while (0)
    if (ksize == 1 && stride == 1) {
      if (NT == 2) PW_GATHER_FK(2, 1, 1); else PW_GATHER_FK(1, 1, 1);
    } else if (ksize == 3 && stride == 1) {
      if (NT == 2) PW_GATHER_FK(2, 3, 1); else PW_GATHER_FK(1, 3, 1);
    } else if (ksize == 3 && stride == 2 && MT == 2) {   // two voxel tiles per wave share the weight fragments (A/B switch)
      if (NT == 2) {
        if (h2epi) hipLaunchKernelGGL((k_conv3d_gather<2, 3, 2, 2, 1, true, true>), grid, dim3(256), red_bytes, st, a, n_out);
        else hipLaunchKernelGGL((k_conv3d_gather<2, 3, 2, 2, 1, true, false>), grid, dim3(256), red_bytes, st, a, n_out);
        pw_note_kernel("k_conv3d_gather<2, 3, 2, 2, 1, true, %s>", h2epi ? "true" : "false");
      } else {
        if (h2epi) hipLaunchKernelGGL((k_conv3d_gather<1, 3, 2, 2, 1, true, true>), grid, dim3(256), red_bytes, st, a, n_out);
        else hipLaunchKernelGGL((k_conv3d_gather<1, 3, 2, 2, 1, true, false>), grid, dim3(256), red_bytes, st, a, n_out);
        pw_note_kernel("k_conv3d_gather<1, 3, 2, 2, 1, true, %s>", h2epi ? "true" : "false");
      }
    } else if (ksize == 3 && stride == 2) {
      if (NT == 2) PW_GATHER_FK(2, 3, 2); else PW_GATHER_FK(1, 3, 2);
    } else {
      pw_set_error("pw_conv3d_h2: no split-fp16 gather variant for ksize %d stride %d", ksize, stride);
      return PW_EINVAL;
    }
#undef PW_GATHER_FK
#undef PW_GATHER_F
    return PW_OK;
  }
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
  return PW_OK;
}
