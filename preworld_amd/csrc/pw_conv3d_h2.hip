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

template <int NT>
struct H2Ctx {
  lds3_t lds3;
  rsrc_t xr, wr;
  unsigned lane_off, wstride;
  unsigned wsoff, wsoff_next;
  bool has_next;
  PipeDma dm;
  int wave, lane;
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

template <int NT>
__device__ __forceinline__ void h2_load_b(rsrc_t wr, unsigned wsoff, unsigned lane_off, v4f (&b)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(wr, lane_off + (unsigned)(q * 16), wsoff + (unsigned)(nt * 4096), 0);
      v4f o;
      o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
      b[nt][q] = o;
    }
}

// slots of a lane: q = 2*ks + p (p = 0 hi, 1 lo) for activations (aq) and weights (b) alike
template <int NT>
__device__ __forceinline__ void h2_mfma(const v4f (&aq)[2][4], const v4f (&b)[NT][4], f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int prod = 0; prod < 3; ++prod) {          // hi_w.hi_x, lo_w.hi_x, hi_w.lo_x
      const int pw = prod == 1 ? 1 : 0, px = prod == 2 ? 1 : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, b[nt][2 * ks + pw]),
                                                               __builtin_bit_cast(h8, aq[mt][2 * ks + px]), acc[mt][nt], 0, 0, 0);
    }
  }
}

template <int NT, int TAP>
__device__ __forceinline__ void h2_step(const ConvArgs& a, const H2Ctx<NT>& c, const unsigned (&aaddr)[2][3][4],
                                        v4f (&ac)[2][4], v4f (&an)[2][4], v4f (&b0)[NT][4], v4f (&b1)[NT][4],
                                        v4f (&b2)[NT][4], f32x16 (&acc)[2][NT]) {
  if (!(a.dma_stage & 1)) {
  if constexpr (TAP + 2 < 27) {
    h2_load_b<NT>(c.wr, c.wsoff + (unsigned)(TAP + 2) * c.wstride, c.lane_off, b2);
  } else {
    h2_load_b<NT>(c.wr, c.wsoff_next + (unsigned)(TAP + 2 - 27) * c.wstride, c.lane_off, b2);
  }
  }
  if (!(a.dma_stage & 4)) {
  if constexpr (TAP >= 1 && TAP <= PIPE_ROWS_PER_WAVE) pipe_dma_row<TAP - 1>(a, c.xr, c.lds3, c.dm, c.wave);
  }
  if (!(a.dma_stage & 2)) {
  if constexpr (TAP < 26) h2_read_a_tap<TAP + 1>(c.lds3, aaddr, an);
  }
  __builtin_amdgcn_sched_barrier(0);
  h2_mfma<NT>(ac, b0, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP < 26) h2_step<NT, TAP + 1>(a, c, aaddr, an, ac, b1, b2, b0, acc);
}

template <int NT>
__global__ void __launch_bounds__(256, 1) k_conv3d_h2(ConvArgs a, PipeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);                  // = d-slice of the tile
  const int half = lane >> 5, j = lane & 31;
  const int pj = patch_of_row(j), pr = pj >> 3, pc = pj & 7;
  const int ntiles_total = a.cout_total >> 5;
  const int nchunk = a.Cin / KC;

  // folded-BN scale / bias of every packed column -> LDS (read back as float4 per channel group in the epilogue)
  {
    float* sb = lds + H2_SB_OFF / 4;
    for (int n = tid; n < a.cout_total; n += 256) {
      sb[n] = a.scale ? a.scale[n] : 1.f;
      sb[H2_MAX_COUT + n] = a.bias ? a.bias[n] : 0.f;
    }
  }

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
  c.lane_off = (unsigned)lane * 64u;
  c.wstride = (unsigned)ntiles_total * 4096u;
  c.wave = wave; c.lane = lane;

  PipeTile t = pipe_decode(a, p, item);
  int ch = 0;
  v4f a0[2][4], a1[2][4], b0[NT][4], b1[NT][4], b2[NT][4];
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
    h2_load_b<NT>(c.wr, (unsigned)((t.ng * NT) * 4096), c.lane_off, b0);
    h2_load_b<NT>(c.wr, (unsigned)((t.ng * NT) * 4096) + c.wstride, c.lane_off, b1);
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
          asm volatile("" : "+v"(aaddr[khp][kw][q]));       // one address register per variant, tap offset = immediate
        }
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

    h2_step<NT, 0>(a, c, aaddr, a0, a1, b0, b1, b2, acc);

    __builtin_amdgcn_s_waitcnt(0);     // my DMA rows of the next stage have landed
    __syncthreads();                   // everyone's have; everyone is done with this buffer

    if (ch == nchunk - 1) {
      // ---- epilogue: y = acc*scale + bias (+residual) (ReLU); lane = one voxel per M-tile, 16 channels per N-tile
      const int od = t.d0 + wave;
      const unsigned out_vox = (unsigned)((size_t)a.B * a.Do * a.Ho * a.Wo);
      const float* sb = lds + H2_SB_OFF / 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (t.ng * NT + nt) * 32;
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
        v4f sc[4], bi[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          sc[g] = *reinterpret_cast<const v4f*>(sb + n0 + 8 * g + 4 * half);
          bi[g] = *reinterpret_cast<const v4f*>(sb + H2_MAX_COUT + n0 + 8 * g + 4 * half);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bool ok = od < a.Do && (t.h0 + mt * 4 + pr) < a.Ho && (t.w0 + pc) < a.Wo;
          const unsigned soff = (unsigned)(((((t.b * a.Do + od) * a.Ho + (t.h0 + mt * 4)) * a.Wo + t.w0) * ld) * 4);
          const unsigned vox = (unsigned)((pr * a.Wo + pc) * ld) * 4u + (unsigned)col0 * 4u;
          // residual values first (fp32: 4 x 16 B; h2: 4 x (8 + 8) B), then the math, then the stores
          float rv[4][4];
          if (has_res) {
            if (a.fmt_res == 0) {
              const unsigned base = ok ? vox + (unsigned)(16 * half) : PIPE_OOB;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 r4 = buf_load4(rr, base + (unsigned)(32 * g), soff);
                rv[g][0] = r4.x; rv[g][1] = r4.y; rv[g][2] = r4.z; rv[g][3] = r4.w;
              }
            } else {
              const unsigned base = ok ? vox + (unsigned)(8 * half) : PIPE_OOB;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const unsigned so = (unsigned)((4 * (g & 1) + 2 * (g >> 1)) * 16);
                const u2 hi = buf_load2(rr, base + so, soff), lo = buf_load2(rr, base + so + 16u, soff);
                h2_join4(hi, lo, rv[g]);
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[mt][nt][4 * g + e] * sc[g][e] + bi[g][e];
              if (has_res) v[e] += rv[g][e];
              v[e] = fmaxf(v[e], lo_clamp);
            }
            if (fmt == 0) {
              buf_store4(yr, ok ? vox + (unsigned)(16 * half + 32 * g) : PIPE_OOB, soff, v);
            } else {
              u2 hi, lo;
              h2_split4(v, hi, lo);
              const unsigned o = ok ? vox + (unsigned)(8 * half + (4 * (g & 1) + 2 * (g >> 1)) * 16) : PIPE_OOB;
              buf_store2(yr, o, soff, hi);
              buf_store2(yr, ok ? o + 16u : PIPE_OOB, soff, lo);
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
    if (!c.has_next) break;
    t = tn; ch = chn; item = itemn;
  }
}

// ------------------------------------------------------------------------------------ fp32 <-> h2
// one thread per (voxel, 4-channel group); ld_* = floats between consecutive voxels (channel slices of wider buffers)
__global__ void k_f32_to_h2(const float* __restrict__ x, float* __restrict__ y, long long n_vox, int C, int ld_x, int ld_y) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = C >> 2;
  if (idx >= n_vox * groups) return;
  const long long v = idx / groups;
  const int c = (int)(idx - v * groups) * 4;
  const float4 f = *reinterpret_cast<const float4*>(x + v * ld_x + c);
  const float in[4] = {f.x, f.y, f.z, f.w};
  u2 hi, lo;
  h2_split4(in, hi, lo);
  char* dst = reinterpret_cast<char*>(y + v * ld_y + (c & ~31));
  *reinterpret_cast<u2*>(dst + h2_group_off(c & 31, 0)) = hi;
  *reinterpret_cast<u2*>(dst + h2_group_off(c & 31, 1)) = lo;
}

__global__ void k_h2_to_f32(const float* __restrict__ x, float* __restrict__ y, long long n_vox, int C, int ld_x, int ld_y) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = C >> 2;
  if (idx >= n_vox * groups) return;
  const long long v = idx / groups;
  const int c = (int)(idx - v * groups) * 4;
  const char* src = reinterpret_cast<const char*>(x + v * ld_x + (c & ~31));
  const u2 hi = *reinterpret_cast<const u2*>(src + h2_group_off(c & 31, 0));
  const u2 lo = *reinterpret_cast<const u2*>(src + h2_group_off(c & 31, 1));
  float out[4];
  h2_join4(hi, lo, out);
  *reinterpret_cast<float4*>(y + v * ld_y + c) = make_float4(out[0], out[1], out[2], out[3]);
}

PW_API int pw_f32_to_h2(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, void* stream) {
  PW_CHECK_ARG(x && y && n_vox > 0 && C > 0 && C % 32 == 0, "pw_f32_to_h2: C must be a positive multiple of 32");
  if (ld_x <= 0) ld_x = C;
  if (ld_y <= 0) ld_y = C;
  PW_CHECK_ARG(ld_x >= C && ld_y >= C && ld_x % 4 == 0 && ld_y % 32 == 0, "pw_f32_to_h2: bad row strides");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_f32_to_h2: pointers must be 16-B aligned");
  const long long n = n_vox * (C / 4);
  hipLaunchKernelGGL(k_f32_to_h2, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream), x, y, (long long)n_vox, C, ld_x, ld_y);
  pw_note_kernel("k_f32_to_h2");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_h2_to_f32(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, void* stream) {
  PW_CHECK_ARG(x && y && n_vox > 0 && C > 0 && C % 32 == 0, "pw_h2_to_f32: C must be a positive multiple of 32");
  if (ld_x <= 0) ld_x = C;
  if (ld_y <= 0) ld_y = C;
  PW_CHECK_ARG(ld_x >= C && ld_y >= C && ld_x % 32 == 0 && ld_y % 4 == 0, "pw_h2_to_f32: bad row strides");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_h2_to_f32: pointers must be 16-B aligned");
  const long long n = n_vox * (C / 4);
  hipLaunchKernelGGL(k_h2_to_f32, dim3((unsigned)pw_cdiv(n, 256)), dim3(256), 0, pw_stream(stream), x, y, (long long)n_vox, C, ld_x, ld_y);
  pw_note_kernel("k_h2_to_f32");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------ host entry
PW_API int pw_conv3d_h2(const float* x, const float* wpk, const float* scale, const float* bias, const float* residual,
                        float* y0, float* y1, int B, int D, int H, int W, int Cin, int cout_total, int cout0, int cout1,
                        int ld_y0, int ld_y1, int relu0, int relu1, int fmt_y0, int fmt_y1, int fmt_res, void* stream) {
  PW_CHECK_ARG(x && wpk && y0, "pw_conv3d_h2: null pointer");
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
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Do = D; a.Ho = H; a.Wo = W;
  a.cout_total = cout_total; a.cout0 = cout0; a.cout1 = cout1;
  a.ld0 = ld_y0 > 0 ? ld_y0 : cout0; a.ld1 = ld_y1 > 0 ? ld_y1 : cout1;
  PW_CHECK_ARG(a.ld0 >= cout0 && a.ld1 >= cout1 && a.ld0 % 32 == 0 && a.ld1 % 32 == 0,
               "pw_conv3d_h2: ld_y0 / ld_y1 must be multiples of 32 >= the channel counts");
  a.n1_start = cout0;
  a.relu0 = relu0; a.relu1 = relu1;
  a.fmt_y0 = fmt_y0; a.fmt_y1 = fmt_y1; a.fmt_res = fmt_res;
  if (const char* e = getenv("PW_H2_DEBUG")) a.dma_stage = atoi(e);     // TEMP sensitivity runs
  a.tiles_d = (D + BD - 1) / BD; a.tiles_h = (H + BH - 1) / BH; a.tiles_w = (W + BW - 1) / BW;
  PW_CHECK_ARG((size_t)B * D * H * W * Cin * 4 < (1ull << 32) &&
                   (size_t)B * D * H * W * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) * 4 < (1ull << 32),
               "pw_conv3d_h2: tensors must be < 4 GiB (32-bit buffer addressing)");
  const int ntiles = cout_total / 32;
  const long long nblk = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
  // two N-tiles per wave (A fragments shared) when that still leaves every CU two or more work items
  int NT = (ntiles % 2 == 0 && nblk * (ntiles / 2) >= 2 * pw_num_cus()) ? 2 : 1;
  if (const char* e = getenv("PW_H2_NT")) {          // experiments only
    const int f = atoi(e);
    if ((f == 1 || f == 2) && ntiles % f == 0) NT = f;
  }
  PipeArgs p;
  p.ngroups = ntiles / NT;
  PW_CHECK_ARG(nblk * p.ngroups < (1ll << 20), "pw_conv3d_h2: too many work items");
  p.n_items = (int)(nblk * p.ngroups);
  p.m_ng = magic_of(p.ngroups); p.m_tw = magic_of(a.tiles_w); p.m_th = magic_of(a.tiles_h); p.m_td = magic_of(a.tiles_d);
  const unsigned nb = (unsigned)(pw_num_cus() / 8 * 8);
  hipStream_t st = pw_stream(stream);
  if (NT == 2) {
    static int once = set_lds_limit(k_conv3d_h2<2>, H2_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_conv3d_h2<2>, dim3(nb), dim3(256), H2_LDS, st, a, p);
    pw_note_kernel("k_conv3d_h2<2>");
  } else {
    static int once = set_lds_limit(k_conv3d_h2<1>, H2_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_conv3d_h2<1>, dim3(nb), dim3(256), H2_LDS, st, a, p);
    pw_note_kernel("k_conv3d_h2<1>");
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
