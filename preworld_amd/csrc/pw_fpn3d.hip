// LSSFPN3D fused tail (A8) -- entry point pw_fpn3d_fuse in include/preworld_hip.h.
#include "pw_h2.h"

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
  // range exponents (pw_h2.h "Range"): x8 is read under x_rng (y16 / y32 are fp32, true units), out is written under y0_rng.
  // Loaded here, first used in the epilogue.
  const int e_out = a.fmt_y0 ? rng_exp(a.y0_rng) : 0;
  const int e_x = a.dma_stage ? rng_exp(a.x_rng) : 0;
  long long m = m0 + i;
  if (m >= n_vox) m = n_vox - 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nchunk = a.Cin / KC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* src = a.x + (size_t)m * a.Cin + ch * KC + half * 16;
    // a weight tile = four 1024-byte pieces of 64 lanes x 16 B (pw_conv3d_common.h WPIECE), fp32 and split-fp16 alike
    const float* wt = a.wpk + (size_t)ch * 1024 + lane * 4;
    constexpr int wq = (int)(WPIECE / 4);
    float4 aq[4], bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      aq[q] = *reinterpret_cast<const float4*>(src + q * 4);
      bq[q] = *reinterpret_cast<const float4*>(wt + q * wq);
    }
    if (a.dma_stage) {
      // x8 and wpk8 in split-fp16 storage (pw_h2.h): the lane's four 16-byte pieces are {hi, lo} x {k-step 0, 1}
      typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, aq[2 * ks]), __builtin_bit_cast(fh8, bq[2 * ks]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, aq[2 * ks]), __builtin_bit_cast(fh8, bq[2 * ks + 1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, aq[2 * ks + 1]), __builtin_bit_cast(fh8, bq[2 * ks]), acc, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[4] = {aq[q].x, aq[q].y, aq[q].z, aq[q].w};
      const float bv[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
    }
  }
  const float xmul = rng_pow2(e_x), omul = rng_pow2(-e_out);
  const float sc = (a.scale ? a.scale[i] : 1.f) * omul;
  const float bi = (a.bias ? a.bias[i] : 0.f) * omul;
  float amax = 0.f;
  const float sd2 = a.D > 1 ? (float)(f.D2 - 1) / (float)(a.D - 1) : 0.f;
  const float sh2 = a.H > 1 ? (float)(f.H2 - 1) / (float)(a.H - 1) : 0.f;
  const float sw2 = a.W > 1 ? (float)(f.W2 - 1) / (float)(a.W - 1) : 0.f;
  const float sd4 = a.D > 1 ? (float)(f.D4 - 1) / (float)(a.D - 1) : 0.f;
  const float sh4 = a.H > 1 ? (float)(f.H4 - 1) / (float)(a.H - 1) : 0.f;
  const float sw4 = a.W > 1 ? (float)(f.W4 - 1) / (float)(a.W - 1) : 0.f;
  // The 32 rows of the tile are 32 consecutive voxels along w.  When they stay inside one (b, d, h) line -- 84 % of
  // the tiles at W = 200 -- the voxel decode and the d / h interpolation terms are wave-uniform and computed once;
  // only the w terms differ per row.  (Per-row div/mod + 3-axis coefficients were ~2/3 of this kernel's VALU.)
  const unsigned um0 = (unsigned)m0;
  const unsigned q1 = um0 / (unsigned)a.W;
  const int ow0 = (int)(um0 - q1 * (unsigned)a.W);
  // source-column windows of the tile at both levels (<= 20 columns each for the 1/2 and 1/4 maps)
  const int wb2 = (int)(sw2 * (float)ow0), wl2 = min((int)(sw2 * (float)(ow0 + 31)) + 1, f.W2 - 1);
  const int wb4 = (int)(sw4 * (float)ow0), wl4 = min((int)(sw4 * (float)(ow0 + 31)) + 1, f.W4 - 1);
  if (ow0 + 32 <= a.W && m0 + 32 <= n_vox && wl2 - wb2 < 20 && wl4 - wb4 < 20) {
    const unsigned q2 = q1 / (unsigned)a.H;
    const int oh = (int)(q1 - q2 * (unsigned)a.H);
    const int b = (int)(q2 / (unsigned)a.D);
    const int od = (int)(q2 - (unsigned)b * (unsigned)a.D);
    // Two-stage trilinear: (1) the low-resolution source columns this tile touches (<= 18 at 1/2, <= 10 at 1/4
    // resolution) are interpolated along d and h ONCE per wave and parked in LDS -- 4 loads per column instead of 8
    // corner loads per output voxel and level; (2) every output voxel finishes with one lerp along w between two LDS
    // values per level.  (Interpolating d/h before w reassociates the reference's w-h-d order: ~1e-7 relative.)
    __shared__ float cols[4][2][20][32];
    float (&c2)[20][32] = cols[wave][0];
    float (&c4)[20][32] = cols[wave][1];
    auto stage = [&](const float* y, int Dl, int Hl, int Wl, float sd, float sh, float (&dst)[20][32], int wbase, int wlast) {
      const float fd = sd * (float)od, fh = sh * (float)oh;
      const int d0 = (int)fd, h0 = (int)fh;
      const int d1 = d0 + (d0 < Dl - 1), h1 = h0 + (h0 < Hl - 1);
      const float ld1 = fd - (float)d0, ld0 = 1.f - ld1, lh1 = fh - (float)h0, lh0 = 1.f - lh1;
      const float* base = y + (size_t)b * Dl * Hl * Wl * 32 + i;
      const float* p00 = base + (size_t)((d0 * Hl + h0) * Wl) * 32; const float* p01 = base + (size_t)((d0 * Hl + h1) * Wl) * 32;
      const float* p10 = base + (size_t)((d1 * Hl + h0) * Wl) * 32; const float* p11 = base + (size_t)((d1 * Hl + h1) * Wl) * 32;
      for (int j = wbase + half; j <= wlast; j += 2) {
        const unsigned o = (unsigned)j * 32u;
        dst[j - wbase][i] = ld0 * (lh0 * p00[o] + lh1 * p01[o]) + ld1 * (lh0 * p10[o] + lh1 * p11[o]);
      }
    };
    stage(f.y16, f.D2, f.H2, f.W2, sd2, sh2, c2, wb2, wl2);
    stage(f.y32, f.D4, f.H4, f.W4, sd4, sh4, c4, wb4, wl4);
    __builtin_amdgcn_wave_barrier();                    // a wave's LDS accesses execute in program order: no s_barrier
    auto lerp_w = [&](const float (&src)[20][32], int Wl, float sw, int wbase, int ow) {
      const float fw = sw * (float)ow;
      const int w0 = (int)fw;
      const int w1 = w0 + (w0 < Wl - 1);
      const float lw1 = fw - (float)w0;
      return (1.f - lw1) * src[w0 - wbase][i] + lw1 * src[w1 - wbase][i];
    };
    float* out = a.y0 + (size_t)m0 * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] * xmul;
      v += lerp_w(c2, f.W2, sw2, wb2, ow0 + row);
      v += lerp_w(c4, f.W4, sw4, wb4, ow0 + row);
      v = v * sc + bi;
      if (a.relu0) v = fmaxf(v, 0.f);
      amax = fmaxf(amax, fabsf(v));
      if (a.fmt_y0) h2_store_elem(a.y0 + (size_t)(m0 + row) * 32, i, v);
      else out[(unsigned)row * 32u] = v;
    }
    if (a.fmt_y0) rng_note(a.y0_rng, __float_as_uint(amax), e_out);
    return;
  }
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
      float v = acc[r] * xmul;
      v += trilerp_ac(f.y16, b, f.D2, f.H2, f.W2, sd2, sh2, sw2, od, oh, ow, i);
      v += trilerp_ac(f.y32, b, f.D4, f.H4, f.W4, sd4, sh4, sw4, od, oh, ow, i);
      v = v * sc + bi;
      if (a.relu0) v = fmaxf(v, 0.f);
      amax = fmaxf(amax, fabsf(v));
      if (a.fmt_y0) h2_store_elem(a.y0 + (size_t)vox * 32, i, v);
      else a.y0[(size_t)vox * 32 + i] = v;
    }
  }
  if (a.fmt_y0) rng_note(a.y0_rng, __float_as_uint(amax), e_out);
}

PW_API int pw_fpn3d_fuse(const float* x8, const float* wpk8, const float* y16, const float* y32,
                         const float* scale, const float* bias, float* out, int B, int D, int H,
                         int W, int Cin8, int D2, int H2, int W2, int D4, int H4, int W4, int relu,
                         int x_h2, int out_h2, const int32_t* x_rng, int32_t* out_rng, void* stream) {
  PW_CHECK_ARG(x8 && wpk8 && y16 && y32 && out, "pw_fpn3d_fuse: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin8 > 0 && Cin8 % 32 == 0, "pw_fpn3d_fuse: bad shape");
  PW_CHECK_ARG(D2 > 0 && H2 > 0 && W2 > 0 && D4 > 0 && H4 > 0 && W4 > 0, "pw_fpn3d_fuse: bad level shape");
  ConvArgs a = {};
  a.x = x8; a.wpk = wpk8; a.scale = scale; a.bias = bias; a.y0 = out;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin8; a.relu0 = relu; a.cout_total = 32; a.cout0 = 32; a.ld0 = 32;
  a.dma_stage = x_h2 ? 1 : 0;          // reused as the input-format flag: x8 and wpk8 are split-fp16
  a.fmt_y0 = out_h2 ? 1 : 0;
  a.x_rng = x_rng; a.y0_rng = out_rng;
  FpnArgs f = {y16, y32, D2, H2, W2, D4, H4, W4};
  const long long n = (long long)B * D * H * W;
  PW_CHECK_ARG(n < (1ll << 31), "pw_fpn3d_fuse: more than 2^31 voxels");
  hipLaunchKernelGGL(k_fpn3d_fuse, dim3((unsigned)pw_cdiv(n, 128)), dim3(256), 0, pw_stream(stream),
                     a, f, n);
  pw_note_kernel("k_fpn3d_fuse");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

