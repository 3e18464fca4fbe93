// LSSFPN3D fused tail (A8) -- entry point pw_fpn3d_fuse in include/preworld_hip.h.
#include <type_traits>

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

// WIDE: W >= 32 (the per-line path); otherwise the eight-corner path of tiny grids -- two instantiations so that the
// register budget of the one that matters (132 VGPRs) is not set by the other (206)
template <bool WIDE>
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
  const float sd2 = a.D > 1 ? (float)(f.D2 - 1) / (float)(a.D - 1) : 0.f;
  const float sh2 = a.H > 1 ? (float)(f.H2 - 1) / (float)(a.H - 1) : 0.f;
  const float sw2 = a.W > 1 ? (float)(f.W2 - 1) / (float)(a.W - 1) : 0.f;
  const float sd4 = a.D > 1 ? (float)(f.D4 - 1) / (float)(a.D - 1) : 0.f;
  const float sh4 = a.H > 1 ? (float)(f.H4 - 1) / (float)(a.H - 1) : 0.f;
  const float sw4 = a.W > 1 ? (float)(f.W4 - 1) / (float)(a.W - 1) : 0.f;
  // The 32 voxels of the tile are consecutive along w: for W >= 32 they lie on ONE line (b, d, h) -- 84 % of the tiles at W = 200 --
  // or on two consecutive lines.  Per line the voxel decode and the d / h interpolation terms are wave-uniform.
  const unsigned um0 = (unsigned)m0;
  const unsigned q1 = um0 / (unsigned)a.W;
  const int ow0 = (int)(um0 - q1 * (unsigned)a.W);
  const int n0 = min(32, a.W - ow0);                  // voxels of the tile on its first line
  int wb2[2] = {0, 0}, wb4[2] = {0, 0};
  __shared__ __attribute__((aligned(16))) float cols[WIDE ? 4 : 1][2][2][20][32];       // [wave][line][level][column][channel]
  // (the staging loads are issued BEFORE the x8 loads' MFMAs: nothing here depends on the product)
  if constexpr (WIDE) {
    // Two-stage trilinear: (1) the low-resolution source columns a line segment touches (<= 18 at 1/2, <= 10 at 1/4
    // resolution) are interpolated along d and h ONCE per wave and parked in LDS -- 4 loads per column instead of 8
    // corner loads per output voxel and level; (2) every output voxel finishes with one lerp along w between two LDS
    // rows per level.  (Interpolating d/h before w reassociates the reference's w-h-d order: ~1e-7 relative.)
    const unsigned q2 = q1 / (unsigned)a.H;
    int oh = (int)(q1 - q2 * (unsigned)a.H);
    int b = (int)(q2 / (unsigned)a.D);
    int od = (int)(q2 - (unsigned)b * (unsigned)a.D);
    auto stage = [&](auto nt, const float* y, int Dl, int Hl, int Wl, float sd, float sh, float (&dst)[20][32], int wbase, int wlast) {
      constexpr int NT = decltype(nt)::value;       // columns per lane half: 10 cover the 20-column window
      const float fd = sd * (float)od, fh = sh * (float)oh;
      const int d0 = (int)fd, h0 = (int)fh;
      const int d1 = d0 + (d0 < Dl - 1), h1 = h0 + (h0 < Hl - 1);
      const float ld1 = fd - (float)d0, ld0 = 1.f - ld1, lh1 = fh - (float)h0, lh0 = 1.f - lh1;
      const float* base = y + (size_t)b * Dl * Hl * Wl * 32 + i;
      const float* p00 = base + (size_t)((d0 * Hl + h0) * Wl) * 32; const float* p01 = base + (size_t)((d0 * Hl + h1) * Wl) * 32;
      const float* p10 = base + (size_t)((d1 * Hl + h0) * Wl) * 32; const float* p11 = base + (size_t)((d1 * Hl + h1) * Wl) * 32;
      // all columns' loads go out together (<= 4 NT in flight per lane): as a plain loop every trip waited on its own four
      float v00[NT], v01[NT], v10[NT], v11[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = wbase + half + 2 * t;
        const unsigned o = (unsigned)min(j, wlast) * 32u;
        v00[t] = p00[o]; v01[t] = p01[o]; v10[t] = p10[o]; v11[t] = p11[o];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = wbase + half + 2 * t;
        if (j <= wlast) dst[j - wbase][i] = ld0 * (lh0 * v00[t] + lh1 * v01[t]) + ld1 * (lh0 * v10[t] + lh1 * v11[t]);
      }
    };
#ifndef PW_X_FPN_NOSTAGE     // (timing-only ablation: the kernel without its low-resolution staging)
    // source-column windows of the two segments at both levels
    wb2[0] = (int)(sw2 * (float)ow0); wb4[0] = (int)(sw4 * (float)ow0);
    wb2[1] = 0; wb4[1] = 0;
    stage(std::integral_constant<int, 10>{}, f.y16, f.D2, f.H2, f.W2, sd2, sh2, cols[wave][0][0], wb2[0], min((int)(sw2 * (float)(ow0 + n0 - 1)) + 1, f.W2 - 1));
    stage(std::integral_constant<int, 6>{}, f.y32, f.D4, f.H4, f.W4, sd4, sh4, cols[wave][0][1], wb4[0], min((int)(sw4 * (float)(ow0 + n0 - 1)) + 1, f.W4 - 1));
    if (n0 < 32 && m0 + n0 < n_vox) {                  // the tile runs on into the next line (wave-uniform)
      if (++oh == a.H) { oh = 0; if (++od == a.D) { od = 0; ++b; } }
      stage(std::integral_constant<int, 10>{}, f.y16, f.D2, f.H2, f.W2, sd2, sh2, cols[wave][1][0], 0, min((int)(sw2 * (float)(31 - n0)) + 1, f.W2 - 1));
      stage(std::integral_constant<int, 6>{}, f.y32, f.D4, f.H4, f.W4, sd4, sh4, cols[wave][1][1], 0, min((int)(sw4 * (float)(31 - n0)) + 1, f.W4 - 1));
    }
#endif
  }
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, bq[2 * ks]), __builtin_bit_cast(fh8, aq[2 * ks]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, bq[2 * ks + 1]), __builtin_bit_cast(fh8, aq[2 * ks]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fh8, bq[2 * ks]), __builtin_bit_cast(fh8, aq[2 * ks + 1]), acc, 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[4] = {aq[q].x, aq[q].y, aq[q].z, aq[q].w};
      const float bv[4] = {bq[q].x, bq[q].y, bq[q].z, bq[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], av[e], acc, 0, 0, 0);
    }
    }
  }
  // The product is TRANSPOSED, D[cout][voxel] (weights as the A operand, like the conv kernels): lane (i, half) ends with voxel
  // m0 + i and the four channel quads 8 q + 4 half + 0..3 -- so the split-fp16 output leaves as 8-byte pieces (it was two 2-byte
  // stores per element), the w-interpolation terms are computed once per lane instead of once per accumulator row, and the staged
  // low-resolution columns are read as float4.
  const float xmul = rng_pow2(e_x), omul = rng_pow2(-e_out);
  float4 sc4[4], bi4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = 8 * q + 4 * half;
    sc4[q] = a.scale ? *reinterpret_cast<const float4*>(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    bi4[q] = a.bias ? *reinterpret_cast<const float4*>(a.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned amax = 0u;
  // finish channel quad q of this lane's voxel: + interpolated low-resolution terms, scale / bias, ReLU, store
  auto finish = [&](int q, long long vox, const float4& up) {
    const int c0 = 8 * q + 4 * half;
    float v[4] = {acc[4 * q] * xmul + up.x, acc[4 * q + 1] * xmul + up.y, acc[4 * q + 2] * xmul + up.z, acc[4 * q + 3] * xmul + up.w};
    const float s[4] = {sc4[q].x, sc4[q].y, sc4[q].z, sc4[q].w}, bb[4] = {bi4[q].x, bi4[q].y, bi4[q].z, bi4[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (v[e] * s[e] + bb[e]) * omul;
      if (a.relu0) v[e] = fmaxf(v[e], 0.f);
      amax = max(amax, rng_absbits(v[e]));
    }
    float* row = a.y0 + (size_t)vox * 32;
    if (a.fmt_y0) {
      u2 hi, lo;
      h2_split4(v, hi, lo);
      *reinterpret_cast<u2*>(reinterpret_cast<char*>(row) + h2_group_off(c0, 0)) = hi;
      *reinterpret_cast<u2*>(reinterpret_cast<char*>(row) + h2_group_off(c0, 1)) = lo;
    } else {
      *reinterpret_cast<float4*>(row + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  if constexpr (WIDE) {
    __builtin_amdgcn_wave_barrier();                    // a wave's LDS accesses execute in program order: no s_barrier
    const int line = i >= n0;
    const int ow = line ? i - n0 : ow0 + i;
    const float fw2 = sw2 * (float)ow, fw4 = sw4 * (float)ow;
    const int w20 = (int)fw2, w40 = (int)fw4;
    const int w21 = w20 + (w20 < f.W2 - 1), w41 = w40 + (w40 < f.W4 - 1);
    const float l21 = fw2 - (float)w20, l41 = fw4 - (float)w40;
    const float (&c2)[20][32] = cols[wave][line][0];
    const float (&c4)[20][32] = cols[wave][line][1];
    const int o2 = line ? 0 : wb2[0], o4 = line ? 0 : wb4[0];
    if (m0 + i < n_vox) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 8 * q + 4 * half;
        const float4 a0 = *reinterpret_cast<const float4*>(&c2[w20 - o2][c0]), a1 = *reinterpret_cast<const float4*>(&c2[w21 - o2][c0]);
        const float4 b0 = *reinterpret_cast<const float4*>(&c4[w40 - o4][c0]), b1 = *reinterpret_cast<const float4*>(&c4[w41 - o4][c0]);
        const float k20 = 1.f - l21, k40 = 1.f - l41;
        float4 up;
        up.x = (k20 * a0.x + l21 * a1.x) + (k40 * b0.x + l41 * b1.x);
        up.y = (k20 * a0.y + l21 * a1.y) + (k40 * b0.y + l41 * b1.y);
        up.z = (k20 * a0.z + l21 * a1.z) + (k40 * b0.z + l41 * b1.z);
        up.w = (k20 * a0.w + l21 * a1.w) + (k40 * b0.w + l41 * b1.w);
        finish(q, m0 + i, up);
      }
    }
  } else {
    // W < 32 (tiny grids): eight corner loads per channel and level
    const long long vox = m0 + i;
    if (vox < n_vox) {
      // 32-bit index math (host guarantees n_vox < 2^31): 64-bit div/mod dominated this kernel
      const unsigned uv = (unsigned)vox;
      const unsigned t1 = uv / (unsigned)a.W;
      const int ow = (int)(uv - t1 * (unsigned)a.W);
      const unsigned t2 = t1 / (unsigned)a.H;
      const int oh = (int)(t1 - t2 * (unsigned)a.H);
      const int b = (int)(t2 / (unsigned)a.D);
      const int od = (int)(t2 - (unsigned)b * (unsigned)a.D);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 8 * q + 4 * half;
        float u[4];
#pragma unroll 1
        for (int e = 0; e < 4; ++e)
          u[e] = trilerp_ac(f.y16, b, f.D2, f.H2, f.W2, sd2, sh2, sw2, od, oh, ow, c0 + e) +
                 trilerp_ac(f.y32, b, f.D4, f.H4, f.W4, sd4, sh4, sw4, od, oh, ow, c0 + e);
        finish(q, vox, make_float4(u[0], u[1], u[2], u[3]));
      }
    }
  }
  if (a.fmt_y0) rng_note(a.y0_rng, amax, e_out);
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
  if (W >= 32) hipLaunchKernelGGL(k_fpn3d_fuse<true>, dim3((unsigned)pw_cdiv(n, 128)), dim3(256), 0, pw_stream(stream), a, f, n);
  else hipLaunchKernelGGL(k_fpn3d_fuse<false>, dim3((unsigned)pw_cdiv(n, 128)), dim3(256), 0, pw_stream(stream), a, f, n);
  pw_note_kernel(W >= 32 ? "k_fpn3d_fuse<true>" : "k_fpn3d_fuse<false>");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

