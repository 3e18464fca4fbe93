// "h2": the split-fp16 storage format of the voxel encoder's activations, and the fp16 matrix-core helpers built on it.
//
// Why.  The reference computes in fp32.  gfx950 has no TF32/xf32 and its exact-fp32 MFMA runs at 1/16 of the fp16
// rate (157 vs 2500 TFLOP/s dense).  An fp32 value splits EXACTLY into x = hi + lo + r with hi = fp16(x),
// lo = fp16(x - hi), |r| <= 2^-22 |x| (for |x| >= 2^-3; below that lo is subnormal and |r| <= 2^-25 absolute), and
//     x * w  =  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  (+ lo_x*lo_w ~ 2^-22, dropped)
// is three v_mfma_f32_32x32x16_f16 with fp32 accumulation.  tools/probes/mfma_f16_split.hip (profiles/r02_hw_probes.md)
// shows on the hardware: fp16 subnormal operands are preserved, the 16 products of one instruction are summed exactly
// before the single rounding into the accumulator, and a K = 1728 dot product by three split MFMAs is as accurate as
// an fp32 fmaf chain (5.8e-6 vs 5.6e-6 max error against fp64; plain fp16 operands: 2.4e-3).  3 instructions at 16x
// the rate = 5.3x the throughput of the fp32 MFMA at fp32-level accuracy.
//
// Format.  A (.., C) channels-last activation tensor, C % 32 == 0, keeps its shape and its 4 bytes per element: every
// 32-channel chunk of a voxel is 128 bytes = 8 slots of 16 bytes,
//     slot(half, ks, p) = 4*half + 2*ks + p     holds plane p (0 = hi, 1 = lo) of channels 16*ks + 8*half + 0..7
// so that an MFMA lane of lane-half `half` finds its 8-channel fragments of both k-steps and both planes in the four
// consecutive slots 4*half .. 4*half+3: exactly the 64 contiguous bytes the fp32 kernels read per lane, so the halo
// tile layout, its XOR swizzle (conflict-free ds_read_b128) and the `buffer_load ... lds` staging of
// pw_conv3d_common.h apply unchanged.  Values saturate at +-65504 (fp16 range).
#ifndef PW_H2_H_
#define PW_H2_H_
#include "pw_conv3d_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr float H2_MAX = 65504.f;

// byte offset inside a 128-byte chunk of the 8-byte group holding plane p of channels c .. c+3 (c % 4 == 0)
__host__ __device__ __forceinline__ constexpr int h2_group_off(int c, int p) {
  return (4 * ((c >> 3) & 1) + 2 * (c >> 4) + p) * 16 + 2 * (c & 7);
}

__device__ __forceinline__ void h2_split4(const float (&v)[4], u2& hi, u2& lo) {
  h4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = __builtin_amdgcn_fmed3f(v[e], -H2_MAX, H2_MAX);
    h[e] = (_Float16)x;
    l[e] = (_Float16)(x - (float)h[e]);
  }
  hi = __builtin_bit_cast(u2, h);
  lo = __builtin_bit_cast(u2, l);
}

__device__ __forceinline__ void h2_join4(u2 hi, u2 lo, float (&v)[4]) {
  const h4 h = __builtin_bit_cast(h4, hi), l = __builtin_bit_cast(h4, lo);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)h[e] + (float)l[e];
}

__device__ __forceinline__ u2 buf_load2(rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  u2 o;
  o[0] = v[0]; o[1] = v[1];
  return o;
}
__device__ __forceinline__ void buf_store2(rsrc_t r, unsigned voff, unsigned soff, u2 v) {
  typedef unsigned bu2 __attribute__((ext_vector_type(2)));
  bu2 t;
  t[0] = v[0]; t[1] = v[1];
  __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store4(rsrc_t r, unsigned voff, unsigned soff, const float (&v)[4]) {
  typedef unsigned bu4 __attribute__((ext_vector_type(4)));
  bu4 t;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(v[e]);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 0);
  // gfx950 hazard (found as wrong channels c & 7 in {2,3} of one epilogue variant): a 128-bit buffer store reads its data VGPRs over several cycles;
  // hipcc (ROCm 7.2) scheduled a v_pk_add_f32 that overwrites two of them ONE instruction after the store and lanes 12-15 of
  // every 16 stored the new value of dword 1.  Keeping the four registers live across two wait states closes the window.
  asm volatile("s_nop 1" ::"v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}

#endif  // PW_H2_H_
