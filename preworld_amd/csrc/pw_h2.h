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
// pw_conv3d_common.h apply unchanged.
//
// Range.  fp16 has 5 exponent bits: hi overflows above 65504 and lo goes subnormal (absolute resolution 2^-25) once |x| < 2^-3.
// The reference is fp32 with no such domain (backbones/resnet.py:88-123 is plain Conv3d), so every h2 tensor carries a
// per-tensor power-of-two exponent e in a RANGE SLOT (two int32 in device memory, see RngSlot below):
//     value = (hi + lo) * 2^e
// Producers divide by 2^e before the split (folded into their epilogue scale / bias: powers of two, exact) and record the
// largest |value| they wrote; consumers fold 2^e into THEIR epilogue scale exactly like the weights' pre-scale.  The host
// (preworld_amd.ops.RangeCtx) chooses e so that the tensor's largest magnitude lands in [2^12, 2^13) of the stored units --
// 22-bit significands for everything within 2^-15 of the maximum, an absolute floor of 2^-38 of it below -- and re-runs a
// sample whose recorded maximum left [2^6, 65504] under the exponents it was run with.  Nothing is clamped: a value beyond
// +-65504 stored units becomes Inf (NaN stays NaN), is recorded as such in the slot, and propagates like it would in fp32.
#ifndef PW_H2_H_
#define PW_H2_H_
#include "pw_conv3d_common.h"

// Packed split-fp16 weight tile (ops.pack_conv_weight_h2): 4096 B per (chunk, tap, 32-column tile) = four 1024-byte PIECES
// q = 2 ks + p (k-step, hi / lo plane), a piece = 64 lanes x 16 B in lane order.  One load instruction of a wave therefore reads
// 1 KB of consecutive bytes = 8 cache lines.  (Rounds 1-2 kept a lane's four pieces together, 64 B per lane: every instruction
// then touched 32 lines for the same 1 KB, and the vector L1 looks up one line per cycle -- 0.82 lookups per cycle over the whole
// 64->64 kernel, 62 % TA busy, a quarter of the wave cycles in s_waitcnt: profiles/r03_conv_h2_pmc.md.)
constexpr unsigned H2W_PIECE = WPIECE;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr float H2_MAX = 65504.f;

// byte offset inside a 128-byte chunk of the 8-byte group holding plane p of channels c .. c+3 (c % 4 == 0)
__host__ __device__ __forceinline__ constexpr int h2_group_off(int c, int p) {
  return (4 * ((c >> 3) & 1) + 2 * (c >> 4) + p) * 16 + 2 * (c & 7);
}

// x -> (hi, lo).  No saturation: |x| >= 65520 gives hi = +-Inf (lo is then clamped to a finite value so that hi + lo stays Inf
// instead of Inf - Inf), NaN gives NaN.
__device__ __forceinline__ void h2_split1(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  l = (_Float16)__builtin_amdgcn_fmed3f(x - (float)h, -H2_MAX, H2_MAX);
}

// 8 values at once, written pair-wise: v_cvt_pk_f16_f32 for the two hi halves, the residual x - hi as ONE v_fma_mix_f32 per
// value (hi * -1 + x with hi read as the low / high half of the packed register: exact, like the convert + subtract pair it
// replaces -- hipcc does not form the mixed-precision FMA by itself), clamp, v_cvt_pk_f16_f32 for the two lo halves
typedef float h2_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_f2 h2_residual2(h2_h2 h, h2_f2 x) {
#ifdef PW_SPLIT_NOMIX                      // A/B: the convert + subtract form
  return x - __builtin_convertvector(h, h2_f2);
#endif
  const unsigned hp = __builtin_bit_cast(unsigned, h);
  h2_f2 d;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d[0]) : "v"(hp), "v"(x[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d[1]) : "v"(hp), "v"(x[1]));
  return d;
}
__device__ __forceinline__ void h2_split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const h2_f2 x = {v[2 * p], v[2 * p + 1]};
    const h2_h2 h = __builtin_convertvector(x, h2_h2);
    h2_f2 d = h2_residual2(h, x);
    d[0] = __builtin_amdgcn_fmed3f(d[0], -H2_MAX, H2_MAX);
    d[1] = __builtin_amdgcn_fmed3f(d[1], -H2_MAX, H2_MAX);
    const h2_h2 l = __builtin_convertvector(d, h2_h2);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

__device__ __forceinline__ void h2_split4(const float (&v)[4], u2& hi, u2& lo) {
  h4 h, l;
#pragma unroll
  for (int p = 0; p < 2; ++p) {                      // pair-wise, like h2_split8
    const h2_f2 x = {v[2 * p], v[2 * p + 1]};
    const h2_h2 a = __builtin_convertvector(x, h2_h2);
    h2_f2 d = h2_residual2(a, x);
    d[0] = __builtin_amdgcn_fmed3f(d[0], -H2_MAX, H2_MAX);
    d[1] = __builtin_amdgcn_fmed3f(d[1], -H2_MAX, H2_MAX);
    const h2_h2 b = __builtin_convertvector(d, h2_h2);
    h[2 * p] = a[0]; h[2 * p + 1] = a[1];
    l[2 * p] = b[0]; l[2 * p + 1] = b[1];
  }
  hi = __builtin_bit_cast(u2, h);
  lo = __builtin_bit_cast(u2, l);
}

// ---- range slots (see "Range" above).  A slot is PW_RNG_ROW int32: rng[0] = exponent e; rng[1] = bit pattern of the largest
// |value| in TRUE units (non-negative floats order like unsigned integers; a NaN pattern is above every number), valid after
// pw_rng_fold; rng[PW_RNG_SCRATCH ..] = PW_RNG_WORDS partial maxima the waves of the producing kernels raise.  Why partials:
// device-scope atomics on ONE address serialise at ~100 ns each -- 8 192 waves raising one word cost the pooling
// kernel 150 us; spread over 1 024 words, without a returned value, they cost nothing measurable.  pw_rng_fold (one small
// launch per pass) folds them into rng[1] and clears them.  A null slot = exponent 0, nothing recorded.
#ifdef PW_RNG_EXP0        // experiment builds only (tools/build_variant.py): how much do the exponent loads cost?
__device__ __forceinline__ int rng_exp(const int*) { return 0; }
#else
__device__ __forceinline__ int rng_exp(const int* r) { return r ? __builtin_amdgcn_readfirstlane(r[0]) : 0; }
#endif
__device__ __forceinline__ float rng_pow2(int e) { return __builtin_ldexpf(1.0f, e); }
__device__ __forceinline__ unsigned rng_absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// maximum over the 64 lanes, as a wave-uniform value: 4 DPP steps inside the rows of 16 lanes (xor 1, xor 2, half-row mirror, row
// mirror), then the four row results through v_readlane + s_max -- no LDS round trips (six ds_bpermute cost ~1 k cycles at
// the tail of every kernel)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));     // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));     // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
// every lane of a wave calls this once, at the end of the kernel (all 64 lanes active): bits = |largest STORED value| of the lane,
// e = the exponent it was stored under
__device__ __forceinline__ void rng_note(int* r, unsigned bits, int e) {
#ifdef PW_RNG_NOAMAX      // experiment builds only: how much does recording the maxima cost?
  return;
#endif
  if (!r) return;
  bits = wave_umax(bits);
  if ((threadIdx.x & 63) == 0 && bits) {
    const unsigned m = rng_absbits(__builtin_ldexpf(__uint_as_float(bits), e));
    const unsigned w = ((unsigned)blockIdx.x * 16u + (threadIdx.x >> 6)) & (unsigned)(PW_RNG_WORDS - 1);
    (void)__hip_atomic_fetch_max(reinterpret_cast<unsigned*>(r) + PW_RNG_SCRATCH + w, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the exponent the host's calibration picks for a tensor whose largest magnitude has this bit pattern: it puts that magnitude
// in [2^12, 2^13) of the stored units (preworld_amd.ops.RangeCtx.ideal_exp computes the same number); 0 for an all-zero or
// non-finite tensor
__host__ __device__ __forceinline__ int rng_ideal_exp(unsigned bits) {
  if (bits == 0u || bits >= 0x7f800000u) return 0;
  const int k = (int)(bits >> 23) - 127 - 12;
  return k < -100 ? -100 : (k > 100 ? 100 : k);
}
// how a conv-shaped kernel folds the exponents of its operands into its epilogue constants
struct RngScale {
  int e0, e1;              // exponents y0 / y1 are stored under (0 for fp32 destinations)
  float s0, s1;            // multiply scale[n] of y0 / y1 columns:  2^(e_x - e_y)
  float b0, b1;            // multiply bias[n]:                      2^(-e_y)
  float res;               // residual as stored -> y0 as stored:    2^(e_res - e_y0)
};
__device__ __forceinline__ RngScale rng_scales(const ConvArgs& a) {
  RngScale s;
  const int ex = rng_exp(a.x_rng);
  const int er = a.fmt_res ? rng_exp(a.res_rng) : 0;
  s.e0 = a.fmt_y0 ? rng_exp(a.y0_rng) : 0;
  s.e1 = a.fmt_y1 ? rng_exp(a.y1_rng) : 0;
  s.s0 = rng_pow2(ex - s.e0); s.s1 = rng_pow2(ex - s.e1);
  s.b0 = rng_pow2(-s.e0); s.b1 = rng_pow2(-s.e1);
  s.res = rng_pow2(er - s.e0);
  return s;
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
