// Device helpers shared by pw_lss.hip (sort-based pooling, reference-ABI ops) and pw_lss_fused.hip (slot-based single-frame lift +
// pooling).  Both files are compiled with -ffp-contract=off: the geometry chain and the pooled sums follow the op order of the test
// oracle, so the two paths and the oracle agree bit for bit.
#pragma once
#include "pw_common.h"
#include "pw_h2.h"

__device__ __forceinline__ void inv3x3(const float* m, float* o) {
  float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  float A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
  float D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
  float G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
  float det = (a * A + b * D) + c * G;
  float r = 1.0f / det;
  o[0] = A * r; o[1] = B * r; o[2] = C * r;
  o[3] = D * r; o[4] = E * r; o[5] = F * r;
  o[6] = G * r; o[7] = H * r; o[8] = I * r;
}

// camera c: inverse(post_rot), sensor2ego[:3,:3] @ inverse(cam2img), sensor2ego[:3,3] (view_transformer.py:141-150; closed-form
// 3x3 inverse in the op order of the oracle's inv3x3_f32)
__device__ __forceinline__ void lss_camera_matrix_one(int c, const float* __restrict__ s2e, const float* __restrict__ K,
                                                      const float* __restrict__ pr, float* __restrict__ ipr,
                                                      float* __restrict__ comb, float* __restrict__ tr) {
  float R[9], Kin[9], Kinv[9], P[9], Pinv[9];
  const float* S = s2e + c * 16;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = S[i * 4 + j];
  for (int i = 0; i < 9; ++i) { Kin[i] = K[c * 9 + i]; P[i] = pr[c * 9 + i]; }
  inv3x3(P, Pinv);
  inv3x3(Kin, Kinv);
  for (int i = 0; i < 9; ++i) ipr[c * 9 + i] = Pinv[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * Kinv[k * 3 + j];
      comb[c * 9 + i * 3 + j] = acc;
    }
  tr[c * 3 + 0] = S[3];
  tr[c * 3 + 1] = S[7];
  tr[c * 3 + 2] = S[11];
}

struct GridParams {
  float lx, ly, lz, ix, iy, iz;
  int gx, gy, gz;
};


__device__ __forceinline__ int count_smaller_lds(const int32_t* ids, int m16, int id) {
  const int4* ids4 = reinterpret_cast<const int4*>(ids);
  int r = 0;
  for (int j = 0; j < m16 / 4; j += 4) {                  // 4 x ds_read_b128 (broadcast) per trip
    const int4 a = ids4[j], b = ids4[j + 1], c = ids4[j + 2], d = ids4[j + 3];
    r += (a.x < id) + (a.y < id) + (a.z < id) + (a.w < id);
    r += (b.x < id) + (b.y < id) + (b.z < id) + (b.w < id);
    r += (c.x < id) + (c.y < id) + (c.z < id) + (c.w < id);
    r += (d.x < id) + (d.y < id) + (d.z < id) + (d.w < id);
  }
  return r;
}


__device__ __forceinline__ void fma4_nc(float4& acc, const float4& f, float d) {
  acc.x = acc.x + f.x * d;
  acc.y = acc.y + f.y * d;
  acc.z = acc.z + f.z * d;
  acc.w = acc.w + f.w * d;
}


// frustum entry (fr0, fr1, fr2) = (x pixel, y pixel, depth) of camera `cam` -> voxel id (or -1): get_lidar_coor + the index half of
// voxel_pooling_prepare_v2 (view_transformer.py:114-153, :226-245), one multiply / add at a time in the oracle's order.  o_out (or null)
// receives the point in ego coordinates.
__device__ __forceinline__ int32_t lss_voxel_of_fr(float fr0, float fr1, float fr2, int cam, int b, const float* __restrict__ ipr,
                                                   const float* __restrict__ ptr, const float* __restrict__ comb,
                                                   const float* __restrict__ trn, const float* __restrict__ bda, const GridParams& gp,
                                                   float* __restrict__ o_out) {
  const float* M = ipr + cam * 9;
  const float* C = comb + cam * 9;
  const float* T = trn + cam * 3;
  const float* PT = ptr + cam * 3;
  const float* A = bda + b * 9;
  float p0 = fr0 - PT[0], p1 = fr1 - PT[1], p2 = fr2 - PT[2];
  float q[3], r[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float acc = 0.f;
    acc += M[k * 3 + 0] * p0;
    acc += M[k * 3 + 1] * p1;
    acc += M[k * 3 + 2] * p2;
    q[k] = acc;
  }
  float u0 = q[0] * q[2], u1 = q[1] * q[2], u2 = q[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float acc = 0.f;
    acc += C[k * 3 + 0] * u0;
    acc += C[k * 3 + 1] * u1;
    acc += C[k * 3 + 2] * u2;
    r[k] = acc + T[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float acc = 0.f;
    acc += A[k * 3 + 0] * r[0];
    acc += A[k * 3 + 1] * r[1];
    acc += A[k * 3 + 2] * r[2];
    o[k] = acc;
  }
  if (o_out) {
    o_out[0] = o[0];
    o_out[1] = o[1];
    o_out[2] = o[2];
  }
  float fx = (o[0] - gp.lx) / gp.ix;
  float fy = (o[1] - gp.ly) / gp.iy;
  float fz = (o[2] - gp.lz) / gp.iz;
  // .long() truncates toward zero (view_transformer.py:228): trunc(f) in [0,g) <=> -1 < f < g
  bool in = fx > -1.f && fx < (float)gp.gx && fy > -1.f && fy < (float)gp.gy && fz > -1.f &&
            fz < (float)gp.gz;
  int32_t v = -1;
  if (in) {
    int x = (int)fx, y = (int)fy, z = (int)fz;
    v = ((b * gp.gz + z) * gp.gy + y) * gp.gx + x;
  }
  return v;
}

// frustum point i -> voxel id (or -1)
__device__ __forceinline__ int32_t lss_voxel_of_point(int64_t i, int N, int64_t DHW, const float* __restrict__ frustum,
                                                      const float* __restrict__ ipr, const float* __restrict__ ptr,
                                                      const float* __restrict__ comb, const float* __restrict__ trn,
                                                      const float* __restrict__ bda, const GridParams& gp,
                                                      float* __restrict__ coor_out) {
  int cam = (int)(i / DHW);
  int64_t p = i - (int64_t)cam * DHW;
  const float* fr = frustum + p * 3;
  return lss_voxel_of_fr(fr[0], fr[1], fr[2], cam, cam / N, ipr, ptr, comb, trn, bda, gp, coor_out ? coor_out + i * 3 : nullptr);
}

// one lane's 4 channels (quad `sub`) of voxel row v: fp32, or split-fp16 (h2, pw_h2.h) for the fp16-matrix-core encoder
template <int LPV>
__device__ __forceinline__ void pool_store(float4* __restrict__ out, int64_t v, int sub, const float4& acc, int out_h2,
                                           float mul, unsigned& amax) {
  if (!out_h2) {
    out[v * LPV + sub] = acc;
  } else {
    // h2: the sums are stored divided by 2^e of the destination's range slot (mul = 2^-e, exact) and their largest magnitude
    // is recorded (pw_h2.h "Range"); bit-pattern maximum, so a NaN among the inputs shows up in the slot
    const float f[4] = {acc.x * mul, acc.y * mul, acc.z * mul, acc.w * mul};
    amax = max(max(amax, rng_absbits(f[0])), max(rng_absbits(f[1]), max(rng_absbits(f[2]), rng_absbits(f[3]))));
    u2 hi, lo;
    h2_split4(f, hi, lo);
    char* row = reinterpret_cast<char*>(out + v * LPV) + (sub >> 3) * 128;
    const int c = (sub & 7) * 4;
    *reinterpret_cast<u2*>(row + h2_group_off(c, 0)) = hi;
    *reinterpret_cast<u2*>(row + h2_group_off(c, 1)) = lo;
  }
}

