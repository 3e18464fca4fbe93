// Data gradient of the 3x3x3 stride-2 pad-1 convolutions (the down-sampling conv1 / downsample of a BasicBlock3D,
// mmdet3d/models/backbones/resnet.py:88-123, 149-163, under autograd) on the fp16 matrix cores with split-fp16 operands.
//
//   y[o] = sum_k W[k] x[2 o + k - 1]      =>      dX[i] = sum_{(o, k): 2 o + k - 1 = i} W[k]^T dY[o]
//
// Along one axis an even i = 2 j is reached by k = 1 from o = j only; an odd i = 2 j + 1 by k = 0 from o = j + 1 and by k = 2
// from o = j.  The 8 parity classes (pd, ph, pw) of the fine grid therefore are 8 dense little convolutions over dY with
// 1, 2, 2, 2, 4, 4, 4, 8 taps -- 27 tap products per 8 fine voxels.  (Round 3 inserted zeros between the voxels of dY and ran the
// stride-1 kernel over the fine grid: 27 tap products per fine voxel, 7/8 of them with zeros, plus a 164 MB zero fill.)
//
// One wave = 32 consecutive coarse voxels j (linear index) of ONE class; it gathers its A fragments straight from dY (h2
// storage, L2-resident: dY is 1/8 of dX), multiplies by the packed W^T tiles and stores its 32 fine rows dX[2 j + p] once.
// Transposed product (weights as the A operand), so a lane ends up with 4 consecutive channels of one voxel = one 16-byte store.
#include "pw_h2.h"

namespace {
constexpr unsigned DG_OOB = 0xfffffff0u;
typedef _Float16 dgh8 __attribute__((ext_vector_type(8)));

struct DgArgs {
  const float* dy;        // (B, Do, Ho, Wo, Cout) h2 storage
  const float* wpk;       // [Cout/32][27][Cin/32][4 pieces][64 lanes][4]: split-fp16 S[n] * W[c][n][tap] (n = input channel of the conv)
  const float* inv;       // [Cin] 1 / S[n]
  float* dx;              // (B, D, H, W, Cin) fp32
  const int* dy_rng;
  int B, D, H, W, Do, Ho, Wo, Cin, Cout;
};

__host__ __device__ constexpr int dg_k(int P, int t) { return P ? (t ? 2 : 0) : 1; }      // kernel tap along one axis
__host__ __device__ constexpr int dg_off(int P, int t) { return P ? (t ? 0 : 1) : 0; }    // coarse offset it reads

struct DgCtx {
  const float* dy;
  rsrc_t wr;
  unsigned voff, lane_off, wtile;      // byte offset of the lane's coarse voxel (+ lane half); lane * 16; first weight tile of this N-group
  unsigned ntiles;
  bool ok, okd, okh, okw;              // lane's voxel exists; its +1 neighbour along d / h / w exists
  int Ho, Wo, Cout;
};

template <int NT, int PD, int PH, int PW_, int TAP>
__device__ __forceinline__ void dg_load(const DgCtx& c, int ch, float4 (&aq)[4], float4 (&bq)[NT][4]) {
  constexpr int NH = 1 + PH, NW = 1 + PW_;
  constexpr int td = TAP / (NH * NW), th = (TAP / NW) % NH, tw = TAP % NW;
  constexpr int od = dg_off(PD, td), oh = dg_off(PH, th), ow = dg_off(PW_, tw);
  constexpr int ktap = (dg_k(PD, td) * 3 + dg_k(PH, th)) * 3 + dg_k(PW_, tw);
  const long long delta = ((long long)(od * c.Ho + oh) * c.Wo + ow) * c.Cout + ch * KC;
  const rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.dy + delta), 0, 0xffffffe0u, 0x00020000);
  const bool ok = c.ok && (!od || c.okd) && (!oh || c.okh) && (!ow || c.okw);
  const unsigned v = ok ? c.voff : DG_OOB;
#pragma unroll
  for (int q = 0; q < 4; ++q) aq[q] = buf_load4(xr, v, (unsigned)(q * 16));
  load_b<NT>(c.wr, c.wtile + (unsigned)(ch * 27 + ktap) * c.ntiles * 4096u, c.lane_off, bq);
}

template <int NT>
__device__ __forceinline__ void dg_mfma(const float4 (&aq)[4], const float4 (&bq)[NT][4], f32x16 (&acc)[NT]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int prod = 0; prod < 3; ++prod) {            // hi.hi, hi_x.lo_w, lo_x.hi_w
      const int pw = prod == 1 ? 1 : 0, px = prod == 2 ? 1 : 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dgh8, bq[nt][2 * ks + pw]),
                                                         __builtin_bit_cast(dgh8, aq[2 * ks + px]), acc[nt], 0, 0, 0);
    }
}

// tap TAP computes from (ac, bc) while (an, bn) receive tap TAP + 1 (or tap 0 of the next chunk)
template <int NT, int PD, int PH, int PW_, int TAP>
__device__ __forceinline__ void dg_step(const DgCtx& c, int ch, bool more, float4 (&ac)[4], float4 (&bc)[NT][4], float4 (&an)[4],
                                        float4 (&bn)[NT][4], f32x16 (&acc)[NT]) {
  constexpr int TAPS = (1 + PD) * (1 + PH) * (1 + PW_);
  if constexpr (TAP + 1 < TAPS) {
    dg_load<NT, PD, PH, PW_, TAP + 1>(c, ch, an, bn);
  } else {
    if (more) dg_load<NT, PD, PH, PW_, 0>(c, ch + 1, an, bn);
  }
  __builtin_amdgcn_sched_barrier(0);
  dg_mfma<NT>(ac, bc, acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP + 1 < TAPS) dg_step<NT, PD, PH, PW_, TAP + 1>(c, ch, more, an, bn, ac, bc, acc);
}

template <int NT, int PD, int PH, int PW_>
__device__ __forceinline__ void dg_class(const DgCtx& c, int nchunk, f32x16 (&acc)[NT]) {
  constexpr int TAPS = (1 + PD) * (1 + PH) * (1 + PW_);
  float4 a0[4], a1[4], b0[NT][4], b1[NT][4];
  dg_load<NT, PD, PH, PW_, 0>(c, 0, a0, b0);
  for (int ch = 0; ch < nchunk; ++ch) {
    dg_step<NT, PD, PH, PW_, 0>(c, ch, ch + 1 < nchunk, a0, b0, a1, b1, acc);
    if constexpr (TAPS & 1) {                          // one tap: the next chunk's fragments arrived in (a1, b1)
#pragma unroll
      for (int q = 0; q < 4; ++q) a0[q] = a1[q];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) b0[nt][q] = b1[nt][q];
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(256) k_conv3d_dgrad_s2_h2(DgArgs a, long long n_coarse) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int half = lane >> 5, i = lane & 31;
  const int cls = 7 - (int)blockIdx.z;                 // the 8-tap class first: the tail of the launch is made of the short ones
  const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
  const int ng = blockIdx.y;
  const long long m0 = ((long long)blockIdx.x * 4 + wave) * 32;
  if (m0 >= n_coarse) return;
  long long m = m0 + i;
  const bool mvalid = m < n_coarse;
  if (!mvalid) m = n_coarse - 1;
  const int jw = (int)(m % a.Wo); long long t = m / a.Wo;
  const int jh = (int)(t % a.Ho); t /= a.Ho;
  const int jd = (int)(t % a.Do);
  const int b = (int)(t / a.Do);

  DgCtx c;
  c.dy = a.dy; c.Ho = a.Ho; c.Wo = a.Wo; c.Cout = a.Cout;
  c.voff = (unsigned)(((size_t)m * a.Cout + half * 16) * 4);
  c.ok = mvalid; c.okd = jd + 1 < a.Do; c.okh = jh + 1 < a.Ho; c.okw = jw + 1 < a.Wo;
  const int nchunk = a.Cout / KC;
  c.ntiles = (unsigned)(a.Cin >> 5);
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 27 * c.ntiles * 4096));
  c.lane_off = (unsigned)lane * 16u;
  c.wtile = (unsigned)(ng * NT) * 4096u;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  switch (cls) {                                       // wave-uniform
    case 0: dg_class<NT, 0, 0, 0>(c, nchunk, acc); break;
    case 1: dg_class<NT, 0, 0, 1>(c, nchunk, acc); break;
    case 2: dg_class<NT, 0, 1, 0>(c, nchunk, acc); break;
    case 3: dg_class<NT, 0, 1, 1>(c, nchunk, acc); break;
    case 4: dg_class<NT, 1, 0, 0>(c, nchunk, acc); break;
    case 5: dg_class<NT, 1, 0, 1>(c, nchunk, acc); break;
    case 6: dg_class<NT, 1, 1, 0>(c, nchunk, acc); break;
    default: dg_class<NT, 1, 1, 1>(c, nchunk, acc); break;
  }

  // accumulators: column = coarse voxel i of the wave, register r = channel (r & 3) + 8 (r >> 2) + 4 half of the 32-column tile
  const int id = 2 * jd + pd, ih = 2 * jh + ph, iw = 2 * jw + pw;
  if (!mvalid || id >= a.D || ih >= a.H || iw >= a.W) return;
  const float sx = rng_pow2(rng_exp(a.dy_rng));
  float* row = a.dx + ((((size_t)b * a.D + id) * a.H + ih) * a.W + iw) * a.Cin;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = (ng * NT + nt) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = n0 + 8 * q + 4 * half;
      const float4 s = *reinterpret_cast<const float4*>(a.inv + ch);
      float4 v;
      v.x = acc[nt][4 * q + 0] * (s.x * sx); v.y = acc[nt][4 * q + 1] * (s.y * sx);
      v.z = acc[nt][4 * q + 2] * (s.z * sx); v.w = acc[nt][4 * q + 3] * (s.w * sx);
      *reinterpret_cast<float4*>(row + ch) = v;
    }
  }
}

// S[n] = 2^k with the largest |W[.][n][.]| * S in [512, 1024) (the rule of preworld_amd.ops.pack_conv_weight_h2); inv[n] = 1 / S[n]
__global__ void __launch_bounds__(256) k_dgrad_w_scale(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ S,
                                                       float* __restrict__ inv) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  float m = 0.f;
  for (int idx = threadIdx.x; idx < Cout * 27; idx += 256) {
    const int co = idx / 27, tap = idx - co * 27;
    m = fmaxf(m, fabsf(w[((size_t)co * Cin + n) * 27 + tap]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double amax = fmax((double)red[0], 1e-30);
    int e;
    (void)frexp(1023.0 / amax, &e);                    // 1023 / amax = f * 2^e, f in [0.5, 1): floor(log2) = e - 1
    S[n] = (float)ldexp(1.0, e - 1);
    inv[n] = (float)ldexp(1.0, 1 - e);
  }
}

// one thread = the 8 halves of both planes of (chunk ch, tap, tile nt, k-step ks, lane half h, column j)
__global__ void __launch_bounds__(256) k_dgrad_w_pack(const float* __restrict__ w, const float* __restrict__ S, int Cout, int Cin,
                                                      float* __restrict__ out) {
  const int ntiles = Cin >> 5;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)(Cout >> 5) * 27 * ntiles * 128;
  if (idx >= total) return;
  size_t t = idx;
  const int j = (int)(t & 31); t >>= 5;
  const int h = (int)(t & 1); t >>= 1;
  const int ks = (int)(t & 1); t >>= 1;
  const int nt = (int)(t % ntiles); t /= ntiles;
  const int tap = (int)(t % 27);
  const int ch = (int)(t / 27);
  const int n = nt * 32 + j;
  const float s = S[n];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w[((size_t)(ch * 32 + 16 * ks + 8 * h + e) * Cin + n) * 27 + tap] * s;
  h8 hi, lo;
  h2_split8(v, hi, lo);
  char* tile = reinterpret_cast<char*>(out) + (((size_t)ch * 27 + tap) * ntiles + nt) * 4096 + (size_t)(h * 32 + j) * 16;
  *reinterpret_cast<h8*>(tile + (2 * ks + 0) * H2W_PIECE) = hi;
  *reinterpret_cast<h8*>(tile + (2 * ks + 1) * H2W_PIECE) = lo;
}
}  // namespace

PW_API size_t pw_conv3d_dgrad_s2_h2_workspace_bytes(int Cin, int Cout) {
  return (size_t)Cout * 27 * Cin * 4 + (size_t)2 * Cin * 4 + 256;
}

PW_API int pw_conv3d_dgrad_s2_h2(const float* dy, const int32_t* dy_rng, const float* w, float* dx, void* workspace,
                                 size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  PW_CHECK_ARG(dy && w && dx && workspace, "pw_conv3d_dgrad_s2_h2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0,
               "pw_conv3d_dgrad_s2_h2: Cin and Cout must be multiples of 32");
  PW_CHECK_ARG(workspace_bytes >= pw_conv3d_dgrad_s2_h2_workspace_bytes(Cin, Cout), "pw_conv3d_dgrad_s2_h2: workspace too small");
  DgArgs a = {};
  a.dy = dy; a.dy_rng = dy_rng; a.dx = dx;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  const long long n_coarse = (long long)B * a.Do * a.Ho * a.Wo;
  PW_CHECK_ARG((double)n_coarse * Cout * 4 < 4.0e9 && (double)Cout * 27 * Cin * 4 < 4.0e9, "pw_conv3d_dgrad_s2_h2: dY / W beyond 32-bit offsets");
  float* wpk = static_cast<float*>(workspace);
  float* S = wpk + (size_t)Cout * 27 * Cin;
  float* inv = S + Cin;
  a.wpk = wpk; a.inv = inv;
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_dgrad_w_scale, dim3((unsigned)Cin), dim3(256), 0, st, w, Cout, Cin, S, inv);
  const size_t total = (size_t)(Cout >> 5) * 27 * (Cin >> 5) * 128;
  hipLaunchKernelGGL(k_dgrad_w_pack, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, st, w, S, Cout, Cin, wpk);
  const int ntiles = Cin >> 5;
  const int NT = ntiles % 2 == 0 ? 2 : 1;
  dim3 grid((unsigned)pw_cdiv(n_coarse, 128), (unsigned)(ntiles / NT), 8u);
  if (NT == 2) hipLaunchKernelGGL((k_conv3d_dgrad_s2_h2<2>), grid, dim3(256), 0, st, a, n_coarse);
  else hipLaunchKernelGGL((k_conv3d_dgrad_s2_h2<1>), grid, dim3(256), 0, st, a, n_coarse);
  pw_note_kernel("k_conv3d_dgrad_s2_h2<%d>", NT);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
