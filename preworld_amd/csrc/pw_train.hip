// Training-side kernels of the voxel encoder: what torch autograd runs behind the reference's BasicBlock3D / CustomResNet3D
// (mmdet3d/models/backbones/resnet.py:88-184: Conv3d(bias=False) -> BatchNorm3d (batch statistics) -> ReLU, residual add) when
// the detector's forward_train calls them.  Entry points (include/preworld_hip.h):
//   pw_conv3d_wgrad      dW[co][ci][tap] = sum_vox dY[vox][co] X[vox*s + tap - pad][ci]       (fp32 MFMA, K = voxels)
//   pw_conv3d_dgrad_s2   dX of a 3x3x3 stride-2 pad-1 conv (the stride-1 / 1x1x1 dgrads are forward convs with the weights
//                        flipped / transposed and run on the forward kernels: preworld_amd/train.py)
//   pw_bn_stats / pw_bn_apply / pw_bn_bwd_reduce / pw_bn_bwd_apply   BatchNorm3d in training mode on channels-last rows
// Channels-last fp32 everywhere: x (B, D, H, W, Cin), dy (B, Do, Ho, Wo, Cout).
#include "pw_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------
// wgrad.  One wave accumulates the 32 x 32 blocks (co0.., ci0..) of the KS taps of one (kd, kh) over a slice of output rows (b, od, oh):
// v_mfma_f32_32x32x2_f32 with K = two consecutive output voxels of the row.  Lane (i = l & 31, k = l >> 5) feeds
//   A[i][k] = dY[row, ow + k][co0 + i]          B[k][j] = X[input row of the tap, (ow + k) s + kw - pad][ci0 + j]
// straight from global memory: consecutive lanes read consecutive channels of a voxel row (128-byte segments), the validity of
// the tap's input row is wave-uniform; columns outside the input along w load as zero.  The four waves of a block take interleaved rows of the block's chunk and add their tiles through LDS; a second
// kernel sums the per-chunk partial tiles in a fixed order (deterministic, no atomics) into torch's [Cout][Cin][kd][kh][kw].
// ------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;
  const float* dy;
  float* partial;        // [n_chunks][taps][co_blocks][ci_blocks][32][32]
  int B, D, H, W, Cin, Do, Ho, Wo, Cout;
  int ks, stride, pad, taps;
  int co_blocks, ci_blocks, n_chunks, rows_per_chunk, n_rows;
};

// KS = kernel extent along w handled by ONE block (all kw taps of a (kd, kh) pair): the dY operand is loaded once for the KS taps,
// and at stride 1 the shifted X operands of a lane overlap (index 2 u + kw): 4 + 9 loads per 12 MFMAs instead of 24.
template <int KS, int STRIDE>
__global__ void __launch_bounds__(256) k_conv3d_wgrad(WgradArgs a) {
  __shared__ float red[3][KS][64 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, k = lane >> 5;
  int bid = blockIdx.x;
  const int cib = bid % a.ci_blocks; bid /= a.ci_blocks;
  const int cob = bid % a.co_blocks; bid /= a.co_blocks;
  const int kh = bid % KS, kd = bid / KS;
  const int chunk = blockIdx.y;
  const int co = cob * 32 + i, ci = cib * 32 + i;
  const bool co_ok = co < a.Cout, ci_ok = ci < a.Cin;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int r0 = chunk * a.rows_per_chunk, r1 = min(a.n_rows, r0 + a.rows_per_chunk);
  for (int row = r0 + wave; row < r1; row += 4) {
    const int oh = row % a.Ho, t = row / a.Ho;
    const int od = t % a.Do, b = t / a.Do;
    const int id = od * STRIDE + kd - a.pad, ih = oh * STRIDE + kh - a.pad;
    if ((unsigned)id >= (unsigned)a.D || (unsigned)ih >= (unsigned)a.H) continue;     // wave-uniform
    const float* dyr = a.dy + ((size_t)row * a.Wo) * a.Cout + co;
    const float* xr = a.x + ((((size_t)b * a.D + id) * a.H + ih) * a.W) * a.Cin + ci;
    for (int ow = 0; ow < a.Wo; ow += 8) {               // 4 k-pairs of output voxels per trip: 4 KS MFMAs
      float av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = ow + 2 * u + k;
        av[u] = (co_ok && o < a.Wo) ? dyr[(size_t)o * a.Cout] : 0.f;
      }
      if constexpr (STRIDE == 1) {
        // X index of (u, kw) for this lane: ow + 2 u + k + kw - pad = base + (2 u + kw): 2 * 3 + KS distinct values
        float xv[6 + KS];
        const int base = ow + k - a.pad;
#pragma unroll
        for (int q = 0; q < 6 + KS; ++q) {
          const int iw = base + q;
          xv[q] = (ci_ok && (unsigned)iw < (unsigned)a.W) ? xr[(size_t)iw * a.Cin] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], xv[2 * u + kw], acc[kw], 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float xv[KS];
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const int iw = (ow + 2 * u + k) * STRIDE + kw - a.pad;
            xv[kw] = (ci_ok && (unsigned)iw < (unsigned)a.W) ? xr[(size_t)iw * a.Cin] : 0.f;
          }
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], xv[kw], acc[kw], 0, 0, 0);
        }
      }
    }
  }
  // block reduction of the four waves' tiles (fixed order: wave 0 + 1 + 2 + 3)
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][t][r * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const int tap = (kd * KS + kh) * KS + t;
      float* dst = a.partial + ((((size_t)chunk * a.taps + tap) * a.co_blocks + cob) * a.ci_blocks + cib) * 1024;
      // D row (co) = (r & 3) + 8 (r >> 2) + 4 k, column (ci) = i
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + i] =
            ((acc[t][r] + red[0][t][r * 64 + lane]) + red[1][t][r * 64 + lane]) + red[2][t][r * 64 + lane];
    }
  }
}

// sum of the per-chunk partial tiles -> torch's [Cout][Cin][kd][kh][kw].  A block owns 32 consecutive elements of a tile; its 8
// groups of 32 threads each add every 8th chunk (four loads in flight), the 8 sums meet in LDS and are added in group order: a fixed
// order, so the result is deterministic.  (One thread per element walking all chunks -- up to 1 024 strided loads in a row -- took
// 54-94 us per call, 1.1 ms of the training step.)
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, int n_chunks,
                                                      int taps, int co_blocks, int ci_blocks, int Cout, int Cin) {
  __shared__ float part[8][32];
  const size_t per_chunk = (size_t)taps * co_blocks * ci_blocks * 1024;
  const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t idx = (size_t)blockIdx.x * 32 + e;                         // over [tap][cob][cib][32][32]
  const float* src = partial + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = q;
  for (; c + 24 < n_chunks; c += 32) {
    s0 += src[(size_t)c * per_chunk];
    s1 += src[(size_t)(c + 8) * per_chunk];
    s2 += src[(size_t)(c + 16) * per_chunk];
    s3 += src[(size_t)(c + 24) * per_chunk];
  }
  for (; c < n_chunks; c += 8) s0 += src[(size_t)c * per_chunk];
  part[q][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0) {
    float s = part[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][e];
    const int j = (int)(idx & 31), i = (int)((idx >> 5) & 31);
    size_t t = idx >> 10;
    const int cib = (int)(t % ci_blocks); t /= ci_blocks;
    const int cob = (int)(t % co_blocks); t /= co_blocks;
    const int tap = (int)t;
    const int co = cob * 32 + i, ci = cib * 32 + j;
    if (co < Cout && ci < Cin) dw[((size_t)co * Cin + ci) * taps + tap] = s;
  }
}

static inline int wgrad_pad(int ksize) { return ksize == 2 ? 0 : ksize / 2; }   // 2x2x2 stride-2 convs (trajectory branch) are unpadded

// output rows are split into chunks so that (kd, kh) pairs x channel blocks x chunks gives ~3 000 blocks (enough waves to hide the
// operand loads' latency), with at least 8 rows per chunk
static int wgrad_chunks(int n_rows, int ksize, int Cin, int Cout) {
  const int groups = ksize * ksize * ((Cout + 31) / 32) * ((Cin + 31) / 32);
  int n = 3072 / groups;
  if (n > n_rows / 8) n = n_rows / 8;
  return n < 1 ? 1 : (n > 1024 ? 1024 : n);
}

PW_API size_t pw_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int ksize, int stride) {
  const int pad = wgrad_pad(ksize);
  const int Do = (D + 2 * pad - ksize) / stride + 1, Ho = (H + 2 * pad - ksize) / stride + 1;
  const size_t tiles = (size_t)ksize * ksize * ksize * ((Cout + 31) / 32) * ((Cin + 31) / 32);
  return (size_t)wgrad_chunks(B * Do * Ho, ksize, Cin, Cout) * tiles * 1024 * 4;
}

PW_API int pw_conv3d_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int B, int D,
                           int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
  PW_CHECK_ARG(x && dy && dw && workspace, "pw_conv3d_wgrad: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "pw_conv3d_wgrad: bad shape");
  PW_CHECK_ARG(((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2)) || (ksize == 2 && stride == 2),
               "pw_conv3d_wgrad: ksize 1 | 3 with stride 1 | 2 (padding ksize / 2), or ksize 2 with stride 2 (no padding)");
  PW_CHECK_ARG(workspace_bytes >= pw_conv3d_wgrad_workspace_bytes(B, D, H, W, Cin, Cout, ksize, stride),
               "pw_conv3d_wgrad: workspace too small");
  WgradArgs a;
  a.x = x; a.dy = dy; a.partial = (float*)workspace;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ks = ksize; a.stride = stride; a.pad = wgrad_pad(ksize); a.taps = ksize * ksize * ksize;
  a.Do = (D + 2 * a.pad - ksize) / stride + 1; a.Ho = (H + 2 * a.pad - ksize) / stride + 1; a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
  a.co_blocks = (Cout + 31) / 32; a.ci_blocks = (Cin + 31) / 32;
  a.n_rows = B * a.Do * a.Ho;
  a.n_chunks = wgrad_chunks(a.n_rows, ksize, Cin, Cout);
  a.rows_per_chunk = (a.n_rows + a.n_chunks - 1) / a.n_chunks;
  hipStream_t st = pw_stream(stream);
  const dim3 grid((unsigned)(ksize * ksize * a.co_blocks * a.ci_blocks), (unsigned)a.n_chunks);
  if (ksize == 3 && stride == 1) hipLaunchKernelGGL((k_conv3d_wgrad<3, 1>), grid, dim3(256), 0, st, a);
  else if (ksize == 3) hipLaunchKernelGGL((k_conv3d_wgrad<3, 2>), grid, dim3(256), 0, st, a);
  else if (ksize == 2) hipLaunchKernelGGL((k_conv3d_wgrad<2, 2>), grid, dim3(256), 0, st, a);
  else if (stride == 1) hipLaunchKernelGGL((k_conv3d_wgrad<1, 1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_conv3d_wgrad<1, 2>), grid, dim3(256), 0, st, a);
  const size_t per_chunk = (size_t)a.taps * a.co_blocks * a.ci_blocks * 1024;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(per_chunk / 32)), dim3(256), 0, st, a.partial, dw, a.n_chunks,
                     a.taps, a.co_blocks, a.ci_blocks, Cout, Cin);
  pw_note_kernel("k_conv3d_wgrad");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// dgrad of the 3x3x3 stride-2 pad-1 conv (the two down-sampling stages of CustomResNet3D): a thread owns 4 consecutive input
// channels of one input voxel and walks the taps whose output coordinate is integral ((i + 1 - k) even: one tap on even
// coordinates, two on odd ones, 1..8 taps per voxel).  wt: the weights as [kd][kh][kw][Cout][Cin] (w.permute(2,3,4,0,1)), so the
// 8 lanes of a voxel read one 128-byte row per (tap, co).  Small layers (8.8 + 4.4 GFLOP at the C3 shape): plain fp32 FMAs.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_conv3d_dgrad_s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                        float* __restrict__ dx, int B, int D, int H, int W, int Cin, int Do,
                                                        int Ho, int Wo, int Cout) {
  const int cq = Cin / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * D * H * W * cq;
  if (idx >= total) return;
  const int c4 = (int)(idx % cq) * 4;
  size_t v = idx / cq;
  const int iw = (int)(v % W); v /= W;
  const int ih = (int)(v % H); v /= H;
  const int id = (int)(v % D);
  const int b = (int)(v / D);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kd = (id + 1) & 1; kd < 3; kd += 2) {
    const int od = (id + 1 - kd) >> 1;
    if ((unsigned)od >= (unsigned)Do) continue;
    for (int kh = (ih + 1) & 1; kh < 3; kh += 2) {
      const int oh = (ih + 1 - kh) >> 1;
      if ((unsigned)oh >= (unsigned)Ho) continue;
      for (int kw = (iw + 1) & 1; kw < 3; kw += 2) {
        const int ow = (iw + 1 - kw) >> 1;
        if ((unsigned)ow >= (unsigned)Wo) continue;
        const float* dyr = dy + ((((size_t)b * Do + od) * Ho + oh) * Wo + ow) * Cout;
        const int tap = (kd * 3 + kh) * 3 + kw;
        const float4* wr = reinterpret_cast<const float4*>(w + ((size_t)tap * Cout) * Cin + c4);      // [tap][co][ci]
        const int cq4 = Cin / 4;
#pragma unroll 4
        for (int co = 0; co < Cout; ++co) {
          const float g = dyr[co];
          const float4 wv = wr[(size_t)co * cq4];
          acc[0] = fmaf(g, wv.x, acc[0]); acc[1] = fmaf(g, wv.y, acc[1]);
          acc[2] = fmaf(g, wv.z, acc[2]); acc[3] = fmaf(g, wv.w, acc[3]);
        }
      }
    }
  }
  *reinterpret_cast<float4*>(dx + idx * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

PW_API int pw_conv3d_dgrad_s2(const float* dy, const float* wt, float* dx, int B, int D, int H, int W, int Cin, int Cout,
                              void* stream) {
  PW_CHECK_ARG(dy && wt && dx, "pw_conv3d_dgrad_s2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0, "pw_conv3d_dgrad_s2: bad shape (Cin %% 4)");
  PW_CHECK_ARG((((uintptr_t)dx | (uintptr_t)wt) & 15) == 0, "pw_conv3d_dgrad_s2: dx / wt must be 16-byte aligned");
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)B * D * H * W * (Cin / 4);
  hipLaunchKernelGGL(k_conv3d_dgrad_s2, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, pw_stream(stream), dy, wt, dx, B,
                     D, H, W, Cin, Do, Ho, Wo, Cout);
  pw_note_kernel("k_conv3d_dgrad_s2");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// dgrad of the unpadded 2x2x2 stride-2 convs of the trajectory branch (heads/occupancy_head.py:180-200): every input voxel feeds
// exactly one output voxel through exactly one tap.  wt: [2][2][2][Cout][Cin].  Input voxels beyond 2 * (D / 2) etc. get 0.
__global__ void __launch_bounds__(256) k_conv3d_dgrad_k2s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, int B, int D, int H, int W, int Cin, int Do,
                                                          int Ho, int Wo, int Cout) {
  const int cq = Cin / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * D * H * W * cq;
  if (idx >= total) return;
  const int c4 = (int)(idx % cq) * 4;
  size_t v = idx / cq;
  const int iw = (int)(v % W); v /= W;
  const int ih = (int)(v % H); v /= H;
  const int id = (int)(v % D);
  const int b = (int)(v / D);
  const int od = id >> 1, oh = ih >> 1, ow = iw >> 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (od < Do && oh < Ho && ow < Wo) {
    const float* dyr = dy + ((((size_t)b * Do + od) * Ho + oh) * Wo + ow) * Cout;
    const int tap = ((id & 1) * 2 + (ih & 1)) * 2 + (iw & 1);
    const float4* wr = reinterpret_cast<const float4*>(w + ((size_t)tap * Cout) * Cin + c4);
#pragma unroll 4
    for (int co = 0; co < Cout; ++co) {
      const float g = dyr[co];
      const float4 wv = wr[(size_t)co * cq];
      acc[0] = fmaf(g, wv.x, acc[0]); acc[1] = fmaf(g, wv.y, acc[1]);
      acc[2] = fmaf(g, wv.z, acc[2]); acc[3] = fmaf(g, wv.w, acc[3]);
    }
  }
  *reinterpret_cast<float4*>(dx + idx * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

PW_API int pw_conv3d_dgrad_k2s2(const float* dy, const float* wt, float* dx, int B, int D, int H, int W, int Cin, int Cout,
                                void* stream) {
  PW_CHECK_ARG(dy && wt && dx, "pw_conv3d_dgrad_k2s2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 1 && H > 1 && W > 1 && Cin > 0 && Cin % 4 == 0 && Cout > 0, "pw_conv3d_dgrad_k2s2: bad shape (Cin %% 4)");
  PW_CHECK_ARG((((uintptr_t)dx | (uintptr_t)wt) & 15) == 0, "pw_conv3d_dgrad_k2s2: dx / wt must be 16-byte aligned");
  const size_t total = (size_t)B * D * H * W * (Cin / 4);
  hipLaunchKernelGGL(k_conv3d_dgrad_k2s2, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, pw_stream(stream), dy, wt, dx,
                     B, D, H, W, Cin, D / 2, H / 2, W / 2, Cout);
  pw_note_kernel("k_conv3d_dgrad_k2s2");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// BatchNorm3d in training mode on channels-last rows x (N, C), C in {32, 64, 128, 256} (256 % C == 0).
// Sums are accumulated in double per thread (rows strided over the block), reduced through LDS and over blocks in a fixed
// order: deterministic, and closer to the exact mean / variance than a float cascade.
//   stats:      mean[c], var[c] (biased, what normalisation uses), rstd[c] = 1 / sqrt(var + eps)
//   apply:      y = x_hat gamma + beta (+ residual) (ReLU)
//   bwd_reduce: dz = dy (y > 0 if ReLU);  sum_dz[c], sum_dz_xhat[c]
//   bwd_apply:  dx = gamma rstd (dz - sum_dz / N - x_hat sum_dz_xhat / N);  dres = dz (optional)
// ------------------------------------------------------------------------------------
constexpr int BN_BLOCKS = 1024;
// Largest magnitude of a tensor, recorded by the kernel that writes it: BN_AMAX_PARTS partial maxima (bit patterns of non-negative
// floats) in the format of pw_absmax2 -- pw_conv3d_wgrad_h2 derives its per-tensor power-of-two pre-scales from them, and the
// separate 25-50 us absmax pass over both operands of every weight gradient goes away.  One return-less atomic per block.
constexpr int BN_AMAX_PARTS = 256;
__device__ __forceinline__ void bn_note_amax(unsigned* __restrict__ amax, const float (&o)[4]) {
  __shared__ unsigned wm[4];
  unsigned m = max(max(__float_as_uint(fabsf(o[0])), __float_as_uint(fabsf(o[1]))), max(__float_as_uint(fabsf(o[2])), __float_as_uint(fabsf(o[3]))));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    (void)__hip_atomic_fetch_max(amax + (blockIdx.x & (BN_AMAX_PARTS - 1)), max(max(wm[0], wm[1]), max(wm[2], wm[3])), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
}

// one row of the reduction: 4 channels of a lane.  BWD: dz = dy (masked by y > 0), sums of dz and dz * x_hat; else sums of x and x^2
template <bool BWD>
__device__ __forceinline__ void bn_acc4(const float4& xv, const float4& gv, const float4& yv, bool relu, const float (&m)[4],
                                        const float (&rs)[4], double (&s0)[4], double (&s1)[4]) {
  const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
  if (BWD) {
    float dz[4] = {gv.x, gv.y, gv.z, gv.w};
    if (relu) {
      const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(ye[e] > 0.f)) dz[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0[e] += (double)dz[e]; s1[e] += (double)dz[e] * (double)((xe[e] - m[e]) * rs[e]); }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0[e] += (double)xe[e]; s1[e] += (double)xe[e] * (double)xe[e]; }
  }
}

// A lane owns 4 channels (16-byte loads) and every (256 / (C/4))-th row of its block's row range, four rows in flight.  (The first
// version loaded 4 bytes per lane with one load in flight: 1.7-2.6 TB/s, 1.8 ms of the training step in these two reductions.)
template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_reduce(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                   const float4* __restrict__ y, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, int64_t N, int C, int relu,
                                                   double* __restrict__ partial /* [blocks][2][C] */) {
  __shared__ double red[2][4][256];
  const int cq = C >> 2;
  const int q = threadIdx.x % cq, rsub = threadIdx.x / cq, rstep = 256 / cq;
  const int64_t per = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < N ? r0 + per : N;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  float m[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
  if (BWD) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = mean[4 * q + e]; rs[e] = rstd[4 * q + e]; }
  }
  const bool mask = BWD && relu;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0 + rsub;
  for (; r + 3 * rstep < r1; r += 4 * rstep) {
    float4 xv[4], gv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = (r + u * rstep) * cq + q;
      xv[u] = x[i];
      gv[u] = BWD ? dy[i] : zero;
      yv[u] = mask ? y[i] : zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) bn_acc4<BWD>(xv[u], gv[u], yv[u], mask, m, rs, s0, s1);
  }
  for (; r < r1; r += rstep) {
    const int64_t i = r * cq + q;
    bn_acc4<BWD>(x[i], BWD ? dy[i] : zero, mask ? y[i] : zero, mask, m, rs, s0, s1);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] = s0[e]; red[1][e][threadIdx.x] = s1[e]; }
  __syncthreads();
  if (threadIdx.x < C) {
    const int qq = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a0 = 0.0, a1 = 0.0;
    for (int g = 0; g < rstep; ++g) { a0 += red[0][e][g * cq + qq]; a1 += red[1][e][g * cq + qq]; }
    partial[((size_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = a0;
    partial[((size_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = a1;
  }
}

// out0 / out1: (mean, var) + rstd for the forward statistics, (sum_dz, sum_dz_xhat) for the backward sums.  One wave per channel:
// lane l adds the partials of blocks l, l + 64, ... in order, then a fixed xor tree combines the lanes (deterministic; the first
// version walked all 512 partials with one thread per channel: 129 us per call, 16 % of a training step)
__global__ void __launch_bounds__(64) k_bn_finish(const double* __restrict__ partial, int blocks, int C, int64_t N, float eps,
                                                  int fwd, float* __restrict__ out0, float* __restrict__ out1,
                                                  float* __restrict__ rstd, unsigned* __restrict__ amax_clear,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                                  int64_t* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x, lane = threadIdx.x;
  // the BN_AMAX_PARTS partial maxima the apply kernel that follows on this stream raises (bn_note_amax) start from zero
  if (amax_clear)
    for (int i = c * 64 + lane; i < BN_AMAX_PARTS; i += gridDim.x * 64) amax_clear[i] = 0u;
  double a0 = 0.0, a1 = 0.0;
  for (int b = lane; b < blocks; b += 64) { a0 += partial[((size_t)b * 2 + 0) * C + c]; a1 += partial[((size_t)b * 2 + 1) * C + c]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
  }
  if (lane != 0) return;
  if (fwd) {
    const double mu = a0 / (double)N;
    double var = a1 / (double)N - mu * mu;
    if (var < 0.0) var = 0.0;
    out0[c] = (float)mu;
    out1[c] = (float)var;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {                          // nn.BatchNorm3d's bookkeeping, as k_bn_update_running does it
      const float n = (float)N, unbias = n / fmaxf(n - 1.f, 1.f);
      running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * (float)mu;
      running_var[c] = running_var[c] * (1.f - momentum) + momentum * ((float)var * unbias);
      if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    }
  } else {
    out0[c] = (float)a0;
    out1[c] = (float)a1;
  }
}

PW_API size_t pw_bn_workspace_bytes(int C) { return (size_t)BN_BLOCKS * 2 * C * sizeof(double); }

// blocks of a reduction: at least 4 rows per lane (the finishing kernel walks the block partials), at most BN_BLOCKS
static int bn_blocks(int64_t N, int C) {
  const int64_t rows_per_pass = 256 / (C / 4);
  const int64_t want = pw_cdiv(N, rows_per_pass * 4);
  return (int)(want < 1 ? 1 : (want > BN_BLOCKS ? BN_BLOCKS : want));
}

static int bn_check(int64_t N, int C, const void* ws, size_t ws_bytes, const char* who) {
  if (!(N > 0 && C > 0 && C <= 256 && 256 % C == 0 && C % 4 == 0)) {
    pw_set_error("%s: C must divide 256 and be a multiple of 4 (got N=%lld C=%d)", who, (long long)N, C);
    return PW_EINVAL;
  }
  if (!ws || ws_bytes < pw_bn_workspace_bytes(C)) {
    pw_set_error("%s: workspace too small", who);
    return PW_ENOSPC;
  }
  return PW_OK;
}

PW_API int pw_bn_stats(const float* x, int64_t N, int C, float eps, void* workspace, size_t workspace_bytes, float* mean,
                       float* var, float* rstd, float* amax_clear, float* running_mean, float* running_var, float momentum,
                       int64_t* num_batches_tracked, void* stream) {
  PW_CHECK_ARG(x && mean && var && rstd && ((uintptr_t)x & 15) == 0, "pw_bn_stats: null or misaligned pointer");
  PW_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pw_bn_stats: running_mean and running_var come together");
  if (int rc = bn_check(N, C, workspace, workspace_bytes, "pw_bn_stats")) return rc;
  const int blocks = bn_blocks(N, C);
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_bn_reduce<false>, dim3(blocks), dim3(256), 0, st, (const float4*)x, (const float4*)nullptr, (const float4*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, N, C, 0, (double*)workspace);
  hipLaunchKernelGGL(k_bn_finish, dim3(C), dim3(64), 0, st, (const double*)workspace, blocks, C, N, eps, 1, mean, var, rstd,
                     (unsigned*)amax_clear, running_mean, running_var, momentum, num_batches_tracked);
  pw_note_kernel("k_bn_reduce<false>");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// nn.BatchNorm3d's running-statistics bookkeeping (torch/nn/modules/batchnorm.py) in one launch: exponential average with the
// UNBIASED batch variance, num_batches_tracked += 1.  (As torch ops it is ten tiny kernels per BatchNorm -- 270 launches per step.)
__global__ void __launch_bounds__(256) k_bn_update_running(const float* __restrict__ mean, const float* __restrict__ var, int C,
                                                           double n_host, const float* __restrict__ n_dev, float momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           int64_t* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
  if (c >= C) return;
  const float n = n_dev ? n_dev[0] : (float)n_host;
  const float unbias = n / fmaxf(n - 1.f, 1.f);
  running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mean[c];
  running_var[c] = running_var[c] * (1.f - momentum) + momentum * (var[c] * unbias);
}

PW_API int pw_bn_update_running(const float* mean, const float* var, int C, double n_rows, const float* n_rows_dev, float momentum,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream) {
  PW_CHECK_ARG(mean && var && running_mean && running_var && C > 0 && (n_rows_dev || n_rows > 0), "pw_bn_update_running: bad arguments");
  hipLaunchKernelGGL(k_bn_update_running, dim3((unsigned)pw_cdiv(C, 256)), dim3(256), 0, pw_stream(stream), mean, var, C, n_rows,
                     n_rows_dev, momentum, running_mean, running_var, num_batches_tracked);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

__global__ void __launch_bounds__(256) k_bn_apply(const float4* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float4* __restrict__ residual,
                                                  int64_t n4, int C, int relu, float4* __restrict__ y,
                                                  unsigned* __restrict__ amax) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const int c = (int)((i * 4) % C);
    const float4 v = x[i];
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (o[e] - mean[c + e]) * rstd[c + e] * gamma[c + e] + beta[c + e];
    if (residual) {
      const float4 r = residual[i];
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    y[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (amax) bn_note_amax(amax, o);
}

PW_API int pw_bn_apply(const float* x, int64_t N, int C, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* residual, int relu, float* y, float* y_amax, void* stream) {
  PW_CHECK_ARG(x && mean && rstd && gamma && beta && y && N > 0 && C > 0 && C % 4 == 0, "pw_bn_apply: bad arguments");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, "pw_bn_apply: tensors must be 16-byte aligned");
  const int64_t n4 = N * C / 4;
  hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)pw_cdiv(n4, 256)), dim3(256), 0, pw_stream(stream), (const float4*)x, mean, rstd,
                     gamma, beta, (const float4*)residual, n4, C, relu, (float4*)y, (unsigned*)y_amax);
  pw_note_kernel("k_bn_apply");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_bn_bwd_reduce(const float* x, const float* dy, const float* y, int64_t N, int C, const float* mean,
                            const float* rstd, int relu, void* workspace, size_t workspace_bytes, float* sum_dz,
                            float* sum_dz_xhat, float* amax_clear, void* stream) {
  PW_CHECK_ARG(x && dy && mean && rstd && sum_dz && sum_dz_xhat && (!relu || y) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)y) & 15) == 0,
               "pw_bn_bwd_reduce: null or misaligned pointer");
  if (int rc = bn_check(N, C, workspace, workspace_bytes, "pw_bn_bwd_reduce")) return rc;
  const int blocks = bn_blocks(N, C);
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_bn_reduce<true>, dim3(blocks), dim3(256), 0, st, (const float4*)x, (const float4*)dy, (const float4*)y, mean, rstd, N, C,
                     relu, (double*)workspace);
  hipLaunchKernelGGL(k_bn_finish, dim3(C), dim3(64), 0, st, (const double*)workspace, blocks, C, N, 0.f, 0, sum_dz, sum_dz_xhat,
                     (float*)nullptr, (unsigned*)amax_clear, (float*)nullptr, (float*)nullptr, 0.f, (int64_t*)nullptr);
  pw_note_kernel("k_bn_reduce<true>");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                      const float4* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ sum_dz, const float* __restrict__ sum_dz_xhat,
                                                      int64_t n4, int C, float inv_n, int relu, float4* __restrict__ dx,
                                                      float4* __restrict__ dres, unsigned* __restrict__ amax) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const int c = (int)((i * 4) % C);
    const float4 xv = x[i], gv = dy[i];
    float xe[4] = {xv.x, xv.y, xv.z, xv.w}, dz[4] = {gv.x, gv.y, gv.z, gv.w};
    if (relu) {
      const float4 yv = y[i];
      const float ye[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(ye[e] > 0.f)) dz[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xe[e] - mean[c + e]) * rstd[c + e];
      o[e] = gamma[c + e] * rstd[c + e] * (dz[e] - sum_dz[c + e] * inv_n - xh * sum_dz_xhat[c + e] * inv_n);
    }
    dx[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) dres[i] = make_float4(dz[0], dz[1], dz[2], dz[3]);
  }
  if (amax) bn_note_amax(amax, o);
}

PW_API int pw_bn_bwd_apply(const float* x, const float* dy, const float* y, int64_t N, int C, const float* mean,
                           const float* rstd, const float* gamma, const float* sum_dz, const float* sum_dz_xhat, int relu,
                           float* dx, float* dres, float* dx_amax, void* stream) {
  PW_CHECK_ARG(x && dy && mean && rstd && gamma && sum_dz && sum_dz_xhat && dx && (!relu || y) && N > 0 && C > 0 && C % 4 == 0,
               "pw_bn_bwd_apply: bad arguments");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0,
               "pw_bn_bwd_apply: tensors must be 16-byte aligned");
  const int64_t n4 = N * C / 4;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)pw_cdiv(n4, 256)), dim3(256), 0, pw_stream(stream), (const float4*)x,
                     (const float4*)dy, (const float4*)y, mean, rstd, gamma, sum_dz, sum_dz_xhat, n4, C, 1.f / (float)N, relu,
                     (float4*)dx, (float4*)dres, (unsigned*)dx_amax);
  pw_note_kernel("k_bn_bwd_apply");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// Trilinear up-sampling (align_corners=True) of a channels-last map and its adjoint -- the LSSFPN3D of
// mmdet3d/models/necks/lss_fpn.py:132-148 in training: with the 1x1x1 conv commuted below the up-sampling only the
// 32-channel maps are interpolated.  Source index / weights exactly as torch's upsample_trilinear3d: src = dst (in-1)/(out-1).
//   pw_upsample_trilinear_add      hi (+)= up(lo)                        one thread per (hi voxel, 4 channels)
//   pw_upsample_trilinear_adjoint  dlo = up^T(dhi)                        one axis at a time (k_upsample_axis_adjoint)
// ------------------------------------------------------------------------------------
struct UpArgs {
  const float* src;
  float* dst;
  int B, Dl, Hl, Wl, Dh, Hh, Wh, C;
  float sd, sh, sw;                 // (in - 1) / (out - 1), 0 when out == 1
  int accumulate;
};

__device__ __forceinline__ void up_axis(int d, float s, int n_in, int& i0, int& i1, float& l0, float& l1) {
  const float src = s * (float)d;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256) k_upsample_add(UpArgs a) {
  const int cq = a.C / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.B * a.Dh * a.Hh * a.Wh * cq;
  if (idx >= total) return;
  const int c4 = (int)(idx % cq);
  size_t v = idx / cq;
  const int x = (int)(v % a.Wh); v /= a.Wh;
  const int y = (int)(v % a.Hh); v /= a.Hh;
  const int z = (int)(v % a.Dh);
  const int b = (int)(v / a.Dh);
  int z0, z1, y0, y1, x0, x1;
  float lz0, lz1, ly0, ly1, lx0, lx1;
  up_axis(z, a.sd, a.Dl, z0, z1, lz0, lz1);
  up_axis(y, a.sh, a.Hl, y0, y1, ly0, ly1);
  up_axis(x, a.sw, a.Wl, x0, x1, lx0, lx1);
  const float4* lo = reinterpret_cast<const float4*>(a.src) + (size_t)b * a.Dl * a.Hl * a.Wl * cq + c4;
  auto at = [&](int zz, int yy, int xx) { return lo[(((size_t)zz * a.Hl + yy) * a.Wl + xx) * cq]; };
  const float4 p000 = at(z0, y0, x0), p001 = at(z0, y0, x1), p010 = at(z0, y1, x0), p011 = at(z0, y1, x1);
  const float4 p100 = at(z1, y0, x0), p101 = at(z1, y0, x1), p110 = at(z1, y1, x0), p111 = at(z1, y1, x1);
  float4* out = reinterpret_cast<float4*>(a.dst) + idx;
  float4 o = a.accumulate ? *out : make_float4(0.f, 0.f, 0.f, 0.f);
#define PW_TRI(f) (lz0 * (ly0 * (lx0 * p000.f + lx1 * p001.f) + ly1 * (lx0 * p010.f + lx1 * p011.f)) + \
                   lz1 * (ly0 * (lx0 * p100.f + lx1 * p101.f) + ly1 * (lx0 * p110.f + lx1 * p111.f)))
  o.x += PW_TRI(x); o.y += PW_TRI(y); o.z += PW_TRI(z); o.w += PW_TRI(w);
#undef PW_TRI
  *out = o;
}

// hi-index range whose corners can include lo index j: floor(s d) in {j - 1, j}
__device__ __forceinline__ void adj_range(int j, float s, int n_out, int& lo, int& hi) {
  if (s <= 0.f) { lo = 0; hi = n_out - 1; return; }
  lo = (int)floorf((float)(j - 1) / s) - 1;
  hi = (int)ceilf((float)(j + 1) / s) + 1;
  if (lo < 0) lo = 0;
  if (hi > n_out - 1) hi = n_out - 1;
}
__device__ __forceinline__ float adj_weight(int d, int j, float s, int n_in) {
  int i0, i1;
  float l0, l1;
  up_axis(d, s, n_in, i0, i1, l0, l1);
  return (i0 == j ? l0 : 0.f) + (i1 == j ? l1 : 0.f);
}

// one axis of the adjoint: src (outer, n_hi, inner) -> dst (outer, n_lo, inner), inner = float4s per index along the axis.  The
// interpolation is a tensor product of three 1-D operators, so is its adjoint: W, then H, then D, each thread gathering the <= 2/s + 3
// fine indices whose corners include its coarse index (deterministic, no atomics).  (The 3-D gather of round 3 looped over up to 11^3
// fine voxels from 80 000 threads: 354 us for the 4x level at 200x200x16; the three passes move 82 + 20 + 5 MB.)
__global__ void __launch_bounds__(256) k_upsample_axis_adjoint(const float4* __restrict__ src, float4* __restrict__ dst, size_t outer,
                                                               int n_hi, int n_lo, int inner, float s) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= outer * n_lo * inner) return;
  const int i4 = (int)(idx % inner);
  size_t v = idx / inner;
  const int j = (int)(v % n_lo);
  const size_t o = v / n_lo;
  int lo, hi;
  adj_range(j, s, n_hi, lo, hi);
  const float4* p = src + o * n_hi * inner + i4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int d = lo; d <= hi; ++d) {
    const float w = adj_weight(d, j, s, n_lo);
    if (w == 0.f) continue;
    const float4 g = p[(size_t)d * inner];
    acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y); acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
  }
  dst[idx] = acc;
}

static int up_args(UpArgs& a, const float* src, float* dst, int B, int Dl, int Hl, int Wl, int Dh, int Hh, int Wh, int C,
                   const char* who) {
  if (!(src && dst && B > 0 && Dl > 0 && Hl > 0 && Wl > 0 && Dh >= Dl && Hh >= Hl && Wh >= Wl && C > 0 && C % 4 == 0)) {
    pw_set_error("%s: bad arguments", who);
    return PW_EINVAL;
  }
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0) {
    pw_set_error("%s: tensors must be 16-byte aligned", who);
    return PW_EINVAL;
  }
  a.src = src; a.dst = dst; a.B = B; a.Dl = Dl; a.Hl = Hl; a.Wl = Wl; a.Dh = Dh; a.Hh = Hh; a.Wh = Wh; a.C = C;
  a.sd = Dh > 1 ? (float)(Dl - 1) / (float)(Dh - 1) : 0.f;
  a.sh = Hh > 1 ? (float)(Hl - 1) / (float)(Hh - 1) : 0.f;
  a.sw = Wh > 1 ? (float)(Wl - 1) / (float)(Wh - 1) : 0.f;
  a.accumulate = 0;
  return PW_OK;
}

PW_API int pw_upsample_trilinear_add(const float* lo, float* hi, int B, int Dl, int Hl, int Wl, int Dh, int Hh, int Wh, int C,
                                     int accumulate, void* stream) {
  UpArgs a;
  if (int rc = up_args(a, lo, hi, B, Dl, Hl, Wl, Dh, Hh, Wh, C, "pw_upsample_trilinear_add")) return rc;
  a.accumulate = accumulate;
  const size_t total = (size_t)B * Dh * Hh * Wh * (C / 4);
  hipLaunchKernelGGL(k_upsample_add, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, pw_stream(stream), a);
  pw_note_kernel("k_upsample_add");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API size_t pw_upsample_trilinear_adjoint_workspace_bytes(int B, int Dl, int Hl, int Wl, int Dh, int Hh, int Wh, int C) {
  return ((size_t)B * Dh * Hh * Wl + (size_t)B * Dh * Hl * Wl) * C * sizeof(float) + 256;
}

PW_API int pw_upsample_trilinear_adjoint(const float* dhi, float* dlo, void* workspace, size_t workspace_bytes, int B, int Dl, int Hl,
                                         int Wl, int Dh, int Hh, int Wh, int C, void* stream) {
  UpArgs a;
  if (int rc = up_args(a, dhi, dlo, B, Dl, Hl, Wl, Dh, Hh, Wh, C, "pw_upsample_trilinear_adjoint")) return rc;
  PW_CHECK_ARG(workspace && workspace_bytes >= pw_upsample_trilinear_adjoint_workspace_bytes(B, Dl, Hl, Wl, Dh, Hh, Wh, C) &&
               ((uintptr_t)workspace & 15) == 0, "pw_upsample_trilinear_adjoint: workspace missing, misaligned or too small");
  float4* t1 = static_cast<float4*>(workspace);                                    // (B, Dh, Hh, Wl, C)
  float4* t2 = t1 + (size_t)B * Dh * Hh * Wl * (C / 4);                            // (B, Dh, Hl, Wl, C)
  hipStream_t st = pw_stream(stream);
  const int cq = C / 4;
  auto pass = [&](const float4* src, float4* dst, size_t outer, int n_hi, int n_lo, int inner, float s) {
    const size_t total = outer * n_lo * inner;
    hipLaunchKernelGGL(k_upsample_axis_adjoint, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, st, src, dst, outer, n_hi,
                       n_lo, inner, s);
  };
  pass(reinterpret_cast<const float4*>(dhi), t1, (size_t)B * Dh * Hh, Wh, Wl, cq, a.sw);
  pass(t1, t2, (size_t)B * Dh, Hh, Hl, Wl * cq, a.sh);
  pass(t2, reinterpret_cast<float4*>(dlo), (size_t)B, Dh, Dl, Hl * Wl * cq, a.sd);
  pw_note_kernel("k_upsample_axis_adjoint");
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// ------------------------------------------------------------------------------------
// Per-voxel dense layers with a handful of channels (OccHead's 1x1x1 convs 16 -> 8 -> 18 and its soft-weight branch 16 -> 8 -> 1,
// heads/occupancy_head.py:124-161; their data gradients are the same op with the weight transposed).  y[n][j] = sum_k x[n][k] w[j][k].
// As library GEMMs (M = 640 000, K = 16, N = 8) these took 1.0-1.5 ms each in the training step; they are a 60 MB stream:
// one thread per row, the weights are wave-uniform (scalar loads), K x N FMAs per row.
// ------------------------------------------------------------------------------------
template <int K, int N>
__global__ void __launch_bounds__(256) k_linear_rows(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xv[K];
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(x + i * K)[q];
      xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
    }
  } else if constexpr (K % 2 == 0) {
#pragma unroll
    for (int q = 0; q < K / 2; ++q) {
      const float2 v = reinterpret_cast<const float2*>(x + i * K)[q];
      xv[2 * q] = v.x; xv[2 * q + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) xv[k] = x[i * K + k];
  }
  float out[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) acc = fmaf(xv[k], w[j * K + k], acc);
    out[j] = acc;
  }
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q)
      reinterpret_cast<float4*>(y + i * N)[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int q = 0; q < N / 2; ++q) reinterpret_cast<float2*>(y + i * N)[q] = make_float2(out[2 * q], out[2 * q + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) y[i * N + j] = out[j];
  }
}

PW_API int pw_linear_rows(const float* x, const float* w, float* y, int64_t n, int K, int N, void* stream) {
  PW_CHECK_ARG(x && w && y && n > 0, "pw_linear_rows: bad arguments");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_linear_rows: x / y must be 16-byte aligned");
  hipStream_t st = pw_stream(stream);
  const dim3 grid((unsigned)pw_cdiv(n, 256));
#define PW_LR(KK, NN) \
  if (K == KK && N == NN) { hipLaunchKernelGGL((k_linear_rows<KK, NN>), grid, dim3(256), 0, st, x, w, y, n); PW_CHECK_LAUNCH(); return PW_OK; }
  PW_LR(16, 8) PW_LR(8, 18) PW_LR(8, 1) PW_LR(8, 16) PW_LR(18, 8) PW_LR(1, 8) PW_LR(32, 16) PW_LR(16, 32)
#undef PW_LR
  pw_set_error("pw_linear_rows: (K, N) = (%d, %d) is not built (16x8, 8x18, 8x1, 8x16, 18x8, 1x8, 32x16, 16x32)", K, N);
  return PW_EUNSUP;
}

// ------------------------------------------------------------------------------------
// Weight packing for the fp32 conv kernels in ONE launch each (training re-packs every weight every step: as torch ops -- double(),
// einsum, zeros, copy, permute, contiguous, float -- a pack cost ~10 launches and ~0.3 ms of host time, 60 packs per step made the
// eager training step host-bound).  flip_t: pack w' = w.flip(2, 3, 4).transpose(0, 1), the weight of the stride-1 data gradient,
// without materialising it.  Layouts: preworld_amd.ops.pack_conv_weight / pack_conv_weight_wino (include/preworld_hip.h).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float packed_w(const float* __restrict__ w, int Cout_w, int Cin_w, int k, int flip_t, int n, int c, int a,
                                          int b, int cc) {
  // element [n][c][a][b][cc] of w (flip_t = 0) or of w.flip(2,3,4).transpose(0,1) (flip_t = 1); zero outside
  const int co = flip_t ? c : n, ci = flip_t ? n : c;
  if (co >= Cout_w || ci >= Cin_w) return 0.f;
  if (flip_t) { a = k - 1 - a; b = k - 1 - b; cc = k - 1 - cc; }
  return w[((((size_t)co * Cin_w + ci) * k + a) * k + b) * k + cc];
}

__global__ void __launch_bounds__(256) k_pack_conv_weight(const float* __restrict__ w, int Cout_w, int Cin_w, int k, int flip_t,
                                                          int cout_total, int cin_total, float* __restrict__ out) {
  // out[ch][tap][nt][q][h * 32 + j][e] = w'[nt * 32 + j][ch * 32 + h * 16 + 4 q + e][tap]
  const int taps = k * k * k, nt_n = cout_total / 32;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)cin_total * taps * cout_total;
  if (idx >= total) return;
  size_t t = idx;
  const int e = (int)(t & 3); t >>= 2;
  const int j = (int)(t & 31); t >>= 5;
  const int h = (int)(t & 1); t >>= 1;
  const int q = (int)(t & 3); t >>= 2;
  const int nt = (int)(t % nt_n); t /= nt_n;
  const int tap = (int)(t % taps);
  const int ch = (int)(t / taps);
  const int n = nt * 32 + j, c = ch * 32 + h * 16 + 4 * q + e;
  out[idx] = packed_w(w, Cout_w, Cin_w, k, flip_t, n, c, tap / (k * k), (tap / k) % k, tap % k);
}

__global__ void __launch_bounds__(256) k_pack_conv_weight_wino(const float* __restrict__ w, int Cout_w, int Cin_w, int flip_t,
                                                               int cout_total, int cin_total, float* __restrict__ out) {
  // out[ch][p][n16][g][j][s] = U[n16 * 16 + j][ch * 32 + g * 8 + s][p], U = G w' G^T along d, h, w in float64, rounded once.
  // One thread per (output channel n, input channel c): 27 loads, the transform one axis at a time (G = [1 0 0; .5 .5 .5;
  // .5 -.5 .5; 0 0 1]: halvings and sums), 64 stores.  (One thread per OUTPUT element with the 27-term triple sum: 1.1 ms per step.)
  const int n16_n = cout_total / 16;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // c fastest: 8 consecutive threads write 32 contiguous bytes
  if (idx >= (size_t)cin_total * cout_total) return;
  const int c = (int)(idx % cin_total), n = (int)(idx / cin_total);
  double t0[3][3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double w0 = packed_w(w, Cout_w, Cin_w, 3, flip_t, n, c, a, b, 0), w1 = packed_w(w, Cout_w, Cin_w, 3, flip_t, n, c, a, b, 1),
                   w2 = packed_w(w, Cout_w, Cin_w, 3, flip_t, n, c, a, b, 2);
      t0[a][b][0] = w0; t0[a][b][1] = 0.5 * ((w0 + w1) + w2); t0[a][b][2] = 0.5 * ((w0 - w1) + w2); t0[a][b][3] = w2;
    }
  double t1[3][4][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double w0 = t0[a][0][k], w1 = t0[a][1][k], w2 = t0[a][2][k];
      t1[a][0][k] = w0; t1[a][1][k] = 0.5 * ((w0 + w1) + w2); t1[a][2][k] = 0.5 * ((w0 - w1) + w2); t1[a][3][k] = w2;
    }
  const int ch = c >> 5, g = (c >> 3) & 3, s8 = c & 7, n16 = n >> 4, j = n & 15;
  float* dst = out + ((((size_t)ch * 64 * n16_n + n16) * 4 + g) * 16 + j) * 8 + s8;          // point 0; points are n16_n * 512 floats apart
  const size_t pstride = (size_t)n16_n * 512;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double w0 = t1[0][jj][k], w1 = t1[1][jj][k], w2 = t1[2][jj][k];
      const double u[4] = {w0, 0.5 * ((w0 + w1) + w2), 0.5 * ((w0 - w1) + w2), w2};
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(size_t)(i * 16 + jj * 4 + k) * pstride] = (float)u[i];
    }
}

PW_API int pw_pack_conv_weight(const float* w, int Cout, int Cin, int ksize, int flip_t, int cout_total, float* out, int wino,
                               void* stream) {
  PW_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && ksize >= 1 && ksize <= 3 && cout_total > 0, "pw_pack_conv_weight: bad arguments");
  const int cout_p = flip_t ? Cin : Cout, cin_p = flip_t ? Cout : Cin;         // shape of the packed (possibly transposed) weight
  PW_CHECK_ARG(cin_p % 32 == 0 && cout_total % 32 == 0 && cout_total >= cout_p, "pw_pack_conv_weight: packed Cin and cout_total must be multiples of 32");
  PW_CHECK_ARG(!wino || ksize == 3, "pw_pack_conv_weight: the Winograd layout is for 3x3x3 weights");
  hipStream_t st = pw_stream(stream);
  if (wino) {
    const size_t total = (size_t)cin_p * cout_total;
    hipLaunchKernelGGL(k_pack_conv_weight_wino, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, st, w, Cout, Cin, flip_t,
                       cout_total, cin_p, out);
  } else {
    const size_t total = (size_t)cin_p * ksize * ksize * ksize * cout_total;
    hipLaunchKernelGGL(k_pack_conv_weight, dim3((unsigned)pw_cdiv((int64_t)total, 256)), dim3(256), 0, st, w, Cout, Cin, ksize, flip_t,
                       cout_total, cin_p, out);
  }
  PW_CHECK_LAUNCH();
  return PW_OK;
}
