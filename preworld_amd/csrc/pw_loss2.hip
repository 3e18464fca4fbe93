// The finetune configs' other two voxel losses (preworld.py:146-155 with
// use_focal_loss=True, weight_voxel_lovasz=1.0):
//
//  * CustomFocalLoss (mmdet3d/models/loss_utils/focal_loss.py:163-262): per valid voxel the sigmoid focal loss of
//    its C logits (mmcv-full's sigmoid_focal_loss op, third-party: -[t==c] a (1-p)^g log(max(p,FLT_MIN))
//    - [t!=c] (1-a) p^g log(max(1-p,FLT_MIN)), p = sigmoid(x)) weighted by class_weights[c] * radial_map[h][w],
//    summed over classes, averaged over the valid voxels, times loss_weight.  The reference gathers the valid rows
//    (nonzero + index), builds an (N, C) weight matrix and calls the op; here it is one pass over the logits
//    (HBM-bound: 46 MB read once) and one more for the gradient.
//
//  * lovasz_softmax (mmdet3d/models/detectors/lovasz_softmax.py:157-232, classes='present'): for every class
//    present among the valid voxels, errors |fg - p_c| sorted descending . lovasz_grad(fg_sorted).  The reference
//    runs C sequential torch.sort + cumsum passes; here ALL classes are sorted at once as 32-bit keys
//    (class << 27 | 27-bit descending-error key, lv_key) by one device-wide radix sort (rocPRIM -- the only library call on this path),
//    and the Jaccard gradient of every element comes from a two-level scan of the sorted foreground flags.
//    Counts stay integers (exact below 2^24, where the reference's float cumsum is exact too).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "pw_common.h"
#include "pw_vox.h"

namespace {
struct VoxArgs {
  const float* x;          // logits (focal) or probabilities (lovasz), (B, C, X, Y, Z) with element strides
  const uint8_t* target;   // (B, X, Y, Z) dense
  const uint8_t* cam;      // or null
  const float* cw;         // class weights or null
  int B, C, X, Y, Z;
  long long sb, sc, sx, sy, sz;
  int ignore;
  VoxWalk walk;
};

__device__ __forceinline__ long long vox_decode(const VoxArgs& a, long long v, int& xh, int& yw) {
  const int z = (int)(v % a.Z); long long t = v / a.Z;
  yw = (int)(t % a.Y); t /= a.Y;
  xh = (int)(t % a.X);
  const int b = (int)(t / a.X);
  return b * a.sb + xh * a.sx + yw * a.sy + z * a.sz;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;                                   // valid in thread 0
}

// -------------------------------------------------------------------------------------------- focal
struct FocalPar { float gamma, alpha, hx, hy, inv_cmax; };

__device__ __forceinline__ float radial(const FocalPar& f, int xh, int yw) {
  const float dx = (float)xh - f.hx, dy = (float)yw - f.hy;
  return sqrtf(dx * dx + dy * dy) * f.inv_cmax + 1.f;
}

__device__ __forceinline__ float powg(float b, float gamma) { return gamma == 2.f ? b * b : powf(b, gamma); }

__global__ void __launch_bounds__(256) k_focal_stats(VoxArgs a, FocalPar f, long long n_vox, double* __restrict__ stats) {
  __shared__ float red[4];
  float sum = 0.f, cnt = 0.f;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_vox; u += (long long)gridDim.x * blockDim.x) {
    int b, xh, yw, z;
    long long v;
    const long long off = vox_walk(a, u, b, xh, yw, z, v);
    const int t = a.target[v];
    if (t == a.ignore || (a.cam && !a.cam[v])) continue;
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float p = 1.f / (1.f + __expf(-a.x[off + c * a.sc]));
      const float el = c == t ? -f.alpha * powg(1.f - p, f.gamma) * __logf(fmaxf(p, 1.17549435e-38f))
                              : -(1.f - f.alpha) * powg(p, f.gamma) * __logf(fmaxf(1.f - p, 1.17549435e-38f));
      s += (a.cw ? a.cw[c] : 1.f) * el;
    }
    sum += s * radial(f, xh, yw);
    cnt += 1.f;
  }
  const float bs = block_sum(sum, red), bc = block_sum(cnt, red);
  if (threadIdx.x == 0) { atomicAdd(stats, (double)bs); atomicAdd(stats + 1, (double)bc); }
}

__global__ void k_focal_finish(const double* __restrict__ stats, float loss_weight, float* __restrict__ loss) {
  loss[0] = (float)((double)loss_weight * stats[0] / stats[1]);
}

__global__ void __launch_bounds__(256) k_focal_grad(VoxArgs a, FocalPar f, long long n_vox, const double* __restrict__ stats,
                                                    float loss_weight, const float* __restrict__ gout, float* __restrict__ grad) {
  const float coef = gout[0] * loss_weight / (float)stats[1];
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_vox; u += (long long)gridDim.x * blockDim.x) {
    int b, xh, yw, z;
    long long v;
    const long long off = vox_walk(a, u, b, xh, yw, z, v);
    const int t = a.target[v];
    const bool valid = t != a.ignore && !(a.cam && !a.cam[v]);
    const float wv = valid ? coef * radial(f, xh, yw) : 0.f;
    for (int c = 0; c < a.C; ++c) {
      float g = 0.f;
      if (valid) {
        const float p = 1.f / (1.f + __expf(-a.x[off + c * a.sc]));
        // mmcv sigmoid_focal_loss backward: d/dx of the two terms
        const float gp = powg(1.f - p, f.gamma) * (1.f - p - f.gamma * p * __logf(fmaxf(p, 1.17549435e-38f)));
        const float gn = powg(p, f.gamma) * (f.gamma * (1.f - p) * __logf(fmaxf(1.f - p, 1.17549435e-38f)) - p);
        g = (c == t ? -f.alpha * gp : -(1.f - f.alpha) * gn) * (a.cw ? a.cw[c] : 1.f) * wv;
      }
      grad[off + c * a.sc] = g;
    }
  }
}

// -------------------------------------------------------------------------------------------- lovasz
// Sort key of one (voxel, class) error e = |fg - p| in [0, 1]: ascending key = descending error.  The bit pattern of e is at most
// 0x3F800000, so the top two bits of its complement are always set: dropped; so are the 3 lowest mantissa bits (errors that agree to
// 2^-20 sort as ties, in walk order, and are decoded to the middle of their bucket) -- class (5 bits) + 27 bits make a 32-bit key:
// 4 radix passes over 8-byte pairs where the 64-bit key took 5 passes over 12-byte pairs (546 -> ~250 us on 10.9 M pairs).
// LV_KEY_ZERO: e == 0 (and denormals below 2^-146), whose gradient is 0; LV_KEY_INVALID (ignored / masked voxels) sorts last.
constexpr int LV_KEY_BITS = 27;
constexpr unsigned LV_KEY_INVALID = (1u << LV_KEY_BITS) - 1u, LV_KEY_ZERO = LV_KEY_INVALID - 1u;
__device__ __forceinline__ unsigned lv_key(float e) { return min((~__float_as_uint(e) & 0x3FFFFFFFu) >> 3, LV_KEY_ZERO); }
__device__ __forceinline__ float lv_key_error(unsigned k) { return __uint_as_float(~(0xC0000000u | (k << 3) | 4u)); }

constexpr int LV_BLOCK = 1024;                 // sorted elements per block of the scan passes (256 threads x 4)

struct LovaszWs {
  unsigned* keys_in; unsigned* keys_out;   // (class << 27) | 27-bit descending-error key, see lv_key
  unsigned* vals_in; unsigned* vals_out;
  unsigned* cnt;           // [C] foreground count per class, [C] = number of valid voxels
  unsigned* bsum;          // [C][nb] foreground count per block of the sorted segment -> exclusive prefix
  double* lossc;           // [C]
  void* temp; size_t temp_bytes;
};

__device__ __forceinline__ int seg_of(int c, int ignore, int C) { return c - ((ignore >= 0 && ignore < C && c > ignore) ? 1 : 0); }

__global__ void __launch_bounds__(256) k_lovasz_keys(VoxArgs a, long long n_vox, LovaszWs w) {
  __shared__ unsigned scnt[33];
  if (threadIdx.x < 33) scnt[threadIdx.x] = 0;
  __syncthreads();
  const bool ign_in = a.ignore >= 0 && a.ignore < a.C;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_vox; u += (long long)gridDim.x * blockDim.x) {
    int b, xh, yw, z;
    long long v;
    const long long off = vox_walk(a, u, b, xh, yw, z, v);
    const int t = a.target[v];
    const bool valid = t != a.ignore && !(a.cam && !a.cam[v]);
    if (valid) {
      atomicAdd(&scnt[32], 1u);
      if (t < a.C) atomicAdd(&scnt[t], 1u);
    }
    for (int c = 0; c < a.C; ++c) {
      if (ign_in && c == a.ignore) continue;
      unsigned k = LV_KEY_INVALID;
      // probas are softmax outputs in [0, 1] (precondition of losses.lovasz_softmax); an error above 1 would wrap in the 27-bit key: clamped
      if (valid) k = lv_key(fminf(fabsf((c == t ? 1.f : 0.f) - a.x[off + c * a.sc]), 1.f));
      const size_t pos = (size_t)seg_of(c, a.ignore, a.C) * (size_t)n_vox + (size_t)u;      // walk order: coalesced (ties sort in it)
      w.keys_in[pos] = ((unsigned)c << LV_KEY_BITS) | k;
      w.vals_in[pos] = (unsigned)v;
    }
  }
  __syncthreads();
  if (threadIdx.x < a.C && scnt[threadIdx.x]) atomicAdd(&w.cnt[threadIdx.x], scnt[threadIdx.x]);
  if (threadIdx.x == 32 && scnt[32]) atomicAdd(&w.cnt[a.C], scnt[32]);
}

// block (b, c): foreground count of sorted elements [b*1024, b*1024+1024) of class c's segment
__global__ void __launch_bounds__(256) k_lovasz_blocksum(VoxArgs a, long long n_vox, LovaszWs w, int nb) {
  __shared__ float red[4];
  const int c = blockIdx.y;
  if ((a.ignore >= 0 && a.ignore < a.C && c == a.ignore) || w.cnt[c] == 0) return;
  const unsigned nvalid = w.cnt[a.C];
  const size_t base = (size_t)seg_of(c, a.ignore, a.C) * (size_t)n_vox;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned i = blockIdx.x * LV_BLOCK + threadIdx.x * 4 + j;
    if (i < nvalid) s += a.target[w.vals_out[base + i]] == c ? 1.f : 0.f;
  }
  const float bs = block_sum(s, red);
  if (threadIdx.x == 0) w.bsum[(size_t)c * nb + blockIdx.x] = (unsigned)bs;
}

// one block per class: exclusive prefix over its block sums (nb is ~1 k: a serial pass by 256 threads in chunks)
__global__ void __launch_bounds__(256) k_lovasz_scan(VoxArgs a, LovaszWs w, int nb) {
  __shared__ unsigned part[256];
  const int c = blockIdx.x;
  if (w.cnt[c] == 0) return;
  unsigned* bs = w.bsum + (size_t)c * nb;
  const int per = (nb + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(lo + per, nb);
  unsigned s = 0;
  for (int i = lo; i < hi; ++i) s += bs[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int i = 0; i < 256; ++i) { const unsigned t = part[i]; part[i] = run; run += t; }
  }
  __syncthreads();
  unsigned run = part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { const unsigned t = bs[i]; bs[i] = run; run += t; }
}

// Jaccard gradient of every sorted element, the class loss, and d(loss_c)/d(p) scattered back to the voxel
__global__ void __launch_bounds__(256) k_lovasz_dot(VoxArgs a, long long n_vox, LovaszWs w, int nb, float* __restrict__ dprob) {
  __shared__ float red[4];
  __shared__ unsigned wsum[4];
  const int c = blockIdx.y;
  if ((a.ignore >= 0 && a.ignore < a.C && c == a.ignore) || w.cnt[c] == 0) return;
  const unsigned nvalid = w.cnt[a.C];
  const float gts = (float)w.cnt[c];
  const size_t base = (size_t)seg_of(c, a.ignore, a.C) * (size_t)n_vox;
  const unsigned i0 = blockIdx.x * LV_BLOCK + threadIdx.x * 4;
  unsigned fg[4], vv[4];
  float err[4];
  bool nz[4];                                    // error > 0 (key below the clamp of k_lovasz_keys)
  unsigned local = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned i = i0 + j;
    fg[j] = 0; vv[j] = 0; err[j] = 0.f; nz[j] = false;
    if (i < nvalid) {
      vv[j] = w.vals_out[base + i];
      fg[j] = a.target[vv[j]] == c ? 1u : 0u;
      const unsigned kb = w.keys_out[base + i] & LV_KEY_INVALID;
      nz[j] = kb != LV_KEY_ZERO;
      err[j] = nz[j] ? lv_key_error(kb) : 0.f;
    }
    local += fg[j];
  }
  // exclusive prefix of `local` over the 256 threads: wave scan + wave totals
  unsigned incl = local;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned before = w.bsum[(size_t)c * nb + blockIdx.x];
  for (int q = 0; q < wave; ++q) before += wsum[q];
  unsigned cf = before + incl - local;           // foreground count strictly before this thread's first element
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned i = i0 + j;
    if (i < nvalid) {
      const float cfp = (float)cf;                // inclusive count up to i-1
      cf += fg[j];
      const float cfi = (float)cf;
      // lovasz_grad: intersection = gts - cumsum(fg), union = gts + cumsum(1 - fg), jaccard = 1 - inter / union
      const float jac = 1.f - (gts - cfi) / (gts + ((float)(i + 1) - cfi));
      const float jprev = i ? 1.f - (gts - cfp) / (gts + ((float)i - cfp)) : 0.f;
      const float g = jac - jprev;
      part += err[j] * g;
      if (dprob) {
        int xh, yw;
        const long long off = vox_decode(a, (long long)vv[j], xh, yw);
        // d|fg - p|/dp = -sign(fg - p), and p in [0, 1]: the sign is that of fg - 1/2 unless the error is zero -- no second
        // (random) read of the probability
        dprob[off + c * a.sc] = nz[j] ? (fg[j] ? -g : g) : 0.f;
      }
    }
  }
  const float bs = block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(&w.lossc[c], (double)bs);
}

__global__ void k_lovasz_finish(LovaszWs w, int C, float* __restrict__ loss, float* __restrict__ inv_present) {
  double s = 0.0;
  int n = 0;
  for (int c = 0; c < C; ++c)
    if (w.cnt[c]) { s += w.lossc[c]; ++n; }
  loss[0] = n ? (float)(s / n) : 0.f;
  inv_present[0] = n ? 1.f / (float)n : 0.f;
}

__global__ void k_zero_bytes(unsigned* __restrict__ p, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int lovasz_layout(long long n_vox, int C, int ignore, char* base, size_t have, LovaszWs& w, size_t& need, int& nb, size_t& zero_off,
                  size_t& zero_bytes) {
  const int nseg = C - ((ignore >= 0 && ignore < C) ? 1 : 0);
  const size_t n = (size_t)nseg * (size_t)n_vox;
  nb = (int)((n_vox + LV_BLOCK - 1) / LV_BLOCK);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
  const size_t o_ki = take(n * 4), o_ko = take(n * 4), o_vi = take(n * 4), o_vo = take(n * 4);
  zero_off = off;
  const size_t o_cnt = take((size_t)(C + 1) * 4), o_ls = take((size_t)C * 8);
  zero_bytes = off - zero_off;
  const size_t o_bs = take((size_t)C * nb * 4);
  size_t temp = 0;
  unsigned bits = LV_KEY_BITS;
  for (int c = C - 1; c > 0; c >>= 1) ++bits;
  if (rocprim::radix_sort_pairs(nullptr, temp, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                (unsigned*)nullptr, n, 0u, bits, (hipStream_t)0) != hipSuccess)
    return PW_EHIP;
  const size_t o_tmp = take(temp);
  need = off;
  if (base && have >= need) {
    w.keys_in = (unsigned*)(base + o_ki); w.keys_out = (unsigned*)(base + o_ko);
    w.vals_in = (unsigned*)(base + o_vi); w.vals_out = (unsigned*)(base + o_vo);
    w.cnt = (unsigned*)(base + o_cnt); w.lossc = (double*)(base + o_ls); w.bsum = (unsigned*)(base + o_bs);
    w.temp = base + o_tmp; w.temp_bytes = temp;
  }
  return PW_OK;
}
}  // namespace

#define PW_VOX_ARGS(a, ptr)                                                                                          \
  VoxArgs a = {ptr, target, cam_mask, class_weights, B, C, X, Y, Z, sb, sc, sx, sy, sz, ignore_index, vox_walk_order(sx, sy, sz)}; \
  const long long n_vox = (long long)B * X * Y * Z;                                                                  \
  PW_CHECK_ARG(ptr && target && B > 0 && C > 0 && X > 0 && Y > 0 && Z > 0, "voxel loss: bad arguments");            \
  const unsigned grid = (unsigned)((n_vox + 255) / 256 < 2048 ? (n_vox + 255) / 256 : 2048)

static FocalPar focal_par(int X, int Y, float gamma, float alpha) {
  FocalPar f;
  f.gamma = gamma; f.alpha = alpha; f.hx = (float)(X / 2.0); f.hy = (float)(Y / 2.0);
  f.inv_cmax = 1.f / sqrtf(f.hx * f.hx + f.hy * f.hy);
  return f;
}

PW_API int pw_focal_loss_stats(const float* logits, const uint8_t* target, const uint8_t* cam_mask, const float* class_weights,
                               int B, int C, int X, int Y, int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy,
                               int64_t sz, int ignore_index, float gamma, float alpha, double* stats, void* stream) {
  PW_VOX_ARGS(a, logits);
  PW_CHECK_ARG(stats, "pw_focal_loss_stats: null stats");
  hipLaunchKernelGGL(k_focal_stats, dim3(grid), dim3(256), 0, pw_stream(stream), a, focal_par(X, Y, gamma, alpha), n_vox, stats);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_focal_loss_finish(const double* stats, float loss_weight, float* loss, void* stream) {
  PW_CHECK_ARG(stats && loss, "pw_focal_loss_finish: null pointer");
  hipLaunchKernelGGL(k_focal_finish, dim3(1), dim3(1), 0, pw_stream(stream), stats, loss_weight, loss);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API int pw_focal_loss_grad(const float* logits, const uint8_t* target, const uint8_t* cam_mask, const float* class_weights,
                              int B, int C, int X, int Y, int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy,
                              int64_t sz, int ignore_index, float gamma, float alpha, const double* stats, float loss_weight,
                              const float* grad_loss, float* grad_logits, void* stream) {
  PW_VOX_ARGS(a, logits);
  PW_CHECK_ARG(stats && grad_loss && grad_logits, "pw_focal_loss_grad: null pointer");
  hipLaunchKernelGGL(k_focal_grad, dim3(grid), dim3(256), 0, pw_stream(stream), a, focal_par(X, Y, gamma, alpha), n_vox, stats,
                     loss_weight, grad_loss, grad_logits);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

PW_API size_t pw_lovasz_workspace_bytes(int64_t n_vox, int n_cls, int ignore_index) {
  LovaszWs w;
  size_t need = 0, zo, zb;
  int nb;
  if (n_vox <= 0 || n_cls <= 0 || n_cls > 32) return 0;
  if (lovasz_layout(n_vox, n_cls, ignore_index, nullptr, 0, w, need, nb, zo, zb) != PW_OK) return 0;
  return need;
}

PW_API int pw_lovasz_softmax(const float* probas, const uint8_t* target, const uint8_t* cam_mask, int B, int C, int X, int Y,
                             int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy, int64_t sz, int ignore_index,
                             void* workspace, size_t workspace_bytes, float* loss, float* inv_present, float* dprob,
                             void* stream) {
  const float* class_weights = nullptr;
  PW_VOX_ARGS(a, probas);
  PW_CHECK_ARG(C <= 32 && n_vox < (1ll << 31), "pw_lovasz_softmax: at most 32 classes and 2^31 voxels");
  PW_CHECK_ARG(workspace && loss && inv_present, "pw_lovasz_softmax: null pointer");
  LovaszWs w;
  size_t need = 0, zo = 0, zb = 0;
  int nb = 0;
  const int rc = lovasz_layout(n_vox, C, ignore_index, (char*)workspace, workspace_bytes, w, need, nb, zo, zb);
  if (rc != PW_OK) { pw_set_error("pw_lovasz_softmax: rocprim workspace query failed"); return rc; }
  PW_CHECK_ARG(workspace_bytes >= need, "pw_lovasz_softmax: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = pw_stream(stream);
  hipLaunchKernelGGL(k_zero_bytes, dim3(1), dim3(256), 0, st, (unsigned*)((char*)workspace + zo), zb / 4);
  hipLaunchKernelGGL(k_lovasz_keys, dim3(grid), dim3(256), 0, st, a, n_vox, w);
  const int nseg = C - ((ignore_index >= 0 && ignore_index < C) ? 1 : 0);
  const size_t n = (size_t)nseg * (size_t)n_vox;
  unsigned bits = LV_KEY_BITS;
  for (int c = C - 1; c > 0; c >>= 1) ++bits;
  size_t tb = w.temp_bytes;
  if (rocprim::radix_sort_pairs(w.temp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, n, 0u, bits, st) != hipSuccess) {
    pw_set_error("pw_lovasz_softmax: radix sort failed");
    return PW_EHIP;
  }
  hipLaunchKernelGGL(k_lovasz_blocksum, dim3((unsigned)nb, (unsigned)C), dim3(256), 0, st, a, n_vox, w, nb);
  hipLaunchKernelGGL(k_lovasz_scan, dim3((unsigned)C), dim3(256), 0, st, a, w, nb);
  hipLaunchKernelGGL(k_lovasz_dot, dim3((unsigned)nb, (unsigned)C), dim3(256), 0, st, a, n_vox, w, nb, dprob);
  hipLaunchKernelGGL(k_lovasz_finish, dim3(1), dim3(1), 0, st, w, C, loss, inv_present);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
