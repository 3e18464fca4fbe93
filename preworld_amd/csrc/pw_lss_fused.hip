// One frame's lift + voxel pooling in five launches, for C = 32 context channels (the PreWorld configs' numC_Trans):
// camera matrices -> voxel id per frustum point -> per-voxel point lists -> pooled (B,Z,Y,X,C) grid, fp32 or split-fp16.
// Replaces, like pw_lss.hip (whose sort-based path stays for the reference-ABI rank tensors, training and `accelerate`):
//   mmdet3d/models/necks/view_transformer.py:114-153  get_lidar_coor
//   mmdet3d/models/necks/view_transformer.py:203-261  voxel_pooling_prepare_v2
//   mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48  bev_pool_v2 forward
//
// Why another form.  The sort-based path spends 9 launches per frame: zero, histogram, two scan passes, scatter, in-segment rank
// sort, then a voxel-driven sweep that follows seg_start -> order -> depth / feat for every one of the 640 000 voxels (46 % of them
// empty, 330 000 holding 1..8 points).  Here a voxel owns EIGHT ID SLOTS (32 bytes at slots[v * 8]): the returning atomic that counts
// a point's voxel also says which slot it gets, so for 95 % of the non-empty voxels the histogram IS the point list -- no scan, no
// scatter, no sort kernel.  The arrival order in the slots is whatever the atomics gave; the pooling sweep sorts the <= 8 ids of a
// voxel inside its 8-lane group (8 shuffles + one ds_permute), so sums still run in ascending point order and stay bit-identical
// to the oracle.  Voxels with more than 8 points ("heavy": 2.5 % of the voxels, 36 % of the points) register themselves when their
// ninth point arrives; their points beyond the slots go to an overflow list and are scattered into per-voxel storage by two small
// kernels; one wave per heavy voxel sorts up to 64 ids in registers, a block sorts longer ones through LDS.
//
//   k_lss_prologue      zero the counters; camera matrices                                                  (2.6 MB)
//   k_lss_index_slots   geometry, run-compressed returning atomics, slot writes, arrival rank per point (4 bytes)
//   k_lss_heavy_alloc   storage for the heavy voxels (wave scan + one atomic per wave), lists by size class
//   k_lss_ovf_scatter   points of rank >= 8 (their voxel recomputed from the geometry) -> tmp[start(voxel) + arrival rank]
//   k_lss_pool_slots    light voxels: 64 counters per wave, non-empty ones compacted in the wave, eight at a time; the EMPTY voxels
//                       of the wave's 64 get their zero rows here, so the grid is written exactly once (round 4: the zero fill of
//                       all 82 MB used to ride in k_lss_index_slots and 44 MB of it was overwritten); heavy voxels: first blocks
// HBM-bound integer/byte work: no MFMA.  Compiled with -ffp-contract=off (see pw_lss_common.h).
#include "pw_lss_common.h"

namespace {

constexpr int LS = 8;                 // id slots per voxel
constexpr int LPV = 8;                // lanes per voxel (float4 of channels per lane): C = 32
constexpr int GROUPS = 64 / LPV;      // voxels a wave works on at a time
constexpr int HEAVY_WAVE_MAX = 64;    // heavy voxels up to this size: one wave, ids sorted in registers
constexpr int SORT_LDS_IDS = 4096;    // ids per LDS pass of the block sort (16 KB)
constexpr int C_BLOCKS = 512;         // blocks of k_lss_pool_heavy that take voxels above HEAVY_WAVE_MAX, one at a time
constexpr int B_BLOCKS = 1024;        // blocks (x 8 half waves) of it that take the other heavy voxels
// cursors, zeroed with the counters: count[n_vox + ...]
enum { CUR_PTS = 0, CUR_B = 1, CUR_C = 2, CUR_WORDS = 8 };

struct FeatIdx {
  int DHW, HW;
  float inv_dhw, inv_hw;
};

// frustum point id -> feat pixel (view_transformer.py:219-224: ranks_feat drops the depth axis): (id / DHW) * HW + id % HW.  Both
// quotients are small (camera index, depth bin), so a float estimate is off by at most one; exact after one correction step.
__device__ __forceinline__ int feat_index(int id, const FeatIdx& fi) {
  int cam = (int)((float)id * fi.inv_dhw);
  int p = id - cam * fi.DHW;
  if (p < 0) { --cam; p += fi.DHW; } else if (p >= fi.DHW) { ++cam; p -= fi.DHW; }
  const int d = (int)((float)p * fi.inv_hw);
  int hw = p - d * fi.HW;
  if (hw < 0) hw += fi.HW; else if (hw >= fi.HW) hw -= fi.HW;
  return cam * fi.HW + hw;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

}  // namespace

__global__ void __launch_bounds__(256)
k_lss_prologue(int BN, const float* __restrict__ s2e, const float* __restrict__ K, const float* __restrict__ pr,
               float* __restrict__ ipr, float* __restrict__ comb, float* __restrict__ tr, int4* __restrict__ zero, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) zero[i] = make_int4(0, 0, 0, 0);
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < BN; c += blockDim.x) lss_camera_matrix_one(c, s2e, K, pr, ipr, comb, tr);
}

// Returning atomics execute below the per-XCD L2 and sustain ~19 G ops/s, so the kernel is bound by how many it
// issues: consecutive lanes are consecutive pixels of one depth bin and fall into the same voxel in runs; a run inside a wave is
// served by ONE atomicAdd of the run length, issued by its first lane (520 k atomics for 880 k kept points at the C3 shape).
__global__ void __launch_bounds__(256)
k_lss_index_slots(int N, int64_t DHW, int64_t total, const float* __restrict__ frustum, const float* __restrict__ ipr,
                  const float* __restrict__ ptr, const float* __restrict__ comb, const float* __restrict__ trn,
                  const float* __restrict__ bda, GridParams gp, int32_t* __restrict__ count,
                  int32_t* __restrict__ slots, int32_t* __restrict__ rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int k = i < total ? lss_voxel_of_point(i, N, DHW, frustum, ipr, ptr, comb, trn, bda, gp, nullptr) : -1;
  const int kprev = __shfl_up(k, 1, 64);
  const unsigned long long heads = __ballot(lane == 0 || k != kprev);
  const unsigned long long upto = heads & (~0ull >> (63 - lane));           // heads at lanes <= mine
  const int start = 63 - __builtin_clzll(upto);
  const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
  const int runlen = (above ? __builtin_ctzll(above) : 64) - start;
  const bool head = k >= 0 && lane == start;
  int base = 0;
  if (head) base = atomicAdd(&count[k], runlen);
  base = __shfl(base, start, 64);
  const int r = base + lane - start;                                        // arrival rank of this point in its voxel
  if (k >= 0 && r < LS) slots[(int64_t)k * LS + r] = (int32_t)i;
  // points beyond the slots are placed by k_lss_ovf_scatter once their voxel has storage.  (An overflow LIST appended to here
  // -- one atomic per wave on a cursor -- cost 80 us: same-address atomics serialise.  4 bytes per point, coalesced, cost nothing;
  // the voxel is recomputed there instead of being carried: 6 MB less written and read.)
  if (i < total) rank[i] = k >= 0 ? r : -1;
}

// storage and size class of the heavy voxels: a sweep over the counters, 2 048 voxels per block, ONE atomic per block and cursor
// (the order of the lists follows the order the blocks arrive in; nothing downstream depends on it)
constexpr int HA_ITEMS = 8;
__global__ void __launch_bounds__(256)
k_lss_heavy_alloc(const int32_t* __restrict__ count, int64_t n_vox, int32_t* __restrict__ cur, int32_t* __restrict__ hstart,
                  int4* __restrict__ list_b, int4* __restrict__ list_c) {
  __shared__ int wsum[3][4];
  __shared__ int bbase[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * HA_ITEMS;
  int c[HA_ITEMS];
  int pts = 0, nb = 0, nc = 0;
#pragma unroll
  for (int u = 0; u < HA_ITEMS; ++u) {
    c[u] = v0 + u < n_vox ? count[v0 + u] : 0;
    if (c[u] > LS) {
      pts += c[u];
      if (c[u] > HEAVY_WAVE_MAX) ++nc; else ++nb;
    }
  }
  const int ip = wave_incl_scan(pts, lane), ib = wave_incl_scan(nb, lane), ic = wave_incl_scan(nc, lane);
  if (lane == 63) { wsum[0][wave] = ip; wsum[1][wave] = ib; wsum[2][wave] = ic; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int t = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
    bbase[threadIdx.x] = t ? atomicAdd(cur + (threadIdx.x == 0 ? CUR_PTS : threadIdx.x == 1 ? CUR_B : CUR_C), t) : 0;
  }
  __syncthreads();
  int hs = bbase[0] + ip - pts, pb = bbase[1] + ib - nb, pc = bbase[2] + ic - nc;
  for (int w = 0; w < wave; ++w) { hs += wsum[0][w]; pb += wsum[1][w]; pc += wsum[2][w]; }
#pragma unroll
  for (int u = 0; u < HA_ITEMS; ++u)
    if (c[u] > LS) {
      hstart[v0 + u] = hs;
      const int4 e = make_int4((int)(v0 + u), c[u], hs, 0);
      if (c[u] > HEAVY_WAVE_MAX) list_c[pc++] = e; else list_b[pb++] = e;
      hs += c[u];
    }
}

__global__ void __launch_bounds__(256)
k_lss_ovf_scatter(int N, int64_t DHW, int64_t total, const float* __restrict__ frustum, const float* __restrict__ ipr,
                  const float* __restrict__ ptr, const float* __restrict__ comb, const float* __restrict__ trn,
                  const float* __restrict__ bda, GridParams gp, const int32_t* __restrict__ rank, const int32_t* __restrict__ hstart,
                  int32_t* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r = rank[i];
  if (r < LS) return;                                              // outside the grid (-1) or in a slot
  const int k = lss_voxel_of_point(i, N, DHW, frustum, ipr, ptr, comb, trn, bda, gp, nullptr);      // same code, same bits as in the index kernel
  tmp[hstart[k] + r] = (int32_t)i;                                 // positions start .. start + 7 stay unused: those ids sit in the slots
}

namespace {

// channel-per-lane sums of a half wave -> the float4-per-lane row layout of pool_store (lanes 0..7 of the half)
__device__ __forceinline__ float4 half_to_quads(float acc, int lane) {
  const int src = (lane & 32) + 4 * (lane & 7);
  float4 f;
  f.x = __shfl(acc, src, 64);
  f.y = __shfl(acc, src + 1, 64);
  f.z = __shfl(acc, src + 2, 64);
  f.w = __shfl(acc, src + 3, 64);
  return f;
}

constexpr int HC_ROWS = 128;          // rows per LDS chunk of the block-pooled voxels (16 KB)

}  // namespace

// The heavy voxels (more than 8 points), sorted and summed.  The sum of a voxel is sequential in its points -- that is what
// makes it bit-exact -- so a voxel costs (points) x (latency of one step) however many lanes work on it; the two classes below
// keep that step short: the rows a voxel needs are requested all at once, the sequential part reads registers or LDS.
//   blocks [0, C_BLOCKS): voxels above HEAVY_WAVE_MAX points, one block each.  The block ranks the ids through LDS (rank = number
//     of smaller ids: they are distinct; segments beyond SORT_LDS_IDS take LDS-sized passes), leaves (feat pixel, depth) in
//     ascending id order, then streams the rows through LDS: all 256 threads gather 128 rows (x depth) per chunk, the next
//     chunk's rows are in flight while the first half wave adds this chunk's, lane c = channel c.
//   other blocks: voxels of 9 .. 64 points, HALF a wave each: lane c of the half ranks the c-th and (c + 32)-th arrival with 64
//     shuffles at most, the half's LDS strip then lists (pixel, depth) in order, all rows are requested, lane c adds channel c.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_lss_pool_slots(const float* __restrict__ depth, const float4* __restrict__ feat, const int32_t* __restrict__ count,
                 const int32_t* __restrict__ slots, int64_t n_vox, const int32_t* __restrict__ cur, const int4* __restrict__ list_b, const int4* __restrict__ list_c,
                 const int32_t* __restrict__ tmp, int32_t* __restrict__ sorted_pf, float* __restrict__ sorted_d, FeatIdx fi,
                 float4* __restrict__ out, int out_h2, int* __restrict__ out_rng) {
#ifdef PW_X_SKIP_LSS            // ablation builds only
  return;
#endif
  extern __shared__ __attribute__((aligned(16))) int32_t ids[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPV;
  const int e_out = out_h2 ? rng_exp(out_rng) : 0;
  const float omul = rng_pow2(-e_out);
  unsigned amax = 0u;
  if ((int)blockIdx.x < C_BLOCKS) {
    const int nc = cur[CUR_C];
    for (int li = blockIdx.x; li < nc; li += C_BLOCKS) {
      const int4 e = list_c[li];
      const int64_t k = e.x;
      const int n = e.y, hs = e.z;
      if (n <= SORT_LDS_IDS) {                             // the usual case: one staging pass
        const int m16 = (n + 15) & ~15;                     // padded with INT_MAX: never "smaller"
        __syncthreads();
        for (int j = threadIdx.x; j < m16; j += blockDim.x) ids[j] = j < n ? (j < LS ? slots[k * LS + j] : tmp[hs + j]) : 0x7fffffff;
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
          const int id = ids[t];
          const float d = depth[id];
          const int r = count_smaller_lds(ids, m16, id);
          sorted_pf[hs + r] = feat_index(id, fi);
          sorted_d[hs + r] = d;
        }
      } else {
        for (int t0 = 0; t0 < n; t0 += blockDim.x) {
          const int t = t0 + threadIdx.x;
          const int id = t < n ? (t < LS ? slots[k * LS + t] : tmp[hs + t]) : 0;
          const float d = t < n ? depth[id] : 0.f;
          int r = 0;
          for (int p0 = 0; p0 < n; p0 += SORT_LDS_IDS) {
            const int m = min(SORT_LDS_IDS, n - p0);
            const int m16 = (m + 15) & ~15;
            __syncthreads();
            for (int j = threadIdx.x; j < m16; j += blockDim.x) {
              const int q = p0 + j;
              ids[j] = j < m ? (q < LS ? slots[k * LS + q] : tmp[hs + q]) : 0x7fffffff;
            }
            __syncthreads();
            if (t < n) r += count_smaller_lds(ids, m16, id);
          }
          if (t < n) {
            sorted_pf[hs + r] = feat_index(id, fi);
            sorted_d[hs + r] = d;
          }
        }
      }
      // the sorted pairs are read back by this block only: a workgroup-scope fence.  (A device-scope __threadfence() here writes
      // back the whole L2 of the XCD -- 82 MB of zero fill sit in it -- and cost 40 us.)
      __threadfence_block();
      __syncthreads();
      // ---- rows through LDS, HC_ROWS at a time: thread (rs, q) holds quad q of rows rs, rs + 32, rs + 64, rs + 96 of a chunk
      const int rs = threadIdx.x >> 3, q = threadIdx.x & 7;
      float4* rows4 = reinterpret_cast<float4*>(ids);
      const float* rowsf = reinterpret_cast<const float*>(ids);
      int pfr[4];
      float dr[4], dc[4];
      float4 reg[4];
      auto load_pd = [&](int base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = base + u * 32 + rs;
          pfr[u] = j < n ? sorted_pf[hs + j] : -1;
          dr[u] = j < n ? sorted_d[hs + j] : 0.f;
        }
      };
      auto load_rows = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          reg[u] = pfr[u] >= 0 ? feat[(int64_t)pfr[u] * LPV + q] : make_float4(0.f, 0.f, 0.f, 0.f);
          dc[u] = dr[u];
        }
      };
      load_pd(0);
      load_rows();
      if (HC_ROWS < n) load_pd(HC_ROWS);
      float acc1 = 0.f;
      for (int base = 0; base < n; base += HC_ROWS) {
        __syncthreads();                                   // the previous chunk has been added (and the sort is done with `ids`)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          rows4[(u * 32 + rs) * LPV + q] = make_float4(reg[u].x * dc[u], reg[u].y * dc[u], reg[u].z * dc[u], reg[u].w * dc[u]);
        __syncthreads();
        if (base + HC_ROWS < n) {
          load_rows();
          if (base + 2 * HC_ROWS < n) load_pd(base + 2 * HC_ROWS);
        }
        if (threadIdx.x < 32) {
          const int m = min(HC_ROWS, n - base);
          for (int j = 0; j < m; j += 16) {                // 16 LDS reads in flight, then 16 dependent adds
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = rowsf[(j + u) * 32 + threadIdx.x];     // rows past m: zeros or stale, not added
#pragma unroll
            for (int u = 0; u < 16; ++u)
              if (j + u < m) acc1 = acc1 + v[u];
          }
        }
      }
      if (wave == 0) {
        const float4 acc = half_to_quads(acc1, lane);
        if (lane < LPV) pool_store<LPV>(out, k, sub, acc, out_h2, omul, amax);
      }
    }
  } else if ((int)blockIdx.x < C_BLOCKS + B_BLOCKS) {
    const int nb = cur[CUR_B];
    const int hb = lane & 32, c = lane & 31;
    int32_t* lpf = ids + (wave * 2 + (lane >> 5)) * 128;          // 64 pixels + 64 depths per half: 4 KB of the block's LDS
    float* ld = reinterpret_cast<float*>(lpf + 64);
    const float* __restrict__ featf = reinterpret_cast<const float*>(feat);
    for (int t = ((int)blockIdx.x - C_BLOCKS) * 4 + wave; 2 * t < nb; t += B_BLOCKS * 4) {
      const int li = 2 * t + (lane >> 5);
      const bool on = li < nb;
      const int4 e = on ? list_b[li] : make_int4(0, 0, 0, 0);
      const int64_t k = e.x;
      const int n = e.y, hs = e.z;
      const int nmax = max(__shfl(n, 0, 64), __shfl(n, 32, 64));
      const int a = c < n ? (c < LS ? slots[k * LS + c] : tmp[hs + c]) : 0x7fffffff;
      const int b = c + 32 < n ? tmp[hs + c + 32] : 0x7fffffff;
      const float da = c < n ? depth[a] : 0.f, db = c + 32 < n ? depth[b] : 0.f;
      int ra = 0, rb = 0;
      if (nmax > 32) {
        for (int j = 0; j < 32; ++j) {
          const int oa = __shfl(a, hb + j, 64), ob = __shfl(b, hb + j, 64);
          ra += (oa < a) + (ob < a);
          rb += (oa < b) + (ob < b);
        }
      } else {
        for (int j = 0; j < nmax; ++j) ra += __shfl(a, hb + j, 64) < a;
      }
      if (c < n) { lpf[ra] = feat_index(a, fi); ld[ra] = da; }
      if (c + 32 < n) { lpf[rb] = feat_index(b, fi); ld[rb] = db; }
      float acc1 = 0.f;
#pragma unroll
      for (int h = 0; h < HEAVY_WAVE_MAX; h += 32)
        if (h < nmax) {                                    // wave-uniform: rows of 32 points requested together
          float x[32];
#pragma unroll
          for (int p = 0; p < 32; ++p) {
            x[p] = 0.f;
            if (h + p < nmax) {
              const int pf = h + p < n ? lpf[h + p] : 0;
              if (h + p < n) x[p] = featf[(unsigned)(pf * 32 + c)];
            }
          }
#pragma unroll
          for (int p = 0; p < 32; ++p)
            if (h + p < nmax) {
              const float d = h + p < n ? ld[h + p] : 0.f;
              if (h + p < n) acc1 = acc1 + x[p] * d;
            }
        }
      const float4 acc = half_to_quads(acc1, lane);
      if (on && c < LPV) pool_store<LPV>(out, k, sub, acc, out_h2, omul, amax);
    }
  } else {
    // ---- the light voxels (1 .. 8 points).  A wave reads the counters of 64 consecutive voxels, moves the (voxel, count) pairs
    // of the non-empty light ones to its first lanes (ds_permute over a full permutation), and its eight lane groups take eight of
    // them at a time: the group sorts the voxel's ids (8 shuffles + one ds_permute), gathers the rows, adds them in order.  The
    // EMPTY voxels among the 64 get their zero row here (8 rows per store instruction, one per lane group; all-zero bytes are zero
    // in both output formats), so every row of the grid is written by exactly one kernel, once.
    const int grp = lane / LPV, gbase = lane - sub;
    const int64_t wid = (int64_t)((int)blockIdx.x - C_BLOCKS - B_BLOCKS) * 4 + wave;
    const int64_t nw = (int64_t)((int)gridDim.x - C_BLOCKS - B_BLOCKS) * 4;
    for (int64_t v0 = wid * 64; v0 < n_vox; v0 += nw * 64) {
      const int c = v0 + lane < n_vox ? count[v0 + lane] : -1;
      {
        const unsigned long long empty = __ballot(c == 0);
        if (empty) {
#pragma unroll
          for (int j = 0; j < GROUPS; ++j)
            if ((empty >> (j * GROUPS + grp)) & 1ull) out[(v0 + j * GROUPS + grp) * LPV + sub] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      const bool light = c > 0 && c <= LS;
      const unsigned long long m = __ballot(light);
      const int nl = __builtin_popcountll(m);
      const int below = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      const int dst = light ? below : nl + (lane - below);
      const int packed = __builtin_amdgcn_ds_permute(dst << 2, lane | (c << 8));
      for (int it = 0; it * GROUPS < nl; ++it) {
        const int q = it * GROUPS + grp;
        const int pk = __shfl(packed, q, 64);
        const bool on = q < nl;
        const int cnt = on ? pk >> 8 : 0;
        const int64_t v = v0 + (pk & 255);
        const int id = sub < cnt ? slots[v * LS + sub] : 0x7ffffff8 + sub;      // padding: distinct, above every id
        int r = 0;
#pragma unroll
        for (int u = 0; u < LPV; ++u) r += __shfl(id, gbase + u, 64) < id;
        const int sid = __builtin_amdgcn_ds_permute((gbase + r) << 2, id);      // lane sub: the sub-th smallest id of the voxel
        int my_pf = 0;
        float my_d = 0.f;
        if (sub < cnt) {
          my_pf = feat_index(sid, fi);
          my_d = depth[sid];
        }
        // most voxels hold 1-3 points: stop at the largest count among the wave's eight voxels (wave-uniform test)
        float4 f[LPV];
#pragma unroll
        for (int u = 0; u < LPV; ++u) f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < LPV; ++u) {
          if (__ballot(u < cnt) == 0ull) break;
          const int pf = __shfl(my_pf, gbase + u, 64);
          if (u < cnt) f[u] = feat[(int64_t)pf * LPV + sub];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < LPV; ++u) {
          if (__ballot(u < cnt) == 0ull) break;
          const float d = __shfl(my_d, gbase + u, 64);
          if (u < cnt) fma4_nc(acc, f[u], d);
        }
        if (on) pool_store<LPV>(out, v, sub, acc, out_h2, omul, amax);
      }
    }
  }
  if (out_h2) rng_note(out_rng, amax, e_out);
}

namespace {
struct FusedWs {
  int32_t* count;      // n_vox counters + CUR_WORDS cursors (zeroed by the prologue)
  size_t zero_bytes;
  float *ipr, *comb, *tr;
  int32_t* slots;
  int32_t* rank;
  int32_t* hstart;
  int4 *list_b, *list_c;
  int32_t *tmp, *sorted_pf;
  float* sorted_d;
  size_t bytes;
};

FusedWs fused_ws(char* base, int64_t n, int64_t n_vox, int BN) {
  FusedWs w;
  char* p = base;
  auto take = [&](size_t bytes) { char* q = p; p += pw_align_up(bytes, 256); return q; };
  w.zero_bytes = pw_align_up((size_t)(n_vox + CUR_WORDS) * 4, 256);
  w.count = (int32_t*)take(w.zero_bytes);
  w.ipr = (float*)take((size_t)BN * 9 * 4);
  w.comb = (float*)take((size_t)BN * 9 * 4);
  w.tr = (float*)take((size_t)BN * 3 * 4);
  w.slots = (int32_t*)take((size_t)n_vox * LS * 4);
  w.rank = (int32_t*)take((size_t)n * 4);
  w.hstart = (int32_t*)take((size_t)n_vox * 4);
  w.list_b = (int4*)take((size_t)(n / (LS + 1) + 1) * 16);
  w.list_c = (int4*)take((size_t)(n / (HEAVY_WAVE_MAX + 1) + 1) * 16);
  w.tmp = (int32_t*)take((size_t)n * 4);
  w.sorted_pf = (int32_t*)take((size_t)n * 4);
  w.sorted_d = (float*)take((size_t)n * 4);
  w.bytes = (size_t)(p - base);
  return w;
}
}  // namespace

PW_API size_t pw_lss_lift_pool_workspace_bytes(int64_t n_points, int64_t n_voxels, int BN) {
  return fused_ws(nullptr, n_points, n_voxels, BN).bytes;
}

PW_API int pw_lss_lift_pool(int B, int N, int D, int H, int W, const float* frustum, const float* sensor2ego,
                            const float* cam2imgs, const float* post_rots, const float* post_trans, const float* bda,
                            const float* lower3_host, const float* interval3_host, int gx, int gy, int gz, const float* depth,
                            const float* feat, int c, void* workspace, size_t workspace_bytes, float* out, int out_h2,
                            int32_t* out_rng, void* stream) {
  PW_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && W > 0 && gx > 0 && gy > 0 && gz > 0, "pw_lss_lift_pool: bad shape");
  PW_CHECK_ARG(frustum && sensor2ego && cam2imgs && post_rots && post_trans && bda && lower3_host && interval3_host && depth &&
                   feat && workspace && out,
               "pw_lss_lift_pool: null pointer");
  PW_CHECK_ARG(c == 4 * LPV, "pw_lss_lift_pool: C must be 32 (use pw_segment_sort + pw_bev_pool_dense for other widths)");
  PW_CHECK_ARG(((uintptr_t)feat & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               "pw_lss_lift_pool: feat / out must be 16-byte aligned, the workspace 256-byte aligned");
  const int64_t n_vox = (int64_t)B * gx * gy * gz;
  const int64_t DHW = (int64_t)D * H * W, total = DHW * B * N;
  PW_CHECK_ARG(n_vox < (int64_t)1 << 31 && total < (int64_t)1 << 31, "pw_lss_lift_pool: sizes must fit int32");
  PW_CHECK_ARG(DHW < (1 << 24) && (int64_t)B * N < (1 << 20), "pw_lss_lift_pool: D*H*W must be below 2^24, B*N below 2^20");
  const FusedWs w = fused_ws((char*)workspace, total, n_vox, B * N);
  if (workspace_bytes < w.bytes) {
    pw_set_error("pw_lss_lift_pool: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    return PW_ENOSPC;
  }
  hipStream_t st = pw_stream(stream);
  const GridParams gp{lower3_host[0], lower3_host[1], lower3_host[2], interval3_host[0], interval3_host[1], interval3_host[2],
                      gx, gy, gz};
  const int64_t nz = (int64_t)(w.zero_bytes / 16);
  hipLaunchKernelGGL(k_lss_prologue, dim3((unsigned)pw_cdiv(nz, 256)), dim3(256), 0, st, B * N, sensor2ego, cam2imgs, post_rots,
                     w.ipr, w.comb, w.tr, (int4*)w.count, nz);
  int32_t* cur = w.count + n_vox;
  hipLaunchKernelGGL(k_lss_index_slots, dim3((unsigned)pw_cdiv(total, 256)), dim3(256), 0, st, N, DHW, total, frustum, w.ipr,
                     post_trans, w.comb, w.tr, bda, gp, w.count, w.slots, w.rank);
  hipLaunchKernelGGL(k_lss_heavy_alloc, dim3((unsigned)pw_cdiv(n_vox, 256 * HA_ITEMS)), dim3(256), 0, st, w.count, n_vox, cur,
                     w.hstart, w.list_b, w.list_c);
  hipLaunchKernelGGL(k_lss_ovf_scatter, dim3((unsigned)pw_cdiv(total, 256)), dim3(256), 0, st, N, DHW, total, frustum, w.ipr,
                     post_trans, w.comb, w.tr, bda, gp, w.rank, w.hstart, w.tmp);
  const FeatIdx fi{(int)DHW, H * W, 1.0f / (float)DHW, 1.0f / (float)(H * W)};
  const int64_t want = pw_cdiv(pw_cdiv(n_vox, 64), 4);
  hipLaunchKernelGGL(k_lss_pool_slots, dim3((unsigned)(want < 4096 ? want : 4096) + C_BLOCKS + B_BLOCKS), dim3(256), SORT_LDS_IDS * 4,
                     st, depth, (const float4*)feat, w.count, w.slots, n_vox, cur, w.list_b, w.list_c, w.tmp, w.sorted_pf, w.sorted_d,
                     fi, (float4*)out, out_h2, out_rng);
  pw_note_kernel("k_lss_pool_slots");
  PW_CHECK_LAUNCH();
  return PW_OK;
}
