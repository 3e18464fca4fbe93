// Weight gradient of the 3x3x3 stride-1 pad-1 convolutions on the fp16 matrix cores with split-fp16 operands (pw_h2.h):
//   dW[co][ci][kd][kh][kw] = sum over output voxels v of dY[v][co] * X[v + (kd, kh, kw) - 1][ci]
// (what torch autograd computes for mmdet3d/models/backbones/resnet.py:88-123's Conv3d in training).  pw_train.hip's
// k_conv3d_wgrad does this with v_mfma_f32_32x32x2_f32 and both operands straight from global memory (65-83 TFLOP/s); here
// K = 16 output voxels of a row per v_mfma_f32_32x32x16_f16 and three MFMAs per product block (hi*hi + lo*hi + hi*lo, 22-bit
// significands, fp32 accumulation), 5.3x the matrix rate of the fp32 instruction.
//
// The MFMA wants 8 CONSECUTIVE K per lane, i.e. 8 consecutive voxels of ONE channel, and the tensors are channels-last: operands
// go through LDS transposed.  A block walks a strip of 64 output columns down the rows (b, od, oh0 .. oh1) for one kd:
//   * per row it stages dY[row][strip][co tile(s)] and the ONE new input row X[id][oh + 1][strip -1 .. +64][ci tile(s)] -- the rows
//     oh - 1 and oh are still in LDS from the previous steps (ring of three), so every X row is read from global memory once per
//     kd instead of once per (kd, kh) -- split into hi / lo fp16 planes, stored channel-major: [plane][channel][position], two
//     voxels per 32-bit write, row stride 176 B = 4 x 11 dwords so that the 16 lanes of a ds_read_b128 phase cover all 64 banks;
//   * a wave owns one (32 co x 32 ci) pair of the block's tile and keeps its 9 accumulator tiles (kh, kw) of this kd in registers
//     (144 AGPRs).  Per K-step: 2 ds_read_b128 of dY (hi, lo), per kh 2 ds_read_b128 of X (hi, lo) + 4 ds_read_b32 (the halves
//     before and after the lane's block); the kw = 0 / 2 operands are the block shifted by one half, built with 4 v_alignbit each;
//     27 MFMAs per (K-step, wave).
// Tiles narrower than 64 x 64 (Cin or Cout == 32) give the spare waves every second / fourth K-step of the same pair; they write
// their own partial tiles.  Values are scaled by a per-tensor power of two taken from the tensors' largest magnitudes (amax2, device
// memory, no host sync) so that they sit in fp16's range whatever the scale of the gradients; the product of the two is taken out
// again when the partial tile is written.  Partial tiles per (chunk, tap, co block, ci block) are summed in a fixed order by
// pw_train.hip's k_wgrad_reduce: deterministic, no atomics.
#include <type_traits>

#include "pw_h2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int WG_NS = 4;                       // K-steps of 16 output columns per strip
constexpr int WG_COLS = 16 * WG_NS;            // 64
constexpr int WG_STRIDE = 176;                 // bytes per channel row of a plane: (8 + 64 + 8) halves = 160 B, padded to 4 x 11 dwords
constexpr int WG_PLANE = 32 * WG_STRIDE + 16;  // one plane (hi or lo) of a 32-channel tile; rows 16..31 sit 16 B further (wg_row)
constexpr int WG_TILE = 2 * WG_PLANE + 32;     // 11 328 B; the second tile of an operand sits 8 banks further
// byte offset of channel row c inside a plane.  The 16 extra bytes after row 15 and the 32 after a tile are for the STAGING writes:
// a wave writes 8 channel quads x 2 tiles x 4 position pairs per instruction, and with plain strides quads q / q + 4 and the two
// tiles fell on the same banks (4-way conflicts on all 16 writes of a task); the ds_read_b128 phases (16 consecutive rows) are
// unaffected.
__host__ __device__ constexpr int wg_row(int c) { return c * WG_STRIDE + ((c >> 4) << 4); }
constexpr int WG_THREADS = 768;                // 12 waves: 3 kh x 4 (pair, K-subset) -- three per SIMD, so one wave's staging VALU work runs under another's MFMAs

struct WgH2Args {
  const float* x;
  const float* dy;
  float* partial;            // [chunk][27][co_blocks][ci_blocks][32 * 32]
  const float* amax_x;       // WG_AMAX_PARTS partial maxima of |x| / |dy| (pw_absmax2, or recorded by the kernel that wrote the
  const float* amax_y;       // tensor: pw_bn_apply / pw_bn_bwd_apply), or null = no pre-scale
  int B, D, H, W, Cin, Cout;
  int co_t, ci_t;            // 32-channel tiles per block along co / ci (1 or 2)
  int cog, cig;              // tile groups along co / ci
  int n_strips, oh_splits, rows_per_split;
  int co_blocks, ci_blocks;
};

constexpr int WG_AMAX_PARTS = 256;             // partial maxima per tensor (pw_absmax2)
// two voxels' values of one channel -> packed hi pair, packed lo pair (low half = first voxel)
__device__ __forceinline__ void wg_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const h2_f2 x = {a, b};
  const h2_h2 h = __builtin_convertvector(x, h2_h2);
  h2_f2 d = h2_residual2(h, x);
  d[0] = __builtin_amdgcn_fmed3f(d[0], -H2_MAX, H2_MAX);
  d[1] = __builtin_amdgcn_fmed3f(d[1], -H2_MAX, H2_MAX);
  const h2_h2 l = __builtin_convertvector(d, h2_h2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ h8 wg_as_h8(const wg_u4& v) { return __builtin_bit_cast(h8, v); }
}  // namespace

template <int CO_T, int CI_T>
__global__ void __launch_bounds__(WG_THREADS) k_conv3d_wgrad_h2(WgH2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = CO_T * CI_T;               // (co tile, ci tile) pairs of the block
  constexpr int KSUB = 4 / P;                  // waves per pair, each taking every KSUB-th K-step
  constexpr int X_TASKS = 34 * 8 * CI_T, Y_TASKS = 32 * 8 * CO_T;
  constexpr int X_ROUNDS = (X_TASKS + WG_THREADS - 1) / WG_THREADS, Y_ROUNDS = (Y_TASKS + WG_THREADS - 1) / WG_THREADS;
  unsigned char* xring = lds;                                  // [4][CI_T] tiles: input rows oh - 1 .. oh + 2
  unsigned char* ybuf = lds + 4 * CI_T * WG_TILE;              // [2][CO_T] tiles: dY rows oh, oh + 1
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kh = wave % 3, pair = (wave / 3) % P, ksub = wave / (3 * P);
  const int ct = pair / CI_T, it = pair % CI_T;
  // one-dimensional grid, remapped so that an XCD works on a contiguous range of logical blocks (xcd_contiguous): the three kd
  // blocks of a (strip, b, od, rows) -- which read the SAME dY rows in lockstep -- and their neighbours then share one L2; dealt
  // round-robin they sat on three XCDs and every row crossed the fabric three times
  int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int ig = bid % a.cig; bid /= a.cig;
  const int cg = bid % a.cog; bid /= a.cog;
  const int kd = bid % 3; bid /= 3;
  const int strip = bid % a.n_strips;
  const int zi = bid / a.n_strips;
  const int split = zi % a.oh_splits, bd = zi / a.oh_splits;
  const int od = bd % a.D, b = bd / a.D;
  const int id = od + kd - 1;
  const int ow0 = strip * WG_COLS;
  const int ns = min(WG_NS, (a.W - ow0 + 15) / 16);
  const int oh0 = split * a.rows_per_split, oh1 = min(a.H, oh0 + a.rows_per_split);
  // exponents e such that amax * 2^-e lies in [2^12, 2^13) (0 for zero / non-finite / absent): the block reduces pw_absmax2's partials
  int ex = 0, ey = 0;
  if (a.amax_x && a.amax_y) {
    __shared__ unsigned amx[2][WG_THREADS / 64];
    const unsigned* px = reinterpret_cast<const unsigned*>(a.amax_x);
    const unsigned* py = reinterpret_cast<const unsigned*>(a.amax_y);
    unsigned m0 = threadIdx.x < WG_AMAX_PARTS ? px[threadIdx.x] : 0u, m1 = threadIdx.x < WG_AMAX_PARTS ? py[threadIdx.x] : 0u;
    m0 = wave_umax(m0); m1 = wave_umax(m1);
    if (lane == 0) { amx[0][wave] = m0; amx[1][wave] = m1; }
    __syncthreads();
    m0 = 0u; m1 = 0u;
#pragma unroll
    for (int w = 0; w < WG_THREADS / 64; ++w) { m0 = max(m0, amx[0][w]); m1 = max(m1, amx[1][w]); }
    ex = rng_ideal_exp(m0); ey = rng_ideal_exp(m1);
  }
  const float sx = rng_pow2(-ex), sy = rng_pow2(-ey);

  f32x16 acc[3];                                 // kw = 0, 1, 2 of this wave's kh
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if ((unsigned)id < (unsigned)a.D && oh0 < oh1) {
    const float* xplane = a.x + (((size_t)b * a.D + id) * a.H) * (size_t)a.W * a.Cin + (size_t)(ig * CI_T) * 32;
    const float* yplane = a.dy + (((size_t)b * a.D + od) * a.H) * (size_t)a.W * a.Cout + (size_t)(cg * CO_T) * 32;
    float4 xr0[X_ROUNDS][2], yr0[Y_ROUNDS][2], xr1[X_ROUNDS][2], yr1[Y_ROUNDS][2];     // two row sets in flight
    // A thread's staging tasks never change: (position pair, tile, channel quad) per round.  Everything that depends only on them
    // is computed ONCE -- element offset inside a row, LDS byte offset, and a per-voxel scale that is the tensor's pre-scale or 0
    // for a column outside the volume (the load then goes to a clamped column, unconditionally: no branches, no 64-bit address
    // arithmetic in the loop -- the first version spent ~180 VALU instructions per task there and the staging phase, not the
    // MFMAs, set the step time).
    int xo[X_ROUNDS][2], yo[Y_ROUNDS][2], xl[X_ROUNDS], yl[Y_ROUNDS];
    float xs[X_ROUNDS][2], ys[Y_ROUNDS][2];
#pragma unroll
    for (int r = 0; r < X_ROUNDS; ++r) {
      const int t = (r * WG_THREADS + (int)threadIdx.x) % X_TASKS;      // surplus threads repeat a task: identical stores, no branch
      const int cq = t & 7, tile = (t >> 3) % CI_T, pp = t / (8 * CI_T);
      xl[r] = tile * WG_TILE + wg_row(cq * 4) + (6 + 2 * pp) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int w = ow0 - 2 + 2 * pp + e;
        xs[r][e] = (unsigned)w < (unsigned)a.W ? sx : 0.f;
        xo[r][e] = min(max(w, 0), a.W - 1) * a.Cin + tile * 32 + cq * 4;
      }
    }
#pragma unroll
    for (int r = 0; r < Y_ROUNDS; ++r) {
      const int t = (r * WG_THREADS + (int)threadIdx.x) % Y_TASKS;
      const int cq = t & 7, tile = (t >> 3) % CO_T, pp = t / (8 * CO_T);
      yl[r] = tile * WG_TILE + wg_row(cq * 4) + (8 + 2 * pp) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int w = ow0 + 2 * pp + e;
        ys[r][e] = w < a.W ? sy : 0.f;
        yo[r][e] = min(w, a.W - 1) * a.Cout + tile * 32 + cq * 4;
      }
    }
    // global loads of one input row (ih) / one dY row (oh) into registers: a uniform row base + the thread's offsets
    auto load_x = [&](float4 (&xr)[X_ROUNDS][2], int ih) {
      const float* row = xplane + (size_t)min(max(ih, 0), a.H - 1) * a.W * a.Cin;
#pragma unroll
      for (int r = 0; r < X_ROUNDS; ++r) {
        xr[r][0] = *reinterpret_cast<const float4*>(row + xo[r][0]);
        xr[r][1] = *reinterpret_cast<const float4*>(row + xo[r][1]);
      }
    };
    auto load_y = [&](float4 (&yr)[Y_ROUNDS][2], int oh) {
      const float* row = yplane + (size_t)oh * a.W * a.Cout;
#pragma unroll
      for (int r = 0; r < Y_ROUNDS; ++r) {
        yr[r][0] = *reinterpret_cast<const float4*>(row + yo[r][0]);
        yr[r][1] = *reinterpret_cast<const float4*>(row + yo[r][1]);
      }
    };
    // registers -> LDS, split and transposed: channel c of the tile at row c of both planes, position pair pp at half index
    // 6 + 2 pp (X: column ow0 - 2 + 2 pp) resp. 8 + 2 pp (dY: column ow0 + 2 pp)
    auto store_task = [&](unsigned char* p, const float4 (&reg)[2], float s0, float s1) {
      const float v0[4] = {reg[0].x * s0, reg[0].y * s0, reg[0].z * s0, reg[0].w * s0};
      const float v1[4] = {reg[1].x * s1, reg[1].y * s1, reg[1].z * s1, reg[1].w * s1};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned hi, lo;
        wg_split_pair(v0[c], v1[c], hi, lo);
        *reinterpret_cast<unsigned*>(p + c * WG_STRIDE) = hi;
        *reinterpret_cast<unsigned*>(p + c * WG_STRIDE + WG_PLANE) = lo;
      }
    };
    auto store_x = [&](const float4 (&xr)[X_ROUNDS][2], int ih) {
      unsigned char* base = xring + ((ih + 4) & 3) * CI_T * WG_TILE;
      const float rowf = (unsigned)ih < (unsigned)a.H ? 1.f : 0.f;          // a row outside the volume is stored as zeros
#pragma unroll
      for (int r = 0; r < X_ROUNDS; ++r)
        store_task(base + xl[r], xr[r], xs[r][0] * rowf, xs[r][1] * rowf);
    };
    auto store_y = [&](const float4 (&yr)[Y_ROUNDS][2], int oh) {
      unsigned char* ybase = ybuf + (oh & 1) * CO_T * WG_TILE;
#pragma unroll
      for (int r = 0; r < Y_ROUNDS; ++r)
        store_task(ybase + yl[r], yr[r], ys[r][0], ys[r][1]);
    };
    const int m = lane & 31, h = lane >> 5;
    // the MFMAs of output row oh: dY slot oh & 1, input rows oh - 1 .. oh + 1 in slots (row & 3)
    auto compute = [&](int oh) {
      const int ih = oh + kh - 1;
      if ((unsigned)ih >= (unsigned)a.H) return;         // wave-uniform: this wave's input row lies outside the volume
      const unsigned char* yt = ybuf + ((oh & 1) * CO_T + ct) * WG_TILE + wg_row(m);
      for (int s = ksub; s < ns; s += KSUB) {
        const int off = (8 + 16 * s + 8 * h) * 2;        // byte offset of the lane's 8-column block in a channel row
        const wg_u4 ah = *reinterpret_cast<const wg_u4*>(yt + off);
        const wg_u4 al = *reinterpret_cast<const wg_u4*>(yt + WG_PLANE + off);
        {
          const unsigned char* xt = xring + (((ih + 4) & 3) * CI_T + it) * WG_TILE + wg_row(m) + off;
          wg_u4 bv[3][2];                                // [kw][plane]
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const wg_u4 c = *reinterpret_cast<const wg_u4*>(xt + pl * WG_PLANE);
            const unsigned bp = *reinterpret_cast<const unsigned*>(xt + pl * WG_PLANE - 4);
            const unsigned bn = *reinterpret_cast<const unsigned*>(xt + pl * WG_PLANE + 16);
            bv[1][pl] = c;
            // columns shifted by -1: halves [-1 .. 6]; by +1: halves [1 .. 8]
            bv[0][pl] = wg_u4{__builtin_amdgcn_alignbit(c.x, bp, 16), __builtin_amdgcn_alignbit(c.y, c.x, 16),
                              __builtin_amdgcn_alignbit(c.z, c.y, 16), __builtin_amdgcn_alignbit(c.w, c.z, 16)};
            bv[2][pl] = wg_u4{__builtin_amdgcn_alignbit(c.y, c.x, 16), __builtin_amdgcn_alignbit(c.z, c.y, 16),
                              __builtin_amdgcn_alignbit(c.w, c.z, 16), __builtin_amdgcn_alignbit(bn, c.w, 16)};
          }
          // product-major: the three MFMAs into one accumulator are two other accumulators apart
#pragma unroll
          for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg_as_h8(prod == 1 ? al : ah), wg_as_h8(bv[kw][prod == 2 ? 1 : 0]), acc[kw], 0, 0, 0);
        }
      }
    };
    // prologue: input rows oh0 - 1 .. oh0 + 1 and dY row oh0 into their slots (exposed, once per block); the rows of step
    // oh0 + 1 in flight in set 0
    load_x(xr0, oh0 - 1); load_x(xr1, oh0); load_y(yr0, oh0);
    store_x(xr0, oh0 - 1); store_x(xr1, oh0); store_y(yr0, oh0);
    load_x(xr0, oh0 + 1); store_x(xr0, oh0 + 1);
    if (oh0 + 1 < oh1) { load_x(xr0, oh0 + 2); load_y(yr0, oh0 + 1); }
    if (oh0 + 2 < oh1) { load_x(xr1, oh0 + 3); load_y(yr1, oh0 + 2); }
    __syncthreads();
    // step oh: split the rows of step oh + 1 (requested a step ago, set a) into the slots row oh does not read, request the rows
    // of step oh + 2 into the same registers, run row oh's MFMAs; one barrier per step.  The steady-state steps are BRANCH-FREE:
    // with a condition around a load or a store the compiler's s_waitcnt bookkeeping merges a path on which the registers'
    // earlier loads were never consumed and drains the (in-order) load counter before every new request -- each step then cost an
    // exposed memory latency.  For the same reason the barrier is not __syncthreads() (its release fence waits for vmcnt(0)), and
    // the store-first / compute-first order is a compile-time variant chosen once per wave.
    auto run = [&](auto sf_tag) {
      constexpr bool SF = decltype(sf_tag)::value;
      auto step_u = [&](int oh, float4 (&xa)[X_ROUNDS][2], float4 (&ya)[Y_ROUNDS][2], float4 (&xb)[X_ROUNDS][2], float4 (&yb)[Y_ROUNDS][2]) {
        if constexpr (SF) {
          store_x(xa, oh + 2); store_y(ya, oh + 1);
          load_x(xa, oh + 4); load_y(ya, oh + 3);
          compute(oh);
        } else {
          compute(oh);
          store_x(xa, oh + 2); store_y(ya, oh + 1);
          load_x(xa, oh + 4); load_y(ya, oh + 3);
        }
        (void)xb; (void)yb;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      };
      auto step_c = [&](int oh, float4 (&xa)[X_ROUNDS][2], float4 (&ya)[Y_ROUNDS][2]) {
        if (SF && oh + 1 < oh1) { store_x(xa, oh + 2); store_y(ya, oh + 1); }
        if (SF && oh + 3 < oh1) { load_x(xa, oh + 4); load_y(ya, oh + 3); }
        compute(oh);
        if (!SF && oh + 1 < oh1) { store_x(xa, oh + 2); store_y(ya, oh + 1); }
        if (!SF && oh + 3 < oh1) { load_x(xa, oh + 4); load_y(ya, oh + 3); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      };
      int oh = oh0;
      for (; oh + 4 < oh1; oh += 2) {
        step_u(oh, xr0, yr0, xr1, yr1);
        step_u(oh + 1, xr1, yr1, xr0, yr0);
      }
      for (; oh < oh1; oh += 2) {                         // the last (up to four) steps
        step_c(oh, xr0, yr0);
        if (oh + 1 < oh1) step_c(oh + 1, xr1, yr1);
      }
    };
    // waves 4 .. 7 store first, the others compute first: on every SIMD one wave's staging VALU work meets the other two's MFMAs
    if ((wave >> 2) & 1) run(std::true_type{}); else run(std::false_type{});
  }
  // narrow tiles: the KSUB waves of a (pair, kh) add their tiles through LDS in a fixed order (wave 0 + 1 (+ 2 + 3)), one kw at a time
  if constexpr (KSUB > 1) {
    float* red = reinterpret_cast<float*>(lds);          // [pair][kh][KSUB - 1][64 lanes x 16] floats
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      __syncthreads();
      if (ksub > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((pair * 3 + kh) * (KSUB - 1) + ksub - 1)) * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (ksub == 0) {
#pragma unroll
        for (int q = 0; q < KSUB - 1; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += red[((((pair * 3 + kh) * (KSUB - 1) + q)) * 16 + r) * 64 + lane];
      }
    }
    if (ksub > 0) return;
  }
  // partial tiles of this (chunk, K-subset): D row (co) = (r & 3) + 8 (r >> 2) + 4 h, column (ci) = lane & 31
  const float unx = rng_pow2(ex), uny = rng_pow2(ey);      // applied one after the other: 2^(ex + ey) alone can leave fp32's range
  const int chunk = zi * a.n_strips + strip;
  const int cob = cg * CO_T + ct, cib = ig * CI_T + it;
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = kd * 9 + kh * 3 + t;
    float* dst = a.partial + ((((size_t)chunk * 27 + tap) * a.co_blocks + cob) * a.ci_blocks + cib) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + i] = acc[t][r] * unx * uny;
  }
}

// sum of the per-chunk partial tiles -> torch's [Cout][Cin][kd][kh][kw].  A block owns 32 consecutive elements of a tile; its 8
// groups of 32 threads each add every 8th chunk (four loads in flight), the 8 sums meet in LDS and are added in group order:
// a fixed order, so the result is deterministic.
__global__ void __launch_bounds__(256) k_wgrad_h2_reduce(const float* __restrict__ partial, float* __restrict__ dw, int n_chunks,
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
    dw[((size_t)(cob * 32 + i) * Cin + cib * 32 + j) * taps + tap] = s;
  }
}

namespace {
struct WgPlan {
  int co_t, ci_t, cog, cig, n_strips, oh_splits, rows_per_split, ksub, n_chunks;
  size_t lds;
};
WgPlan wg_plan(int B, int D, int H, int W, int Cin, int Cout) {
  WgPlan p;
  p.co_t = Cout % 64 == 0 ? 2 : 1;
  p.ci_t = Cin % 64 == 0 ? 2 : 1;
  p.cog = Cout / (32 * p.co_t);
  p.cig = Cin / (32 * p.ci_t);
  p.n_strips = (W + WG_COLS - 1) / WG_COLS;
  p.ksub = 4 / (p.co_t * p.ci_t);
  // enough blocks for ~4 per CU: 3 kd x tile groups x strips x (b, od) x row splits
  const int base = 3 * p.cog * p.cig * p.n_strips * B * D;
  int splits = (640 + base - 1) / base;
  if (splits > H / 8) splits = H / 8;
  if (splits < 1) splits = 1;
  p.rows_per_split = (H + splits - 1) / splits;
  p.oh_splits = (H + p.rows_per_split - 1) / p.rows_per_split;
  p.n_chunks = B * D * p.oh_splits * p.n_strips;
  p.lds = (size_t)(4 * p.ci_t + 2 * p.co_t) * WG_TILE;
  return p;
}
}  // namespace

PW_API size_t pw_conv3d_wgrad_h2_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return 0;
  const WgPlan p = wg_plan(B, D, H, W, Cin, Cout);
  return (size_t)p.n_chunks * 27 * (Cout / 32) * (Cin / 32) * 1024 * 4;
}

PW_API int pw_conv3d_wgrad_h2(const float* x, const float* dy, float* dw, const float* amax_x, const float* amax_y, void* workspace,
                              size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
  PW_CHECK_ARG(x && dy && dw && workspace, "pw_conv3d_wgrad_h2: null pointer");
  PW_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0,
               "pw_conv3d_wgrad_h2: 3x3x3 stride 1, Cin and Cout multiples of 32 (pw_conv3d_wgrad takes everything else)");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "pw_conv3d_wgrad_h2: x / dy must be 16-byte aligned");
  PW_CHECK_ARG(workspace_bytes >= pw_conv3d_wgrad_h2_workspace_bytes(B, D, H, W, Cin, Cout), "pw_conv3d_wgrad_h2: workspace too small");
  const WgPlan p = wg_plan(B, D, H, W, Cin, Cout);
  WgH2Args a;
  a.x = x; a.dy = dy; a.partial = (float*)workspace; a.amax_x = amax_x; a.amax_y = amax_y;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.co_t = p.co_t; a.ci_t = p.ci_t; a.cog = p.cog; a.cig = p.cig;
  a.n_strips = p.n_strips; a.oh_splits = p.oh_splits; a.rows_per_split = p.rows_per_split;
  a.co_blocks = Cout / 32; a.ci_blocks = Cin / 32;
  hipStream_t st = pw_stream(stream);
  const dim3 grid((unsigned)(3 * p.cog * p.cig * p.n_strips * B * D * p.oh_splits));
#define PW_WG_LAUNCH(CO, CI)                                                                                                    \
  do {                                                                                                                          \
    /* per launch: the attribute is per device, and the call is cheap next to a ~100 us kernel */                  \
    PW_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv3d_wgrad_h2<CO, CI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds)); \
    hipLaunchKernelGGL((k_conv3d_wgrad_h2<CO, CI>), grid, dim3(WG_THREADS), p.lds, st, a);                                             \
  } while (0)
  if (p.co_t == 2 && p.ci_t == 2) PW_WG_LAUNCH(2, 2);
  else if (p.co_t == 2) PW_WG_LAUNCH(2, 1);
  else if (p.ci_t == 2) PW_WG_LAUNCH(1, 2);
  else PW_WG_LAUNCH(1, 1);
#undef PW_WG_LAUNCH
  const size_t per_chunk = (size_t)27 * a.co_blocks * a.ci_blocks * 1024;
  hipLaunchKernelGGL(k_wgrad_h2_reduce, dim3((unsigned)(per_chunk / 32)), dim3(256), 0, st, a.partial, dw, p.n_chunks, 27,
                     a.co_blocks, a.ci_blocks, Cout, Cin);
  pw_note_kernel("k_conv3d_wgrad_h2<%d, %d>", p.co_t, p.ci_t);
  PW_CHECK_LAUNCH();
  return PW_OK;
}

// largest magnitudes of two tensors in one launch (the per-tensor pre-scales of pw_conv3d_wgrad_h2): block b of tensor i leaves the
// maximum of its grid-stride share at out[i * WG_AMAX_PARTS + b] as a float (bit-pattern maximum: a NaN ends above every number).
// No atomics (2 x 8 192 same-address atomic maxima cost 50 us), no zero-initialised output; the consumer reduces the partials.
__global__ void __launch_bounds__(256) k_absmax2(const float4* __restrict__ x, int64_t nx4, const float4* __restrict__ y, int64_t ny4,
                                                 unsigned* __restrict__ out) {
  __shared__ unsigned wm[4];
  const bool second = blockIdx.y == 1;
  const float4* p = second ? y : x;
  const int64_t n4 = second ? ny4 : nx4;
  unsigned m = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    m = max(max(m, rng_absbits(v.x)), max(rng_absbits(v.y), max(rng_absbits(v.z), rng_absbits(v.w))));
  }
  m = wave_umax(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[(second ? WG_AMAX_PARTS : 0) + blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}

PW_API int pw_absmax2(const float* x, int64_t nx, const float* y, int64_t ny, float* out, void* stream) {
  PW_CHECK_ARG(x && y && out && nx >= 0 && ny >= 0 && nx % 4 == 0 && ny % 4 == 0, "pw_absmax2: bad arguments (element counts must be multiples of 4)");
  PW_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pw_absmax2: x / y must be 16-byte aligned");
  hipLaunchKernelGGL(k_absmax2, dim3(WG_AMAX_PARTS, 2), dim3(256), 0, pw_stream(stream), (const float4*)x, nx / 4, (const float4*)y, ny / 4,
                     (unsigned*)out);
  PW_CHECK_LAUNCH();
  return PW_OK;
}
