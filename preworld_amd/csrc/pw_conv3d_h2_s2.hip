// 3x3x3 STRIDE-2 convolution of the voxel encoder on the fp16 matrix cores, split-fp16 operands, input tile staged in LDS.
//   mmdet3d/models/backbones/resnet.py:88-184 -- the first BasicBlock3D of stages 1 and 2: conv1 (stride 2, BN, ReLU) and the
//   `downsample` conv (stride 2, BN) read the same input and run as ONE pass over N = 2 x Cout packed columns.
//
// The gather kernel (pw_conv3d_gather.hip) fetches every tap's activations AND weights from L2 per 32 output voxels: 135 us /
// 82 us for the two layers of the C3 encoder, 0.11-0.15 of the fp16 peak.  Staging the input tile in LDS removes the
// activation re-reads, but a first version with a 2 x 4 x 4 output tile per block (5 x 9 x 9 halo of 128-byte voxel chunks,
// three blocks per CU) stopped at 94 / 60 us: every block streams the whole weight set of a chunk (27 taps x cout tiles x 4 KB =
// 432 KB for 128 columns) through L2 -> 1.08 GB per launch at the ~11 TB/s the L2s deliver.  Weight bytes per output voxel are
// what matters, so the tile has to be as large as LDS allows:
//   * a pass covers ONE k-step (16 input channels = 64 bytes per voxel: {hi, lo} x {k-half 0, 1}) instead of a 32-channel
//     chunk, which lets a block own 2 x 4 x 8 = 64 OUTPUT voxels (two 32-column MFMA tiles per wave) with a 5 x 9 x 17 halo
//     of 48 960 bytes -- still three blocks per CU -- and halves the weight traffic per voxel;
//   * GEMM transposed like k_conv3d_h2: D[cout][voxel] += W[cout][k] X[k][voxel]; wave w owns the cout tiles w, w + 4, ...;
//     per tap and voxel tile 2 ds_read_b128 (hi, lo of the lane's voxel) feed 3 MFMAs per cout tile (hi_w.hi_x, lo_w.hi_x,
//     hi_w.lo_x); the weights of the next tap (L2, packed [chunk][tap][tile][lane][64 B]) are requested before this tap's MFMAs;
//   * LDS layout: with stride 2 every lane's halo coordinate is even + tap, so in a voxel-major layout all lanes of a
//     fragment read would share the low bits of the voxel index and collide.  A halo row (17 voxels) is stored as its 9 even
//     columns then its 8 odd ones, so that consecutive output columns read consecutive 64-byte voxels, and the 16-byte piece
//     index is XORed with (hh >> 1) & 3: under ds_read_b128's lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, same + 32) the
//     16 lanes of a group then hit 16 different 16-byte bank groups for every tap;
//   * out-of-volume halo voxels come back as zeros from an out-of-range buffer offset; the epilogue is the vector h2 epilogue
//     of the gather kernel (float4 scale / bias, ReLU per destination, 8-byte hi + 8-byte lo stores).
// Built for what the encoder uses: every destination in h2 storage, no residual, cout_total = 128 or 256 (4 waves x 32 x NTW);
// anything else stays on the gather kernel.
#include "pw_h2.h"

namespace {
constexpr int S2_TD = 2, S2_TH = 4, S2_TW = 8;                 // output tile: two voxel tiles of 2 x 4 x 4 side by side in w
constexpr int S2_HD = 5, S2_HH = 9, S2_HW = 17;                // input halo
constexpr int S2_EVEN = 9;                                     // even columns of a halo row come first
constexpr int S2_VOX = S2_HD * S2_HH * S2_HW;                  // 765
constexpr int S2_LDS = S2_VOX * 64;                            // 48 960
constexpr unsigned S2_OOB = 0xfffffff0u;
// Compile-time A/B switches of two measured-and-rejected variants, kept because the code paths are tested by
// building with them: -DS2_NG_V=3 puts 3 (NTW = 1) / 2 (NTW = 2) tile groups of 4 waves in one block, in lockstep through the
// block barriers, so that their identical weight requests meet in L1 (taps 3x faster, but nothing covers the staging round
// trip: 99 / 74 us); -DS2_PREFETCH=1 additionally issues the next pass's halo loads before the taps (95 / 71 us; vector
// loads return in order, so the first weight wait of the pass then waits for the halo as well).  Default: one tile per block,
// three (two) blocks per CU, 77 / 45 us.
#ifndef S2_NG_V
#define S2_NG_V 1
#endif
#ifndef S2_PREFETCH
#define S2_PREFETCH 0
#endif
template <int NTW> constexpr int S2_NG = S2_NG_V == 1 ? 1 : (NTW == 1 ? 3 : 2);      // tiles (4-wave groups) per block

__device__ __forceinline__ v4f s2_load4(rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);       // keep `auto` (pw_conv3d_common.h)
  v4f o;
  o[0] = __uint_as_float(v[0]); o[1] = __uint_as_float(v[1]); o[2] = __uint_as_float(v[2]); o[3] = __uint_as_float(v[3]);
  return o;
}

struct S2Ctx {
  const char* lds;
  rsrc_t wr;
  unsigned lane_off, tile_off, wstride;        // lane * 16 (+ k-step pieces); first cout tile of the wave x 4096; bytes per tap
  unsigned ra[2][2];                           // fragment addresses [kh >> 1][plane]
};

template <int NTW>
__device__ __forceinline__ void s2_load_w(const S2Ctx& c, unsigned wsoff, v4f (&bw)[NTW][2]) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int p = 0; p < 2; ++p) bw[nt][p] = s2_load4(c.wr, c.lane_off + (unsigned)p * H2W_PIECE, wsoff + c.tile_off + (unsigned)(nt * 4 * 4096));
}

template <int TAP>
__device__ __forceinline__ void s2_read_x(const S2Ctx& c, v4f (&ax)[2][2]) {
  constexpr int kd = TAP / 9, kh = (TAP / 3) % 3, kw = TAP % 3;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int imm = ((kd * S2_HH + kh) * S2_HW + (kw & 1) * S2_EVEN + (kw >> 1) + 4 * nb) * 64;
#pragma unroll
    for (int p = 0; p < 2; ++p) ax[nb][p] = *reinterpret_cast<const v4f*>(c.lds + c.ra[kh >> 1][p] + imm);
  }
}

template <int NTW>
__device__ __forceinline__ void s2_mfma(const v4f (&ax)[2][2], const v4f (&bw)[NTW][2], f32x16 (&acc)[2][NTW]) {
#pragma unroll
  for (int prod = 0; prod < 3; ++prod) {                 // hi_w.hi_x, lo_w.hi_x, hi_w.lo_x
    const int pw = prod == 1 ? 1 : 0, px = prod == 2 ? 1 : 0;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        acc[nb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, bw[nt][pw]), __builtin_bit_cast(h8, ax[nb][px]),
                                                             acc[nb][nt], 0, 0, 0);
  }
}

// tap TAP computes on ax[TAP & 1] and ring slot TAP % S2_WD; the activation fragments of tap TAP + 1 and the weights of tap
// TAP + S2_WD - 1 are requested first.  (With the weights only one tap ahead a tap took ~1 200 cycles -- an L2 round trip under
// load -- against 192 cycles of MFMAs; three waves per SIMD cannot cover that.)
constexpr int S2_WD = S2_PREFETCH ? 3 : 5;
template <int NTW, int TAP>
__device__ __forceinline__ void s2_step(const S2Ctx& c, unsigned wsoff, v4f (&ax)[2][2][2], v4f (&bw)[S2_WD][NTW][2], f32x16 (&acc)[2][NTW]) {
  if constexpr (TAP + S2_WD - 1 < 27) s2_load_w<NTW>(c, wsoff + (unsigned)(TAP + S2_WD - 1) * c.wstride, bw[(TAP + S2_WD - 1) % S2_WD]);
  if constexpr (TAP + 1 < 27) s2_read_x<TAP + 1>(c, ax[(TAP + 1) & 1]);
  __builtin_amdgcn_sched_barrier(0);
  s2_mfma<NTW>(ax[TAP & 1], bw[TAP % S2_WD], acc);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TAP + 1 < 27) s2_step<NTW, TAP + 1>(c, wsoff, ax, bw, acc);
}
}  // namespace

template <int NTW>      // cout tiles per wave: cout_total = 128 NTW
__global__ void __launch_bounds__(256 * S2_NG<NTW>, S2_NG_V == 1 ? (NTW == 1 ? 3 : 2) : 1) k_conv3d_h2_s2(ConvArgs a, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char s2_lds[];
  // S2_NG<NTW> tile groups of 4 waves per block (1 by default; see the switches at the top of the file)
  const int group = uni((int)threadIdx.x >> 8);
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = uni(tid >> 6);                  // scalar: it feeds the weight loads' scalar offset (no waterfall loop)
  const int half = lane >> 5, j = lane & 31;
  const int od = j >> 4, oh = (j >> 2) & 3, ow = j & 3;       // this lane's output voxel inside a 2 x 4 x 4 voxel tile
  int t = (int)blockIdx.x * S2_NG<NTW> + group;
  const bool live = t < n_tiles;                   // a dead group only keeps the barriers company
  if (!live) t = n_tiles - 1;
  const int tw = t % a.tiles_w; t /= a.tiles_w;
  const int th = t % a.tiles_h; t /= a.tiles_h;
  const int td = t % a.tiles_d;
  const int b = t / a.tiles_d;
  const int d0 = td * S2_TD, h0 = th * S2_TH, w0 = tw * S2_TW;
  const int ntiles_total = a.cout_total >> 5, nchunk = a.Cin / KC;
  // range exponents of x / y0 / y1 (pw_h2.h "Range"), folded into scale / bias in the epilogue: loaded here, first used there
  const RngScale rs = rng_scales(a);

  const rsrc_t xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.D * a.H * a.W * a.Cin * 4));
  S2Ctx c;
  char* const my_lds = s2_lds + group * S2_LDS;
  c.lds = my_lds;
  c.wr = make_rsrc(a.wpk, (unsigned)((size_t)nchunk * 27 * ntiles_total * 4096));
  c.tile_off = (unsigned)wave * 4096u;
  c.wstride = (unsigned)ntiles_total * 4096u;
  {
    const int base_v = ((2 * od) * S2_HH + 2 * oh) * S2_HW + ow;
#pragma unroll
    for (int khh = 0; khh < 2; ++khh)
#pragma unroll
      for (int p = 0; p < 2; ++p) c.ra[khh][p] = (unsigned)(base_v * 64 + (((2 * half + p) ^ ((oh + khh) & 3)) * 16));
  }

  f32x16 acc[2][NTW];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][nt][r] = 0.f;

  long long ts[6] = {0, 0, 0, 0, 0, 0};
  if (a.probe) ts[0] = __builtin_readcyclecounter();
  // halo of a pass: 4 pieces of 16 bytes per voxel (t = 2 k-half + plane); all loads of a thread go out before its first LDS
  // write (one round trip per pass); the co-resident blocks cover each other's staging
  constexpr int NIT = (S2_VOX * 4 + 255) / 256;
  v4f val[NIT];
  auto issue_halo = [&](int pass) {
    const int ch = pass >> 1, ks = pass & 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it;
      const int v = idx >> 2, tp = idx & 3;
      const int dd = v / (S2_HH * S2_HW), r = v - dd * (S2_HH * S2_HW);
      const int hh = r / S2_HW, ww = r - hh * S2_HW;
      const int gd = 2 * d0 - 1 + dd, gh = 2 * h0 - 1 + hh, gw = 2 * w0 - 1 + ww;
      const bool ok = live && idx < S2_VOX * 4 && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
      const int slot = 4 * (tp >> 1) + 2 * ks + (tp & 1);
      const unsigned goff = ok ? (unsigned)((((((size_t)b * a.D + gd) * a.H + gh) * a.W + gw) * a.Cin + ch * KC) * 4 + slot * 16) : S2_OOB;
      val[it] = s2_load4(xr, goff, 0);
    }
  };
  if (S2_PREFETCH) issue_halo(0);
  for (int pass = 0; pass < 2 * nchunk; ++pass) {
    const int ch = pass >> 1, ks = pass & 1;
    if (!S2_PREFETCH) issue_halo(pass);
    if (pass) __syncthreads();                       // everyone is done with the previous pass's halo
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it;
      const int v = idx >> 2, tp = idx & 3;
      const int dd = v / (S2_HH * S2_HW), r = v - dd * (S2_HH * S2_HW);
      const int hh = r / S2_HW, ww = r - hh * S2_HW;
      const int vp = (dd * S2_HH + hh) * S2_HW + (ww & 1) * S2_EVEN + (ww >> 1);
      if (idx < S2_VOX * 4) *reinterpret_cast<v4f*>(my_lds + vp * 64 + ((tp ^ ((hh >> 1) & 3)) * 16)) = val[it];
    }
    __syncthreads();
    if (a.probe && pass == 0) ts[1] = __builtin_readcyclecounter();
    if (S2_PREFETCH && pass + 1 < 2 * nchunk) issue_halo(pass + 1);
    // ---- 27 taps of this k-step
    c.lane_off = (unsigned)lane * 16u + (unsigned)(2 * ks) * H2W_PIECE;
    const unsigned wsoff = (unsigned)(ch * 27) * c.wstride;
    v4f ax[2][2][2], bw[S2_WD][NTW][2];
#pragma unroll
    for (int k = 0; k < S2_WD - 1; ++k) s2_load_w<NTW>(c, wsoff + (unsigned)k * c.wstride, bw[k]);
    s2_read_x<0>(c, ax[0]);
    s2_step<NTW, 0>(c, wsoff, ax, bw, acc);
    if (a.probe && pass == 0) ts[2] = __builtin_readcyclecounter();
  }
  if (a.probe) ts[3] = __builtin_readcyclecounter();

  // ---- epilogue: lane = output voxel j, register r = output channel (r & 3) + 8 (r >> 2) + 4 half of the cout tile.  Straight
  // from the accumulators a lane owns 8-byte pieces of 32 different rows, i.e. every store instruction would touch 64 partial
  // lines (measured: the longest phase of the block).  So each (voxel tile, cout tile) goes through this wave's 4 KB of the
  // now idle halo buffer in h2 row format -- 16-byte slot XOR (voxel & 7) against the 128-byte row stride -- and comes back as
  // 16 bytes per lane with 8 consecutive lanes covering one voxel's 128-byte chunk: whole lines per store.
  __syncthreads();                                   // every wave is done with the halo
  char* const stg = my_lds + wave * 4096;
  float amax0 = 0.f, amax1 = 0.f;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n0 = (wave + 4 * nt) * 32;                      // a 32-column tile lies in one destination (cout0 % 32 == 0)
    const bool first = n0 < a.cout0;
    const int nn0 = first ? n0 : n0 - a.n1_start;
    if (!first && !(a.y1 && nn0 >= 0 && nn0 < a.cout1)) continue;          // wave-uniform
    float* const dst = first ? a.y0 : a.y1;
    const int ld = first ? a.ld0 : a.ld1;
    const bool relu = first ? a.relu0 != 0 : a.relu1 != 0;
    float sc[4][4], bi[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = 8 * q + 4 * half;
      const float4 s4 = a.scale ? *reinterpret_cast<const float4*>(a.scale + n0 + cc) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + n0 + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float sm = first ? rs.s0 : rs.s1, bm = first ? rs.b0 : rs.b1;
      sc[q][0] = s4.x * sm; sc[q][1] = s4.y * sm; sc[q][2] = s4.z * sm; sc[q][3] = s4.w * sm;
      bi[q][0] = b4.x * bm; bi[q][1] = b4.y * bm; bi[q][2] = b4.z * bm; bi[q][3] = b4.w * bm;
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      // this lane's accumulators belong to output voxel (od, oh, 4 nb + ow) of the tile: inside the volume?
      const bool mine = live && d0 + od < a.Do && h0 + oh < a.Ho && w0 + 4 * nb + ow < a.Wo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = 8 * q + 4 * half;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[nb][nt][4 * q + e] * sc[q][e] + bi[q][e];
          v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
        }
        const float m4 = mine ? fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) : 0.f;
        if (first) amax0 = fmaxf(amax0, m4); else amax1 = fmaxf(amax1, m4);
        u2 hi, lo;
        h2_split4(v, hi, lo);
        const int o0 = h2_group_off(cc, 0), o1 = h2_group_off(cc, 1);      // 16-byte slot = off >> 4, position inside = off & 15
        *reinterpret_cast<u2*>(stg + j * 128 + ((((o0 >> 4) ^ (j & 7)) << 4) | (o0 & 15))) = hi;
        *reinterpret_cast<u2*>(stg + j * 128 + ((((o1 >> 4) ^ (j & 7)) << 4) | (o1 & 15))) = lo;
      }
      // (same wave wrote and reads: program order, no barrier)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int jj = 8 * i + (lane >> 3), sl = lane & 7;
        const v4f piece = *reinterpret_cast<const v4f*>(stg + jj * 128 + ((sl ^ (jj & 7)) << 4));
        const int gd = d0 + (jj >> 4), gh = h0 + ((jj >> 2) & 3), gw = w0 + 4 * nb + (jj & 3);
        if (live && gd < a.Do && gh < a.Ho && gw < a.Wo) {
          const size_t vox = (((size_t)b * a.Do + gd) * a.Ho + gh) * a.Wo + gw;
          *reinterpret_cast<v4f*>(reinterpret_cast<char*>(dst + vox * ld + (nn0 & ~31)) + sl * 16) = piece;
        }
      }
    }
  }
  rng_note(a.y0_rng, __float_as_uint(amax0), rs.e0);
  if (a.y1) rng_note(a.y1_rng, __float_as_uint(amax1), rs.e1);
  if (a.probe && lane == 0 && blockIdx.x == 100 && group == 0) {     // {start, pass 0 staged, pass 0 taps done, all passes done, epilogue done}
    ts[4] = __builtin_readcyclecounter();
    long long* pp = a.probe + wave * 8;
    pp[0] = ts[0]; pp[1] = ts[1]; pp[2] = ts[2]; pp[3] = ts[3]; pp[4] = ts[4];
  }
}

// a: as filled by pw_conv3d_h2 (Do / Ho / Wo of the stride-2 output); returns PW_EUNSUP when the shape is not built
int pw_launch_conv3d_h2_s2(const ConvArgs& a0, hipStream_t st) {
  const int ntiles = a0.cout_total / 32;
  const bool h2epi = a0.fmt_y0 == 1 && (a0.cout1 == 0 || a0.fmt_y1 == 1) && !a0.residual;
  if (!h2epi || (ntiles != 4 && ntiles != 8)) return PW_EUNSUP;
  ConvArgs a = a0;
  a.tiles_d = (a.Do + S2_TD - 1) / S2_TD; a.tiles_h = (a.Ho + S2_TH - 1) / S2_TH; a.tiles_w = (a.Wo + S2_TW - 1) / S2_TW;
  const long long nblk = (long long)a.B * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk >= (1ll << 31)) return PW_EUNSUP;
  if (ntiles == 4) {
    constexpr int NG = S2_NG<1>;
    static int once = set_lds_limit(k_conv3d_h2_s2<1>, NG * S2_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_conv3d_h2_s2<1>, dim3((unsigned)((nblk + NG - 1) / NG)), dim3(256 * NG), NG * S2_LDS, st, a, (int)nblk);
    pw_note_kernel("k_conv3d_h2_s2<1>");
  } else {
    constexpr int NG = S2_NG<2>;
    static int once = set_lds_limit(k_conv3d_h2_s2<2>, NG * S2_LDS);
    if (once) return once;
    hipLaunchKernelGGL(k_conv3d_h2_s2<2>, dim3((unsigned)((nblk + NG - 1) / NG)), dim3(256 * NG), NG * S2_LDS, st, a, (int)nblk);
    pw_note_kernel("k_conv3d_h2_s2<2>");
  }
  return PW_OK;
}
