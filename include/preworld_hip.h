/*
 * preworld_hip.h -- C ABI of libpreworld_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the native ops on PreWorld's camera->voxel
 * occupancy hot path.  Plain pointers and sizes only: every pointer is a DEVICE
 * pointer unless the parameter name ends in _host; `stream` is a hipStream_t passed
 * as void* (NULL = the null stream).  The caller owns all memory; kernels write in
 * place; nothing here allocates, synchronises or copies to the host.
 *
 * Return value: 0 on success, a negative PW_E* code on failure; pw_last_error()
 * returns a thread-local message.  (The reference's pybind ops throw C++ exceptions
 * -- mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:30-57 does no checking at all; the
 * Python layer in preworld_amd/_lib.py turns a non-zero code into RuntimeError.)
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   pw_bev_pool_v2_forward / _backward  <- mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:30-57,74-104
 *                                          (kernels bev_pool_cuda.cu:21-48,67-121)
 *   pw_lss_* (geometry, rank build)     <- mmdet3d/models/necks/view_transformer.py:114-153,203-261
 *   pw_conv3d_* / pw_fpn3d_*            <- torch Conv3d/BatchNorm3d/Upsample as composed in
 *                                          mmdet3d/models/backbones/resnet.py:88-184,
 *                                          mmdet3d/models/necks/lss_fpn.py:103-148
 *   pw_forecast_* / pw_occ_head_*       <- mmdet3d/models/detectors/preworld_temporal_traj.py:303-368,
 *                                          mmdet3d/models/heads/occupancy_head.py:124-177
 *   pw_raw2alpha*, pw_alpha2weight*     <- mmdet3d/models/nerf/cuda/render_utils.cpp:120-167
 *                                          (kernels render_utils_kernel.cu:431-443,507-517,577-677)
 *   pw_cumdist_thres                    <- mmdet3d/models/nerf/cuda/ub360_utils.cpp:15-18
 *                                          (kernel ub360_utils_kernel.cu:13-47)
 *   pw_render_*                         <- mmdet3d/models/nerf/nerf_head.py:32-55,165-353
 */
#ifndef PREWORLD_HIP_H_
#define PREWORLD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PW_OK 0
#define PW_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define PW_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define PW_ENOSPC (-3)   /* workspace too small */
#define PW_EUNSUP (-4)   /* configuration not supported by the kernels */

int pw_version(void);
const char* pw_build_id(void);   /* hash of the sources the library was built from (preworld_amd.build.source_hash) */
const char* pw_last_error(void);
/* name, as rocprofv3 --kernel-trace prints it, of the dominant kernel the calling thread's last pw_* compute call
 * launched (which variant the library picked); measurement aid for bench.py's roofline object */
const char* pw_last_kernel(void);
/* measurement aid (bench.py): TFLOP/s a bare v_mfma_f32_32x32x16_f16 stream sustains on the current device for `seconds`
 * (<= 10) of wall time -- random fp16 operands in registers, no memory traffic, one wave per SIMD on every CU: what the socket's
 * power cap leaves of the 2.5 PFLOP/s data-sheet peak (profiles/r03_power_wall.txt).  Synchronous, default stream. */
int pw_probe_mfma_f16(double seconds, double* tflops);
/* device the library was built for / runs on: fills CU count, returns 0 */
int pw_device_info(int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);

/* ---------------------------------------------------------------------------------------
 * A2  camera matrices: inv(post_rots), combine = R(sensor2ego) @ inv(cam2imgs), t
 * view_transformer.py:141-150 (the reference calls torch.inverse twice per forward).
 * BN = B*N cameras.  sensor2ego (BN,4,4), cam2imgs/post_rots (BN,3,3) -> (BN,3,3)x2,(BN,3) */
int pw_lss_camera_matrices(int BN, const float* sensor2ego, const float* cam2imgs,
                           const float* post_rots, float* inv_post_rot, float* combine,
                           float* trans, void* stream);

/* A2+A3a  frustum point -> ego coordinate -> voxel id (or -1 when outside the grid).
 * frustum (D,H,W,3); inv_post_rot/combine (B*N,3,3); post_trans/trans (B*N,3); bda (B,3,3).
 * lower3_host / interval3_host: 3 floats each, HOST memory (grid_lower_bound, grid_interval).
 * vox: int32[B*N*D*H*W].  coor_out: float[B*N*D*H*W*3] (the reference's `coor`) or NULL. */
int pw_lss_voxel_index(int B, int N, int D, int H, int W, const float* frustum,
                       const float* inv_post_rot, const float* post_trans, const float* combine,
                       const float* trans, const float* bda, const float* lower3_host,
                       const float* interval3_host, int gx, int gy, int gz, int32_t* vox,
                       float* coor_out, void* stream);

/* A3b  stable segmented sort of point ids by key (key<0 = dropped).  Builds what
 * voxel_pooling_prepare_v2 (view_transformer.py:239-261) and the backward's re-sort
 * (bev_pool.py:47-57) need.  Outputs:
 *   seg_start  int32[n_keys+1]  dense exclusive prefix (seg_start[n_keys] = kept count)
 *   order      int32[n]         order[0..kept) = original indices, ascending inside a segment
 *   order_aux  int32[n] or NULL order_aux[pos] = (id/aux_div)*aux_mod + id%aux_mod for id=order[pos]
 *                               (LSS: aux_div=D*H*W, aux_mod=H*W gives ranks_feat, :219-224)
 *   long_list  int32[n/(long_threshold+1)+1] or NULL: keys whose segment has more than
 *              long_threshold entries (unordered), count in n_long (device int32)
 * workspace: pw_segment_sort_workspace_bytes(n, n_keys) bytes, 256-byte aligned. */
size_t pw_segment_sort_workspace_bytes(int64_t n, int64_t n_keys);
int pw_segment_sort(int64_t n, int64_t n_keys, const int32_t* keys, void* workspace,
                    size_t workspace_bytes, int32_t* seg_start, int32_t* order, int aux_div,
                    int aux_mod, int32_t* order_aux, int long_threshold, int32_t* long_list,
                    int32_t* n_long, void* stream);

/* A3c  expand the sort into the reference's five tensors (view_transformer.py:246-261):
 * ranks_bev/ranks_depth/ranks_feat int32[>=kept], interval_starts/lengths int32[>=n_intervals]
 * (compacted over non-empty voxels), counts int32[2] = {kept, n_intervals} (device).
 * D, HW: depth bins and H*W of the image-view feature (ranks_feat drops the depth axis). */
size_t pw_lss_ranks_workspace_bytes(int64_t n_voxels);
int pw_lss_ranks(int64_t n_voxels, const int32_t* seg_start, const int32_t* order, int D, int HW,
                 void* workspace, size_t workspace_bytes, int32_t* ranks_bev,
                 int32_t* ranks_depth, int32_t* ranks_feat, int32_t* interval_starts,
                 int32_t* interval_lengths, int32_t* counts, void* stream);

/* A4  bev_pool_v2 forward, reference ABI (bev_pool.cpp:30-57; note lengths BEFORE starts).
 * depth (B,N,D,H,W) flat, feat (B,N,H,W,C) flat, out (B,Z,Y,X,C) PRE-ZEROED by the caller. */
int pw_bev_pool_v2_forward(const float* depth, const float* feat, float* out,
                           const int32_t* ranks_depth, const int32_t* ranks_feat,
                           const int32_t* ranks_bev, const int32_t* interval_lengths,
                           const int32_t* interval_starts, int c, int n_intervals, void* stream);

/* A5  bev_pool_v2 backward, reference ABI (bev_pool.cpp:74-104); intervals are per feat pixel
 * (bev_pool.py:47-57); depth_grad/feat_grad pre-zeroed by the caller. */
int pw_bev_pool_v2_backward(const float* out_grad, float* depth_grad, float* feat_grad,
                            const float* depth, const float* feat, const int32_t* ranks_depth,
                            const int32_t* ranks_feat, const int32_t* ranks_bev,
                            const int32_t* interval_lengths, const int32_t* interval_starts, int c,
                            int n_intervals, void* stream);

/* A4 (fused fast path)  dense voxel-driven pooling: every voxel of out (n_voxels, C),
 * channels-last = (B,Z,Y,X,C), is written exactly once (sum or zero) -- no memset, no permute.
 * seg_start/order/order_feat(=order_aux)/long_list/n_long come from pw_segment_sort over the
 * voxel ids (long_list/n_long may be NULL).  Sums run in ascending point order per voxel.
 * out_h2 != 0 (C % 32 == 0, 16-byte aligned feat / out): the same fp32 sums are written in split-fp16 "h2" storage under the
 * range slot out_rng (see pw_f32_to_h2; NULL = exponent 0, nothing recorded). */
int pw_bev_pool_dense(const float* depth, const float* feat, const int32_t* seg_start,
                      const int32_t* order, const int32_t* order_feat, int64_t n_voxels, int c,
                      int long_threshold, const int32_t* long_list, const int32_t* n_long,
                      float* out, int out_h2, int32_t* out_rng, void* stream);

/* A1-A4 in one call (inference, C == 32): camera tensors + frustum + (depth, feat) of one batch of frames -> pooled grid
 * out (B,Z,Y,X,C), every voxel written (sum in ascending point order, or zero); fp32 or (out_h2) split-fp16 under out_rng.
 * Same results, bit for bit, as pw_lss_camera_matrices + pw_lss_voxel_index + pw_segment_sort + pw_bev_pool_dense
 * (view_transformer.py:114-153, :203-261; bev_pool_cuda.cu:21-48) in 5 launches instead of 11: a voxel's first eight points are
 * listed by the counting atomics themselves and sorted inside the pooling sweep; only voxels with more points take a scatter.
 * sensor2ego (B,N,4,4), cam2imgs / post_rots (B,N,3,3), post_trans (B,N,3), bda (B,3,3), frustum (D,H,W,3), depth (B,N,D,H,W),
 * feat (B,N,H,W,C) device; lower3_host / interval3_host host float[3].  workspace: 256-byte aligned scratch, no state kept. */
size_t pw_lss_lift_pool_workspace_bytes(int64_t n_points, int64_t n_voxels, int BN);
int pw_lss_lift_pool(int B, int N, int D, int H, int W, const float* frustum, const float* sensor2ego,
                     const float* cam2imgs, const float* post_rots, const float* post_trans, const float* bda,
                     const float* lower3_host, const float* interval3_host, int gx, int gy, int gz, const float* depth,
                     const float* feat, int c, void* workspace, size_t workspace_bytes, float* out, int out_h2,
                     int32_t* out_rng, void* stream);

/* ---------------------------------------------------------------------------------------
 * A6-A9, A11  3-D convolution on channels-last activations, exact-fp32 MFMA implicit GEMM.
 * Replaces torch Conv3d(+BatchNorm3d eval)(+residual)(+ReLU) as composed by
 * mmdet3d/models/backbones/resnet.py:88-184, necks/lss_fpn.py:120-129,
 * detectors/preworld.py:72-79, heads/occupancy_head.py:80-105.
 *   x        (B, D, H, W, Cin) fp32, Cin % 32 == 0
 *   wpk      packed weights, float[Cin/32][ksize^3][cout_total/32][4][64][4] with
 *            wpk[ch][tap][nt][q][h*32+j][e] = w[nt*32+j][ch*32+h*16+4*q+e][tap]  (w = torch
 *            (Cout,Cin,kD,kH,kW); columns >= Cout zero) -- built by preworld_amd.ops.pack_conv_weight.
 *            (A 4096-byte tile is four 1024-byte pieces of 64 lanes x 16 B, so that one wave-wide 16-byte load
 *            reads 8 consecutive cache lines.)
 *   scale,bias  float[cout_total] or NULL: y = acc*scale + bias  (BatchNorm eval folded, or conv bias)
 *   residual    same layout as y0 or NULL (added before ReLU; BasicBlock3D.forward resnet.py:120-123)
 *   y0 (B,Do,Ho,Wo,cout0) gets packed columns [0,cout0); y1 (B,Do,Ho,Wo,cout1) gets columns
 *   [roundup32(cout0), +cout1) or is NULL -- lets conv1 and downsample of a BasicBlock3D share
 *   one pass over x.  ld_y0 / ld_y1: floats between consecutive voxels of y0 (and residual) / y1;
 *   0 = dense (cout0 / cout1); larger when the output is a channel slice of a wider channels-last
 *   buffer (the [adjacent, key] concat of bevdet_occ.py:266 is produced in place this way).
 *   ksize in {1,2,3} (pad = (ksize-1)/2; ksize 2 needs stride 2), stride in {1,2}.
 *   algo: 0 auto; the others force one kernel (tests, A/B): 1 tile-per-block LDS kernel (k3 s1), 2 gather kernel, 3 gather kernel
 *   with the input-channel chunks split over the 4 waves of a block and summed in LDS in a fixed order (small grids), 4 persistent
 *   DMA-pipelined LDS kernel (k3 s1). */
int pw_conv3d_ndhwc(const float* x, const float* wpk, const float* scale, const float* bias,
                    const float* residual, float* y0, float* y1, int B, int D, int H, int W,
                    int Cin, int cout_total, int cout0, int cout1, int ld_y0, int ld_y1, int ksize,
                    int stride, int relu0, int relu1, int algo, void* stream);

/* A8  LSSFPN3D fused (mmdet3d/models/necks/lss_fpn.py:132-148): out = ReLU(BN(W8 x8 +
 * up2(y16) + up4(y32))) where y16/y32 are the 1x1x1 conv already applied at 1/2 and 1/4
 * resolution (32 channels each; interpolation and 1x1x1 conv commute).  trilinear,
 * align_corners=True.  x8 (B,D,H,W,Cin8), wpk8 packed [Cin8/32][1][1][4][64][4], out (B,D,H,W,32).
 * x_h2 != 0: x8 is in split-fp16 "h2" storage and wpk8 comes from pack_conv_weight_h2 (its inv_scale folded into `scale`);
 * out_h2 != 0: out is written in h2 storage (see pw_conv3d_h2).  y16 / y32 are always fp32.  x_rng / out_rng: range slots
 * of x8 / out when they are h2 (see pw_f32_to_h2; NULL = exponent 0). */
int pw_fpn3d_fuse(const float* x8, const float* wpk8, const float* y16, const float* y32,
                  const float* scale, const float* bias, float* out, int B, int D, int H, int W,
                  int Cin8, int D2, int H2, int W2, int D4, int H4, int W4, int relu, int x_h2, int out_h2,
                  const int32_t* x_rng, int32_t* out_rng, void* stream);

/* 3x3x3 stride-1 pad-1 convolution by Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores: same contract
 * as pw_conv3d_ndhwc (scale/bias, residual, ReLU, two destinations, row strides) with weights in the
 * transform domain: uwpk = float[Cin/32][64 points][cout_total/16][64 lanes][8],
 *   uwpk[ch][p][n16][g*16+j][s] = U[n16*16+j][ch*32+g*8+s][p],  U = G w G^T along d, h, w
 *   (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], p = i_d*16 + i_h*4 + i_w) -- preworld_amd.ops.pack_conv_weight_wino.
 * 3.375x fewer multiplies than the direct sum; results differ from it by fp32 rounding only.
 * cout_total: any multiple of 32 (work items = 4x8x8 tile x group of 32 or 64 columns of the wave-specialised
 * persistent kernel; PW_WINO_WS=0 selects the tile-per-block kernel, which takes 32 or 64). */
int pw_conv3d_wino(const float* x, const float* uwpk, const float* scale, const float* bias,
                   const float* residual, float* y0, float* y1, int B, int D, int H, int W, int Cin,
                   int cout_total, int cout0, int cout1, int ld_y0, int ld_y1, int relu0, int relu1,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * Split-fp16 ("h2") path of the voxel encoder: fp32-level accuracy on the fp16 matrix cores.
 * x = hi + lo with hi = fp16(x), lo = fp16(x - hi); a product block is three v_mfma_f32_32x32x16_f16
 * (hi.hi + lo.hi + hi.lo) with fp32 accumulation (preworld_amd/csrc/pw_h2.h; measured error = that of an fp32 FMA chain).
 * An h2 tensor has the shape and byte size of its fp32 counterpart ((.., C) channels-last, C % 32 == 0); each 32-channel
 * chunk of a voxel is 8 slots of 16 bytes, slot 4*half + 2*ks + p = plane p (0 hi, 1 lo) of channels 16*ks + 8*half + 0..7.
 * pw_f32_to_h2 / pw_h2_to_f32 convert (n_vox, C) rows with row strides ld_x / ld_y floats (0 = dense).
 *
 * RANGE SLOTS.  fp16 has 5 exponent bits, the reference's fp32 (backbones/resnet.py:88-123: plain Conv3d) has 8.  Every h2 tensor
 * therefore carries a per-tensor power-of-two exponent in a range slot `rng` = int32[2] in device memory:
 *     value = (hi + lo) * 2^rng[0];   rng[1] = bit pattern of the largest |value| (true units, a float >= 0; a NaN pattern if a
 *     NaN was written) recorded by the kernels that wrote the tensor since the host last zeroed it.
 * Producers store value / 2^rng[0] (unsaturated: beyond +-65504 stored units the element becomes Inf and the slot records it) and
 * atomically raise rng[1]; consumers fold 2^rng[0] into their epilogue scale like the weights' pre-scale (powers of two, exact).
 * The host (preworld_amd.ops.RangeCtx) picks rng[0] so that the largest magnitude lands in [2^12, 2^13) stored units -- 22-bit
 * significands down to 2^-15 of the maximum, an absolute floor of 2^-38 of it below -- and re-runs a sample whose recorded
 * maxima left [2^6, 65504] stored units.  A NULL slot means exponent 0 and nothing recorded.
 * Layout: a slot is PW_RNG_ROW int32; behind rng[0..1] sit PW_RNG_WORDS partial maxima (from rng[PW_RNG_SCRATCH]) that the waves of
 * the producing kernels raise with return-less atomics -- thousands of atomics on one address would serialise at ~100 ns each.
 * pw_rng_fold(tab, n_slots, compact): rng[1] = max(rng[1], partials), partials cleared, for n_slots consecutive slots, and (if
 * compact != NULL) the (n_slots, 2) pairs [exponent, maximum] copied there contiguously -- one small launch at the end of a pass.
 * pw_rng_audit(compact, n_slots, sticky): the steady-state test ON THE DEVICE, for passes nobody waits for (graph replays in
 * flight): sticky[0] += slots of this pass whose recorded maximum left [2^6, 65504] stored units (or is not finite), sticky[1] += 1
 * if there was one, sticky[2] += 1 (passes audited).  The counters are never cleared by the library: the host reads them whenever
 * it likes and knows whether EVERY pass since stayed inside its calibrated ranges.
 * pw_f32_to_h2: auto_exp != 0 first derives rng[0] from the largest finite |x| (three extra small launches) instead of
 * taking it as it is. */
#define PW_RNG_ROW 1056
#define PW_RNG_SCRATCH 32
#define PW_RNG_WORDS 1024
int pw_rng_fold(int32_t* tab, int n_slots, int32_t* compact, void* stream);
int pw_rng_audit(const int32_t* compact, int n_slots, int32_t* sticky, void* stream);
int pw_f32_to_h2(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, int32_t* rng, int auto_exp, void* stream);
int pw_h2_to_f32(const float* x, float* y, int64_t n_vox, int C, int ld_x, int ld_y, const int32_t* rng, void* stream);

/* Convolution with split-fp16 operands, same contract as pw_conv3d_ndhwc -- scale/bias, residual, ReLU, two destinations,
 * row strides -- for ksize 3 (stride 1: persistent LDS-tiled kernel; stride 2: LDS-tiled k_conv3d_h2_s2 for the shapes it is built
 * for, else the gather kernel) and ksize 1 (stride 1); algo 0 auto, algo 2 / 3 force the gather kernel (3 = input-channel chunks
 * split over the 4 waves, tiny grids), with
 *   x        (B, D, H, W, Cin) in h2 storage;
 *   wpk      split-fp16 packed weights float[Cin/32][ksize^3][cout_total/32][4 pieces][64 lanes][4]: piece q = 2*ks + p of
 *            lane l (j = l & 31, h = l >> 5) holds the 8 halves plane p of S[n] * w[n = nt*32 + j][c = ch*32 + 16*ks + 8*h + 0..7][tap],
 *            S[n] a power of two that the caller folds back into scale[n] (preworld_amd.ops.pack_conv_weight_h2);
 *   cout0 / cout1 / ld_y0 / ld_y1 multiples of 32;
 *   fmt_y0 / fmt_y1 / fmt_res: 0 = fp32, 1 = h2 storage of y0 / y1 / residual (the residual shares y0's row stride and
 *   may be y0 itself);
 *   x_rng / res_rng / y0_rng / y1_rng: range slots (see pw_f32_to_h2) of the operands that are in h2 storage, NULL otherwise. */
int pw_conv3d_h2(const float* x, const float* wpk, const float* scale, const float* bias, const float* residual,
                 float* y0, float* y1, int B, int D, int H, int W, int Cin, int cout_total, int cout0, int cout1,
                 int ld_y0, int ld_y1, int ksize, int stride, int relu0, int relu1, int algo, int fmt_y0, int fmt_y1,
                 int fmt_res, const int32_t* x_rng, const int32_t* res_rng, int32_t* y0_rng, int32_t* y1_rng, void* stream);


/* A11  OccHead fused (mmdet3d/models/heads/occupancy_head.py:124-177, num_level=1,
 * use_deblock=False): conv3x3x3 Cin->16 + BN + ReLU, 1x1x1 16->8 + BN + ReLU, 1x1x1 8->18,
 * argmax -> uint8, in one kernel.  x (B,D,H,W,Cin); scale/bias float[>=16] (folded BN of
 * occ_convs.0.1); w1 [8][16], s1/b1 [8] (folded BN of occ_pred_conv.1), w2 [18][8]; occ
 * uint8[B*D*H*W]; logits float[B*D*H*W][18] or NULL; geo uint8[B*D*H*W] or NULL receives the
 * reference's geo_occ (preworld_temporal_traj.py:313-319): 0 where occ != empty_idx, n_cls-1 elsewhere.
 * wpk_layout 16: wpk = float[Cin/32][27][64][8], wpk[ch][tap][g*16+j][s] = w[j][ch*32+g*8+s][tap]
 *   (direct-form v_mfma_f32_16x16x4_f32 kernel, no padded output columns);
 * wpk_layout 64: Cin == 32, wpk = pw_conv3d_wino's transform-domain layout with cout_total = 16
 *   (float[1][64 points][1][64 lanes][8]): the Winograd kernel k_occ_head_wino, 1.4x faster at 16x200x200. */
int pw_occ_head_fused(const float* x, const float* wpk, const float* scale, const float* bias,
                      const float* w1, const float* s1, const float* b1, const float* w2,
                      uint8_t* occ, float* logits, uint8_t* geo, int empty_idx, int B, int D, int H,
                      int W, int Cin, int n_mid, int n_hid, int n_cls, int wpk_layout, void* stream);

/* A11 on the fp16 matrix cores (split-fp16 operands, pw_h2 storage; same reference lines as pw_occ_head_fused):
 * x (B,D,H,W,32) in h2 storage; wpk = 27 taps x {hi, lo} x 64 lanes x 8 fp16 (55 296 bytes):
 *   wpk[tap][p][g*16 + j][e] = plane p of S_j * w[j][16*(g>>1) + 8*(g&1) + e][tap], S_j the per-output-channel power of two
 *   that puts max |w[j]| in [512, 1024); scale [16] = folded BN scale / S_j, bias [16];
 * tailpk = 800 floats (preworld_amd.ops.pack_occ_tail_h2): six fragments [64 lanes][4 fp16] = {hi, lo} of S1 * W1 (rows = 8
 *   hidden channels padded to 16, lane (row = l & 15, k = 4 (l >> 4) + e)), {hi, lo} of S2 * W2 rows 0..15 and rows 16..17
 *   (k = hidden channel, padded to 16), then s1 / S1 and b1 padded to 16 floats each (folded BN of occ_pred_conv.1);
 *   inv2 = 1 / S2 rescales the logits output.  Outputs as pw_occ_head_fused.
 * x_rng: range slot of x (see pw_f32_to_h2).  The two hidden layers are split in registers, in units chosen from a-priori bounds:
 *   |mid| <= mid_a * 2^(16 + x_rng[0]) + mid_b with mid_a = max_c |BN scale_c| * ||w[c]||_1, mid_b = max_c |bias_c|, and
 *   |hid| <= hid_a * max|mid| + hid_b with hid_a = max_r |s1_r| * ||W1[r]||_1, hid_b = max_r |b1_r| (preworld_amd.ops computes them).
 * Kernel k_occ_head_h2<LOGITS>: v_mfma_f32_16x16x32_f16 with all conv weights register-resident; the tail runs on
 * v_mfma_f32_16x16x16_f16 in registers, as a phase behind each tile's tap loop. */
int pw_occ_head_h2(const float* x, const float* wpk, const float* scale, const float* bias, const float* tailpk, float inv2,
                   uint8_t* occ, float* logits, uint8_t* geo, int empty_idx, int B, int D, int H, int W, int Cin, int n_mid,
                   int n_hid, int n_cls, const int32_t* x_rng, float mid_a, float mid_b, float hid_a, float hid_b, void* stream);
/* the same with the uint8 grids written through caller-given BYTE strides (b, d, h, w) -- e.g. a (Z,Y,X) result stored as the (X,Y,Z)-contiguous
 * array detectors/preworld_temporal_traj.py:311-366 hands out, directly into a row of the host-payload buffer (no transposing copy);
 * out_strides4_host NULL = contiguous; out_span_bytes = bytes from occ / geo to the end of their buffer (store bounds).  logits stay contiguous. */
int pw_occ_head_h2_strided(const float* x, const float* wpk, const float* scale, const float* bias, const float* tailpk,
                           float inv2, uint8_t* occ, float* logits, uint8_t* geo, const int64_t* out_strides4_host,
                           int64_t out_span_bytes, int empty_idx, int B, int D, int H, int W, int Cin, int n_mid, int n_hid,
                           int n_cls, const int32_t* x_rng, float mid_a, float mid_b, float hid_a, float hid_b, void* stream);

/* A10  state-conditioned forecast (mmdet3d/models/detectors/preworld_temporal_traj.py:329-368).
 * pw_forecast_pack: fusion_head.{0,2}.weight ([128][64], [32][128]) -> per-lane MFMA operand
 *   order, w1p/w2p float[4096] each (once per weight update).
 * pw_forecast_prologue: per sample, e = plan_head(ego) (21->256 ReLU->256 ReLU->32) and the
 *   hoisted ego term c1 = fusion_head.0.weight[:,32:] e + fusion_head.0.bias;
 *   ego (n_samples, ego_dim), ego_feat (n_samples,32), c1 (n_samples,128) in natural order and
 *   c1p (n_samples,128) in the MFMA accumulator order pw_forecast_steps consumes.
 * pw_forecast_steps: v_{k+1} = v_k + W2 softplus(W1a v_k + c1) + b2 for k < n_steps, all steps
 *   in registers; v0 (n_samples*n_vox, 32) channels-last; states float[n_steps][n_samples*n_vox][32]. */
int pw_forecast_pack(const float* fusion_w1, const float* fusion_w2, float* w1p, float* w2p,
                     void* stream);
int pw_forecast_prologue(const float* ego, int n_samples, int ego_dim, const float* plan_w0,
                         const float* plan_b0, const float* plan_w2, const float* plan_b2,
                         const float* plan_w4, const float* plan_b4, const float* fusion_w1,
                         const float* fusion_b1, float* ego_feat, float* c1, float* c1p,
                         void* stream);
int pw_forecast_steps(const float* v0, int64_t n_vox_per_sample, int n_samples, const float* w1p,
                      const float* w2p, const float* c1p, const float* fusion_b2, int n_steps,
                      float* states, void* stream);
/* pw_forecast_steps on the fp16 matrix cores with split-fp16 operands (see pw_conv3d_h2): w1p / w2p are the split weights
 * float[4 tiles][2 k-blocks][2 planes][64 lanes][4] built by preworld_amd.ops.forecast_pack_h2 with power-of-two pre-scales
 * whose inverses are inv1 / inv2; everything else as pw_forecast_steps.  v0_h2 / out_h2 != 0: v0 is read / the states are
 * written in h2 storage (same 4 bytes per element, see pw_f32_to_h2) instead of fp32 -- what pw_conv3d_h2 writes and
 * pw_occ_head_h2 reads.  v0_rng: range slot of an h2 v0.  states_rng: range slot the recursion runs under -- the units of the
 * split operands of every step and of h2 states (fp32 states are written in true units; the largest state magnitude is recorded
 * either way).  w1_l1max = max row L1 norm of fusion_head.0.weight[:, :32] bounds the hidden activations a priori. */
int pw_forecast_steps_h2(const float* v0, int64_t n_vox_per_sample, int n_samples, const float* w1p,
                         const float* w2p, float inv1, float inv2, const float* c1p, const float* fusion_b2,
                         int n_steps, float* states, int v0_h2, int out_h2, const int32_t* v0_rng, int32_t* states_rng,
                         float w1_l1max, void* stream);

/* ---------------------------------------------------------------------------------------
 * A14/A16/A17  render ops with the reference's semantics on compacted point arrays
 * (mmdet3d/models/nerf/cuda/render_utils.cpp:120-167, ub360_utils.cpp:15-18; kernels
 * render_utils_kernel.cu:431-443,507-517,577-677, ub360_utils_kernel.cu:13-47).  fp32 only
 * (the reference dispatches float/double; every call site passes float).
 *   pw_raw2alpha:           exp_d = exp(density+shift); alpha = 1-(1+exp_d)^(-interval)
 *   pw_raw2alpha_backward:  grad = min(exp_d,1e10)*(1+exp_d)^(-interval-1)*interval*grad_back
 *   pw_alpha2weight:        ray_id int64[n_pts] sorted; fills weight/T [n_pts], alphainv_last
 *                           [n_rays], i_start/i_end int64[n_rays] (all initialised here)
 *   pw_alpha2weight_backward, pw_cumdist_thres (dist (n_rays,n_pts) -> uint8 mask) */
int pw_raw2alpha(const float* density, float shift, float interval, int64_t n, float* exp_d,
                 float* alpha, void* stream);
int pw_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval, int64_t n,
                          float* grad, void* stream);
int pw_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int n_rays,
                    float* weight, float* T, float* alphainv_last, int64_t* i_start,
                    int64_t* i_end, void* stream);
int pw_alpha2weight_backward(const float* alpha, const float* weight, const float* T,
                             const float* alphainv_last, const int64_t* i_start,
                             const int64_t* i_end, int n_rays, const float* grad_weights,
                             const float* grad_last, int64_t n_pts, float* grad, void* stream);
int pw_cumdist_thres(const float* dist, float thres, int n_rays, int n_pts, uint8_t* mask,
                     void* stream);

/* A13-A18 fused forward of NerfHead.render_one_scene + render_depth/semantic/color
 * (mmdet3d/models/nerf/nerf_head.py:32-55,165-269,331-353), one wavefront per ray.
 *   rays_o/rays_d (n_rays,3); t float[n_samples] (<= 448); grid = packed attribute grid
 *   (Z,Y,X,grid_channels) channels-last holding sigma at c_sigma, n_sem(=17) semantic logits at
 *   c_sem.., rgb at c_rgb..; consts_host = 27 HOST floats: scene_center[3], scene_radius[3],
 *   bda[9], xyz_min[3], xyz_max[3], bg_len, act_shift, interval, dist_thres, fast_color_thres,
 *   depth_scale(=radius).
 *   outputs: out_depth (n_rays), out_sem (n_rays,17), out_rgb (n_rays,3), out_last (n_rays) =
 *   alphainv_last; optional out_counts int32 (n_rays,3) = #samples after each of the reference's
 *   three compactions, out_weights (n_rays,n_samples) dense weights (0 where culled), out_mask
 *   uint8 (n_rays,n_samples) the inner|cumdist mask.  *   grid_bf16 != 0: `grid` points at bfloat16 values (same (Z,Y,X,grid_channels) layout, half the bytes per trilinear corner;
 *   BASELINE configs[4] "bf16 storage"): widened to fp32 on load, fp32 arithmetic and accumulation throughout. */
int pw_render_rays(const float* rays_o, const float* rays_d, int n_rays, const float* t,
                   int n_samples, const float* grid, int X, int Y, int Z, int grid_channels,
                   int c_sigma, int c_sem, int n_sem, int c_rgb, const float* consts_host,
                   float* out_depth, float* out_sem, float* out_rgb, float* out_last,
                   int32_t* out_counts, float* out_weights, uint8_t* out_mask, int grid_bf16, void* stream);

/* Backward of pw_render_rays in ONE kernel (reference: torch autograd through nerf_head.py:211-225 grid_sample, utils.py:37-68
 * Raw2Alpha / Alphas2Weights with render_utils_kernel.cu:507-517,654-677, and the three segment sums :331-353): the same
 * per-ray march (identical masks / compactions / early stop), the reverse transmittance scan, and the trilinear corner
 * scatter-adds into grad_grid (Z,Y,X,grid_channels) -- ACCUMULATED with float atomics, zero it first.
 * g_depth (R), g_sem (R,n_sem), g_rgb (R,3), g_last (R) = d loss / d {depth, semantic, color, alphainv_last};
 * g_weights (R,n_samples) = d loss / d out_weights (the dense per-sample weights) or NULL. */
int pw_render_rays_backward(const float* rays_o, const float* rays_d, int n_rays, const float* t, int n_samples,
                            const float* grid, int X, int Y, int Z, int grid_channels, int c_sigma, int c_sem,
                            int n_sem, int c_rgb, const float* consts_host, const float* g_depth, const float* g_sem,
                            const float* g_rgb, const float* g_last, const float* g_weights, float* grad_grid,
                            void* stream);
/* The same gradient WITHOUT float atomics, bit-reproducible from run to run (the autograd graph it replaces, nerf_head.py:211-225
 * under torch, is deterministic given sorted ray_id): the march emits one (voxel, ray, w*corner weight, d sigma*corner weight)
 * entry per visited sample and in-bounds corner, a counting sort groups the entries by voxel, and one wave per voxel sums
 * [d sigma | w g_sem[ray] | w g_rgb[ray]] in 64-bit fixed point (exact integer sums: order-free), written by a single writer.
 * g_semrgb: (R, 20) rows [g_sem | g_rgb]; g_absmax: device scalar max |g_semrgb|; workspace: pw_render_backward_workspace_bytes,
 * 256-byte aligned, 40 bytes per entry the arrays are laid out for.  max_entries: an upper bound of the entries the march will emit
 * -- 8 x the number of samples above the alpha threshold, i.e. 8 x the sum of pw_render_rays' out_counts[:, 1], is one -- or 0 for
 * the worst case n_rays*n_samples*8 (5.1 GB at 38 400 x 417; the forward's count brings it to a few hundred MB).  A bound that turns
 * out too small is a caller bug: nothing is written out of range, grad_grid[0] becomes NaN.
 * Exactness: the fixed-point units leave room for 2^22 largest-magnitude terms per voxel and channel group (a voxel near the cameras
 * sees 10^4..10^5); non-finite upstream gradients are not representable in them and count as 0 (the float-atomics entry point
 * propagates them). */
size_t pw_render_backward_workspace_bytes(int n_rays, int n_samples, int X, int Y, int Z, int64_t max_entries);
int pw_render_rays_backward_sorted(const float* rays_o, const float* rays_d, int n_rays, const float* t, int n_samples,
                                   const float* grid, int X, int Y, int Z, int grid_channels, int c_sigma, int c_sem,
                                   int n_sem, int c_rgb, const float* consts_host, const float* g_depth, const float* g_sem,
                                   const float* g_rgb, const float* g_last, const float* g_weights,
                                   const float* g_semrgb, const float* g_absmax, void* workspace, size_t workspace_bytes,
                                   int64_t max_entries, float* grad_grid, void* stream);

/* A12  attribute projection (preworld_temporal_traj.py:81-104): density/semantic/color MLPs
 * (each 32 -> 64 Softplus -> {2,17,3}) fused; v0 (n_vox,32) channels-last; out (n_vox,24) packed
 * {density_prob[2], semantic[17], color[3], 0, 0}.  w1p/w2p: float[6144] per-lane MFMA operand
 * order, b1p float[2][96] accumulator order, b2 float[32] (built by preworld_amd.ops.pack_attr_mlp).
 * final_softplus: apply Softplus to the two density outputs (final_softplus=True configs). */
int pw_attr_mlp(const float* v0, int64_t n_vox, const float* w1p, const float* w2p,
                const float* b1p, const float* b2, int final_softplus, float* out, void* stream);

/* A22  confusion matrix of the occupancy metric (mmdet3d/datasets/occ_metrics.py:82-105):
 * hist[n_cl*gt + pred] += 1 over voxels with gt < n_cl (and mask != 0 when mask is given).
 * pred/gt/mask uint8[n]; hist int64[n_cl*n_cl] ACCUMULATES (zero it for a fresh metric). */
int pw_confusion_hist(const uint8_t* pred, const uint8_t* gt, const uint8_t* mask, int64_t n,
                      int n_cl, int64_t* hist, void* stream);

/* nn.Softplus(beta=1, threshold=20) elementwise (the activation inside fusion_head and the
 * attribute MLPs, preworld_temporal_traj.py:81-132), same device function as the fused kernels. */
int pw_softplus(const float* x, float* y, int64_t n, void* stream);

/* A20  trajectory branch, train-time only (preworld_temporal_traj.py:464-470).
 * pw_global_avgpool_ndhwc: nn.AdaptiveAvgPool3d((1,1,1)) at the end of DownScaleModule3DCustom
 * (mmdet3d/models/heads/occupancy_head.py:180-200): x (B, n_vox, C) channels-last -> y (B, C),
 * sequential voxel order per channel.  The three Conv3d(k=2, s=2, bias) in front of it are
 * pw_conv3d_ndhwc with ksize=2, stride=2.
 * pw_linear_act: one nn.Linear (+activation) of ego_fusion_head / traj_head
 * (preworld_temporal_traj.py:136-150): y[r][o] = act(b[o] + sum_k x[r][k] * w[o][k]), w row-major
 * (out, in) as in the state dict; act 0 none, 1 ReLU, 2 Softplus(beta=1, threshold=20). */
int pw_global_avgpool_ndhwc(const float* x, int B, int64_t n_vox, int C, float* y, void* stream);
int pw_linear_act(const float* x, const float* w, const float* b, float* y, int rows, int n_in,
                  int n_out, int act, void* stream);

/* SURVEY 8f row 1 (first half): the DepthNet tail, mmdet3d/models/necks/view_transformer.py:797-801:
 *   depth_digit = x[:, :D]; tran_feat = x[:, D:D+C]; depth = depth_digit.softmax(dim=1)
 * plus the (B,N,H,W,C) re-layout bev_pool_v2 makes of tran_feat (view_transformer.py:189), in one
 * pass over the DepthNet output.
 *   x         float[BN][D + C (+ rest)][HW]  DepthNet output, channel stride HW; x_channels = its channel count
 *   depth     float[BN][D][HW]               softmax over D per pixel (max-subtracted, like ATen)
 *   feat_cl   float[BN][HW][C]               context features, channels-last (what pooling gathers)
 * C must be a multiple of 4. */
int pw_depthnet_tail(const float* x, int BN, int x_channels, int D, int C, int HW, float* depth,
                     float* feat_cl, void* stream);

/* SURVEY 8f row 1 (second half): the DepthNet's stereo cost volume,
 * mmdet3d/models/necks/view_transformer.py:546-604 (gen_grid + calculate_cost_volumn), one kernel.
 *   prev, curr  stereo features of the previous / current frame, element (bn, c, y, x) at
 *               bn*s_bn + c*s_c + y*s_y + x*s_x (NCHW or channels_last storage; channels_last is the
 *               fast path), C % 4 == 0 channels, H x W = the cv_frustum's height x width
 *   ds[D], xs[W], ys[H]  the separable cv_frustum (depth bins; pixel coordinates in the wi x hi input image)
 *   inv_post_rot, combine (= k2s_sensor[:3,:3] @ inv(intrins)), trans (= k2s_sensor[:3,3]): from
 *               pw_lss_camera_matrices(k2s_sensor, intrins, post_rots); post_trans [BN][3], intrins, post_rots [BN][9]
 *   out         float[BN][D][H][W] = softmax_D( -(sum_c |curr - warp(prev)| + bias * [warp outside]) ) */
int pw_stereo_cost_volume(const float* prev, const float* curr, int BN, int C, int H, int W,
                          int64_t s_bn, int64_t s_c, int64_t s_y, int64_t s_x, const float* ds, int D,
                          const float* xs, const float* ys, const float* inv_post_rot,
                          const float* post_trans, const float* combine, const float* trans,
                          const float* intrins, const float* post_rots, float wi, float hi, float bias,
                          float* out, void* stream);

/* ray table + weighted-ray-sampling weights of the pre-train dataloader
 * (mmdet3d/datasets/ray.py:34-119), on the GPU.
 * pw_pts2ray (ray.py:34-55, one camera): for pixel p = coor[i] (x, y):
 *   dirs = ((x+0.5-K02)/K00, (y+0.5-K12)/K11, 1); rays_d = c2w[:3,:3] . dirs; rays_o = c2w[:3,3];
 *   viewdirs = rays_d/|rays_d|;  rays[i] = {x, y, depth, seg, rays_o[3], rays_d[3], viewdirs[3], img[3]}
 *   coor float[n][2], depth/seg float[n], img float[n][3], c2w float[16] (row-major 4x4, device),
 *   K float[9] (device), rays float[n][16].
 * pw_class_count (ray.py:94-97): counts[c] += #(rays[:,3] == c), c < n_cls; counts int64[n_cls] accumulates.
 * pw_wrs_weights (ray.py:99-112): weight = balance_weight[class] * (frame_id == 0 ? 1 : (class in
 *   dynamic_class ? weight_dyn : weight_adj)); rays float[n][16], balance_weight float[n_cls],
 *   dynamic_class int32[n_dyn], weights float[n]. */
int pw_pts2ray(const float* coor, const float* depth, const float* seg, const float* img,
               const float* c2w, const float* K, int64_t n, float* rays, void* stream);
int pw_class_count(const float* rays, int64_t n, int n_cls, int64_t* counts, void* stream);
int pw_wrs_weights(const float* rays, int64_t n, int frame_id, const float* balance_weight, int n_cls,
                   const int32_t* dynamic_class, int n_dyn, float weight_adj, float weight_dyn,
                   float* weights, void* stream);

/* voxel-grid training losses of mmdet3d/models/detectors/loss.py -- CE_ssc_loss
 * (:20-29, class-weighted cross entropy, ignore_index), sem_scal_loss (:32-80) and geo_scal_loss
 * (:83-113) as used by loss_voxel (preworld_temporal_traj.py:176-199) -- forward statistics in one
 * pass over the logits and the gradient w.r.t. the logits in a second pass.
 *   logits   float, element (b, c, x, y, z) at b*sb + c*sc + x*sx + y*sy + z*sz (any layout: the
 *            reference's (B,C,X,Y,Z) tensor or a channels-last buffer viewed that way)
 *   target   uint8[B][X][Y][Z] (255 = ignore), cam_mask uint8[B][X][Y][Z] or NULL
 *   n_cls <= 32 classes, class_weights float[n_cls] (CE), empty_idx = geo_scal's non_empty_idx
 * pw_voxel_loss_stats ACCUMULATES into stats (double[PW_VOXEL_LOSS_NSTATS], zero it first):
 *   [0] sum_{t!=ignore} w_t * (-log p_t)   [1] sum_{t!=ignore} w_t
 *   [2] N = #(t != ignore & cam)           [3+i] S_p[i] = sum_M p_i   [35+i] S_t[i] = #_M (t == i)
 *   [67+i] S_pt[i] = sum_M p_i [t == i]    (M = the N voxels)
 *   [99] I = sum nt*(1-p_e)  [100] sum (1-p_e)  [101] sum nt  [102] sum (1-nt)*p_e  [103] sum (1-nt)
 *        with nt = (t != empty_idx) & cam over ALL voxels (geo_scal_loss :93-110)
 * pw_voxel_loss_grad: grad_logits (same strides as logits) = d(loss)/d(logits) where
 *   d(loss)/d(p_i(v)) = [v in M] * (ga[i] + gb[i]*[t(v) == i]) + [i == empty_idx] * (gc0 + gc1*nt(v)),
 *   plus the cross-entropy part ce_scale * w_t * (p - onehot(t)) for t != ignore; the softmax Jacobian
 *   is applied in the kernel.  coef = float[2*32 + 3] = {ga[32], gb[32], gc0, gc1, ce_scale} (device).
 * pw_voxel_loss_finish: losses float[3] = {CE, sem_scal, geo_scal} from the sums (loss.py:58-80,104-113).
 * pw_voxel_loss_coef: coef for pw_voxel_loss_grad from the sums and the upstream gradients
 *   grad_losses float[3] (device) of the three losses. */
#define PW_VOXEL_LOSS_NSTATS 104
int pw_voxel_loss_stats(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                        const float* class_weights, int B, int n_cls, int X, int Y, int Z, int64_t sb,
                        int64_t sc, int64_t sx, int64_t sy, int64_t sz, int ignore_index, int empty_idx,
                        double* stats, void* stream);
int pw_voxel_loss_grad(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                       const float* class_weights, const float* coef, int B, int n_cls, int X, int Y,
                       int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy, int64_t sz,
                       int ignore_index, int empty_idx, float* grad_logits, void* stream);
int pw_voxel_loss_finish(const double* stats, int n_cls, float* losses, void* stream);
int pw_voxel_loss_coef(const double* stats, int n_cls, const float* grad_losses, float* coef, void* stream);

/* SURVEY 8f row 2, the finetune configs' other two voxel losses (preworld.py:146-155).
 * Same logits / target / cam_mask / stride conventions as pw_voxel_loss_*.
 *
 * CustomFocalLoss.forward (mmdet3d/models/loss_utils/focal_loss.py:206-262): over the valid voxels (target !=
 *   ignore_index, cam_mask) mean of sum_c class_weights[c] * r(x, y) * FL(logit_c, [target == c]) times loss_weight,
 *   FL = mmcv-full's sigmoid_focal_loss element (gamma, alpha), r = |(x - X/2, y - Y/2)| / max + 1 (:196-203; the
 *   reference hard-codes X = Y = 200).
 *   pw_focal_loss_stats ACCUMULATES into stats double[2] = {weighted sum, valid count} (zero it first);
 *   pw_focal_loss_finish: loss float[1] = loss_weight * stats[0] / stats[1];
 *   pw_focal_loss_grad: grad_logits (logits' strides) = grad_loss[0] (device float) * d(loss)/d(logits).
 *
 * lovasz_softmax (mmdet3d/models/detectors/lovasz_softmax.py:157-232; classes='present', per_image=False):
 *   probas = softmax probabilities (B,C,X,Y,Z) by strides; valid voxels = target != ignore_index (and cam_mask);
 *   loss float[1] = mean over the classes present among them of  sort_desc(|[t==c] - p_c|) . lovasz_grad(sorted fg);
 *   inv_present float[1] = 1 / number of present classes; dprob (probas' strides, ZERO-FILLED by the caller, or NULL)
 *   receives d(sum_c loss_c)/d(probas) with lovasz_grad held constant like the reference, i.e. the gradient of the loss
 *   is dprob * inv_present.  workspace: pw_lovasz_workspace_bytes(n_vox = B*X*Y*Z, n_cls, ignore_index) bytes of
 *   device memory (sort keys/values double-buffered + rocPRIM temporary); n_cls <= 32, n_vox < 2^31. */
int pw_focal_loss_stats(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                        const float* class_weights, int B, int n_cls, int X, int Y, int Z, int64_t sb,
                        int64_t sc, int64_t sx, int64_t sy, int64_t sz, int ignore_index, float gamma,
                        float alpha, double* stats, void* stream);
int pw_focal_loss_finish(const double* stats, float loss_weight, float* loss, void* stream);
int pw_focal_loss_grad(const float* logits, const uint8_t* target, const uint8_t* cam_mask,
                       const float* class_weights, int B, int n_cls, int X, int Y, int Z, int64_t sb,
                       int64_t sc, int64_t sx, int64_t sy, int64_t sz, int ignore_index, float gamma,
                       float alpha, const double* stats, float loss_weight, const float* grad_loss,
                       float* grad_logits, void* stream);
/* Distortion loss of the render head (mmdet3d/models/nerf/nerf_head.py:316-327 -> flatten_eff_distloss of torch_efficient_distloss,
 * third-party; mip-NeRF 360 eq. 15 as DVGOv2 evaluates it) on the dense per-sample weights of pw_render_rays:
 *   weights (n_rays, n_samples), 0 where a sample was culled; s (n_samples) the normalised sample positions;
 *   loss[0] = sum_rays [ (1/3) sum_i w_i^2 / n_kept + 2 sum_i w_i (s_i sum_{j<i} w_j - sum_{j<i} w_j s_j) ] / n_rays_used
 *   (n_kept = samples with w > 0 in the batch, at least 1; n_rays_used = 1 + index of the last ray that kept one, at least 1);
 *   scal[0..1] = the two coefficients pw_distortion_loss_backward needs; grad_weights = grad_loss[0] * d loss / d weights.
 * One pass over the weights each way (a wave per ray, wave scans with a carry); block partials folded in double: deterministic. */
size_t pw_distortion_workspace_bytes(int n_rays);
int pw_distortion_loss(const float* weights, const float* s, int n_rays, int n_samples, void* workspace, size_t workspace_bytes,
                       float* loss, float* scal, void* stream);
int pw_distortion_loss_backward(const float* weights, const float* s, int n_rays, int n_samples, const float* scal,
                                const float* grad_loss, float* grad_weights, void* stream);
size_t pw_lovasz_workspace_bytes(int64_t n_vox, int n_cls, int ignore_index);
int pw_lovasz_softmax(const float* probas, const uint8_t* target, const uint8_t* cam_mask, int B, int n_cls,
                      int X, int Y, int Z, int64_t sb, int64_t sc, int64_t sx, int64_t sy, int64_t sz,
                      int ignore_index, void* workspace, size_t workspace_bytes, float* loss,
                      float* inv_present, float* dprob, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training side of the voxel encoder (what torch autograd runs behind mmdet3d/models/backbones/resnet.py:88-184 in
 * forward_train): conv3d weight / data gradients and BatchNorm3d with batch statistics.  fp32, channels-last.
 * ------------------------------------------------------------------------------------- */
/* dW of a Conv3d(kernel ksize in {1,3}, stride in {1,2}, padding ksize/2; or ksize 2, stride 2, no padding): x (B,D,H,W,Cin), dy (B,Do,Ho,Wo,Cout),
 * dw float[Cout][Cin][k][k][k] (torch's layout).  fp32 MFMA with K = voxels, per-chunk partial tiles in `workspace`
 * (pw_conv3d_wgrad_workspace_bytes) summed in a fixed order: deterministic. */
size_t pw_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int ksize, int stride);
int pw_conv3d_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int B, int D, int H,
                    int W, int Cin, int Cout, int ksize, int stride, void* stream);

/* torch Conv3d weight w (Cout, Cin, k, k, k) -> the packed operand layout of pw_conv3d_ndhwc (wino = 0: float[Cin'/32][k^3]
 * [cout_total/32][4][64][4]) or of pw_conv3d_wino (wino = 1, k = 3: float[Cin'/32][64][cout_total/16][64][8], U = G w G^T in
 * float64), in one launch.  flip_t != 0 packs w' = w.flip(2,3,4).transpose(0,1) -- the weight of the stride-1 data gradient --
 * without materialising it (then Cin' = Cout, columns = Cin).  Training re-packs every weight every step. */
int pw_pack_conv_weight(const float* w, int Cout, int Cin, int ksize, int flip_t, int cout_total, float* out, int wino, void* stream);

/* per-voxel dense layer with few channels (OccHead's 1x1x1 convs in training, occupancy_head.py:124-161, and their data
 * gradients): y[n][j] = sum_k x[n][k] w[j][k], x (n, K), w (N, K), y (n, N); built for (K, N) in {16x8, 8x18, 8x1, 8x16, 18x8, 1x8,
 * 32x16, 16x32}, PW_EUNSUP otherwise. */
int pw_linear_rows(const float* x, const float* w, float* y, int64_t n, int K, int N, void* stream);

/* the same gradient for 3x3x3 stride-1 layers with Cin, Cout multiples of 32 on the fp16 matrix cores with split-fp16 operands
 * (22-bit products, fp32 accumulation; operands transposed through LDS, three input rows resident): 3-4x pw_conv3d_wgrad.
 * amax_x / amax_y: 256 partial maxima of |x| / |dy| each -- the two halves of pw_absmax2's output, or what pw_bn_apply / pw_bn_bwd_apply
 * recorded when they wrote the tensor -- for the per-tensor power-of-two pre-scales; both NULL = none.  Deterministic. */
/* largest magnitudes of two fp32 tensors in one launch, as 2 x 256 partial maxima (no atomics): out (device float[512]) [0..255]
 * over x, [256..511] over y; the maximum of a half is max |.|.  Element counts multiples of 4 (0 = skip that tensor: zeros), 16-byte
 * aligned.  Feeds pw_conv3d_wgrad_h2's amax_x / amax_y, which reduces the partials itself. */
int pw_absmax2(const float* x, int64_t nx, const float* y, int64_t ny, float* out, void* stream);
size_t pw_conv3d_wgrad_h2_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int pw_conv3d_wgrad_h2(const float* x, const float* dy, float* dw, const float* amax_x, const float* amax_y, void* workspace,
                       size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* dX of Conv3d(k=3, stride=2, padding=1): dy (B,Do,Ho,Wo,Cout) with Do = (D-1)/2+1 .., wt float[3][3][3][Cout][Cin] (torch's
 * weight.permute(2,3,4,0,1)), dx (B,D,H,W,Cin), Cin % 4 == 0.  (Stride-1 and 1x1x1 data gradients are forward convolutions
 * with flipped / transposed weights: pw_conv3d_ndhwc.) */
int pw_conv3d_dgrad_s2(const float* dy, const float* wt, float* dx, int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* The same data gradient on the fp16 matrix cores with split-fp16 operands (see pw_conv3d_h2; 22-bit products, fp32 accumulation),
 * as the 8 parity classes of the fine grid: dX[2j+p] is a dense 1-, 2-, 4- or 8-tap convolution over dY -- 27 tap products per 8 fine
 * voxels, every dX row written once (no zero fill).  dy (B,Do,Ho,Wo,Cout) in h2 storage under range slot dy_rng (pw_f32_to_h2 with
 * auto_exp), w = torch's (Cout,Cin,3,3,3) weight as it is (transposed, pre-scaled and split on the device into the workspace),
 * dx (B,D,H,W,Cin) fp32; Cin % 32 == 0, Cout % 32 == 0.  Kernel k_conv3d_dgrad_s2_h2<NT>.  Deterministic. */
size_t pw_conv3d_dgrad_s2_h2_workspace_bytes(int Cin, int Cout);
int pw_conv3d_dgrad_s2_h2(const float* dy, const int32_t* dy_rng, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                          int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* dX of the unpadded Conv3d(k=2, stride=2) of the trajectory branch (heads/occupancy_head.py:180-200): dy (B,D/2,H/2,W/2,Cout),
 * wt float[2][2][2][Cout][Cin], dx (B,D,H,W,Cin).  pw_conv3d_wgrad takes ksize 2 / stride 2 for the matching dW. */
int pw_conv3d_dgrad_k2s2(const float* dy, const float* wt, float* dx, int B, int D, int H, int W, int Cin, int Cout, void* stream);
/* BatchNorm3d, training mode, on channels-last rows x (N, C), 256 % C == 0:
 *   pw_bn_stats       mean[c], var[c] (biased), rstd[c] = 1/sqrt(var + eps)              (double accumulation, deterministic)
 *   pw_bn_apply       y = (x - mean) rstd gamma + beta (+ residual) (ReLU if relu)
 *   pw_bn_bwd_reduce  dz = dy (masked by y > 0 if relu);  sum_dz[c] = sum dz,  sum_dz_xhat[c] = sum dz x_hat
 *   pw_bn_bwd_apply   dx = gamma rstd (dz - sum_dz / N - x_hat sum_dz_xhat / N);  dres = dz (or NULL)
 * d gamma = sum_dz_xhat, d beta = sum_dz.  workspace: pw_bn_workspace_bytes(C).
 * Recorded maxima (all four pointers may be NULL): y_amax / dx_amax = float[256] partial maxima of |y| / |dx| in pw_absmax2's format, what
 * pw_conv3d_wgrad_h2 takes as amax_x / amax_y -- the layer that consumes y (or dx) then needs no absmax pass over it.  The buffer is
 * cleared by the reduction that precedes the apply on the same stream: hand the SAME pointer to pw_bn_stats (amax_clear) and
 * pw_bn_apply (y_amax), or to pw_bn_bwd_reduce and pw_bn_bwd_apply.
 * pw_bn_stats with running_mean / running_var != NULL also does pw_bn_update_running's bookkeeping (n = N) in the same launch. */
size_t pw_bn_workspace_bytes(int C);
int pw_bn_stats(const float* x, int64_t N, int C, float eps, void* workspace, size_t workspace_bytes, float* mean, float* var,
                float* rstd, float* amax_clear, float* running_mean, float* running_var, float momentum,
                int64_t* num_batches_tracked, void* stream);
int pw_bn_apply(const float* x, int64_t N, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                const float* residual, int relu, float* y, float* y_amax, void* stream);
int pw_bn_bwd_reduce(const float* x, const float* dy, const float* y, int64_t N, int C, const float* mean, const float* rstd,
                     int relu, void* workspace, size_t workspace_bytes, float* sum_dz, float* sum_dz_xhat, float* amax_clear,
                     void* stream);
int pw_bn_bwd_apply(const float* x, const float* dy, const float* y, int64_t N, int C, const float* mean, const float* rstd,
                    const float* gamma, const float* sum_dz, const float* sum_dz_xhat, int relu, float* dx, float* dres,
                    float* dx_amax, void* stream);
/* nn.BatchNorm3d's running-statistics update after a training-mode forward, one launch: running_mean = (1 - momentum) running_mean
 * + momentum mean; running_var likewise with the UNBIASED batch variance var * n / max(n - 1, 1); num_batches_tracked (int64, may be
 * NULL) += 1.  n = n_rows_dev[0] (device float: the SyncBN row count over all ranks) when n_rows_dev != NULL, else n_rows. */
int pw_bn_update_running(const float* mean, const float* var, int C, double n_rows, const float* n_rows_dev, float momentum,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream);

/* y = act(x + bias[c]) on channels-last rows x (N, C), C <= 256, and its backward in one pass each: dx = dy act'(x + bias),
 * dbias[c] = sum over rows of dx (double accumulation, deterministic; NULL = not wanted).  act: 0 none, 1 ReLU, 2 Softplus (beta 1,
 * threshold 20: torch.nn.Softplus()).  bias may be NULL (zeros).  Training side of the attribute MLPs' Linear -> Softplus
 * (detectors/preworld.py:101-110), final_conv's bias + ReLU (:72-79) and the trajectory branch's biased convs. */
size_t pw_bias_act_workspace_bytes(int C);
int pw_bias_act(const float* x, const float* bias, int64_t N, int C, int act, float* y, void* stream);
int pw_bias_act_backward(const float* x, const float* bias, const float* dy, int64_t N, int C, int act, float* dx, float* dbias,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Trilinear up-sampling with align_corners=True (torch's upsample_trilinear3d index / weight rule) of a channels-last map
 * lo (B,Dl,Hl,Wl,C) to hi (B,Dh,Hh,Wh,C), C % 4 == 0: hi = up(lo) or hi += up(lo) (accumulate != 0); and its adjoint
 * dlo = up^T(dhi), one axis at a time (W, H, D: the operator is a tensor product) through two intermediates in the workspace, each pass a
 * gather (deterministic).  Training side of LSSFPN3D (necks/lss_fpn.py:132-148). */
int pw_upsample_trilinear_add(const float* lo, float* hi, int B, int Dl, int Hl, int Wl, int Dh, int Hh, int Wh, int C,
                              int accumulate, void* stream);
size_t pw_upsample_trilinear_adjoint_workspace_bytes(int B, int Dl, int Hl, int Wl, int Dh, int Hh, int Wh, int C);
int pw_upsample_trilinear_adjoint(const float* dhi, float* dlo, void* workspace, size_t workspace_bytes, int B, int Dl, int Hl, int Wl,
                                  int Dh, int Hh, int Wh, int C, void* stream);

/* n device-to-device copies (src[i] -> dst[i], bytes[i] bytes; host arrays of device pointers) in one launch per 32 segments: a
 * sample's lifted inputs going into the static buffers of a captured step (preworld_amd.pipeline.CapturedSample.run). */
int pw_copy_many(const void* const* src, void* const* dst, const size_t* bytes, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PREWORLD_HIP_H_ */
