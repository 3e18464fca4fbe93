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
const char* pw_last_error(void);
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
 * workspace: pw_segment_sort_workspace_bytes(n, n_keys) bytes, 256-byte aligned. */
size_t pw_segment_sort_workspace_bytes(int64_t n, int64_t n_keys);
int pw_segment_sort(int64_t n, int64_t n_keys, const int32_t* keys, void* workspace,
                    size_t workspace_bytes, int32_t* seg_start, int32_t* order, void* stream);

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
 * seg_start/order come from pw_segment_sort over the voxel ids; D/HW as above. */
int pw_bev_pool_dense(const float* depth, const float* feat, const int32_t* seg_start,
                      const int32_t* order, int64_t n_voxels, int c, int D, int HW, float* out,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PREWORLD_HIP_H_ */
