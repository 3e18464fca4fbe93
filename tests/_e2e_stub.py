"""Shared by tools/gen_golden.py (which runs the REFERENCE's detector classes in the development container) and
tests/test_gpu_e2e_reference.py (which runs the drop-in detectors on the GPU): seeded `img_inputs`, and stand-ins for the image
side -- the part north_star leaves on PyTorch and SURVEY 8a excludes from the hot path.  Both sides install the same stand-ins,
so everything downstream of the DepthNet output (the frame loop, pose algebra, lift, pre_process, concat order, encoder, neck,
final_conv, permutes, forecast recursion, heads, argmax, result keys) is the code under test.  No reference source here."""
import numpy as np
import torch
import torch.nn as nn

from preworld_amd import synth as S

GRID = {'x': [-40, 40, 2.0], 'y': [-40, 40, 2.0], 'z': [-1, 5.4, 0.8], 'depth': [1.0, 45.0, 0.5]}
INPUT_SIZE = (128, 352)
N_CAMS = 2
IN_CH = 16           # channels of the (stand-in) image-view feature map
D, C = 88, 32
TEST_THRESHOLD = 0.7  # density threshold of the attribute-MLP decode (8.5 in the configs; random weights never reach that)
VARIANTS = {
    'small': dict(grid=GRID, cams=[1, 4]),
    # BASELINE.json configs[0]'s voxel grid with the full camera rig: 6 cameras, 100 x 100 x 8
    'c6': dict(grid={'x': [-40, 40, 0.8], 'y': [-40, 40, 0.8], 'z': [-1, 5.4, 0.8], 'depth': [1.0, 45.0, 0.5]}, cams=[0, 1, 2, 3, 4, 5]),
    # BASELINE.json's headline config as it is (C3, round 5 / VERDICT r04 item 2): 200 x 200 x 16 grid, the full camera rig AND the
    # 512 x 1408 image -- a 32 x 88 feature map, 88 x 32 x 88 x 6 = 1 486 848 frustum points per frame
    # (necks/view_transformer.py:84-112), intrinsics unscaled, the test-time resize 0.88 / crop -280 of loading.py:988-1000
    'full': dict(grid={'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 0.4], 'depth': [1.0, 45.0, 0.5]}, cams=[0, 1, 2, 3, 4, 5],
                 input_size=(512, 1408), k_scale=1.0, post_rot=(0.88, 0.004), post_tran=(3.0, -280.0, 2.0)),
}


def input_size(variant='small'):
    return VARIANTS[variant].get('input_size', INPUT_SIZE)
RUNS = [('p4d_ft', 'PreWorld4DTraj', True, True), ('p4d_ft_noprev', 'PreWorld4DTraj', True, False),
        ('p4d_attr', 'PreWorld4DTraj', False, True), ('pw_ft', 'PreWorld', True, True), ('pw_attr', 'PreWorld', False, True)]


def model_cfg(detector, if_post_finetune, with_prev=True, variant='small'):
    """the `model = dict(...)` of configs/preworld/**.py at the reduced grid above (cf. harness.model_cfg)"""
    from preworld_amd import harness
    cfg = harness.model_cfg(VARIANTS[variant]['grid'], with_prev=with_prev, if_post_finetune=if_post_finetune, detector=detector)
    cfg['img_view_transformer'].update(input_size=input_size(variant), in_channels=IN_CH)
    cfg['use_focal_loss'] = False        # loss objects are not part of the inference path (mmdet registry, not in the tree)
    cfg['test_threshold'] = TEST_THRESHOLD
    return cfg


def _pose(yaw, t):
    m = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    m[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    m[:3, 3] = t
    return m


def _img_inputs_one(seed=0, variant='small', b=0):
    """the 7-tuple `img_inputs` the dataset pipeline delivers (datasets/pipelines/loading.py:1091-1123): imgs (B, N*T, 3, H, W)
    camera-major / frame-minor, sensor2egos / ego2globals (B, T*N, 4, 4) frame-major, intrins, post_rots, post_trans, bda.
    T = 3 frames (key, adjacent, extra stereo reference) with a moving ego; per-camera image augmentation; a BEV augmentation.
    b > 0 (further batch elements, round 6): another ego track, other image / BEV augmentations, its own random image."""
    rs = np.random.RandomState(1000 + seed + 100 * b)
    T = 3
    rig = S.synthetic_rig(6, dtype=np.float64)
    cams = VARIANTS[variant]['cams']
    N_CAMS = len(cams)
    s2e = np.stack([rig['sensor2ego'][0, cams]] * T, 0)                      # (T, N, 4, 4): same rig every frame
    e2g = np.stack([np.stack([_pose(0.10 - 0.03 * t - 0.4 * b, [10.0 - 2.4 * t + 7.0 * b, 5.0 - 0.3 * t + 1.1 * b * t, 0.2])] * N_CAMS, 0)
                    for t in range(T)], 0)
    K = np.stack([rig['intrin'][0, cams]] * T, 0)
    v = VARIANTS[variant]
    ks = v.get('k_scale', 0.25)                                                                 # 0.25: a 128 x 352 image
    K[..., 0, 0] *= ks; K[..., 1, 1] *= ks; K[..., 0, 2] *= ks; K[..., 1, 2] *= ks
    r0, dr = v.get('post_rot', (0.9, 0.05))
    tx, ty, dty = v.get('post_tran', (3.0, -20.0, 2.0))
    r0, ty = r0 * (1.0 - 0.06 * b), ty * (1.0 + 0.25 * b)
    pr = np.stack([np.stack([np.diag([r0 + dr * n, r0 + dr * n, 1.0]) for n in range(N_CAMS)], 0)] * T, 0)
    pt = np.stack([np.stack([np.array([tx * n, ty + dty * n, 0.0]) for n in range(N_CAMS)], 0)] * T, 0)
    a = 0.05 - 0.12 * b
    bda = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * np.array([1.02 - 0.05 * b, 1.02 - 0.05 * b, 1.0])[None, :]
    if b % 2:                                                                                   # flip_dy of the BEV augmentation (loading.py:1048-1062)
        bda = np.diag([1.0, -1.0, 1.0]) @ bda
    f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))       # noqa: E731
    if v.get('input_size') is None:
        imgs = torch.from_numpy(rs.standard_normal((1, N_CAMS * T, 3) + INPUT_SIZE).astype(np.float32))
    else:                       # the stand-in image side reads only the shape (155 MB at 512 x 1408): no random fill
        imgs = torch.zeros((1, N_CAMS * T, 3) + tuple(v['input_size']))
    return (imgs, f32(s2e.reshape(1, T * N_CAMS, 4, 4)), f32(e2g.reshape(1, T * N_CAMS, 4, 4)),
            f32(K.reshape(1, T * N_CAMS, 3, 3)), f32(pr.reshape(1, T * N_CAMS, 3, 3)), f32(pt.reshape(1, T * N_CAMS, 3)),
            f32(bda[None]))


def img_inputs(seed=0, variant='small', batch=1):
    """`batch` samples collated along dim 0 (samples_per_gpu = 2 in configs/preworld/nuscenes/preworld-7frame-finetune.py:58)"""
    ones = [_img_inputs_one(seed, variant, b) for b in range(batch)]
    return ones[0] if batch == 1 else tuple(torch.cat([o[i] for o in ones], 0) for i in range(7))


def ego_states(seed=0):
    """kwargs['temporal_ego_states'] as the detectors index it: [0][0] is the (B, 1, 21) tensor (preworld_temporal_traj.py:228,259)"""
    return [[torch.from_numpy(S.ego_state(40 + seed))]]


class SeededDepthNet(nn.Module):
    """stands in for view_transformer.DepthNet: call k returns a seeded (B*N, D + C, H, W) tensor (depth logits + context) and
    records the `mlp_input` it was handed"""

    def __init__(self, seed=0, variant='small'):
        super().__init__()
        self.seed, self.k, self.mlp_inputs, self.size = seed, 0, [], input_size(variant)

    def reset(self):
        self.k, self.mlp_inputs = 0, []

    def forward(self, x, mlp_input, stereo_metas=None):
        rs = np.random.RandomState(2000 + 10 * self.seed + self.k)
        self.k += 1
        self.mlp_inputs.append(mlp_input.detach().cpu().clone())
        H, W = self.size[0] // 16, self.size[1] // 16
        out = rs.standard_normal((x.shape[0], D + C, H, W)).astype(np.float32)
        out[:, :D] *= 2.0
        return torch.from_numpy(out).to(x.device)


def install_image_side(model, seed=0, variant='small'):
    """replace the image side of a detector (reference class or drop-in) by the stand-ins; returns the SeededDepthNet"""
    H, W = input_size(variant)[0] // 16, input_size(variant)[1] // 16

    def image_encoder(img, stereo=False):
        B, N = img.shape[:2]
        return img.new_zeros(B, N, IN_CH, H, W), img.new_zeros(B * N, 8, 4 * H, 4 * W)

    def extract_stereo_ref_feat(x):
        B, N = x.shape[:2]
        return x.new_zeros(B * N, 8, 4 * H, 4 * W)
    model.image_encoder = image_encoder
    model.extract_stereo_ref_feat = extract_stereo_ref_feat
    dn = SeededDepthNet(seed, variant)
    model.img_view_transformer.depth_net = dn
    return dn


# ---- training (tools/gen_golden.py gen_e2e_train / tests/test_gpu_e2e_reference.py): the fine-tune flags of
# configs/preworld/preworld-7frame-finetune*.py with every voxel loss switched on
TRAIN_CFG = dict(if_render=False, if_post_finetune=True, use_lss_depth_loss=False, use_focal_loss=False, weight_voxel_ce=1.0,
                 weight_voxel_sem_scal=1.0, weight_voxel_geo_scal=1.0, weight_voxel_lovasz=1.0)


TRAIN_EPOCH = 7      # PreWorld4DTraj.set_epoch: without rendering, epoch 7 supervises the current state and three forecast intervals


def grid_shape(variant='small'):
    g = VARIANTS[variant]['grid']
    return tuple(int(round((g[a][1] - g[a][0]) / g[a][2])) for a in 'xyz')


def train_kwargs(seed, detector, device='cpu', variant='small', batch=1):
    """the keyword arguments forward_train reads (preworld.py:256-263, preworld_temporal_traj.py:412-524): voxel_semantics (B,X,Y,Z),
    mask_camera, and for the temporal detector temporal_semantics[k]['voxel_semantics'], temporal_ego_states, temporal_trajs"""
    rs = np.random.RandomState(3000 + seed)
    X, Y, Z = grid_shape(variant)
    sem = lambda: torch.from_numpy(rs.randint(0, 18, (batch, X, Y, Z))).to(device)       # noqa: E731
    kw = dict(voxel_semantics=sem(), mask_camera=None)
    if detector == 'PreWorld4DTraj':
        kw['temporal_semantics'] = [dict(voxel_semantics=sem()) for _ in range(7)]
        kw['temporal_ego_states'] = [torch.from_numpy(np.concatenate([S.ego_state(40 + seed + 7 * b) for b in range(batch)], 0)).to(device)]
        kw['temporal_trajs'] = torch.from_numpy(rs.standard_normal((batch, 6, 2)).astype(np.float32)).to(device)
    return kw


# ---- pre-training (round 6): the flags of configs/preworld/nuscenes/preworld-7frame-pretrain.py:10-33 and
# nuscenes-temporal/preworld-7frame-pretrain-traj.py (the same with use_lss_depth_loss=False): rendering losses with the released
# NerfHead weights, the voxel losses replaced by the zero-weight loss_sup_voxel, the LSS depth loss
NERF_HEAD_CFG = dict(type='NerfHead', point_cloud_range=[-40., -40., -1., 40., 40., 5.4], voxel_size=0.4, scene_center=[0, 0, 2.2],
                     radius=39, use_depth_sup=True, weight_depth=1.0, weight_semantic=1.0, weight_color=1.0,
                     weight_entropy_last=0.01, weight_distortion=0.01)


def pretrain_cfg(detector):
    return dict(final_softplus=True, use_lss_depth_loss=detector == 'PreWorld', use_3d_loss=False, if_render=True, if_post_finetune=False,
                weight_voxel_ce=0.0, weight_voxel_sem_scal=0.0, weight_voxel_geo_scal=0.0, weight_voxel_lovasz=0.0, empty_idx=17,
                use_focal_loss=False, nerf_head=dict(NERF_HEAD_CFG))


PRETRAIN_EPOCH = 4   # PreWorld4DTraj with rendering: epoch 4 -> future intervals 0, 1, 2, i.e. temporal_rays[1..3] (:436-440)
PRETRAIN_RAYS = 160  # rays per sample (38 400 / 19 200 in the configs)


def ray_batch(seed, batch, R=PRETRAIN_RAYS):
    """`rays` (B, R, 16) in the layout NerfHead.forward slices (nerf_head.py:362-367; datasets/ray.py builds it): [2] lidar depth --
    a tenth of the rays 0 (no label), a tenth beyond the 52 m cut of :379 -- [3] semantic label 0..16, [4:7] origin, [7:10] direction,
    [13:16] colour; the other columns (pixel coordinates, camera id) are not read"""
    rs = np.random.RandomState(4000 + seed)
    rays = np.zeros((batch, R, 16), np.float32)
    for b in range(batch):
        o, d = S.rays_mixed(4100 + 10 * seed + b, R, n_special=0)
        depth = rs.uniform(1.0, 50.0, R)
        depth[rs.rand(R) < 0.1] = 0.0
        far = rs.rand(R) < 0.1
        depth[far] = rs.uniform(52.5, 70.0, int(far.sum()))
        rays[b, :, 0:2] = rs.uniform(0, 100, (R, 2))
        rays[b, :, 2], rays[b, :, 3], rays[b, :, 4:7], rays[b, :, 7:10] = depth, rs.randint(0, 17, R), o, d
        rays[b, :, 13:16] = rs.uniform(0, 1, (R, 3))
    return torch.from_numpy(rays)


def pretrain_kwargs(seed, detector, device='cpu', variant='small', batch=2):
    """what forward_train reads under the pre-train flags: `rays` (preworld.py:289), `gt_depth` (B,N,H,W) for the LSS depth loss
    (:303-304, view_transformer.py:736-789), `voxel_semantics` (asserted in range, weight 0), and for the temporal detector
    `temporal_rays[k]` for the forecast states (preworld_temporal_traj.py:510) with ego states and trajectories"""
    kw = train_kwargs(seed, detector, device, variant, batch)
    kw['rays'] = ray_batch(seed, batch).to(device)
    if detector == 'PreWorld4DTraj':
        kw['temporal_rays'] = [ray_batch(seed + 10 * (k + 1), batch).to(device) for k in range(7)]
    rs = np.random.RandomState(5000 + seed)
    H, W = input_size(variant)
    n = len(VARIANTS[variant]['cams'])
    gd = rs.uniform(0.5, 60.0, (batch, n, H, W)).astype(np.float32)
    gd[rs.rand(batch, n, H, W) < 0.97] = 0.0                               # projected lidar: sparse
    kw['gt_depth'] = torch.from_numpy(gd).to(device)
    return kw


def opaque_density_state(sd, gain=16.0, shift=-8.0):
    """the synthetic density_mlp never leaves the transparent regime (softplus of O(1) numbers against act_shift = -13.8): scale and shift
    its output row 0 so that part of the volume is opaque and rays TERMINATE (T < 1e-3, render_utils_kernel.cu:591-603) under the
    pre-train fixtures.  Returns a copy of the state dict."""
    sd = dict(sd)
    w, b = np.array(sd['density_mlp.2.weight'], copy=True), np.array(sd['density_mlp.2.bias'], copy=True)
    w[0] *= gain
    b[0] = b[0] * gain + shift
    sd['density_mlp.2.weight'], sd['density_mlp.2.bias'] = w, b
    return sd


def grad_probes(model, detector):
    """(name, parameter) pairs whose gradients the training fixture stores"""
    pr = [('final_conv', model.final_conv.conv.weight), ('occ_conv', model.occupancy_head.occ_convs[0][0].weight),
          ('encoder_l0_conv1', model.img_bev_encoder_backbone.layers[0][0].conv1.conv.weight),
          ('pre_process_conv2', model.pre_process_net.layers[0][0].conv2.conv.weight)]
    if detector == 'PreWorld4DTraj':
        pr += [('fusion_head0', model.fusion_head[0].weight), ('plan_head0', model.plan_head[0].weight),
               ('traj_head2', model.traj_head[2].weight), ('downscale1', model.downscale.downscale1.weight)]
    return pr


def pretrain_grad_probes(model, detector):
    """the parameters the rendering losses reach: the three attribute MLPs, final_conv, the encoder, (temporal) the forecast heads;
    the OccHead only through the zero-weight loss_sup_voxel (its gradient is exactly zero)"""
    pr = [('final_conv', model.final_conv.conv.weight), ('density_mlp0', model.density_mlp[0].weight), ('density_mlp2', model.density_mlp[2].weight),
          ('semantic_mlp0', model.semantic_mlp[0].weight), ('semantic_mlp2', model.semantic_mlp[2].weight),
          ('color_mlp0', model.color_mlp[0].weight), ('color_mlp2', model.color_mlp[2].weight),
          ('encoder_l0_conv1', model.img_bev_encoder_backbone.layers[0][0].conv1.conv.weight),
          ('pre_process_conv2', model.pre_process_net.layers[0][0].conv2.conv.weight),
          ('occ_conv', model.occupancy_head.occ_convs[0][0].weight)]
    if detector == 'PreWorld4DTraj':
        pr += [('fusion_head0', model.fusion_head[0].weight), ('plan_head0', model.plan_head[0].weight),
               ('traj_head2', model.traj_head[2].weight)]
    return pr
