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


def img_inputs(seed=0, variant='small'):
    """the 7-tuple `img_inputs` the dataset pipeline delivers (datasets/pipelines/loading.py:1091-1123): imgs (B, N*T, 3, H, W)
    camera-major / frame-minor, sensor2egos / ego2globals (B, T*N, 4, 4) frame-major, intrins, post_rots, post_trans, bda.
    T = 3 frames (key, adjacent, extra stereo reference) with a moving ego; per-camera image augmentation; a BEV augmentation."""
    rs = np.random.RandomState(1000 + seed)
    T = 3
    rig = S.synthetic_rig(6, dtype=np.float64)
    cams = VARIANTS[variant]['cams']
    N_CAMS = len(cams)
    s2e = np.stack([rig['sensor2ego'][0, cams]] * T, 0)                      # (T, N, 4, 4): same rig every frame
    e2g = np.stack([np.stack([_pose(0.10 - 0.03 * t, [10.0 - 2.4 * t, 5.0 - 0.3 * t, 0.2])] * N_CAMS, 0) for t in range(T)], 0)
    K = np.stack([rig['intrin'][0, cams]] * T, 0)
    v = VARIANTS[variant]
    ks = v.get('k_scale', 0.25)                                                                 # 0.25: a 128 x 352 image
    K[..., 0, 0] *= ks; K[..., 1, 1] *= ks; K[..., 0, 2] *= ks; K[..., 1, 2] *= ks
    r0, dr = v.get('post_rot', (0.9, 0.05))
    tx, ty, dty = v.get('post_tran', (3.0, -20.0, 2.0))
    pr = np.stack([np.stack([np.diag([r0 + dr * n, r0 + dr * n, 1.0]) for n in range(N_CAMS)], 0)] * T, 0)
    pt = np.stack([np.stack([np.array([tx * n, ty + dty * n, 0.0]) for n in range(N_CAMS)], 0)] * T, 0)
    a = 0.05
    bda = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * np.array([1.02, 1.02, 1.0])[None, :]
    f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))       # noqa: E731
    if v.get('input_size') is None:
        imgs = torch.from_numpy(rs.standard_normal((1, N_CAMS * T, 3) + INPUT_SIZE).astype(np.float32))
    else:                       # the stand-in image side reads only the shape (155 MB at 512 x 1408): no random fill
        imgs = torch.zeros((1, N_CAMS * T, 3) + tuple(v['input_size']))
    return (imgs, f32(s2e.reshape(1, T * N_CAMS, 4, 4)), f32(e2g.reshape(1, T * N_CAMS, 4, 4)),
            f32(K.reshape(1, T * N_CAMS, 3, 3)), f32(pr.reshape(1, T * N_CAMS, 3, 3)), f32(pt.reshape(1, T * N_CAMS, 3)),
            f32(bda[None]))


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


def train_kwargs(seed, detector, device='cpu', variant='small'):
    """the keyword arguments forward_train reads (preworld.py:256-263, preworld_temporal_traj.py:412-524): voxel_semantics (B,X,Y,Z),
    mask_camera, and for the temporal detector temporal_semantics[k]['voxel_semantics'], temporal_ego_states, temporal_trajs"""
    rs = np.random.RandomState(3000 + seed)
    X, Y, Z = grid_shape(variant)
    sem = lambda: torch.from_numpy(rs.randint(0, 18, (1, X, Y, Z))).to(device)       # noqa: E731
    kw = dict(voxel_semantics=sem(), mask_camera=None)
    if detector == 'PreWorld4DTraj':
        kw['temporal_semantics'] = [dict(voxel_semantics=sem()) for _ in range(7)]
        kw['temporal_ego_states'] = [torch.from_numpy(S.ego_state(40 + seed)).to(device)]
        kw['temporal_trajs'] = torch.from_numpy(rs.standard_normal((1, 6, 2)).astype(np.float32)).to(device)
    return kw


def grad_probes(model, detector):
    """(name, parameter) pairs whose gradients the training fixture stores"""
    pr = [('final_conv', model.final_conv.conv.weight), ('occ_conv', model.occupancy_head.occ_convs[0][0].weight),
          ('encoder_l0_conv1', model.img_bev_encoder_backbone.layers[0][0].conv1.conv.weight),
          ('pre_process_conv2', model.pre_process_net.layers[0][0].conv2.conv.weight)]
    if detector == 'PreWorld4DTraj':
        pr += [('fusion_head0', model.fusion_head[0].weight), ('plan_head0', model.plan_head[0].weight),
               ('traj_head2', model.traj_head[2].weight), ('downscale1', model.downscale.downscale1.weight)]
    return pr
