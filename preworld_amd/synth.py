"""Seeded synthetic inputs and weights (numpy only; no torch, no oracle).

Used by bench.py, __graft_entry__.smoke(), tools/gen_golden.py and the tests so that
every side regenerates identical tensors from a seed instead of shipping them as
fixtures.  Uses numpy's legacy RandomState, whose streams are stable across numpy
versions.  Shapes/kwargs follow SURVEY.md 8b/8d (reference configs
configs/preworld/nuscenes/bevstereo-occ.py:90-108, preworld-7frame-finetune.py).
"""
import math

import numpy as np

GRID_CONFIG_FULL = {'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 0.4],
                    'depth': [1.0, 45.0, 0.5]}
# C1 (BASELINE.json configs[0]): 1 camera, 100x100x8 grid
GRID_CONFIG_C1 = {'x': [-40, 40, 0.8], 'y': [-40, 40, 0.8], 'z': [-1, 5.4, 0.8],
                  'depth': [1.0, 45.0, 0.5]}
INPUT_SIZE = (512, 1408)
DOWNSAMPLE = 16


def synthetic_rig(n_cams=6, dx=0.0, dtype=np.float32):
    """Analytic nuScenes-like 6-camera rig. dx = ego translation of the
    adjacent frame along x.  Returns (1,N,...) arrays keyed like the reference's inputs
    (sensor2ego, intrin, post_rot, post_tran, bda) -- view_transformer.py:791-796."""
    yaws = [55, 0, -55, -110, 180, 110][:n_cams] if n_cams > 1 else [0]
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    n = len(yaws)
    s2e = np.zeros((1, n, 4, 4), np.float64)
    for i, y in enumerate(yaws):
        a = math.radians(y)
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        s2e[0, i, :3, :3] = Rz @ base
        s2e[0, i, :3, 3] = [1.5 * math.cos(a) + dx, 1.5 * math.sin(a), 1.5]
        s2e[0, i, 3, 3] = 1
    K = np.array([[1266.4, 0, 816.27], [0, 1266.4, 491.5], [0, 0, 1]], np.float64)
    return dict(
        sensor2ego=s2e.astype(dtype),
        intrin=np.broadcast_to(K, (1, n, 3, 3)).astype(dtype).copy(),
        post_rot=np.broadcast_to(np.diag([0.88, 0.88, 1.0]), (1, n, 3, 3)).astype(dtype).copy(),
        post_tran=np.broadcast_to(np.array([0.0, -280.0, 0.0]), (1, n, 3)).astype(dtype).copy(),
        bda=np.eye(3, dtype=dtype)[None].copy(),
    )


def lift_inputs(seed, B=1, N=6, D=88, H=32, W=88, C=32):
    """depth = softmax(N(0,1)) over D bins, feat ~ N(0,1): (B,N,D,H,W), (B,N,C,H,W)."""
    rs = np.random.RandomState(seed)
    logits = rs.standard_normal((B, N, D, H, W)).astype(np.float32)
    logits -= logits.max(2, keepdims=True)
    e = np.exp(logits)
    depth = (e / e.sum(2, keepdims=True)).astype(np.float32)
    feat = rs.standard_normal((B, N, C, H, W)).astype(np.float32)
    return depth, feat


def _conv_w(rs, cout, cin, k):
    fan_in = cin * k ** 3
    return (rs.standard_normal((cout, cin, k, k, k)) * math.sqrt(2.0 / fan_in)).astype(np.float32)


def _bn(rs, sd, prefix, c):
    sd[prefix + '.weight'] = (rs.rand(c) + 0.5).astype(np.float32)
    sd[prefix + '.bias'] = (rs.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + '.running_mean'] = (rs.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + '.running_var'] = (rs.rand(c) + 0.5).astype(np.float32)


def _conv_module(rs, sd, prefix, cin, cout, k=3):
    sd[prefix + '.conv.weight'] = _conv_w(rs, cout, cin, k)
    _bn(rs, sd, prefix + '.bn', cout)


def _resnet3d(rs, sd, prefix, numC_input, num_layer, num_channels):
    cur = numC_input
    for lid, nl in enumerate(num_layer):
        for b in range(nl):
            p = '%s.layers.%d.%d' % (prefix, lid, b)
            cin = cur if b == 0 else num_channels[lid]
            _conv_module(rs, sd, p + '.conv1', cin, num_channels[lid])
            _conv_module(rs, sd, p + '.conv2', num_channels[lid], num_channels[lid])
            if b == 0:
                _conv_module(rs, sd, p + '.downsample', cin, num_channels[lid])
        cur = num_channels[lid]


def _linear(rs, sd, prefix, fin, fout):
    sd[prefix + '.weight'] = (rs.standard_normal((fout, fin)) / math.sqrt(fin)).astype(np.float32)
    sd[prefix + '.bias'] = (rs.standard_normal(fout) * 0.1).astype(np.float32)


def synth_state_dict(seed=0, out_dim=32, num_classes=18):
    """Random weights for every hot-path module with the reference's state-dict keys
    (SURVEY.md 8b): pre_process_net, img_bev_encoder_backbone, img_bev_encoder_neck,
    final_conv, occupancy_head, density/semantic/color_mlp, plan_head, fusion_head."""
    rs = np.random.RandomState(seed)
    sd = {}
    _resnet3d(rs, sd, 'pre_process_net', 32, [1], [32])
    _resnet3d(rs, sd, 'img_bev_encoder_backbone', 64, [1, 2, 4], [32, 64, 128])
    _conv_module(rs, sd, 'img_bev_encoder_neck.conv', 224, 32, k=1)
    sd['final_conv.conv.weight'] = _conv_w(rs, out_dim, 32, 3)
    sd['final_conv.conv.bias'] = (rs.standard_normal(out_dim) * 0.1).astype(np.float32)
    # OccHead (occupancy_head.py:80-105): occ_convs.0 = [conv3 32->16, BN, ReLU];
    # occ_pred_conv = [1x1 16->8, BN, ReLU, 1x1 8->18]; voxel_soft_weights likewise
    sd['occupancy_head.occ_convs.0.0.weight'] = _conv_w(rs, 16, 32, 3)
    _bn(rs, sd, 'occupancy_head.occ_convs.0.1', 16)
    sd['occupancy_head.occ_pred_conv.0.weight'] = _conv_w(rs, 8, 16, 1)
    _bn(rs, sd, 'occupancy_head.occ_pred_conv.1', 8)
    sd['occupancy_head.occ_pred_conv.3.weight'] = _conv_w(rs, num_classes, 8, 1)
    sd['occupancy_head.voxel_soft_weights.0.weight'] = _conv_w(rs, 8, 16, 1)
    _bn(rs, sd, 'occupancy_head.voxel_soft_weights.1', 8)
    sd['occupancy_head.voxel_soft_weights.3.weight'] = _conv_w(rs, 1, 8, 1)
    for name, outs in (('density_mlp', 2), ('semantic_mlp', num_classes - 1), ('color_mlp', 3)):
        _linear(rs, sd, name + '.0', out_dim, out_dim * 2)
        _linear(rs, sd, name + '.2', out_dim * 2, outs)
    _linear(rs, sd, 'plan_head.0', 21, 256)
    _linear(rs, sd, 'plan_head.2', 256, 256)
    _linear(rs, sd, 'plan_head.4', 256, out_dim)
    _linear(rs, sd, 'fusion_head.0', out_dim * 2, out_dim * 4)
    _linear(rs, sd, 'fusion_head.2', out_dim * 4, out_dim)
    # A20 trajectory branch -- appended LAST so the random stream of every earlier key (and the
    # fixtures generated from it) is unchanged
    for name, ci, co in (('downscale.downscale1', out_dim, out_dim * 2),
                         ('downscale.downscale2', out_dim * 2, out_dim * 4),
                         ('downscale.downscale3', out_dim * 4, out_dim * 4)):
        bound = 1.0 / np.sqrt(ci * 8)
        sd[name + '.weight'] = rs.uniform(-bound, bound, (co, ci, 2, 2, 2)).astype(np.float32)
        sd[name + '.bias'] = rs.uniform(-bound, bound, (co,)).astype(np.float32)
    _linear(rs, sd, 'ego_fusion_head.0', out_dim * 5, out_dim * 8)
    _linear(rs, sd, 'ego_fusion_head.2', out_dim * 8, out_dim * 4)
    _linear(rs, sd, 'ego_fusion_head.4', out_dim * 4, out_dim * 2)
    _linear(rs, sd, 'ego_fusion_head.6', out_dim * 2, out_dim)
    _linear(rs, sd, 'traj_head.0', out_dim, out_dim * 2)
    _linear(rs, sd, 'traj_head.2', out_dim * 2, 2)
    return sd


def ego_state(seed):
    return np.random.RandomState(seed).standard_normal((1, 1, 21)).astype(np.float32)


def render_grids(seed, X=200, Y=200, Z=16, n_sem=17):
    """density (X,Y,Z) sparse positive, semantic (X,Y,Z,17), color (X,Y,Z,3)."""
    rs = np.random.RandomState(seed)
    raw = rs.standard_normal((X, Y, Z)).astype(np.float32) * 4 - 6
    density = np.where(raw > 20, raw, np.log1p(np.exp(np.minimum(raw, 20)))).astype(np.float32)
    semantic = rs.standard_normal((X, Y, Z, n_sem)).astype(np.float32)
    color = rs.standard_normal((X, Y, Z, 3)).astype(np.float32)
    return density, semantic, color


def rays(seed, R, n_cams=6):
    """(R,3) origins at the rig's camera centres, (R,3) directions mostly horizontal."""
    rs = np.random.RandomState(seed)
    rig = synthetic_rig(n_cams)
    cam = rs.randint(0, n_cams, R)
    o = rig['sensor2ego'][0, :, :3, 3][cam].astype(np.float32)
    d = rs.standard_normal((R, 3)).astype(np.float32)
    d[:, 2] *= 0.15
    return o, d


HARD_BLOCK = (150, 150)        # (x, y) voxel of the density-30 block of render_grids_mixed


def render_grids_mixed(seed, X=200, Y=200, Z=16, n_sem=17, n_blocks=120):
    """A scene in the regime released checkpoints sit in (occupied <=> density > 8.5, detectors/preworld.py:32,180): free space
    = softplus of a negative number (~2e-3), a ground slab (two layers) and n_blocks boxes with density ~ U(10, 22) -- a ray that
    meets them loses its transmittance within a few samples and stops at T < 1e-3 (render_utils_kernel.cu:591-603) -- plus one
    box of density 30 around HARD_BLOCK (alpha > 0.999 per sample: a ray starting inside it keeps exactly one sample).  The
    cameras' neighbourhood stays free.  Returns density (X,Y,Z), semantic (X,Y,Z,17), color (X,Y,Z,3)."""
    rs = np.random.RandomState(seed)
    raw = rs.standard_normal((X, Y, Z)).astype(np.float32) * 2 - 6
    density = np.log1p(np.exp(raw)).astype(np.float32)
    occ = np.zeros((X, Y, Z), bool)
    occ[:, :, :2] = True
    for _ in range(n_blocks):
        x0, y0 = rs.randint(0, X - 12), rs.randint(0, Y - 12)
        sx, sy = rs.randint(2, 12, 2)
        occ[x0:x0 + sx, y0:y0 + sy, 2:rs.randint(3, 14)] = True
    occ[X // 2 - 8:X // 2 + 8, Y // 2 - 8:Y // 2 + 8, 2:] = False
    density = np.where(occ, rs.uniform(10, 22, (X, Y, Z)).astype(np.float32), density)
    hx, hy = HARD_BLOCK
    if hx + 4 <= X and hy + 4 <= Y:
        density[hx - 4:hx + 4, hy - 4:hy + 4, :] = 30.0
    semantic = rs.standard_normal((X, Y, Z, n_sem)).astype(np.float32)
    color = rs.standard_normal((X, Y, Z, 3)).astype(np.float32)
    return density, semantic, color


def rays_mixed(seed, R, n_cams=6, n_special=8):
    """rays for render_grids_mixed: R - n_special from the camera centres with a wide, downward-biased pitch spread (most meet the
    ground slab or a box), then n_special rays starting INSIDE the density-30 block (one kept sample each)."""
    rs = np.random.RandomState(seed)
    rig = synthetic_rig(n_cams)
    cam = rs.randint(0, n_cams, R)
    o = rig['sensor2ego'][0, :, :3, 3][cam].astype(np.float32)
    d = rs.standard_normal((R, 3)).astype(np.float32)
    d[:, 2] = d[:, 2] * 0.5 - 0.2
    n_special = min(n_special, R)
    if n_special:
        hx, hy = HARD_BLOCK
        o[R - n_special:] = np.array([-40 + 0.4 * hx, -40 + 0.4 * hy, 1.0], np.float32) + \
            rs.uniform(-0.5, 0.5, (n_special, 3)).astype(np.float32)
    return o, d


def render_grids_void(seed, X=200, Y=200, Z=16, n_sem=17):
    """density -5 everywhere: alpha = 1 - (1 + e^(-5 - 13.8))^-0.5 ~ 3e-9 is below fast_color_thres = 1e-7, so every sample INSIDE
    the grid goes in the first compaction (nerf_head.py:229-238); out-of-grid samples read 0 and stay.  A horizontal ray
    (rays_void) never leaves the grid and keeps nothing."""
    rs = np.random.RandomState(seed)
    return (np.full((X, Y, Z), -5.0, np.float32), rs.standard_normal((X, Y, Z, n_sem)).astype(np.float32),
            rs.standard_normal((X, Y, Z, 3)).astype(np.float32))


def rays_void(seed, R, n_cams=6):
    """every second ray exactly horizontal (d_z = 0: all of its samples are inside the grid's z range)"""
    o, d = rays(seed, R, n_cams)
    d[::2, 2] = 0.0
    return o, d


def ray_label_inputs(seed, n_cams=4, n_pts=(300, 257, 64, 1)):
    """Seeded labelled pixels per camera for the ray-table rows (mmdet3d/datasets/ray.py):
    lists of coor (n,2) pixel xy, depth (n), seg (n) class ids as float, rgb (n,3), c2w (4,4), K (3,3)."""
    rs = np.random.RandomState(seed)
    rig = synthetic_rig(6)
    coors, depths, segs, imgs, c2ws, Ks = [], [], [], [], [], []
    for c in range(n_cams):
        n = n_pts[c]
        coors.append(np.stack([rs.randint(0, 1600, n), rs.randint(0, 900, n)], 1).astype(np.float32))
        depths.append(rs.uniform(1, 52, n).astype(np.float32))
        segs.append(rs.randint(0, 17, n).astype(np.float32))
        imgs.append(rs.standard_normal((n, 3)).astype(np.float32))
        c2ws.append(rig['sensor2ego'][0, c].astype(np.float32))
        Ks.append(rig['intrin'][0, c].astype(np.float32))
    return coors, depths, segs, imgs, c2ws, Ks


def voxel_loss_inputs(seed, shape=(2, 18, 10, 12, 6)):
    rs = np.random.RandomState(seed)
    B, C, X, Y, Z = shape
    pred = (rs.standard_normal(shape) * 2).astype(np.float32)
    target = rs.randint(0, 18, (B, X, Y, Z)).astype(np.int64)
    target[rs.rand(B, X, Y, Z) < 0.1] = 255
    target[rs.rand(B, X, Y, Z) < 0.4] = 17
    cam = rs.rand(B, X, Y, Z) < 0.8
    return pred, target, cam


def stereo_inputs(seed, C=8, H=6, W=11, D=12, n_cams=2):
    """Seeded inputs of the DepthNet cost volume (view_transformer.py:546-604) at a reduced size:
    prev/curr stereo features (n_cams, C, H, W); a pinhole camera sized for the (4H x 4W) input image
    with a mild image augmentation; k2s_sensor (1,n,4,4) = a small ego motion; the cv_frustum
    (D,H,W,3) built like create_frustum(downsample=4) (view_transformer.py:84-112)."""
    rs = np.random.RandomState(seed)
    hi, wi = 4 * H, 4 * W
    prev = rs.standard_normal((n_cams, C, H, W)).astype(np.float32)
    curr = rs.standard_normal((n_cams, C, H, W)).astype(np.float32)
    K = np.tile(np.array([[0.9 * wi, 0, wi / 2.0], [0, 0.9 * wi, hi / 2.0], [0, 0, 1]], np.float32), (1, n_cams, 1, 1))
    post_rot = np.tile(np.eye(3, dtype=np.float32), (1, n_cams, 1, 1))
    post_tran = np.zeros((1, n_cams, 3), np.float32)
    k2s = np.tile(np.eye(4, dtype=np.float32), (1, n_cams, 1, 1))
    for c in range(n_cams):
        post_rot[0, c, 0, 0] = post_rot[0, c, 1, 1] = 0.95 + 0.03 * c
        post_tran[0, c, :2] = (1.5 - c, -0.7 * (c + 1))
        ang = 0.03 * (c + 1)
        k2s[0, c, :3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        k2s[0, c, :3, 3] = np.array([0.15 * (c + 1), 0.03, -0.8 - 0.4 * c], np.float32)
    d = (1.0 + 2.0 * np.arange(D)).astype(np.float32)
    x = np.linspace(0, wi - 1, W, dtype=np.float32)
    y = np.linspace(0, hi - 1, H, dtype=np.float32)
    frustum = np.stack(np.broadcast_arrays(x[None, None, :], y[None, :, None], d[:, None, None]), -1).astype(np.float32)
    return prev, curr, k2s, K, post_rot, post_tran, np.ascontiguousarray(frustum)


def seeded_module_state(module, seed):
    """Deterministic state dict for any torch module, identified by its (sorted) keys and shapes only, so that the
    reference module (tools/gen_golden.py) and the restated one (tests) get identical numbers without shipping weights:
    BN running_var / LayerNorm & BN weights positive, everything else ~N(0, 0.08^2)."""
    import torch
    rs = np.random.RandomState(seed)
    out = {}
    for k in sorted(module.state_dict().keys()):
        v = module.state_dict()[k]
        if 'num_batches_tracked' in k or 'relative_position_index' in k:
            continue
        shape = tuple(v.shape)
        if k.endswith('running_var'):
            a = rs.uniform(0.5, 1.5, shape)
        elif k.endswith('.weight') and v.dim() == 1:
            a = rs.uniform(0.5, 1.5, shape)
        else:
            a = rs.standard_normal(shape) * 0.08
        out[k] = torch.from_numpy(a.astype(np.float32))
    return out


def small_swin_cfg():
    """A 4-stage Swin at toy width with the PreWorld settings that change the data flow (out_indices (2,3),
    return_stereo_feat, shifted windows that need padding at 64x96 input)."""
    return dict(pretrain_img_size=224, patch_size=4, window_size=4, mlp_ratio=4, embed_dims=16, depths=[2, 2, 2, 2],
                num_heads=[2, 2, 4, 4], strides=(4, 2, 2, 2), out_indices=(2, 3), qkv_bias=True, qk_scale=None,
                patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, use_abs_pos_embed=False,
                return_stereo_feat=True, pretrain_style='official', output_missing_index_as_none=False)
