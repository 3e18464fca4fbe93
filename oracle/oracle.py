"""CPU oracle for the PreWorld camera->voxel occupancy hot path (numpy + C via ctypes).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by preworld_amd/.  Each function cites the reference
file:line it restates (paths relative to /root/reference).  State dicts use the
reference's parameter names so reference fixtures load unchanged.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libpw_oracle.so')
_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/pw_oracle.c with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, 'pw_oracle.c')
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(['make', '-C', _HERE, '-B', 'libpw_oracle.so'],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.pwo_version.restype = ctypes.c_int
        _lib.pwo_num_threads.restype = ctypes.c_int
        _lib.pwo_voxel_prepare.restype = ctypes.c_int64
        _lib.pwo_bev_pool_bp_prepare.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=c_f32p):
    return a.ctypes.data_as(t)


def num_threads():
    return lib().pwo_num_threads()


def set_num_threads(n):
    lib().pwo_set_num_threads(int(n))


# --------------------------------------------------------------------------- geometry
def create_frustum(depth_cfg, input_size, downsample):
    """view_transformer.py:84-112 -> (D,H,W,3) float32."""
    H_in, W_in = int(input_size[0]), int(input_size[1])
    downsample = int(downsample)
    Hf, Wf = H_in // downsample, W_in // downsample
    depth_cfg = [float(v) for v in depth_cfg]
    D = len(np.arange(depth_cfg[0], depth_cfg[1], depth_cfg[2]))
    out = np.empty((D, Hf, Wf, 3), np.float32)
    lib().pwo_create_frustum(D, Hf, Wf, ctypes.c_float(depth_cfg[0]),
                             ctypes.c_float(depth_cfg[2]), H_in, W_in, _p(out))
    return out


def grid_infos(grid_config):
    """view_transformer.py:66-82: lower bound, interval (fp32) and integer size."""
    axes = [grid_config[k] for k in ('x', 'y', 'z')]
    lower = np.array([c[0] for c in axes], np.float32)
    interval = np.array([c[2] for c in axes], np.float32)
    size = [int(np.float32((c[1] - c[0]) / c[2])) for c in axes]
    return lower, interval, size


def camera_matrices(sensor2ego, cam2imgs, post_rots):
    """inverse(post_rots), R @ inverse(K), t  (view_transformer.py:141-150)."""
    s = _f32(sensor2ego).reshape(-1, 4, 4)
    k = _f32(cam2imgs).reshape(-1, 3, 3)
    r = _f32(post_rots).reshape(-1, 3, 3)
    BN = s.shape[0]
    ipr = np.empty((BN, 3, 3), np.float32)
    comb = np.empty((BN, 3, 3), np.float32)
    tr = np.empty((BN, 3), np.float32)
    lib().pwo_camera_matrices(BN, _p(s), _p(k), _p(r), _p(ipr), _p(comb), _p(tr))
    return ipr, comb, tr


def lidar_coor(frustum, inv_post_rot, post_trans, combine, trans, bda, B, N):
    """view_transformer.py:114-153 with the 3x3 inverses already applied."""
    D, H, W, _ = frustum.shape
    coor = np.empty((B, N, D, H, W, 3), np.float32)
    lib().pwo_lidar_coor(B, N, D, H, W, _p(_f32(frustum)), _p(_f32(inv_post_rot)),
                         _p(_f32(post_trans)), _p(_f32(combine)), _p(_f32(trans)),
                         _p(_f32(bda)), _p(coor))
    return coor


def voxel_index(coor, lower, interval, size):
    B = coor.shape[0]
    n_per_b = int(np.prod(coor.shape[1:5]))
    vox = np.empty((B * n_per_b,), np.int32)
    lib().pwo_voxel_index(ctypes.c_size_t(n_per_b), B, _p(_f32(coor)), _p(_f32(lower)),
                          _p(_f32(interval)), size[0], size[1], size[2], _p(vox, c_i32p))
    return vox


def voxel_pooling_prepare_v2(coor, lower, interval, size):
    """view_transformer.py:203-261 -> (ranks_bev, ranks_depth, ranks_feat, starts, lengths)
    or five Nones when nothing is kept."""
    B, N, D, H, W, _ = coor.shape
    vox = voxel_index(coor, lower, interval, size)
    n_total = vox.shape[0]
    n_vox = B * size[0] * size[1] * size[2]
    rb = np.empty(n_total, np.int32)
    rd = np.empty(n_total, np.int32)
    rf = np.empty(n_total, np.int32)
    st = np.empty(min(n_total, n_vox), np.int32)
    ln = np.empty(min(n_total, n_vox), np.int32)
    ni = ctypes.c_int32(0)
    kept = lib().pwo_voxel_prepare(ctypes.c_size_t(n_total), n_vox, _p(vox, c_i32p), D, H * W,
                                   _p(rb, c_i32p), _p(rd, c_i32p), _p(rf, c_i32p),
                                   _p(st, c_i32p), _p(ln, c_i32p), ctypes.byref(ni))
    if kept == 0:
        return None, None, None, None, None
    return rb[:kept].copy(), rd[:kept].copy(), rf[:kept].copy(), \
        st[:ni.value].copy(), ln[:ni.value].copy()


def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                        interval_lengths, interval_starts):
    """bev_pool.cpp:30-57 argument order (lengths before starts); out is in/out."""
    c = feat.shape[-1]
    lib().pwo_bev_pool_v2_forward(c, len(interval_lengths), _p(depth), _p(feat),
                                  _p(ranks_depth, c_i32p), _p(ranks_feat, c_i32p),
                                  _p(ranks_bev, c_i32p), _p(interval_starts, c_i32p),
                                  _p(interval_lengths, c_i32p), _p(out))


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
    """bev_pool.py:86-92: returns (B,C,Z,Y,X)."""
    depth = _f32(depth)
    feat = _f32(feat)
    out = np.zeros(bev_feat_shape, np.float32)
    bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                        interval_lengths, interval_starts)
    return np.ascontiguousarray(out.transpose(0, 4, 1, 2, 3))


def bev_pool_v2_backward(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev):
    """bev_pool.py:43-83 -> (depth_grad, feat_grad). out_grad is (B,Z,Y,X,C)."""
    depth = _f32(depth)
    feat = _f32(feat)
    out_grad = _f32(out_grad)
    n = len(ranks_bev)
    n_pix = int(np.prod(feat.shape[:-1]))
    ob = np.empty(n, np.int32)
    od = np.empty(n, np.int32)
    of = np.empty(n, np.int32)
    st = np.empty(n_pix, np.int32)
    ln = np.empty(n_pix, np.int32)
    ni = lib().pwo_bev_pool_bp_prepare(ctypes.c_int64(n), n_pix, _p(ranks_bev, c_i32p),
                                       _p(ranks_depth, c_i32p), _p(ranks_feat, c_i32p),
                                       _p(ob, c_i32p), _p(od, c_i32p), _p(of, c_i32p),
                                       _p(st, c_i32p), _p(ln, c_i32p))
    dg = np.zeros_like(depth)
    fg = np.zeros_like(feat)
    lib().pwo_bev_pool_v2_backward(feat.shape[-1], ni, _p(out_grad), _p(depth), _p(feat),
                                   _p(od, c_i32p), _p(of, c_i32p), _p(ob, c_i32p),
                                   _p(st, c_i32p), _p(ln, c_i32p), _p(dg), _p(fg))
    return dg, fg


def lss_view_transform(depth, tran_feat, sensor2ego, cam2imgs, post_rots, post_trans, bda,
                       grid_config, input_size, downsample):
    """LSSViewTransformer.view_transform_core (accelerate=False) view_transformer.py:269-291.
    depth (B,N,D,H,W), tran_feat (B,N,C,H,W) -> bev_feat (B,C,Z,Y,X)."""
    B, N = depth.shape[:2]
    frustum = create_frustum(grid_config['depth'], input_size, downsample)
    lower, interval, size = grid_infos(grid_config)
    ipr, comb, tr = camera_matrices(sensor2ego, cam2imgs, post_rots)
    coor = lidar_coor(frustum, ipr, post_trans, comb, tr, bda, B, N)
    rb, rd, rf, st, ln = voxel_pooling_prepare_v2(coor, lower, interval, size)
    C = tran_feat.shape[2]
    if rb is None:
        return np.zeros((B, C, size[2], size[1], size[0]), np.float32)
    feat = np.ascontiguousarray(_f32(tran_feat).transpose(0, 1, 3, 4, 2))
    return bev_pool_v2(depth, feat, rd, rf, rb, (B, size[2], size[1], size[0], C), st, ln)


def depthnet_tail(x, D, C):
    """view_transformer.py:797-801: depth = x[:, :D].softmax(dim=1), tran_feat = x[:, D:D+C];
    the context is returned channels-last (BN,H,W,C) as bev_pool_v2 makes it (:189)."""
    x = _f32(x)
    z = x[:, :D]
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m, dtype=np.float32)
    s = np.zeros_like(m)
    for d in range(D):                       # sequential fp32 sum, the order a scalar loop uses
        s[:, 0] += e[:, d]
    depth = (e / s).astype(np.float32)
    feat = np.ascontiguousarray(x[:, D:D + C].transpose(0, 2, 3, 1))
    return depth, feat


# --------------------------------------------------------------------------- conv stack
def conv3d(x, w, bias=None, stride=1, pad=1):
    x = _f32(x)
    w = _f32(w)
    N, Cin, D, H, W = x.shape
    Cout, _, k = w.shape[:3]
    Do = (D + 2 * pad - k) // stride + 1
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    y = np.empty((N, Cout, Do, Ho, Wo), np.float32)
    b = _p(_f32(bias)) if bias is not None else None
    lib().pwo_conv3d(_p(x), _p(w), b, N, Cin, D, H, W, Cout, k, stride, pad, _p(y))
    return y


def bn_act(x, sd, prefix, residual=None, relu=False, eps=1e-5):
    """BatchNorm3d.eval() (+residual, +ReLU) in place on x."""
    N, C = x.shape[:2]
    sp = int(np.prod(x.shape[2:]))
    g, b = _f32(sd[prefix + '.weight']), _f32(sd[prefix + '.bias'])
    m, v = _f32(sd[prefix + '.running_mean']), _f32(sd[prefix + '.running_var'])
    r = _p(_f32(residual)) if residual is not None else None
    lib().pwo_bn_act(_p(x), N, C, ctypes.c_size_t(sp), _p(g), _p(b), _p(m), _p(v),
                     ctypes.c_float(eps), r, int(relu))
    return x


def relu_(x):
    lib().pwo_relu(_p(x), ctypes.c_size_t(x.size))
    return x


def conv_module(x, sd, prefix, stride=1, pad=1, relu=False, residual=None):
    """mmcv ConvModule conv(no bias)->BN3d->act (order conv,norm,act)."""
    y = conv3d(x, sd[prefix + '.conv.weight'], sd.get(prefix + '.conv.bias'), stride, pad)
    return bn_act(y, sd, prefix + '.bn', residual=residual, relu=relu)


def basic_block3d(x, sd, prefix, stride=1):
    """resnet.py:88-123: relu(conv2(conv1(x)) + downsample(x))."""
    if (prefix + '.downsample.conv.weight') in sd:
        identity = conv_module(x, sd, prefix + '.downsample', stride=stride)
    else:
        identity = x
    y = conv_module(x, sd, prefix + '.conv1', stride=stride, relu=True)
    return conv_module(y, sd, prefix + '.conv2', residual=identity, relu=True)


def custom_resnet3d(x, sd, prefix, num_layer, stride, backbone_output_ids=None):
    """resnet.py:126-184."""
    ids = range(len(num_layer)) if backbone_output_ids is None else backbone_output_ids
    feats = []
    for lid, nl in enumerate(num_layer):
        for b in range(nl):
            x = basic_block3d(x, sd, '%s.layers.%d.%d' % (prefix, lid, b),
                              stride=stride[lid] if b == 0 else 1)
        if lid in ids:
            feats.append(x)
    return feats


def upsample_trilinear(x, s):
    x = _f32(x)
    N, C, D, H, W = x.shape
    y = np.empty((N, C, D * s, H * s, W * s), np.float32)
    lib().pwo_upsample_trilinear_ac(_p(x), N * C, D, H, W, s, _p(y))
    return y


def lss_fpn3d(feats, sd, prefix):
    """lss_fpn.py:132-148 (levels=3)."""
    x8, x16, x32 = feats
    x = np.concatenate([x8, upsample_trilinear(x16, 2), upsample_trilinear(x32, 4)], axis=1)
    return conv_module(x, sd, prefix + '.conv', pad=0, relu=True)


def final_conv(x, sd, prefix='final_conv'):
    """preworld.py:72-79 ConvModule(bias=True, no norm, default act ReLU); then
    .permute(0,4,3,2,1) -> (B,X,Y,Z,C) (preworld_temporal_traj.py:222)."""
    y = conv3d(x, sd[prefix + '.conv.weight'], sd[prefix + '.conv.bias'], 1, 1)
    relu_(y)
    return np.ascontiguousarray(y.transpose(0, 4, 3, 2, 1))


def occ_head(voxel_feat, sd, prefix='occupancy_head'):
    """occupancy_head.py:124-177 with num_level=1, use_deblock=False: the soft-weight
    branch is numerically the identity (softmax over 1 channel, same-size interpolate).
    voxel_feat (1,32,X,Y,Z) -> logits (1,18,X,Y,Z)."""
    x = conv3d(voxel_feat, sd[prefix + '.occ_convs.0.0.weight'], None, 1, 1)
    bn_act(x, sd, prefix + '.occ_convs.0.1', relu=True)
    x = conv3d(x, sd[prefix + '.occ_pred_conv.0.weight'], None, 1, 0)
    bn_act(x, sd, prefix + '.occ_pred_conv.1', relu=True)
    return conv3d(x, sd[prefix + '.occ_pred_conv.3.weight'], None, 1, 0)


def linear(x, w, b, act=0):
    x = _f32(x)
    M = int(np.prod(x.shape[:-1]))
    In = x.shape[-1]
    Out = w.shape[0]
    y = np.empty(x.shape[:-1] + (Out,), np.float32)
    lib().pwo_linear(_p(x), ctypes.c_size_t(M), In, _p(_f32(w)),
                     _p(_f32(b)) if b is not None else None, Out, act, _p(y))
    return y


def plan_head(ego, sd, prefix='plan_head'):
    """preworld_temporal_traj.py:121-127: 21->256 ReLU ->256 ReLU ->32."""
    h = linear(ego, sd[prefix + '.0.weight'], sd[prefix + '.0.bias'], 1)
    h = linear(h, sd[prefix + '.2.weight'], sd[prefix + '.2.bias'], 1)
    return linear(h, sd[prefix + '.4.weight'], sd[prefix + '.4.bias'], 0)


def forecast_step(v, e, sd, prefix='fusion_head'):
    """preworld_temporal_traj.py:335-342; v (..., C) channels-last, e (C,)."""
    v = _f32(v)
    C = v.shape[-1]
    M = v.size // C
    out = np.empty_like(v)
    lib().pwo_forecast_step(_p(v), ctypes.c_size_t(M), C, _p(_f32(e)),
                            _p(_f32(sd[prefix + '.0.weight'])), _p(_f32(sd[prefix + '.0.bias'])),
                            _p(_f32(sd[prefix + '.2.weight'])), _p(_f32(sd[prefix + '.2.bias'])),
                            _p(out))
    return out


def downscale_module(fused_xyzc, sd, prefix='downscale'):
    """A20  DownScaleModule3DCustom, mmdet3d/models/heads/occupancy_head.py:180-200:
    (B,X,Y,Z,C) -> permute to (B,C,X,Y,Z) -> 3x Conv3d(k=2,s=2,bias) -> AdaptiveAvgPool3d(1) -> (B,4C).
    Also returns the three conv outputs (B,C',X',Y',Z')."""
    x = np.ascontiguousarray(np.transpose(_f32(fused_xyzc), (0, 4, 1, 2, 3)))
    levels = []
    for k in ('downscale1', 'downscale2', 'downscale3'):
        x = conv3d(x, sd['%s.%s.weight' % (prefix, k)], sd['%s.%s.bias' % (prefix, k)], stride=2, pad=0)
        levels.append(x)
    B, C = x.shape[:2]
    pooled = x.reshape(B, C, -1).sum(axis=2, dtype=np.float32) / np.float32(x[0, 0].size)
    return pooled.astype(np.float32), levels


def traj_branch(fused_xyzc, ego_feat, sd):
    """A20  preworld_temporal_traj.py:457-470: down = downscale(fused); h = ego_fusion_head(cat(identity,
    down)); fused_ego = identity + h; pred_traj = traj_head(fused_ego)."""
    down, _ = downscale_module(fused_xyzc, sd)
    h = np.concatenate([_f32(ego_feat), down], axis=-1)
    for i in (0, 2, 4):
        h = linear(h, sd['ego_fusion_head.%d.weight' % i], sd['ego_fusion_head.%d.bias' % i], 2)
    res = linear(h, sd['ego_fusion_head.6.weight'], sd['ego_fusion_head.6.bias'], 0)
    fused_ego = _f32(ego_feat) + res
    t = linear(fused_ego, sd['traj_head.0.weight'], sd['traj_head.0.bias'], 2)
    return linear(t, sd['traj_head.2.weight'], sd['traj_head.2.bias'], 0), fused_ego


def argmax_u8(x):
    x = _f32(x)
    C = x.shape[-1]
    out = np.empty(x.shape[:-1], np.uint8)
    lib().pwo_argmax_u8(_p(x), ctypes.c_size_t(x.size // C), C, _p(out, c_u8p))
    return out


def occ_decode(voxel_feats_xyzc, sd):
    """preworld_temporal_traj.py:306-324: OccHead on (1,C,X,Y,Z), argmax over classes.
    returns (occ uint8 (X,Y,Z), logits (X,Y,Z,18))."""
    vf = np.ascontiguousarray(voxel_feats_xyzc[0].transpose(3, 0, 1, 2))[None]
    logits = occ_head(vf, sd)[0].transpose(1, 2, 3, 0)
    logits = np.ascontiguousarray(logits)
    return argmax_u8(logits), logits


def attribute_decode(voxel_feats, sd, test_threshold=8.5, num_classes=18):
    """preworld_temporal_traj.py:231-250: density/semantic MLP decode."""
    h = linear(voxel_feats, sd['density_mlp.0.weight'], sd['density_mlp.0.bias'], 2)
    dens = linear(h, sd['density_mlp.2.weight'], sd['density_mlp.2.bias'], 2)[..., 0]
    h = linear(voxel_feats, sd['semantic_mlp.0.weight'], sd['semantic_mlp.0.bias'], 2)
    sem = linear(h, sd['semantic_mlp.2.weight'], sd['semantic_mlp.2.bias'], 0)
    occ = np.where(dens > test_threshold, argmax_u8(sem), num_classes - 1).astype(np.uint8)
    return occ, dens, sem


def encoder_forward(bev_adj, bev_key, sd):
    """bevdet_occ.py:266-267 + bevdet.py:52-58: cat([adj,key]) -> backbone -> neck."""
    x = np.concatenate([bev_adj, bev_key], axis=1)
    feats = custom_resnet3d(x, sd, 'img_bev_encoder_backbone', [1, 2, 4], [1, 2, 2])
    return lss_fpn3d(feats, sd, 'img_bev_encoder_neck')


def pre_process(bev, sd):
    """bevdet_occ.py:164: CustomResNet3D(numC_input=32, num_layer=[1], stride=[1])."""
    return custom_resnet3d(bev, sd, 'pre_process_net', [1], [1])[0]


def preworld4d_decode(voxel_feats, ego, sd, n_steps=6, post_finetune=True):
    """preworld_temporal_traj.py:303-368 (post-finetune) / :224-301 (attribute decode).
    voxel_feats (1,X,Y,Z,C).  Returns list of uint8 (X,Y,Z) states and list of features."""
    e = plan_head(_f32(ego).reshape(1, -1), sd)[0]
    states, feats = [], [voxel_feats]
    v = voxel_feats
    dec = (lambda f: occ_decode(f, sd)[0]) if post_finetune else \
        (lambda f: attribute_decode(f, sd)[0][0])
    states.append(dec(v))
    for _ in range(n_steps):
        v = forecast_step(v, e, sd)
        feats.append(v)
        states.append(dec(v))
    return states, feats


# --------------------------------------------------------------------------- detector-level composition
def prepare_inputs(inputs, num_frame=3, temporal_frame=2):
    """BEVStereo4DOCC.prepare_inputs (bevdet_occ.py:88-139) in numpy: poses of every sweep expressed in the KEY frame's ego
    system, fp64 algebra rounded to fp32 at the end (:103-106), and curr2adjsensor of the stereo pairs (:108-124).
    inputs = (imgs (B, N*T, C, H, W), sensor2egos (B, T*N, 4, 4), ego2globals, intrins (B, T*N, 3, 3), post_rots, post_trans
    (B, T*N, 3), bda).  Returns per-frame lists (sensor2keyegos, ego2globals, intrins, post_rots, post_trans), bda, curr2adjsensor."""
    B = inputs[0].shape[0]
    N = inputs[0].shape[1] // num_frame
    s2e = np.asarray(inputs[1], np.float32).reshape(B, num_frame, N, 4, 4)
    e2g = np.asarray(inputs[2], np.float32).reshape(B, num_frame, N, 4, 4)
    keyego2global = e2g[:, 0, 0][:, None, None].astype(np.float64)
    s2k = (np.linalg.inv(keyego2global) @ e2g.astype(np.float64) @ s2e.astype(np.float64)).astype(np.float32)
    cur = e2g[:, :temporal_frame].astype(np.float64) @ s2e[:, :temporal_frame].astype(np.float64)
    adj = e2g[:, 1:temporal_frame + 1].astype(np.float64) @ s2e[:, 1:temporal_frame + 1].astype(np.float64)
    c2a = (np.linalg.inv(adj) @ cur).astype(np.float32)
    per = lambda a, tail: [np.asarray(a, np.float32).reshape((B, num_frame, N) + tail)[:, t] for t in range(num_frame)]   # noqa: E731
    return ([s2k[:, t] for t in range(num_frame)], [e2g[:, t] for t in range(num_frame)], per(inputs[3], (3, 3)),
            per(inputs[4], (3, 3)), per(inputs[5], (3,)), np.asarray(inputs[6], np.float32),
            [c2a[:, t] for t in range(temporal_frame)] + [None] * (num_frame - temporal_frame))


def get_mlp_input(sensor2ego, intrin, post_rot, post_tran, bda):
    """LSSViewTransformerBEVDepth.get_mlp_input (view_transformer.py:713-734): 15 camera scalars + sensor2ego[:3, :] = 27"""
    B, N = sensor2ego.shape[:2]
    b = np.broadcast_to(bda.reshape(B, 1, 3, 3), (B, N, 3, 3))
    v = np.stack([intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2], intrin[:, :, 1, 2], post_rot[:, :, 0, 0],
                  post_rot[:, :, 0, 1], post_tran[:, :, 0], post_rot[:, :, 1, 0], post_rot[:, :, 1, 1], post_tran[:, :, 1],
                  b[:, :, 0, 0], b[:, :, 0, 1], b[:, :, 1, 0], b[:, :, 1, 1], b[:, :, 2, 2]], axis=-1)
    return np.concatenate([v, sensor2ego[:, :, :3, :].reshape(B, N, -1)], axis=-1).astype(np.float32)


def detector_simple_test(inputs, depthnet, ego, sd, grid_config, input_size, downsample, detector='PreWorld4DTraj',
                         post_finetune=True, with_prev=True, test_threshold=8.5, D=88, C=32):
    """PreWorld4DTraj.simple_test (preworld_temporal_traj.py:212-370) / PreWorld.simple_test (preworld.py:159-226) on top of
    BEVStereo4DOCC.extract_img_feat's frame loop (bevdet_occ.py:167-269), downstream of the DepthNet:
    depthnet(k, mlp_input) -> the (B*N, D+C, H, W) DepthNet output of the k-th processed frame (frames are visited
    extra-reference, adjacent, key: :191; only adjacent and key reach the DepthNet; with_prev=False visits the key frame only
    and feeds zeros as the adjacent BEV feature, :243-258).  Returns (result dict like the reference's, bev_feat, voxel_feats)."""
    s2k, e2g, K, pr, pt, bda, _ = prepare_inputs(inputs)
    bevs, k = [], 0
    for fid in (1, 0):
        if fid != 0 and not with_prev:
            continue
        mlp = get_mlp_input(s2k[0], K[fid], pr[fid], pt[fid], bda)          # always the KEY frame's sensor2keyego (:197-199)
        x = depthnet(k, mlp)
        k += 1
        B, N = s2k[fid].shape[:2]
        depth, feat_cl = depthnet_tail(x, D, C)
        H, W = depth.shape[-2:]
        tran = np.ascontiguousarray(feat_cl.transpose(0, 3, 1, 2)).reshape(B, N, C, H, W)
        bev = lss_view_transform(depth.reshape(B, N, D, H, W), tran, s2k[fid], K[fid], pr[fid], pt[fid], bda, grid_config,
                                 input_size, downsample)
        bevs.append(pre_process(bev, sd))
    if not with_prev:
        bevs = [np.zeros_like(bevs[0])] + bevs                                  # [zeros, key]
    bev_feat = encoder_forward(bevs[0], bevs[1], sd)                            # cat([adjacent, key]) (:266)
    vf = final_conv(bev_feat, sd)                                               # (B,X,Y,Z,C) like :222
    res = {}
    if detector == 'PreWorld':
        occ = occ_decode(vf, sd)[0] if post_finetune else attribute_decode(vf, sd, test_threshold)[0][0]
        res['semantic_occ'] = [occ]
        res['geo_occ'] = [np.where(occ != 17, 0, 17).astype(np.uint8)]
        return res, bev_feat, vf
    e = plan_head(_f32(ego).reshape(1, -1), sd)[0]
    v = vf
    dec = (lambda f: occ_decode(f, sd)[0]) if post_finetune else (lambda f: attribute_decode(f, sd, test_threshold)[0][0])
    for step in range(7):
        if step:
            v = forecast_step(v, e, sd)
        occ = dec(v)
        name = step if (post_finetune or step == 0) else step + 1               # :361 names k+1, :294 names k+2
        res['semantic_occ_%ds' % name] = [occ]
        res['geo_occ_%ds' % name] = [np.where(occ != 17, 0, 17).astype(np.uint8)]
    return res, bev_feat, vf


# --------------------------------------------------------------------------- render head
class NerfConsts:
    """nerf_head.py:105-162 buffers/constants for a given config."""

    def __init__(self, point_cloud_range=(-40, -40, -1, 40, 40, 5.4), voxel_size=0.4, radius=39,
                 step_size=0.5, alpha_init=1e-6, fast_color_thres=1e-7, world_size=(200, 200, 16)):
        xyz_min = np.array(point_cloud_range[:3], np.float32)
        xyz_max = np.array(point_cloud_range[3:], np.float32)
        rng = xyz_max - xyz_min
        self.bg_len = np.float32((rng[0] // 2 - radius) / radius)
        self.radius = radius
        self.scene_center = ((xyz_min + xyz_max) * np.float32(0.5)).astype(np.float32)
        self.scene_radius = np.array([radius] * 3, np.float32)
        z_ = np.float32(rng[2] / rng[0])
        self.xyz_min = np.array([-1 - self.bg_len, -1 - self.bg_len, -z_], np.float32)
        self.xyz_max = np.array([1 + self.bg_len, 1 + self.bg_len, z_], np.float32)
        self.act_shift = np.float32(np.log(1 / (1 - alpha_init) - 1))
        self.step_size = step_size
        self.world_len = world_size[0]
        self.fast_color_thres = fast_color_thres

    def t_table(self):
        """nerf_head.py:35-43 (uses torch.linspace semantics, fp32)."""
        # the reference evaluates this with bg_len a 0-dim fp32 tensor: every op rounds to fp32
        f = np.float32
        N_inner = int(f(f(f(2) / f(f(2) + f(f(2) * self.bg_len))) * f(self.world_len)) / f(self.step_size)) + 1
        N_outer = N_inner // 15

        def linspace(a, b, n):
            # nerf_head.py:37-38 call torch.linspace; its vectorised CPU kernel rounds a few
            # entries differently from the scalar formula, so use torch itself when present.
            try:
                import torch
                return torch.linspace(a, b, n).numpy()
            except ImportError:
                a, b = np.float32(a), np.float32(b)
                step = np.float32((b - a) / np.float32(n - 1))
                i = np.arange(n)
                lo = (a + step * i.astype(np.float32)).astype(np.float32)
                hi = (b - step * (n - i - 1).astype(np.float32)).astype(np.float32)
                return np.where(i < n // 2, lo, hi).astype(np.float32)

        b_inner = linspace(0, 2, N_inner + 1)
        b_outer = (np.float32(2) / linspace(1, 1 / 64, N_outer + 1)).astype(np.float32)
        t = np.concatenate([(b_inner[1:] + b_inner[:-1]) * np.float32(0.5),
                            (b_outer[1:] + b_outer[:-1]) * np.float32(0.5)]).astype(np.float32)
        return t


def sample_ray(rays_o, rays_d, consts, bda):
    t = consts.t_table()
    R, S = rays_o.shape[0], t.shape[0]
    pts = np.empty((R, S, 3), np.float32)
    inner = np.empty((R, S), np.uint8)
    lib().pwo_sample_ray(R, S, _p(_f32(rays_o)), _p(_f32(rays_d)), _p(t),
                         _p(consts.scene_center), _p(consts.scene_radius),
                         ctypes.c_float(consts.bg_len), _p(_f32(bda)), _p(pts), _p(inner, c_u8p))
    return pts, inner.astype(bool), t


def cumdist_thres(dist, thres):
    dist = _f32(dist)
    R, S = dist.shape
    m = np.empty((R, S), np.uint8)
    lib().pwo_cumdist_thres(R, S, _p(dist), ctypes.c_float(thres), _p(m, c_u8p))
    return m.astype(bool)


def grid_sample_xyz(grid_cxyz, xyz, consts):
    g = _f32(grid_cxyz)
    C, X, Y, Z = g.shape
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty((xyz.shape[0], C), np.float32)
    lib().pwo_grid_sample_xyz(_p(g), C, X, Y, Z, _p(xyz), ctypes.c_size_t(xyz.shape[0]),
                              _p(consts.xyz_min), _p(consts.xyz_max), _p(out))
    return out


def raw2alpha(density, shift, interval):
    d = _f32(density).reshape(-1)
    e = np.empty_like(d)
    a = np.empty_like(d)
    lib().pwo_raw2alpha(_p(d), ctypes.c_float(shift), ctypes.c_float(interval),
                        ctypes.c_size_t(d.size), _p(e), _p(a))
    return e, a


def raw2alpha_backward(exp_d, grad_back, interval):
    e = _f32(exp_d)
    g = _f32(grad_back)
    out = np.empty_like(e)
    lib().pwo_raw2alpha_backward(_p(e), _p(g), ctypes.c_float(interval),
                                 ctypes.c_size_t(e.size), _p(out))
    return out


def alpha2weight(alpha, ray_id, n_rays):
    a = _f32(alpha)
    rid = np.ascontiguousarray(ray_id, dtype=np.int64)
    n = a.size
    w = np.empty(n, np.float32)
    T = np.empty(n, np.float32)
    last = np.empty(n_rays, np.float32)
    i_s = np.empty(n_rays, np.int64)
    i_e = np.empty(n_rays, np.int64)
    lib().pwo_alpha2weight(_p(a), _p(rid, c_i64p), ctypes.c_size_t(n), n_rays, _p(w), _p(T),
                           _p(last), _p(i_s, c_i64p), _p(i_e, c_i64p))
    return w, T, last, i_s, i_e


def alpha2weight_backward(alpha, weight, T, last, i_s, i_e, n_rays, grad_w, grad_last):
    n = alpha.size
    g = np.empty(n, np.float32)
    lib().pwo_alpha2weight_backward(_p(_f32(alpha)), _p(_f32(weight)), _p(_f32(T)),
                                    _p(_f32(last)), _p(i_s, c_i64p), _p(i_e, c_i64p), n_rays,
                                    _p(_f32(grad_w)), _p(_f32(grad_last)), ctypes.c_size_t(n),
                                    _p(g))
    return g


def segment_sum(src, index, n_seg):
    src = _f32(src)
    C = 1 if src.ndim == 1 else src.shape[1]
    idx = np.ascontiguousarray(index, dtype=np.int64)
    out = np.empty((n_seg, C), np.float32)
    lib().pwo_segment_sum(_p(src), _p(idx, c_i64p), ctypes.c_size_t(idx.size), C, n_seg, _p(out))
    return out[:, 0] if src.ndim == 1 else out


def render_one_scene(rays_o, rays_d, bda, density, semantic, color, consts):
    """nerf_head.py:165-269.  density (X,Y,Z), semantic (X,Y,Z,17), color (X,Y,Z,3)."""
    pts, inner, t = sample_ray(rays_o, rays_d, consts, bda)
    R, S = inner.shape
    mask = inner.copy()
    dist_thres = np.float32((2 + 2 * float(consts.bg_len)) / consts.world_len * consts.step_size * 0.95)
    d = pts[:, 1:] - pts[:, :-1]
    dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    mask[:, 1:] |= cumdist_thres(dist, dist_thres)
    ray_id = np.repeat(np.arange(R), S).reshape(R, S)[mask]
    step_id = np.tile(np.arange(S), R).reshape(R, S)[mask]
    xyz = pts[mask]
    tt = np.broadcast_to(t[None], (R, S))[mask]
    dens = grid_sample_xyz(density[None], xyz, consts)[:, 0]
    sem = grid_sample_xyz(np.ascontiguousarray(semantic.transpose(3, 0, 1, 2)), xyz, consts)
    col = grid_sample_xyz(np.ascontiguousarray(color.transpose(3, 0, 1, 2)), xyz, consts)
    _, alpha = raw2alpha(dens, consts.act_shift, 0.5)
    m1 = alpha > consts.fast_color_thres
    _alpha_all, _rid_all = alpha, ray_id
    ray_id, step_id, tt, dens, alpha, sem, col = [a[m1] for a in (ray_id, step_id, tt, dens, alpha, sem, col)]
    weights, T, last, i_s, i_e = alpha2weight(alpha, ray_id, R)
    m2 = weights > consts.fast_color_thres
    trace = dict(ray_id=ray_id, alpha=alpha, T=T, weights=weights, alpha_all=_alpha_all, ray_id_all=_rid_all)
    ray_id, step_id, tt, alpha, sem, col, weights = [a[m2] for a in (ray_id, step_id, tt, alpha, sem, col, weights)]
    s = (1 - 1 / (1 + tt)).astype(np.float32)
    return dict(alphainv_last=last, weights=weights, ray_id=ray_id, step_id=step_id, s=s, t=tt,
                N_ray=R, semantic=sem, color=col, mask1=m1, mask2=m2, sample_mask=mask, trace=trace)


def render_near_tie_rays(res, rel=2e-3):
    """Rays of a render_one_scene result on which one of the three data-dependent decisions of the reference sits within `rel` of
    its threshold: alpha > 1e-7 (nerf_head.py:229), weight > 1e-7 (:244), T_cum < 1e-3 (render_utils_kernel.cu:597).  A correct
    fp32 implementation with another evaluation order may decide such a sample the other way; on every other ray the kept
    sample set must be identical.  Returns a bool (N_ray,) array."""
    tr, R = res['trace'], res['N_ray']
    tie = np.zeros(R, bool)
    a = tr['alpha_all']
    tie[tr['ray_id_all'][np.abs(a - 1e-7) <= rel * 1e-7]] = True
    w = tr['weights']
    visited = (tr['T'] != 1) | (w != 0) | (np.r_[True, tr['ray_id'][1:] != tr['ray_id'][:-1]] if len(w) else np.zeros(0, bool))
    tie[tr['ray_id'][visited & (np.abs(w - 1e-7) <= rel * 1e-7)]] = True
    t_next = tr['T'].astype(np.float64) * (1.0 - tr['alpha'].astype(np.float64))
    tie[tr['ray_id'][visited & (np.abs(t_next - 1e-3) <= rel * 1e-3)]] = True
    return tie


def render_outputs(res, consts):
    """nerf_head.py:331-353."""
    depth = (segment_sum(res['weights'] * res['s'], res['ray_id'], res['N_ray'])
             + np.float32(1e-7)) * np.float32(consts.radius)
    sem = segment_sum(res['weights'][:, None] * res['semantic'], res['ray_id'], res['N_ray'])
    col = segment_sum(res['weights'][:, None] * res['color'], res['ray_id'], res['N_ray'])
    return depth.astype(np.float32), sem, col


def flatten_eff_distloss(w, m, interval, ray_id):
    """torch_efficient_distloss.flatten_eff_distloss (un-vendored dependency, PyPI
    torch_efficient_distloss; algorithm = Mip-NeRF-360 distortion loss, segmented prefix sums):
    loss = sum_i [ (1/3) interval w_i^2 + 2 w_i (m_i Wcum_i - WMcum_i) ] / n_rays,
    with exclusive per-ray prefix sums Wcum, WMcum.  PARITY UNPINNED."""
    w = w.astype(np.float64)
    m = m.astype(np.float64)
    n_rays = int(ray_id.max()) + 1 if len(ray_id) else 1
    wm = w * m
    cw = np.cumsum(w)
    cwm = np.cumsum(wm)
    first = np.r_[True, ray_id[1:] != ray_id[:-1]] if len(ray_id) else np.zeros(0, bool)
    seg_start_cw = np.where(first, cw - w, 0)
    seg_start_cwm = np.where(first, cwm - wm, 0)
    idx = np.maximum.accumulate(np.where(first, np.arange(len(w)), 0)) if len(w) else np.zeros(0, int)
    w_pre = (cw - w) - seg_start_cw[idx]
    wm_pre = (cwm - wm) - seg_start_cwm[idx]
    uni = (1 / 3) * interval * (w ** 2)
    bi = 2 * w * (m * w_pre - wm_pre)
    return float((uni.sum() + bi.sum()) / n_rays)


def nerf_losses(res, depth, sem, col, target_depth, target_sem, target_col, class_weights,
                weight_entropy_last=0.01, weight_distortion=0.01):
    """nerf_head.py:271-299 (fp64 numpy restatement of scalar losses)."""
    out = {}
    d = np.log(depth.astype(np.float64) + 1e-7) - np.log(target_depth.astype(np.float64))
    out['loss_render_depth'] = float(np.sqrt((d ** 2).mean() - 0.85 * d.mean() ** 2))
    z = sem.astype(np.float64)
    z = z - z.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    tgt = target_sem.astype(np.int64)
    wts = class_weights[tgt]
    out['loss_render_semantic'] = float(-(wts * logp[np.arange(len(tgt)), tgt]).sum() / wts.sum())
    out['loss_render_color'] = float(np.abs(col.astype(np.float64) - target_col).mean(0).sum())
    p = np.clip(res['alphainv_last'].astype(np.float64), 1e-6, 1 - 1e-6)
    out['loss_sdf_entropy'] = float(weight_entropy_last * -(p * np.log(p) + (1 - p) * np.log(1 - p)).mean())
    n_max = len(res['t'])
    out['loss_sdf_distortion'] = weight_distortion * flatten_eff_distloss(
        res['weights'], res['s'], 1 / max(n_max, 1), res['ray_id'])
    return out


def nerf_head_forward(density, semantic, color, rays, bda, class_weights, consts=None, if_temporal=False, interval=0,
                      use_depth_sup=True, weight_depth=1.0, weight_semantic=1.0, weight_color=1.0, weight_entropy_last=0.01,
                      weight_distortion=0.01):
    """NerfHead.forward (nerf_head.py:355-420), numpy: density (B,X,Y,Z), semantic (B,X,Y,Z,17), color (B,X,Y,Z,3), rays (B,R,16),
    bda (B,3,3).  Per batch element: lidar depths beyond 52 m are dropped (:379, in place on `rays` as the reference does), rays
    with gt_depth > 0 are rendered (:380, :172-174), compute_loss[_temporal] (:271-329), the per-key sum over the batch divided by
    the batch size (:411-418).  Pinned by tests/golden/nerf_losses_small.npz (the imported reference's own forward)."""
    consts = consts or NerfConsts()
    sfx = '_%ds' % int(interval) if if_temporal else ''
    weights = dict(loss_render_depth=weight_depth, loss_render_semantic=weight_semantic, loss_render_color=weight_color,
                   loss_sdf_entropy=1.0, loss_sdf_distortion=1.0)            # (nerf_losses applies the last two itself)
    losses = {}
    for b in range(rays.shape[0]):
        gt_depth = rays[b, :, 2]
        gt_depth[gt_depth > 52] = 0
        m = gt_depth > 0
        res = render_one_scene(_f32(rays[b, :, 4:7][m]), _f32(rays[b, :, 7:10][m]), bda[b], density[b], semantic[b], color[b], consts)
        depth, sem, col = render_outputs(res, consts)
        single = nerf_losses(res, depth, sem, col, gt_depth[m], rays[b, :, 3][m], rays[b, :, 13:16][m], class_weights,
                             weight_entropy_last, weight_distortion)
        if not use_depth_sup:
            single.pop('loss_render_depth')
        if not weight_entropy_last > 0:
            single.pop('loss_sdf_entropy')
        if not weight_distortion > 0:
            single.pop('loss_sdf_distortion')
        for k, v in single.items():
            losses[k + sfx] = losses.get(k + sfx, 0.0) + v * weights[k]
    return {k: v / semantic.shape[0] for k, v in losses.items()}


# --------------------------------------------------------------------------- metric
CLASS_NAMES = ['others', 'barrier', 'bicycle', 'bus', 'car', 'construction_vehicle', 'motorcycle',
               'pedestrian', 'traffic_cone', 'trailer', 'truck', 'driveable_surface', 'other_flat',
               'sidewalk', 'terrain', 'manmade', 'vegetation', 'free']


class MetricMIoU:
    """occ_metrics.py:52-185 (hist_info / per_class_iu / add_batch / count_miou)."""

    def __init__(self, num_classes=18, use_lidar_mask=False, use_image_mask=False):
        self.num_classes = num_classes
        self.use_lidar_mask = use_lidar_mask
        self.use_image_mask = use_image_mask
        self.hist = np.zeros((num_classes, num_classes), np.int64)
        self.cnt = 0

    def add_batch(self, pred, gt, mask_lidar=None, mask_camera=None):
        self.cnt += 1
        mask = mask_camera if self.use_image_mask else (mask_lidar if self.use_lidar_mask else None)
        p = np.ascontiguousarray(pred, np.uint8).reshape(-1)
        g = np.ascontiguousarray(gt, np.uint8).reshape(-1)
        m = np.ascontiguousarray(mask, np.uint8).reshape(-1) if mask is not None else None
        lib().pwo_confusion_hist(_p(p, c_u8p), _p(g, c_u8p), _p(m, c_u8p) if m is not None else None,
                                 ctypes.c_size_t(p.size), self.num_classes, _p(self.hist, c_i64p))

    def per_class_iu(self):
        h = self.hist.astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.diag(h) / (h.sum(1) + h.sum(0) - np.diag(h))

    def count_miou(self):
        iu = self.per_class_iu()
        return round(float(np.nanmean(iu[:self.num_classes - 1])) * 100, 2), iu


# --------------------------------------------------------------------------- stereo cost volume
def stereo_cost_volume(prev, curr, frustum, k2s_sensor, intrins, post_rots, post_trans, bias=0.0):
    """mmdet3d/models/necks/view_transformer.py:546-604 restated in numpy (float32 steps).
    prev/curr (BN,C,H,W); frustum (D,H,W,3); k2s (B,N,4,4); intrins/post_rots (B,N,3,3); post_trans (B,N,3)."""
    f32 = np.float32
    prev, curr = _f32(prev), _f32(curr)
    BN, C, H, W = curr.shape
    D = frustum.shape[0]
    hi, wi = H * 4, W * 4
    k2s = _f32(k2s_sensor).reshape(BN, 4, 4)
    K = _f32(intrins).reshape(BN, 3, 3)
    pr = _f32(post_rots).reshape(BN, 3, 3)
    pt = _f32(post_trans).reshape(BN, 3)
    ipr, comb, tr = camera_matrices(k2s[None], K[None], pr[None])
    ipr, comb, tr = ipr.reshape(BN, 3, 3), comb.reshape(BN, 3, 3), tr.reshape(BN, 3)
    out = np.empty((BN, D, H, W), f32)

    def mv(m, v):                                  # (3,3) x (...,3), left-to-right fp32 sums
        return np.stack([(m[r, 0] * v[..., 0] + m[r, 1] * v[..., 1]) + m[r, 2] * v[..., 2] for r in range(3)], -1).astype(f32)
    for bn in range(BN):
        p = (_f32(frustum) - pt[bn]).astype(f32)
        q = mv(ipr[bn], p)
        r = np.stack([q[..., 0] * q[..., 2], q[..., 1] * q[..., 2], q[..., 2]], -1).astype(f32)
        q = (mv(comb[bn], r) + tr[bn]).astype(f32)
        neg = q[..., 2] < f32(1e-3)
        r = mv(K[bn], q)
        u, v = r[..., 0] / r[..., 2], r[..., 1] / r[..., 2]
        x = ((pr[bn, 0, 0] * u + pr[bn, 0, 1] * v) + pt[bn, 0]).astype(f32)
        y = ((pr[bn, 1, 0] * u + pr[bn, 1, 1] * v) + pt[bn, 1]).astype(f32)
        px = (x / f32(wi - 1.0) * f32(2.0) - f32(1.0)).astype(f32)
        py = (y / f32(hi - 1.0) * f32(2.0) - f32(1.0)).astype(f32)
        px[neg] = -2
        py[neg] = -2
        ix = ((px + f32(1)) / f32(2) * f32(W - 1)).astype(f32)
        iy = ((py + f32(1)) / f32(2) * f32(H - 1)).astype(f32)
        x0, y0 = np.floor(ix), np.floor(iy)
        tx1, tx0 = ix - x0, (x0 + 1) - ix
        ty1, ty0 = iy - y0, (y0 + 1) - iy
        x0i = np.clip(x0, -2, W).astype(np.int64)
        y0i = np.clip(y0, -2, H).astype(np.int64)
        cost = np.zeros((D, H, W), f32)
        last0 = None
        for c0 in range(0, C, 4):
            grp = np.zeros((D, H, W), f32)
            for k in range(4):
                img = prev[bn, c0 + k]
                acc = np.zeros((D, H, W), f32)
                for (dx, dy, wgt) in ((0, 0, tx0 * ty0), (1, 0, tx1 * ty0), (0, 1, tx0 * ty1), (1, 1, tx1 * ty1)):
                    xi, yi = x0i + dx, y0i + dy
                    ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                    val = img[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
                    acc = np.where(ok, acc + val * wgt.astype(f32), acc).astype(f32)
                if k == 0:
                    last0 = acc
                grp = (grp + np.abs(curr[bn, c0 + k][None] - acc)).astype(f32)
            cost = (cost + grp).astype(f32)
        if bias != 0:
            cost = np.where(last0 == 0, cost + f32(bias), cost).astype(f32)
        z = -cost
        e = np.exp(z - z.max(axis=0, keepdims=True), dtype=f32)
        out[bn] = e / e.sum(axis=0, keepdims=True)
    return out


# --------------------------------------------------------------------------- voxel-grid training losses
def _bce1(x):
    with np.errstate(divide='ignore'):
        return -np.maximum(np.log(np.float64(x)), -100.0)


def voxel_losses(pred, target, class_weights=None, ignore_index=255, empty_idx=17, camera_mask=None):
    """mmdet3d/models/detectors/loss.py:20-113 restated (CE_ssc_loss, sem_scal_loss, geo_scal_loss);
    pred (B,C,X,Y,Z) logits, target (B,X,Y,Z) ints.  Sums in float64 (the reference sums fp32 tensors;
    the difference is inside the stated tolerance)."""
    z = np.asarray(pred, np.float64)
    B, C = z.shape[:2]
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m)
    p = e / e.sum(axis=1, keepdims=True)
    t = np.asarray(target).astype(np.int64)
    cam = np.ones_like(t, bool) if camera_mask is None else np.asarray(camera_mask).astype(bool)
    w = np.ones(C) if class_weights is None else np.asarray(class_weights, np.float64)
    valid = t != ignore_index
    logp = np.log(p)
    tt = np.where(valid, t, 0)
    lp_t = np.take_along_axis(logp, tt[:, None], axis=1)[:, 0]
    ce = (w[tt] * -lp_t)[valid].sum() / w[tt][valid].sum()
    M = valid & cam
    loss, count = 0.0, 0
    for i in range(C):
        pi = p[:, i][M]
        ct = (t[M] == i).astype(np.float64)
        if ct.sum() > 0:
            count += 1
            nom = (pi * ct).sum()
            lc = 0.0
            if pi.sum() > 0:
                lc += _bce1(nom / pi.sum())
            lc += _bce1(nom / ct.sum())
            if (1 - ct).sum() > 0:
                lc += _bce1(((1 - pi) * (1 - ct)).sum() / (1 - ct).sum())
            loss += lc
    sem = loss / count
    pe = p[:, empty_idx]
    nt = ((t != empty_idx) & cam).astype(np.float64)
    inter = (nt * (1 - pe)).sum()
    geo = _bce1(inter / (1 - pe).sum()) + _bce1(inter / nt.sum()) + _bce1(((1 - nt) * pe).sum() / (1 - nt).sum())
    return float(ce), float(sem), float(geo)


def focal_radial_map(H, W):
    """CustomFocalLoss.__init__ (mmdet3d/models/loss_utils/focal_loss.py:196-203): c = |(h - H/2, w - W/2)| / max + 1
    on the (H, W) = first two grid axes (the reference hard-codes 200 x 200)."""
    xy, yx = np.meshgrid(np.arange(H, dtype=np.float32) - np.float32(H / 2), np.arange(W, dtype=np.float32) - np.float32(W / 2),
                         indexing='ij')
    c = np.sqrt(xy * xy + yx * yx).astype(np.float32)
    return (c / c.max() + np.float32(1)).astype(np.float32)


def focal_loss_voxel(pred, target, class_weights, ignore_index=255, camera_mask=None, gamma=2.0, alpha=0.25,
                     loss_weight=100.0, want_grad=False):
    """CustomFocalLoss.forward (focal_loss.py:206-262) as loss_voxel calls it (preworld.py:146-148): per valid voxel
    (target != ignore, camera mask) the sigmoid focal loss of its C logits, weighted by class_weights[c] * radial_map[h, w],
    summed over classes, averaged over the valid voxels, times loss_weight.
    The per-element term is mmcv-full 1.6.0's sigmoid_focal_loss CUDA op (third-party dependency, not vendored in the
    reference tree; its published kernel: -[t==c] a (1-p)^g log(max(p, FLT_MIN)) - [t!=c] (1-a) p^g log(max(1-p, FLT_MIN)),
    p = sigmoid(x)); the reference's own CPU branch (py_sigmoid_focal_loss, focal_loss.py:12-57) is the same function
    and is what the golden fixture was generated with.  pred (B,C,X,Y,Z), target (B,X,Y,Z).  float64 sums."""
    z = np.asarray(pred, np.float64)
    B, C, X, Y, Z = z.shape
    t = np.asarray(target).astype(np.int64)
    valid = t != ignore_index
    if camera_mask is not None:
        valid &= np.asarray(camera_mask).astype(bool)
    cmap = focal_radial_map(X, Y).astype(np.float64)[None, :, :, None]
    w = np.asarray(class_weights, np.float64)
    p = 1.0 / (1.0 + np.exp(-z))
    tiny = np.finfo(np.float32).tiny
    onehot = (t[:, None] == np.arange(C)[None, :, None, None, None])
    term_p = (1 - p) ** gamma * np.log(np.maximum(p, tiny))
    term_n = p ** gamma * np.log(np.maximum(1 - p, tiny))
    el = np.where(onehot, -alpha * term_p, -(1 - alpha) * term_n)
    wm = w[None, :, None, None, None] * cmap[:, None]
    n = valid.sum()
    loss = loss_weight * (el * wm * valid[:, None]).sum() / n
    if not want_grad:
        return float(loss)
    gp = (1 - p) ** gamma * (1 - p - gamma * p * np.log(np.maximum(p, tiny)))
    gn = p ** gamma * (gamma * (1 - p) * np.log(np.maximum(1 - p, tiny)) - p)
    g = np.where(onehot, -alpha * gp, -(1 - alpha) * gn) * wm * valid[:, None] * (loss_weight / n)
    return float(loss), g.astype(np.float32)


def lovasz_softmax(probas, labels, ignore=None, camera_mask=None, want_grad=False):
    """mmdet3d/models/detectors/lovasz_softmax.py:157-232 (classes='present', per_image=False): flatten_probas keeps
    the voxels with label != ignore (and camera mask); for every class present among them the errors |fg - p_c| are
    sorted descending and dotted with lovasz_grad (:20-33) of the sorted foreground indicator; mean over those classes.
    probas (B,C,X,Y,Z), labels (B,X,Y,Z).  The gradient treats lovasz_grad as a constant, like the reference
    (Variable(lovasz_grad(fg_sorted)))."""
    P = np.asarray(probas, np.float32)
    B, C = P.shape[:2]
    lab = np.asarray(labels).astype(np.int64).reshape(-1)
    pr = np.moveaxis(P.reshape(B, C, -1), 1, 2).reshape(-1, C)
    valid = np.ones_like(lab, bool) if ignore is None else lab != ignore
    if camera_mask is not None:
        valid &= np.asarray(camera_mask).astype(bool).reshape(-1)
    vp, vl = pr[valid], lab[valid]
    losses, grads = [], np.zeros_like(vp)
    for c in range(C):
        fg = (vl == c).astype(np.float32)
        if fg.sum() == 0:
            continue
        err = np.abs(fg - vp[:, c])
        perm = np.argsort(-err, kind='stable')
        es, fs = err[perm], fg[perm]
        gts = fs.sum()
        inter = gts - np.cumsum(fs, dtype=np.float32)
        union = gts + np.cumsum(1 - fs, dtype=np.float32)
        jac = (np.float32(1) - inter / union).astype(np.float32)
        jac[1:] = jac[1:] - jac[:-1]
        losses.append(np.dot(es.astype(np.float64), jac.astype(np.float64)))
        grads[perm, c] = jac * np.where(fg[perm] > 0, -1.0, 1.0) * (es != 0)
    loss = float(np.mean(losses)) if losses else 0.0
    if not want_grad:
        return loss
    g = np.zeros_like(pr)
    g[valid] = grads / max(len(losses), 1)
    return loss, np.moveaxis(g.reshape(B, -1, C), 2, 1).reshape(P.shape)


# --------------------------------------------------------------------------- ray table + WRS weights
def pts2ray(coor, label_depth, label_seg, label_img, c2w, K):
    """mmdet3d/datasets/ray.py:34-55: get_rays(x+0.5, y+0.5, K, c2w, inverse_y=True) + the (n,16) row."""
    coor, c2w, K = _f32(coor), _f32(c2w), _f32(K)
    i, j = coor[:, 0] + np.float32(0.5), coor[:, 1] + np.float32(0.5)
    dirs = np.stack([(i - K[0, 2]) / K[0, 0], (j - K[1, 2]) / K[1, 1], np.ones_like(i)], -1).astype(np.float32)
    prod = dirs[:, None, :] * c2w[None, :3, :3]
    rays_d = ((prod[..., 0] + prod[..., 1]) + prod[..., 2]).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape)
    sq = rays_d * rays_d
    nrm = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2]).astype(np.float32)
    view = rays_d / nrm[:, None]
    return np.concatenate([coor, _f32(label_depth)[:, None], _f32(label_seg)[:, None], rays_o, rays_d, view,
                           _f32(label_img)], axis=1).astype(np.float32)


def wrs_weights(rays_list, ids, dynamic_class, balance_weight=None, weight_adj=0.3, weight_dyn=0.0):
    """ray.py:88-114: (balance_weight, concatenated weights)."""
    if balance_weight is None:
        classes = np.concatenate([r[:, 3] for r in rays_list])
        class_nums = np.array([(classes == c).sum() for c in range(17)], np.float32)
        with np.errstate(divide='ignore'):
            balance_weight = np.exp(np.float32(0.005) * (class_nums.max() / class_nums - np.float32(1))).astype(np.float32)
    out = []
    for r, fid in zip(rays_list, ids):
        wt = np.full(r.shape[0], 1.0 if fid == 0 else weight_adj, np.float32)
        if fid != 0:
            dyn = np.isin(r[:, 3], np.asarray(dynamic_class, np.float32))
            wt[dyn] = weight_dyn
        out.append(_f32(balance_weight)[r[:, 3].astype(np.int64)] * wt)
    return _f32(balance_weight), np.concatenate(out).astype(np.float32)


# --------------------------------------------------------------------------- synthetic rig
def synthetic_rig(n_cams=6, dx=0.0, dtype=np.float32):
    """SURVEY.md 8d analytic 6-camera rig (nuScenes-like). Returns dict of (1,N,...) arrays."""
    yaws = [55, 0, -55, -110, 180, 110][:n_cams] if n_cams > 1 else [0]
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    s2e = np.zeros((1, len(yaws), 4, 4), np.float64)
    for i, y in enumerate(yaws):
        a = math.radians(y)
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        s2e[0, i, :3, :3] = Rz @ base
        s2e[0, i, :3, 3] = [1.5 * math.cos(a) + dx, 1.5 * math.sin(a), 1.5]
        s2e[0, i, 3, 3] = 1
    K = np.array([[1266.4, 0, 816.27], [0, 1266.4, 491.5], [0, 0, 1]], np.float64)
    n = len(yaws)
    return dict(
        sensor2ego=s2e.astype(dtype),
        intrin=np.broadcast_to(K, (1, n, 3, 3)).astype(dtype).copy(),
        post_rot=np.broadcast_to(np.diag([0.88, 0.88, 1.0]), (1, n, 3, 3)).astype(dtype).copy(),
        post_tran=np.broadcast_to(np.array([0.0, -280.0, 0.0]), (1, n, 3)).astype(dtype).copy(),
        bda=np.eye(3, dtype=dtype)[None].copy(),
    )
