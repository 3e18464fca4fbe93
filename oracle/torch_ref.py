"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product path (preworld_amd/).

PyTorch-CPU composition of the camera -> occupancy path: what the reference's own nn.Modules compute when they run on
host cores (Conv3d / BatchNorm3d / Upsample / Linear / Softplus are plain torch modules in the reference:
mmdet3d/models/backbones/resnet.py:88-184, necks/lss_fpn.py:103-148, heads/occupancy_head.py:124-177,
detectors/preworld_temporal_traj.py:71-150,303-368).  The three native ops have no CPU implementation in the reference
(SURVEY.md section 0), so the voxel pooling comes from oracle/pw_oracle.c.  Used (a) as the second CPU baseline SURVEY 8d
asks for (`torch.set_num_threads(n)`, oneDNN convolutions) beside the OpenMP port and (b) as an independent cross-check of
the C oracle (tests/test_oracle_golden.py).  State dict: numpy arrays under the reference's key names."""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def conv_module(x, sd, prefix, stride=1, pad=1, relu=False, residual=None):
    """mmcv ConvModule (conv -> norm -> act) with eval-mode BatchNorm3d."""
    b = sd.get(prefix + '.conv.bias')
    y = F.conv3d(x, _t(sd[prefix + '.conv.weight']), None if b is None else _t(b), stride=stride, padding=pad)
    if (prefix + '.bn.weight') in sd:
        y = F.batch_norm(y, _t(sd[prefix + '.bn.running_mean']), _t(sd[prefix + '.bn.running_var']),
                         _t(sd[prefix + '.bn.weight']), _t(sd[prefix + '.bn.bias']), False, 0.0, 1e-5)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def basic_block3d(x, sd, prefix, stride=1):
    identity = conv_module(x, sd, prefix + '.downsample', stride=stride) if (prefix + '.downsample.conv.weight') in sd else x
    y = conv_module(x, sd, prefix + '.conv1', stride=stride, relu=True)
    return conv_module(y, sd, prefix + '.conv2', residual=identity, relu=True)


def custom_resnet3d(x, sd, prefix, num_layer, stride):
    feats = []
    for lid, nl in enumerate(num_layer):
        for b in range(nl):
            x = basic_block3d(x, sd, '%s.layers.%d.%d' % (prefix, lid, b), stride=stride[lid] if b == 0 else 1)
        feats.append(x)
    return feats


def lss_fpn3d(feats, sd, prefix='img_bev_encoder_neck'):
    x8, x16, x32 = feats
    x = torch.cat([x8, F.interpolate(x16, scale_factor=2, mode='trilinear', align_corners=True),
                   F.interpolate(x32, scale_factor=4, mode='trilinear', align_corners=True)], dim=1)
    return conv_module(x, sd, prefix + '.conv', pad=0, relu=True)


def occ_logits(v_xyzc, sd, prefix='occupancy_head'):
    """v (1,X,Y,Z,C) -> logits (X,Y,Z,18) (occupancy_head.py:124-177, soft-weight branch inert)."""
    x = v_xyzc[0].permute(3, 0, 1, 2)[None]
    x = F.conv3d(x, _t(sd[prefix + '.occ_convs.0.0.weight']), None, padding=1)
    p = prefix + '.occ_convs.0.1'
    x = F.relu(F.batch_norm(x, _t(sd[p + '.running_mean']), _t(sd[p + '.running_var']), _t(sd[p + '.weight']),
                            _t(sd[p + '.bias']), False, 0.0, 1e-5))
    x = F.conv3d(x, _t(sd[prefix + '.occ_pred_conv.0.weight']))
    p = prefix + '.occ_pred_conv.1'
    x = F.relu(F.batch_norm(x, _t(sd[p + '.running_mean']), _t(sd[p + '.running_var']), _t(sd[p + '.weight']),
                            _t(sd[p + '.bias']), False, 0.0, 1e-5))
    return F.conv3d(x, _t(sd[prefix + '.occ_pred_conv.3.weight']))[0].permute(1, 2, 3, 0)


def _mlp(x, sd, prefix, idx, acts):
    for i, act in zip(idx, acts):
        x = F.linear(x, _t(sd['%s.%d.weight' % (prefix, i)]), _t(sd['%s.%d.bias' % (prefix, i)]))
        if act == 'relu':
            x = F.relu(x)
        elif act == 'softplus':
            x = F.softplus(x)
    return x


@torch.no_grad()
def c3_sample(bevs, ego, sd, n_steps=6, with_prev=True, want_stages=False):
    """bevs: [key, adjacent] pooled features (1,32,Z,Y,X) numpy (from oracle.lss_view_transform).  Returns the list of
    uint8 (X,Y,Z) states [, dict of stage tensors]."""
    pre = [custom_resnet3d(_t(b), sd, 'pre_process_net', [1], [1])[0] for b in (bevs if with_prev else bevs[:1])]
    adj = pre[1] if with_prev else torch.zeros_like(pre[0])
    x = torch.cat([adj, pre[0]], dim=1)
    feats = custom_resnet3d(x, sd, 'img_bev_encoder_backbone', [1, 2, 4], [1, 2, 2])
    neck = lss_fpn3d(feats, sd)
    fc = F.relu(F.conv3d(neck, _t(sd['final_conv.conv.weight']), _t(sd['final_conv.conv.bias']), padding=1))
    v = fc.permute(0, 4, 3, 2, 1)                                             # (1,X,Y,Z,C) view
    states, logits = [], []
    lg = occ_logits(v, sd)
    logits.append(lg)
    states.append(lg.argmax(-1).to(torch.uint8).numpy())
    if n_steps:
        e = _mlp(_t(ego).reshape(1, -1), sd, 'plan_head', (0, 2, 4), ('relu', 'relu', None))
        for _ in range(n_steps):
            h = torch.cat([v, e.view(1, 1, 1, 1, -1).expand(*v.shape[:-1], -1)], dim=-1)
            v = v + _mlp(h, sd, 'fusion_head', (0, 2), ('softplus', None))
            lg = occ_logits(v, sd)
            logits.append(lg)
            states.append(lg.argmax(-1).to(torch.uint8).numpy())
    if want_stages:
        return states, dict(pre=pre, enc=feats, neck=neck, final_conv=fc, logits=logits)
    return states


def lifted_bevs(seed, n_cams, grid_config, n_frames=2):
    """pooled (1,32,Z,Y,X) features of the synthetic frames of SURVEY 8d (key first) through the C oracle's pooling."""
    from preworld_amd import synth as S
    bevs = []
    for f in range(n_frames):
        depth, feat = S.lift_inputs(seed * 16 + f, N=n_cams)
        r = S.synthetic_rig(n_cams, dx=-2.5 * f)
        bevs.append(O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'], r['post_tran'],
                                         r['bda'], grid_config, S.INPUT_SIZE, S.DOWNSAMPLE))
    return bevs
