"""Registry adapter and module construction from the reference's config dicts (CPU, no kernels)."""
import pytest
import torch

from preworld_amd import modules as M
from preworld_amd import registry as R
from preworld_amd import synth as S


class FakeRegistry:
    """mmcv 1.6.0 Registry.register_module(name=None, force=False, module=None) semantics."""

    def __init__(self):
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        if not force and name in self.module_dict:
            raise KeyError(name + ' is already registered')
        self.module_dict[name] = module
        return module

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop('type')](**cfg)


def test_force_reregistration_and_build_from_reference_cfg():
    regs = {k: FakeRegistry() for k in ('mmdet3d.NECKS', 'mmdet.BACKBONES', 'mmdet.NECKS', 'mmdet.HEADS')}
    regs['mmdet.BACKBONES'].register_module(name='CustomResNet3D', module=object)    # the reference's
    with pytest.raises(KeyError):
        regs['mmdet.BACKBONES'].register_module(name='CustomResNet3D', module=object)
    done = R.register(regs)
    assert ('mmdet.BACKBONES', 'CustomResNet3D') in done
    assert regs['mmdet.BACKBONES'].module_dict['CustomResNet3D'] is M.CustomResNet3D
    # config dicts copied from configs/preworld/nuscenes/bevstereo-occ.py:90-108 and
    # preworld-7frame-finetune.py (resolved values)
    bb = regs['mmdet.BACKBONES'].build(dict(type='CustomResNet3D', numC_input=64, num_layer=[1, 2, 4],
                                            with_cp=False, num_channels=[32, 64, 128], stride=[1, 2, 2],
                                            backbone_output_ids=[0, 1, 2]))
    nk = regs['mmdet.NECKS'].build(dict(type='LSSFPN3D', in_channels=224, out_channels=32))
    hd = regs['mmdet.HEADS'].build(dict(type='OccHead', with_cp=False, use_deblock=False,
                                        norm_cfg=dict(type='SyncBN', requires_grad=True),
                                        soft_weights=True, final_occ_size=[200, 200, 16], empty_idx=17,
                                        num_level=1, in_channels=[32], out_channel=18,
                                        point_cloud_range=[-40, -40, -1, 40, 40, 5.4]))
    nh = regs['mmdet.HEADS'].build(dict(type='NerfHead', point_cloud_range=[-40, -40, -1, 40, 40, 5.4],
                                        voxel_size=0.4, scene_center=[0, 0, 2.2], radius=39,
                                        use_depth_sup=True, weight_depth=1.0, weight_semantic=1.0,
                                        weight_color=1.0))
    # parameter counts of the reference modules
    assert sum(p.numel() for p in bb.parameters()) == 4122688
    assert sum(p.numel() for p in nk.parameters()) == 7232
    assert sum(p.numel() for p in hd.parameters()) == 14296
    assert set(nh.state_dict()) == {'scene_center', 'scene_radius', 'xyz_min', 'xyz_max', 'act_shift'}
    assert ('mmdet.LOSSES', 'CustomFocalLoss') not in done           # no LOSSES registry was passed: skipped, not an error
    regs['mmdet.LOSSES'] = FakeRegistry()
    assert ('mmdet.LOSSES', 'CustomFocalLoss') in R.register(regs)
    fl = regs['mmdet.LOSSES'].build(dict(type='CustomFocalLoss'))      # preworld.py:117
    assert (fl.gamma, fl.alpha, fl.loss_weight) == (2.0, 0.25, 100.0)


def test_state_dict_keys_match_reference_names():
    """SURVEY 8b key list: a state dict with the reference's names loads without surprises."""
    net = M.PreWorld4DTraj(
        img_view_transformer=dict(grid_config=S.GRID_CONFIG_FULL, input_size=S.INPUT_SIZE,
                                  in_channels=512, out_channels=32, collapse_z=False, downsample=16),
        img_bev_encoder_backbone=dict(numC_input=64, num_layer=[1, 2, 4], num_channels=[32, 64, 128],
                                      stride=[1, 2, 2], backbone_output_ids=[0, 1, 2]),
        img_bev_encoder_neck=dict(in_channels=224, out_channels=32),
        pre_process=dict(numC_input=32, num_layer=[1], num_channels=[32], stride=[1],
                         backbone_output_ids=[0]))
    keys = set(net.state_dict())
    for k in ['pre_process_net.layers.0.0.conv1.conv.weight', 'pre_process_net.layers.0.0.downsample.bn.running_var',
              'img_bev_encoder_backbone.layers.2.3.conv2.bn.weight', 'img_bev_encoder_neck.conv.conv.weight',
              'img_bev_encoder_neck.conv.bn.running_mean', 'final_conv.conv.weight', 'final_conv.conv.bias',
              'occupancy_head.occ_convs.0.0.weight', 'occupancy_head.occ_convs.0.1.running_mean',
              'occupancy_head.occ_pred_conv.0.weight', 'occupancy_head.occ_pred_conv.1.bias',
              'occupancy_head.occ_pred_conv.3.weight', 'occupancy_head.voxel_soft_weights.3.weight',
              'density_mlp.0.weight', 'density_mlp.2.bias', 'semantic_mlp.2.weight', 'color_mlp.0.bias',
              'plan_head.0.weight', 'plan_head.4.bias', 'fusion_head.0.weight', 'fusion_head.2.bias']:
        assert k in keys, k
    sd = S.synth_state_dict(0)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected
    assert all('depth_net' in k or 'num_batches_tracked' in k for k in missing)


def test_modules_refuse_cpu_tensors():
    """no silent fallback: the HIP path raises on CPU tensors"""
    m = M.CustomResNet3D(numC_input=32, num_layer=[1], num_channels=[32], stride=[1]).eval()
    with pytest.raises(Exception):
        m(torch.zeros(1, 32, 4, 8, 8))


def test_prepare_inputs_pose_algebra():
    """A21: bevdet_occ.py:88-139 -- sweep poses expressed in the key ego frame (fp64), frame split
    orders (images camera-major/frame-minor, poses frame-major)."""
    import numpy as np
    net = M.PreWorld4DTraj(
        img_view_transformer=dict(grid_config=S.GRID_CONFIG_FULL, input_size=S.INPUT_SIZE,
                                  in_channels=512, out_channels=32, collapse_z=False, downsample=16),
        img_bev_encoder_backbone=dict(numC_input=64, num_layer=[1], num_channels=[32], stride=[1]),
        img_bev_encoder_neck=dict(in_channels=224, out_channels=32))
    B, N, T = 1, 6, 3
    rs = np.random.RandomState(0)

    def rand_pose(n):
        P = np.tile(np.eye(4), (n, 1, 1))
        for k in range(n):
            q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
            P[k, :3, :3] = q * np.sign(np.linalg.det(q))
            P[k, :3, 3] = rs.standard_normal(3) * 10
        return P
    s2e = rand_pose(T * N)[None]
    e2g = rand_pose(T * N)[None]
    imgs = torch.arange(B * N * T).float().view(B, N * T, 1, 1, 1).expand(B, N * T, 3, 2, 2)
    inputs = (imgs, torch.from_numpy(s2e).float(), torch.from_numpy(e2g).float(),
              torch.rand(B, T * N, 3, 3), torch.rand(B, T * N, 3, 3), torch.rand(B, T * N, 3),
              torch.eye(3)[None])
    out = net.prepare_inputs(inputs, stereo=True, num_frame=T, temporal_frame=2, extra_ref_frames=1)
    im, s2k, e2gs, K, pr, pt, bda, c2a = out
    assert len(im) == T and im[1].shape == (B, N, 3, 2, 2)
    # image (camera n, frame t) sits at stacked index n*T + t
    assert float(im[2][0, 4, 0, 0, 0]) == 4 * T + 2
    ref = np.linalg.inv(e2g[0, 0]) @ e2g[0].reshape(T, N, 4, 4) @ s2e[0].reshape(T, N, 4, 4)
    np.testing.assert_allclose(torch.stack(s2k, 1)[0].numpy(), ref, rtol=1e-4, atol=1e-4)
    a, c = e2g[0].reshape(T, N, 4, 4), s2e[0].reshape(T, N, 4, 4)
    ref_c2a = np.linalg.inv(a[1] @ c[1]) @ a[0] @ c[0]
    np.testing.assert_allclose(c2a[0][0].numpy(), ref_c2a, rtol=1e-4, atol=1e-4)
    assert c2a[2] is None and len(c2a) == T
