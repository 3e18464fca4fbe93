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
