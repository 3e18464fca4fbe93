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


ALL_REGS = ('mmdet3d.NECKS', 'mmdet.BACKBONES', 'mmdet.NECKS', 'mmdet.HEADS', 'mmdet.DETECTORS', 'mmdet.LOSSES')


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


def test_training_registration_keeps_inference_only_classes_on_the_reference():
    """classes without a backward must not reach a training run.  training=True swaps only what can train (forward
    under autograd + HIP backward, pinned by gradient fixtures: tests/test_gpu_train.py) -- the REGISTRY_OF flag is what gates it."""
    regs = {k: FakeRegistry() for k in ALL_REGS}
    done = R.register(regs, training=True)
    assert all(can_train for _, can_train in R.REGISTRY_OF.values())       # round 2: every drop-in has a training path
    assert {n for _, n in done} == set(R.REGISTRY_OF)
    regs = {k: FakeRegistry() for k in ALL_REGS}
    done = R.register(regs)
    assert {n for _, n in done} == set(R.REGISTRY_OF)


def _hot_types(cfg, found):
    if isinstance(cfg, dict):
        if 'type' in cfg:
            found.append(cfg['type'])
        for v in cfg.values():
            _hot_types(v, found)
    return found


def test_every_preworld_config_builds_through_the_registry():
    """The resolved `model` dict of each of the six configs/preworld/**.py files (tests/golden/preworld_configs.json,
    produced by tools/gen_golden.py executing the reference configs) builds through the stand-in registry after
    R.register(), and every component on the camera -> occupancy path is a preworld_amd class -- in particular the
    view transformer the configs name (LSSViewTransformerBEVStereo) and the detectors PreWorld / PreWorld4DTraj."""
    import json
    import os
    from preworld_amd import detectors as D, image_encoder as IE, losses as L
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'preworld_configs.json')))
    assert len(cfgs) == 6
    regs = {k: FakeRegistry() for k in ALL_REGS}
    R.register(regs)
    want_cls = {'BEVStereo4DOCC': D.BEVStereo4DOCC, 'PreWorld': D.PreWorld, 'PreWorld4DTraj': D.PreWorld4DTraj}
    for name, model in cfgs.items():
        types_used = set(_hot_types(model, []))
        # every type on the path resolves to a class of this package (GELU / LN / SyncBN / CrossEntropyLoss are layer
        # or loss names inside act_cfg / norm_cfg / loss_occ, not registry builds on this path)
        from preworld_amd import builder
        path_types = types_used - {'GELU', 'LN', 'SyncBN', 'CrossEntropyLoss'}
        assert path_types <= set(builder.table()), (name, path_types - set(builder.table()))
        model = dict(model)
        model['img_backbone'] = dict(model['img_backbone'], depths=[1, 1, 1, 1])     # shallow Swin: the test builds, it does not run
        det = regs['mmdet.DETECTORS'].build(model)
        assert type(det) is want_cls[model['type']], name
        assert type(det.img_view_transformer) is M.LSSViewTransformerBEVStereo
        assert type(det.img_view_transformer.depth_net) is IE.DepthNet
        assert type(det.img_bev_encoder_backbone) is M.CustomResNet3D and type(det.pre_process_net) is M.CustomResNet3D
        assert type(det.img_bev_encoder_neck) is M.LSSFPN3D
        assert type(det.img_backbone) is IE.SwinTransformer and type(det.img_neck) is IE.FPN_LSS
        assert det.num_frame == 3 and det.temporal_frame == 2 and det.num_adj == 1
        if model['type'] != 'BEVStereo4DOCC':
            assert type(det.occupancy_head) is M.OccHead and type(det.nerf_head) is M.NerfHead
            assert type(det.focal_loss) is L.CustomFocalLoss
            assert det.if_post_finetune == model.get('if_post_finetune', False)
            keys = set(det.state_dict())
            for k in ('final_conv.conv.bias', 'density_mlp.2.weight', 'semantic_mlp.0.bias', 'color_mlp.2.bias',
                      'occupancy_head.occ_pred_conv.3.weight', 'nerf_head.act_shift',
                      'img_view_transformer.depth_net.reduce_conv.0.weight',
                      'img_view_transformer.depth_net.depth_conv.4.weight', 'img_neck.conv.0.weight',
                      'img_backbone.patch_embed.projection.weight'):
                assert k in keys, (name, k)
            has_traj = model['type'] == 'PreWorld4DTraj'
            assert ('fusion_head.0.weight' in keys) == has_traj and ('plan_head.4.bias' in keys) == has_traj
            assert ('downscale.downscale3.weight' in keys) == has_traj
        else:
            assert 'predicter.2.weight' in det.state_dict()
        with pytest.raises(RuntimeError, match='training mode'):          # every detector has a training step (tests/test_gpu_train.py)
            det.eval().forward_train()


def test_bevdepth_host_methods_match_reference(golden):
    """LSSViewTransformerBEVStereo.get_mlp_input / get_downsampled_gt_depth / get_depth_loss / cv_frustum
    (view_transformer.py:713-789, 807-813) against outputs of the imported reference class."""
    g = golden('bevdepth_small.npz')
    vt = M.LSSViewTransformerBEVStereo(grid_config=S.GRID_CONFIG_FULL, input_size=(64, 96), in_channels=16, out_channels=8,
                                       sid=False, collapse_z=False, loss_depth_weight=0.05, downsample=16,
                                       depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=8, stereo=True, bias=5.0))
    T = torch.from_numpy
    mlp = vt.get_mlp_input(T(g['sensor2ego']), T(g['ego2global']), T(g['intrin']), T(g['post_rot']), T(g['post_tran']), T(g['bda']))
    assert torch.equal(mlp, T(g['mlp_input']))
    assert torch.equal(vt.get_downsampled_gt_depth(T(g['depth_gt'])), T(g['onehot']))
    loss = vt.get_depth_loss(T(g['depth_gt']), T(g['depth_pred']))
    assert abs(float(loss) - float(g['depth_loss'])) <= 1e-6 * abs(float(g['depth_loss']))
    assert torch.equal(vt.cv_frustum, T(g['cv_frustum'])) and torch.equal(vt.frustum, T(g['frustum']))
    assert vt.D == 88 and vt.loss_depth_weight == 0.05


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
        img_bev_encoder_neck=dict(in_channels=224, out_channels=32), num_adj=1)
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


def test_detectors_expose_the_runner_api():
    """tools/train.py:244 calls model.init_weights(); mmcv's runner calls model.train_step(data, optimizer) and reads
    outputs['loss'] / ['log_vars'] / ['num_samples'] (mmdet 2.24.0 BaseDetector.train_step / _parse_losses): the drop-in detectors
    provide the same API.  The loss dict is stubbed here -- the kernels need a GPU; tests/test_gpu_train.py runs
    the real forward_train."""
    import torch
    from preworld_amd import harness
    from preworld_amd import synth as S
    for det in ('PreWorld4DTraj', 'PreWorld', 'BEVStereo4DOCC'):
        cfg = harness.model_cfg(S.GRID_CONFIG_C1, detector=det)
        if det == 'BEVStereo4DOCC':
            cfg.pop('occupancy_head'); cfg.pop('if_post_finetune')
        from preworld_amd import builder
        net = builder.build(cfg)
        net.init_weights()
        w = torch.nn.Parameter(torch.tensor(2.0))
        net.forward_train = lambda **kw: {'loss_a': w * 3.0, 'loss_list': [w * 1.0, w * 0.5], 'acc': w.detach() * 0 + 0.25}
        out = net.train_step(dict(img_metas=[{}, {}], img_inputs=None), None)
        assert set(out) == {'loss', 'log_vars', 'num_samples'} and out['num_samples'] == 2
        assert abs(float(out['loss']) - 9.0) < 1e-6                        # 'acc' has no 'loss' in its key: logged, not summed
        assert out['log_vars'] == {'loss_a': 6.0, 'loss_list': 3.0, 'acc': 0.25, 'loss': 9.0}
        out['loss'].backward()
        assert abs(float(w.grad) - 4.5) < 1e-6
        assert set(net.val_step(dict(img_metas=[{}]), None)) == {'loss', 'log_vars', 'num_samples'}
