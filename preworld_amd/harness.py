"""Row H of SURVEY.md 8a: what replaces tools/test_temporal.py + mmdet3d/apis/test.py for the hot
path -- build the detector from a restated config dict, load a state dict by the reference's key
names, feed lifted inputs, collect `semantic_occ_{k}s`, stack states 0/2/4/6
(mmdet3d/apis/test.py:218-223) and score them with Metric_mIoU_Temporal
(mmdet3d/datasets/occ_metrics.py:413-594).  GPU only: every op goes through libpreworld_hip.so."""
import numpy as np
import torch

from . import builder, metrics, synth


def model_cfg(grid_config=None, with_prev=True, if_post_finetune=True, detector='PreWorld4DTraj'):
    """The `model = dict(...)` section of configs/preworld/**.py restricted to the hot path
    (bevstereo-occ.py:62-131 + preworld-7frame-finetune[-traj].py), as a plain dict; the image side
    (img_backbone / img_neck) is left out: benches and parity tests start from the lifted inputs."""
    gc = grid_config or synth.GRID_CONFIG_FULL
    sx = int(round((gc['x'][1] - gc['x'][0]) / gc['x'][2]))
    sy = int(round((gc['y'][1] - gc['y'][0]) / gc['y'][2]))
    sz = int(round((gc['z'][1] - gc['z'][0]) / gc['z'][2]))
    return dict(
        type=detector,
        img_view_transformer=dict(type='LSSViewTransformerBEVStereo', grid_config=gc,
                                  input_size=synth.INPUT_SIZE, in_channels=512, out_channels=32, sid=False,
                                  collapse_z=False, loss_depth_weight=0.05,
                                  depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True, bias=5.0),
                                  downsample=16),
        img_bev_encoder_backbone=dict(type='CustomResNet3D', numC_input=64, num_layer=[1, 2, 4], with_cp=False,
                                      num_channels=[32, 64, 128], stride=[1, 2, 2], backbone_output_ids=[0, 1, 2]),
        img_bev_encoder_neck=dict(type='LSSFPN3D', in_channels=224, out_channels=32),
        pre_process=dict(type='CustomResNet3D', numC_input=32, with_cp=False, num_layer=[1], num_channels=[32],
                         stride=[1], backbone_output_ids=[0]),
        occupancy_head=dict(type='OccHead', with_cp=False, use_deblock=False,
                            norm_cfg=dict(type='SyncBN', requires_grad=True), soft_weights=True,
                            final_occ_size=[sx, sy, sz], empty_idx=17, num_level=1, in_channels=[32],
                            out_channel=18, point_cloud_range=[gc['x'][0], gc['y'][0], gc['z'][0],
                                                               gc['x'][1], gc['y'][1], gc['z'][1]]),
        if_post_finetune=if_post_finetune, with_prev=with_prev)


def build_model(cfg, state_dict, device='cuda:0'):
    """cfg: model_cfg(...) or the reference's own `model` dict (type PreWorld / PreWorld4DTraj / BEVStereo4DOCC);
    state_dict: numpy or torch tensors under the reference's key names.  Hot-path keys must all be present; image-side
    keys (img_backbone / img_neck / depth_net) may be absent when only lifted inputs are fed."""
    net = builder.build(cfg, 'PreWorld4DTraj')
    sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in state_dict.items()}
    missing, _ = net.load_state_dict(sd, strict=False)
    # (nerf_head.*: five registered buffers the constructor recomputes from the config, nerf_head.py:134-144)
    image_side = ('depth_net', 'img_backbone', 'img_neck', 'num_batches_tracked', 'nerf_head.')
    bad = [k for k in missing if not any(t in k for t in image_side)]
    if bad:
        raise KeyError('state dict lacks hot-path keys: %s' % bad[:8])
    return net.to(device).eval()


def lifted_frames(seed, grid_cams, device='cuda:0', n_frames=2):
    """Synthetic (depth, context, camera) inputs of SURVEY.md 8d for `n_frames` frames (key first)."""
    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    frames = []
    for f in range(n_frames):
        rig = synth.synthetic_rig(grid_cams, dx=-2.5 * f)
        depth, feat = synth.lift_inputs(seed * 16 + f, N=grid_cams)
        frames.append(dict(depth=T(depth).view(grid_cams, 88, 32, 88), tran_feat=T(feat).view(grid_cams, 32, 32, 88),
                           sensor2keyego=T(rig['sensor2ego']), intrin=T(rig['intrin']),
                           post_rot=T(rig['post_rot']), post_tran=T(rig['post_tran']), bda=T(rig['bda'])))
    return frames


def stack_states(result, horizons=(0, 2, 4, 6)):
    """apis/test.py:218-223: [np.stack([semantic_occ_0s, _2s, _4s, _6s])] of sample 0."""
    return np.stack([result['semantic_occ_%ds' % h][0].cpu().numpy() for h in horizons], axis=0)


@torch.no_grad()
def evaluate(net, samples, device='cuda:0', use_image_mask=True):
    """samples: iterable of dict(frames, ego, gt {idx: (X,Y,Z) uint8}, mask_camera (X,Y,Z) bool).
    Scores the stacked states {0,2,4,6} like tools/test_temporal.py -> dataset.evaluate
    (nuscenes_dataset_occ_trajectory.py:478-526).  Returns (Metric_mIoU_Temporal.report() dict incl. the 0 s horizon and
    'avg_future', list of stacked predictions); the reference's own return values are metric.count_miou() /
    count_iou() of the same object (third return value)."""
    metric = metrics.Metric_mIoU_Temporal(num_classes=18, use_image_mask=use_image_mask, device=device)
    stacks = []
    for s in samples:
        res = net.simple_test_from_lift(s['frames'], s['ego'], n_steps=6)
        st = stack_states(res)
        stacks.append(st)
        mc = s.get('mask_camera')
        metric.add_batch(st, s['gt'], None, {h: mc for h in s['gt']} if mc is not None else None)
    return metric.report(), stacks, metric


@torch.no_grad()
def simple_test_sharded(net, frames, ego, n_steps=6, group=None, gather_on_host=False, timings=None):
    """The latency mode of DESIGN.md section 7 wired to the real modules (one process per GPU, torch.distributed
    initialised): frame f is lifted + pre-processed on rank f % W and all ranks receive every frame's (B,Z,Y,X,32)
    feature through ONE all_gather (parallel.lift_frames_sharded), every rank runs the encoder, state k is forecast +
    decoded on rank k % W, one all_gather of the uint8 grids assembles all states on every rank.
    Returns {'semantic_occ_%ds': [(X,Y,Z) uint8]} like simple_test_from_lift.  `with_prev=False` drops the adjacent
    frames: their channel slice is zeros (bevdet_occ.py:243-258), exactly as in extract_bev_feat_cl.
    gather_on_host=True moves the 0.64 MB grids through host memory (for process groups that cannot all_gather device
    tensors: gloo in the tests; RCCL takes device tensors).
    timings: a dict that receives the milliseconds of the LAST pass's phases on this rank -- lift (LSS + pre_process of this rank's
    frames), gather_frames (the 81.92 MB per frame all_gather), encoder (cat + bev_encoder + final_conv), decode (this rank's
    share of the recursion + OccHead), gather_states (the uint8 all_gather) -- from HIP events on the current stream (synchronises) --
    and the payload bytes of the two exchanges on this rank (frames_bytes_received / _sent, states_bytes_received)."""
    from . import ops, parallel
    from .modules import precision
    vt = net.img_view_transformer
    _, _, size = vt._grid()
    f0 = frames[0]
    B, C = f0['sensor2keyego'].shape[0], vt.out_channels
    n = net.num_adj + 1
    use = frames[:n] if net.with_prev else frames[:1]
    # The exchanged features are fp32 values: an h2 buffer is only meaningful together with the exponent of its range slot,
    # which every rank calibrates for itself (ops.RangeCtx).  On the split-fp16 path a frame is lifted in h2, expanded
    # (hi + lo) * 2^e -- exact -- for the all_gather, and the concatenated buffer is split again under ONE exponent derived
    # from its own maximum, which is the same number on every rank.
    h2 = precision() == 'h2' and C % 32 == 0
    from .modules import as_f32

    def lift(fr):
        return as_f32(net.lift_frame_cl(out_h2=h2, **fr))

    def decode(f):
        occ = net.occupancy_head.decode_cl(f, transposed=True)
        occ = occ.permute(0, 3, 2, 1)[0].contiguous()                          # batch element 0, (X,Y,Z) (:306)
        return occ.cpu() if gather_on_host else occ

    events = []

    def mark(name):
        if timings is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append((name, e))

    def one_pass():
        del events[:]
        mark('start')
        lifted = parallel.lift_frames_sharded(use, lift, (B, size[2], size[1], size[0], C), torch.float32, f0['depth'].device,
                                              group, via_host=gather_on_host, mark=mark, stats=timings)
        x = torch.cat(lifted[1:][::-1] + lifted[:1], dim=-1)                   # [adjacent ..., key] (bevdet_occ.py:266)
        if len(lifted) < n:
            x = torch.cat([x.new_zeros(x.shape[:-1] + ((n - len(lifted)) * C,)), x], dim=-1)
        # final_conv -> forecast -> OccHead keep h2 storage like simple_test_from_lift (post-finetune decode)
        v0 = net.final_conv.forward_cl(net.bev_encoder_cl(ops.f32_to_h2(x) if h2 else x, out_h2=h2), out_h2=h2)
        mark('encoder')
        return parallel.decode_states_sharded(v0, lambda v, k: net.forecast_cl(v, ego, k, out_h2=h2)[0], decode,
                                              n_steps + 1, group, mark=mark, stats=timings,
                                              grid_like=((int(size[0]), int(size[1]), int(size[2])), torch.uint8,
                                                         'cpu' if gather_on_host else f0['depth'].device))

    if h2:
        # ranks own different tensors, so they must agree on whether another calibration pass runs (the passes contain
        # collectives): one tiny MIN all-reduce per pass
        ctx = net.__dict__.get('_range_ctx_sharded')
        if ctx is None or ctx.device != f0['depth'].device:
            ctx = net.__dict__['_range_ctx_sharded'] = ops.RangeCtx(f0['depth'].device)
        grids = ops.ranged(one_pass, ctx, agree=lambda ok: parallel.all_agree(ok, f0['depth'].device, group, gather_on_host))
    else:
        grids = one_pass()
    if timings is not None:
        torch.cuda.synchronize()
        for (_, a), (name, b) in zip(events[:-1], events[1:]):
            timings[name] = a.elapsed_time(b)
    return {'semantic_occ_%ds' % k: [g] for k, g in enumerate(grids)}
