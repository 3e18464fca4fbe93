"""Composition of the hot path for bench.py / smoke(): synthetic inputs in HBM -> LSS voxel
pooling -> (later stages are added as their kernels land).  Mirrors the call order of
PreWorld4DTraj.simple_test (mmdet3d/models/detectors/preworld_temporal_traj.py:212-370)."""
import numpy as np
import torch

from . import modules, ops, synth as S


def to_dev(a, dev='cuda:0'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def smoke_check():
    """One small hot-path invocation on cuda:0, checked against the CPU oracle.
    (The oracle is imported here only as the checker -- see oracle/pw_oracle.c header.)"""
    from oracle import oracle as O
    dev = 'cuda:0'
    gc = S.GRID_CONFIG_C1
    rig = S.synthetic_rig(1)
    depth, feat = S.lift_inputs(3, N=1)
    vt = modules.LSSViewTransformer(grid_config=gc, input_size=S.INPUT_SIZE, downsample=S.DOWNSAMPLE,
                                    in_channels=8, out_channels=32, collapse_z=False).to(dev)
    inp = [torch.empty(1, 1, 8, 32, 88, device=dev)] + \
        [to_dev(rig[k]) for k in ('sensor2ego',)] + [None] + \
        [to_dev(rig[k]) for k in ('intrin', 'post_rot', 'post_tran', 'bda')]
    with torch.no_grad():
        bev, _ = vt.view_transform(inp, to_dev(depth).view(1, 88, 32, 88), to_dev(feat).view(1, 32, 32, 88))
    want = O.lss_view_transform(depth, feat, rig['sensor2ego'], rig['intrin'], rig['post_rot'],
                                rig['post_tran'], rig['bda'], gc, S.INPUT_SIZE, S.DOWNSAMPLE)
    got = bev.cpu().numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want), 'LSS pooling differs from the oracle'
    return True
