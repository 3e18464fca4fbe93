"""Composition of the hot path for bench.py / smoke(): synthetic inputs in HBM -> LSS voxel
pooling -> voxel encoder -> forecast decode -> occupancy heads.  Mirrors the call order of
PreWorld4DTraj.simple_test (mmdet3d/models/detectors/preworld_temporal_traj.py:212-370)."""
import numpy as np
import torch


def to_dev(a, dev='cuda:0'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
