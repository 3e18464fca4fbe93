"""GPU drop-ins for mmdet3d/models/detectors/loss.py:20-113: CE_ssc_loss,
sem_scal_loss, geo_scal_loss -- same names, same arguments -- and `voxel_losses`, the three of them
from ONE pass over the logits (what loss_voxel, preworld_temporal_traj.py:176-199, needs), with an
autograd backward that is one more pass (pw_voxel_loss_stats / pw_voxel_loss_grad).

The scalar algebra on the 104 accumulated sums runs in two one-block kernels (pw_voxel_loss_finish,
pw_voxel_loss_coef): no host sync, 2 launches forward and 2 backward."""
import torch

from . import _lib, ops

_NS = 104
_MAXC = 32


def _args(pred, target, camera_mask, class_weights):
    if not pred.is_cuda or pred.dtype != torch.float32 or pred.dim() != 5:
        raise _lib.PreworldHipError('pred must be a float32 device tensor (B, C, X, Y, Z)')
    B, C, X, Y, Z = pred.shape
    t = target.reshape(B, X, Y, Z).to(torch.uint8).contiguous()
    cm = camera_mask.reshape(B, X, Y, Z).to(torch.uint8).contiguous() if camera_mask is not None else None
    cw = class_weights.to(device=pred.device, dtype=torch.float32).contiguous() if class_weights is not None else None
    return (B, C, X, Y, Z), t, cm, cw


def _stats(pred, t, cm, cw, dims, ignore_index, empty_idx):
    B, C, X, Y, Z = dims
    stats = torch.zeros(_NS, device=pred.device, dtype=torch.float64)
    sb, sc, sx, sy, sz = pred.stride()
    _lib.call('pw_voxel_loss_stats', ops._p(pred), ops._p(t), ops._p(cm), ops._p(cw), B, C, X, Y, Z, sb, sc, sx,
              sy, sz, int(ignore_index), int(empty_idx), ops._p(stats), ops._stream())
    return stats


class _VoxelLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, camera_mask, class_weights, ignore_index, empty_idx):
        dims, t, cm, cw = _args(pred, target, camera_mask, class_weights)
        s = _stats(pred, t, cm, cw, dims, ignore_index, empty_idx)
        out = torch.empty(3, device=pred.device, dtype=torch.float32)
        _lib.call('pw_voxel_loss_finish', ops._p(s), dims[1], ops._p(out), ops._stream())
        ctx.save_for_backward(pred, t, cm if cm is not None else torch.empty(0), cw if cw is not None else torch.empty(0), s)
        ctx.meta = (dims, ignore_index, empty_idx, cm is not None, cw is not None)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_ce, g_sem, g_geo):
        pred, t, cm, cw, s = ctx.saved_tensors
        dims, ignore_index, empty_idx, has_cm, has_cw = ctx.meta
        B, C, X, Y, Z = dims
        gout = torch.stack([g_ce, g_sem, g_geo]).to(torch.float32).contiguous()
        coef = torch.empty(2 * _MAXC + 3, device=pred.device, dtype=torch.float32)
        _lib.call('pw_voxel_loss_coef', ops._p(s), C, ops._p(gout), ops._p(coef), ops._stream())
        grad = torch.empty_strided(pred.shape, pred.stride(), device=pred.device, dtype=pred.dtype)
        sb, sc, sx, sy, sz = pred.stride()
        _lib.call('pw_voxel_loss_grad', ops._p(pred), ops._p(t), ops._p(cm) if has_cm else None,
                  ops._p(cw) if has_cw else None, ops._p(coef), B, C, X, Y, Z, sb, sc, sx, sy, sz,
                  int(ignore_index), int(empty_idx), ops._p(grad), ops._stream())
        return grad, None, None, None, None, None


def voxel_losses(pred, target, class_weights=None, ignore_index=255, empty_idx=17, camera_mask=None):
    """(CE_ssc_loss, sem_scal_loss, geo_scal_loss) of loss.py from one pass over `pred` (B,C,X,Y,Z)."""
    return _VoxelLosses.apply(pred, target, camera_mask, class_weights, ignore_index, empty_idx)


def CE_ssc_loss(pred, target, class_weights, ignore_index):
    """loss.py:20-29."""
    return voxel_losses(pred, target, class_weights, ignore_index, 0)[0]


def sem_scal_loss(pred, ssc_target, ignore_index, camera_mask=None):
    """loss.py:32-80."""
    return voxel_losses(pred, ssc_target, None, ignore_index, 0, camera_mask)[1]


def geo_scal_loss(pred, ssc_target, ignore_index, non_empty_idx=0, camera_mask=None):
    """loss.py:83-113 (ignore_index is unused by the reference's geo term as well)."""
    return voxel_losses(pred, ssc_target, None, ignore_index, non_empty_idx, camera_mask)[2]
