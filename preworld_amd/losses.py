"""GPU drop-ins for the voxel-grid training losses with the reference's names and arguments:
CE_ssc_loss, sem_scal_loss, geo_scal_loss (mmdet3d/models/detectors/loss.py:20-113) -- and `voxel_losses`, the three
of them from ONE pass over the logits (what loss_voxel, preworld_temporal_traj.py:176-199, needs), with an autograd
backward that is one more pass (pw_voxel_loss_stats / pw_voxel_loss_grad) -- plus the two the finetune configs add
(preworld.py:146-155): CustomFocalLoss (loss_utils/focal_loss.py:163-262) and lovasz_softmax
(detectors/lovasz_softmax.py:157-232), see the end of this file and csrc/pw_loss2.hip.

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


# ------------------------------------------------------------------------------- focal + Lovasz
class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, camera_mask, class_weights, ignore_index, gamma, alpha, loss_weight):
        dims, t, cm, cw = _args(pred, target, camera_mask, class_weights)
        B, C, X, Y, Z = dims
        stats = torch.zeros(2, device=pred.device, dtype=torch.float64)
        sb, sc, sx, sy, sz = pred.stride()
        _lib.call('pw_focal_loss_stats', ops._p(pred), ops._p(t), ops._p(cm), ops._p(cw), B, C, X, Y, Z, sb, sc, sx, sy,
                  sz, int(ignore_index), float(gamma), float(alpha), ops._p(stats), ops._stream())
        out = torch.empty(1, device=pred.device, dtype=torch.float32)
        _lib.call('pw_focal_loss_finish', ops._p(stats), float(loss_weight), ops._p(out), ops._stream())
        ctx.save_for_backward(pred, t, cm if cm is not None else torch.empty(0), cw if cw is not None else torch.empty(0), stats)
        ctx.meta = (dims, ignore_index, gamma, alpha, loss_weight, cm is not None, cw is not None)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, t, cm, cw, stats = ctx.saved_tensors
        dims, ignore_index, gamma, alpha, loss_weight, has_cm, has_cw = ctx.meta
        B, C, X, Y, Z = dims
        gout = g.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_strided(pred.shape, pred.stride(), device=pred.device, dtype=pred.dtype)
        sb, sc, sx, sy, sz = pred.stride()
        _lib.call('pw_focal_loss_grad', ops._p(pred), ops._p(t), ops._p(cm) if has_cm else None,
                  ops._p(cw) if has_cw else None, B, C, X, Y, Z, sb, sc, sx, sy, sz, int(ignore_index), float(gamma),
                  float(alpha), ops._p(stats), float(loss_weight), ops._p(gout), ops._p(grad), ops._stream())
        return grad, None, None, None, None, None, None, None


class CustomFocalLoss(torch.nn.Module):
    """Drop-in for mmdet3d/models/loss_utils/focal_loss.py:163-262 (same constructor arguments, same forward
    signature) on the fused kernels: forward(pred (B,C,X,Y,Z) logits, target (B,X,Y,Z), weight (C,), ...,
    ignore_index=255, camera_mask=None) -> scalar.  The radial map follows the grid's first two axes (the reference
    hard-codes 200 x 200, which is what the configs use)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=100.0, activated=False):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        if activated:
            raise NotImplementedError('activated=True (probabilities as input) is not built')
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, ignore_index=255, reduction_override=None,
                camera_mask=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        # the reference's weighted branch ends in loss.sum(-1).mean() whatever `reduction` says (focal_loss.py:156-158)
        return _FocalLoss.apply(pred, target, camera_mask, weight, ignore_index, self.gamma, self.alpha, self.loss_weight)


class _Lovasz(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probas, labels, camera_mask, ignore):
        dims, t, cm, _ = _args(probas, labels, camera_mask, None)
        B, C, X, Y, Z = dims
        ign = -1 if ignore is None else int(ignore)
        nbytes = _lib.call_size('pw_lovasz_workspace_bytes', B * X * Y * Z, C, ign)
        if nbytes == 0:
            raise _lib.PreworldHipError('pw_lovasz_workspace_bytes failed (n_cls > 32?)')
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=probas.device)
        out = torch.empty(2, device=probas.device, dtype=torch.float32)
        need_grad = probas.requires_grad
        dprob = torch.zeros_like(probas, memory_format=torch.preserve_format) if need_grad else None
        sb, sc, sx, sy, sz = probas.stride()
        if need_grad and dprob.stride() != probas.stride():
            dprob = torch.empty_strided(probas.shape, probas.stride(), device=probas.device, dtype=probas.dtype).zero_()
        _lib.call('pw_lovasz_softmax', ops._p(probas), ops._p(t), ops._p(cm), B, C, X, Y, Z, sb, sc, sx, sy, sz, ign,
                  ops._p(ws), nbytes, ops._p(out[0:1]), ops._p(out[1:2]), ops._p(dprob), ops._stream())
        if need_grad:
            ctx.save_for_backward(dprob, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        dprob, out = ctx.saved_tensors
        return dprob * (g * out[1]), None, None, None


def lovasz_softmax(probas, labels, classes='present', per_image=False, ignore=None, camera_mask=None):
    """Drop-in for mmdet3d/models/detectors/lovasz_softmax.py:157-174 as loss_voxel calls it (preworld.py:155):
    probas (B,C,X,Y,Z) softmax probabilities, labels (B,X,Y,Z); classes='present', per_image=False only.
    Precondition: probas in [0, 1] (what a softmax delivers).  The sort key holds errors |fg - p| in [0, 1] (larger ones are clamped
    to 1) and the gradient takes its sign from fg alone (d|fg - p| / dp = -1 for fg = 1, +1 for fg = 0, true only inside [0, 1])."""
    if classes != 'present' or per_image:
        raise NotImplementedError("only classes='present', per_image=False (the reference's call) is built")
    if probas.dim() != 5:
        raise _lib.PreworldHipError('probas must be (B, C, X, Y, Z)')
    return _Lovasz.apply(probas, labels, camera_mask, ignore)


class _Sanitise(torch.autograd.Function):
    """NaN / +-Inf -> 0, the gradient passing where the value was finite (what the reference's two masked assignments do)"""

    @staticmethod
    def forward(ctx, x):
        y = torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return g * (x == y)                      # NaN != 0 and Inf != 0: exactly the replaced elements drop out


def _softmax_classes(x):
    """torch.softmax(x, dim=1) of (B,C,X,Y,Z) logits.  The OccHead's logits are a permuted view of a channels-last buffer (class axis
    innermost in memory); torch.softmax would first copy them into (B,C,X,Y,Z) order -- a 46 MB transpose forward and another for the
    gradient.  Here the softmax runs over the innermost memory axis and the result is viewed back (the loss kernels take strides)."""
    order = sorted(range(x.dim()), key=lambda i: -x.stride(i))
    xl = x.permute(order)
    if order[-1] != 1 or not xl.is_contiguous():
        return torch.softmax(x, dim=1)
    inv = [order.index(i) for i in range(x.dim())]
    return torch.softmax(xl, dim=-1).permute(inv)


def loss_voxel(output_voxels, target_voxels, class_weights, camera_mask=None, empty_idx=17, use_focal_loss=True,
               weight_voxel_ce=1.0, weight_voxel_sem_scal=1.0, weight_voxel_geo_scal=1.0, weight_voxel_lovasz=1.0,
               focal_loss=None):
    """PreWorld.loss_voxel (mmdet3d/models/detectors/preworld.py:136-157) with the same dictionary keys:
    class_weights = the detector's 17 `1/log(freq)` weights (a 0 for the free class is appended here as there, :147);
    NaN / Inf logits count as zeros (:137-138; the caller's tensor is not modified).  Four passes over the logits forward (focal or CE+sem+geo
    share one when use_focal_loss is False) instead of the reference's several hundred masked reductions."""
    # :137-138 zero NaN / Inf logits with two in-place masked assignments.  Same values from ONE elementwise pass in the tensor's
    # own memory layout: in place on the permuted view of an autograd tensor the assignments cost ~9 transposing copies of the
    # 46 MB logits per step (CopySlices forward and backward, 0.2 ms each); the caller's tensor is left as it was
    output_voxels = _Sanitise.apply(output_voxels)
    cw = torch.cat([class_weights.to(output_voxels.device), torch.zeros(1, device=output_voxels.device)]).type_as(output_voxels)
    ce, sem, geo = voxel_losses(output_voxels, target_voxels, cw, 255, empty_idx, camera_mask)
    if use_focal_loss:
        ce = (focal_loss or CustomFocalLoss())(output_voxels, target_voxels, cw, 255, camera_mask=camera_mask)
    return {'loss_voxel_ce': weight_voxel_ce * ce,
            'loss_voxel_sem': weight_voxel_sem_scal * sem,
            'loss_voxel_geo': weight_voxel_geo_scal * geo,
            'loss_voxel_lovasz': weight_voxel_lovasz * lovasz_softmax(_softmax_classes(output_voxels), target_voxels,
                                                                      ignore=empty_idx, camera_mask=camera_mask)}
