"""Training-mode forward + backward of the voxel encoder blocks on the HIP kernels of csrc/pw_train.hip.

What torch autograd runs behind the reference's `BasicBlock3D` / `CustomResNet3D` (mmdet3d/models/backbones/resnet.py:88-184:
Conv3d(bias=False) -> BatchNorm3d with batch statistics -> ReLU, residual add) when `forward_train` calls them:
  * conv forward   : the fp32 MFMA kernels of the inference path with unit scale / zero bias (raw conv output)
  * conv dgrad     : stride 1 (3x3x3 and 1x1x1) = a forward conv with the weights flipped and transposed, on the same kernels;
                     3x3x3 stride 2 = pw_conv3d_dgrad_s2
  * conv wgrad     : pw_conv3d_wgrad (fp32 MFMA, K = voxels, deterministic two-stage reduction)
  * BatchNorm3d    : pw_bn_stats / pw_bn_apply / pw_bn_bwd_reduce / pw_bn_bwd_apply (batch statistics in double, running
                     statistics updated like nn.BatchNorm3d: momentum, unbiased variance, num_batches_tracked)
Everything is fp32 channels-last (B, D, H, W, C).  `ConvModule3d.forward_cl`, `BasicBlock3D.forward_cl` and therefore
`CustomResNet3D.forward_cl` / `.forward` dispatch here when the module is in training mode.  The rest of `forward_train`
(FPN, final_conv, OccHead, forecast backward) is not built."""
import torch

from . import _lib, ops

_f32 = torch.float32


def _cl(t, name):
    if t.dtype != _f32 or not t.is_cuda or not t.is_contiguous():
        raise _lib.PreworldHipError('%s must be a contiguous float32 device tensor' % name)
    return t


# ------------------------------------------------------------------------------ raw ops
def conv3d_raw(x, w, stride=1):
    """Conv3d(bias=False, padding=k//2) of channels-last x (B,D,H,W,Cin) with torch-layout w (Cout,Cin,k,k,k) -> (B,Do,Ho,Wo,Cout);
    fp32 MFMA kernels of the inference path, no scale / bias / activation."""
    k = w.shape[2]
    return ops.conv3d_ndhwc(_cl(x, 'x'), ops.pack_conv_weight(w.detach()), cout0=w.shape[0], ksize=k, stride=stride)


def conv3d_dgrad(dy, w, x_shape, stride=1):
    """d loss / d x of conv3d_raw: dy (B,Do,Ho,Wo,Cout) -> (B,D,H,W,Cin)."""
    k = w.shape[2]
    Cout, Cin = w.shape[:2]
    B, D, H, W, _ = x_shape
    if stride == 1:
        if Cout % 32:
            raise _lib.PreworldHipError('conv3d_dgrad: Cout %% 32 == 0 expected (encoder layers)')
        wt = w.detach().flip(2, 3, 4).transpose(0, 1).contiguous() if k == 3 else w.detach().transpose(0, 1).contiguous()
        return ops.conv3d_ndhwc(_cl(dy, 'dy'), ops.pack_conv_weight(wt), cout0=Cin, ksize=k, stride=1)
    if k != 3 or stride != 2:
        raise _lib.PreworldHipError('conv3d_dgrad: only stride 1 (k 1 | 3) and 3x3x3 stride 2 are built')
    dx = torch.empty(B, D, H, W, Cin, device=dy.device, dtype=_f32)
    _lib.call('pw_conv3d_dgrad_s2', ops._p(_cl(dy, 'dy')), ops._p(_cl(w.detach().permute(2, 3, 4, 0, 1).contiguous(), 'w')), ops._p(dx), B, D, H, W, Cin,
              Cout, ops._stream())
    return dx


def conv3d_wgrad(x, dy, w_shape, stride=1):
    """d loss / d w of conv3d_raw, torch layout (Cout,Cin,k,k,k)."""
    Cout, Cin, k = w_shape[0], w_shape[1], w_shape[2]
    B, D, H, W, _ = x.shape
    nbytes = _lib.call_size('pw_conv3d_wgrad_workspace_bytes', B, D, H, W, Cin, Cout, k, stride)
    ws = ops._workspace(nbytes, x.device)
    dw = torch.empty(tuple(w_shape), device=x.device, dtype=_f32)
    _lib.call('pw_conv3d_wgrad', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), ops._p(dw), ops._p(ws), nbytes, B, D, H, W, Cin, Cout,
              k, stride, ops._stream())
    return dw


def bn_stats(x, eps):
    C = x.shape[-1]
    N = x.numel() // C
    nbytes = _lib.call_size('pw_bn_workspace_bytes', C)
    ws = ops._workspace(nbytes, x.device)
    mean, var, rstd = (torch.empty(C, device=x.device, dtype=_f32) for _ in range(3))
    _lib.call('pw_bn_stats', ops._p(_cl(x, 'x')), N, C, float(eps), ops._p(ws), nbytes, ops._p(mean), ops._p(var), ops._p(rstd),
              ops._stream())
    return mean, var, rstd


def bn_apply(x, mean, rstd, gamma, beta, residual=None, relu=False):
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.call('pw_bn_apply', ops._p(_cl(x, 'x')), x.numel() // C, C, ops._p(mean), ops._p(rstd), ops._p(_cl(gamma, 'gamma')),
              ops._p(_cl(beta, 'beta')), ops._p(_cl(residual, 'residual') if residual is not None else None), int(relu), ops._p(y),
              ops._stream())
    return y


def bn_backward(x, dy, y, mean, rstd, gamma, relu, want_dres):
    """-> (dx, dgamma, dbeta, dres or None)"""
    C = x.shape[-1]
    N = x.numel() // C
    nbytes = _lib.call_size('pw_bn_workspace_bytes', C)
    ws = ops._workspace(nbytes, x.device)
    s0, s1 = torch.empty(C, device=x.device, dtype=_f32), torch.empty(C, device=x.device, dtype=_f32)
    yp = ops._p(_cl(y, 'y')) if relu else None
    _lib.call('pw_bn_bwd_reduce', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), yp, N, C, ops._p(mean), ops._p(rstd), int(relu),
              ops._p(ws), nbytes, ops._p(s0), ops._p(s1), ops._stream())
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    _lib.call('pw_bn_bwd_apply', ops._p(x), ops._p(dy), yp, N, C, ops._p(mean), ops._p(rstd), ops._p(_cl(gamma, 'gamma')),
              ops._p(s0), ops._p(s1), int(relu), ops._p(dx), ops._p(dres), ops._stream())
    return dx, s1, s0, dres


# ------------------------------------------------------------------------------ autograd
class Conv3dCL(torch.autograd.Function):
    """y = conv3d(x, w), channels-last, bias-free, padding k//2"""

    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return conv3d_raw(x, w, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = conv3d_dgrad(dy, w, x.shape, ctx.stride) if ctx.needs_input_grad[0] else None
        dw = conv3d_wgrad(x, dy, w.shape, ctx.stride) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class BatchNormCL(torch.autograd.Function):
    """y = relu?(batch_norm(x; batch statistics) * gamma + beta (+ residual)) -> (y, batch mean, biased batch variance)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, relu):
        mean, var, rstd = bn_stats(x, eps)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = bn_apply(x, mean, rstd, g, b, residual, relu)
        ctx.save_for_backward(x, y, mean, rstd, g)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, y, mean, rstd, g = ctx.saved_tensors
        dx, dgamma, dbeta, dres = bn_backward(x, dy.contiguous(), y, mean, rstd, g, ctx.relu, ctx.has_res and ctx.needs_input_grad[3])
        return dx, dgamma, dbeta, dres, None, None


def _update_running(bn, mean, var, n):
    """nn.BatchNorm3d bookkeeping (torch/nn/modules/batchnorm.py): exponential average with the UNBIASED batch variance"""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1.0 - m).add_(var * (n / max(n - 1.0, 1.0)), alpha=m)


# ------------------------------------------------------------------------------ module-level training forwards
def conv_module_forward(m, x, residual=None, relu=None):
    """ConvModule3d (conv -> BN -> ReLU) in training mode; `relu` overrides the module's own activation flag and, with
    `residual`, gives BasicBlock3D's relu(bn(conv(x)) + identity)."""
    if m.conv.bias is not None or not m.with_norm:
        raise NotImplementedError('training path: bias-free conv + BatchNorm3d modules only (the encoder blocks)')
    y = Conv3dCL.apply(x.contiguous(), m.conv.weight, m.stride)
    relu = m.with_activation if relu is None else relu
    out, mean, var = BatchNormCL.apply(y, m.bn.weight, m.bn.bias, residual, m.bn.eps, relu)
    _update_running(m.bn, mean, var, float(y.numel() // y.shape[-1]))
    return out


def basic_block_forward(blk, x):
    """resnet.py:108-123 in training mode: relu(conv2(conv1(x)) + downsample(x)), every conv followed by batch-stat BN"""
    identity = conv_module_forward(blk.downsample, x) if blk.downsample is not None else x
    y = conv_module_forward(blk.conv1, x)
    return conv_module_forward(blk.conv2, y, residual=identity.contiguous(), relu=True)
