"""Training side of the voxel path: forward under torch autograd + backward on the HIP kernels of csrc/pw_train.hip.

What torch autograd runs behind the reference's modules when `forward_train` (preworld.py:229-309,
preworld_temporal_traj.py:372-530) calls them:
  * BasicBlock3D / CustomResNet3D (backbones/resnet.py:88-184): Conv3d(bias=False) -> BatchNorm3d (batch statistics) -> ReLU,
    residual add                                                   conv_module_forward / basic_block_forward
  * LSSFPN3D (necks/lss_fpn.py:132-148)                            fpn_forward (per-level 1x1x1 convs, trilinear up-sampling + adjoint)
  * final_conv (conv + bias + ReLU)                                conv_bias_act_forward
  * OccHead.forward_coarse_voxel (heads/occupancy_head.py:124-161) occ_head_forward
  * the forecast step and the trajectory branch (:448-470)         fusion_step / downscale_forward
Kernels:
  * conv forward   : the fp32 MFMA kernels of the inference path with unit scale / zero bias (raw conv output)
  * conv dgrad     : stride 1 (3x3x3 and 1x1x1) = a forward conv with the weights flipped and transposed, on the same kernels;
                     3x3x3 stride 2 = pw_conv3d_dgrad_s2; 2x2x2 stride 2 = pw_conv3d_dgrad_k2s2
  * conv wgrad     : pw_conv3d_wgrad (fp32 MFMA, K = voxels, deterministic two-stage reduction)
  * BatchNorm3d    : pw_bn_stats / pw_bn_apply / pw_bn_bwd_reduce / pw_bn_bwd_apply (batch statistics in double, running
                     statistics updated like nn.BatchNorm3d: momentum, unbiased variance, num_batches_tracked)
  * up-sampling    : pw_upsample_trilinear_add / pw_upsample_trilinear_adjoint
Everything is fp32 channels-last (B, D, H, W, C); the per-sample dense layers (and per-voxel ones of unusual width) are library GEMMs (torch.matmul /
F.linear).  `ConvModule3d.forward_cl`, `BasicBlock3D.forward_cl`, `CustomResNet3D`, `LSSFPN3D`, `OccHead.forward` dispatch here
when the module is in training mode; `PreWorld.forward_train` / `PreWorld4DTraj.forward_train` (detectors.py) compose them."""
import os

import torch

from . import _lib, ops

_f32 = torch.float32
# module-level settings (tests set them; no environment switches):
# _WGRAD 'f32' keeps the exact-fp32 MFMA weight-gradient kernel for the 3x3x3 stride-1 layers too (default: split-fp16, pw_train_h2.hip)
_WGRAD = 'h2'
# _DGRAD_S2 'valu' keeps the plain-FMA stride-2 data-gradient kernel (pw_conv3d_dgrad_s2, exact fp32 order), which also serves channel
# counts that are not multiples of 32 (default: split-fp16 parity-class kernel, pw_dgrad_s2_h2.hip)
_DGRAD_S2 = 'h2'


def _cl(t, name):
    if t.dtype != _f32 or not t.is_cuda or not t.is_contiguous():
        raise _lib.PreworldHipError('%s must be a contiguous float32 device tensor' % name)
    return t


# ------------------------------------------------------------------------------ raw ops
def pack_weight(w, wino, flip_t=False):
    """ops.pack_conv_weight(w) / ops.pack_conv_weight_wino(w) -- or, with flip_t, of w.flip(2, 3, 4).transpose(0, 1) -- in one
    launch (pw_pack_conv_weight): training re-packs every weight every step, and as a dozen torch ops per pack that made the eager
    step host-bound."""
    Cout, Cin, k = w.shape[0], w.shape[1], w.shape[2]
    cout_p, cin_p = (Cin, Cout) if flip_t else (Cout, Cin)
    cout_total = (cout_p + 31) // 32 * 32
    if wino:
        out = torch.empty(cin_p // 32, 64, cout_total // 16, 64, 8, device=w.device, dtype=_f32)
    else:
        out = torch.empty(cin_p // 32, k ** 3, cout_total // 32, 64, 16, device=w.device, dtype=_f32)
    _lib.call('pw_pack_conv_weight', ops._p(_cl(w.detach().contiguous(), 'w')), Cout, Cin, k, int(bool(flip_t)), cout_total, ops._p(out),
              int(bool(wino)), ops._stream())
    return out


def _conv_fwd(x, w, stride, flip_t=False, residual=None):
    """bias-free conv of channels-last x with torch-layout w (or, flip_t, with w.flip(2,3,4).transpose(0,1): the stride-1 data
    gradient) on the fp32 inference kernels: Winograd F(2x2x2,3x3x3) where the module stack uses it (3x3x3 stride 1, enough tiles),
    else direct"""
    from .modules import _use_wino
    k, cout = w.shape[2], (w.shape[1] if flip_t else w.shape[0])
    cout_total = (cout + 31) // 32 * 32
    if _use_wino(x, cout_total, k, stride):
        return ops.conv3d_wino(x, pack_weight(w, True, flip_t), cout0=cout, residual=residual)
    return ops.conv3d_ndhwc(x, pack_weight(w, False, flip_t), cout0=cout, ksize=k, stride=stride, residual=residual)


def _conv_fwd_pair(x, w1, w2, stride):
    """(conv(x, w1), conv(x, w2)) from ONE pass over x: the packed columns of both weights side by side, two destinations
    (pw_conv3d_ndhwc / pw_conv3d_wino cout0 / cout1) -- conv1 and downsample of a BasicBlock3D read the same input."""
    from .modules import _use_wino
    k, c1, c2 = w1.shape[2], w1.shape[0], w2.shape[0]
    w = torch.cat([w1, w2], 0)
    if _use_wino(x, c1 + c2, k, stride):
        return ops.conv3d_wino(x, pack_weight(w, True), cout0=c1, cout1=c2)
    return ops.conv3d_ndhwc(x, pack_weight(w, False), cout0=c1, cout1=c2, ksize=k, stride=stride)


def conv3d_raw(x, w, stride=1):
    """Conv3d(bias=False, padding=k//2) of channels-last x (B,D,H,W,Cin) with torch-layout w (Cout,Cin,k,k,k) -> (B,Do,Ho,Wo,Cout);
    fp32 MFMA kernels of the inference path, no scale / bias / activation."""
    k = w.shape[2]
    if k == 2 and stride != 2:
        raise _lib.PreworldHipError('2x2x2 convs are built for stride 2 (the trajectory branch)')
    return _conv_fwd(_cl(x, 'x'), w.detach(), stride)


def conv3d_dgrad(dy, w, x_shape, stride=1, accumulate=None):
    """d loss / d x of conv3d_raw: dy (B,Do,Ho,Wo,Cout) -> (B,D,H,W,Cin).  accumulate (stride 1): a (B,D,H,W,Cin) tensor the result
    is added to in the kernel's epilogue (returned; the gradient of an input that two convolutions read)."""
    k = w.shape[2]
    Cout, Cin = w.shape[:2]
    B, D, H, W, _ = x_shape
    if stride == 1:
        if Cout % 32:
            raise _lib.PreworldHipError('conv3d_dgrad: Cout %% 32 == 0 expected (encoder layers)')
        return _conv_fwd(_cl(dy, 'dy'), w, 1, flip_t=True, residual=accumulate)
    if accumulate is not None:
        raise _lib.PreworldHipError('conv3d_dgrad: accumulate is built for stride 1')
    if stride != 2 or k not in (2, 3):
        raise _lib.PreworldHipError('conv3d_dgrad: only stride 1 (k 1 | 3), 3x3x3 stride 2 and 2x2x2 stride 2 are built')
    if k == 3 and Cout % 32 == 0 and Cin % 32 == 0 and _DGRAD_S2 != 'valu':
        # the 8 parity classes of the fine grid as dense 1- to 8-tap convolutions over dY, split-fp16 operands: 27 tap products per
        # 8 fine voxels and no zero fill (round 3 inserted zeros into dY and ran the stride-1 Winograd kernel: 27 per voxel)
        dyh = ops.f32_to_h2(_cl(dy, 'dy'))
        nbytes = _lib.call_size('pw_conv3d_dgrad_s2_h2_workspace_bytes', Cin, Cout)
        ws = ops._workspace(nbytes, dy.device)
        dx = torch.empty(B, D, H, W, Cin, device=dy.device, dtype=_f32)
        _lib.call('pw_conv3d_dgrad_s2_h2', ops._p(dyh.buf), ops._p(dyh.rng), ops._p(_cl(w.detach(), 'w')), ops._p(dx), ops._p(ws), nbytes,
                  B, D, H, W, Cin, Cout, ops._stream())
        return dx
    dx = torch.empty(B, D, H, W, Cin, device=dy.device, dtype=_f32)
    _lib.call('pw_conv3d_dgrad_s2' if k == 3 else 'pw_conv3d_dgrad_k2s2', ops._p(_cl(dy, 'dy')),
              ops._p(_cl(w.detach().permute(2, 3, 4, 0, 1).contiguous(), 'w')), ops._p(dx), B, D, H, W, Cin, Cout, ops._stream())
    return dx


def _amax_key(t):
    return (t.data_ptr(), t._version, t.numel())


def _amax_of(t):
    """the 256 partial maxima a BatchNorm kernel recorded when it wrote this tensor (BatchNormCL: y forward, dx backward), or None.
    They ride on the tensor object: autograd hands the same object to the consuming Function; a tensor that was summed, padded or
    copied on the way simply arrives without them and gets an absmax pass, and so does one that was modified in place afterwards or
    re-pointed: the record is keyed on (data_ptr, version counter, numel).  A write through an alias that does not share the version
    counter (`y.data`) is not seen by the key; PW_AMAX_CHECK=1 (tests) compares every recorded maximum with a fresh pw_absmax2 pass."""
    rec = getattr(t, '_pw_amax', None)
    if rec is None or rec[1] != _amax_key(t):         # written to in place / re-pointed since the maxima were recorded
        return None
    if _AMAX_CHECK:
        fresh = torch.empty(512, device=t.device, dtype=_f32)
        _lib.call('pw_absmax2', ops._p(_cl(t, 't')), t.numel(), ops._p(_cl(t, 't')), 0, ops._p(fresh), ops._stream())
        got, want = float(rec[0].max()), float(fresh[:256].max())
        _AMAX_STATS['checked'] += 1
        assert got == want, 'recorded absmax %r does not describe the tensor (fresh pass: %r)' % (got, want)
    return rec[0]


_AMAX_CHECK = os.environ.get('PW_AMAX_CHECK', '0') == '1'
_AMAX_STATS = {'checked': 0}


def _set_amax(t, amax):
    t._pw_amax = (amax, _amax_key(t))


def conv3d_wgrad(x, dy, w_shape, stride=1, x_amax=None):
    """d loss / d w of conv3d_raw, torch layout (Cout,Cin,k,k,k).  x_amax: recorded maxima of x (see _amax_of) when the caller kept them."""
    Cout, Cin, k = w_shape[0], w_shape[1], w_shape[2]
    B, D, H, W, _ = x.shape
    if k == 3 and stride == 1 and Cin % 32 == 0 and Cout % 32 == 0 and _WGRAD != 'f32':
        # split-fp16 operands on the fp16 matrix cores (pw_conv3d_wgrad_h2): 22-bit products, fp32 accumulation; the two maxima
        # (device side, no sync) give the per-tensor power-of-two pre-scales
        ax, ay = _amax_of(x) if x_amax is None else x_amax, _amax_of(dy)
        if ax is None or ay is None:
            # one pass over whichever operand did not come with its maximum recorded by the kernel that wrote it
            amax2 = torch.empty(512, device=x.device, dtype=_f32)
            _lib.call('pw_absmax2', ops._p(_cl(x, 'x')), x.numel() if ax is None else 0, ops._p(_cl(dy, 'dy')), dy.numel() if ay is None else 0,
                      ops._p(amax2), ops._stream())
            ax, ay = amax2[:256] if ax is None else ax, amax2[256:] if ay is None else ay
        nbytes = _lib.call_size('pw_conv3d_wgrad_h2_workspace_bytes', B, D, H, W, Cin, Cout)
        ws = ops._workspace(nbytes, x.device)
        dw = torch.empty(tuple(w_shape), device=x.device, dtype=_f32)
        _lib.call('pw_conv3d_wgrad_h2', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), ops._p(dw), ops._p(ax), ops._p(ay), ops._p(ws), nbytes,
                  B, D, H, W, Cin, Cout, ops._stream())
        return dw
    nbytes = _lib.call_size('pw_conv3d_wgrad_workspace_bytes', B, D, H, W, Cin, Cout, k, stride)
    ws = ops._workspace(nbytes, x.device)
    dw = torch.empty(tuple(w_shape), device=x.device, dtype=_f32)
    _lib.call('pw_conv3d_wgrad', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), ops._p(dw), ops._p(ws), nbytes, B, D, H, W, Cin, Cout,
              k, stride, ops._stream())
    return dw


def bn_stats(x, eps, amax=None, running=None):
    """amax: float32[256] buffer that bn_apply(..., amax=) will record max |y| into (cleared here); running: (running_mean, running_var,
    momentum, num_batches_tracked or None) to update like nn.BatchNorm3d in the same launch"""
    C = x.shape[-1]
    N = x.numel() // C
    nbytes = _lib.call_size('pw_bn_workspace_bytes', C)
    ws = ops._workspace(nbytes, x.device)
    mean, var, rstd = (torch.empty(C, device=x.device, dtype=_f32) for _ in range(3))
    _lib.call('pw_bn_stats', ops._p(_cl(x, 'x')), N, C, float(eps), ops._p(ws), nbytes, ops._p(mean), ops._p(var), ops._p(rstd),
              ops._p(amax), ops._p(running[0]) if running else None, ops._p(running[1]) if running else None,
              float(running[2]) if running else 0.0, ops._p(running[3]) if running and running[3] is not None else None, ops._stream())
    return mean, var, rstd


def bn_apply(x, mean, rstd, gamma, beta, residual=None, relu=False, amax=None):
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.call('pw_bn_apply', ops._p(_cl(x, 'x')), x.numel() // C, C, ops._p(mean), ops._p(rstd), ops._p(_cl(gamma, 'gamma')),
              ops._p(_cl(beta, 'beta')), ops._p(_cl(residual, 'residual') if residual is not None else None), int(relu), ops._p(y),
              ops._p(amax), ops._stream())
    if amax is not None:
        _set_amax(y, amax)
    return y


def bn_backward(x, dy, y, mean, rstd, gamma, relu, want_dres, record_amax=False):
    """-> (dx, dgamma, dbeta, dres or None); record_amax: dx carries its recorded maxima (_amax_of)"""
    amax = torch.empty(256, device=x.device, dtype=_f32) if record_amax else None
    C = x.shape[-1]
    N = x.numel() // C
    nbytes = _lib.call_size('pw_bn_workspace_bytes', C)
    ws = ops._workspace(nbytes, x.device)
    s0, s1 = torch.empty(C, device=x.device, dtype=_f32), torch.empty(C, device=x.device, dtype=_f32)
    yp = ops._p(_cl(y, 'y')) if relu else None
    _lib.call('pw_bn_bwd_reduce', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), yp, N, C, ops._p(mean), ops._p(rstd), int(relu),
              ops._p(ws), nbytes, ops._p(s0), ops._p(s1), ops._p(amax), ops._stream())
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    _lib.call('pw_bn_bwd_apply', ops._p(x), ops._p(dy), yp, N, C, ops._p(mean), ops._p(rstd), ops._p(_cl(gamma, 'gamma')),
              ops._p(s0), ops._p(s1), int(relu), ops._p(dx), ops._p(dres), ops._p(amax), ops._stream())
    if amax is not None:
        _set_amax(dx, amax)
    return dx, s1, s0, dres


# ------------------------------------------------------------------------------ autograd
class Conv3dCL(torch.autograd.Function):
    """y = conv3d(x, w), channels-last, bias-free, padding k//2"""

    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.x_amax = stride, _amax_of(x)
        return conv3d_raw(x, w, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        pad = (-w.shape[0]) % 32
        if pad and ctx.stride == 1:
            # a narrow layer (OccHead's 32 -> 16 conv, the attribute MLPs' last Linear): the gradient kernels take multiples of 32
            # output channels, so dY and W get zero columns HERE -- one pad kernel, instead of a padded weight in the forward graph
            # whose output slice autograd turns into a strided copy forward and a zero fill + strided copy backward
            dy = torch.nn.functional.pad(dy, (0, pad))
            wp = torch.cat([w.detach(), w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
            dx = conv3d_dgrad(dy, wp, x.shape, 1) if ctx.needs_input_grad[0] else None
            dw = conv3d_wgrad(x, dy, wp.shape, 1, ctx.x_amax)[:w.shape[0]] if ctx.needs_input_grad[1] else None
            return dx, dw, None
        dx = conv3d_dgrad(dy, w, x.shape, ctx.stride) if ctx.needs_input_grad[0] else None
        dw = conv3d_wgrad(x, dy, w.shape, ctx.stride, ctx.x_amax) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def _sync_world(sync):
    """process group size when this BatchNorm is a SyncBN (norm_cfg type 'SyncBN', configs/preworld/**: OccHead) and a group is up"""
    import torch.distributed as dist
    return dist.get_world_size() if (sync and dist.is_available() and dist.is_initialized()) else 1


def _all_reduce_sum(t):
    """in-place SUM over the ranks; RCCL takes the device tensor, gloo (CPU tests) goes through host memory"""
    import torch.distributed as dist
    if dist.get_backend() == 'gloo' and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


class BatchNormCL(torch.autograd.Function):
    """y = relu?(batch_norm(x; batch statistics) * gamma + beta (+ residual)) -> (y, batch mean, biased batch variance, count).
    sync: SyncBatchNorm semantics (torch/nn/modules/_functions.py SyncBatchNorm, what mmcv builds for norm_cfg 'SyncBN'): the
    statistics are those of the batch over ALL ranks -- one all-reduce of [sum x, sum x^2, n] per channel in the forward pass,
    one of [sum dz, sum dz x_hat] in the backward pass; d gamma / d beta stay per rank (DDP averages parameter gradients)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, relu, sync=False, running=None):
        ctx.set_materialize_grads(False)          # the statistics outputs never carry gradients: no zero tensors made for them
        amax = torch.empty(256, device=x.device, dtype=_f32)
        # running: (running_mean, running_var, momentum, num_batches_tracked) updated inside the statistics launch -- local statistics
        # only (a SyncBN over several ranks updates them from the all-reduced ones: _update_running)
        mean, var, rstd = bn_stats(x, eps, amax, running if _sync_world(sync) == 1 else None)
        n_local = float(x.numel() // x.shape[-1])
        n_total = n_local
        if _sync_world(sync) > 1:
            C = x.shape[-1]
            st = torch.cat([mean.double() * n_local, (var.double() + mean.double() ** 2) * n_local,
                            torch.full((1,), n_local, dtype=torch.float64, device=x.device)])
            _all_reduce_sum(st)
            n_total = float(st[-1])
            m64 = st[:C] / n_total
            v64 = (st[C:2 * C] / n_total - m64 ** 2).clamp_min(0)
            mean, var = m64.float(), v64.float()
            rstd = torch.rsqrt(v64 + eps).float()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = bn_apply(x, mean, rstd, g, b, residual, relu, amax)
        ctx.save_for_backward(x, y, mean, rstd, g)
        ctx.relu, ctx.has_res, ctx.sync, ctx.n_ratio = bool(relu), residual is not None, bool(sync), n_local / n_total
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var, n_total                                  # the row count goes out as a plain number

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dc):
        x, y, mean, rstd, g = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None, None, None, None
        if _sync_world(ctx.sync) > 1:
            return BatchNormCL._backward_sync(ctx, x, dy.contiguous(), y, mean, rstd, g)
        dx, dgamma, dbeta, dres = bn_backward(x, dy.contiguous(), y, mean, rstd, g, ctx.relu, ctx.has_res and ctx.needs_input_grad[3], True)
        return dx, dgamma, dbeta, dres, None, None, None, None

    @staticmethod
    def _backward_sync(ctx, x, dy, y, mean, rstd, g):
        """bn_backward with the two per-channel sums taken over all ranks.  pw_bn_bwd_apply divides the sums by its row count
        (the local N); handing it the global sums scaled by N_local / N_total makes that the global mean."""
        C = x.shape[-1]
        N = x.numel() // C
        nbytes = _lib.call_size('pw_bn_workspace_bytes', C)
        ws = ops._workspace(nbytes, x.device)
        s0, s1 = torch.empty(C, device=x.device, dtype=_f32), torch.empty(C, device=x.device, dtype=_f32)
        yp = ops._p(_cl(y, 'y')) if ctx.relu else None
        _lib.call('pw_bn_bwd_reduce', ops._p(_cl(x, 'x')), ops._p(_cl(dy, 'dy')), yp, N, C, ops._p(mean), ops._p(rstd), int(ctx.relu),
                  ops._p(ws), nbytes, ops._p(s0), ops._p(s1), None, ops._stream())
        tot = _all_reduce_sum(torch.cat([s0, s1]).double()) * ctx.n_ratio
        g0, g1 = tot[:C].float().contiguous(), tot[C:].float().contiguous()
        dx = torch.empty_like(x)
        want_dres = ctx.has_res and ctx.needs_input_grad[3]
        dres = torch.empty_like(x) if want_dres else None
        _lib.call('pw_bn_bwd_apply', ops._p(x), ops._p(dy), yp, N, C, ops._p(mean), ops._p(rstd), ops._p(_cl(g, 'gamma')),
                  ops._p(g0), ops._p(g1), int(ctx.relu), ops._p(dx), ops._p(dres), None, ops._stream())
        return dx, s1, s0, dres, None, None, None, None


def _update_running(bn, mean, var, n):
    """nn.BatchNorm3d bookkeeping (torch/nn/modules/batchnorm.py): exponential average with the UNBIASED batch variance;
    n: (1,) device tensor (or a plain number), the number of rows the statistics were taken over (all ranks for a SyncBN) -- no host sync"""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    if bn.momentum is not None and bn.running_mean.is_cuda and bn.running_mean.dtype == _f32 and bn.running_var.dtype == _f32:
        on_dev = torch.is_tensor(n)
        _lib.call('pw_bn_update_running', ops._p(mean), ops._p(var), mean.numel(), 0.0 if on_dev else float(n),
                  ops._p(n.float().contiguous()) if on_dev else None, float(bn.momentum), ops._p(bn.running_mean), ops._p(bn.running_var),
                  ops._p(bn.num_batches_tracked) if bn.num_batches_tracked is not None else None, ops._stream())
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
        unbias = n / (n - 1.0).clamp_min(1.0) if torch.is_tensor(n) else float(n) / max(float(n) - 1.0, 1.0)
        bn.running_var.mul_(1.0 - m).add_(var * unbias, alpha=m)


class ConvPairCL(torch.autograd.Function):
    """(conv3d(x, w1), conv3d(x, w2)), same kernel size and stride: conv1 and downsample of a BasicBlock3D (resnet.py:108-123) read
    the same input.  Forward: one kernel pass over x with two destinations.  Backward: the two data gradients meet in ONE tensor --
    stride 1: the second convolution adds the first one's result in its epilogue; stride 2: one parity-class kernel over the
    channel-concatenated dY (the pair is a convolution with Cout1 + Cout2 output channels) -- where autograd would launch two
    convolutions and an add over the input-sized gradient."""

    @staticmethod
    def forward(ctx, x, w1, w2, stride):
        ctx.save_for_backward(x, w1, w2)
        ctx.stride, ctx.x_amax = stride, _amax_of(x)
        return _conv_fwd_pair(_cl(x, 'x'), w1.detach(), w2.detach(), stride)

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, w1, w2 = ctx.saved_tensors
        dy1, dy2 = dy1.contiguous(), dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.stride == 1:
                dx = conv3d_dgrad(dy2, w2, x.shape, 1, accumulate=conv3d_dgrad(dy1, w1, x.shape, 1))
            else:
                dx = conv3d_dgrad(torch.cat([dy1, dy2], -1), torch.cat([w1.detach(), w2.detach()], 0), x.shape, ctx.stride)
        xa = ctx.x_amax
        if xa is None and ctx.stride == 1 and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            xa = torch.empty(512, device=x.device, dtype=_f32)           # one pass over x serves both weight gradients
            _lib.call('pw_absmax2', ops._p(x), x.numel(), ops._p(x), 0, ops._p(xa), ops._stream())
            xa = xa[:256]
        dw1 = conv3d_wgrad(x, dy1, w1.shape, ctx.stride, xa) if ctx.needs_input_grad[1] else None
        dw2 = conv3d_wgrad(x, dy2, w2.shape, ctx.stride, xa) if ctx.needs_input_grad[2] else None
        return dx, dw1, dw2, None


# ------------------------------------------------------------------------------ module-level training forwards
def conv_module_forward(m, x, residual=None, relu=None):
    """ConvModule3d (conv -> BN -> ReLU) in training mode; `relu` overrides the module's own activation flag and, with
    `residual`, gives BasicBlock3D's relu(bn(conv(x)) + identity)."""
    if m.conv.bias is not None or not m.with_norm:
        raise NotImplementedError('training path: bias-free conv + BatchNorm3d modules only (the encoder blocks)')
    return _norm_act(m, Conv3dCL.apply(x.contiguous(), m.conv.weight, m.stride), residual, relu)


def _norm_act(m, y, residual=None, relu=None):
    """the BatchNorm (batch statistics) + activation half of a ConvModule3d on its conv output y"""
    return _bn_train(m.bn, y, residual, m.with_activation if relu is None else relu)


def _bn_train(bn, x, residual=None, relu=False):
    """training-mode BatchNorm3d of a channels-last tensor (+ residual, + ReLU) with nn.BatchNorm3d's running-statistics bookkeeping --
    inside the statistics launch when the statistics are local and the momentum fixed, else by _update_running"""
    sync = getattr(bn, 'pw_sync', False)
    fuse = (bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None and bn.running_mean.is_cuda and
            bn.running_mean.dtype == _f32 and bn.running_var.dtype == _f32 and _sync_world(sync) == 1)
    running = (bn.running_mean, bn.running_var, bn.momentum, bn.num_batches_tracked) if fuse else None
    out, mean, var, cnt = BatchNormCL.apply(x, bn.weight, bn.bias, residual, bn.eps, relu, sync, running)
    if not fuse:
        _update_running(bn, mean, var, cnt)
    return out


def _pairable(a, b):
    """two ConvModule3d over the same input that one kernel pass can serve: bias-free, normed, same kernel size / stride, output
    channels in whole 32-column tiles"""
    return (a.conv.bias is None and b.conv.bias is None and a.with_norm and b.with_norm and a.stride == b.stride and
            a.conv.weight.shape[1:] == b.conv.weight.shape[1:] and a.conv.weight.shape[2] == 3 and
            a.conv.weight.shape[0] % 32 == 0 and b.conv.weight.shape[0] % 32 == 0 and a.conv.weight.shape[1] % 32 == 0)


def basic_block_forward(blk, x):
    """resnet.py:108-123 in training mode: relu(conv2(conv1(x)) + downsample(x)), every conv followed by batch-stat BN"""
    if blk.downsample is not None and _pairable(blk.conv1, blk.downsample):
        y1, yd = ConvPairCL.apply(x.contiguous(), blk.conv1.conv.weight, blk.downsample.conv.weight, blk.conv1.stride)
        identity = _norm_act(blk.downsample, yd)          # the reference normalises the identity branch first (BN bookkeeping order)
        y = _norm_act(blk.conv1, y1)
    else:
        identity = conv_module_forward(blk.downsample, x) if blk.downsample is not None else x
        y = conv_module_forward(blk.conv1, x)
    return conv_module_forward(blk.conv2, y, residual=identity.contiguous(), relu=True)


# ------------------------------------------------------------------------------ FPN / final_conv / OccHead in training mode
def upsample_add(lo, hi, accumulate):
    """hi (B,Dh,Hh,Wh,C) (+)= trilinear(lo (B,Dl,Hl,Wl,C)), align_corners=True (torch's index rule)"""
    B, Dl, Hl, Wl, C = lo.shape
    _lib.call('pw_upsample_trilinear_add', ops._p(_cl(lo, 'lo')), ops._p(_cl(hi, 'hi')), B, Dl, Hl, Wl, hi.shape[1], hi.shape[2],
              hi.shape[3], C, int(accumulate), ops._stream())
    return hi


def upsample_adjoint(dhi, lo_shape):
    B, Dl, Hl, Wl, C = lo_shape
    dlo = torch.empty(tuple(lo_shape), device=dhi.device, dtype=_f32)
    dims = (B, Dl, Hl, Wl, dhi.shape[1], dhi.shape[2], dhi.shape[3], C)
    nbytes = _lib.call_size('pw_upsample_trilinear_adjoint_workspace_bytes', *dims)
    ws = ops._workspace(nbytes, dhi.device)
    _lib.call('pw_upsample_trilinear_adjoint', ops._p(_cl(dhi, 'dhi')), ops._p(dlo), ops._p(ws), nbytes, *dims, ops._stream())
    return dlo


class UpsampleSumCL(torch.autograd.Function):
    """base + up(a) + up(b): the three per-level 1x1x1 conv outputs of LSSFPN3D summed at full resolution"""

    @staticmethod
    def forward(ctx, base, a, b):
        ctx.shapes = (tuple(a.shape), tuple(b.shape))
        # in place on `base` (a conv output nothing else reads or saved): no 82 MB clone
        out = base if base.is_contiguous() else base.contiguous()
        upsample_add(a.contiguous(), out, True)
        upsample_add(b.contiguous(), out, True)
        if out is base:
            ctx.mark_dirty(base)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        sa, sb = ctx.shapes
        return (dy if ctx.needs_input_grad[0] else None, upsample_adjoint(dy, sa) if ctx.needs_input_grad[1] else None,
                upsample_adjoint(dy, sb) if ctx.needs_input_grad[2] else None)


_ACT = {None: 0, 'none': 0, 'relu': 1, 'softplus': 2}


class BiasActCL(torch.autograd.Function):
    """act(x + bias) on channels-last rows, one kernel each way (pw_bias_act / pw_bias_act_backward: d x and the column sums for
    d bias from the same pass).  act: None | 'relu' | 'softplus' (torch.nn.Softplus defaults)."""

    @staticmethod
    def forward(ctx, x, bias, act):
        C = x.shape[-1]
        b = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        _lib.call('pw_bias_act', ops._p(_cl(x, 'x')), ops._p(b), x.numel() // C, C, _ACT[act], ops._p(y), ops._stream())
        ctx.save_for_backward(x, b if b is not None else torch.empty(0))
        ctx.act, ctx.has_bias = _ACT[act], b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b = ctx.saved_tensors
        C = x.shape[-1]
        dx = torch.empty_like(x)
        want_db = ctx.has_bias and ctx.needs_input_grad[1]
        db = torch.empty(C, device=x.device, dtype=_f32) if want_db else None
        nbytes = _lib.call_size('pw_bias_act_workspace_bytes', C)
        ws = ops._workspace(nbytes, x.device)
        _lib.call('pw_bias_act_backward', ops._p(x), ops._p(b) if ctx.has_bias else None, ops._p(_cl(dy.contiguous(), 'dy')), x.numel() // C, C,
                  ctx.act, ops._p(dx), ops._p(db), ops._p(ws), nbytes, ops._stream())
        return dx, db, None


def fpn_forward(neck, feats):
    """LSSFPN3D in training mode (lss_fpn.py:132-148).  The bias-free 1x1x1 conv commutes with the trilinear up-sampling, so
    each level is convolved at its own resolution with its slice of the 224-column weight and only 32-channel maps are
    up-sampled; BatchNorm (batch statistics) + ReLU follow on the sum -- the same function as upsample -> concat -> conv."""
    x8, x16, x32 = [f.contiguous() for f in feats]
    cm = neck.conv
    w = cm.conv.weight
    c8, c16 = x8.shape[-1], x16.shape[-1]
    y8 = Conv3dCL.apply(x8, w[:, :c8], 1)
    y16 = Conv3dCL.apply(x16, w[:, c8:c8 + c16], 1)
    y32 = Conv3dCL.apply(x32, w[:, c8 + c16:], 1)
    pre = UpsampleSumCL.apply(y8, y16, y32)
    return _bn_train(cm.bn, pre, None, cm.with_activation)


def conv_bias_act_forward(m, x):
    """ConvModule3d without norm (final_conv: conv + bias + ReLU, preworld.py:72-79) in training mode"""
    y = Conv3dCL.apply(x.contiguous(), m.conv.weight, m.stride)
    if m.conv.bias is None and not m.with_activation:
        return y
    return BiasActCL.apply(y, m.conv.bias, 'relu' if m.with_activation else None)


def _bn_cl(bn, x, relu):
    return _bn_train(bn, x.contiguous(), None, relu)


_LINEAR_ROWS = {(16, 8), (8, 18), (8, 1), (8, 16), (18, 8), (1, 8), (32, 16), (16, 32)}


def _linear_rows(x, w):
    """x (..., K) channels-last, w (N, K) -> (..., N) on pw_linear_rows"""
    K, N = w.shape[1], w.shape[0]
    y = torch.empty(tuple(x.shape[:-1]) + (N,), device=x.device, dtype=_f32)
    _lib.call('pw_linear_rows', ops._p(_cl(x, 'x')), ops._p(w.contiguous()), ops._p(y), x.numel() // K, K, N, ops._stream())
    return y


class LinearRowsCL(torch.autograd.Function):
    """1x1x1 Conv3d without bias on a channels-last (B,Z,Y,X,K) tensor with a handful of channels (OccHead's occ_pred_conv /
    voxel_soft_weights, occupancy_head.py:124-161): forward and data gradient on pw_linear_rows, weight gradient on
    pw_conv3d_wgrad (ksize 1).  Replaces torch.matmul, whose GEMM kernels took 1.0-1.5 ms per call at M = 640 000, K = 16, N = 8."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _linear_rows(x, w.detach())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _linear_rows(dy, w.detach().t().contiguous()) if ctx.needs_input_grad[0] else None
        dw = conv3d_wgrad(x, dy, (w.shape[0], w.shape[1], 1, 1, 1), 1).reshape(w.shape) if ctx.needs_input_grad[1] else None
        return dx, dw


def _conv1x1_cl(x, conv):
    """x (B,Z,Y,X,K) @ the (N,K,1,1,1) weight of a bias-free 1x1x1 conv"""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    if x.dim() == 5 and (w.shape[1], w.shape[0]) in _LINEAR_ROWS and (w.shape[0], w.shape[1]) in _LINEAR_ROWS:
        return LinearRowsCL.apply(x.contiguous(), w)
    return torch.matmul(x, w.t())


def linear_cl(x_cl, lin, act=None):
    """act(nn.Linear(x)) per voxel of a channels-last (B,Z,Y,X,K) tensor: the product as a 1x1x1 convolution on the MFMA conv kernels
    (forward, data and weight gradients: Conv3dCL; narrow outputs are padded to 32 columns inside its backward), bias and activation
    in one more pass (BiasActCL).  For K % 32 == 0 (the attribute MLPs 32 -> 64 -> {2, 17, 3} of preworld.py:101-110, which the
    library GEMM ran at 1.0-1.5 ms per call on 640 000 voxels); anything else goes to F.linear."""
    N, K = lin.weight.shape
    if x_cl.dim() != 5 or K % 32 or not x_cl.is_cuda:
        y = torch.nn.functional.linear(x_cl, lin.weight, lin.bias)
        return torch.nn.functional.softplus(y) if act == 'softplus' else (torch.relu(y) if act == 'relu' else y)
    y = Conv3dCL.apply(x_cl.contiguous(), lin.weight.reshape(N, K, 1, 1, 1), 1)
    return BiasActCL.apply(y, lin.bias, act) if (lin.bias is not None or act is not None) else y


def _plain_act(m):
    """'softplus' / 'relu' for an activation module BiasActCL reproduces exactly, else None"""
    if isinstance(m, torch.nn.Softplus) and m.beta == 1 and m.threshold == 20:
        return 'softplus'
    return 'relu' if isinstance(m, torch.nn.ReLU) else None


def mlp_cl(seq, x_cl):
    """an nn.Sequential of Linear / activation modules per voxel of a channels-last tensor: every Linear through linear_cl, a Softplus /
    ReLU that follows one is fused into its bias pass"""
    mods, i = list(seq), 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            act = _plain_act(mods[i + 1]) if i + 1 < len(mods) else None
            x_cl = linear_cl(x_cl, m, act)
            i += 2 if act is not None else 1
        else:
            x_cl = m(x_cl)
            i += 1
    return x_cl


def occ_head_forward(head, x_cl, transposed=True):
    """OccHead.forward_coarse_voxel (occupancy_head.py:124-161) in training mode on channels-last x (B,Z,Y,X,32) -> logits
    (B,Z,Y,X,18).  transposed: x is the encoder's native (Z,Y,X) buffer while the reference convolves (X,Y,Z): the taps are
    permuted instead of the activation.  3x3x3 conv 32 -> 16 on the MFMA kernels (Conv3dCL pads dY / W to 32 columns for the
    gradient kernels), batch-statistics BatchNorm on the HIP kernels; the per-voxel 16 -> 8 -> 18 layers (and the soft-weight
    branch) on pw_linear_rows (LinearRowsCL)."""
    c0, bn0 = head.occ_convs[0][0], head.occ_convs[0][1]
    c1, bn1, c2 = head.occ_pred_conv[0], head.occ_pred_conv[1], head.occ_pred_conv[3]
    w0 = c0.weight.permute(0, 1, 4, 3, 2) if transposed else c0.weight
    mid = _bn_cl(bn0, Conv3dCL.apply(x_cl.contiguous(), w0.contiguous(), 1), True)
    if head.soft_weights:
        # occupancy_head.py:141-151 with num_level = 1: softmax over ONE channel == 1, so the features are unchanged and the branch
        # receives exactly zero gradients -- but its BatchNorm sees the batch (running statistics move) and its parameters get
        # zero-valued .grad tensors (weight decay applies to them), so it is evaluated, not skipped
        s0, sbn, s3 = head.voxel_soft_weights[0], head.voxel_soft_weights[1], head.voxel_soft_weights[3]
        sw = _bn_cl(sbn, _conv1x1_cl(mid, s0), True)
        sw = torch.softmax(_conv1x1_cl(sw, s3), dim=-1)
        mid = mid * sw
    hid = _bn_cl(bn1, _conv1x1_cl(mid, c1), True)
    return _conv1x1_cl(hid, c2)


def downscale_forward(mod, v_cl):
    """DownScaleModule3DCustom (heads/occupancy_head.py:180-200) in training mode on channels-last v (B,Z,Y,X,C) -> (B, 4C): three
    unpadded 2x2x2 stride-2 convs with bias (taps permuted for the (Z,Y,X) buffer, like the inference path) and the global average."""
    x = v_cl.contiguous()
    for conv in (mod.downscale1, mod.downscale2, mod.downscale3):
        y = Conv3dCL.apply(x, conv.weight.permute(0, 1, 4, 3, 2).contiguous(), 2)
        x = BiasActCL.apply(y, conv.bias, None)
    return x.mean(dim=(1, 2, 3))


def fusion_step(fusion_head, v_cl, ego_feat):
    """One state-conditioned forecast step (preworld_temporal_traj.py:452-455) under autograd: v + W2 softplus(W1 [v, e] + b1) + b2
    with the ego half of W1 applied once per sample instead of on a repeat_interleave'd copy (same function; plain library GEMMs)."""
    l0, l2 = fusion_head[0], fusion_head[2]
    C = v_cl.shape[-1]
    hid = torch.nn.functional.linear(v_cl, l0.weight[:, :C]) + torch.nn.functional.linear(ego_feat, l0.weight[:, C:], l0.bias)[:, None, None, None, :]
    return v_cl + torch.nn.functional.linear(fusion_head[1](hid), l2.weight, l2.bias)
