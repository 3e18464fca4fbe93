"""Functional wrappers over the C ABI for torch tensors resident in HBM.

torch is plumbing here: device memory, the current HIP stream and autograd glue.
All arithmetic happens in libpreworld_hip.so.  Names and argument order follow the
reference operators these replace (cited per function, paths relative to the reference).
"""
import ctypes
import os

import torch

from . import _lib

_i32 = torch.int32
_f32 = torch.float32
RNG_ROW = 1056       # PW_RNG_ROW of include/preworld_hip.h: int32 per range slot


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PreworldHipError('%s must be a CUDA(HIP) tensor' % name)
    if t.dtype != dtype:
        raise _lib.PreworldHipError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.PreworldHipError('%s must be contiguous' % name)
    return _ptr(t)


class _Ptr(ctypes.c_void_p):
    """A device-pointer argument that keeps its tensor alive until the argument list is dropped, i.e. until the kernel has been
    enqueued.  Round 6: `_chk(x.contiguous(), ...)` on a NON-contiguous x used to pass the address of a temporary that was freed
    before the call -- the caching allocator handed the same block to the next temporary of the same call and its copy overwrote the
    first (four camera tensors of a B = 2 batch aliasing one buffer: found by the reference-class B = 2 fixtures).  After the
    enqueue a reuse is stream-ordered behind the kernel and safe."""


def _ptr(t):
    p = _Ptr(t.data_ptr())
    p._keep = t
    return p


def _p(t):
    return _ptr(t) if t is not None else None


def _host3(vals):
    return (ctypes.c_float * 3)(*[float(v) for v in vals])


def _workspace(nbytes, device):
    # torch's caching allocator returns >=512-byte aligned blocks
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)


def copy_many(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of same-shaped contiguous device tensors in ONE launch (pw_copy_many) -- a sample's inputs going
    into the static buffers of a captured step are 15 small copies, mostly launch latency as separate kernels"""
    n = len(dsts)
    if n != len(srcs):
        raise _lib.PreworldHipError('copy_many: %d destinations, %d sources' % (n, len(srcs)))
    for d, s_ in zip(dsts, srcs):
        if d.shape != s_.shape or d.dtype != s_.dtype or not (d.is_contiguous() and s_.is_contiguous() and d.is_cuda and s_.is_cuda):
            raise _lib.PreworldHipError('copy_many: tensors must be contiguous device tensors of equal shape and dtype')
    if n == 0:
        return
    S = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
    D = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    B = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d in dsts])
    _lib.call('pw_copy_many', S, D, B, n, _stream())


def device_info():
    cu = ctypes.c_int(0)
    lds = ctypes.c_int(0)
    name = ctypes.create_string_buffer(64)
    _lib.call('pw_device_info', ctypes.byref(cu), ctypes.byref(lds), name, 64)
    return dict(cu_count=cu.value, lds_bytes_per_cu=lds.value, arch=name.value.decode())


# ------------------------------------------------------------------------------ LSS geometry
def lss_camera_matrices(sensor2ego, cam2imgs, post_rots):
    """inverse(post_rots), sensor2ego[:3,:3] @ inverse(cam2imgs), sensor2ego[:3,3]
    (mmdet3d/models/necks/view_transformer.py:141-150).  Inputs (B,N,4,4),(B,N,3,3),(B,N,3,3)."""
    B, N = sensor2ego.shape[:2]
    s = sensor2ego.contiguous().float()
    k = cam2imgs.contiguous().float()
    r = post_rots.contiguous().float()
    dev = s.device
    ipr = torch.empty(B, N, 3, 3, device=dev, dtype=_f32)
    comb = torch.empty(B, N, 3, 3, device=dev, dtype=_f32)
    tr = torch.empty(B, N, 3, device=dev, dtype=_f32)
    _lib.call('pw_lss_camera_matrices', B * N, _chk(s, _f32, 'sensor2ego'), _chk(k, _f32, 'cam2imgs'),
              _chk(r, _f32, 'post_rots'), _p(ipr), _p(comb), _p(tr), _stream())
    return ipr, comb, tr


def lss_voxel_index(frustum, inv_post_rot, post_trans, combine, trans, bda, lower, interval,
                    grid_size, B, N, return_coor=False):
    """get_lidar_coor + the voxel-id half of voxel_pooling_prepare_v2
    (view_transformer.py:114-153, :226-245).  Returns int32 voxel id per frustum point (-1 =
    outside) and, optionally, the reference's coor tensor (B,N,D,H,W,3)."""
    D, H, W, _ = frustum.shape
    dev = frustum.device
    vox = torch.empty(B * N * D * H * W, device=dev, dtype=_i32)
    coor = torch.empty(B, N, D, H, W, 3, device=dev, dtype=_f32) if return_coor else None
    _lib.call('pw_lss_voxel_index', B, N, D, H, W, _chk(frustum, _f32, 'frustum'),
              _chk(inv_post_rot, _f32, 'inv_post_rot'), _chk(post_trans, _f32, 'post_trans'),
              _chk(combine, _f32, 'combine'), _chk(trans, _f32, 'trans'), _chk(bda, _f32, 'bda'),
              _host3(lower), _host3(interval), int(grid_size[0]), int(grid_size[1]),
              int(grid_size[2]), _p(vox), _p(coor), _stream())
    return (vox, coor) if return_coor else vox


LONG_SEGMENT = 64   # voxels holding more points than this are pooled by whole waves


class VoxelSort:
    """Result of the device counting sort over voxel ids (all tensors stay on the GPU)."""
    __slots__ = ('seg_start', 'order', 'order_feat', 'long_list', 'n_long', 'n_keys')

    def __init__(self, seg_start, order, order_feat, long_list, n_long, n_keys):
        self.seg_start, self.order, self.order_feat = seg_start, order, order_feat
        self.long_list, self.n_long, self.n_keys = long_list, n_long, n_keys


def segment_sort(keys, n_keys, aux_div=0, aux_mod=0, long_threshold=0):
    """Stable counting sort of indices by int32 key (key<0 dropped).
    Returns VoxelSort(seg_start int32[n_keys+1], order int32[n], order_feat, long_list, n_long)."""
    n = keys.numel()
    dev = keys.device
    nbytes = _lib.call_size('pw_segment_sort_workspace_bytes', n, n_keys)
    ws = _workspace(nbytes, dev)
    seg_start = torch.empty(n_keys + 1, device=dev, dtype=_i32)
    order = torch.empty(n, device=dev, dtype=_i32)
    aux = torch.empty(n, device=dev, dtype=_i32) if aux_div else None
    ll = torch.empty(n // (long_threshold + 1) + 1, device=dev, dtype=_i32) if long_threshold else None
    nl = torch.empty(1, device=dev, dtype=_i32) if long_threshold else None
    _lib.call('pw_segment_sort', n, n_keys, _chk(keys, _i32, 'keys'), _p(ws), nbytes, _p(seg_start),
              _p(order), aux_div, aux_mod, _p(aux), long_threshold, _p(ll), _p(nl), _stream())
    return VoxelSort(seg_start, order, aux, ll, nl, n_keys)


def lss_ranks(seg_start, order, n_voxels, D, HW):
    """Expand a voxel sort into the reference's (ranks_bev, ranks_depth, ranks_feat,
    interval_starts, interval_lengths) -- view_transformer.py:246-261.  One D2H sync (the
    reference's boolean-mask indexing syncs several times)."""
    dev = seg_start.device
    n = order.numel()
    nbytes = _lib.call_size('pw_lss_ranks_workspace_bytes', n_voxels)
    ws = _workspace(nbytes, dev)
    rb = torch.empty(n, device=dev, dtype=_i32)
    rd = torch.empty(n, device=dev, dtype=_i32)
    rf = torch.empty(n, device=dev, dtype=_i32)
    nmax = min(n, n_voxels)
    st = torch.empty(nmax, device=dev, dtype=_i32)
    ln = torch.empty(nmax, device=dev, dtype=_i32)
    counts = torch.empty(2, device=dev, dtype=_i32)
    _lib.call('pw_lss_ranks', n_voxels, _p(seg_start), _p(order), D, HW, _p(ws), nbytes, _p(rb),
              _p(rd), _p(rf), _p(st), _p(ln), _p(counts), _stream())
    kept, ni = counts.tolist()
    if kept == 0:
        return None, None, None, None, None
    return rb[:kept], rd[:kept], rf[:kept], st[:ni], ln[:ni]


def bev_pool_dense(depth, feat, vs, out=None, out_h2=False):
    """Write-once dense pooling.  depth (B,N,D,H,W) flat, feat (B,N,H,W,C), vs = VoxelSort built
    with aux_div=D*H*W, aux_mod=H*W  ->  (n_voxels, C) fp32, or (out_h2=True) the same sums as an ops.H2 under a new
    range slot (or under `out`'s when an ops.H2 is passed)."""
    C = feat.shape[-1]
    if vs.order_feat is None:
        raise _lib.PreworldHipError('bev_pool_dense needs a VoxelSort built with aux_div/aux_mod')
    if out is None:
        out = torch.empty(vs.n_keys, C, device=feat.device, dtype=_f32)
    orng = None
    if out_h2:
        out, orng = _out_h2(out, feat.device)
    _lib.call('pw_bev_pool_dense', _chk(depth, _f32, 'depth'), _chk(feat, _f32, 'feat'),
              _chk(vs.seg_start, _i32, 'seg_start'), _chk(vs.order, _i32, 'order'),
              _chk(vs.order_feat, _i32, 'order_feat'), vs.n_keys, C,
              LONG_SEGMENT if vs.long_list is not None else 0, _p(vs.long_list), _p(vs.n_long),
              _chk(out, _f32, 'out'), int(bool(out_h2)), _p(orng), _stream())
    return H2(out, orng) if out_h2 else out


def lss_lift_pool(frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda, lower, interval, grid_size, depth, feat,
                  out=None, out_h2=False):
    """get_lidar_coor + voxel_pooling_prepare_v2 + bev_pool_v2 forward of one batch of frames in one call (inference, C == 32;
    view_transformer.py:114-153, :203-261, bev_pool_cuda.cu:21-48): the same bits as lss_camera_matrices -> lss_voxel_index ->
    segment_sort -> bev_pool_dense in 5 launches.  depth (B,N,D,H,W), feat (B,N,H,W,C) fp32; returns (n_voxels, C) fp32 or an
    ops.H2 (out_h2).  (A voxel-driven gather form was built and measured in round 5 -- same bits, 1.2 x instead of 2.0 x the
    algorithmic HBM bytes, but instruction-bound at 2-3 x the time: profiles/r05_lss_gather_experiment.md, commit 69d8f73.)"""
    B, N = sensor2ego.shape[:2]
    D, H, W, _ = frustum.shape
    C = feat.shape[-1]
    dev = feat.device
    n_vox = B * int(grid_size[0]) * int(grid_size[1]) * int(grid_size[2])
    if out is None:
        out = torch.empty(n_vox, C, device=dev, dtype=_f32)
    orng = None
    if out_h2:
        out, orng = _out_h2(out, dev)
    nbytes = _lib.call_size('pw_lss_lift_pool_workspace_bytes', B * N * D * H * W, n_vox, B * N)
    ws = _workspace(nbytes, dev)
    f = lambda t: t.contiguous().float()
    _lib.call('pw_lss_lift_pool', B, N, D, H, W, _chk(frustum, _f32, 'frustum'), _chk(f(sensor2ego), _f32, 'sensor2ego'),
              _chk(f(cam2imgs), _f32, 'cam2imgs'), _chk(f(post_rots), _f32, 'post_rots'), _chk(f(post_trans), _f32, 'post_trans'),
              _chk(f(bda), _f32, 'bda'), _host3(lower), _host3(interval), int(grid_size[0]), int(grid_size[1]),
              int(grid_size[2]), _chk(depth, _f32, 'depth'), _chk(feat, _f32, 'feat'), C, _p(ws), nbytes,
              _chk(out, _f32, 'out'), int(bool(out_h2)), _p(orng), _stream())
    return H2(out, orng) if out_h2 else out


# ------------------------------------------------------------------------------ bev_pool_v2
def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                        interval_starts):
    """Same call as bev_pool_v2_ext.bev_pool_v2_forward
    (mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:30-57): out (B,Z,Y,X,C) pre-zeroed, in/out;
    interval_lengths BEFORE interval_starts."""
    _lib.call('pw_bev_pool_v2_forward', _chk(depth, _f32, 'depth'), _chk(feat, _f32, 'feat'),
              _chk(out, _f32, 'out'), _chk(ranks_depth, _i32, 'ranks_depth'),
              _chk(ranks_feat, _i32, 'ranks_feat'), _chk(ranks_bev, _i32, 'ranks_bev'),
              _chk(interval_lengths, _i32, 'interval_lengths'),
              _chk(interval_starts, _i32, 'interval_starts'), feat.shape[-1],
              interval_lengths.numel(), _stream())


def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                         ranks_bev, interval_lengths, interval_starts):
    """Same call as bev_pool_v2_ext.bev_pool_v2_backward (bev_pool.cpp:74-104)."""
    _lib.call('pw_bev_pool_v2_backward', _chk(out_grad, _f32, 'out_grad'),
              _chk(depth_grad, _f32, 'depth_grad'), _chk(feat_grad, _f32, 'feat_grad'),
              _chk(depth, _f32, 'depth'), _chk(feat, _f32, 'feat'),
              _chk(ranks_depth, _i32, 'ranks_depth'), _chk(ranks_feat, _i32, 'ranks_feat'),
              _chk(ranks_bev, _i32, 'ranks_bev'), _chk(interval_lengths, _i32, 'interval_lengths'),
              _chk(interval_starts, _i32, 'interval_starts'), out_grad.shape[-1],
              interval_lengths.numel(), _stream())


class QuickCumsumCuda(torch.autograd.Function):
    """Drop-in for mmdet3d/ops/bev_pool_v2/bev_pool.py:11-83 (same name, same signature)."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
        ranks_bev = ranks_bev.int()
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        ranks_depth = ranks_depth.contiguous().int()
        ranks_feat = ranks_feat.contiguous().int()
        interval_lengths = interval_lengths.contiguous().int()
        interval_starts = interval_starts.contiguous().int()
        out = feat.new_zeros(bev_feat_shape)
        bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                            interval_starts)
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        n_pix = feat.numel() // feat.shape[-1]
        # re-sort by feat pixel (bev_pool.py:47-57) with the same device counting sort
        vs = segment_sort(ranks_feat.contiguous(), n_pix)
        seg_start = vs.seg_start
        order = vs.order[:ranks_feat.numel()].long()
        rf, rd, rb = ranks_feat[order].contiguous(), ranks_depth[order].contiguous(), \
            ranks_bev[order].contiguous()
        lens = seg_start[1:] - seg_start[:-1]
        nz = lens > 0
        interval_starts_bp = seg_start[:-1][nz].contiguous()
        interval_lengths_bp = lens[nz].contiguous()
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        bev_pool_v2_backward(out_grad.contiguous(), depth_grad, feat_grad, depth, feat, rd, rf, rb,
                             interval_lengths_bp, interval_starts_bp)
        return depth_grad, feat_grad, None, None, None, None, None, None


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
    """mmdet3d/ops/bev_pool_v2/bev_pool.py:86-92.  Returns (B,C,Z,Y,X); unlike the reference
    the permute is a stride view of the channels-last buffer (no 82 MB copy)."""
    x = QuickCumsumCuda.apply(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths)
    return x.permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------ conv3d
def pack_conv_weight(w, cout_total=None):
    """torch Conv3d weight (Cout, Cin, k, k, k) -> the MFMA operand order documented in
    include/preworld_hip.h: float[Cin/32][k^3][cout_total/32][4 pieces][64 lanes][4] with
    wpk[ch][tap][nt][q][h*32+j][e] = w[nt*32+j][ch*32+h*16+4*q+e][tap]; columns >= Cout are zero."""
    Cout, Cin, k = w.shape[0], w.shape[1], w.shape[2]
    if Cin % 32:
        raise _lib.PreworldHipError('Cin must be a multiple of 32, got %d' % Cin)
    if cout_total is None:
        cout_total = (Cout + 31) // 32 * 32
    taps = k ** 3
    wp = w.new_zeros(cout_total, Cin, taps)
    wp[:Cout] = w.reshape(Cout, Cin, taps)
    nt, nch = cout_total // 32, Cin // 32
    wp = wp.view(nt, 32, nch, 2, 4, 4, taps)            # (nt, j, ch, h, q, e, tap): s = 4 q + e
    wp = wp.permute(2, 6, 0, 4, 3, 1, 5).contiguous()   # (ch, tap, nt, q, h, j, e): piece-major, a piece = 64 lanes x 16 B
    return wp.view(nch, taps, nt, 64, 16).float().contiguous()


def pack_conv_weights_concat(ws):
    """Concatenate several convs over the SAME input along packed output columns, each padded
    to a multiple of 32 columns (conv1 + downsample of a BasicBlock3D share one pass)."""
    return torch.cat([pack_conv_weight(w) for w in ws], dim=2).contiguous()


def fold_bn(bn_weight, bn_bias, running_mean, running_var, eps=1e-5, conv_bias=None):
    """BatchNorm(eval) -> per-channel (scale, bias): y = conv*scale + bias."""
    inv = 1.0 / torch.sqrt(running_var.float() + eps)
    scale = bn_weight.float() * inv if bn_weight is not None else inv
    bias = (bn_bias.float() if bn_bias is not None else 0) - running_mean.float() * scale
    if conv_bias is not None:
        bias = bias + conv_bias.float() * scale
    return scale.contiguous(), bias.contiguous()


def _pad32(v, fill):
    n = v.numel()
    n32 = (n + 31) // 32 * 32
    if n32 == n:
        return v.contiguous()
    out = v.new_full((n32,), fill)
    out[:n] = v
    return out


def _row_stride(t, shape, name):
    """t must be (B,Do,Ho,Wo,c) laid out as a channel slice of a dense channels-last buffer:
    strides (Do*Ho*Wo*ld, Ho*Wo*ld, Wo*ld, ld, 1).  Returns ld."""
    if not t.is_cuda or t.dtype != _f32 or tuple(t.shape) != tuple(shape):
        raise _lib.PreworldHipError('%s must be a float32 device tensor of shape %s' % (name, tuple(shape)))
    B, Do, Ho, Wo, c = shape
    ld, below = c, 1                      # the stride of a size-1 axis is arbitrary: take ld from the innermost axis > 1
    for i in (3, 2, 1, 0):
        if shape[i] > 1:
            ld = t.stride(i) // below if t.stride(i) % below == 0 else -1
            break
        below *= shape[i]
    want = (Do * Ho * Wo * ld, Ho * Wo * ld, Wo * ld, ld, 1)
    ok = ld >= c and all(t.stride(i) == want[i] or t.shape[i] == 1 for i in range(5))
    if not ok:
        raise _lib.PreworldHipError('%s must be dense or a channel slice of a dense channels-last buffer' % name)
    return ld


def conv3d_ndhwc(x, wpk, scale=None, bias=None, residual=None, cout0=None, cout1=0, ksize=3,
                 stride=1, relu0=False, relu1=False, algo=0, out0=None, out1=None):
    """x (B,D,H,W,Cin) channels-last -> y0 (B,Do,Ho,Wo,cout0) [, y1 (B,Do,Ho,Wo,cout1)].
    scale/bias: per packed column (length cout_total, see include/preworld_hip.h).
    out0 / out1 may be channel slices of a wider channels-last buffer (e.g. buf[..., 32:64]); a
    residual then has to share out0's layout (it may BE out0: each element is read, then written)."""
    B, D, H, W, Cin = x.shape
    nch, taps, nt = wpk.shape[:3]
    cout_total = nt * 32
    if cout0 is None:
        cout0 = cout_total
    pad = (ksize - 1) // 2
    Do = (D + 2 * pad - ksize) // stride + 1
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    y0 = out0 if out0 is not None else torch.empty(B, Do, Ho, Wo, cout0, device=x.device, dtype=_f32)
    ld0 = _row_stride(y0, (B, Do, Ho, Wo, cout0), 'y0')
    y1, ld1 = None, 0
    if cout1:
        y1 = out1 if out1 is not None else torch.empty(B, Do, Ho, Wo, cout1, device=x.device, dtype=_f32)
        ld1 = _row_stride(y1, (B, Do, Ho, Wo, cout1), 'y1')
    if residual is not None and _row_stride(residual, (B, Do, Ho, Wo, cout0), 'residual') != ld0:
        raise _lib.PreworldHipError('residual must have the same row stride as y0')
    if scale is not None and scale.numel() != cout_total:
        raise _lib.PreworldHipError('scale must have cout_total=%d entries' % cout_total)
    if bias is not None and bias.numel() != cout_total:
        raise _lib.PreworldHipError('bias must have cout_total=%d entries' % cout_total)
    if nch * 32 != Cin or taps != ksize ** 3:
        raise _lib.PreworldHipError('packed weight does not match Cin/ksize')
    _lib.call('pw_conv3d_ndhwc', _chk(x, _f32, 'x'), _chk(wpk, _f32, 'wpk'), _p(scale), _p(bias),
              _p(residual), _p(y0), _p(y1), B, D, H, W, Cin, cout_total, cout0, cout1, ld0, ld1,
              ksize, stride, int(relu0), int(relu1), algo, _stream())
    return (y0, y1) if cout1 else y0


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def pack_conv_weight_wino(w, cout_total=None):
    """torch Conv3d weight (Cout, Cin, 3, 3, 3) -> Winograd F(2,3)^3 transform-domain weights in the
    operand order of pw_conv3d_wino: float[Cin/32][64][cout_total/16][64][8] (include/preworld_hip.h).
    The transform U = G w G^T along d, h, w is done in float64 and rounded once."""
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 32:
        raise _lib.PreworldHipError('pack_conv_weight_wino expects (Cout, 32k, 3, 3, 3)')
    if cout_total is None:
        cout_total = (Cout + 31) // 32 * 32
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    U = torch.einsum('ia,jb,kc,oeabc->oeijk', G, G, G, w.double())           # (Cout, Cin, 4, 4, 4)
    Up = U.new_zeros(cout_total, Cin, 64)
    Up[:Cout] = U.reshape(Cout, Cin, 64)
    n16, nch = cout_total // 16, Cin // 32
    Up = Up.view(n16, 16, nch, 4, 8, 64)                   # (n16, j, ch, g, s, p)
    Up = Up.permute(2, 5, 0, 3, 1, 4).contiguous()         # (ch, p, n16, g, j, s)
    return Up.view(nch, 64, n16, 64, 8).float().contiguous()


def pack_conv_weights_wino_concat(ws):
    """conv1 + downsample of a BasicBlock3D over the same input: each padded to a multiple of 32 columns."""
    return torch.cat([pack_conv_weight_wino(w) for w in ws], dim=2).contiguous()


def conv3d_wino(x, uwpk, scale=None, bias=None, residual=None, cout0=None, cout1=0, relu0=False, relu1=False,
                out0=None, out1=None):
    """3x3x3 stride-1 pad-1 conv by Winograd F(2x2x2,3x3x3) (pw_conv3d_wino): same contract as
    conv3d_ndhwc(ksize=3, stride=1) with weights from pack_conv_weight_wino."""
    B, D, H, W, Cin = x.shape
    nch, npts, n16 = uwpk.shape[:3]
    cout_total = n16 * 16
    if npts != 64 or nch * 32 != Cin or cout_total % 32:
        raise _lib.PreworldHipError('packed Winograd weight does not match the input')
    if cout0 is None:
        cout0 = cout_total
    y0 = out0 if out0 is not None else torch.empty(B, D, H, W, cout0, device=x.device, dtype=_f32)
    ld0 = _row_stride(y0, (B, D, H, W, cout0), 'y0')
    y1, ld1 = None, 0
    if cout1:
        y1 = out1 if out1 is not None else torch.empty(B, D, H, W, cout1, device=x.device, dtype=_f32)
        ld1 = _row_stride(y1, (B, D, H, W, cout1), 'y1')
    if residual is not None and _row_stride(residual, (B, D, H, W, cout0), 'residual') != ld0:
        raise _lib.PreworldHipError('residual must have the same row stride as y0')
    if scale is not None and scale.numel() != cout_total:
        raise _lib.PreworldHipError('scale must have cout_total=%d entries' % cout_total)
    if bias is not None and bias.numel() != cout_total:
        raise _lib.PreworldHipError('bias must have cout_total=%d entries' % cout_total)
    _lib.call('pw_conv3d_wino', _chk(x, _f32, 'x'), _chk(uwpk, _f32, 'uwpk'), _p(scale), _p(bias), _p(residual),
              _p(y0), _p(y1), B, D, H, W, Cin, cout_total, cout0, cout1, ld0, ld1, int(relu0), int(relu1), _stream())
    return (y0, y1) if cout1 else y0


# ------------------------------------------------------------------------------ split-fp16 ("h2") path
class RangeCtx:
    """Activation ranges of the split-fp16 path (include/preworld_hip.h "RANGE SLOTS").

    fp16 has 5 exponent bits, the reference's fp32 activations 8 (backbones/resnet.py:88-123 is plain Conv3d): an h2 tensor
    stores value / 2^e with a per-tensor exponent e kept in a device-resident *range slot* (int32[2] = [e, bit pattern of the
    largest |value| written]).  A RangeCtx owns a table of slots.  Producers take the next slot in CALL ORDER
    (`new_slot`), so the i-th h2 tensor of a deterministic forward pass always maps to slot i and its exponent persists from
    one pass to the next; kernels read the exponents from the table itself, so a captured hipGraph follows later updates.

    `ranged(fn)` is the calibration loop: run the pass, read the recorded maxima back (one small D2H copy), move every
    exponent whose tensor left the comfortable window, and repeat until nothing moves -- typically two passes on a cold
    table, one afterwards.  `check()` is the cheap steady-state test (hard window) a graph replay is followed by."""
    TARGET = 12          # ideal exponent puts the largest magnitude in [2^12, 2^13) stored units
    SETTLE_LO, SETTLE_HI = 8, 15      # calibration accepts a stored maximum in [2^8, 2^15)
    HARD_LO, HARD_HI = 6, 65504.0     # steady state accepts [2^6, 65504]

    def __init__(self, device, n_slots=128):
        # a slot = RNG_ROW int32: [exponent, folded maximum, ..., 1 024 partial maxima] (include/preworld_hip.h "RANGE SLOTS")
        self.tab = torch.zeros(n_slots, RNG_ROW, dtype=_i32, device=device)
        self.compact = torch.zeros(n_slots, 2, dtype=_i32, device=device)      # [exponent, maximum] pairs, written by fold()
        self.sticky = torch.zeros(4, dtype=_i32, device=device)                 # audit(): [bad slots, bad passes, passes audited, -]
        self.n = 0
        self.device = torch.device(device)

    def begin(self):
        """start of a pass: slots are handed out from 0 again, recorded maxima are cleared (a kernel, capturable)"""
        self.n = 0
        self.tab[:, 1].zero_()

    def fold(self):
        """end of a pass: fold the partial maxima the kernels raised into the slots and refresh `compact` (one small launch)"""
        _lib.call('pw_rng_fold', _p(self.tab), self.tab.shape[0], _p(self.compact), _stream())

    def audit(self):
        """after fold(): the hard-window test of check() on the device, accumulated in `sticky` (pw_rng_audit) -- capturable, so
        every replay of a graph is tested without the host waiting for it; read the verdict with audited()"""
        _lib.call('pw_rng_audit', _p(self.compact), self.compact.shape[0], _p(self.sticky), _stream())

    def audited(self):
        """(passes that left their calibrated ranges, passes audited) since the table was created (synchronises)"""
        bad_slots, bad_passes, passes, _ = self.sticky.tolist()
        return bad_passes, passes

    def new_slot(self):
        if self.n >= self.tab.shape[0]:
            raise _lib.PreworldHipError('RangeCtx: more than %d h2 tensors in one pass' % self.tab.shape[0])
        self.n += 1
        return self.tab[self.n - 1]

    @staticmethod
    def ideal_exp(amax):
        """the exponent k_f32_to_h2 / rng_ideal_exp (pw_h2.h) derive from a maximum: largest magnitude in [2^12, 2^13)"""
        import math
        if not (amax > 0.0) or math.isinf(amax) or math.isnan(amax):
            return 0
        return max(-100, min(100, math.frexp(amax)[1] - 1 - RangeCtx.TARGET))

    def _read(self, host=None):
        import numpy as np
        t = (self.compact.cpu() if host is None else host).numpy()
        return t[:, 0].astype(np.int64), t[:, 1].astype(np.int32).view(np.float32).astype(np.float64)

    def check(self, host=None):
        """indices of the slots whose recorded maximum left the hard window under the exponent it was written with, as of the
        last fold() (host: an already copied `compact` table, e.g. the one a graph replay delivers)"""
        import numpy as np
        e, amax = self._read(host)
        with np.errstate(over='ignore', invalid='ignore'):
            stored = amax * np.exp2(-e.astype(np.float64))
        bad = (amax != 0) & ~((stored >= 2.0 ** self.HARD_LO) & (stored <= self.HARD_HI))
        return np.nonzero(bad)[0].tolist()

    def settle(self):
        """after a pass: move the exponents of tensors outside the calibration window; True when nothing had to move"""
        import numpy as np
        self.fold()
        e, amax = self._read()
        new = e.copy()
        for i in range(len(e)):
            a = float(amax[i])
            if a == 0.0:
                continue
            if not np.isfinite(a):                    # overflowed under this exponent (or fed by a tensor that did)
                new[i] = min(int(e[i]) + 12, 100)
                continue
            ideal = self.ideal_exp(a)
            if not (ideal + self.TARGET - self.SETTLE_HI < e[i] <= ideal + self.TARGET - self.SETTLE_LO):
                new[i] = ideal
        if (new == e).all():
            return True
        self.tab[:len(e), 0] = torch.from_numpy(new.astype(np.int32)).to(self.device)
        return False


_range_stack = []


class use_range:
    """`with ops.use_range(ctx):` h2 producers called inside take their slots from ctx"""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        _range_stack.append(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        _range_stack.pop()


def current_range():
    return _range_stack[-1] if _range_stack else None


def new_slot(device):
    """range slot for a new h2 tensor: the active RangeCtx's next one, else a private slot with exponent 0 (the stored units
    are the values themselves -- fine for O(1) data; wrap the computation in ops.ranged() for anything else)"""
    ctx = current_range()
    return ctx.new_slot() if ctx is not None else torch.zeros(RNG_ROW, dtype=_i32, device=device)


def slot_state(slot):
    """(exponent, largest recorded |value|) of one range slot as host numbers (folds its partial maxima first; syncs)"""
    _lib.call('pw_rng_fold', _p(slot), 1, None, _stream())
    e, bits = slot[:2].tolist()
    import struct
    return e, struct.unpack('f', struct.pack('i', bits))[0]


def ranged(fn, ctx, max_iter=12, agree=None):
    """Run fn() under ctx until its activation ranges have settled (RangeCtx.settle) and return the last result.  Under
    stream capture fn() runs once with the exponents as they are (calibrate before capturing; `check()` after replays).
    agree(ok) -> bool: multi-process passes (fn contains collectives) pass a function that ANDs `ok` over the ranks."""
    with use_range(ctx):
        for _ in range(max_iter):
            ctx.begin()
            out = fn()
            if torch.cuda.is_current_stream_capturing():
                ctx.fold()
                return out
            ok = ctx.settle()
            if agree(ok) if agree is not None else ok:
                return out
    raise _lib.PreworldHipError('activation ranges did not settle in %d passes (non-finite inputs or weights?): slots %s'
                                % (max_iter, ctx.check()))


class H2:
    """A channels-last activation tensor in h2 storage (include/preworld_hip.h): `.buf` is a float32-typed torch
    tensor of the logical shape (.., C), C % 32 == 0, whose bytes are the split-fp16 encoding of value / 2^e; `.rng` is the
    range slot (int32[2] device tensor: [e, recorded maximum]) the tensor was written under, or None for e = 0.  Channel
    slices at multiples of 32 and leading-axis slices of `.buf` are again valid h2 tensors under the same slot."""
    __slots__ = ('buf', 'rng')

    def __init__(self, buf, rng=None):
        if buf.dtype != _f32 or buf.shape[-1] % 32:
            raise _lib.PreworldHipError('h2 tensors are float32-typed with a multiple of 32 channels')
        if rng is not None and (rng.dtype != _i32 or rng.numel() != RNG_ROW or not rng.is_contiguous() or rng.device != buf.device):
            raise _lib.PreworldHipError('h2 range slot must be a contiguous int32[%d] tensor on the same device' % RNG_ROW)
        self.buf = buf
        self.rng = rng

    @property
    def shape(self):
        return self.buf.shape

    @property
    def device(self):
        return self.buf.device

    def __getitem__(self, idx):
        return H2(self.buf[idx], self.rng)

    def view(self, *shape):
        return H2(self.buf.view(*shape), self.rng)


def _rng(h):
    """ctypes pointer of an H2's range slot (NULL without one)"""
    return _p(h.rng) if isinstance(h, H2) and h.rng is not None else None


def _out_h2(o, device):
    """(buffer, slot) of an h2 destination: an ops.H2 keeps its slot, a raw buffer / a new tensor gets a new one"""
    if isinstance(o, H2):
        return o.buf, (o.rng if o.rng is not None else None)
    return o, new_slot(device)


def _rows(t, name):
    """(n_vox, C, ld) of a channels-last tensor that is dense or a channel slice of a dense buffer."""
    C = t.shape[-1]
    n = t.numel() // C
    if t.is_contiguous():
        return n, C, C
    ld, expect = None, None
    for i in range(t.dim() - 2, -1, -1):                 # innermost voxel axis of extent > 1 defines the row stride
        if t.shape[i] == 1:
            continue
        if ld is None:
            ld, expect = t.stride(i), t.stride(i) * t.shape[i]
        elif t.stride(i) != expect:
            ld = -1
            break
        else:
            expect *= t.shape[i]
    if ld is None:
        ld = C
    if t.stride(-1) != 1 or ld < C:
        raise _lib.PreworldHipError('%s must be dense or a channel slice of a dense channels-last buffer' % name)
    return n, C, ld


def f32_to_h2(x, out=None):
    """fp32 channels-last (.., C) -> H2 of the same shape (pw_f32_to_h2).  out: None (new tensor under a private slot whose
    exponent is derived from max|x| on the device: exact for any scale), an ops.H2 (written under ITS slot's exponent) or a
    raw float32 buffer (same, new private slot)."""
    n, C, ldx = _rows(x, 'x')
    if isinstance(out, H2) and out.rng is not None:
        buf, slot, auto = out.buf, out.rng, 0
    else:
        buf = out.buf if isinstance(out, H2) else out
        if buf is None:
            buf = torch.empty(x.shape, device=x.device, dtype=_f32)
        slot, auto = torch.zeros(RNG_ROW, dtype=_i32, device=x.device), 1
    _, _, ldy = _rows(buf, 'out')
    _lib.call('pw_f32_to_h2', _p(x), _p(buf), n, C, ldx, ldy, _p(slot), auto, _stream())
    return H2(buf, slot)


def h2_to_f32(x, out=None):
    """H2 -> fp32 channels-last tensor of the same shape (pw_h2_to_f32): (hi + lo) * 2^e, exact."""
    n, C, ldx = _rows(x.buf, 'x')
    if out is None:
        out = torch.empty(x.shape, device=x.buf.device, dtype=_f32)
    _, _, ldy = _rows(out, 'out')
    _lib.call('pw_h2_to_f32', _p(x.buf), _p(out), n, C, ldx, ldy, _rng(x), _stream())
    return out


def pack_conv_weight_h2(w, cout_total=None):
    """torch Conv3d weight (Cout, Cin, 3,3,3) -> (wpk, inv_scale): split-fp16 weights in the operand order of
    pw_conv3d_h2 (include/preworld_hip.h) as a float32-typed tensor [Cin/32][27][cout_total/32][4 pieces][64 lanes][4], and the
    per-column factor (cout_total,) = 1 / S[n] to multiply into the epilogue scale.  S[n] = 2^k puts the largest
    |w[n]| in [512, 1024): both halves of the split are then normal fp16 numbers for every weight above 2^-13 of it."""
    Cout, Cin = w.shape[:2]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    if Cin % 32:
        raise _lib.PreworldHipError('pack_conv_weight_h2 expects Cin % 32 == 0')
    if cout_total is None:
        cout_total = (Cout + 31) // 32 * 32
    wf = w.reshape(Cout, -1).double()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    S = torch.exp2(torch.floor(torch.log2(1023.0 / amax)))                       # power of two per output channel
    ws = torch.zeros(cout_total, Cin, taps, dtype=torch.float64, device=w.device)
    ws[:Cout] = (wf * S[:, None]).view(Cout, Cin, taps)
    hi = ws.to(torch.float16)
    lo = (ws - hi.double()).to(torch.float16)
    planes = torch.stack([hi, lo], 0)                                            # (p, n, c, tap)
    nt, nch = cout_total // 32, Cin // 32
    t = planes.view(2, nt, 32, nch, 2, 2, 8, taps)                               # (p, nt, j, ch, ks, h, e, tap)
    t = t.permute(3, 7, 1, 4, 0, 5, 2, 6).contiguous()                           # (ch, tap, nt, ks, p, h, j, e): piece-major
    wpk = t.view(nch, taps, nt, 4 * 64 * 8).view(torch.float32).view(nch, taps, nt, 64, 16).contiguous()
    inv = torch.ones(cout_total, dtype=torch.float64, device=w.device)
    inv[:Cout] = 1.0 / S
    return wpk, inv.float()


def pack_conv_weights_h2_concat(ws):
    """conv1 + downsample of a BasicBlock3D over the same input: packed columns side by side."""
    parts = [pack_conv_weight_h2(w) for w in ws]
    return torch.cat([p[0] for p in parts], dim=2).contiguous(), torch.cat([p[1] for p in parts]).contiguous()


def conv3d_h2(x, wpk, scale, bias=None, residual=None, cout0=None, cout1=0, relu0=False, relu1=False, out0=None,
              out1=None, out_h2=(True, True), ksize=3, stride=1, algo=0):
    """3x3x3 stride-1 pad-1 conv on the fp16 matrix cores with split-fp16 operands (pw_conv3d_h2).
    x: H2 (B,D,H,W,Cin); wpk from pack_conv_weight_h2; scale (cout_total,) MUST already contain the packer's
    inv_scale (scale = bn_scale * inv_scale).  residual: H2 or fp32 tensor with y0's layout (may be out0 itself).
    out_h2: storage of (y0, y1): True -> H2, False -> fp32 tensor.  out0 / out1: destinations -- an ops.H2 is written under
    its own range slot, a raw buffer (or None) in h2 format gets a new slot (ops.new_slot).  Returns y0 [, y1]."""
    if not isinstance(x, H2):
        raise _lib.PreworldHipError('conv3d_h2 takes an ops.H2 input (ops.f32_to_h2)')
    B, D, H, W, Cin = x.shape
    nch, taps, nt = wpk.shape[:3]
    cout_total = nt * 32
    if nch * 32 != Cin or taps != ksize ** 3:
        raise _lib.PreworldHipError('packed h2 weight does not match Cin / ksize')
    pad = (ksize - 1) // 2
    D, H, W = [(v + 2 * pad - ksize) // stride + 1 for v in (D, H, W)]      # output grid from here on
    Din, Hin, Win = x.shape[1:4]
    if cout0 is None:
        cout0 = cout_total
    fm0, fm1 = int(bool(out_h2[0])), int(bool(out_h2[1]))

    def _dst(o, c, fm):
        if o is None:
            o = torch.empty(B, D, H, W, c, device=x.buf.device, dtype=_f32)
        if not fm:
            return (o.buf if isinstance(o, H2) else o), None
        return _out_h2(o, x.buf.device)
    y0, rng0 = _dst(out0, cout0, fm0)
    ld0 = _row_stride(y0, (B, D, H, W, cout0), 'y0')
    y1, ld1, rng1 = None, 0, None
    if cout1:
        y1, rng1 = _dst(out1, cout1, fm1)
        ld1 = _row_stride(y1, (B, D, H, W, cout1), 'y1')
    res, fmr = None, 0
    if residual is not None:
        fmr = int(isinstance(residual, H2))
        res = residual.buf if fmr else residual
        if _row_stride(res, (B, D, H, W, cout0), 'residual') != ld0:
            raise _lib.PreworldHipError('residual must have the same row stride as y0')
    if scale is None:
        raise _lib.PreworldHipError('conv3d_h2 needs scale = (BN scale or 1) * inv_scale of pack_conv_weight_h2')
    if scale.numel() != cout_total or (bias is not None and bias.numel() != cout_total):
        raise _lib.PreworldHipError('scale (with the weight pre-scale folded in) / bias must have cout_total=%d entries' % cout_total)
    xb = x.buf
    if not xb.is_contiguous():
        raise _lib.PreworldHipError('x must be dense')
    _lib.call('pw_conv3d_h2', _p(xb), _chk(wpk, _f32, 'wpk'), _p(scale), _p(bias), _p(res), _p(y0), _p(y1), B, Din, Hin, Win,
              Cin, cout_total, cout0, cout1, ld0, ld1, ksize, stride, int(relu0), int(relu1), algo, fm0, fm1, fmr,
              _rng(x), _rng(residual), _p(rng0), _p(rng1), _stream())
    r0 = H2(y0, rng0) if fm0 else y0
    if cout1:
        return r0, (H2(y1, rng1) if fm1 else y1)
    return r0


def fpn3d_fuse(x8, wpk8, y16, y32, scale, bias, relu=True, out=None, out_h2=False):
    """LSSFPN3D tail: ReLU(BN(W8 x8 + up2(y16) + up4(y32))) -- lss_fpn.py:132-148.  x8: fp32 tensor (wpk8 from
    pack_conv_weight) or ops.H2 (wpk8 from pack_conv_weight_h2, its inv_scale folded into scale); out_h2: return ops.H2."""
    x_h2 = isinstance(x8, H2)
    xb = x8.buf if x_h2 else x8
    B, D, H, W, C8 = xb.shape
    if out is None:
        out = torch.empty(B, D, H, W, 32, device=xb.device, dtype=_f32)
    orng = None
    if out_h2:
        out, orng = _out_h2(out, xb.device)
    _lib.call('pw_fpn3d_fuse', _chk(xb, _f32, 'x8'), _chk(wpk8, _f32, 'wpk8'), _chk(y16, _f32, 'y16'),
              _chk(y32, _f32, 'y32'), _p(scale), _p(bias), _chk(out, _f32, 'out'), B, D, H, W, C8,
              y16.shape[1], y16.shape[2], y16.shape[3], y32.shape[1], y32.shape[2], y32.shape[3],
              int(relu), int(x_h2), int(bool(out_h2)), _rng(x8), _p(orng), _stream())
    return H2(out, orng) if out_h2 else out


def pack_conv_weight16(w):
    """(16, Cin, 3,3,3) conv weight -> float[Cin/32][27][64][8] for the 16x16x4 MFMA OccHead
    kernel: wpk[ch][tap][g*16+j][s] = w[j][ch*32+g*8+s][tap]."""
    Cout, Cin = w.shape[:2]
    if Cout != 16 or Cin % 32:
        raise _lib.PreworldHipError('pack_conv_weight16 expects (16, 32k, 3,3,3)')
    nch = Cin // 32
    wp = w.reshape(16, nch, 4, 8, 27).permute(1, 4, 2, 0, 3).contiguous()    # (ch, tap, g, j, s)
    return wp.view(nch, 27, 64, 8).float().contiguous()


def occ_head_fused(x, wpk, scale, bias, w1, s1, b1, w2, want_logits=False, occ=None, want_geo=False,
                   empty_idx=17):
    """OccHead (occupancy_head.py:124-177) on channels-last x (B,D,H,W,32); wpk from pack_conv_weight16 (direct
    MFMA kernel) or pack_conv_weight_wino(w, cout_total=16) (Winograd kernel, 32 input channels):
    returns uint8 argmax (B,D,H,W) [, logits (B,D,H,W,18)] [, geo_occ uint8 (B,D,H,W) =
    0 where occupied / 17 where empty, preworld_temporal_traj.py:313-319]."""
    B, D, H, W, Cin = x.shape
    if occ is None:
        occ = torch.empty(B, D, H, W, device=x.device, dtype=torch.uint8)
    logits = torch.empty(B, D, H, W, 18, device=x.device, dtype=_f32) if want_logits else None
    geo = torch.empty(B, D, H, W, device=x.device, dtype=torch.uint8) if want_geo else None
    _lib.call('pw_occ_head_fused', _chk(x, _f32, 'x'), _chk(wpk, _f32, 'wpk'), _chk(scale, _f32, 'scale'),
              _chk(bias, _f32, 'bias'), _chk(w1, _f32, 'w1'), _chk(s1, _f32, 's1'), _chk(b1, _f32, 'b1'),
              _chk(w2, _f32, 'w2'), _p(occ), _p(logits), _p(geo), int(empty_idx), B, D, H, W, Cin, 16, 8, 18,
              64 if wpk.dim() == 5 else (16 if wpk.shape[-1] == 8 else 32), _stream())
    out = (occ,) + ((logits,) if want_logits else ()) + ((geo,) if want_geo else ())
    return out if len(out) > 1 else occ


def pack_occ_weight_h2(w):
    """OccHead conv weight (16, 32, 3,3,3) -> (wpk, inv_scale (16,)) for pw_occ_head_h2 (include/preworld_hip.h):
    wpk[tap][plane][g*16 + j][e] = plane of S_j * w[j][16*(g>>1) + 8*(g&1) + e][tap] as a float32-typed (27, 2, 64, 4) tensor."""
    Cout, Cin = w.shape[:2]
    if (Cout, Cin) != (16, 32) or tuple(w.shape[2:]) != (3, 3, 3):
        raise _lib.PreworldHipError('pack_occ_weight_h2 expects a (16, 32, 3, 3, 3) weight')
    wf = w.reshape(Cout, -1).double()
    S = torch.exp2(torch.floor(torch.log2(1023.0 / wf.abs().amax(dim=1).clamp_min(1e-30))))
    ws = (wf * S[:, None]).view(Cout, Cin, 27)
    hi = ws.to(torch.float16)
    lo = (ws - hi.double()).to(torch.float16)
    t = torch.stack([hi, lo], 0).view(2, 16, 2, 2, 8, 27)                          # (p, j, ks, half, e, tap): c = 16 ks + 8 half + e
    t = t.permute(5, 0, 2, 3, 1, 4).contiguous()                                   # (tap, p, ks, half, j, e): g = 2 ks + half
    wpk = t.view(27, 2, 64, 8).view(torch.float32).view(27, 2, 64, 4).contiguous()
    return wpk, (1.0 / S).float()


def occ_head_bounds(w0, s0, b0, w1, s1, b1):
    """a-priori magnitude bounds of OccHead's two hidden layers for pw_occ_head_h2 (include/preworld_hip.h): w0 (16,32,3,3,3) /
    w1 (8,16) conv weights, s* / b* folded BN.  Returns host floats (mid_a, mid_b, hid_a, hid_b)."""
    mid_a = float((s0.double().abs()[:w0.shape[0]] * w0.double().abs().flatten(1).sum(1)).max())
    hid_a = float((s1.double().abs() * w1.double().abs().flatten(1).sum(1)).max())
    return mid_a, float(b0.double().abs().max()), hid_a, float(b1.double().abs().max())


def pack_occ_tail_h2(w1, s1, b1, w2):
    """The 16 -> 8 (+BN+ReLU) -> 18 tail of OccHead as operands of pw_occ_head_h2: w1 (8,16), s1 / b1 (8,) folded BN, w2 (18,8)
    -> (tailpk float32 (800,), inv2).  One power-of-two pre-scale per matrix (argmax must see one common scale)."""
    dev = w1.device
    S1, h1, l1 = _split_planes(w1.double())
    S2, h2_, l2 = _split_planes(w2.double())
    lane = torch.arange(64, device=dev)
    row, kg = lane & 15, lane >> 4
    k = (4 * kg)[:, None] + torch.arange(4, device=dev)[None, :]                    # (64, 4): k = 4 g + e

    def frag(plane, nrow, ncol, row0=0):
        full = torch.zeros(32, 16, dtype=torch.float16, device=dev)
        full[:nrow, :ncol] = plane
        return full[(row0 + row)[:, None], k]                                       # (64, 4) fp16

    frags = [frag(h1, 8, 16), frag(l1, 8, 16), frag(h2_, 18, 8), frag(l2, 18, 8), frag(h2_, 18, 8, 16), frag(l2, 18, 8, 16)]
    pk = torch.stack(frags, 0).contiguous().view(torch.float32).reshape(-1)        # 6 * 64 * 2 floats
    s1p = torch.zeros(16, dtype=_f32, device=dev)
    b1p = torch.zeros(16, dtype=_f32, device=dev)
    s1p[:8] = (s1.double() / S1).float()
    b1p[:8] = b1.float()
    return torch.cat([pk, s1p, b1p]).contiguous(), 1.0 / S2


def occ_head_h2(x, wpk, scale, bias, tailpk, inv2, bounds, want_logits=False, occ=None, want_geo=False, empty_idx=17, geo=None):
    """OccHead (occupancy_head.py:124-177) on the fp16 matrix cores: x = ops.H2 (B,D,H,W,32); wpk from pack_occ_weight_h2;
    scale (16,) MUST already contain the packer's inv_scale; (tailpk, inv2) from pack_occ_tail_h2; bounds from occ_head_bounds.
    Returns like occ_head_fused.  occ / geo: optional uint8 (B,D,H,W) destinations with ARBITRARY (equal) strides -- e.g. the
    .permute(0,3,2,1) view of an (B,X,Y,Z)-contiguous payload buffer: the kernel then writes the reference's (X,Y,Z) arrays in
    place (pw_occ_head_h2_strided) and no transposing copy is needed afterwards."""
    if not isinstance(x, H2):
        raise _lib.PreworldHipError('occ_head_h2 takes an ops.H2 input (ops.f32_to_h2)')
    B, D, H, W, Cin = x.shape
    if geo is not None:
        want_geo = True
    if occ is None:
        occ = torch.empty(B, D, H, W, device=x.device, dtype=torch.uint8)
    logits = torch.empty(B, D, H, W, 18, device=x.device, dtype=_f32) if want_logits else None
    if want_geo and geo is None:
        # (empty_strided sizes the storage for the gapped / interleaved strides; as_strided over a dense allocation did not)
        geo = torch.empty_strided(tuple(occ.shape), tuple(occ.stride()), device=x.device, dtype=torch.uint8)
    if tailpk.numel() != 800 or scale.numel() < 16 or bias.numel() < 16:
        raise _lib.PreworldHipError('occ_head_h2: tailpk (800,), scale / bias (16,) expected')
    for t in (occ, geo):
        if t is not None and (tuple(t.shape) != (B, D, H, W) or t.dtype != torch.uint8 or not t.is_cuda):
            raise _lib.PreworldHipError('occ_head_h2: occ / geo must be uint8 device tensors of shape (B, D, H, W)')
    strides, span = None, 0
    if not occ.is_contiguous() or (geo is not None and not geo.is_contiguous()):
        if geo is not None and tuple(geo.stride()) != tuple(occ.stride()):
            raise _lib.PreworldHipError('occ_head_h2: occ and geo must have the same strides')
        st = [int(v) for v in occ.stride()]
        strides = (ctypes.c_int64 * 4)(*st)
        span = 1 + sum((n - 1) * v for n, v in zip((B, D, H, W), st))
        # (the C side rejects overlapping stride sets -- two voxels storing to one byte would race)
    _lib.call('pw_occ_head_h2_strided', _chk(x.buf, _f32, 'x'), _chk(wpk, _f32, 'wpk'), _chk(scale, _f32, 'scale'),
              _chk(bias, _f32, 'bias'), _chk(tailpk, _f32, 'tailpk'), float(inv2), _p(occ), _p(logits), _p(geo), strides, int(span),
              int(empty_idx), B, D, H, W, Cin, 16, 8, 18, _rng(x), float(bounds[0]), float(bounds[1]), float(bounds[2]),
              float(bounds[3]), _stream())
    out = (occ,) + ((logits,) if want_logits else ()) + ((geo,) if want_geo else ())
    return out if len(out) > 1 else occ


def forecast_pack(fusion_w1, fusion_w2):
    w1p = torch.empty(4096, device=fusion_w1.device, dtype=_f32)
    w2p = torch.empty(4096, device=fusion_w1.device, dtype=_f32)
    _lib.call('pw_forecast_pack', _chk(fusion_w1, _f32, 'fusion_head.0.weight'),
              _chk(fusion_w2, _f32, 'fusion_head.2.weight'), _p(w1p), _p(w2p), _stream())
    return w1p, w2p


def forecast_prologue(ego, plan, fusion_w1, fusion_b1):
    """ego (n_samples, 21); plan = [(w0,b0),(w2,b2),(w4,b4)] ->
    (ego_feat (n,32), c1 (n,128) natural order, c1p (n,128) accumulator order)."""
    n, dim = ego.shape
    ef = torch.empty(n, 32, device=ego.device, dtype=_f32)
    c1 = torch.empty(n, 128, device=ego.device, dtype=_f32)
    c1p = torch.empty(n, 128, device=ego.device, dtype=_f32)
    (w0, b0), (w2, b2), (w4, b4) = plan
    _lib.call('pw_forecast_prologue', _chk(ego, _f32, 'ego'), n, dim, _chk(w0, _f32, 'w0'),
              _chk(b0, _f32, 'b0'), _chk(w2, _f32, 'w2'), _chk(b2, _f32, 'b2'), _chk(w4, _f32, 'w4'),
              _chk(b4, _f32, 'b4'), _chk(fusion_w1, _f32, 'fw1'), _chk(fusion_b1, _f32, 'fb1'),
              _p(ef), _p(c1), _p(c1p), _stream())
    return ef, c1, c1p


def forecast_steps(v0, n_samples, w1p, w2p, c1p, fusion_b2, n_steps, states=None):
    """v0 (n_samples, ..., 32) channels-last -> states (n_steps, *v0.shape)."""
    n_total = v0.numel() // 32
    if states is None:
        states = torch.empty((n_steps,) + tuple(v0.shape), device=v0.device, dtype=_f32)
    _lib.call('pw_forecast_steps', _chk(v0, _f32, 'v0'), n_total // n_samples, n_samples,
              _chk(w1p, _f32, 'w1p'), _chk(w2p, _f32, 'w2p'), _chk(c1p, _f32, 'c1p'),
              _chk(fusion_b2, _f32, 'fb2'), n_steps, _chk(states, _f32, 'states'), _stream())
    return states


def _split_planes(w):
    """float64 tensor -> (power-of-two scale S, fp16 hi, fp16 lo) of S * w, largest |S w| in [512, 1024)"""
    S = float(2.0 ** torch.floor(torch.log2(1023.0 / w.abs().max().clamp_min(1e-30))))
    ws = w * S
    hi = ws.to(torch.float16)
    return S, hi, (ws - hi.double()).to(torch.float16)


def forecast_pack_h2(fusion_w1, fusion_w2):
    """fusion_head.{0,2}.weight ([128][64], [32][128]) -> (w1p, w2p, inv1, inv2, w1_l1max): split-fp16 MFMA A operands in the
    order pw_forecast_steps_h2 documents (include/preworld_hip.h), the inverses of their power-of-two pre-scales and the largest
    row L1 norm of the voxel part of W1 (the kernel's a-priori bound on the hidden activations)."""
    import numpy as np
    dev = fusion_w1.device
    lane = np.arange(64)
    i, h = lane & 31, lane >> 5
    e = np.arange(8)
    # row_of(8 kb + e, h) = ((8kb+e) & 3) + 8 * ((8kb+e) >> 2) + 4 h
    t, kb, ln, ee = np.meshgrid(np.arange(4), np.arange(2), lane, e, indexing='ij')
    r = 8 * kb + ee
    k = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5)
    row = ln & 31
    T = lambda a: torch.from_numpy(a.reshape(-1)).to(dev)                      # noqa: E731
    S1, h1, l1 = _split_planes(fusion_w1[:, :32].double())                     # voxel part of W1 (128, 32)
    S2, h2_, l2 = _split_planes(fusion_w2.double())                            # (32, 128)
    idx1 = (T(t) * 32 + T(row), T(k))
    idx2 = (T(row), T(t) * 32 + T(k))
    w1p = torch.stack([h1[idx1].view(4, 2, 64, 8), l1[idx1].view(4, 2, 64, 8)], dim=2)     # (t, kb, p, lane, e)
    w2p = torch.stack([h2_[idx2].view(4, 2, 64, 8), l2[idx2].view(4, 2, 64, 8)], dim=2)
    return (w1p.contiguous().view(torch.float32).contiguous(), w2p.contiguous().view(torch.float32).contiguous(),
            1.0 / S1, 1.0 / S2, float(fusion_w1[:, :32].double().abs().sum(1).max()))


def forecast_steps_h2(v0, n_samples, packed, c1p, fusion_b2, n_steps, states=None, out_h2=False):
    """forecast_steps on the fp16 matrix cores with split-fp16 operands; packed = forecast_pack_h2(...).
    v0: fp32 tensor or ops.H2; out_h2=True returns the states as ONE ops.H2 of shape (n_steps, *v0.shape).  The recursion
    runs in the units of the states' range slot (a new one, or `states.rng`) whatever the output format."""
    w1p, w2p, inv1, inv2, l1 = packed
    v0_h2 = isinstance(v0, H2)
    vbuf = v0.buf if v0_h2 else v0
    n_total = vbuf.numel() // 32
    if states is None:
        states = torch.empty((n_steps,) + tuple(vbuf.shape), device=vbuf.device, dtype=_f32)
    states, srng = _out_h2(states, vbuf.device)
    if srng is None:
        srng = new_slot(vbuf.device)
    _lib.call('pw_forecast_steps_h2', _chk(vbuf, _f32, 'v0'), n_total // n_samples, n_samples, _chk(w1p, _f32, 'w1p'),
              _chk(w2p, _f32, 'w2p'), float(inv1), float(inv2), _chk(c1p, _f32, 'c1p'), _chk(fusion_b2, _f32, 'fb2'),
              n_steps, _chk(states, _f32, 'states'), int(v0_h2), int(bool(out_h2)), _rng(v0), _p(srng), float(l1), _stream())
    return H2(states, srng) if out_h2 else states


def softplus(x):
    y = torch.empty_like(x)
    _lib.call('pw_softplus', _chk(x, _f32, 'x'), _p(y), x.numel(), _stream())
    return y


# ------------------------------------------------------------------------------ ray table + WRS weights
def pts2ray(coor, label_depth, label_seg, label_img, c2w, cam_intrinsic):
    """mmdet3d/datasets/ray.py:47-55 (one camera): (n,16) rows
    {x, y, depth, seg, rays_o[3], rays_d[3], viewdirs[3], rgb[3]}."""
    n = coor.shape[0]
    rays = torch.empty(n, 16, device=coor.device, dtype=_f32)
    _lib.call('pw_pts2ray', _chk(coor.float().contiguous(), _f32, 'coor'),
              _chk(label_depth.float().contiguous(), _f32, 'label_depth'),
              _chk(label_seg.float().contiguous(), _f32, 'label_seg'),
              _chk(label_img.float().contiguous(), _f32, 'label_img'),
              _chk(c2w.float().contiguous(), _f32, 'c2w'), _chk(cam_intrinsic.float().contiguous(), _f32, 'K'),
              n, _p(rays), _stream())
    return rays


def class_count(rays, n_cls=17, counts=None):
    if counts is None:
        counts = torch.zeros(n_cls, device=rays.device, dtype=torch.int64)
    _lib.call('pw_class_count', _chk(rays, _f32, 'rays'), rays.shape[0], n_cls, _p(counts), _stream())
    return counts


def wrs_weights(rays, frame_id, balance_weight, dynamic_class, weight_adj=0.3, weight_dyn=0.0):
    """ray.py:99-112 for one camera's rays of frame `frame_id`."""
    n = rays.shape[0]
    w = torch.empty(n, device=rays.device, dtype=_f32)
    dyn = dynamic_class.to(device=rays.device, dtype=_i32).contiguous()
    _lib.call('pw_wrs_weights', _chk(rays, _f32, 'rays'), n, int(frame_id),
              _chk(balance_weight.float().contiguous(), _f32, 'balance_weight'), balance_weight.numel(),
              _p(dyn), dyn.numel(), float(weight_adj), float(weight_dyn), _p(w), _stream())
    return w


# ------------------------------------------------------------------------------ DepthNet tail
def depthnet_tail(x, D, C):
    """x (BN, >=D+C, H, W) DepthNet output -> (depth (BN,D,H,W) softmaxed over D,
    feat_cl (BN,H,W,C) channels-last context) -- view_transformer.py:797-801 and :189 in one pass."""
    BN, XC, H, W = x.shape
    depth = torch.empty(BN, D, H, W, device=x.device, dtype=_f32)
    feat = torch.empty(BN, H, W, C, device=x.device, dtype=_f32)
    _lib.call('pw_depthnet_tail', _chk(x, _f32, 'x'), BN, XC, D, C, H * W, _p(depth), _p(feat), _stream())
    return depth, feat


def stereo_cost_volume(prev, curr, frustum, k2s_sensor, intrins, post_rots, post_trans, bias=0.0):
    """DepthNet.calculate_cost_volumn (view_transformer.py:546-604).  prev/curr (B*N, C, H, W) stereo
    features (any strides; torch.channels_last storage is the fast path), frustum (D,H,W,3) the
    cv_frustum, k2s_sensor (B,N,4,4), intrins/post_rots (B,N,3,3), post_trans (B,N,3).
    Returns the softmaxed cost volume (B*N, D, H, W)."""
    BN, C, H, W = curr.shape
    D = frustum.shape[0]
    prev, curr = prev.float(), curr.float()          # no-op on fp32; keeps the strides of a dense tensor
    if tuple(frustum.shape[1:3]) != (H, W) or tuple(prev.shape) != tuple(curr.shape) or prev.stride() != curr.stride():
        raise _lib.PreworldHipError('stereo_cost_volume: prev/curr/frustum shapes or strides disagree')
    dev = curr.device
    ds = frustum[:, 0, 0, 2].contiguous().float()
    xs = frustum[0, 0, :, 0].contiguous().float()
    ys = frustum[0, :, 0, 1].contiguous().float()
    ipr, comb, tr = lss_camera_matrices(k2s_sensor, intrins, post_rots)
    out = torch.empty(BN, D, H, W, device=dev, dtype=_f32)
    sbn, sc, sy, sx = curr.stride()
    _lib.call('pw_stereo_cost_volume', _p(prev), _p(curr), BN, C, H, W, sbn, sc, sy, sx, _p(ds), D, _p(xs), _p(ys),
              _p(ipr), _chk(post_trans.reshape(BN, 3).float().contiguous(), _f32, 'post_trans'), _p(comb), _p(tr),
              _chk(intrins.reshape(BN, 9).float().contiguous(), _f32, 'intrins'),
              _chk(post_rots.reshape(BN, 9).float().contiguous(), _f32, 'post_rots'),
              float(4 * W), float(4 * H), float(bias), _p(out), _stream())
    return out


# ------------------------------------------------------------------------------ A20 trajectory branch
def global_avgpool_ndhwc(x):
    """x (B, ..., C) channels-last -> (B, C): nn.AdaptiveAvgPool3d((1,1,1)) of
    DownScaleModule3DCustom (occupancy_head.py:187,198)."""
    B, C = x.shape[0], x.shape[-1]
    n_vox = x.numel() // (B * C)
    y = torch.empty(B, C, device=x.device, dtype=_f32)
    _lib.call('pw_global_avgpool_ndhwc', _chk(x, _f32, 'x'), B, n_vox, C, _p(y), _stream())
    return y


_ACT = {None: 0, 'none': 0, 'relu': 1, 'softplus': 2}


def linear_act(x, weight, bias=None, act=None):
    """nn.Linear (+ReLU / Softplus) on a few rows: x (rows, in), weight (out, in) as stored."""
    rows, n_in = x.shape
    n_out = weight.shape[0]
    y = torch.empty(rows, n_out, device=x.device, dtype=_f32)
    _lib.call('pw_linear_act', _chk(x, _f32, 'x'), _chk(weight, _f32, 'weight'),
              _p(bias.contiguous()) if bias is not None else None, _p(y), rows, n_in, n_out, _ACT[act],
              _stream())
    return y


# ------------------------------------------------------------------------------ attribute MLPs
def _row_of(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def _mfma_pack_indices(n_tiles):
    """index helpers for the transposed MFMA chain (see pw_forecast.hip): for packed position
    [t][q][lane][e] returns (tile-row i, k index row_of(4q+e, lane>>5))."""
    import numpy as np
    t, q, lane, e = np.meshgrid(np.arange(n_tiles), np.arange(4), np.arange(64), np.arange(4),
                                indexing='ij')
    i, h = lane & 31, lane >> 5
    k = ((4 * q + e) & 3) + 8 * ((4 * q + e) >> 2) + 4 * h
    return t, i, k


def pack_mlp_blocks(mlps):
    """Up to three nn.Sequential(Linear(32,64), Softplus, Linear(64,n_i)[, ...]) over the same 32-channel input ->
    (w1p, w2p, b1p, b2) in the operand order pw_attr_mlp consumes: hidden block i occupies units [64i, 64i+64), its n_i
    outputs follow block i-1's in the packed output row (sum n_i <= 24); absent blocks are zero."""
    import numpy as np
    dev = mlps[0][0].weight.device
    W1 = torch.zeros(192, 32, device=dev)
    b1 = torch.zeros(192, device=dev)
    W2 = torch.zeros(32, 192, device=dev)
    b2 = torch.zeros(32, device=dev)
    row = 0
    for blk, m in enumerate(mlps):
        if tuple(m[0].weight.shape) != (64, 32) or m[2].weight.shape[1] != 64:
            raise _lib.PreworldHipError('pw_attr_mlp is built for Linear(32,64) -> Softplus -> Linear(64,n) blocks')
        n = m[2].weight.shape[0]
        W1[blk * 64:(blk + 1) * 64] = m[0].weight.float()
        b1[blk * 64:(blk + 1) * 64] = m[0].bias.float()
        W2[row:row + n, blk * 64:(blk + 1) * 64] = m[2].weight.float()
        b2[row:row + n] = m[2].bias.float()
        row += n
    if row > 24 or len(mlps) > 3:
        raise _lib.PreworldHipError('pw_attr_mlp writes 24 packed output channels from at most 3 blocks')
    t, i, k = _mfma_pack_indices(6)
    ti, ii, ki = [torch.from_numpy(a.reshape(-1)).to(dev) for a in (t, i, k)]
    w1p = W1[ti * 32 + ii, ki].contiguous()                       # W1[t*32+i][k]
    w2p = W2[ii, ti * 32 + ki].contiguous()                       # W2[i][t*32+k]
    hh, tt, rr = np.meshgrid(np.arange(2), np.arange(6), np.arange(16), indexing='ij')
    idx = tt * 32 + ((rr & 3) + 8 * (rr >> 2) + 4 * hh)
    b1p = b1[torch.from_numpy(idx.reshape(-1)).to(dev)].contiguous()
    return w1p, w2p, b1p, b2.contiguous()


def pack_attr_mlp(density_mlp, semantic_mlp, color_mlp):
    """density (2) / semantic (17) / color (3) MLPs of preworld.py:81-104 -> packed channels [0:2], [2:19], [19:22]."""
    return pack_mlp_blocks([density_mlp, semantic_mlp, color_mlp])


def attr_mlp(v_cl, packed, final_softplus=True, out=None):
    """v_cl (..., 32) channels-last -> packed attribute grid (..., 24):
    [0:2] density_prob, [2:19] semantic, [19:22] color (preworld_temporal_traj.py:231-236)."""
    w1p, w2p, b1p, b2 = packed
    n = v_cl.numel() // 32
    if out is None:
        out = torch.empty(tuple(v_cl.shape[:-1]) + (24,), device=v_cl.device, dtype=_f32)
    _lib.call('pw_attr_mlp', _chk(v_cl, _f32, 'v'), n, _chk(w1p, _f32, 'w1p'), _chk(w2p, _f32, 'w2p'),
              _chk(b1p, _f32, 'b1p'), _chk(b2, _f32, 'b2'), int(final_softplus), _chk(out, _f32, 'out'),
              _stream())
    return out


# ------------------------------------------------------------------------------ render ops
_i64 = torch.int64


def raw2alpha(density, shift, interval):
    """render_utils_cuda.raw2alpha (render_utils.cpp:120-124) -> (exp, alpha)."""
    d = density.contiguous()
    e = torch.empty_like(d)
    a = torch.empty_like(d)
    _lib.call('pw_raw2alpha', _chk(d, _f32, 'density'), float(shift), float(interval), d.numel(),
              _p(e), _p(a), _stream())
    return e, a


def raw2alpha_backward(exp, grad_back, interval):
    g = torch.empty_like(exp)
    _lib.call('pw_raw2alpha_backward', _chk(exp, _f32, 'exp'), _chk(grad_back.contiguous(), _f32, 'grad_back'),
              float(interval), exp.numel(), _p(g), _stream())
    return g


def alpha2weight(alpha, ray_id, n_rays):
    """render_utils_cuda.alpha2weight (render_utils.cpp:142-149) ->
    (weight, T, alphainv_last, i_start, i_end)."""
    a = alpha.contiguous()
    n = a.numel()
    dev = a.device
    w = torch.empty(n, device=dev, dtype=_f32)
    T = torch.empty(n, device=dev, dtype=_f32)
    last = torch.empty(n_rays, device=dev, dtype=_f32)
    i_s = torch.empty(n_rays, device=dev, dtype=_i64)
    i_e = torch.empty(n_rays, device=dev, dtype=_i64)
    _lib.call('pw_alpha2weight', _chk(a, _f32, 'alpha'), _chk(ray_id.contiguous(), _i64, 'ray_id'), n,
              int(n_rays), _p(w), _p(T), _p(last), _p(i_s), _p(i_e), _stream())
    return w, T, last, i_s, i_e


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights,
                          grad_last):
    g = torch.empty_like(alpha)
    _lib.call('pw_alpha2weight_backward', _chk(alpha, _f32, 'alpha'), _chk(weight, _f32, 'weight'),
              _chk(T, _f32, 'T'), _chk(alphainv_last, _f32, 'alphainv_last'), _chk(i_start, _i64, 'i_start'),
              _chk(i_end, _i64, 'i_end'), int(n_rays), _chk(grad_weights.contiguous(), _f32, 'grad_weights'),
              _chk(grad_last.contiguous(), _f32, 'grad_last'), alpha.numel(), _p(g), _stream())
    return g


def cumdist_thres(dist, thres):
    """ub360_utils_cuda.cumdist_thres (ub360_utils.cpp:15-18): (R,S) float -> bool mask."""
    d = dist.contiguous()
    R, S = d.shape
    m = torch.empty(R, S, device=d.device, dtype=torch.uint8)
    _lib.call('pw_cumdist_thres', _chk(d, _f32, 'dist'), float(thres), R, S, _p(m), _stream())
    return m.bool()


class Raw2Alpha(torch.autograd.Function):
    """Drop-in for mmdet3d/models/nerf/utils.py:26-50."""

    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = raw2alpha(density, shift, interval)
        if density.requires_grad:
            ctx.save_for_backward(exp)
            ctx.interval = interval
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        exp = ctx.saved_tensors[0]
        return raw2alpha_backward(exp, grad_back.contiguous(), ctx.interval), None, None


class Alphas2Weights(torch.autograd.Function):
    """Drop-in for mmdet3d/models/nerf/utils.py:52-68."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        weights, T, alphainv_last, i_start, i_end = alpha2weight(alpha, ray_id, N)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
            ctx.n_rays = N
        return weights, alphainv_last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
        grad = alpha2weight_backward(alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays,
                                     grad_weights, grad_last)
        return grad, None, None


def render_rays(rays_o, rays_d, t, grid, consts, c_sigma=0, c_sem=2, n_sem=17, c_rgb=19,
                want_debug=False):
    """Fused forward of NerfHead.render_one_scene + render_depth/semantic/color
    (nerf_head.py:165-269,331-353).  grid: packed (Z,Y,X,GC) attribute grid.  consts: 27 floats
    (see include/preworld_hip.h).  Returns dict(depth (R), semantic (R,17), color (R,3),
    alphainv_last (R) [, counts (R,3), weights (R,S), mask (R,S)])."""
    R, S = rays_o.shape[0], t.numel()
    Z, Y, X, GC = grid.shape
    dev = rays_o.device
    bf16 = grid.dtype == torch.bfloat16           # bf16 STORAGE of the packed grid (grid.to(torch.bfloat16)); fp32 arithmetic
    if not bf16 and grid.dtype != _f32:
        raise _lib.PreworldHipError('grid must be float32 or bfloat16')
    if not grid.is_cuda or not grid.is_contiguous():
        raise _lib.PreworldHipError('grid must be a contiguous device tensor')
    depth = torch.empty(R, device=dev, dtype=_f32)
    sem = torch.empty(R, n_sem, device=dev, dtype=_f32)
    rgb = torch.empty(R, 3, device=dev, dtype=_f32)
    last = torch.empty(R, device=dev, dtype=_f32)
    counts = torch.empty(R, 3, device=dev, dtype=_i32) if want_debug else None
    weights = torch.empty(R, S, device=dev, dtype=_f32) if want_debug else None
    mask = torch.empty(R, S, device=dev, dtype=torch.uint8) if want_debug else None
    ch = (ctypes.c_float * 27)(*[float(v) for v in consts])
    _lib.call('pw_render_rays', _chk(rays_o.contiguous(), _f32, 'rays_o'),
              _chk(rays_d.contiguous(), _f32, 'rays_d'), R, _chk(t, _f32, 't'), S,
              _p(grid), X, Y, Z, GC, c_sigma, c_sem, n_sem, c_rgb, ch, _p(depth),
              _p(sem), _p(rgb), _p(last), _p(counts), _p(weights), _p(mask), int(bf16), _stream())
    out = dict(depth=depth, semantic=sem, color=rgb, alphainv_last=last)
    if want_debug:
        out.update(counts=counts, weights=weights, mask=mask.bool())
    return out


def render_rays_backward(rays_o, rays_d, t, grid, consts, g_depth, g_sem, g_rgb, g_last, g_weights=None, c_sigma=0, c_sem=2,
                         n_sem=17, c_rgb=19, grad_grid=None, algo=None, max_entries=0):
    """Backward of render_rays: gradient of the packed (Z,Y,X,GC) grid given the gradients of depth (R), semantic (R,17),
    color (R,3), alphainv_last (R) [and of the dense weights (R,S)].
    algo 'sorted' (default): pw_render_rays_backward_sorted -- entries sorted by voxel,
    fixed-point segmented sums, no float atomics, bit-reproducible; 'atomics': pw_render_rays_backward (168 float atomics per
    kept sample, arrival order decides the last bits; kept for A/B).  max_entries: upper bound of the (sample, corner) entries of the
    sorted form -- 8 x render_rays' counts[:, 1].sum() -- which sizes its workspace (0: worst case, 40 B x R x S x 8)."""
    import os
    R, S = rays_o.shape[0], t.numel()
    Z, Y, X, GC = grid.shape
    if grad_grid is None:
        grad_grid = torch.zeros_like(grid)
    if R == 0:
        return grad_grid
    algo = algo or 'sorted'
    ch = (ctypes.c_float * 27)(*[float(v) for v in consts])
    gw = _chk(g_weights.contiguous(), _f32, 'g_weights') if g_weights is not None else None
    args = (_chk(rays_o.contiguous(), _f32, 'rays_o'), _chk(rays_d.contiguous(), _f32, 'rays_d'), R,
            _chk(t, _f32, 't'), S, _chk(grid, _f32, 'grid'), X, Y, Z, GC, c_sigma, c_sem, n_sem, c_rgb, ch,
            _chk(g_depth.contiguous(), _f32, 'g_depth'), _chk(g_sem.contiguous(), _f32, 'g_sem'),
            _chk(g_rgb.contiguous(), _f32, 'g_rgb'), _chk(g_last.contiguous(), _f32, 'g_last'), gw)
    if algo == 'atomics':
        _lib.call('pw_render_rays_backward', *args, _chk(grad_grid, _f32, 'grad_grid'), _stream())
        return grad_grid
    g20 = torch.cat([g_sem.float(), g_rgb.float()], dim=1).contiguous()
    gmax = g20.abs().max().reshape(1).contiguous()
    nbytes = _lib.call_size('pw_render_backward_workspace_bytes', R, S, X, Y, Z, int(max_entries))
    ws = _workspace(nbytes, grid.device)
    off = (-ws.data_ptr()) % 256
    _lib.call('pw_render_rays_backward_sorted', *args, _chk(g20, _f32, 'g_semrgb'), _chk(gmax, _f32, 'g_absmax'),
              ctypes.c_void_p(ws.data_ptr() + off), nbytes, int(max_entries), _chk(grad_grid, _f32, 'grad_grid'), _stream())
    return grad_grid


_PINNED = []          # spare pinned int64[1] buffers (hipHostMalloc is slow: reused)


def _async_host_scalar(t):
    """start the copy of a device scalar to the host and return poll() -> its value as an int once it has arrived, else None (never
    blocks, never synchronises)"""
    buf = _PINNED.pop() if _PINNED else torch.empty(1, dtype=torch.int64, pin_memory=True)
    buf.copy_(t.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    state = {}

    def poll():
        if 'v' not in state and ev.query():
            state['v'] = int(buf[0])
            _PINNED.append(buf)
        return state.get('v')
    return poll


class RenderRays(torch.autograd.Function):
    """Differentiable fused render head: forward = pw_render_rays, backward = pw_render_rays_backward.  Replaces the autograd
    graph the reference builds through 3x F.grid_sample, Raw2Alpha, Alphas2Weights and 3x segment_coo
    (nerf_head.py:165-269,331-353; utils.py:26-68).  Returns (depth, semantic, color, alphainv_last, weights (R,S) dense)."""

    @staticmethod
    def forward(ctx, grid, rays_o, rays_d, t, consts):
        grid = grid.contiguous()
        out = render_rays(rays_o, rays_d, t, grid, consts, want_debug=True)
        ctx.save_for_backward(grid, rays_o, rays_d, t)
        ctx.consts = tuple(float(v) for v in consts)
        # how many samples passed the alpha threshold: on its way to the host without anyone waiting for it; if it has arrived by
        # the time the backward runs, the backward's entry arrays are sized by it instead of by the worst case (5.1 GB at 38 400 x 417)
        ctx.kept = _async_host_scalar(out['counts'][:, 1].sum(dtype=torch.int64))
        return out['depth'], out['semantic'], out['color'], out['alphainv_last'], out['weights']

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_depth, g_sem, g_rgb, g_last, g_w):
        grid, rays_o, rays_d, t = ctx.saved_tensors
        kept = ctx.kept()
        gg = render_rays_backward(rays_o, rays_d, t, grid, ctx.consts, g_depth, g_sem, g_rgb, g_last, g_w,
                                  max_entries=0 if kept is None else 8 * max(kept, 1))
        return gg, None, None, None, None


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, s):
        R, S_ = w.shape
        nbytes = _lib.call_size('pw_distortion_workspace_bytes', R)
        ws = _workspace(nbytes, w.device)
        out = torch.empty(3, device=w.device, dtype=_f32)
        _lib.call('pw_distortion_loss', _p(w), _p(s), R, S_, _p(ws), nbytes, _p(out[0:1]), _p(out[1:3]), _stream())
        ctx.save_for_backward(w, s, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        w, s, out = ctx.saved_tensors
        gw = torch.empty_like(w)
        _lib.call('pw_distortion_loss_backward', _p(w), _p(s), w.shape[0], w.shape[1], _p(out[1:3]), _p(g.reshape(1).float().contiguous()),
                  _p(gw), _stream())
        return gw, None


def distortion_loss(weights, s):
    """flatten_eff_distloss (torch_efficient_distloss, called at nerf_head.py:316-327) on the dense (R, S) render weights (0 where a
    sample was culled) and the normalised sample positions s (S,): sum over rays of (1/3) sum w_i^2 / n_kept + 2 sum_i w_i (s_i
    sum_{j<i} w_j - sum_{j<i} w_j s_j), divided by 1 + the index of the last ray that kept a sample (pw_distortion_loss); autograd
    through pw_distortion_loss_backward."""
    if weights.dim() != 2 or s.numel() != weights.shape[1]:
        raise _lib.PreworldHipError('distortion_loss: weights (R, S), s (S,)')
    if weights.shape[0] == 0 or weights.shape[1] == 0:          # an empty ray batch: flatten_eff_distloss' sums are empty, the loss 0
        return weights.sum() * 0.0
    return _Distortion.apply(_chk_t(weights), _chk_t(s.reshape(-1)))


def _chk_t(t):
    return t if (t.dtype == _f32 and t.is_contiguous()) else t.float().contiguous()


def confusion_hist(pred, gt, mask, n_cl, hist):
    """hist (n_cl,n_cl) int64 += bincount(n_cl*gt + pred) over (masked) voxels -- occ_metrics.py:82-105."""
    p = pred.contiguous().view(-1)
    g = gt.contiguous().view(-1)
    m = mask.contiguous().view(-1).to(torch.uint8) if mask is not None else None
    if p.dtype != torch.uint8 or g.dtype != torch.uint8:
        raise _lib.PreworldHipError('pred/gt must be uint8')
    _lib.call('pw_confusion_hist', _chk(p, torch.uint8, 'pred'), _chk(g, torch.uint8, 'gt'), _p(m),
              p.numel(), int(n_cl), _chk(hist, _i64, 'hist'), _stream())
    return hist
