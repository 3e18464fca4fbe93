"""Functional wrappers over the C ABI for torch tensors resident in HBM.

torch is plumbing here: device memory, the current HIP stream and autograd glue.
All arithmetic happens in libpreworld_hip.so.  Names and argument order follow the
reference operators these replace (cited per function, paths relative to the reference).
"""
import ctypes

import torch

from . import _lib

_i32 = torch.int32
_f32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PreworldHipError('%s must be a CUDA(HIP) tensor' % name)
    if t.dtype != dtype:
        raise _lib.PreworldHipError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.PreworldHipError('%s must be contiguous' % name)
    return ctypes.c_void_p(t.data_ptr())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _host3(vals):
    return (ctypes.c_float * 3)(*[float(v) for v in vals])


def _workspace(nbytes, device):
    # torch's caching allocator returns >=512-byte aligned blocks
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)


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


def segment_sort(keys, n_keys):
    """Stable counting sort of indices by int32 key (key<0 dropped).
    Returns seg_start int32[n_keys+1], order int32[n]."""
    n = keys.numel()
    dev = keys.device
    nbytes = _lib.call_size('pw_segment_sort_workspace_bytes', n, n_keys)
    ws = _workspace(nbytes, dev)
    seg_start = torch.empty(n_keys + 1, device=dev, dtype=_i32)
    order = torch.empty(n, device=dev, dtype=_i32)
    _lib.call('pw_segment_sort', n, n_keys, _chk(keys, _i32, 'keys'), _p(ws), nbytes, _p(seg_start),
              _p(order), _stream())
    return seg_start, order


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


def bev_pool_dense(depth, feat, seg_start, order, n_voxels, D, HW, out=None):
    """Write-once dense pooling.  depth (B,N,D,H,W) flat, feat (B,N,H,W,C) -> (n_voxels, C)."""
    C = feat.shape[-1]
    if out is None:
        out = torch.empty(n_voxels, C, device=feat.device, dtype=_f32)
    _lib.call('pw_bev_pool_dense', _chk(depth, _f32, 'depth'), _chk(feat, _f32, 'feat'),
              _chk(seg_start, _i32, 'seg_start'), _chk(order, _i32, 'order'), n_voxels, C, D, HW,
              _chk(out, _f32, 'out'), _stream())
    return out


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
        seg_start, order = segment_sort(ranks_feat.contiguous(), n_pix)
        order = order[:ranks_feat.numel()].long()
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
