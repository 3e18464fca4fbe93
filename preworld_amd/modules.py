"""Drop-in nn.Modules mirroring the reference's registry classes for the hot path.

Same class names, constructor kwargs, forward signatures and state-dict keys as the
reference, so `configs/preworld/*.py` model dicts build them unchanged via
preworld_amd.registry.  All hot arithmetic goes through libpreworld_hip.so (preworld_amd.ops);
torch only provides parameters, device memory and the stream.  Modules raise if the HIP
library is missing or if asked to run on CPU tensors -- there is no fallback path.
"""
import os

import torch
import torch.nn as nn

from . import ops

# PW_LSS=sort keeps the sort-based lift (pw_segment_sort + pw_bev_pool_dense) on the inference path: A/B switch for
# ops.lss_lift_pool, which is the default for C == 32 (same bits either way, tests/test_gpu_lss.py)
_LSS_FORM = os.environ.get('PW_LSS', 'slots')


def create_frustum(depth_cfg, input_size, downsample):
    """mmdet3d/models/necks/view_transformer.py:84-112 (sid=False): (D,H,W,3) fp32,
    built with the same torch calls as the reference so the table is bit-identical."""
    H_in, W_in = input_size
    H_feat, W_feat = H_in // downsample, W_in // downsample
    d = torch.arange(*depth_cfg, dtype=torch.float).view(-1, 1, 1).expand(-1, H_feat, W_feat)
    D = d.shape[0]
    x = torch.linspace(0, W_in - 1, W_feat, dtype=torch.float).view(1, 1, W_feat).expand(D, H_feat, W_feat)
    y = torch.linspace(0, H_in - 1, H_feat, dtype=torch.float).view(1, H_feat, 1).expand(D, H_feat, W_feat)
    return torch.stack((x, y, d), -1).contiguous()


class LSSViewTransformer(nn.Module):
    """Lift-Splat-Shoot view transformer (BEVPoolv2) -- drop-in for
    mmdet3d/models/necks/view_transformer.py:15-291.

    forward(input) with input = [x (B,N,C_in,H,W), sensor2ego (B,N,4,4), ego2global, cam2imgs
    (B,N,3,3), post_rots (B,N,3,3), post_trans (B,N,3), bda (B,3,3)] returns
    (bev_feat (B,C,Z,Y,X) [or (B,C*Z,Y,X) with collapse_z], depth (B*N,D,H,W)).
    The returned bev_feat is a stride view of a channels-last (B,Z,Y,X,C) buffer.
    """

    def __init__(self, grid_config, input_size, downsample=16, in_channels=512, out_channels=64,
                 accelerate=False, sid=False, collapse_z=True):
        super().__init__()
        if sid:
            raise NotImplementedError('sid=True is not used by any PreWorld config')
        self.grid_config = grid_config
        self.downsample = downsample
        self.create_grid_infos(**grid_config)
        self.sid = sid
        self.input_size = input_size
        self.frustum = self.create_frustum(grid_config['depth'], input_size, downsample)
        self.out_channels = out_channels
        self.in_channels = in_channels
        self.depth_net = nn.Conv2d(in_channels, self.D + self.out_channels, kernel_size=1, padding=0)
        self.accelerate = accelerate
        self.initial_flag = True
        self.collapse_z = collapse_z
        self._cache = None

    # ---- view_transformer.py:66-82
    def create_grid_infos(self, x, y, z, **kwargs):
        self.grid_lower_bound = torch.Tensor([cfg[0] for cfg in [x, y, z]])
        self.grid_interval = torch.Tensor([cfg[2] for cfg in [x, y, z]])
        self.grid_size = torch.Tensor([(cfg[1] - cfg[0]) / cfg[2] for cfg in [x, y, z]])

    # ---- view_transformer.py:84-112
    def create_frustum(self, depth_cfg, input_size, downsample):
        fr = create_frustum(depth_cfg, input_size, downsample)
        self.D = fr.shape[0]
        return fr

    def _grid(self):
        key = tuple((id(t), t._version, t.data_ptr()) for t in (self.grid_lower_bound, self.grid_interval, self.grid_size))
        if getattr(self, '_grid_host', (None,))[0] != key:                 # host copies of the three 3-vectors, made once
            self._grid_host = (key, ([float(v) for v in self.grid_lower_bound], [float(v) for v in self.grid_interval],
                                     [int(v) for v in self.grid_size]))
        return self._grid_host[1]

    def _frustum_on(self, ref):
        if self.frustum.device != ref.device:
            self.frustum = self.frustum.to(ref.device)       # plain attribute, like the reference
        return self.frustum

    # ---- view_transformer.py:114-153
    def get_lidar_coor(self, sensor2ego, ego2global, cam2imgs, post_rots, post_trans, bda):
        B, N = sensor2ego.shape[:2]
        lower, interval, size = self._grid()
        ipr, comb, tr = ops.lss_camera_matrices(sensor2ego, cam2imgs, post_rots)
        _, coor = ops.lss_voxel_index(self._frustum_on(sensor2ego), ipr,
                                      post_trans.contiguous().float(), comb, tr,
                                      bda.contiguous().float(), lower, interval, size, B, N,
                                      return_coor=True)
        return coor

    def _sort(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        """geometry -> voxel ids -> device counting sort.  With accelerate=True the result is kept and reused for as
        long as the camera tensors are the SAME tensors with unchanged contents (identity + torch version counters;
        the reference's accelerate flag, view_transformer.py:155-174,263-267, assumes a fixed rig and never checks)."""
        key = None
        if self.accelerate:
            tensors = (sensor2ego, cam2imgs, post_rots, post_trans, bda)
            # identity + version counter of the tensor OBJECTS, which the cache entry keeps alive: a freed tensor's address can be
            # handed to a new one by the caching allocator (with _version 0 again), so addresses alone would alias
            # (+ the data pointer: `t.data = other` swaps the storage without touching id or _version)
            key = tuple((id(t), t._version, t.data_ptr()) for t in tensors)
            if self._cache is not None and self._cache[0] == key:
                return self._cache[2]
        vs = self._sort_now(sensor2ego, cam2imgs, post_rots, post_trans, bda)
        if key is not None:
            self._cache = (key, tensors, vs)
        return vs

    def _sort_now(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        B, N = sensor2ego.shape[:2]
        lower, interval, size = self._grid()
        ipr, comb, tr = ops.lss_camera_matrices(sensor2ego, cam2imgs, post_rots)
        vox = ops.lss_voxel_index(self._frustum_on(sensor2ego), ipr, post_trans.contiguous().float(),
                                  comb, tr, bda.contiguous().float(), lower, interval, size, B, N)
        n_vox = B * size[0] * size[1] * size[2]
        H, W = self.frustum.shape[1:3]
        return ops.segment_sort(vox, n_vox, aux_div=self.D * H * W, aux_mod=H * W,
                                long_threshold=ops.LONG_SEGMENT)

    # ---- view_transformer.py:203-261 (takes the camera tensors instead of coor: the 17.8 MB
    # coordinate tensor is never materialised; use get_lidar_coor() if you need it)
    def voxel_pooling_prepare_v2(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        vs = self._sort(sensor2ego, cam2imgs, post_rots, post_trans, bda)
        H, W = self.frustum.shape[1:3]
        return ops.lss_ranks(vs.seg_start, vs.order, vs.n_keys, self.D, H * W)

    # ---- view_transformer.py:155-174,263-267
    def init_acceleration_v2(self, input):
        self._sort(input[1], input[3], input[4], input[5], input[6])

    def pre_compute(self, input):
        if self.initial_flag:
            self.init_acceleration_v2(input)
            self.initial_flag = False

    # ---- view_transformer.py:176-201,269-291
    def view_transform_core(self, input, depth, tran_feat):
        B, N, C, H, W = input[0].shape
        _, _, size = self._grid()
        if getattr(tran_feat, '_pw_channels_last', False):        # already (B*N,H,W,C) from ops.depthnet_tail
            feat = tran_feat.view(B, N, H, W, self.out_channels)
        else:
            feat = tran_feat.view(B, N, self.out_channels, H, W).permute(0, 1, 3, 4, 2).contiguous().float()
        dep = depth.view(B, N, self.D, H, W).contiguous().float()
        train = (dep.requires_grad or feat.requires_grad) and torch.is_grad_enabled()
        if not train and not self.accelerate and self.out_channels == 32 and _LSS_FORM != 'sort':
            # one frame's lift + pooling without a sort (ops.lss_lift_pool): same bits, 5 launches
            lower, interval, _ = self._grid()
            out = ops.lss_lift_pool(self._frustum_on(input[1]), input[1], input[3], input[4], input[5], input[6], lower,
                                    interval, size, dep, feat, out_h2=getattr(self, '_pool_h2', False))
            if isinstance(out, ops.H2):
                self._pool_rng, out = out.rng, out.buf
            bev = out.view(B, size[2], size[1], size[0], self.out_channels).permute(0, 4, 1, 2, 3)
            if self.collapse_z:
                bev = torch.cat(bev.unbind(dim=2), 1)
            return bev, depth
        vs = self._sort(input[1], input[3], input[4], input[5], input[6])
        seg_start, order, n_vox = vs.seg_start, vs.order, vs.n_keys
        if train:
            rb, rd, rf, st, ln = ops.lss_ranks(seg_start, order, n_vox, self.D, H * W)
            if rb is None:
                out = feat.new_zeros(B, size[2], size[1], size[0], self.out_channels)
                bev = out.permute(0, 4, 1, 2, 3)
            else:
                bev = ops.bev_pool_v2(dep, feat, rd, rf, rb,
                                      (B, size[2], size[1], size[0], self.out_channels), st, ln)
        else:
            out = ops.bev_pool_dense(dep, feat, vs, out_h2=getattr(self, '_pool_h2', False))
            if isinstance(out, ops.H2):                   # pool_cl re-wraps the buffer under this range slot
                self._pool_rng, out = out.rng, out.buf
            bev = out.view(B, size[2], size[1], size[0], self.out_channels).permute(0, 4, 1, 2, 3)
        if self.collapse_z:
            bev = torch.cat(bev.unbind(dim=2), 1)
        return bev, depth

    def view_transform(self, input, depth, tran_feat):
        if self.accelerate:
            self.pre_compute(input)
        return self.view_transform_core(input, depth, tran_feat)

    def pool_cl(self, input, depth, tran_feat, out_h2=False):
        """view_transform without the (B,C,Z,Y,X) view: the pooled channels-last (B,Z,Y,X,C) buffer itself, as fp32 or
        (out_h2, inference, C % 32 == 0) as ops.H2 -- the same fp32 sums written in the split-fp16 storage the encoder's
        first convolution consumes, so no conversion pass runs between pooling and pre_process."""
        keep, self.collapse_z = self.collapse_z, False
        self._pool_h2 = bool(out_h2) and not (torch.is_grad_enabled() and (depth.requires_grad or tran_feat.requires_grad))
        try:
            bev, _ = self.view_transform(input, depth, tran_feat)
        finally:
            self.collapse_z = keep
            h2, self._pool_h2 = self._pool_h2, False
        x = to_channels_last_3d(bev)
        return ops.H2(x, self.__dict__.pop('_pool_rng', None)) if h2 else x

    def forward(self, input):
        x = input[0]
        B, N, C, H, W = x.shape
        x = x.view(B * N, C, H, W)
        x = self.depth_net(x)
        depth, tran_feat = self.depthnet_tail(x)
        return self.view_transform(input, depth, tran_feat)

    def depthnet_tail(self, x):
        """view_transformer.py:797-801: split the DepthNet output, softmax the depth logits; the
        context comes back channels-last (what the pooling gathers), tagged so that
        view_transform_core skips its permute copy."""
        depth, feat = ops.depthnet_tail(x.float().contiguous(), self.D, self.out_channels)
        feat._pw_channels_last = True
        return depth, feat



class LSSViewTransformerBEVDepth(LSSViewTransformer):
    """Drop-in for view_transformer.py:702-804: the view transformer the PreWorld configs reach through
    `type='LSSViewTransformerBEVStereo'`.  `depth_net` is the DepthNet of image_encoder.py (plain PyTorch-ROCm, as
    north_star prescribes for the image side; its stereo cost volume and its softmax/split tail are HIP kernels), the
    lifting + pooling is the HIP path of the base class.  State-dict keys `depth_net.*` equal the reference's."""

    def __init__(self, loss_depth_weight=3.0, depthnet_cfg=dict(), **kwargs):
        super().__init__(**kwargs)
        from .image_encoder import DepthNet
        self.loss_depth_weight = loss_depth_weight
        self.depth_net = DepthNet(self.in_channels, self.in_channels, self.out_channels, self.D, **depthnet_cfg)

    # ---- view_transformer.py:713-734
    def get_mlp_input(self, sensor2ego, ego2global, intrin, post_rot, post_tran, bda):
        from .image_encoder import get_mlp_input
        return get_mlp_input(sensor2ego, ego2global, intrin, post_rot, post_tran, bda)

    # ---- view_transformer.py:736-773
    def get_downsampled_gt_depth(self, gt_depths):
        """gt_depths (B,N,H,W) -> one-hot (B*N*h*w, D): minimum non-zero depth of every downsample x downsample patch,
        binned with the depth grid (sid=False)."""
        B, N, H, W = gt_depths.shape
        ds = self.downsample
        g = gt_depths.view(B * N, H // ds, ds, W // ds, ds, 1).permute(0, 1, 3, 5, 2, 4).contiguous().view(-1, ds * ds)
        g = torch.where(g == 0.0, 1e5 * torch.ones_like(g), g).min(dim=-1).values.view(B * N, H // ds, W // ds)
        d0, _, dstep = self.grid_config['depth']
        g = (g - (d0 - dstep)) / dstep
        g = torch.where((g < self.D + 1) & (g >= 0.0), g, torch.zeros_like(g))
        return torch.nn.functional.one_hot(g.long(), num_classes=self.D + 1).view(-1, self.D + 1)[:, 1:].float()

    # ---- view_transformer.py:775-789
    def get_depth_loss(self, depth_labels, depth_preds):
        depth_labels = self.get_downsampled_gt_depth(depth_labels)
        depth_preds = depth_preds.permute(0, 2, 3, 1).contiguous().view(-1, self.D)
        fg_mask = torch.max(depth_labels, dim=1).values > 0.0
        loss = torch.nn.functional.binary_cross_entropy(depth_preds[fg_mask], depth_labels[fg_mask], reduction='none')
        return self.loss_depth_weight * (loss.sum() / max(1.0, fg_mask.sum()))

    # ---- view_transformer.py:791-804
    def forward(self, input, stereo_metas=None, depth_gt=None):
        x, mlp_input = input[0], input[7]
        B, N, C, H, W = x.shape
        x = self.depth_net(x.view(B * N, C, H, W), mlp_input, stereo_metas)
        if torch.is_grad_enabled() and x.requires_grad:
            # training: keep the softmax/split in autograd; pooling goes through ops.bev_pool_v2 (QuickCumsumCuda)
            depth = x[:, :self.D].softmax(dim=1)
            tran_feat = x[:, self.D:self.D + self.out_channels]
        else:
            depth, tran_feat = self.depthnet_tail(x)
        return self.view_transform(input, depth, tran_feat)


class LSSViewTransformerBEVStereo(LSSViewTransformerBEVDepth):
    """view_transformer.py:807-813: adds the 1/4-resolution frustum of the stereo cost volume."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.cv_frustum = create_frustum(kwargs['grid_config']['depth'], kwargs['input_size'], 4)


# =====================================================================================
# voxel encoder: ConvModule / BasicBlock3D / CustomResNet3D / LSSFPN3D
# =====================================================================================
def to_channels_last_3d(x):
    """(B,C,D,H,W) logical tensor -> contiguous channels-last storage (B,D,H,W,C).
    Free when x is already a view of such a buffer (everything this package returns is)."""
    y = x.permute(0, 2, 3, 4, 1)
    return y if y.is_contiguous() else y.contiguous()


def from_channels_last_3d(y):
    """(B,D,H,W,C) storage -> (B,C,D,H,W) view (what the reference's callers index)."""
    return y.permute(0, 4, 1, 2, 3)


def precision():
    """'h2' (default): the voxel encoder's convolutions run on the fp16 matrix cores with split-fp16 operands
    (ops.conv3d_h2: x = hi + lo, three MFMAs per product block, fp32 accumulate -- as accurate as an fp32 FMA chain,
    DESIGN.md section 5.2) and activations travel between them in h2 storage; 'f32' (PW_PRECISION=f32): the exact-fp32
    MFMA kernels (Winograd / direct) of round 1."""
    import os
    return os.environ.get('PW_PRECISION', 'h2')


def as_h2(x):
    """fp32 channels-last tensor or ops.H2 -> ops.H2 (one conversion pass if needed)"""
    return x if isinstance(x, ops.H2) else ops.f32_to_h2(x.contiguous())


def as_f32(x):
    return ops.h2_to_f32(x) if isinstance(x, ops.H2) else x


def _use_wino(x_cl, cout_total, ksize, stride):
    """3x3x3 stride-1 convs whose (4x8x8 tile, 32-column group) work items number >= 128 run on the Winograd
    F(2x2x2,3x3x3) kernel (pw_conv3d_wino: 3.4x fewer multiplies, 1.7-2.2x faster than the direct MFMA
    kernels at the C3 shapes); smaller grids and everything else: the direct kernels (ops.conv3d_ndhwc)."""
    if ksize != 3 or stride != 1 or cout_total % 32:
        return False
    B, D, H, W, _ = x_cl.shape
    return B * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8) * (cout_total // 32) >= 128


class _PackedCache:
    """Packed/folded weights are derived data: rebuilt lazily when parameters change."""

    def __init__(self):
        self._key = None
        self._val = None
        self._live = None

    def get(self, tensors, build):
        # identity + version of the parameter OBJECTS (kept alive by the entry; see LSSViewTransformer._sort for why not addresses)
        live = [t for t in tensors if t is not None]
        key = tuple((id(t), t._version, t.data_ptr(), t.device) for t in live)
        if key != self._key:
            with torch.no_grad():
                self._val = build()
            self._key, self._live = key, live
        return self._val


class ConvModule3d(nn.Module):
    """mmcv ConvModule restricted to what the hot path uses (conv_cfg=Conv3d, norm_cfg=BN3d or
    None, act ReLU or None; order conv -> norm -> act).  Child names `conv` / `bn` keep the
    reference's state-dict keys (`...conv.weight`, `...bn.running_mean`, ...)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type') == 'Conv3d'
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, bias=bias)
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            assert norm_cfg['type'] in ('BN3d', 'SyncBN', 'BN')
            self.bn = nn.BatchNorm3d(out_channels)
            self.bn.pw_sync = norm_cfg['type'] == 'SyncBN'        # training: statistics over all ranks (train.BatchNormCL)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        assert padding == kernel_size // 2
        self._cache = _PackedCache()
        self._wcache = _PackedCache()

    def folded(self):
        """(packed weight, scale[cout32], bias[cout32]) for the HIP conv."""
        params = [self.conv.weight, self.conv.bias]
        if self.with_norm:
            params += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]

        def build():
            wpk = ops.pack_conv_weight(self.conv.weight)
            if self.with_norm:
                sc, bi = ops.fold_bn(self.bn.weight, self.bn.bias, self.bn.running_mean,
                                     self.bn.running_var, self.bn.eps, self.conv.bias)
            else:
                sc = torch.ones_like(self.conv.weight[:, 0, 0, 0, 0])
                bi = self.conv.bias.float() if self.conv.bias is not None else torch.zeros_like(sc)
            return wpk, ops._pad32(sc, 1.0), ops._pad32(bi, 0.0)
        return self._cache.get(params, build)

    def folded_h2(self):
        """(split-fp16 packed weight, scale * weight pre-scale, bias) for ops.conv3d_h2."""
        params = [self.conv.weight, self.conv.bias]
        if self.with_norm:
            params += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        if not hasattr(self, '_h2cache'):
            self._h2cache = _PackedCache()

        def build():
            _, sc, bi = self.folded()
            wpk, inv = ops.pack_conv_weight_h2(self.conv.weight)
            return wpk, (sc * inv).contiguous(), bi
        return self._h2cache.get(params, build)

    def _check_eval(self):
        if self.training and self.with_norm:
            raise NotImplementedError('this fused HIP path folds BatchNorm (eval mode); training mode goes through '
                                      'preworld_amd.train (ConvModule3d.forward_cl / BasicBlock3D.forward_cl dispatch there)')

    def forward_cl(self, x_cl, residual=None, algo=0, out_h2=False):
        """channels-last in (fp32 tensor or ops.H2) -> channels-last out (fp32, or ops.H2 with out_h2).
        In training mode (BatchNorm with batch statistics, autograd): preworld_amd.train, fp32."""
        if self.training and self.with_norm:                # fp32 tensors under autograd (out_h2 does not apply)
            from . import train
            return train.conv_module_forward(self, as_f32(x_cl), residual=as_f32(residual) if residual is not None else None)
        if self.training and torch.is_grad_enabled() and residual is None:     # conv + bias + act without norm (final_conv)
            from . import train
            return train.conv_bias_act_forward(self, as_f32(x_cl))
        self._check_eval()
        if precision() == 'h2' and self.kernel_size in (1, 3) and self.out_channels % 32 == 0 and algo in (0, 2, 3):
            wpk, sc, bi = self.folded_h2()
            return ops.conv3d_h2(as_h2(x_cl), wpk, sc, bi, residual=residual, cout0=self.out_channels,
                                 relu0=self.with_activation, ksize=self.kernel_size, stride=self.stride, algo=algo,
                                 out_h2=(out_h2, out_h2))
        x_cl, residual = as_f32(x_cl), (as_f32(residual) if residual is not None else None)
        y = self._forward_cl_f32(x_cl, residual, algo)
        return as_h2(y) if out_h2 else y

    def _forward_cl_f32(self, x_cl, residual=None, algo=0):
        wpk, sc, bi = self.folded()
        if algo == 0 and _use_wino(x_cl, sc.numel(), self.kernel_size, self.stride):
            uw = self._wcache.get([self.conv.weight], lambda: ops.pack_conv_weight_wino(self.conv.weight))
            return ops.conv3d_wino(x_cl, uw, sc, bi, residual=residual, cout0=self.out_channels,
                                   relu0=self.with_activation)
        return ops.conv3d_ndhwc(x_cl, wpk, sc, bi, residual=residual, cout0=self.out_channels,
                                ksize=self.kernel_size, stride=self.stride,
                                relu0=self.with_activation, algo=algo)

    def forward(self, x):
        return from_channels_last_3d(self.forward_cl(to_channels_last_3d(x)))


class BasicBlock3D(nn.Module):
    """mmdet3d/models/backbones/resnet.py:88-123: relu(conv2(conv1(x)) + downsample(x)).
    conv1 and downsample read the same input, so they run as ONE conv with 2*Cout columns."""

    def __init__(self, channels_in, channels_out, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvModule3d(channels_in, channels_out, 3, stride=stride, padding=1, bias=False,
                                  conv_cfg=dict(type='Conv3d'), norm_cfg=dict(type='BN3d'),
                                  act_cfg=dict(type='ReLU', inplace=True))
        self.conv2 = ConvModule3d(channels_out, channels_out, 3, stride=1, padding=1, bias=False,
                                  conv_cfg=dict(type='Conv3d'), norm_cfg=dict(type='BN3d'),
                                  act_cfg=None)
        self.downsample = downsample
        self._cache = _PackedCache()

    def forward_cl(self, x, out=None, out_h2=False):
        """x: fp32 channels-last tensor or ops.H2.  out: optional destination (a channel slice of a wider channels-last
        buffer is fine; an ops.H2 when out_h2); the downsample branch is then written there first and conv2 adds onto it
        in place.  Returns fp32, or ops.H2 with out_h2.  Training mode: preworld_amd.train (fp32, autograd)."""
        if self.training:                                  # fp32 tensors under autograd; h2 storage is an inference format
            from . import train
            if out is not None:
                raise NotImplementedError('BasicBlock3D training path returns a new fp32 tensor (no preallocated destination)')
            return train.basic_block_forward(self, as_f32(x))
        if precision() == 'h2':
            return self._forward_cl_h2(x, out, out_h2)
        x = as_f32(x)
        if out_h2:
            return as_h2(self._forward_cl_f32(x, None)) if out is None else self._forward_f32_into_h2(x, out)
        return self._forward_cl_f32(x, out.buf if isinstance(out, ops.H2) else out)

    def _forward_f32_into_h2(self, x, out):
        y = self._forward_cl_f32(x, None)
        ops.f32_to_h2(y, out=out.buf)
        return out

    def _forward_cl_h2(self, x, out, out_h2):
        """split-fp16 path: conv1 + downsample as ONE pass over x (N = 2 x Cout), everything between the convs in h2.  An
        ops.H2 destination keeps its range slot: the identity branch and the block output are both written under it."""
        c1, c2, ds = self.conv1, self.conv2, self.downsample
        c1._check_eval()
        x = as_h2(x)
        if ds is not None:
            params = [c1.conv.weight, c1.bn.weight, c1.bn.bias, c1.bn.running_mean, c1.bn.running_var,
                      ds.conv.weight, ds.bn.weight, ds.bn.bias, ds.bn.running_mean, ds.bn.running_var]
            if not hasattr(self, '_h2cache'):
                self._h2cache = _PackedCache()

            def build():
                w1, s1, b1 = c1.folded_h2()
                wd, sd, bd = ds.folded_h2()
                return (torch.cat([w1, wd], dim=2).contiguous(), torch.cat([s1, sd]).contiguous(),
                        torch.cat([b1, bd]).contiguous())
            wpk, sc, bi = self._h2cache.get(params, build)
            y, identity = ops.conv3d_h2(x, wpk, sc, bi, cout0=c1.out_channels, cout1=ds.out_channels, relu0=True, relu1=False,
                                        out1=out, ksize=3, stride=c1.stride, out_h2=(True, out_h2))
        else:
            w1, s1, b1 = c1.folded_h2()
            y = ops.conv3d_h2(x, w1, s1, b1, cout0=c1.out_channels, relu0=True, ksize=3, stride=c1.stride, out_h2=(True, True))
            w2, s2, b2 = c2.folded_h2()
            if out is None:                                   # identity = the block input itself, result in a new buffer
                return ops.conv3d_h2(y, w2, s2, b2, residual=x, cout0=c2.out_channels, relu0=True, out_h2=(out_h2, out_h2))
            if out_h2:                                        # re-express x under the destination's exponent
                identity = ops.f32_to_h2(ops.h2_to_f32(x), out=out)
            else:
                identity = ops.h2_to_f32(x, out=out.buf if isinstance(out, ops.H2) else out)
        w2, s2, b2 = c2.folded_h2()
        return ops.conv3d_h2(y, w2, s2, b2, residual=identity, cout0=c2.out_channels, relu0=True, out0=identity,
                             out_h2=(out_h2, out_h2))

    def _forward_cl_f32(self, x, out=None):
        """exact-fp32 path (PW_PRECISION=f32): Winograd / direct MFMA kernels"""
        c1, c2, ds = self.conv1, self.conv2, self.downsample
        c1._check_eval()
        if ds is not None:
            params = [c1.conv.weight, c1.bn.weight, c1.bn.bias, c1.bn.running_mean, c1.bn.running_var,
                      ds.conv.weight, ds.bn.weight, ds.bn.bias, ds.bn.running_mean, ds.bn.running_var]

            def build():
                w1, s1, b1 = c1.folded()
                wd, sd, bd = ds.folded()
                return (torch.cat([w1, wd], dim=2).contiguous(), torch.cat([s1, sd]).contiguous(),
                        torch.cat([b1, bd]).contiguous())
            wpk, sc, bi = self._cache.get(params, build)
            if _use_wino(x, sc.numel(), 3, c1.stride):
                if not hasattr(self, '_wcache'):
                    self._wcache = _PackedCache()
                uw = self._wcache.get([c1.conv.weight, ds.conv.weight],
                                      lambda: ops.pack_conv_weights_wino_concat([c1.conv.weight, ds.conv.weight]))
                y, identity = ops.conv3d_wino(x, uw, sc, bi, cout0=c1.out_channels, cout1=ds.out_channels,
                                              relu0=True, relu1=False, out1=out)
            else:
                y, identity = ops.conv3d_ndhwc(x, wpk, sc, bi, cout0=c1.out_channels,
                                               cout1=ds.out_channels, ksize=3, stride=c1.stride,
                                               relu0=True, relu1=False, out1=out)
        else:
            identity = x
            y = c1._forward_cl_f32(x)
        w2, s2, b2 = c2.folded()
        if out is not None and ds is None:
            out.copy_(identity)
            identity = out
        if _use_wino(y, s2.numel(), 3, 1):
            uw2 = c2._wcache.get([c2.conv.weight], lambda: ops.pack_conv_weight_wino(c2.conv.weight))
            return ops.conv3d_wino(y, uw2, s2, b2, residual=identity, cout0=c2.out_channels, relu0=True, out0=out)
        return ops.conv3d_ndhwc(y, w2, s2, b2, residual=identity, cout0=c2.out_channels, ksize=3,
                                stride=1, relu0=True, out0=out)

    def forward(self, x):
        return from_channels_last_3d(self.forward_cl(to_channels_last_3d(x)))


class CustomResNet3D(nn.Module):
    """mmdet3d/models/backbones/resnet.py:126-184 (same kwargs, same `layers.{s}.{b}` keys)."""

    def __init__(self, numC_input, num_layer=[2, 2, 2], num_channels=None, stride=[2, 2, 2],
                 backbone_output_ids=None, with_cp=False):
        super().__init__()
        assert len(num_layer) == len(stride)
        num_channels = [numC_input * 2 ** (i + 1) for i in range(len(num_layer))] \
            if num_channels is None else num_channels
        self.backbone_output_ids = range(len(num_layer)) \
            if backbone_output_ids is None else backbone_output_ids
        layers = []
        curr = numC_input
        for i in range(len(num_layer)):
            layer = [BasicBlock3D(curr, num_channels[i], stride=stride[i],
                                  downsample=ConvModule3d(curr, num_channels[i], 3, stride=stride[i],
                                                          padding=1, bias=False,
                                                          conv_cfg=dict(type='Conv3d'),
                                                          norm_cfg=dict(type='BN3d'), act_cfg=None))]
            curr = num_channels[i]
            layer.extend([BasicBlock3D(curr, curr) for _ in range(num_layer[i] - 1)])
            layers.append(nn.Sequential(*layer))
        self.layers = nn.Sequential(*layers)
        self.with_cp = with_cp

    def forward_cl(self, x, out_last=None, keep_h2=False):
        """x: fp32 channels-last tensor or ops.H2.  out_last: optional destination of the LAST block's output (see
        BasicBlock3D.forward_cl).  keep_h2: return the stage outputs as ops.H2 (what the next split-fp16 kernel consumes)
        instead of fp32 tensors.  In the default 'h2' precision every block-to-block tensor stays in h2 storage."""
        feats = []
        n_layers = len(self.layers)
        h2 = precision() == 'h2' and not self.training
        for lid, layer in enumerate(self.layers):
            for bid, blk in enumerate(layer):
                last = out_last is not None and lid == n_layers - 1 and bid == len(layer) - 1
                x = blk.forward_cl(x, out=out_last if last else None,
                                   out_h2=isinstance(out_last, ops.H2) if last else h2)
            if lid in self.backbone_output_ids:
                feats.append(x)
        return [as_h2(f) if (keep_h2 and not self.training) else as_f32(f) for f in feats]

    def forward(self, x):
        return [from_channels_last_3d(f) for f in self.forward_cl(to_channels_last_3d(x))]


class LSSFPN3D(nn.Module):
    """mmdet3d/models/necks/lss_fpn.py:103-148 (levels=3).  The 1x1x1 conv is applied before the
    trilinear upsampling (they commute), so neither the upsampled maps nor the 224-channel
    concat are ever materialised -- see pw_fpn3d_fuse in include/preworld_hip.h."""

    def __init__(self, in_channels, out_channels, levels=3, with_cp=False):
        super().__init__()
        assert levels == 3, 'PreWorld configs use levels=3'
        assert out_channels == 32, 'fused neck kernel is built for out_channels=32'
        self.levels = levels
        self.conv = ConvModule3d(in_channels, out_channels, 1, stride=1, padding=0, bias=False,
                                 conv_cfg=dict(type='Conv3d'), norm_cfg=dict(type='BN3d'),
                                 act_cfg=dict(type='ReLU', inplace=True))
        self.with_cp = with_cp
        self._cache = _PackedCache()

    def forward_cl(self, feats, out_h2=False):
        """feats: fp32 channels-last tensors or ops.H2 -> fp32 tensor (or ops.H2 with out_h2)"""
        if self.training:                                  # fp32 under autograd (preworld_amd.train)
            from . import train
            return train.fpn_forward(self, [as_f32(f) for f in feats])
        x8, x16, x32 = feats
        c8, c16, c32 = x8.shape[-1], x16.shape[-1], x32.shape[-1]
        cm = self.conv
        cm._check_eval()
        params = [cm.conv.weight, cm.bn.weight, cm.bn.bias, cm.bn.running_mean, cm.bn.running_var]
        w = cm.conv.weight
        assert w.shape[1] == c8 + c16 + c32
        if precision() == 'h2':
            if not hasattr(self, '_h2cache'):
                self._h2cache = _PackedCache()

            def build_h2():
                sc, bi = ops.fold_bn(cm.bn.weight, cm.bn.bias, cm.bn.running_mean, cm.bn.running_var, cm.bn.eps)
                w8, i8 = ops.pack_conv_weight_h2(w[:, :c8].contiguous())
                w16, i16 = ops.pack_conv_weight_h2(w[:, c8:c8 + c16].contiguous())
                w32, i32 = ops.pack_conv_weight_h2(w[:, c8 + c16:].contiguous())
                # the three partial sums share ONE BN scale: y16 / y32 are scaled by inv16 / inv8 and inv32 / inv8 so that
                # the fused kernel can apply scale * inv8 to the sum (all factors are powers of two: exact)
                return w8, w16, w32, (i16 / i8).contiguous(), (i32 / i8).contiguous(), (sc * i8).contiguous(), bi
            w8, w16, w32, r16, r32, sc8, bi = self._h2cache.get(params, build_h2)
            y16 = ops.conv3d_h2(as_h2(x16), w16, r16, ksize=1, out_h2=(False, False))
            y32 = ops.conv3d_h2(as_h2(x32), w32, r32, ksize=1, out_h2=(False, False))
            return ops.fpn3d_fuse(as_h2(x8), w8, y16, y32, sc8, bi, relu=True, out_h2=out_h2)

        def build():
            sc, bi = ops.fold_bn(cm.bn.weight, cm.bn.bias, cm.bn.running_mean, cm.bn.running_var,
                                 cm.bn.eps)
            return (ops.pack_conv_weight(w[:, :c8].contiguous()),
                    ops.pack_conv_weight(w[:, c8:c8 + c16].contiguous()),
                    ops.pack_conv_weight(w[:, c8 + c16:].contiguous()), sc, bi)
        w8, w16, w32, sc, bi = self._cache.get(params, build)
        x8, x16, x32 = as_f32(x8), as_f32(x16), as_f32(x32)
        y16 = ops.conv3d_ndhwc(x16, w16, ksize=1)
        y32 = ops.conv3d_ndhwc(x32, w32, ksize=1)
        y = ops.fpn3d_fuse(x8, w8, y16, y32, sc, bi, relu=True)
        return as_h2(y) if out_h2 else y

    def forward(self, feats):
        return from_channels_last_3d(self.forward_cl([to_channels_last_3d(f) for f in feats]))


# =====================================================================================
# heads
# =====================================================================================
class OccHead(nn.Module):
    """mmdet3d/models/heads/occupancy_head.py:45-177 for the configuration PreWorld uses
    (num_level=1, use_deblock=False, soft_weights=True): with one level the soft-weight branch
    is softmax over a single channel (== 1) times a same-size interpolate (== identity), so it
    does not change the logits; its parameters are kept so checkpoints load.

    forward(voxel_feats=[(1,C,X,Y,Z)]) -> {'output_voxels': [(1,18,X,Y,Z)]} like the reference;
    decode(...) returns the uint8 argmax directly from the fused kernel."""

    def __init__(self, in_channels, out_channel, num_level=1, soft_weights=False,
                 conv_cfg=dict(type='Conv3d', bias=False),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 point_cloud_range=[-40., -40., -1., 40., 40., 5.4], final_occ_size=[200, 200, 16],
                 empty_idx=17, balance_cls_weight=True, with_cp=False, use_deblock=False):
        super().__init__()
        if type(in_channels) is not list:
            in_channels = [in_channels]
        assert num_level == 1 and not use_deblock, 'PreWorld uses num_level=1, use_deblock=False'
        assert norm_cfg['type'] in ('SyncBN', 'BN3d', 'BN'), 'PreWorld uses SyncBN (eval == BN)'
        assert conv_cfg.get('bias', False) is False
        self.in_channels, self.out_channel, self.num_level = in_channels, out_channel, num_level
        self.with_cp, self.use_deblock, self.empty_idx = with_cp, use_deblock, empty_idx
        mid = in_channels[0] // 2
        self.occ_convs = nn.ModuleList([nn.Sequential(
            nn.Conv3d(in_channels[0], mid, 3, stride=1, padding=1, bias=False),
            nn.BatchNorm3d(mid), nn.ReLU(inplace=True))])
        self.occ_pred_conv = nn.Sequential(
            nn.Conv3d(mid, mid // 2, 1, bias=False), nn.BatchNorm3d(mid // 2), nn.ReLU(inplace=True),
            nn.Conv3d(mid // 2, out_channel, 1, bias=False))
        self.soft_weights = soft_weights
        self.num_point_sampling_feat = num_level
        self.norm_type = norm_cfg['type']
        if soft_weights:
            self.voxel_soft_weights = nn.Sequential(
                nn.Conv3d(mid, mid // 2, 1, bias=False), nn.BatchNorm3d(mid // 2),
                nn.ReLU(inplace=True), nn.Conv3d(mid // 2, self.num_point_sampling_feat, 1, bias=False))
        # norm_cfg type 'SyncBN' (every PreWorld config: preworld-7frame-finetune.py:39): in a multi-process training run the batch
        # statistics and the backward sums of these layers are all-reduced over the ranks (train.BatchNormCL, sync=True); eval mode
        # is plain BatchNorm with the running statistics either way
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.pw_sync = norm_cfg['type'] == 'SyncBN'
        self._cache = _PackedCache()

    def _folded(self, transposed=False, wino=False):
        c0, bn0 = self.occ_convs[0][0], self.occ_convs[0][1]
        c1, bn1, c2 = self.occ_pred_conv[0], self.occ_pred_conv[1], self.occ_pred_conv[3]
        params = [c0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, c1.weight,
                  bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, c2.weight]

        def build():
            s0, b0 = ops.fold_bn(bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, bn0.eps)
            s1, b1 = ops.fold_bn(bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps)
            w0 = c0.weight
            w0t = w0.permute(0, 1, 4, 3, 2).contiguous()
            return ((ops.pack_conv_weight16(w0), ops.pack_conv_weight16(w0t)),
                    (ops.pack_conv_weight_wino(w0, cout_total=16), ops.pack_conv_weight_wino(w0t, cout_total=16)),
                    ops._pad32(s0, 1.0), ops._pad32(b0, 0.0),
                    c1.weight.reshape(c1.weight.shape[0], -1).float().contiguous(), s1, b1,
                    c2.weight.reshape(c2.weight.shape[0], -1).float().contiguous())
        wpk, uwpk, s0, b0, w1, s1, b1, w2 = self._cache.get(params, build)
        return (uwpk if wino else wpk)[1 if transposed else 0], s0, b0, w1, s1, b1, w2

    def _folded_h2(self, transposed=False):
        """operands of ops.occ_head_h2: split-fp16 conv weights (both tap orders), folded BN with the weights' pre-scale
        divided out, the packed 16->8->18 tail"""
        c0, bn0 = self.occ_convs[0][0], self.occ_convs[0][1]
        c1, bn1, c2 = self.occ_pred_conv[0], self.occ_pred_conv[1], self.occ_pred_conv[3]
        params = [c0.weight, bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, c1.weight,
                  bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, c2.weight]

        def build():
            s0, b0 = ops.fold_bn(bn0.weight, bn0.bias, bn0.running_mean, bn0.running_var, bn0.eps)
            s1, b1 = ops.fold_bn(bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.eps)
            packs = []
            for w in (c0.weight, c0.weight.permute(0, 1, 4, 3, 2).contiguous()):
                wpk, inv = ops.pack_occ_weight_h2(w.float())
                packs.append((wpk, (s0 * inv).contiguous()))
            tailpk, inv2 = ops.pack_occ_tail_h2(c1.weight.reshape(c1.weight.shape[0], -1).float(), s1, b1,
                                                c2.weight.reshape(c2.weight.shape[0], -1).float())
            bounds = ops.occ_head_bounds(c0.weight, s0, b0, c1.weight.reshape(c1.weight.shape[0], -1), s1, b1)
            return packs, b0.contiguous(), tailpk, inv2, bounds
        if not hasattr(self, '_cache_h2'):
            self._cache_h2 = _PackedCache()
        packs, b0, tailpk, inv2, bounds = self._cache_h2.get(params, build)
        wpk, s0 = packs[1 if transposed else 0]
        return wpk, s0, b0, tailpk, inv2, bounds

    def decode_cl(self, x_cl, want_logits=False, transposed=False, want_geo=False, occ_out=None, geo_out=None):
        """x_cl (B,D,H,W,C) channels-last (fp32 tensor or ops.H2) -> uint8 argmax (B,D,H,W) [, logits (B,D,H,W,18)].
        The reference feeds (1,C,X,Y,Z), i.e. kernel axes (kD,kH,kW) <-> (X,Y,Z).  With
        transposed=True, x_cl is the encoder's native (B,Z,Y,X,C) buffer and the kernel taps are
        permuted instead of the 82 MB activation; the result is then the
        (Z,Y,X) array whose .permute(0,3,2,1) view is the reference's (X,Y,Z) output.  occ_out / geo_out (split-fp16 path only):
        uint8 (B,D,H,W) destinations of any strides, written in place (ops.occ_head_h2)."""
        if self.training:
            raise NotImplementedError('OccHead HIP path is eval-only')
        import os
        B, D, H, W, C = x_cl.shape
        is_h2 = isinstance(x_cl, ops.H2)
        if C == 32 and (is_h2 or precision() == 'h2'):
            # split-fp16 kernel (k_occ_head_h2); an fp32 input is converted first (30 us at 16x200x200, still ahead)
            wpk, s0, b0, tailpk, inv2, bounds = self._folded_h2(transposed)
            return ops.occ_head_h2(x_cl if is_h2 else ops.f32_to_h2(x_cl.contiguous()), wpk, s0, b0, tailpk, inv2, bounds,
                                   want_logits=want_logits, want_geo=want_geo, empty_idx=self.empty_idx, occ=occ_out, geo=geo_out)
        if occ_out is not None or geo_out is not None:
            raise NotImplementedError('occ_out / geo_out are built for the split-fp16 OccHead kernel')
        if is_h2:
            x_cl = ops.h2_to_f32(x_cl)
        # fp32: the 32->16 conv runs as Winograd F(2x2x2,3x3x3) (k_occ_head_wino) on grids with enough 4x8x8 tiles to
        # keep the persistent blocks busy, the direct 16x16x4 MFMA kernel on smaller ones
        wino = C == 32 and B * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8) >= 256
        wpk, s0, b0, w1, s1, b1, w2 = self._folded(transposed, wino)
        return ops.occ_head_fused(x_cl, wpk, s0, b0, w1, s1, b1, w2, want_logits=want_logits,
                                  want_geo=want_geo, empty_idx=self.empty_idx)

    def forward(self, voxel_feats, **kwargs):
        assert type(voxel_feats) is list and len(voxel_feats) == self.num_level
        if self.training:                                  # batch-statistics BN, autograd (preworld_amd.train)
            from . import train
            logits = train.occ_head_forward(self, to_channels_last_3d(voxel_feats[0]).float(), transposed=False)
            return {'output_voxels': [from_channels_last_3d(logits)]}
        _, logits = self.decode_cl(to_channels_last_3d(voxel_feats[0]), want_logits=True)
        return {'output_voxels': [from_channels_last_3d(logits)]}


# =====================================================================================
# detector-level composition of the hot path
# =====================================================================================
class DownScaleModule3DCustom(nn.Module):
    """mmdet3d/models/heads/occupancy_head.py:180-200: three Conv3d(k=2, s=2, bias) 32->64->128->128
    and a global average pool.  The reference runs them on (B,C,X,Y,Z); here the feature map
    stays channels-last (B,Z,Y,X,C), so the 2x2x2 taps are permuted (kx,ky,kz)->(kz,ky,kx) when the
    weights are packed."""

    def __init__(self, in_dim):
        super().__init__()
        self.in_dim = in_dim
        self.downscale1 = nn.Conv3d(in_dim, in_dim * 2, 2, stride=2)
        self.downscale2 = nn.Conv3d(in_dim * 2, in_dim * 4, 2, stride=2)
        self.downscale3 = nn.Conv3d(in_dim * 4, in_dim * 4, 2, stride=2)
        self._cache = _PackedCache()

    def _packed(self):
        convs = (self.downscale1, self.downscale2, self.downscale3)
        params = [c.weight for c in convs]
        return self._cache.get(params, lambda: [ops.pack_conv_weight(c.weight.permute(0, 1, 4, 3, 2).contiguous())
                                                for c in convs])

    def forward_cl(self, v_cl, want_levels=False):
        """v_cl (B,Z,Y,X,C) -> (B, 4*in_dim)."""
        x = v_cl
        levels = []
        for conv, wpk in zip((self.downscale1, self.downscale2, self.downscale3), self._packed()):
            x = ops.conv3d_ndhwc(x, wpk, bias=conv.bias, ksize=2, stride=2)
            levels.append(x)
        out = ops.global_avgpool_ndhwc(x)
        return (out, levels) if want_levels else out

    def forward(self, feats):
        """reference layout: feats (B, X, Y, Z, C) -> (B, 1, 1, 1, 4*in_dim)."""
        v_cl = feats.permute(0, 3, 2, 1, 4).contiguous()
        return self.forward_cl(v_cl).view(feats.shape[0], 1, 1, 1, -1)


# PreWorld / PreWorld4DTraj / BEVStereo4DOCC live in detectors.py; `modules.PreWorld4DTraj` keeps working (PEP 562)
def __getattr__(name):
    if name in ('PreWorld', 'PreWorld4DTraj', 'BEVStereo4DOCC'):
        from . import detectors
        return getattr(detectors, name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))


# =====================================================================================
# volume-rendering head
# =====================================================================================
import numpy as _np

nusc_class_frequencies = _np.array([1163161, 2309034, 188743, 2997643, 20317180, 852476, 243808,
                                    2457947, 497017, 2731022, 7224789, 214411435, 5565043, 63191967,
                                    76098082, 128860031, 141625221, 2307405309])


def pack_attribute_grid(density, semantic, color):
    """reference-layout attributes density (X,Y,Z), semantic (X,Y,Z,17), color (X,Y,Z,3) -> the
    packed channels-last grid (Z,Y,X,24) pw_render_rays gathers from (sigma in channel 0)."""
    X, Y, Z = density.shape
    g = density.new_zeros(Z, Y, X, 24)
    g[..., 0] = density.permute(2, 1, 0)
    g[..., 2:19] = semantic.permute(2, 1, 0, 3)
    g[..., 19:22] = color.permute(2, 1, 0, 3)
    return g


class NerfHead(nn.Module):
    """Drop-in for mmdet3d/models/nerf/nerf_head.py:103-420 (same kwargs, same registered
    buffers scene_center / scene_radius / xyz_min / xyz_max / act_shift, same loss keys).

    The forward pass of render_one_scene + render_depth/semantic/color is one HIP kernel
    (pw_render_rays, one wavefront per ray); the scalar losses on the (n_rays,) outputs are a
    handful of torch reductions.  With gradients enabled the render runs through ops.RenderRays, whose backward is
    ONE kernel as well (pw_render_rays_backward: reverse transmittance scan + trilinear corner scatter-adds into the packed
    grid); ops.Raw2Alpha / ops.Alphas2Weights keep the reference's op-level autograd interface."""

    def __init__(self, point_cloud_range, voxel_size, scene_center=None, radius=39, step_size=0.5,
                 use_depth_sup=True, balance_cls_weight=True, weight_depth=1.0, weight_semantic=1.0,
                 weight_color=1.0, weight_entropy_last=0.01, weight_distortion=0.01, alpha_init=1e-6,
                 fast_color_thres=1e-7):
        super().__init__()
        self.weight_entropy_last, self.weight_distortion = weight_entropy_last, weight_distortion
        xyz_min = torch.Tensor(point_cloud_range[:3])
        xyz_max = torch.Tensor(point_cloud_range[3:])
        xyz_range = (xyz_max - xyz_min).float()
        self.bg_len = (xyz_range[0] // 2 - radius) / radius          # 0-dim fp32 tensor (:131)
        self.radius = radius
        self.register_buffer('scene_center', (xyz_min + xyz_max) * 0.5)
        self.register_buffer('scene_radius', torch.Tensor([radius, radius, radius]))
        self.step_size, self.use_depth_sup = step_size, use_depth_sup
        z_ = xyz_range[2] / xyz_range[0]
        self.register_buffer('xyz_min', torch.Tensor([-1 - self.bg_len, -1 - self.bg_len, -z_]))
        self.register_buffer('xyz_max', torch.Tensor([1 + self.bg_len, 1 + self.bg_len, z_]))
        self.alpha_init = alpha_init
        self.register_buffer('act_shift', torch.FloatTensor([_np.log(1 / (1 - alpha_init) - 1)]))
        self.voxel_size = voxel_size / radius
        self.world_size = torch.Tensor([200, 200, 16]).long()
        self.world_len = self.world_size[0].item()
        self.fast_color_thres = fast_color_thres
        self.weight_depth, self.weight_semantic, self.weight_color = weight_depth, weight_semantic, weight_color
        if balance_cls_weight:
            self.class_weights = torch.from_numpy(1 / _np.log(nusc_class_frequencies[:17] + 0.001))
        else:
            self.class_weights = torch.ones(17) / 17
        self._t = None

    def t_table(self, device):
        """nerf_head.py:35-43, with the same torch calls (N_inner=391, N_outer=26 -> 417)."""
        if self._t is None or self._t.device != torch.device(device):
            N_inner = int(2 / (2 + 2 * self.bg_len) * self.world_len / self.step_size) + 1
            N_outer = N_inner // 15
            b_inner = torch.linspace(0, 2, N_inner + 1)
            b_outer = 2 / torch.linspace(1, 1 / 64, N_outer + 1)
            t = torch.cat([(b_inner[1:] + b_inner[:-1]) * 0.5, (b_outer[1:] + b_outer[:-1]) * 0.5])
            self._t = t.float().contiguous().to(device)
        return self._t

    def consts(self, bda, interval=0.5):
        """the kernel's scalar arguments as host floats.  The registered buffers are read back ONCE per (device, in-place version): as
        `float(v) for v in self.<buffer>` they were 24 one-element D2H syncs per rendered batch element (round 6)."""
        bufs = (self.scene_center, self.scene_radius, self.xyz_min, self.xyz_max, self.act_shift)
        key = tuple((b.device, b._version, b.data_ptr()) for b in bufs)
        host = self.__dict__.get('_consts_host')
        if host is None or host[0] != key:
            host = self.__dict__['_consts_host'] = (key, [[float(v) for v in b.detach().cpu().reshape(-1)] for b in bufs])
        sc, sr, xmin, xmax, shift = host[1]
        dist_thres = (2 + 2 * self.bg_len) / self.world_len * self.step_size * 0.95       # :197
        return sc + sr + [float(v) for v in bda.detach().cpu().reshape(-1)] + xmin + xmax + \
            [float(self.bg_len), shift[0], float(interval), float(dist_thres), float(self.fast_color_thres), float(self.radius)]

    @torch.no_grad()
    def render(self, grid, rays_o, rays_d, bda, want_debug=False):
        """grid: packed (Z,Y,X,24) attribute grid; rays (R,3); bda (3,3)."""
        return ops.render_rays(rays_o.float().contiguous(), rays_d.float().contiguous(),
                               self.t_table(grid.device), grid, self.consts(bda.cpu()),
                               want_debug=want_debug)

    def compute_loss(self, out, target_depth, target_semantic, target_color, suffix=''):
        """nerf_head.py:271-329 on the fused kernel's outputs."""
        losses = {}
        if self.use_depth_sup:
            d = torch.log(out['depth'] + 1e-7) - torch.log(target_depth)
            losses['loss_render_depth' + suffix] = \
                torch.sqrt((d ** 2).mean() - 0.85 * (d.mean() ** 2)) * self.weight_depth
        crit = nn.CrossEntropyLoss(weight=self.class_weights.type_as(out['semantic']), reduction='mean')
        losses['loss_render_semantic' + suffix] = crit(out['semantic'], target_semantic.long()) * self.weight_semantic
        losses['loss_render_color' + suffix] = \
            torch.sum(torch.mean(torch.abs(out['color'] - target_color), dim=0)) * self.weight_color
        if self.weight_entropy_last > 0:
            p = out['alphainv_last'].clamp(1e-6, 1 - 1e-6)
            losses['loss_sdf_entropy' + suffix] = self.weight_entropy_last * \
                -(p * torch.log(p) + (1 - p) * torch.log(1 - p)).mean()
        if self.weight_distortion > 0 and 'weights' in out:
            w = out['weights']                                   # (R,S) dense, 0 where culled
            t = self.t_table(w.device)
            losses['loss_sdf_distortion' + suffix] = self.weight_distortion * ops.distortion_loss(w, 1 - 1 / (1 + t))
        return losses

    def forward(self, density, semantic, color, if_pretrain=False, if_temporal=False,
                dataset_type='Nuscenes', rays=None, bda=None, interval=0, **kwargs):
        """Same signature as nerf_head.py:361-420.  density (B,X,Y,Z), semantic (B,X,Y,Z,17),
        color (B,X,Y,Z,3), rays (B,R,16), bda (B,3,3) -> dict of scalar losses."""
        assert dataset_type == 'Nuscenes'
        losses = {}
        suffix = '_%ds' % int(interval) if if_temporal else ''
        need_grad = torch.is_grad_enabled() and (density.requires_grad or semantic.requires_grad or color.requires_grad)
        bda_host = bda.detach().cpu()                               # one D2H copy for the batch (was one per batch element)
        for b in range(rays.shape[0]):
            gt_depth = rays[b, :, 2]
            gt_depth[gt_depth > 52] = 0                          # in-place, like :379
            mask = gt_depth > 0
            grid = pack_attribute_grid(density[b].float(), semantic[b].float(), color[b].float())
            ro, rd = rays[b, :, 4:7][mask].float().contiguous(), rays[b, :, 7:10][mask].float().contiguous()
            if need_grad:
                # training: the fused forward + ONE backward kernel (reverse transmittance scan + corner scatter-adds)
                d_, s_, c_, l_, w_ = ops.RenderRays.apply(grid, ro, rd, self.t_table(grid.device), self.consts(bda_host[b]))
                out = dict(depth=d_, semantic=s_, color=c_, alphainv_last=l_)
                if self.weight_distortion > 0:
                    out['weights'] = w_
            else:
                out = ops.render_rays(ro, rd, self.t_table(grid.device), grid, self.consts(bda_host[b]),
                                      want_debug=self.weight_distortion > 0)
            single = self.compute_loss(out, gt_depth[mask], rays[b, :, 3][mask], rays[b, :, 13:16][mask],
                                       suffix)
            for k, v in single.items():
                losses[k] = losses[k] + v if k in losses else v
        for k in losses:
            losses[k] = losses[k] / semantic.shape[0]
        return losses
