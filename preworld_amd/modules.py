"""Drop-in nn.Modules mirroring the reference's registry classes for the hot path.

Same class names, constructor kwargs, forward signatures and state-dict keys as the
reference, so `configs/preworld/*.py` model dicts build them unchanged via
preworld_amd.registry.  All hot arithmetic goes through libpreworld_hip.so (preworld_amd.ops);
torch only provides parameters, device memory and the stream.  Modules raise if the HIP
library is missing or if asked to run on CPU tensors -- there is no fallback path.
"""
import torch
import torch.nn as nn

from . import ops


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
        return ([float(v) for v in self.grid_lower_bound], [float(v) for v in self.grid_interval],
                [int(v) for v in self.grid_size])

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
        B, N = sensor2ego.shape[:2]
        lower, interval, size = self._grid()
        ipr, comb, tr = ops.lss_camera_matrices(sensor2ego, cam2imgs, post_rots)
        vox = ops.lss_voxel_index(self._frustum_on(sensor2ego), ipr, post_trans.contiguous().float(),
                                  comb, tr, bda.contiguous().float(), lower, interval, size, B, N)
        n_vox = B * size[0] * size[1] * size[2]
        seg_start, order = ops.segment_sort(vox, n_vox)
        return seg_start, order, n_vox

    # ---- view_transformer.py:203-261 (takes the camera tensors instead of coor: the 17.8 MB
    # coordinate tensor is never materialised; use get_lidar_coor() if you need it)
    def voxel_pooling_prepare_v2(self, sensor2ego, cam2imgs, post_rots, post_trans, bda):
        seg_start, order, n_vox = self._sort(sensor2ego, cam2imgs, post_rots, post_trans, bda)
        H, W = self.frustum.shape[1:3]
        return ops.lss_ranks(seg_start, order, n_vox, self.D, H * W)

    # ---- view_transformer.py:155-174,263-267
    def init_acceleration_v2(self, input):
        self._cache = self._sort(input[1], input[3], input[4], input[5], input[6])

    def pre_compute(self, input):
        if self.initial_flag:
            self.init_acceleration_v2(input)
            self.initial_flag = False

    # ---- view_transformer.py:176-201,269-291
    def view_transform_core(self, input, depth, tran_feat):
        B, N, C, H, W = input[0].shape
        _, _, size = self._grid()
        if self.accelerate and self._cache is not None:
            seg_start, order, n_vox = self._cache
        else:
            seg_start, order, n_vox = self._sort(input[1], input[3], input[4], input[5], input[6])
        feat = tran_feat.view(B, N, self.out_channels, H, W).permute(0, 1, 3, 4, 2).contiguous().float()
        dep = depth.view(B, N, self.D, H, W).contiguous().float()
        if (dep.requires_grad or feat.requires_grad) and torch.is_grad_enabled():
            rb, rd, rf, st, ln = ops.lss_ranks(seg_start, order, n_vox, self.D, H * W)
            if rb is None:
                out = feat.new_zeros(B, size[2], size[1], size[0], self.out_channels)
                bev = out.permute(0, 4, 1, 2, 3)
            else:
                bev = ops.bev_pool_v2(dep, feat, rd, rf, rb,
                                      (B, size[2], size[1], size[0], self.out_channels), st, ln)
        else:
            out = ops.bev_pool_dense(dep, feat, seg_start, order, n_vox, self.D, H * W)
            bev = out.view(B, size[2], size[1], size[0], self.out_channels).permute(0, 4, 1, 2, 3)
        if self.collapse_z:
            bev = torch.cat(bev.unbind(dim=2), 1)
        return bev, depth

    def view_transform(self, input, depth, tran_feat):
        if self.accelerate:
            self.pre_compute(input)
        return self.view_transform_core(input, depth, tran_feat)

    def forward(self, input):
        x = input[0]
        B, N, C, H, W = x.shape
        x = x.view(B * N, C, H, W)
        x = self.depth_net(x)
        depth_digit = x[:, :self.D, ...]
        tran_feat = x[:, self.D:self.D + self.out_channels, ...]
        depth = depth_digit.softmax(dim=1)
        return self.view_transform(input, depth, tran_feat)

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda):
        return None
