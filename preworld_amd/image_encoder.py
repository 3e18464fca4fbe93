"""the image side of the detector in plain torch.nn (no mmcv / mmdet / mmseg), so that a true
image -> occupancy samples/s can be reported.  Nothing here is a custom kernel: the reference keeps these modules on
PyTorch and so does this package (north_star: "the multi-camera backbone stays on PyTorch-ROCm"); only the DepthNet's
stereo cost volume and its softmax tail call the HIP library (ops.stereo_cost_volume, ops.depthnet_tail -- rows 8f.1).

State-dict keys, constructor arguments and forward contracts are the reference's, so released checkpoints load:

  SwinTransformer  mmdet3d/models/backbones/swin.py:680-976 (PatchEmbed :79-171, PatchMerging :174-241, WindowMSA :244-350,
                   ShiftWindowMSA :353-513, SwinBlock :516-592, SwinBlockSequence :595-677); the FFN is mmcv-full 1.6.0's
                   (third-party, not in the tree): layers = Sequential(Sequential(Linear, act, Dropout), Linear, Dropout)
  FPN_LSS          mmdet3d/models/necks/lss_fpn.py:12-101
  DepthNet         mmdet3d/models/necks/view_transformer.py:471-638 (ASPP :323-420, Mlp :423-445, SELayer :448-468);
                   BasicBlock is mmdet 2.24's ResNet block (third-party): conv1, bn1, conv2, bn2, downsample
  get_mlp_input    mmdet3d/models/necks/view_transformer.py:713-734
  ImageBranch      BEVDet.image_encoder (detectors/bevdet.py:34-50), extract_stereo_ref_feat (:573-603) and the per-frame
                   loop of BEVStereo4DOCC.extract_img_feat / prepare_bev_feat (detectors/bevdet_occ.py:142-241) up to the
                   (depth, context) pair the voxel-pooling path starts from.

Inference only (DropPath / Dropout are identities in eval mode; torch.utils.checkpoint is not used)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- Swin
class PatchEmbed(nn.Module):
    def __init__(self, in_channels=3, embed_dims=128, kernel_size=4, stride=4, norm=True):
        super().__init__()
        self.patch_size = (kernel_size, kernel_size)
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size=kernel_size, stride=stride)
        self.norm = nn.LayerNorm(embed_dims) if norm else None

    def forward(self, x):
        H, W = x.shape[2], x.shape[3]
        if H % self.patch_size[0] != 0:
            x = F.pad(x, (0, 0, 0, self.patch_size[0] - H % self.patch_size[0]))
        if W % self.patch_size[1] != 0:
            x = F.pad(x, (0, self.patch_size[1] - W % self.patch_size[1], 0, 0))
        x = self.projection(x)
        self.DH, self.DW = x.shape[2], x.shape[3]
        x = x.flatten(2).transpose(1, 2)
        return self.norm(x) if self.norm is not None else x


class PatchMerging(nn.Module):
    """nn.Unfold grouping (channel-major 2x2 samples), LayerNorm, Linear -- swin.py:174-241."""

    def __init__(self, in_channels, out_channels, stride=2, norm=True):
        super().__init__()
        self.stride, self.out_channels = stride, out_channels
        self.sampler = nn.Unfold(kernel_size=stride, dilation=1, padding=0, stride=stride)
        self.norm = nn.LayerNorm(stride ** 2 * in_channels) if norm else None
        self.reduction = nn.Linear(stride ** 2 * in_channels, out_channels, bias=False)

    def forward(self, x, hw_shape):
        B, L, C = x.shape
        H, W = hw_shape
        x = x.view(B, H, W, C).permute(0, 3, 1, 2)
        if (H % self.stride != 0) or (W % self.stride != 0):
            x = F.pad(x, (0, W % self.stride, 0, H % self.stride))
        x = self.sampler(x).transpose(1, 2)
        x = self.norm(x) if self.norm is not None else x
        return self.reduction(x), ((H + 1) // 2, (W + 1) // 2)


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.embed_dims, self.window_size, self.num_heads = embed_dims, window_size, num_heads
        self.scale = qk_scale or (embed_dims // num_heads) ** -0.5
        Wh, Ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        seq1 = torch.arange(0, (2 * Ww - 1) * Wh, 2 * Ww - 1)
        seq2 = torch.arange(0, Ww, 1)
        coords = (seq1[:, None] + seq2[None, :]).reshape(1, -1)
        self.register_buffer('relative_position_index', (coords + coords.T).flip(1).contiguous())
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)

    def forward(self, x, mask=None):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        bias = bias.permute(2, 0, 1).contiguous()                                 # nH, N, N
        # softmax(q k^T * scale + bias [+ mask]) v  ==  the reference's explicit attention (swin.py:322-344).  Measured
        # on MI355X (Swin-B, 512x1408, fp32): SDPA with the additive term expanded per window 137 ms per sample,
        # SDPA over (B/nW, nW, ...) with a broadcast mask 150 ms, explicit matmul + softmax 154 ms.
        if mask is not None:                                                      # (nW, N, N), 0 / -100
            nW = mask.shape[0]
            add = (bias.view(1, 1, self.num_heads, N, N) + mask.view(1, nW, 1, N, N)).expand(B // nW, nW, self.num_heads, N, N)
            x = F.scaled_dot_product_attention(q, k, v, attn_mask=add.reshape(B, self.num_heads, N, N), scale=self.scale)
        else:
            x = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.unsqueeze(0), scale=self.scale)
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, shift_size=0, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        self.w_msa = WindowMSA(embed_dims, num_heads, (window_size, window_size), qkv_bias, qk_scale)
        self._mask_cache = {}

    def _partition(self, x):
        B, H, W, C = x.shape
        ws = self.window_size
        x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous()
        return x.view(-1, ws, ws, C)

    def _reverse(self, windows, H, W):
        ws = self.window_size
        B = int(windows.shape[0] / (H * W / ws / ws))
        x = windows.view(B, H // ws, W // ws, ws, ws, -1)
        return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)

    def _shift_mask(self, Hp, Wp, device):
        """Additive attention mask of the shifted configuration (swin.py:422-444): after the cyclic shift the last
        window row / column mixes three source regions per axis (|..unshifted..|ws - ss|ss|); tokens of different
        regions must not attend to each other (-100).  Region ids come from index arithmetic and the (nW, N, N) mask
        is cached per padded size."""
        key = (Hp, Wp, str(device))
        m = self._mask_cache.get(key)
        if m is None:
            ws, ss = self.window_size, self.shift_size
            ih, iw = torch.arange(Hp, device=device), torch.arange(Wp, device=device)
            rh = (ih >= Hp - ws).long() + (ih >= Hp - ss).long()
            rw = (iw >= Wp - ws).long() + (iw >= Wp - ss).long()
            region = (3 * rh[:, None] + rw[None, :]).float().view(1, Hp, Wp, 1)
            mw = self._partition(region).view(-1, ws * ws)
            diff = mw.unsqueeze(1) - mw.unsqueeze(2)
            m = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
            self._mask_cache[key] = m
        return m

    def forward(self, query, hw_shape):
        B, L, C = query.shape
        H, W = hw_shape
        ws, ss = self.window_size, self.shift_size
        x = query.view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        mask = None
        if ss > 0:
            x = torch.roll(x, shifts=(-ss, -ss), dims=(1, 2))
            mask = self._shift_mask(Hp, Wp, x.device)
        win = self.w_msa(self._partition(x).view(-1, ws * ws, C), mask=mask)
        x = self._reverse(win.view(-1, ws, ws, C), Hp, Wp)
        if ss > 0:
            x = torch.roll(x, shifts=(ss, ss), dims=(1, 2))
        if pad_r or pad_b:
            x = x[:, :H, :W, :].contiguous()
        return x.view(B, H * W, C)


class FFN(nn.Module):
    """mmcv.cnn.bricks.transformer.FFN (num_fcs=2, add_identity=True): keys layers.0.0.*, layers.1.*"""

    def __init__(self, embed_dims, feedforward_channels):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Dropout(0.)),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.))

    def forward(self, x, identity=None):
        return (x if identity is None else identity) + self.layers(x)


class SwinBlock(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size, shift, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = FFN(embed_dims, feedforward_channels)

    def forward(self, x, hw_shape):
        x = self.attn(self.norm1(x), hw_shape) + x
        return self.ffn(self.norm2(x), identity=x)


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, depth, window_size, downsample, qkv_bias=True,
                 qk_scale=None):
        super().__init__()
        self.blocks = nn.ModuleList([SwinBlock(embed_dims, num_heads, feedforward_channels, window_size, i % 2 == 1,
                                               qkv_bias, qk_scale) for i in range(depth)])
        self.downsample = downsample

    def forward(self, x, hw_shape):
        for block in self.blocks:
            x = block(x, hw_shape)
        if self.downsample is not None:
            x_down, down_hw = self.downsample(x, hw_shape)
            return x_down, down_hw, x, hw_shape
        return x, hw_shape, x, hw_shape


class SwinTransformer(nn.Module):
    """Same constructor arguments as the reference (those that only matter for training / checkpoint loading are
    accepted and ignored).  forward(x (BN,3,H,W)) -> list: [stage-0 feature un-normed if return_stereo_feat] +
    [norm_i(stage i) for i in out_indices], each (BN, C_i, H_i, W_i) -- swin.py:946-971."""

    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7, mlp_ratio=4,
                 depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 use_abs_pos_embed=False, act_cfg=None, norm_cfg=None, pretrain_style='official', pretrained=None,
                 init_cfg=None, with_cp=True, return_stereo_feat=False, output_missing_index_as_none=False,
                 frozen_stages=-1):
        super().__init__()
        assert strides[0] == patch_size, 'Use non-overlapping patch embed.'
        if use_abs_pos_embed:
            raise NotImplementedError('use_abs_pos_embed=True is not used by the PreWorld configs')
        self.out_indices, self.return_stereo_feat = tuple(out_indices), return_stereo_feat
        self.output_missing_index_as_none = output_missing_index_as_none
        self.use_abs_pos_embed = False
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size, strides[0], norm=patch_norm)
        self.drop_after_pos = nn.Dropout(p=drop_rate)
        self.stages = nn.ModuleList()
        c = embed_dims
        for i in range(len(depths)):
            down = PatchMerging(c, 2 * c, strides[i + 1], norm=patch_norm) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(c, num_heads[i], mlp_ratio * c, depths[i], window_size, down, qkv_bias,
                                                 qk_scale))
            if down is not None:
                c = down.out_channels
        self.num_features = [int(embed_dims * 2 ** i) for i in range(len(depths))]
        for i in self.out_indices:
            self.add_module('norm%d' % i, nn.LayerNorm(self.num_features[i]))

    def _to_map(self, out, hw, i, channels_last=False):
        m = out.view(-1, *hw, self.num_features[i]).permute(0, 3, 1, 2)
        # the stereo feature only feeds the cost volume kernel, whose fast path reads channels-last storage: the
        # token layout (B, H*W, C) already is that, so the NCHW-shaped view is handed out without the 138 MB copy
        return m if channels_last else m.contiguous()

    def forward(self, x):
        x = self.drop_after_pos(self.patch_embed(x))
        hw_shape = (self.patch_embed.DH, self.patch_embed.DW)
        outs = []
        for i, stage in enumerate(self.stages):
            x, hw_shape, out, out_hw = stage(x, hw_shape)
            if i == 0 and self.return_stereo_feat:
                outs.append(self._to_map(out, out_hw, i, channels_last=True))
            if i in self.out_indices:
                outs.append(self._to_map(getattr(self, 'norm%d' % i)(out), out_hw, i))
            elif self.output_missing_index_as_none:
                outs.append(None)
        return outs

    def stereo_ref_feat(self, x):
        """extract_stereo_ref_feat's Swin branch (bevdet.py:589-603): patch embedding + stage 0 only."""
        x = self.drop_after_pos(self.patch_embed(x))
        hw_shape = (self.patch_embed.DH, self.patch_embed.DW)
        _, _, out, out_hw = self.stages[0](x, hw_shape)
        return self._to_map(out, out_hw, 0, channels_last=True)


# ----------------------------------------------------------------------------------------------- FPN_LSS
class FPN_LSS(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4, input_feature_index=(0, 2), norm_cfg=None,
                 extra_upsample=2, lateral=None, use_input_conv=False):
        super().__init__()
        if lateral is not None or use_input_conv:
            raise NotImplementedError('lateral / use_input_conv are not used by the PreWorld configs')
        self.input_feature_index = input_feature_index
        self.extra_upsample = extra_upsample is not None
        self.up = nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=True)
        cf = 2 if self.extra_upsample else 1
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels * cf, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels * cf),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels * cf, out_channels * cf, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels * cf),
            nn.ReLU(inplace=True))
        if self.extra_upsample:
            self.up2 = nn.Sequential(
                nn.Upsample(scale_factor=extra_upsample, mode='bilinear', align_corners=True),
                nn.Conv2d(out_channels * cf, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
                nn.ReLU(inplace=True), nn.Conv2d(out_channels, out_channels, 1, padding=0))

    def forward(self, feats):
        x2, x1 = feats[self.input_feature_index[0]], feats[self.input_feature_index[1]]
        x = self.conv(torch.cat([x2, self.up(x1)], dim=1))
        return self.up2(x) if self.extra_upsample else x


# ----------------------------------------------------------------------------------------------- DepthNet
class BasicBlock2d(nn.Module):
    """mmdet.models.backbones.resnet.BasicBlock (2.24, third-party): 3x3 conv-BN-ReLU, 3x3 conv-BN, + downsample(x), ReLU."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size, stride=1, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.relu(self.bn(self.atrous_conv(x)))


class ASPP(nn.Module):
    def __init__(self, inplanes, mid_channels=256):
        super().__init__()
        self.aspp1 = _ASPPModule(inplanes, mid_channels, 1, 0, 1)
        self.aspp2 = _ASPPModule(inplanes, mid_channels, 3, 6, 6)
        self.aspp3 = _ASPPModule(inplanes, mid_channels, 3, 12, 12)
        self.aspp4 = _ASPPModule(inplanes, mid_channels, 3, 18, 18)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(inplanes, mid_channels, 1, bias=False),
                                             nn.BatchNorm2d(mid_channels), nn.ReLU())
        self.conv1 = nn.Conv2d(int(mid_channels * 5), inplanes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        x5 = F.interpolate(self.global_avg_pool(x), size=x.shape[2:], mode='bilinear', align_corners=True)
        x = torch.cat((self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x), x5), dim=1)
        return self.dropout(self.relu(self.bn1(self.conv1(x))))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.ReLU()
        self.drop1 = nn.Dropout(0.0)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class SELayer(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = nn.ReLU()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = nn.Sigmoid()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class DepthNet(nn.Module):
    """view_transformer.py:471-638 with use_dcn=False (the PreWorld configs).  With `stereo_metas` holding a previous
    stereo feature the cost volume comes from ops.stereo_cost_volume (HIP, one kernel instead of the reference's 32
    grid_sample rounds); without one it is zeros, as in the reference (:621-628)."""

    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, use_dcn=True, use_aspp=True,
                 with_cp=False, stereo=False, bias=0.0, aspp_mid_channels=-1, D=100):
        super().__init__()
        if use_dcn:
            raise NotImplementedError('use_dcn=True (mmcv deformable conv) is not used by the PreWorld configs')
        self.reduce_conv = nn.Sequential(nn.Conv2d(in_channels, mid_channels, 3, stride=1, padding=1),
                                         nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.context_conv = nn.Conv2d(mid_channels, context_channels, 1)
        self.bn = nn.BatchNorm1d(27)
        self.depth_mlp = Mlp(27, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        self.context_mlp = Mlp(27, mid_channels, mid_channels)
        self.context_se = SELayer(mid_channels)
        cin, downsample = mid_channels, None
        self.stereo = stereo
        if stereo:
            cin += depth_channels
            downsample = nn.Conv2d(cin, mid_channels, 1, 1, 0)
            net = []
            for _ in range(2):
                net += [nn.Conv2d(depth_channels, depth_channels, 3, stride=2, padding=1), nn.BatchNorm2d(depth_channels)]
            self.cost_volumn_net = nn.Sequential(*net)
            self.bias = bias
        layers = [BasicBlock2d(cin, mid_channels, downsample=downsample), BasicBlock2d(mid_channels, mid_channels),
                  BasicBlock2d(mid_channels, mid_channels)]
        if use_aspp:
            layers.append(ASPP(mid_channels, mid_channels if aspp_mid_channels < 0 else aspp_mid_channels))
        layers.append(nn.Conv2d(mid_channels, depth_channels, 1))
        self.depth_conv = nn.Sequential(*layers)
        self.depth_channels = depth_channels

    def calculate_cost_volumn(self, metas):
        from . import ops
        prev, curr = metas['cv_feat_list']
        return ops.stereo_cost_volume(prev, curr, metas['frustum'], metas['k2s_sensor'], metas['intrins'],
                                      metas['post_rots'], metas['post_trans'], bias=self.bias)

    def forward(self, x, mlp_input, stereo_metas=None):
        mlp_input = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        x = self.reduce_conv(x)
        context = self.context_conv(self.context_se(x, self.context_mlp(mlp_input)[..., None, None]))
        depth = self.depth_se(x, self.depth_mlp(mlp_input)[..., None, None])
        if stereo_metas is not None:
            if stereo_metas['cv_feat_list'][0] is None:
                BN, _, H, W = x.shape
                s = float(stereo_metas['downsample']) / stereo_metas['cv_downsample']
                cv = torch.zeros((BN, self.depth_channels, int(H * s), int(W * s))).to(x)
            else:
                with torch.no_grad():
                    cv = self.calculate_cost_volumn(stereo_metas)
            depth = torch.cat([depth, self.cost_volumn_net(cv)], dim=1)
        return torch.cat([self.depth_conv(depth), context], dim=1)


def get_mlp_input(sensor2ego, ego2global, intrin, post_rot, post_tran, bda):
    """LSSViewTransformerBEVDepth.get_mlp_input (view_transformer.py:713-734): 15 scalars + sensor2ego[:3,:] = 27."""
    B, N, _, _ = sensor2ego.shape
    bda = bda.view(B, 1, 3, 3).repeat(1, N, 1, 1)
    v = torch.stack([intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2], intrin[:, :, 1, 2],
                     post_rot[:, :, 0, 0], post_rot[:, :, 0, 1], post_tran[:, :, 0],
                     post_rot[:, :, 1, 0], post_rot[:, :, 1, 1], post_tran[:, :, 1],
                     bda[:, :, 0, 0], bda[:, :, 0, 1], bda[:, :, 1, 0], bda[:, :, 1, 1], bda[:, :, 2, 2]], dim=-1)
    return torch.cat([v, sensor2ego[:, :, :3, :].reshape(B, N, -1)], dim=-1)


def create_frustum(depth_cfg, input_size, downsample):
    """LSSViewTransformer.create_frustum (view_transformer.py:84-112): (D, H/ds, W/ds, 3) of (u, v, d)."""
    H_in, W_in = input_size
    Hf, Wf = H_in // downsample, W_in // downsample
    d = torch.arange(*depth_cfg, dtype=torch.float).view(-1, 1, 1).expand(-1, Hf, Wf)
    D = d.shape[0]
    x = torch.linspace(0, W_in - 1, Wf, dtype=torch.float).view(1, 1, Wf).expand(D, Hf, Wf)
    y = torch.linspace(0, H_in - 1, Hf, dtype=torch.float).view(1, Hf, 1).expand(D, Hf, Wf)
    return torch.stack((x, y, d), -1)


# ----------------------------------------------------------------------------------------------- image -> (depth, context)
class ImageBranch(nn.Module):
    """img_backbone + img_neck + DepthNet with the reference's attribute names (`img_backbone`, `img_neck`,
    `img_view_transformer.depth_net`) so that a detector checkpoint's keys map one to one.

    frames_from_images(...) runs BEVStereo4DOCC.extract_img_feat's frame loop (bevdet_occ.py:188-241) up to the lifting
    inputs: for fid = num_frame-1 .. 0: the extra reference frame gives only its stage-0 stereo feature; every other
    frame goes through backbone, neck and DepthNet (cost volume against the previously processed, i.e. older, frame's
    stereo feature) and yields dict(depth softmax (B*N, D, H, W), tran_feat channels-last (B*N, H, W, C), sensor2keyego,
    intrin, post_rot, post_tran, bda) -- exactly what modules.PreWorld4DTraj.simple_test_from_lift consumes (key first)."""

    def __init__(self, backbone_cfg, neck_cfg, depthnet_cfg, grid_depth, input_size, downsample=16, out_channels=32,
                 in_channels=512):
        super().__init__()
        self.img_backbone = SwinTransformer(**backbone_cfg)
        self.img_neck = FPN_LSS(**neck_cfg)
        self.img_view_transformer = nn.Module()
        self.D = len(torch.arange(*grid_depth))
        self.img_view_transformer.depth_net = DepthNet(in_channels, in_channels, out_channels, self.D, **depthnet_cfg)
        self.out_channels, self.downsample = out_channels, downsample
        self.register_buffer('cv_frustum', create_frustum(grid_depth, input_size, 4), persistent=False)

    def image_encoder(self, img):
        B, N, C, H, W = img.shape
        x = self.img_backbone(img.view(B * N, C, H, W))
        stereo_feat, x = x[0], self.img_neck(x[1:])
        return x.view(B, N, *x.shape[1:]), stereo_feat

    @torch.no_grad()
    def frames_from_images(self, imgs, sensor2keyegos, ego2globals, intrins, post_rots, post_trans, bda, curr2adjsensor,
                           extra_ref_frames=1, with_prev=True):
        from . import ops
        num_frame = len(imgs)
        frames, feat_prev = [], None
        for fid in range(num_frame - 1, -1, -1):
            img = imgs[fid]
            if fid == num_frame - extra_ref_frames:
                B, N, C, H, W = img.shape
                feat_prev = self.img_backbone.stereo_ref_feat(img.view(B * N, C, H, W))
                continue
            if not (fid == 0 or with_prev):
                continue
            mlp_input = get_mlp_input(sensor2keyegos[0], ego2globals[0], intrins[fid], post_rots[fid], post_trans[fid], bda)
            x, stereo_feat = self.image_encoder(img)
            B, N, C, H, W = x.shape
            metas = dict(k2s_sensor=curr2adjsensor[fid], intrins=intrins[fid], post_rots=post_rots[fid],
                         post_trans=post_trans[fid], frustum=self.cv_frustum.to(x), cv_downsample=4,
                         downsample=self.downsample, cv_feat_list=[feat_prev, stereo_feat])
            out = self.img_view_transformer.depth_net(x.view(B * N, C, H, W), mlp_input, metas)
            depth, feat = ops.depthnet_tail(out.float().contiguous(), self.D, self.out_channels)
            feat._pw_channels_last = True       # (B*N, H, W, C): view_transform_core skips its permute copy
            frames.append(dict(depth=depth, tran_feat=feat, sensor2keyego=sensor2keyegos[fid], intrin=intrins[fid],
                               post_rot=post_rots[fid], post_tran=post_trans[fid], bda=bda))
            feat_prev = stereo_feat
        return frames[::-1]                     # key frame first


def preworld_image_cfg():
    """configs/preworld/nuscenes/bevstereo-occ.py:45-88: Swin-B, FPN_LSS, DepthNet of the PreWorld detectors."""
    return dict(
        backbone_cfg=dict(pretrain_img_size=224, patch_size=4, window_size=12, mlp_ratio=4, embed_dims=128,
                          depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], strides=(4, 2, 2, 2), out_indices=(2, 3),
                          qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                          use_abs_pos_embed=False, return_stereo_feat=True, pretrain_style='official',
                          output_missing_index_as_none=False),
        neck_cfg=dict(in_channels=512 + 1024, out_channels=512, extra_upsample=None, input_feature_index=(0, 1),
                      scale_factor=2),
        depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True, bias=5.),
        grid_depth=[1.0, 45.0, 0.5], input_size=(512, 1408), downsample=16, out_channels=32, in_channels=512)
