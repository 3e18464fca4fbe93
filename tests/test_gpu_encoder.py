"""GPU parity: MFMA conv3d, fused neck, fused OccHead, forecast recursion and their
composition (HIP through the C ABI) against the CPU oracle and the golden vectors generated
from the imported reference.  fp32 everywhere; the MFMA accumulates K in a different order
than the oracle's scalar loops, so tolerances are rtol=2e-4 / atol=2e-4 on O(1) activations
(same bound the oracle itself meets against the reference's torch-CPU output)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from preworld_amd import modules as M
from preworld_amd import ops
from preworld_amd import synth as S

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = dict(rtol=2e-4, atol=2e-4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cl(x):      # numpy (B,C,D,H,W) -> torch channels-last (B,D,H,W,C) on device
    return T(np.ascontiguousarray(x.transpose(0, 2, 3, 4, 1)))


def ncdhw(y):   # torch (B,D,H,W,C) -> numpy (B,C,D,H,W)
    return y.permute(0, 4, 1, 2, 3).contiguous().cpu().numpy()


def _rand_conv(rs, cout, cin, k):
    return (rs.standard_normal((cout, cin, k, k, k)) * np.sqrt(2.0 / (cin * k ** 3))).astype(np.float32)


@pytest.mark.parametrize('shape', [(1, 32, 4, 8, 8), (2, 32, 5, 11, 13), (1, 64, 3, 9, 17),
                                   (1, 128, 4, 6, 10)])
@pytest.mark.parametrize('cout', [32, 64, 16])
@pytest.mark.parametrize('algo', [1, 2])
def test_conv3d_k3s1_vs_oracle(shape, cout, algo):
    rs = np.random.RandomState(hash((shape, cout)) % 2 ** 31)
    x = rs.standard_normal(shape).astype(np.float32)
    w = _rand_conv(rs, cout, shape[1], 3)
    scale = (rs.rand(cout) + 0.5).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32)
    res = rs.standard_normal((shape[0], cout) + shape[2:]).astype(np.float32)
    want = O.conv3d(x, w, None, 1, 1) * scale[None, :, None, None, None] + bias[None, :, None, None, None]
    want = np.maximum(want + res, 0)
    wpk = ops.pack_conv_weight(T(w))
    got = ops.conv3d_ndhwc(cl(x), wpk, ops._pad32(T(scale), 1.0), ops._pad32(T(bias), 0.0),
                           residual=cl(res), cout0=cout, ksize=3, stride=1, relu0=True, algo=algo)
    np.testing.assert_allclose(ncdhw(got), want, **TOL)


@pytest.mark.parametrize('shape,cout', [((1, 32, 8, 12, 12), 64), ((2, 64, 5, 9, 7), 128),
                                        ((1, 32, 16, 20, 20), 64)])
def test_conv3d_stride2_and_1x1(shape, cout):
    rs = np.random.RandomState(7)
    x = rs.standard_normal(shape).astype(np.float32)
    w = _rand_conv(rs, cout, shape[1], 3)
    want = O.conv3d(x, w, None, 2, 1)
    got = ops.conv3d_ndhwc(cl(x), ops.pack_conv_weight(T(w)), cout0=cout, ksize=3, stride=2)
    np.testing.assert_allclose(ncdhw(got), want, **TOL)
    w1 = _rand_conv(rs, 32, shape[1], 1)
    b1 = rs.standard_normal(32).astype(np.float32)
    want = O.conv3d(x, w1, b1, 1, 0)
    got = ops.conv3d_ndhwc(cl(x), ops.pack_conv_weight(T(w1)), None, T(b1), cout0=32, ksize=1)
    np.testing.assert_allclose(ncdhw(got), want, **TOL)


def test_conv3d_two_outputs_share_input():
    """BasicBlock3D's conv1 (+BN+ReLU) and downsample (+BN) in one pass == two separate convs."""
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1, 32, 6, 10, 12)).astype(np.float32)
    wa, wb = _rand_conv(rs, 32, 32, 3), _rand_conv(rs, 32, 32, 3)
    wpk = ops.pack_conv_weights_concat([T(wa), T(wb)])
    y0, y1 = ops.conv3d_ndhwc(cl(x), wpk, cout0=32, cout1=32, ksize=3, relu0=True, relu1=False)
    np.testing.assert_allclose(ncdhw(y0), np.maximum(O.conv3d(x, wa), 0), **TOL)
    np.testing.assert_allclose(ncdhw(y1), O.conv3d(x, wb), **TOL)


@pytest.mark.parametrize('shape', [(1, 16, 40, 48), (2, 5, 9, 11)])
def test_conv3d_outputs_into_channel_slices(shape):
    """ld_y0 / ld_y1: y0 (with an in-place residual) and y1 written into channel slices of wider
    channels-last buffers must equal the dense results, on the tiled, pipelined and gather kernels,
    and must not touch the other channels."""
    import os
    B, D, H, W = shape
    rs = np.random.RandomState(21)
    x = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    w = [torch.from_numpy(rs.standard_normal((32, 32, 3, 3, 3)).astype(np.float32) * 0.05).to(DEV) for _ in range(2)]
    wpk2 = ops.pack_conv_weights_concat(w)
    sc = T(rs.uniform(0.5, 1.5, 64).astype(np.float32)); bi = T(rs.standard_normal(64).astype(np.float32))
    res = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    for algo in (1, 4, 2):                  # tile-per-block, persistent DMA-pipelined, gather
        if True:
            d0, d1 = ops.conv3d_ndhwc(x, wpk2, sc, bi, cout0=32, cout1=32, ksize=3, relu0=True, algo=algo)
            buf = torch.full((B, D, H, W, 96), 7.0, device=DEV)
            s0, s1 = ops.conv3d_ndhwc(x, wpk2, sc, bi, cout0=32, cout1=32, ksize=3, relu0=True, algo=algo,
                                      out0=buf[..., 0:32], out1=buf[..., 64:96])
            assert s0.data_ptr() == buf.data_ptr()
            np.testing.assert_array_equal(buf[..., 0:32].cpu().numpy(), d0.cpu().numpy())
            np.testing.assert_array_equal(buf[..., 64:96].cpu().numpy(), d1.cpu().numpy())
            assert bool((buf[..., 32:64] == 7.0).all())
            # residual in place: slice holds the residual, conv adds onto it
            wp1 = ops.pack_conv_weight(w[0])
            dense = ops.conv3d_ndhwc(x, wp1, sc[:32].contiguous(), bi[:32].contiguous(), residual=res, ksize=3,
                                     relu0=True, algo=algo)
            buf2 = torch.full((B, D, H, W, 64), -3.0, device=DEV)
            buf2[..., 32:64] = res
            ops.conv3d_ndhwc(x, wp1, sc[:32].contiguous(), bi[:32].contiguous(), residual=buf2[..., 32:64], ksize=3,
                             relu0=True, algo=algo, out0=buf2[..., 32:64])
            np.testing.assert_array_equal(buf2[..., 32:64].cpu().numpy(), dense.cpu().numpy())
            assert bool((buf2[..., 0:32] == -3.0).all())
    with pytest.raises(Exception):
        ops.conv3d_ndhwc(x, wp1, residual=res, ksize=3, out0=buf2[..., 32:64])      # residual stride != y0 stride


@pytest.mark.parametrize('shape', [(1, 4, 50, 50, 128, 128, 1), (2, 3, 7, 9, 64, 32, 1), (1, 8, 20, 22, 64, 96, 2),
                                   (1, 5, 6, 7, 32, 32, 1)])
def test_conv3d_channel_split_gather(shape):
    """algo=3: input-channel chunks split over the waves of a block, partial sums reduced in LDS in
    a fixed order -- equal to the plain gather kernel up to fp32 summation order, and deterministic."""
    B, D, H, W, ci, co, st = shape
    rs = np.random.RandomState(4)
    x = T(rs.standard_normal((B, D, H, W, ci)).astype(np.float32))
    wpk = ops.pack_conv_weight(torch.from_numpy(rs.standard_normal((co, ci, 3, 3, 3)).astype(np.float32) * 0.05).to(DEV))
    sc = T(rs.uniform(0.5, 1.5, wpk.shape[2] * 32).astype(np.float32))
    bi = T(rs.standard_normal(wpk.shape[2] * 32).astype(np.float32))
    ref = ops.conv3d_ndhwc(x, wpk, sc, bi, cout0=co, ksize=3, stride=st, relu0=True, algo=2)
    y1 = ops.conv3d_ndhwc(x, wpk, sc, bi, cout0=co, ksize=3, stride=st, relu0=True, algo=3)
    y2 = ops.conv3d_ndhwc(x, wpk, sc, bi, cout0=co, ksize=3, stride=st, relu0=True, algo=3)
    np.testing.assert_allclose(y1.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(y1.cpu().numpy(), y2.cpu().numpy())


WINO_TOL = dict(rtol=3e-4, atol=3e-4)      # transform-domain products: |err| ~ 2e-5 on O(1) outputs


@pytest.mark.parametrize('shape', [(1, 32, 4, 8, 8), (2, 32, 5, 11, 13), (1, 64, 3, 9, 17), (1, 32, 16, 40, 48),
                                   (1, 64, 1, 1, 1), (1, 32, 2, 7, 3)])
@pytest.mark.parametrize('cout', [32, 64, 16])
def test_conv3d_wino_vs_oracle(shape, cout):
    """pw_conv3d_wino (Winograd F(2x2x2,3x3x3), wave-specialised persistent kernel) against the direct-form oracle conv + folded
    BN + residual + ReLU, on whole tiles, ragged edges in every axis and grids smaller than one tile."""
    rs = np.random.RandomState(hash((shape, cout)) % 2 ** 31)
    x = rs.standard_normal(shape).astype(np.float32)
    w = _rand_conv(rs, cout, shape[1], 3)
    scale = (rs.rand(cout) + 0.5).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32)
    res = rs.standard_normal((shape[0], cout) + shape[2:]).astype(np.float32)
    want = O.conv3d(x, w, None, 1, 1) * scale[None, :, None, None, None] + bias[None, :, None, None, None]
    want = np.maximum(want + res, 0)
    got = ops.conv3d_wino(cl(x), ops.pack_conv_weight_wino(T(w)), ops._pad32(T(scale), 1.0), ops._pad32(T(bias), 0.0),
                          residual=cl(res), cout0=cout, relu0=True)
    np.testing.assert_allclose(ncdhw(got), want, **WINO_TOL)
    again = ops.conv3d_wino(cl(x), ops.pack_conv_weight_wino(T(w)), ops._pad32(T(scale), 1.0), ops._pad32(T(bias), 0.0),
                            residual=cl(res), cout0=cout, relu0=True)
    assert torch.equal(got, again)           # no atomics, fixed summation order


def test_conv3d_wino_persistent_many_tiles_per_block():
    """More tiles than CUs x 1: every block of the persistent kernel walks several tiles (halo DMA of the next tile
    issued under the current tile's MFMAs) and two 32-channel chunks; against the direct MFMA kernel on the same operands."""
    rs = np.random.RandomState(11)
    x = T(rs.standard_normal((2, 16, 88, 104, 64)).astype(np.float32))           # 2*4*11*13 = 1144 tiles
    w = T(_rand_conv(rs, 64, 64, 3))
    sc = T(rs.uniform(0.5, 1.5, 64).astype(np.float32)); bi = T(rs.standard_normal(64).astype(np.float32))
    a = ops.conv3d_wino(x, ops.pack_conv_weight_wino(w), sc, bi, relu0=True)
    b = ops.conv3d_ndhwc(x, ops.pack_conv_weight(w), sc, bi, ksize=3, relu0=True)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), **WINO_TOL)
    assert torch.equal(a, ops.conv3d_wino(x, ops.pack_conv_weight_wino(w), sc, bi, relu0=True))


@pytest.mark.parametrize('shape', [(1, 4, 50, 50, 128, 128), (1, 8, 20, 28, 64, 96), (2, 5, 9, 11, 32, 160)])
def test_conv3d_wino_cout_groups(shape):
    """More than 64 output columns: work items = (tile, group of 32 or 64 columns) of the persistent kernel, residual
    and ReLU on every group, against the direct gather kernel."""
    B, D, H, W, ci, co = shape
    rs = np.random.RandomState(13)
    x = T(rs.standard_normal((B, D, H, W, ci)).astype(np.float32))
    w = T(_rand_conv(rs, co, ci, 3))
    sc = T(rs.uniform(0.5, 1.5, co).astype(np.float32)); bi = T(rs.standard_normal(co).astype(np.float32))
    res = T(rs.standard_normal((B, D, H, W, co)).astype(np.float32))
    ref = ops.conv3d_ndhwc(x, ops.pack_conv_weight(w), sc, bi, residual=res, cout0=co, ksize=3, relu0=True, algo=2)
    got = ops.conv3d_wino(x, ops.pack_conv_weight_wino(w), sc, bi, residual=res, cout0=co, relu0=True)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), **WINO_TOL)


def test_conv3d_wino_two_outputs_and_channel_slices():
    """The BasicBlock3D forms: conv1+downsample in one launch (y0 ReLU, y1 not), outputs written into
    channel slices of wider buffers, residual added in place -- same contract as conv3d_ndhwc."""
    rs = np.random.RandomState(5)
    B, D, H, W = 1, 6, 21, 19
    x = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    w = [T(_rand_conv(rs, 32, 32, 3)) for _ in range(2)]
    sc = T(rs.uniform(0.5, 1.5, 64).astype(np.float32)); bi = T(rs.standard_normal(64).astype(np.float32))
    res = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    d0, d1 = ops.conv3d_ndhwc(x, ops.pack_conv_weights_concat(w), sc, bi, cout0=32, cout1=32, ksize=3, relu0=True, algo=1)
    uw2 = ops.pack_conv_weights_wino_concat(w)
    y0, y1 = ops.conv3d_wino(x, uw2, sc, bi, cout0=32, cout1=32, relu0=True, relu1=False)
    np.testing.assert_allclose(y0.cpu().numpy(), d0.cpu().numpy(), **WINO_TOL)
    np.testing.assert_allclose(y1.cpu().numpy(), d1.cpu().numpy(), **WINO_TOL)
    assert float(y1.min()) < 0 <= float(y0.min())
    buf = torch.full((B, D, H, W, 96), 7.0, device=DEV)
    s0, s1 = ops.conv3d_wino(x, uw2, sc, bi, cout0=32, cout1=32, relu0=True, out0=buf[..., 0:32], out1=buf[..., 64:96])
    assert s0.data_ptr() == buf.data_ptr()
    np.testing.assert_array_equal(buf[..., 0:32].cpu().numpy(), y0.cpu().numpy())
    np.testing.assert_array_equal(buf[..., 64:96].cpu().numpy(), y1.cpu().numpy())
    assert bool((buf[..., 32:64] == 7.0).all())
    uw1 = ops.pack_conv_weight_wino(w[0])
    dense = ops.conv3d_wino(x, uw1, sc[:32].contiguous(), bi[:32].contiguous(), residual=res, relu0=True)
    buf2 = torch.full((B, D, H, W, 64), -3.0, device=DEV)
    buf2[..., 32:64] = res
    ops.conv3d_wino(x, uw1, sc[:32].contiguous(), bi[:32].contiguous(), residual=buf2[..., 32:64], relu0=True,
                    out0=buf2[..., 32:64])
    np.testing.assert_array_equal(buf2[..., 32:64].cpu().numpy(), dense.cpu().numpy())
    assert bool((buf2[..., 0:32] == -3.0).all())
    with pytest.raises(Exception):
        ops.conv3d_wino(x, uw1, residual=res, out0=buf2[..., 32:64])           # residual stride != y0 stride
    with pytest.raises(Exception):
        ops.conv3d_wino(x[..., :24].contiguous(), uw1)


def test_module_dispatch_wino_matches_direct():
    """modules._use_wino: under PW_PRECISION=f32 the module stack runs large 3x3x3 stride-1 layers on the Winograd kernel; the same
    block composed from the direct kernel (ops.conv3d_ndhwc) agrees to the conv tolerance and differs in the last bits."""
    import os
    rs = np.random.RandomState(9)
    blk = M.BasicBlock3D(32, 32, stride=1, downsample=M.ConvModule3d(32, 32, 3, stride=1, padding=1, bias=False, norm_cfg=dict(type='BN3d'), act_cfg=None)).to(DEV).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.from_numpy(rs.standard_normal(tuple(p_.shape)).astype(np.float32) * 0.05).to(DEV))
    x = T(rs.standard_normal((1, 16, 48, 56, 32)).astype(np.float32))
    assert M._use_wino(x, 64, 3, 1) and not M._use_wino(x[:, :2, :8, :8], 64, 3, 1) and M._use_wino(x, 128, 3, 1) and not M._use_wino(x, 48, 3, 1)
    with torch.no_grad():
        h = blk.forward_cl(x)                               # default precision: split-fp16 kernels
        os.environ['PW_PRECISION'] = 'f32'
        try:
            a = blk.forward_cl(x)
        finally:
            os.environ.pop('PW_PRECISION', None)
        w1, s1, b1 = blk.conv1.folded()
        wd, sd, bd = blk.downsample.folded()
        w2, s2, b2 = blk.conv2.folded()
        y = ops.conv3d_ndhwc(x, w1, s1, b1, cout0=32, ksize=3, relu0=True)
        idt = ops.conv3d_ndhwc(x, wd, sd, bd, cout0=32, ksize=3)
        b = ops.conv3d_ndhwc(y, w2, s2, b2, residual=idt, cout0=32, ksize=3, relu0=True)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), **WINO_TOL)
    assert not torch.equal(a, b)
    # the three implementations (h2 on the fp16 cores, Winograd fp32, direct fp32) agree; h2 sits with the direct sum
    np.testing.assert_allclose(h.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert not torch.equal(h, b)


def test_conv3d_bad_arguments_raise():
    x = torch.zeros(1, 4, 8, 8, 24, device=DEV)
    with pytest.raises(Exception):
        ops.conv3d_ndhwc(x, torch.zeros(1, 27, 1, 64, 16, device=DEV), ksize=3)   # Cin % 32 != 0


def _load(module, sd, prefix):
    own = module.state_dict()
    with torch.no_grad():
        for k in own:
            if 'num_batches_tracked' not in k:
                own[k].copy_(torch.from_numpy(sd[prefix + k]))


def _net(grid=None):
    net = M.PreWorld4DTraj(
        img_view_transformer=dict(type='LSSViewTransformerBEVStereo',
                                  grid_config=grid or S.GRID_CONFIG_FULL, input_size=S.INPUT_SIZE,
                                  in_channels=512, out_channels=32, sid=False, collapse_z=False,
                                  loss_depth_weight=0.05,
                                  depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True, bias=5.0), downsample=16),
        img_bev_encoder_backbone=dict(type='CustomResNet3D', numC_input=64, num_layer=[1, 2, 4],
                                      with_cp=False, num_channels=[32, 64, 128], stride=[1, 2, 2],
                                      backbone_output_ids=[0, 1, 2]),
        img_bev_encoder_neck=dict(type='LSSFPN3D', in_channels=224, out_channels=32),
        pre_process=dict(type='CustomResNet3D', numC_input=32, with_cp=False, num_layer=[1],
                         num_channels=[32], stride=[1], backbone_output_ids=[0]),
        occupancy_head=dict(type='OccHead', with_cp=False, use_deblock=False,
                            norm_cfg=dict(type='SyncBN', requires_grad=True), soft_weights=True,
                            final_occ_size=[200, 200, 16], empty_idx=17, num_level=1,
                            in_channels=[32], out_channel=18,
                            point_cloud_range=[-40, -40, -1, 40, 40, 5.4]),
        if_post_finetune=True)
    return net


def _load_net(seed):
    sd = S.synth_state_dict(seed)
    net = _net()
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()},
                                              strict=False)
    assert not unexpected
    assert all('depth_net' in k or 'num_batches_tracked' in k for k in missing), missing
    return net.to(DEV).eval(), sd


def test_conv_stack_golden(golden):
    """pre_process -> CustomResNet3D -> LSSFPN3D -> final_conv -> OccHead against the tensors the
    imported reference modules produced (tests/golden/conv_stack_small.npz)."""
    g = golden('conv_stack_small.npz')
    net, sd = _load_net(int(g['seed_sd']))
    Z, Y, X = [int(v) for v in g['shape']]
    rs = np.random.RandomState(int(g['seed_in']))
    bev_key = rs.standard_normal((1, 32, Z, Y, X)).astype(np.float32)
    bev_adj = rs.standard_normal((1, 32, Z, Y, X)).astype(np.float32)
    with torch.no_grad():
        pk = net.pre_process_net.forward_cl(cl(bev_key))[0]
        pa = net.pre_process_net.forward_cl(cl(bev_adj))[0]
        np.testing.assert_allclose(ncdhw(pk), g['pre_key'], **TOL)
        np.testing.assert_allclose(ncdhw(pa), g['pre_adj'], **TOL)
        feats = net.img_bev_encoder_backbone.forward_cl(torch.cat([pa, pk], -1))
        for f, k in zip(feats, ('enc0', 'enc1', 'enc2')):
            np.testing.assert_allclose(ncdhw(f), g[k], **TOL)
        nk = net.img_bev_encoder_neck.forward_cl(feats)
        np.testing.assert_allclose(ncdhw(nk), g['neck'], **TOL)
        fc = net.final_conv.forward_cl(nk)
        np.testing.assert_allclose(ncdhw(fc), g['final_conv'], **TOL)
        # reference path: OccHead.forward on the (1,C,X,Y,Z) view
        vf = M.from_channels_last_3d(fc).permute(0, 4, 3, 2, 1)      # (B,X,Y,Z,C) as :222
        vf_xyz = vf[0].permute(3, 0, 1, 2).unsqueeze(0)                # (1,C,X,Y,Z) as :306-307
        logits = net.occupancy_head([vf_xyz])['output_voxels'][0]
        np.testing.assert_allclose(logits.cpu().numpy(), g['logits'], rtol=5e-4, atol=5e-4)
        # fast path: native (Z,Y,X) buffer + permuted taps gives the same occupancy
        occ, lg = net.occupancy_head.decode_cl(fc, want_logits=True, transposed=True)
        np.testing.assert_allclose(lg.permute(0, 4, 3, 2, 1).cpu().numpy(), g['logits'],
                                   rtol=5e-4, atol=5e-4)
        occ_xyz = occ.permute(0, 3, 2, 1)[0].cpu().numpy()
        from _parity import check_argmax
        check_argmax('conv_stack_small occ vs reference', occ_xyz, g['occ'], np.moveaxis(g['logits'][0], 0, -1), 2e-3)
        assert np.array_equal(occ_xyz, lg.permute(0, 3, 2, 1, 4)[0].argmax(-1).cpu().numpy())
    # module-level (B,C,D,H,W) API == reference call convention
    with torch.no_grad():
        out = net.pre_process_net(T(bev_key))[0]
    np.testing.assert_allclose(out.cpu().numpy(), g['pre_key'], **TOL)


@pytest.mark.parametrize('shape', [(1, 16, 200, 200), (2, 5, 19, 27), (1, 4, 8, 8), (1, 16, 96, 104)])
def test_occ_head_wino_matches_direct_and_oracle(shape):
    """k_occ_head_wino (Winograd conv + fused 16->8->18 + argmax, wave-specialised persistent) against the direct
    16x16x4 MFMA kernel: logits to the conv tolerance, occupancy = argmax of its own logits, geo_occ consistent;
    on a crop the logits are checked against the oracle's OccHead restatement."""
    B, D, H, W = shape
    rs = np.random.RandomState(17)
    x = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    w0 = T(_rand_conv(rs, 16, 32, 3))
    s0 = T(rs.uniform(0.5, 1.5, 16).astype(np.float32)); b0 = T((rs.standard_normal(16) * 0.3).astype(np.float32))
    w1 = T((rs.standard_normal((8, 16)) * 0.4).astype(np.float32))
    s1 = T(rs.uniform(0.5, 1.5, 8).astype(np.float32)); b1 = T((rs.standard_normal(8) * 0.3).astype(np.float32))
    w2 = T((rs.standard_normal((18, 8)) * 0.5).astype(np.float32))
    args = (ops._pad32(s0, 1.0), ops._pad32(b0, 0.0), w1, s1, b1, w2)
    occ_d, lg_d, geo_d = ops.occ_head_fused(x, ops.pack_conv_weight16(w0), *args, want_logits=True, want_geo=True)
    occ_w, lg_w, geo_w = ops.occ_head_fused(x, ops.pack_conv_weight_wino(w0, cout_total=16), *args, want_logits=True,
                                            want_geo=True)
    np.testing.assert_allclose(lg_w.cpu().numpy(), lg_d.cpu().numpy(), rtol=5e-4, atol=5e-4)
    assert torch.equal(occ_w.long(), lg_w.argmax(-1))
    from _parity import check_argmax
    check_argmax('occ_head wino vs direct %s' % (shape,), occ_w, occ_d, lg_d, 2e-3)
    np.testing.assert_array_equal(geo_w.cpu().numpy(), np.where(occ_w.cpu().numpy() != 17, 0, 17).astype(np.uint8))
    occ_only = ops.occ_head_fused(x, ops.pack_conv_weight_wino(w0, cout_total=16), *args)
    assert torch.equal(occ_only, occ_w)                      # deterministic, logits optional
    # oracle on a corner crop (true zero padding on the low sides)
    d1, h1, w1_ = min(D, 6), min(H, 12), min(W, 12)
    xc = x[:1, :d1, :h1, :w1_].permute(0, 4, 1, 2, 3).contiguous().cpu().numpy()
    mid = np.maximum(O.conv3d(xc, w0.cpu().numpy()) * s0.cpu().numpy()[None, :, None, None, None]
                     + b0.cpu().numpy()[None, :, None, None, None], 0)
    hid = np.maximum(np.einsum('oc,bcdhw->bodhw', w1.cpu().numpy(), mid) * s1.cpu().numpy()[None, :, None, None, None]
                     + b1.cpu().numpy()[None, :, None, None, None], 0)
    want = np.einsum('oc,bcdhw->bdhwo', w2.cpu().numpy(), hid)
    got = lg_w[:1, :d1, :h1, :w1_].cpu().numpy()
    sl = (slice(None), slice(0, d1 if d1 == D else d1 - 1), slice(0, h1 if h1 == H else h1 - 1),
          slice(0, w1_ if w1_ == W else w1_ - 1))
    np.testing.assert_allclose(got[sl], want[sl], rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize('shape', [(1, 16, 200, 200), (2, 5, 19, 27), (1, 4, 8, 8), (1, 16, 96, 104), (3, 2, 9, 7)])
def test_occ_head_h2_matches_direct_and_oracle(shape):
    """k_occ_head_h2 (split-fp16 16x16x32 MFMA, register-resident weights, scalar-operand tail) against the direct fp32
    16x16x4 MFMA kernel: logits to split-fp16 accuracy, occupancy = argmax of its own logits (first maximum), geo_occ
    consistent, ragged grids (partial tiles in every axis); on a corner crop the logits against the oracle."""
    B, D, H, W = shape
    rs = np.random.RandomState(23)
    x = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    w0 = T(_rand_conv(rs, 16, 32, 3))
    s0 = T(rs.uniform(0.5, 1.5, 16).astype(np.float32)); b0 = T((rs.standard_normal(16) * 0.3).astype(np.float32))
    w1 = T((rs.standard_normal((8, 16)) * 0.4).astype(np.float32))
    s1 = T(rs.uniform(0.5, 1.5, 8).astype(np.float32)); b1 = T((rs.standard_normal(8) * 0.3).astype(np.float32))
    w2 = T((rs.standard_normal((18, 8)) * 0.5).astype(np.float32))
    occ_d, lg_d, geo_d = ops.occ_head_fused(x, ops.pack_conv_weight16(w0), ops._pad32(s0, 1.0), ops._pad32(b0, 0.0), w1, s1, b1,
                                            w2, want_logits=True, want_geo=True)
    wpk, inv = ops.pack_occ_weight_h2(w0)
    hargs = ((s0 * inv).contiguous(), b0) + ops.pack_occ_tail_h2(w1, s1, b1, w2) + (ops.occ_head_bounds(w0, s0, b0, w1, s1, b1),)
    xh = ops.f32_to_h2(x)
    occ_h, lg_h, geo_h = ops.occ_head_h2(xh, wpk, *hargs, want_logits=True, want_geo=True)
    from _parity import check_argmax, check_close
    check_close('occ_head h2 logits vs direct fp32 %s' % (shape,), lg_h, lg_d, 2e-5)
    assert torch.equal(occ_h.long(), lg_h.argmax(-1))
    check_argmax('occ_head h2 vs direct %s' % (shape,), occ_h, occ_d, lg_d, 2e-4)
    np.testing.assert_array_equal(geo_h.cpu().numpy(), np.where(occ_h.cpu().numpy() != 17, 0, 17).astype(np.uint8))
    occ_only = ops.occ_head_h2(xh, wpk, *hargs)
    assert torch.equal(occ_only, occ_h)                      # deterministic, logits optional
    # strided destinations (pw_occ_head_h2_strided): the (D,H,W) result written as the transposed (W,H,D)-contiguous array -- the
    # reference's (X,Y,Z) payload -- into rows of a (B, 2, W, H, D) buffer; bytes outside the rows stay untouched
    buf = torch.full((B, 2, W, H, D + 3), 255, dtype=torch.uint8, device=DEV)[..., :D]        # rows with a gap after every line
    o2, g2 = ops.occ_head_h2(xh, wpk, *hargs, occ=buf[:, 0].permute(0, 3, 2, 1), geo=buf[:, 1].permute(0, 3, 2, 1))
    assert torch.equal(buf[:, 0], occ_h.permute(0, 3, 2, 1)) and torch.equal(buf[:, 1], geo_h.permute(0, 3, 2, 1))
    assert o2.data_ptr() == buf.data_ptr() and bool((buf._base.reshape(B, 2, W, H, D + 3)[..., D:] == 255).all())
    # a strided occ with an auto-allocated geo (storage sized for the gapped strides), and overlapping stride sets refused
    buf2 = torch.full((B, 2, W, H, D + 3), 255, dtype=torch.uint8, device=DEV)[..., :D]
    o3, g3 = ops.occ_head_h2(xh, wpk, *hargs, occ=buf2[:, 0].permute(0, 3, 2, 1), want_geo=True)
    assert tuple(g3.stride()) == tuple(o3.stride()) and torch.equal(g3, geo_h) and torch.equal(o3, occ_h)
    if B > 1:
        with pytest.raises(Exception, match='overlap'):
            ops.occ_head_h2(xh, wpk, *hargs, occ=buf2[:1, 0].permute(0, 3, 2, 1).expand(B, D, H, W))          # batch stride 0
    if H > 1 and W > 1:
        flat = torch.zeros(B * D * H * W, dtype=torch.uint8, device=DEV)
        with pytest.raises(Exception, match='overlap'):                                                       # h and w both step by D
            ops.occ_head_h2(xh, wpk, *hargs, occ=flat.as_strided((B, D, H, W), (D * H * W, 1, D, D)))
    d1, h1, w1_ = min(D, 6), min(H, 12), min(W, 12)
    xc = x[:1, :d1, :h1, :w1_].permute(0, 4, 1, 2, 3).contiguous().cpu().numpy()
    mid = np.maximum(O.conv3d(xc, w0.cpu().numpy()) * s0.cpu().numpy()[None, :, None, None, None]
                     + b0.cpu().numpy()[None, :, None, None, None], 0)
    hid = np.maximum(np.einsum('oc,bcdhw->bodhw', w1.cpu().numpy(), mid) * s1.cpu().numpy()[None, :, None, None, None]
                     + b1.cpu().numpy()[None, :, None, None, None], 0)
    want = np.einsum('oc,bcdhw->bdhwo', w2.cpu().numpy(), hid)
    got = lg_h[:1, :d1, :h1, :w1_].cpu().numpy()
    sl = (slice(None), slice(0, d1 if d1 == D else d1 - 1), slice(0, h1 if h1 == H else h1 - 1),
          slice(0, w1_ if w1_ == W else w1_ - 1))
    check_close('occ_head h2 logits vs oracle crop %s' % (shape,), got[sl], want[sl], 2e-5)


def test_forecast_h2_storage_in_and_out():
    """pw_forecast_steps_h2 with v0 read from / states written in h2 storage equals the fp32-I/O run of the same kernel
    up to the storage rounding (2^-22 relative per value)."""
    rs = np.random.RandomState(5)
    v0 = T(rs.standard_normal((1, 4, 10, 12, 32)).astype(np.float32))
    fw1 = T((rs.standard_normal((128, 64)) * 0.2).astype(np.float32)); fw2 = T((rs.standard_normal((32, 128)) * 0.2).astype(np.float32))
    fb2 = T((rs.standard_normal(32) * 0.1).astype(np.float32))
    c1p = T((rs.standard_normal((1, 128)) * 0.3).astype(np.float32))
    packed = ops.forecast_pack_h2(fw1, fw2)
    ref = ops.forecast_steps_h2(v0, 1, packed, c1p, fb2, 3)
    out = ops.forecast_steps_h2(ops.f32_to_h2(v0), 1, packed, c1p, fb2, 3, out_h2=True)
    assert isinstance(out, ops.H2) and tuple(out.shape) == (3,) + tuple(v0.shape)
    from _parity import check_close
    check_close('forecast h2 storage I/O', ops.h2_to_f32(out), ref, 2e-6)


@pytest.mark.parametrize('shape', [(1, 8, 40, 72), (2, 4, 12, 200), (1, 16, 200, 200), (1, 4, 8, 24)])
def test_fpn3d_fuse_vs_torch_interpolate(shape):
    """k_fpn3d_fuse against ReLU(BN(conv1x1(x8) + up2(y16) + up4(y32))) built from torch's own trilinear
    align_corners=True upsampling (lss_fpn.py:132-148 after commuting the 1x1x1 conv below the upsample): W >= 32
    takes the per-tile path (decode and d/h terms hoisted), the last shape the per-row path."""
    import torch.nn.functional as F
    B, D, H, W = shape
    rs = np.random.RandomState(23)
    x8 = T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32))
    y16 = T(rs.standard_normal((B, D // 2, H // 2, W // 2, 32)).astype(np.float32))
    y32 = T(rs.standard_normal((B, max(D // 4, 1), H // 4, W // 4, 32)).astype(np.float32))
    w8 = T(_rand_conv(rs, 32, 32, 1))
    sc = T(rs.uniform(0.5, 1.5, 32).astype(np.float32)); bi = T(rs.standard_normal(32).astype(np.float32))
    got = ops.fpn3d_fuse(x8, ops.pack_conv_weight(w8), y16, y32, sc, bi, relu=True)
    up = lambda y: F.interpolate(y.permute(0, 4, 1, 2, 3), size=(D, H, W), mode='trilinear', align_corners=True)
    lin = torch.einsum('bdhwc,oc->bodhw', x8.double(), w8.view(32, 32).double()).float()
    want = torch.relu((lin + up(y16) + up(y32)) * sc.view(1, 32, 1, 1, 1) + bi.view(1, 32, 1, 1, 1)).permute(0, 2, 3, 4, 1)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-4)


def test_forecast_golden(golden):
    g = golden('forecast_small.npz')
    net, sd = _load_net(int(g['seed_sd']))
    v = np.random.RandomState(int(g['seed_v'])).standard_normal((1, 8, 8, 4, 32)).astype(np.float32)
    ego = S.ego_state(int(g['seed_ego']))
    with torch.no_grad():
        states, ef = net.forecast_cl(T(v), T(ego), 6)
    np.testing.assert_allclose(ef.cpu().numpy(), g['ego_feat'], rtol=1e-4, atol=1e-5)
    for k in range(6):
        np.testing.assert_allclose(states[k].cpu().numpy(), g['states'][k + 1], **TOL)


def test_traj_branch_golden(golden):
    """A20 through the C ABI (2x2x2 stride-2 convs on the MFMA gather kernel, global average pool,
    dense layers) against the imported reference module's outputs and the oracle's levels."""
    g = golden('traj_small.npz')
    net, sd = _load_net(int(g['seed_sd']))
    fused = np.random.RandomState(int(g['seed_v'])).standard_normal((1, 16, 16, 8, 32)).astype(np.float32)
    fused_cl = T(np.ascontiguousarray(fused.transpose(0, 3, 2, 1, 4)))           # (B,Z,Y,X,C)
    with torch.no_grad():
        down, levels = net.downscale.forward_cl(fused_cl, want_levels=True)
        ref_out = net.downscale(T(fused))                                        # reference layout API
        traj, fused_ego = net.traj_branch_cl(fused_cl, T(g['identity']))
    _, olev = O.downscale_module(fused, sd)
    for got, want in zip(levels, olev):                                          # (B,Z,Y,X,C) vs (B,C,X,Y,Z)
        np.testing.assert_allclose(got.cpu().numpy().transpose(0, 4, 3, 2, 1), want, **TOL)
    np.testing.assert_allclose(down.cpu().numpy(), g['down'], rtol=2e-4, atol=2e-5)
    assert ref_out.shape == (1, 1, 1, 1, 128)
    np.testing.assert_allclose(ref_out.view(1, 128).cpu().numpy(), g['down'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(fused_ego.cpu().numpy(), g['fused_ego'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(traj.cpu().numpy(), g['traj'], rtol=2e-4, atol=2e-5)


def test_traj_branch_odd_sizes():
    """grid extents that are not multiples of 8 (floor division at every level), two samples."""
    net, sd = _load_net(5)
    rs = np.random.RandomState(3)
    fused = rs.standard_normal((2, 18, 10, 9, 32)).astype(np.float32)             # (B,X,Y,Z,C)
    ego_feat = rs.standard_normal((2, 32)).astype(np.float32)
    with torch.no_grad():
        traj, fused_ego = net.traj_branch_cl(T(np.ascontiguousarray(fused.transpose(0, 3, 2, 1, 4))), T(ego_feat))
    otraj, oego = O.traj_branch(fused, ego_feat, sd)
    np.testing.assert_allclose(fused_ego.cpu().numpy(), oego, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(traj.cpu().numpy(), otraj, rtol=2e-4, atol=2e-5)


def test_forecast_ragged_and_batched():
    """voxel counts that are not a multiple of 32, two samples with different ego states."""
    net, sd = _load_net(5)
    rs = np.random.RandomState(1)
    v = rs.standard_normal((2, 3, 5, 7, 32)).astype(np.float32)
    ego = rs.standard_normal((2, 1, 21)).astype(np.float32)
    with torch.no_grad():
        states, ef = net.forecast_cl(T(v), T(ego), 3)
    for b in range(2):
        e = O.plan_head(ego[b].reshape(1, 21), sd)[0]
        cur = v[b]
        for k in range(3):
            cur = O.forecast_step(cur, e, sd)
            np.testing.assert_allclose(states[k, b].cpu().numpy(), cur, **TOL)


@pytest.mark.parametrize('shape', [(16, 200, 200), (8, 100, 100)])
def test_full_size_tiled_vs_gather_and_oracle_crop(shape):
    """BASELINE grid sizes: the LDS-tiled and the gather kernels are independent implementations
    and must agree everywhere; a crop (with halo) is checked against the oracle."""
    D, H, W = shape
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.standard_normal((1, D, H, W, 32)).astype(np.float32)).to(DEV)
    w = _rand_conv(rs, 32, 32, 3)
    wpk = ops.pack_conv_weight(T(w))
    a = ops.conv3d_ndhwc(x, wpk, ksize=3, algo=1)
    b = ops.conv3d_ndhwc(x, wpk, ksize=3, algo=2)
    assert float((a - b).abs().max()) < 2e-4
    # linearity: conv(2x) == 2 conv(x) exactly (power of two scaling)
    a2 = ops.conv3d_ndhwc((x * 2).contiguous(), wpk, ksize=3, algo=1)
    assert torch.equal(a2, a * 2)
    # oracle on a corner crop (zero padding side) and an interior crop
    for (d0, h0, w0) in [(0, 0, 0), (D - 6, H // 2, W - 12)]:
        xc = x[:, d0:d0 + 6, h0:h0 + 12, w0:w0 + 12].permute(0, 4, 1, 2, 3).contiguous().cpu().numpy()
        want = O.conv3d(xc, w)
        got = ncdhw(a[:, d0:d0 + 6, h0:h0 + 12, w0:w0 + 12])
        # compare voxels whose 3x3x3 support lies inside the crop or on the true boundary
        sl = (slice(None), slice(None),
              slice(0 if d0 == 0 else 1, 5), slice(0 if h0 == 0 else 1, 11), slice(0 if w0 == 0 else 1, 11))
        np.testing.assert_allclose(got[sl], want[sl], **TOL)


def test_softplus_accuracy():
    """hardware exp2/log2 softplus vs fp64 torch over the whole useful range."""
    x = torch.cat([torch.linspace(-100, 100, 200001), torch.tensor([-1e4, -30.0, -1e-8, 0.0, 1e-8, 19.999, 20.0, 20.001, 88.0, 1e4])]).to(DEV)
    y = ops.softplus(x.contiguous())
    ref = torch.nn.functional.softplus(x.double().cpu())
    err = (y.double().cpu() - ref).abs()
    rel = err / ref.clamp_min(1e-30)
    assert float(torch.minimum(err / 3e-7, rel / 1e-6).max()) <= 1.0   # abs 3e-7 or rel 1e-6


# ----------------------------------------------------------------------------- split-fp16 ("h2") path
def test_h2_round_trip_and_slices():
    """fp32 -> h2 -> fp32 is (hi + lo) * 2^e with e derived from the tensor's largest magnitude (pw_f32_to_h2 auto_exp): exact
    to 2^-22 relative for everything within 2^-15 of that maximum, 2^-38 of it absolute below -- at ANY scale (1e9 here; the
    round-2 format saturated at 65504); works on channel slices of wider buffers and under an explicit range slot."""
    rs = np.random.RandomState(5)
    x = (rs.standard_normal((2, 3, 5, 7, 64)) * np.exp(rs.uniform(-12, 6, (2, 3, 5, 7, 64)))).astype(np.float32)
    x[0, 0, 0, 0, :4] = [70000.0, -1e9, 0.0, 65504.0]
    xt = T(x)
    h0 = ops.f32_to_h2(xt)
    back = ops.h2_to_f32(h0).cpu().numpy()
    amax = float(np.abs(x).max())
    assert ops.slot_state(h0.rng) == (ops.RangeCtx.ideal_exp(amax), amax)
    err = np.abs(back.astype(np.float64) - x)
    assert (err <= np.maximum(np.abs(x) * 2.0 ** -21, amax * 2.0 ** -37)).all(), float((err / np.maximum(np.abs(x), 1e-30)).max())
    buf = torch.zeros(2, 3, 5, 7, 96, device=DEV)
    h = ops.f32_to_h2(xt[..., 32:64], out=buf[..., 64:96])
    sub = ops.h2_to_f32(h).cpu().numpy()
    a2 = float(np.abs(x[..., 32:64]).max())
    assert (np.abs(sub.astype(np.float64) - x[..., 32:64]) <= np.maximum(np.abs(x[..., 32:64]) * 2.0 ** -21, a2 * 2.0 ** -37)).all()
    assert float(buf[..., :64].abs().max()) == 0.0
    np.testing.assert_array_equal(ops.h2_to_f32(h0[..., 0:32]).cpu().numpy(), back[..., 0:32])
    # an explicit slot fixes the exponent: same bytes as a tensor whose values are all 2^e times smaller under exponent 0
    slot = torch.zeros(ops.RNG_ROW, dtype=torch.int32, device=DEV)
    slot[0] = 7
    hs = ops.f32_to_h2(xt[..., :32].contiguous(), out=ops.H2(torch.empty(2, 3, 5, 7, 32, device=DEV), slot))
    h1 = ops.f32_to_h2((xt[..., :32] / 128.0).contiguous(), out=ops.H2(torch.empty(2, 3, 5, 7, 32, device=DEV),
                                                                           torch.zeros(ops.RNG_ROW, dtype=torch.int32, device=DEV)))
    assert torch.equal(hs.buf, h1.buf) and ops.slot_state(hs.rng) == (7, float(np.abs(x[..., :32]).max()))


@pytest.mark.parametrize('shape', [(1, 32, 4, 8, 8), (2, 32, 5, 11, 13), (1, 64, 3, 9, 17), (1, 128, 4, 6, 10),
                                   (1, 32, 8, 40, 48)])
@pytest.mark.parametrize('cout', [32, 64, 128])
@pytest.mark.parametrize('fmt', ['f32', 'h2'])
def test_conv3d_h2_vs_oracle(shape, cout, fmt):
    """pw_conv3d_h2 (split-fp16 operands on the fp16 matrix cores) against the fp32 oracle: same tolerance class as the
    exact-fp32 MFMA kernels (measured ~1e-6 relative, printed), fp32 and h2 outputs / residuals, NT = 1 and 2."""
    from _parity import check_close
    rs = np.random.RandomState(hash((shape, cout)) % 2 ** 31)
    x = rs.standard_normal(shape).astype(np.float32)
    x[rs.rand(*shape) < 0.3] = 0
    w = _rand_conv(rs, cout, shape[1], 3)
    scale = (rs.rand(cout) + 0.5).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32)
    res = rs.standard_normal((shape[0], cout) + shape[2:]).astype(np.float32)
    want = O.conv3d(x, w, None, 1, 1) * scale[None, :, None, None, None] + bias[None, :, None, None, None]
    want = np.maximum(want + res, 0)
    wpk, inv = ops.pack_conv_weight_h2(T(w))
    xh = ops.f32_to_h2(cl(x))
    h2 = fmt == 'h2'
    r = ops.f32_to_h2(cl(res)) if h2 else cl(res)
    got = ops.conv3d_h2(xh, wpk, T(scale) * inv, T(bias), residual=r, cout0=cout, relu0=True, out_h2=(h2, h2))
    if h2:
        got = ops.h2_to_f32(got)
    check_close('conv3d_h2 %s %d->%d %s' % (shape[2:], shape[1], cout, fmt), ncdhw(got), want, 3e-6, atol=1e-6)


def test_conv3d_h2_two_outputs_strided_inplace_residual():
    """conv1 (+ReLU) and downsample in one pass into channel slices of a wider h2 buffer; conv2 adds onto its residual in
    place (BasicBlock3D's data flow) -- equal to the separate dense calls bit for bit."""
    rs = np.random.RandomState(3)
    B, D, H, W = 1, 6, 20, 26
    x = ops.f32_to_h2(T(rs.standard_normal((B, D, H, W, 32)).astype(np.float32)))
    wa, wb = T(_rand_conv(rs, 32, 32, 3)), T(_rand_conv(rs, 32, 32, 3))
    wpk, inv = ops.pack_conv_weights_h2_concat([wa, wb])
    sc = T(rs.uniform(0.5, 1.5, 64).astype(np.float32)) * inv
    bi = T(rs.standard_normal(64).astype(np.float32))
    y0, y1 = ops.conv3d_h2(x, wpk, sc, bi, cout0=32, cout1=32, relu0=True)
    buf = torch.full((B, D, H, W, 96), 7.0, device=DEV)
    s0, s1 = ops.conv3d_h2(x, wpk, sc, bi, cout0=32, cout1=32, relu0=True, out0=buf[..., 0:32], out1=buf[..., 64:96])
    assert torch.equal(s0.buf, y0.buf) and torch.equal(s1.buf, y1.buf) and bool((buf[..., 32:64] == 7.0).all())
    wp1, inv1 = ops.pack_conv_weight_h2(wa)
    dense = ops.conv3d_h2(y0, wp1, sc[:32].contiguous() / inv[:32] * inv1, bi[:32].contiguous(), residual=y1, relu0=True)
    ops.conv3d_h2(y0, wp1, sc[:32].contiguous() / inv[:32] * inv1, bi[:32].contiguous(), residual=s1, relu0=True, out0=s1)
    assert torch.equal(buf[..., 64:96], dense.buf)
    # separate fp32-output run of the same conv agrees with the decoded h2 result to the format's resolution
    f = ops.conv3d_h2(y0, wp1, sc[:32].contiguous() / inv[:32] * inv1, bi[:32].contiguous(), residual=y1, relu0=True, out_h2=(False, False))
    np.testing.assert_allclose(ops.h2_to_f32(dense).cpu().numpy(), f.cpu().numpy(), rtol=5e-7, atol=5e-7)


@pytest.mark.parametrize('shape,cout', [((1, 32, 8, 12, 12), 64), ((2, 64, 5, 9, 7), 128), ((1, 32, 16, 20, 20), 64)])
def test_conv3d_h2_stride2_and_1x1(shape, cout):
    """split-fp16 gather kernel: 3x3x3 stride 2 (encoder stage transitions), 1x1x1 (FPN laterals), and the chunk-split
    variant for tiny 3x3x3 grids, fp32 and h2 outputs, against the oracle."""
    from _parity import check_close
    rs = np.random.RandomState(7)
    x = rs.standard_normal(shape).astype(np.float32)
    xh = ops.f32_to_h2(cl(x))
    w = _rand_conv(rs, cout, shape[1], 3)
    wpk, inv = ops.pack_conv_weight_h2(T(w))
    got = ops.conv3d_h2(xh, wpk, inv, cout0=cout, ksize=3, stride=2, out_h2=(True, True))
    check_close('conv3d_h2 k3 s2 %s' % (shape,), ncdhw(ops.h2_to_f32(got)), O.conv3d(x, w, None, 2, 1), 3e-6, atol=1e-6)
    got = ops.conv3d_h2(xh, wpk, inv, cout0=cout, ksize=3, stride=1, algo=3 if (shape[1] // 32) % 2 == 0 else 2, out_h2=(False, False))
    check_close('conv3d_h2 k3 s1 gather %s' % (shape,), ncdhw(got), O.conv3d(x, w, None, 1, 1), 3e-6, atol=1e-6)
    w1 = _rand_conv(rs, 32, shape[1], 1)
    b1 = rs.standard_normal(32).astype(np.float32)
    wp1, inv1 = ops.pack_conv_weight_h2(T(w1))
    got = ops.conv3d_h2(xh, wp1, inv1, T(b1), cout0=32, ksize=1, out_h2=(False, False))
    check_close('conv3d_h2 k1 %s' % (shape,), ncdhw(got), O.conv3d(x, w1, b1, 1, 0), 3e-6, atol=1e-6)


@pytest.mark.parametrize('shape,cout', [((1, 32, 7, 11, 13), 64), ((2, 64, 5, 9, 18), 64), ((1, 64, 8, 20, 20), 128), ((1, 32, 16, 40, 40), 64)])
def test_conv3d_h2_stride2_tiled(shape, cout):
    """LDS-tiled split-fp16 stride-2 kernel (pw_conv3d_h2_s2.hip) in the form the encoder uses it -- conv1 (BN + ReLU) and the
    downsample conv (BN) of a stage's first block as one pass over 2 x Cout columns, two h2 destinations -- against the oracle
    and against the gather kernel (algo=2); ragged grids (tiles cut in every axis), batch 2, a strided destination."""
    from _parity import check_close
    from preworld_amd import _lib
    rs = np.random.RandomState(11)
    x = rs.standard_normal(shape).astype(np.float32)
    xh = ops.f32_to_h2(cl(x))
    w1, w2 = _rand_conv(rs, cout, shape[1], 3), _rand_conv(rs, cout, shape[1], 3)
    wpk, inv = ops.pack_conv_weights_h2_concat([T(w1), T(w2)])
    sc = T(rs.uniform(0.5, 1.5, 2 * cout).astype(np.float32))
    bi = T(rs.standard_normal(2 * cout).astype(np.float32))
    y0, y1 = ops.conv3d_h2(xh, wpk, sc * inv, bi, cout0=cout, cout1=cout, relu0=True, ksize=3, stride=2)
    assert _lib.lib().pw_last_kernel().decode().startswith('k_conv3d_h2_s2')
    scn, bin_ = sc.cpu().numpy(), bi.cpu().numpy()
    want0 = np.maximum(O.conv3d(x, w1, None, 2, 1) * scn[:cout, None, None, None] + bin_[:cout, None, None, None], 0)
    want1 = O.conv3d(x, w2, None, 2, 1) * scn[cout:, None, None, None] + bin_[cout:, None, None, None]
    check_close('s2 tiled conv1 %s' % (shape,), ncdhw(ops.h2_to_f32(y0)), want0, 3e-6, atol=2e-6)
    check_close('s2 tiled downsample %s' % (shape,), ncdhw(ops.h2_to_f32(y1)), want1, 3e-6, atol=2e-6)
    g0, g1 = ops.conv3d_h2(xh, wpk, sc * inv, bi, cout0=cout, cout1=cout, relu0=True, ksize=3, stride=2, algo=2)
    assert _lib.lib().pw_last_kernel().decode().startswith('k_conv3d_gather')
    # (the two kernels add the 27 x Cin products in different orders: same bound as against the oracle)
    check_close('s2 tiled vs gather conv1 %s' % (shape,), ops.h2_to_f32(y0).cpu().numpy(), ops.h2_to_f32(g0).cpu().numpy(), 3e-6, atol=2e-6)
    check_close('s2 tiled vs gather downsample %s' % (shape,), ops.h2_to_f32(y1).cpu().numpy(), ops.h2_to_f32(g1).cpu().numpy(), 3e-6, atol=2e-6)
    # strided destination (channel slice of a wider buffer), untouched neighbours
    B, Do, Ho, Wo = y0.shape[:4]
    buf = torch.full((B, Do, Ho, Wo, 3 * cout), 7.0, device=DEV)
    s0, s1 = ops.conv3d_h2(xh, wpk, sc * inv, bi, cout0=cout, cout1=cout, relu0=True, ksize=3, stride=2,
                           out0=buf[..., 0:cout], out1=buf[..., 2 * cout:3 * cout])
    assert torch.equal(s0.buf, y0.buf) and torch.equal(s1.buf, y1.buf) and bool((buf[..., cout:2 * cout] == 7.0).all())


def _slot_with_exp(e):
    slot = torch.zeros(ops.RNG_ROW, dtype=torch.int32, device=DEV)
    slot[0] = e
    return slot


def test_pool_h2_and_fpn_h2():
    """voxel pooling with h2 output = the fp32 pooled sums split (bit-identical to converting the fp32 result); the fused
    neck with h2 input / output against its exact-fp32 self."""
    from test_gpu_lss import _prepare, vsort
    from _parity import check_close
    gc = S.GRID_CONFIG_C1
    rig = S.synthetic_rig(1)
    fr, lower, interval, size, vox, _ = _prepare(gc, S.INPUT_SIZE, S.DOWNSAMPLE, rig, 1, 1)
    D, H, W = fr.shape[:3]
    vs = vsort(vox, size[0] * size[1] * size[2], D, H * W)
    depth, feat = S.lift_inputs(7, B=1, N=1)
    d_t, f_t = T(depth), T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
    ref = ops.bev_pool_dense(d_t, f_t, vs)
    h = ops.bev_pool_dense(d_t, f_t, vs, out_h2=True)                      # private slot, exponent 0 (sums of softmax weights x N(0,1): O(1))
    assert torch.equal(h.buf, ops.f32_to_h2(ref, out=ops.H2(torch.empty_like(ref), torch.zeros(ops.RNG_ROW, dtype=torch.int32, device=DEV))).buf)
    assert ops.slot_state(h.rng) == (0, float(ref.abs().max()))
    # under a range slot with exponent -3 the sums are stored 8x larger: same values up to the storage rounding
    h3 = ops.bev_pool_dense(d_t, f_t, vs, out_h2=True, out=ops.H2(torch.empty_like(ref), _slot_with_exp(-3)))
    d3 = (ops.h2_to_f32(h3) - ref).abs()
    assert bool((d3 <= torch.maximum(ref.abs() * 2.0 ** -21, torch.full_like(ref, 2.0 ** -27))).all())
    assert ops.slot_state(h3.rng) == (-3, float(ref.abs().max()))
    # neck: x8 (32 ch), x16 (64 ch at 1/2), x32 (128 ch at 1/4)
    rs = np.random.RandomState(2)
    neck = M.LSSFPN3D(in_channels=224, out_channels=32).to(DEV).eval()
    with torch.no_grad():
        neck.conv.bn.running_var.uniform_(0.5, 1.5); neck.conv.bn.running_mean.normal_(0, 0.1)
        neck.conv.bn.weight.uniform_(0.5, 1.5); neck.conv.bn.bias.normal_(0, 0.1)
    feats = [T(rs.standard_normal((1, 8, 16, 24, 32)).astype(np.float32)), T(rs.standard_normal((1, 4, 8, 12, 64)).astype(np.float32)),
             T(rs.standard_normal((1, 2, 4, 6, 128)).astype(np.float32))]
    import os
    os.environ['PW_PRECISION'] = 'f32'
    try:
        with torch.no_grad():
            want = neck.forward_cl(feats)
    finally:
        os.environ.pop('PW_PRECISION')
    with torch.no_grad():
        got = neck.forward_cl(feats)
        got_h2 = neck.forward_cl([ops.f32_to_h2(f) for f in feats], out_h2=True)
    check_close('fpn3d h2 vs f32 path', got, want.cpu().numpy(), 3e-6, atol=1e-6)
    check_close('fpn3d h2 in/out', ops.h2_to_f32(got_h2), want.cpu().numpy(), 3e-6, atol=1e-6)


def test_sustained_mfma_probe_reports_a_plausible_rate():
    """pw_probe_mfma_f16 (bench.py's roofline.sustained_mfma): a bare fp16 MFMA stream on random operands -- above the split-fp16
    conv kernels' executed rate, below the 2.5 PFLOP/s data-sheet peak."""
    import ctypes
    from preworld_amd import _lib as L
    tf = ctypes.c_double(0.0)
    L.call('pw_probe_mfma_f16', 0.2, ctypes.byref(tf))
    print('[probe] bare v_mfma_f32_32x32x16_f16 stream, random operands: %.0f TFLOP/s' % tf.value)
    assert 600.0 < tf.value < 2600.0, tf.value
