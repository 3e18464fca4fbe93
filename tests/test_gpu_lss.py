"""GPU parity: LSS geometry / rank build / voxel pooling (HIP, through the C ABI) against
the CPU oracle and the golden vectors.  Integer outputs are compared bit-exact; the pooled
fp32 sums are bit-exact too because kernel and oracle share op order with FMA contraction off."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from preworld_amd import modules as M
from preworld_amd import ops
from preworld_amd import synth as S

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def vsort(vox, n_vox, D, HW):
    return ops.segment_sort(vox, n_vox, aux_div=D * HW, aux_mod=HW, long_threshold=ops.LONG_SEGMENT)


def test_kat_reference_abi(golden):
    """mmdet3d/ops/bev_pool_v2/bev_pool.py:145-176 on the HIP op (C=1 -> generic kernels)."""
    g = golden('kat_bev_pool_v2.npz')
    depth = T(g['depth']).requires_grad_()
    feat = T(g['feat']).requires_grad_()
    out = ops.bev_pool_v2(depth, feat, T(g['ranks_depth']), T(g['ranks_feat']), T(g['ranks_bev']),
                          (1, 1, 2, 2, 2), T(g['interval_starts']), T(g['interval_lengths']))
    loss = out.sum()
    loss.backward()
    assert abs(loss.item() - 4.4) < 1e-6
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g['out'])
    np.testing.assert_allclose(depth.grad.cpu().numpy(), g['depth_grad'])
    np.testing.assert_allclose(feat.grad.cpu().numpy(), g['feat_grad'])


def test_camera_matrices_bit_exact():
    rig = S.synthetic_rig(6, dx=-2.5)
    ipr, comb, tr = ops.lss_camera_matrices(T(rig['sensor2ego']), T(rig['intrin']), T(rig['post_rot']))
    o_ipr, o_comb, o_tr = O.camera_matrices(rig['sensor2ego'], rig['intrin'], rig['post_rot'])
    np.testing.assert_array_equal(ipr.cpu().numpy().reshape(-1, 3, 3), o_ipr)
    np.testing.assert_array_equal(comb.cpu().numpy().reshape(-1, 3, 3), o_comb)
    np.testing.assert_array_equal(tr.cpu().numpy().reshape(-1, 3), o_tr)


def _prepare(gc, input_size, downsample, rig, B, N):
    fr = O.create_frustum(gc['depth'], input_size, downsample)
    lower, interval, size = O.grid_infos(gc)
    ipr, comb, tr = ops.lss_camera_matrices(T(rig['sensor2ego']), T(rig['intrin']), T(rig['post_rot']))
    vox, coor = ops.lss_voxel_index(T(fr), ipr, T(rig['post_tran']), comb, tr, T(rig['bda']),
                                    lower, interval, size, B, N, return_coor=True)
    return fr, lower, interval, size, vox, coor


def test_golden_small_coor_and_ranks(golden):
    g = golden('lss_small.npz')
    gc = {'x': list(g['grid_x']), 'y': list(g['grid_y']), 'z': list(g['grid_z']),
          'depth': list(g['grid_depth'])}
    lower, interval, size = O.grid_infos(gc)
    tr = T(g['sensor2ego'][:, :, :3, 3])
    # with the reference's own torch.inverse matrices the coordinates are bit-identical
    vox, coor = ops.lss_voxel_index(T(g['frustum']), T(g['inv_post_rot']), T(g['post_tran']),
                                    T(g['combine']), tr, T(g['bda']), lower, interval, size, 1, 2,
                                    return_coor=True)
    np.testing.assert_array_equal(coor.cpu().numpy(), g['coor'])
    n_vox = size[0] * size[1] * size[2]
    D, H, W = g['frustum'].shape[:3]
    vs = vsort(vox, n_vox, D, H * W)
    seg_start, order = vs.seg_start, vs.order
    rb, rd, rf, st, ln = ops.lss_ranks(seg_start, order, n_vox, D, H * W)
    np.testing.assert_array_equal(rb.cpu().numpy(), g['ranks_bev'])
    np.testing.assert_array_equal(st.cpu().numpy(), g['interval_starts'])
    np.testing.assert_array_equal(ln.cpu().numpy(), g['interval_lengths'])
    o = O.voxel_pooling_prepare_v2(g['coor'], lower, interval, size)
    np.testing.assert_array_equal(rd.cpu().numpy(), o[1])      # stable order == oracle's
    np.testing.assert_array_equal(rf.cpu().numpy(), o[2])
    # pooled output vs the reference's (B,C,Z,Y,X)
    feat = T(np.ascontiguousarray(g['feat'].transpose(0, 1, 3, 4, 2)))
    depth = T(g['depth'])
    bev = ops.bev_pool_v2(depth, feat, rd, rf, rb, (1, size[2], size[1], size[0], 8), st, ln)
    np.testing.assert_allclose(bev.cpu().numpy(), g['bev_feat'], rtol=1e-5, atol=1e-6)
    dense = ops.bev_pool_dense(depth, feat, vs)
    np.testing.assert_array_equal(dense.view(1, size[2], size[1], size[0], 8).permute(0, 4, 1, 2, 3)
                                  .cpu().numpy(), bev.cpu().numpy())
    # backward through the autograd wrapper
    depth_g = depth.clone().requires_grad_()
    feat_g = feat.clone().requires_grad_()
    out = ops.bev_pool_v2(depth_g, feat_g, rd, rf, rb, (1, size[2], size[1], size[0], 8), st, ln)
    (out * T(g['out_grad'])).sum().backward()
    np.testing.assert_allclose(depth_g.grad.cpu().numpy(), g['depth_grad'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(feat_g.grad.permute(0, 1, 4, 2, 3).cpu().numpy(), g['feat_grad'],
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('cfg', ['C1', 'full', 'full_adj_b2'])
def test_full_size_bit_exact_vs_oracle(cfg):
    """BASELINE configs: C1 (1 cam, 100x100x8) and the 6-cam 200x200x16 grid, plus B=2 with a
    translated adjacent-frame rig.  Every integer output and the pooled sums are bit-exact."""
    if cfg == 'C1':
        gc, N, B = S.GRID_CONFIG_C1, 1, 1
        rig = S.synthetic_rig(1)
    elif cfg == 'full':
        gc, N, B = S.GRID_CONFIG_FULL, 6, 1
        rig = S.synthetic_rig(6)
    else:
        gc, N, B = S.GRID_CONFIG_FULL, 6, 2
        r0, r1 = S.synthetic_rig(6), S.synthetic_rig(6, dx=-2.5)
        rig = {k: np.concatenate([r0[k], r1[k]], 0) for k in r0}
        rig['bda'][1] = np.array([[0.99, 0.1, 0], [-0.1, 0.99, 0], [0, 0, 1.0]], np.float32)
    fr, lower, interval, size, vox, coor = _prepare(gc, S.INPUT_SIZE, S.DOWNSAMPLE, rig, B, N)
    o_ipr, o_comb, o_tr = O.camera_matrices(rig['sensor2ego'], rig['intrin'], rig['post_rot'])
    o_coor = O.lidar_coor(fr, o_ipr, rig['post_tran'].reshape(-1, 3), o_comb, o_tr, rig['bda'], B, N)
    np.testing.assert_array_equal(coor.cpu().numpy(), o_coor)
    np.testing.assert_array_equal(vox.cpu().numpy(), O.voxel_index(o_coor, lower, interval, size))
    n_vox = B * size[0] * size[1] * size[2]
    D, H, W = fr.shape[:3]
    vs = vsort(vox, n_vox, D, H * W)
    seg_start, order = vs.seg_start, vs.order
    got = ops.lss_ranks(seg_start, order, n_vox, D, H * W)
    # the feat-pixel payload equals ranks_feat; the long list holds exactly the > LONG_SEGMENT voxels
    np.testing.assert_array_equal(vs.order_feat[:len(got[2])].cpu().numpy(), got[2].cpu().numpy())
    lens = (seg_start[1:] - seg_start[:-1])
    long_ref = torch.nonzero(lens > ops.LONG_SEGMENT).flatten().cpu().numpy()
    nl = int(vs.n_long)
    assert nl == len(long_ref) and nl > 0
    np.testing.assert_array_equal(np.sort(vs.long_list[:nl].cpu().numpy()), long_ref)
    want = O.voxel_pooling_prepare_v2(o_coor, lower, interval, size)
    for a, b, name in zip(got, want, ('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths')):
        np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=name)
    depth, feat = S.lift_inputs(7, B=B, N=N)
    featc = np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2))
    o_bev = O.bev_pool_v2(depth, featc, want[1], want[2], want[0],
                          (B, size[2], size[1], size[0], 32), want[3], want[4])
    d_t, f_t = T(depth), T(featc)
    dense = ops.bev_pool_dense(d_t, f_t, vs)
    dense = dense.view(B, size[2], size[1], size[0], 32).permute(0, 4, 1, 2, 3).cpu().numpy()
    np.testing.assert_array_equal(dense, o_bev)
    ref_abi = ops.bev_pool_v2(d_t, f_t, got[1], got[2], got[0], (B, size[2], size[1], size[0], 32),
                              got[3], got[4]).cpu().numpy()
    np.testing.assert_array_equal(ref_abi, o_bev)


def _rig_cfg(cfg):
    if cfg == 'C1':
        return S.GRID_CONFIG_C1, 1, 1, S.synthetic_rig(1)
    if cfg == 'full':
        return S.GRID_CONFIG_FULL, 6, 1, S.synthetic_rig(6)
    if cfg == 'full_aug':
        # training-style augmentation: every camera's image rotated / scaled / shifted, the BEV rotated and scaled
        rig = S.synthetic_rig(6)
        for n in range(6):
            a, sc = 0.05 * (n - 2), 0.9 + 0.03 * n
            rig['post_rot'][0, n, :2, :2] = sc * np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            rig['post_tran'][0, n, :2] = [3.0 * n - 5, -280 + 7 * n]
        a = 0.05
        rig['bda'][0] = (np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * np.array([1.02, 1.02, 1.0])[None, :])
        return S.GRID_CONFIG_FULL, 6, 1, rig
    r0, r1 = S.synthetic_rig(6), S.synthetic_rig(6, dx=-2.5)
    rig = {k: np.concatenate([r0[k], r1[k]], 0) for k in r0}
    rig['bda'][1] = np.array([[0.99, 0.1, 0], [-0.1, 0.99, 0], [0, 0, 1.0]], np.float32)
    return S.GRID_CONFIG_FULL, 6, 2, rig


@pytest.mark.parametrize('cfg', ['C1', 'full', 'full_adj_b2', 'full_aug'])
def test_lift_pool_slots_bit_exact_vs_oracle_and_sort_path(cfg):
    """ops.lss_lift_pool (id slots + in-group sort, no sort kernels) at the BASELINE sizes, and with a rotated / scaled image
    augmentation per camera and a rotated + scaled bda ('full_aug', round 5): the pooled fp32 sums equal the oracle's
    and the sort-based path's bit for bit; the h2 form equals the sort-based path's h2 bytes and range maximum; two runs agree
    (the slot arrival order differs from run to run, the result must not)."""
    gc, N, B, rig = _rig_cfg(cfg)
    fr, lower, interval, size, vox, coor = _prepare(gc, S.INPUT_SIZE, S.DOWNSAMPLE, rig, B, N)
    n_vox = B * size[0] * size[1] * size[2]
    D, H, W = fr.shape[:3]
    want = O.voxel_pooling_prepare_v2(coor.cpu().numpy(), lower, interval, size)
    depth, feat = S.lift_inputs(7, B=B, N=N)
    featc = np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2))
    o_bev = O.bev_pool_v2(depth, featc, want[1], want[2], want[0], (B, size[2], size[1], size[0], 32), want[3], want[4])
    d_t, f_t = T(depth), T(featc)
    cams = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
    out = torch.full((n_vox, 32), float('nan'), device=DEV)          # every voxel must be written
    got = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, d_t, f_t, out=out)
    np.testing.assert_array_equal(got.view(B, size[2], size[1], size[0], 32).permute(0, 4, 1, 2, 3).cpu().numpy(), o_bev)
    again = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, d_t, f_t)
    assert torch.equal(got, again)
    vs = vsort(vox, n_vox, D, H * W)
    ref_h2 = ops.bev_pool_dense(d_t, f_t, vs, out_h2=True)
    got_h2 = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, d_t, f_t, out_h2=True)
    assert torch.equal(got_h2.buf.view(torch.int32), ref_h2.buf.view(torch.int32))
    a, b = ops.slot_state(ref_h2.rng), ops.slot_state(got_h2.rng)
    assert a == b and b[1] > 0


@pytest.mark.parametrize('case', ['one_voxel', 'ragged', 'all_outside'])
def test_lift_pool_slots_edge_cases(case):
    """the heavy-voxel classes of ops.lss_lift_pool at their extremes: every frustum point in ONE voxel (a 5 632-point segment:
    block sort in two LDS passes), a ragged tiny frustum (point count not a multiple of 64, voxel count not a multiple of 64),
    and a rig that looks away from the grid (nothing kept: all zeros)."""
    rs = np.random.RandomState(3)
    if case == 'one_voxel':
        gc = dict(x=[-400., 400., 800.], y=[-400., 400., 800.], z=[-100., 100., 200.], depth=[1.0, 9.0, 1.0])   # a 1x1x1 grid
        N, input_size, ds = 2, (16, 22), 1
    elif case == 'ragged':
        gc = dict(x=[-10., 10., 4.0], y=[-10., 10., 2.5], z=[-1., 5.4, 0.8], depth=[1.0, 14.0, 1.0])
        N, input_size, ds = 3, (80, 176), 16
    else:
        gc = dict(x=[500., 540., 0.4], y=[500., 540., 0.4], z=[-1., 5.4, 0.4], depth=[1.0, 45.0, 0.5])
        N, input_size, ds = 2, (256, 704), 16
    rig = S.synthetic_rig(N)
    fr = O.create_frustum(gc['depth'], input_size, ds)
    lower, interval, size = O.grid_infos(gc)
    D, H, W = fr.shape[:3]
    depth = rs.random_sample((1, N, D, H, W)).astype(np.float32)
    featc = rs.standard_normal((1, N, H, W, 32)).astype(np.float32)
    cams = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
    got = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, T(depth), T(featc))
    ipr, comb, tr = O.camera_matrices(rig['sensor2ego'], rig['intrin'], rig['post_rot'])
    coor = O.lidar_coor(fr, ipr, rig['post_tran'].reshape(-1, 3), comb, tr, rig['bda'], 1, N)
    want = O.voxel_pooling_prepare_v2(coor, lower, interval, size)
    shape = (1, size[2], size[1], size[0], 32)
    if want[0] is None or len(want[0]) == 0:
        assert case == 'all_outside'
        o_bev = np.zeros((1, 32, size[2], size[1], size[0]), np.float32)
    else:
        assert case != 'all_outside'
        o_bev = O.bev_pool_v2(depth, featc, want[1], want[2], want[0], shape, want[3], want[4])
        if case == 'one_voxel':
            assert len(want[0]) == N * D * H * W > 4096
    np.testing.assert_array_equal(got.view(*shape).permute(0, 4, 1, 2, 3).cpu().numpy(), o_bev)


def test_lift_pool_slots_class_boundaries():
    """ops.lss_lift_pool at the exact boundaries of its voxel classes: voxels holding 0, 1, 7, 8 (all slots), 9 (first heavy one),
    32, 33 (second shuffle round of the half-wave sort), 63, 64, 65 (first block-sorted one), 127, 128, 129 (LDS row chunks), 4096
    and 4097 points (second LDS sort pass), their points scattered over the frustum in a random order.  With identity cameras a
    frustum entry (x, y, 1) IS the point, so the counts are exact by construction."""
    counts = [0, 1, 7, 8, 9, 32, 33, 63, 64, 65, 127, 128, 129, 4096, 4097, 2, 3, 0, 8, 9]
    rs = np.random.RandomState(17)
    gx, gy = 5, 4                                        # 20 voxels of size 1 in the z = [0.5, 1.5) slab
    pts = []
    for v, c in enumerate(counts):
        x0, y0 = v % gx, v // gx
        pts.append(np.stack([x0 + rs.uniform(0.05, 0.95, c), y0 + rs.uniform(0.05, 0.95, c), np.ones(c)], 1))
    pts = np.concatenate(pts).astype(np.float32)
    n = len(pts)
    W = 97                                               # ragged: the frustum is padded with points outside the grid
    H = (n + W - 1) // W
    pad = np.tile(np.array([[-5.0, -5.0, 1.0]], np.float32), (H * W - n, 1))
    fr = np.concatenate([pts, pad])[rs.permutation(H * W)].reshape(1, H, W, 3)
    eye = np.eye(3, dtype=np.float32)
    s2e = np.eye(4, dtype=np.float32)[None, None]
    cams = [T(s2e), T(eye[None, None]), T(eye[None, None]), T(np.zeros((1, 1, 3), np.float32)), T(eye[None])]
    lower, interval, size = [0.0, 0.0, 0.5], [1.0, 1.0, 1.0], [gx, gy, 1]
    depth = rs.random_sample((1, 1, 1, H, W)).astype(np.float32)
    featc = rs.standard_normal((1, 1, H, W, 32)).astype(np.float32)
    got = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, T(depth), T(featc))
    ipr, comb, tr = O.camera_matrices(s2e, eye[None, None], eye[None, None])
    coor = O.lidar_coor(fr, ipr, np.zeros((1, 3), np.float32), comb, tr, eye[None], 1, 1)
    want = O.voxel_pooling_prepare_v2(coor, np.array(lower, np.float32), np.array(interval, np.float32), size)
    assert np.bincount(want[0], minlength=gx * gy).tolist() == counts
    o_bev = O.bev_pool_v2(depth, featc, want[1], want[2], want[0], (1, 1, gy, gx, 32), want[3], want[4])
    np.testing.assert_array_equal(got.view(1, 1, gy, gx, 32).permute(0, 4, 1, 2, 3).cpu().numpy(), o_bev)


def test_full_size_properties(golden):
    """Size-independent properties at BASELINE's full size: linearity in feat, conservation
    (sum of pooled == sum over kept points of depth*feat), idempotent re-run, golden stats."""
    g = golden('lss_full_stats.npz')
    rig = S.synthetic_rig(6)
    fr, lower, interval, size, vox, _ = _prepare(S.GRID_CONFIG_FULL, S.INPUT_SIZE, S.DOWNSAMPLE,
                                                 rig, 1, 6)
    n_vox = 640000
    vs = vsort(vox, n_vox, 88, 32 * 88)
    seg_start, order = vs.seg_start, vs.order
    kept = int(seg_start[-1])
    assert abs(kept - int(g['P_kept'])) <= 64                # closed-form vs LAPACK 3x3 inverse
    lens = (seg_start[1:] - seg_start[:-1])
    assert abs(int((lens > 0).sum()) - int(g['n_intervals'])) <= 64
    # sortedness / permutation property of the order array
    o = order[:kept].long()
    assert torch.unique(o).numel() == kept
    assert bool((vox[o] >= 0).all())
    assert bool((vox[o][1:] >= vox[o][:-1]).all())
    depth, feat = S.lift_inputs(int(g['seed_lift']))
    d_t = T(depth)
    f_t = T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
    a = ops.bev_pool_dense(d_t, f_t, vs)
    b = ops.bev_pool_dense(d_t, f_t, vs)
    assert torch.equal(a, b)                                   # deterministic
    a2 = ops.bev_pool_dense(d_t, (f_t * 2).contiguous(), vs)
    assert torch.equal(a2, a * 2)                              # exact linearity (power of two)
    pf = (o // (88 * 32 * 88)) * (32 * 88) + o % (32 * 88)
    direct = (d_t.view(-1)[o].double()[:, None] * f_t.view(-1, 32)[pf].double()).sum(0)
    # conservation: fp32 per-voxel sums vs an fp64 direct sum over the kept points
    abs_sum = float((d_t.view(-1)[o].double()[:, None] * f_t.view(-1, 32)[pf].double().abs()).sum(0).max())
    np.testing.assert_allclose(a.double().sum(0).cpu().numpy(), direct.cpu().numpy(), rtol=0,
                               atol=1e-6 * abs_sum)
    rows = a[torch.from_numpy(g['sample_voxel_idx']).to(DEV)].cpu().numpy()
    bad = np.abs(rows - g['sample_rows']).max(1) > 1e-4
    assert bad.mean() < 0.01


def test_depthnet_tail_softmax_and_layout():
    """view_transformer.py:797-801 (+ :189): softmax over the depth logits and the channels-last
    context copy, against the oracle and torch's own softmax; then LSSViewTransformer.forward with
    an identity DepthNet must equal view_transform on the separately prepared tensors."""
    from preworld_amd import modules as M
    rs = np.random.RandomState(9)
    x = (rs.standard_normal((6, 120, 32, 88)) * 3).astype(np.float32)
    x[0, :88, 0, 0] = 50.0 * rs.standard_normal(88)            # a peaked pixel (max subtraction matters)
    depth, feat = ops.depthnet_tail(T(x), 88, 32)
    od, of = O.depthnet_tail(x, 88, 32)
    np.testing.assert_allclose(depth.cpu().numpy(), od, rtol=2e-6, atol=1e-9)
    np.testing.assert_array_equal(feat.cpu().numpy(), of)
    td = torch.from_numpy(x[:, :88]).softmax(dim=1).numpy()
    np.testing.assert_allclose(depth.cpu().numpy(), td, rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(depth.sum(1).cpu().numpy(), 1.0, rtol=1e-5)
    # D > 96 takes the streaming variant, ragged pixel count
    x2 = rs.standard_normal((2, 140, 5, 7)).astype(np.float32)
    d2, f2 = ops.depthnet_tail(T(x2), 100, 40)
    od2, of2 = O.depthnet_tail(x2, 100, 40)
    np.testing.assert_allclose(d2.cpu().numpy(), od2, rtol=2e-6, atol=1e-9)
    np.testing.assert_array_equal(f2.cpu().numpy(), of2)
    # module level
    rig = S.synthetic_rig(6)
    vt = M.LSSViewTransformer(grid_config=S.GRID_CONFIG_FULL, input_size=S.INPUT_SIZE, downsample=S.DOWNSAMPLE,
                              in_channels=120, out_channels=32, collapse_z=False).to(DEV)
    vt.depth_net = torch.nn.Identity()
    inp = [T(x).view(1, 6, 120, 32, 88), T(rig['sensor2ego']), None] + \
        [T(rig[k]) for k in ('intrin', 'post_rot', 'post_tran', 'bda')]
    with torch.no_grad():
        bev, dep = vt(inp)
        bev2, _ = vt.view_transform(inp, T(td), T(np.ascontiguousarray(x[:, 88:120])))
    np.testing.assert_allclose(bev.cpu().numpy(), bev2.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert dep.shape == (6, 88, 32, 88)


@pytest.mark.parametrize('channels_last', [False, True])
def test_stereo_cost_volume(golden, channels_last):
    """SURVEY 8f row 1 through the C ABI: the DepthNet cost volume in one kernel vs the reference's own
    output (tests/golden/stereo_small.npz) and the oracle, NCHW and channels_last feature storage."""
    g = golden('stereo_small.npz')
    prev, curr, k2s, K, pr, pt, fr = S.stereo_inputs(int(g['seed']))
    tp, tc = T(prev), T(curr)
    if channels_last:
        tp, tc = tp.contiguous(memory_format=torch.channels_last), tc.contiguous(memory_format=torch.channels_last)
    for bias in (0.0, 5.0):
        cv = ops.stereo_cost_volume(tp, tc, T(fr), T(k2s), T(K), T(pr), T(pt), bias=bias)
        np.testing.assert_allclose(cv.cpu().numpy(), g['cv_bias%d' % int(bias)], rtol=2e-4, atol=5e-6)
    # ragged width (W not a multiple of the 8-pixel block), more channels, vs the oracle
    prev, curr, k2s, K, pr, pt, fr = S.stereo_inputs(5, C=16, H=5, W=13, D=20, n_cams=3)
    tp, tc = T(prev), T(curr)
    if channels_last:
        tp, tc = tp.contiguous(memory_format=torch.channels_last), tc.contiguous(memory_format=torch.channels_last)
    cv = ops.stereo_cost_volume(tp, tc, T(fr), T(k2s), T(K), T(pr), T(pt), bias=5.0)
    want = O.stereo_cost_volume(prev, curr, fr, k2s, K, pr, pt, bias=5.0)
    np.testing.assert_allclose(cv.cpu().numpy(), want, rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(cv.sum(1).cpu().numpy(), 1.0, rtol=1e-5)


@pytest.mark.parametrize('shape', [dict(C=128, H=19, W=45, D=88, n_cams=2), dict(C=16, H=9, W=21, D=33, n_cams=1)])
def test_stereo_cost_volume_tile_kernel(shape):
    """The LDS-tiled channels-last kernel (default) against the point-per-lane kernel on the NCHW copy of the same features (itself
    pinned on the reference fixture):
    same taps, the group costs summed by a lane tree instead of serially -> 1e-5-level agreement; ragged tile edges, bins
    beyond the last chunk, bias rule; plus a strong-parallax pose that pushes near bins onto the direct-gather branch."""
    from preworld_amd import _lib
    prev, curr, k2s, K, pr, pt, fr = S.stereo_inputs(9, **shape)
    cl = lambda t: T(t).contiguous(memory_format=torch.channels_last)
    tp, tc = cl(prev), cl(curr)
    for variant in range(2):
        k2 = k2s.copy()
        if variant == 1:
            k2[0, :, :3, 3] = (1.9, 0.4, -2.5)            # large footprints for the near bins
        args = (T(fr), T(k2), T(K), T(pr), T(pt))
        want = ops.stereo_cost_volume(T(prev), T(curr), *args, bias=5.0)
        assert _lib.lib().pw_last_kernel().decode() == 'k_stereo_cost_volume<false>'
        got = ops.stereo_cost_volume(tp, tc, *args, bias=5.0)
        assert _lib.lib().pw_last_kernel().decode() == 'k_stereo_cost_volume_tile'
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=1e-7)
        assert float(got.sum(1).sub(1).abs().max()) < 1e-5
        assert torch.equal(got, ops.stereo_cost_volume(tp, tc, *args, bias=5.0))          # deterministic


def test_edge_cases():
    lower, interval, size = O.grid_infos(S.GRID_CONFIG_FULL)
    # nothing inside the grid -> five Nones like view_transformer.py:237-238
    vox = torch.full((1000,), -1, device=DEV, dtype=torch.int32)
    vs = vsort(vox, 640000, 8, 125)
    assert int(vs.seg_start[-1]) == 0
    assert ops.lss_ranks(vs.seg_start, vs.order, 640000, 8, 125) == (None,) * 5
    out = ops.bev_pool_dense(torch.rand(1000, device=DEV), torch.rand(125, 32, device=DEV), vs)
    assert float(out.abs().max()) == 0.0
    # every point in ONE voxel (maximum collision): long-segment path
    n = 5000
    vox = torch.full((n,), 12345, device=DEV, dtype=torch.int32)
    vs = vsort(vox, 640000, 8, n // 8)
    assert torch.equal(vs.order, torch.arange(n, device=DEV, dtype=torch.int32))
    assert int(vs.n_long) == 1 and int(vs.long_list[0]) == 12345
    depth = torch.rand(n, device=DEV)
    feat = torch.rand(n // 8, 32, device=DEV)
    out = ops.bev_pool_dense(depth, feat, vs)
    pf = torch.arange(n, device=DEV) % (n // 8)
    acc = torch.zeros(32, device=DEV)
    ref = np.zeros(32, np.float32)
    dn, fn = depth.cpu().numpy(), feat.cpu().numpy()
    for i in range(n):
        ref = (ref + fn[int(pf[i])] * dn[i]).astype(np.float32)
    np.testing.assert_array_equal(out[12345].cpu().numpy(), ref)
    assert float(out.abs().sum()) == pytest.approx(float(np.abs(ref).sum()), rel=1e-6)
    # ragged channel count (not a multiple of 4) takes the generic kernels
    feat5 = torch.rand(n // 8, 5, device=DEV)
    out5 = ops.bev_pool_dense(depth, feat5, vs)
    # and without the long-segment list (plain sequential groups)
    vs_plain = ops.segment_sort(vox, 640000, aux_div=n, aux_mod=n // 8)
    assert torch.equal(ops.bev_pool_dense(depth, feat, vs_plain), out)
    np.testing.assert_allclose(out5[12345].cpu().numpy(),
                               (feat5[pf].double() * depth.double()[:, None]).sum(0).cpu().numpy(),
                               rtol=1e-5)
    # bad arguments raise (no silent fallback)
    with pytest.raises(Exception):
        ops.bev_pool_dense(depth.cpu(), feat, vs)


def test_accelerate_reuses_the_sort_and_notices_new_camera_tensors():
    """accelerate=True (view_transformer.py:155-174,263-267: the reference precomputes the ranks once for a fixed rig): the second
    call with the SAME camera tensors skips geometry + sort and gives the same bits; new tensor objects (a new sample's poses
    -- even when the allocator hands them the old addresses) or an in-place update rebuild the sort (ADVICE)."""
    gc = S.GRID_CONFIG_C1
    calls = []

    def make(accelerate):
        vt = M.LSSViewTransformer(grid_config=gc, input_size=S.INPUT_SIZE, downsample=S.DOWNSAMPLE, in_channels=8, out_channels=32,
                                  collapse_z=False, accelerate=accelerate).to(DEV)
        orig = vt._sort_now
        vt._sort_now = lambda *a: (calls.append(1), orig(*a))[1]
        return vt

    def inputs(dx):
        rig = S.synthetic_rig(1, dx=dx)
        return [torch.empty(1, 1, 8, 32, 88, device=DEV), T(rig['sensor2ego']), None] + [T(rig[k]) for k in ('intrin', 'post_rot', 'post_tran', 'bda')]
    depth, feat = S.lift_inputs(3, N=1)
    d, f = T(depth).view(1, 88, 32, 88), T(feat).view(1, 32, 32, 88)
    ref_vt = make(False)
    vt = make(True)
    inp = inputs(0.0)
    with torch.no_grad():
        want, _ = ref_vt.view_transform(inp, d, f)
        n0 = len(calls)
        a, _ = vt.view_transform(inp, d, f)
        b, _ = vt.view_transform(inp, d, f)
        assert len(calls) == n0 + 1, 'second call with the same camera tensors must reuse the sort'
        assert torch.equal(a, want) and torch.equal(b, want)
        # a new sample: new tensor objects with different poses.  Free the old ones first so that the caching allocator is
        # likely to return the very same addresses -- the cache must still notice
        old_ptrs = [t.data_ptr() for t in inp if t is not None]
        del inp
        inp2 = inputs(-2.5)
        want2, _ = ref_vt.view_transform(inp2, d, f)
        n1 = len(calls)
        c, _ = vt.view_transform(inp2, d, f)
        assert len(calls) == n1 + 1 and torch.equal(c, want2) and not torch.equal(c, want)
        print('accelerate: %d tensors of the new sample reuse an old address' % sum(t.data_ptr() in old_ptrs for t in inp2 if t is not None))
        # in-place update of a cached tensor
        inp2[1].copy_(T(S.synthetic_rig(1, dx=0.0)['sensor2ego']))
        n2 = len(calls)
        e, _ = vt.view_transform(inp2, d, f)
        assert len(calls) == n2 + 1 and torch.equal(e, want)


def test_lift_pool_with_non_contiguous_camera_tensors():
    """Round 6 regression (found by the reference-class B = 2 training fixtures): prepare_inputs hands the view transformer per-frame
    SLICES of the (B, T, N, ...) pose tensors -- non-contiguous as soon as B > 1 -- and ops.lss_lift_pool made contiguous temporaries
    whose addresses outlived them: the four camera tensors of a call aliased one allocator block.  The pointer arguments now keep
    their tensors alive (ops._Ptr); strided inputs must give the bits of contiguous ones."""
    gc, N, B, rig = _rig_cfg('full_adj_b2')
    fr, lower, interval, size, vox, coor = _prepare(gc, S.INPUT_SIZE, S.DOWNSAMPLE, rig, B, N)
    depth, feat = S.lift_inputs(7, B=B, N=N)
    d_t, f_t = T(depth), T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
    cams = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
    want = ops.lss_lift_pool(T(fr), *cams, lower, interval, size, d_t, f_t)

    def strided(t):                                   # the (B, T, N, ...)[:, fid] slice prepare_inputs produces, T = 3
        big = torch.randn((t.shape[0], 3) + tuple(t.shape[1:]), device=DEV)
        big[:, 1] = t
        v = big[:, 1]
        assert not v.is_contiguous()
        return v
    got = ops.lss_lift_pool(T(fr), *[strided(c) for c in cams[:4]], cams[4], lower, interval, size, d_t, f_t)
    assert torch.equal(got, want)
