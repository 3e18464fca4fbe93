"""Per-kernel micro-benchmarks (HIP events on the launch stream).  Development aid only."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops, synth as S  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3   # us


def bench_lss():
    dev = 'cuda:0'
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rig = S.synthetic_rig(6)
    gc = S.GRID_CONFIG_FULL
    fr = torch.stack(torch.meshgrid(torch.arange(88.), torch.arange(32.), torch.arange(88.),
                                    indexing='ij'), -1)
    from preworld_amd.modules import create_frustum
    fr = create_frustum(gc['depth'], S.INPUT_SIZE, S.DOWNSAMPLE).to(dev)
    lower = [-40., -40., -1.]
    interval = [0.4, 0.4, 0.4]
    size = [200, 200, 16]
    s2e, K, pr, pt, bda = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
    depth, feat = S.lift_inputs(0)
    d_t = T(depth)
    f_t = T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
    res = {}
    ipr, comb, tr = ops.lss_camera_matrices(s2e, K, pr)
    res['camera_matrices_us'] = timeit(lambda: ops.lss_camera_matrices(s2e, K, pr))
    vox = ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6)
    res['voxel_index_us'] = timeit(lambda: ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6))
    srt = lambda: ops.segment_sort(vox, 640000, aux_div=88 * 32 * 88, aux_mod=32 * 88, long_threshold=ops.LONG_SEGMENT)
    vs = srt()
    seg_start, order = vs.seg_start, vs.order
    res['segment_sort_us'] = timeit(srt)
    out = torch.empty(640000, 32, device=dev)
    res['pool_dense_us'] = timeit(lambda: ops.bev_pool_dense(d_t, f_t, vs, out=out))
    kept = int(seg_start[-1])
    alg = 82e6 + 5.95e6 + 2.16e6 + kept * 4 + 2.56e6
    res['pool_dense_GBps'] = alg / res['pool_dense_us'] / 1e3
    rb, rd, rf, st, ln = ops.lss_ranks(seg_start, order, 640000, 88, 32 * 88)
    res['lss_ranks_us'] = timeit(lambda: ops.lss_ranks(seg_start, order, 640000, 88, 32 * 88))
    o2 = torch.zeros(1, 16, 200, 200, 32, device=dev)
    res['pool_intervals_us'] = timeit(lambda: ops.bev_pool_v2_forward(d_t, f_t, o2, rd, rf, rb, ln, st))
    res['memset_82MB_us'] = timeit(lambda: o2.zero_())
    res['kept'] = kept
    res['n_intervals'] = int(st.numel())
    return res


def bench_enc():
    dev = 'cuda:0'
    rs = np.random.RandomState(0)
    res = {}
    x32 = torch.randn(1, 16, 200, 200, 32, device=dev)
    x64 = torch.randn(1, 16, 200, 200, 64, device=dev)
    w = lambda co, ci, k=3: ops.pack_conv_weight(torch.randn(co, ci, k, k, k, device=dev) * 0.05)
    sc, bi = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    sc64, bi64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    def flops(ci, co, nv=640000, taps=27):
        return 2.0 * nv * taps * ci * co
    for name, x, wp, kw, fl in [
        ('k3s1_32_32_tiled', x32, w(32, 32), dict(algo=1), flops(32, 32)),
        ('k3s1_32_32_gather', x32, w(32, 32), dict(algo=2), flops(32, 32)),
        ('k3s1_32_64_tiled', x32, w(64, 32), dict(algo=1), flops(32, 64)),
        ('k3s1_64_64_tiled', x64, w(64, 64), dict(algo=1), flops(64, 64)),
        ('k3s1_32_16pad_tiled', x32, w(16, 32), dict(algo=1, cout0=16), flops(32, 32)),
    ]:
        y = ops.conv3d_ndhwc(x, wp, ksize=3, **kw)
        t = timeit(lambda: ops.conv3d_ndhwc(x, wp, ksize=3, **kw), iters=10)
        res[name + '_us'] = t
        res[name + '_TFLOPs'] = fl / t / 1e6
    xs = torch.randn(1, 8, 100, 100, 64, device=dev)
    wp = w(64, 64)
    t = timeit(lambda: ops.conv3d_ndhwc(xs, wp, ksize=3, algo=1), iters=10)
    res['L1_64_64_tiled_us'] = t; res['L1_64_64_tiled_TFLOPs'] = flops(64, 64, 80000) / t / 1e6
    xs2 = torch.randn(1, 4, 50, 50, 128, device=dev)
    wp = w(128, 128)
    t = timeit(lambda: ops.conv3d_ndhwc(xs2, wp, ksize=3, algo=1), iters=10)
    res['L2_128_128_tiled_us'] = t; res['L2_128_128_tiled_TFLOPs'] = flops(128, 128, 10000) / t / 1e6
    t = timeit(lambda: ops.conv3d_ndhwc(xs2, wp, ksize=3, algo=2), iters=10)
    res['L2_128_128_gather_us'] = t
    wp = w(64, 32)
    t = timeit(lambda: ops.conv3d_ndhwc(x32, wp, ksize=3, stride=2), iters=10)
    res['s2_32_64_gather_us'] = t
    # neck
    y16 = torch.randn(1, 8, 100, 100, 32, device=dev); y32 = torch.randn(1, 4, 50, 50, 32, device=dev)
    w8 = w(32, 32, 1)
    t = timeit(lambda: ops.fpn3d_fuse(x32, w8, y16, y32, sc, bi), iters=10)
    res['fpn_fuse_us'] = t; res['fpn_fuse_GBps'] = 164e6 / t / 1e3
    # occ head
    w1 = torch.randn(8, 16, device=dev); s1 = torch.ones(8, device=dev); b1 = torch.zeros(8, device=dev)
    w2 = torch.randn(18, 8, device=dev)
    wp16 = ops.pack_conv_weight16(torch.randn(16, 32, 3, 3, 3, device=dev) * 0.05)
    t = timeit(lambda: ops.occ_head_fused(x32, wp16, sc, bi, w1, s1, b1, w2), iters=10)
    res['occ_head_us'] = t; res['occ_head_TFLOPs_useful'] = flops(32, 16) / t / 1e6
    # forecast
    fw1 = torch.randn(128, 64, device=dev) * 0.1; fw2 = torch.randn(32, 128, device=dev) * 0.1
    w1p, w2p = ops.forecast_pack(fw1, fw2)
    c1 = torch.randn(1, 128, device=dev) * 0.1; fb2 = torch.randn(32, device=dev)
    st = torch.empty(6, 1, 16, 200, 200, 32, device=dev)
    t = timeit(lambda: ops.forecast_steps(x32, 1, w1p, w2p, c1, fb2, 6, states=st), iters=5)
    res['forecast6_us'] = t; res['forecast6_TFLOPs'] = 6 * 640000 * 2 * (32 * 128 * 2) / t / 1e6
    return res


def bench_render():
    """C5: attribute MLPs on the full grid + the fused ray march,
    (ii) the reference shape 38 400 rays x 417 samples and (i) 3 072 rays x 96 uniform samples."""
    from preworld_amd import modules as M, synth as S
    dev = 'cuda:0'
    res = {}
    sd = S.synth_state_dict(0)
    mods = []
    for name, nout in (('density_mlp', 2), ('semantic_mlp', 17), ('color_mlp', 3)):
        m = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Softplus(), torch.nn.Linear(64, nout))
        m.load_state_dict({k[len(name) + 1:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(name + '.')})
        mods.append(m.to(dev))
    packed = ops.pack_attr_mlp(*mods)
    v = torch.randn(1, 16, 200, 200, 32, device=dev)
    grid = ops.attr_mlp(v, packed, final_softplus=False)
    t = timeit(lambda: ops.attr_mlp(v, packed, final_softplus=False, out=grid), iters=10)
    res['attr_mlp_us'] = t
    res['attr_mlp_TFLOPs'] = 640000 * 2.0 * (32 * 192 + 192 * 24) / t / 1e6
    res['attr_mlp_GBps'] = 640000 * (32 + 24) * 4 / t / 1e3
    head = M.NerfHead(point_cloud_range=[-40, -40, -1, 40, 40, 5.4], voxel_size=0.4, scene_center=[0, 0, 2.2],
                      radius=39).to(dev)
    bda = torch.eye(3)
    g = grid[0].contiguous()
    g[..., 0] = torch.rand(16, 200, 200, device=dev) * 4 - 2            # sigma logits around the act shift
    for label, R, tt in (('ii_38400x417', 38400, head.t_table(dev)),
                         ('i_3072x96', 3072, torch.linspace(0, 2, 97, device=dev)[:-1] + 1.0 / 96)):
        o, d = S.rays(5, R)
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        fn = lambda: ops.render_rays(o, d, tt.contiguous(), g, head.consts(bda))
        fn()
        t = timeit(fn, iters=10)
        npts = R * tt.numel()
        res['render_%s_us' % label] = t
        res['render_%s_Mpts_per_s' % label] = npts / t
        res['render_%s_gather_GBps' % label] = npts * 8 * 96 / t / 1e3     # cache-level corner rows
        res['render_%s_hbm_alg_GBps' % label] = (g.numel() * 4 + R * 25 * 4) / t / 1e3
    return res


def bench_stereo():
    """DepthNet cost volume at the reference shape (6 cams, 128 channels, 128x352, D=88)."""
    dev = 'cuda:0'
    res = {}
    rig = S.synthetic_rig(6)
    BN, C, H, W, D = 6, 128, 128, 352, 88
    prev = torch.randn(BN, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    curr = torch.randn(BN, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    d = torch.arange(1.0, 45.0, 0.5, device=dev)
    x = torch.linspace(0, 1408 - 1, W, device=dev)
    y = torch.linspace(0, 512 - 1, H, device=dev)
    fr = torch.stack([x.view(1, 1, W).expand(D, H, W), y.view(1, H, 1).expand(D, H, W), d.view(D, 1, 1).expand(D, H, W)], -1).contiguous()
    k2s = torch.eye(4, device=dev).repeat(1, 6, 1, 1)
    k2s[0, :, 2, 3] = -2.5
    args = (fr, k2s) + tuple(torch.from_numpy(np.ascontiguousarray(rig[k])).to(dev) for k in ('intrin', 'post_rot', 'post_tran'))
    fn = lambda: ops.stereo_cost_volume(prev, curr, *args, bias=5.0)
    fn()
    t = timeit(fn, iters=5)
    npts = BN * D * H * W
    res['stereo_cv_us'] = t
    res['stereo_cv_Gpts_per_s'] = npts / t / 1e3
    res['stereo_cv_gather_GBps'] = npts * 4 * C * 4 / t / 1e3        # 4 corners x C floats per point (cache level)
    res['stereo_cv_hbm_alg_GBps'] = (2 * prev.numel() * 4 + npts * 4) / t / 1e3
    t2 = timeit(lambda: ops.stereo_cost_volume(prev.contiguous(), curr.contiguous(), *args, bias=5.0), iters=2)
    res['stereo_cv_nchw_us'] = t2
    # the point-per-lane kernel (NCHW copy of the same features) next to the tiled one
    import os
    ref = fn()
    old = ops.stereo_cost_volume(prev.contiguous(), curr.contiguous(), *args, bias=5.0)
    res['stereo_cv_max_abs_diff_to_point_kernel'] = float((ref - old).abs().max())
    res['stereo_cv_max_rel_diff_to_point_kernel'] = float(((ref - old).abs() / old.clamp_min(1e-30)).max())
    probe = torch.zeros(48, dtype=torch.int64, device=dev)
    os.environ['PW_STEREO_PROBE'] = str(probe.data_ptr())
    fn()
    torch.cuda.synchronize()
    del os.environ['PW_STEREO_PROBE']
    res['stereo_cv_tile_phase_cycles'] = dict(zip(['geometry', 'plan', 'stage', 'compute', 'barrier', 'softmax'],
                                                  probe.view(8, 6).float().mean(0).tolist()))
    # ego motion along the optical axis of every camera (the case above) vs a sideways one
    k2s2 = torch.eye(4, device=dev).repeat(1, 6, 1, 1)
    k2s2[0, :, 0, 3] = -2.5
    args2 = (fr, k2s2) + args[2:]
    fn2 = lambda: ops.stereo_cost_volume(prev, curr, *args2, bias=5.0)
    fn2()
    res['stereo_cv_sideways_us'] = timeit(fn2, iters=5)
    return res


def bench_loss():
    """CE + sem_scal + geo_scal forward (one pass) and backward (one pass) on
    (1,18,200,200,16) logits; algorithmic HBM bytes = logits read (46 MB) [+ gradient written]."""
    from preworld_amd import losses as L
    dev = 'cuda:0'
    res = {}
    pred = torch.randn(1, 18, 200, 200, 16, device=dev, requires_grad=True)
    target = torch.randint(0, 18, (1, 200, 200, 16), device=dev)
    cam = torch.rand(1, 200, 200, 16, device=dev) < 0.8
    cw = torch.rand(18, device=dev) + 0.1

    def fwd():
        return L.voxel_losses(pred, target, cw, 255, 17, cam)

    def fwd_bwd():
        pred.grad = None
        ce, sem, geo = fwd()
        (ce + sem + geo).backward()
    with torch.no_grad():
        t = timeit(fwd, iters=10)
    res['loss_fwd_us'] = t
    res['loss_fwd_GBps'] = (pred.numel() * 4 + 2 * 640000) / t / 1e3
    t2 = timeit(fwd_bwd, iters=10)
    res['loss_fwd_bwd_us'] = t2
    res['loss_bwd_GBps'] = (2 * pred.numel() * 4 + 2 * 640000) / max(t2 - t, 1e-3) / 1e3
    # the finetune configs' other two: CustomFocalLoss and lovasz_softmax (+ the same ops in plain torch as the
    # reference composes them, for scale: C sequential sorts)
    focal = L.CustomFocalLoss()

    def focal_fb():
        pred.grad = None
        focal(pred, target, cw, None, 255, camera_mask=cam).backward()
    with torch.no_grad():
        res['focal_fwd_us'] = timeit(lambda: focal(pred, target, cw, None, 255, camera_mask=cam), iters=10)
    res['focal_fwd_bwd_us'] = timeit(focal_fb, iters=10)

    def lovasz_fb():
        pred.grad = None
        L.lovasz_softmax(torch.softmax(pred, 1), target, ignore=17, camera_mask=cam).backward()
    with torch.no_grad():
        pr = torch.softmax(pred, 1)
        res['lovasz_fwd_us'] = timeit(lambda: L.lovasz_softmax(pr, target, ignore=17, camera_mask=cam), iters=5)

        def torch_lovasz():          # lovasz_softmax_flat's loop in torch ops (no autograd), valid rows gathered first
            valid = (target != 17) & cam
            vp = pr.permute(0, 2, 3, 4, 1)[valid]
            vl = target[valid]
            tot = 0.0
            for c in range(18):
                fg = (vl == c).float()
                if fg.sum() == 0:
                    continue
                es, perm = torch.sort((fg - vp[:, c]).abs(), 0, descending=True)
                fs = fg[perm]
                g = fs.sum()
                jac = 1. - (g - fs.cumsum(0)) / (g + (1 - fs).cumsum(0))
                jac[1:] = jac[1:] - jac[:-1]
                tot = tot + torch.dot(es, jac)
            return tot
        res['lovasz_fwd_torch_ops_us'] = timeit(torch_lovasz, iters=3)
    res['lovasz_fwd_bwd_incl_softmax_us'] = timeit(lovasz_fb, iters=5)
    return res


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='lss')
    a = ap.parse_args()
    out = {}
    if 'lss' in a.what:
        out['lss'] = bench_lss()
    if 'enc' in a.what:
        out['enc'] = bench_enc()
    if 'render' in a.what:
        out['render'] = bench_render()
    if 'loss' in a.what:
        out['loss'] = bench_loss()
    if 'stereo' in a.what:
        out['stereo'] = bench_stereo()
    print(json.dumps(out, indent=1))
