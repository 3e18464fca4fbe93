"""Winograd F(2x2x2,3x3x3) with split-fp16 operands on 32x32x16 MFMA tiles (pw_conv3d_wino_h2) next to the fp32 Winograd kernel
and the direct split-fp16 kernel at the three full-resolution layer shapes: error against a float64 torch conv on two sub-volumes,
and sustained time per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from preworld_amd import ops

DEV = 'cuda:0'


def timeit(fn, secs=0.4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 50 * 1e3


for cin, cout in ((32, 32), (32, 64), (64, 64)):
    torch.manual_seed(0)
    B, D, H, W = 1, 16, 200, 200
    x = torch.relu(torch.randn(B, D, H, W, cin, device=DEV)) * 1.7
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    sc = torch.rand(cout, device=DEV) + 0.5
    bi = torch.randn(cout, device=DEV) * 0.1
    y = torch.empty(B, D, H, W, cout, device=DEV)
    # float64 reference on a corner block (includes the zero padding) and an interior block
    def ref_block(d0, h0, w0, n=12):
        xs = x[:, max(d0 - 1, 0):d0 + n + 1, max(h0 - 1, 0):h0 + n + 1, max(w0 - 1, 0):w0 + n + 1].permute(0, 4, 1, 2, 3).double()
        pad = (1 if w0 == 0 else 0, 0, 1 if h0 == 0 else 0, 0, 1 if d0 == 0 else 0, 0)
        r = F.conv3d(F.pad(xs, pad), w.double())[:, :, :n, :n, :n]
        return (r * sc.double()[None, :, None, None, None] + bi.double()[None, :, None, None, None]).permute(0, 2, 3, 4, 1)
    blocks = [(0, 0, 0), (2, 96, 101)]
    refs = [ref_block(*b) for b in blocks]

    def err(out):
        e, m = 0.0, 0.0
        for (d0, h0, w0), r in zip(blocks, refs):
            o = out[:, d0:d0 + r.shape[1], h0:h0 + 12, w0:w0 + 12].double()
            e = max(e, float((o - r).abs().max())); m = max(m, float(r.abs().max()))
        return e / m
    res = {}
    uw = ops.pack_conv_weight_wino(w)
    os.environ.pop('PW_WINO_F16', None)
    f32 = lambda: ops.conv3d_wino(x, uw, sc, bi, relu0=False, out0=y)
    f32(); res['wino fp32'] = (err(y), timeit(f32))
    uwh, mul = ops.pack_conv_weight_wino_h2(w)
    xh = ops.f32_to_h2(x)
    f16b = lambda: ops.conv3d_wino_h2(xh, uwh, sc * mul, bi, out0=y, out_h2=(False, False))
    f16b(); res['wino h2 (h2 in, f32 out)'] = (err(y), timeit(f16b))
    yh = ops.H2(torch.empty(B, D, H, W, cout, device=DEV), ops.new_slot(DEV))
    f16c = lambda: ops.conv3d_wino_h2(xh, uwh, sc * mul, bi, out0=yh, out_h2=(True, True))
    f16c(); res['wino h2 (h2 in, h2 out)'] = (err(ops.h2_to_f32(yh)), timeit(f16c))
    if cout == 32:
        rr = ops.H2(torch.randn(B, D, H, W, cout, device=DEV), ops.new_slot(DEV))
        f16d = lambda: ops.conv3d_wino_h2(xh, uwh, sc * mul, bi, residual=rr, relu0=True, out0=rr, out_h2=(True, True))
        res['wino h2 (h2 in, h2 out, in-place residual)'] = (float('nan'), timeit(f16d))
    yd = ops.H2(torch.empty(B, D, H, W, cout, device=DEV), ops.new_slot(DEV))
    wpk0, inv0 = ops.pack_conv_weight_h2(w)
    h2b = lambda: ops.conv3d_h2(xh, wpk0, sc * inv0, bi, out0=yd, out_h2=(True, True))
    h2b(); res['direct h2 (h2 in, h2 out)'] = (err(ops.h2_to_f32(yd)), timeit(h2b))
    wpk, inv = ops.pack_conv_weight_h2(w)
    h2 = lambda: ops.conv3d_h2(xh, wpk, sc * inv, bi, out0=y, out_h2=(False, False))
    h2(); res['direct split-fp16'] = (err(y), timeit(h2))
    gf = 2.0 * B * D * H * W * 27 * cin * cout / 1e9
    print('%d->%d\n  ' % (cin, cout) + '\n  '.join('%s: err %.1e, %.1f us (%.0f TF direct)' % (k, v[0], v[1], gf / v[1] * 1e3) for k, v in res.items()), flush=True)
