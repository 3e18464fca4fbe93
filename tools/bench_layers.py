"""Sustained timing of every conv layer shape of the C3 encoder (one line per shape).
LOOP_S seconds per shape.
ALGO=<0..3> picks the direct kernel (pw_conv3d_ndhwc's algo); EPI=1 adds scale/bias + in-place residual + ReLU to the Winograd runs.  ALGO=wino runs the Winograd kernel on the
shapes it supports (k3 s1, <= 64 output columns) and skips the others.  TFLOP/s are direct-form FLOPs / time."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
SHAPES = [  # (D, H, W, Cin, Cout, ksize, stride)
    (16, 200, 200, 32, 32, 3, 1), (16, 200, 200, 32, 64, 3, 1), (16, 200, 200, 64, 64, 3, 1),
    (8, 100, 100, 64, 64, 3, 1), (8, 100, 100, 64, 32, 3, 1), (8, 100, 100, 64, 128, 3, 1), (4, 50, 50, 128, 128, 3, 1), (4, 50, 50, 128, 256, 3, 1),
    (16, 200, 200, 32, 128, 3, 2), (8, 100, 100, 64, 256, 3, 2),
]
only = os.environ.get('ONLY')
loop_s = float(os.environ.get('LOOP_S', 1.0))
res = {}
for (D, H, W, ci, co, ks, st) in SHAPES:
    key = '%dx%dx%d %d->%d k%ds%d' % (D, H, W, ci, co, ks, st)
    if only and only not in key:
        continue
    x = torch.randn(1, D, H, W, ci, device=dev)
    w = ops.pack_conv_weight(torch.randn(co, ci, ks, ks, ks, device=dev) * 0.05)
    if os.environ.get('ALGO') == 'wino':
        if ks != 3 or st != 1:
            continue
        uw = ops.pack_conv_weight_wino(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
        if os.environ.get('EPI'):        # the module stack's epilogue: folded BN, in-place residual, ReLU
            sc, bi = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
            rbuf = torch.randn(1, D, H, W, co, device=dev)
            if os.environ.get('EPI') == '2':      # no residual
                fn = lambda: ops.conv3d_wino(x, uw, sc, bi, relu0=True, out0=rbuf)
            else:
                fn = lambda: ops.conv3d_wino(x, uw, sc, bi, residual=rbuf, relu0=True, out0=rbuf)
        else:
            fn = lambda: ops.conv3d_wino(x, uw)
    else:
        fn = lambda: ops.conv3d_ndhwc(x, w, ksize=ks, stride=st, algo=int(os.environ.get('ALGO', 0)))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = 0
    t0 = time.time()
    while time.time() - t0 < loop_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = (time.time() - t0) / n
    flops = 2.0 * (D // st) * (H // st) * (W // st) * ks ** 3 * ci * co
    res[key] = dict(us=round(dt * 1e6, 1), tflops=round(flops / dt * 1e-12, 1))
    print('%-28s %8.1f us  %6.1f TFLOP/s' % (key, dt * 1e6, flops / dt * 1e-12), flush=True)
print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith('PW_')}, layers=res)))
