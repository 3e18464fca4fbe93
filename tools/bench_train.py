"""Training step of the voxel encoder at the C3 shape: CustomResNet3D [1,2,4] on (1,64,16,200,200), forward with batch-statistics
BatchNorm + backward (conv dgrad / wgrad, BN backward) through preworld_amd.train.  Development aid / profiles/r02_train_encoder.txt."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import _lib, modules as M, synth as S, train  # noqa: E402

dev = 'cuda:0'
sd = S.synth_state_dict(0)
enc = M.CustomResNet3D(numC_input=64, num_layer=[1, 2, 4], with_cp=False, num_channels=[32, 64, 128], stride=[1, 2, 2],
                       backbone_output_ids=[0, 1, 2])
own = enc.state_dict()
with torch.no_grad():
    for k in own:
        if 'num_batches_tracked' not in k:
            own[k].copy_(torch.from_numpy(sd['img_bev_encoder_backbone.' + k]))
enc = enc.to(dev).train()
x = torch.randn(1, 16, 200, 200, 64, device=dev).requires_grad_(True)


def step():
    for p in enc.parameters():
        p.grad = None
    x.grad = None
    feats = enc.forward_cl(x)
    loss = sum(f.sum() for f in feats)
    loss.backward()


def fwd():
    with torch.no_grad():
        enc.forward_cl(x)


for fn, name in ((fwd, 'forward (train-mode BN)'), (step, 'forward + backward')):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-28s %.2f ms' % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)

# single layers
for (D, H, W), cin, cout, s in (((16, 200, 200), 32, 32, 1), ((16, 200, 200), 64, 32, 1), ((16, 200, 200), 32, 64, 2),
                                 ((8, 100, 100), 64, 64, 1), ((8, 100, 100), 64, 128, 2), ((4, 50, 50), 128, 128, 1)):
    xx = torch.randn(1, D, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    y = train.conv3d_raw(xx, w, s)
    g = torch.randn_like(y)
    gf = 2.0 * y.numel() // cout * 27 * cin * cout / 1e9

    def t(fn):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 5 * 1e3
    tw = t(lambda: train.conv3d_wgrad(xx, g, w.shape, s))
    td = t(lambda: train.conv3d_dgrad(g, w, xx.shape, s))
    print('%dx%dx%d %d->%d s%d  %.1f GF: wgrad %.0f us (%.0f TF)  dgrad %.0f us (%.0f TF)' % (D, H, W, cin, cout, s, gf, tw, gf / tw * 1e3,
                                                                                    td, gf / td * 1e3), flush=True)
