"""pw_conv3d_wgrad_h2 alone (no amax passes) at the encoder's 3x3x3 stride-1 shapes.  (The PW_WG_DEBUG phase switches the
recorded staging-only / MFMA-only figures of profiles/r03_train.txt were taken with are no longer in the kernel: a condition around
its loads or stores costs it the overlap they measure.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import _lib, ops  # noqa: E402
from bench_h2 import timeit  # noqa: E402

dev = 'cuda:0'
for (B, D, H, W), cin, cout in (((1, 16, 200, 200), 32, 32), ((1, 16, 200, 200), 64, 32), ((1, 16, 200, 200), 64, 64),
                                ((1, 8, 100, 100), 64, 64), ((1, 4, 50, 50), 128, 128)):
    x = torch.randn(B, D, H, W, cin, device=dev)
    dy = torch.randn(B, D, H, W, cout, device=dev)
    dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
    nbytes = _lib.call_size('pw_conv3d_wgrad_h2_workspace_bytes', B, D, H, W, cin, cout)
    ws = ops._workspace(nbytes, dev)
    fn = lambda: _lib.call('pw_conv3d_wgrad_h2', ops._p(x), ops._p(dy), ops._p(dw), None, None, ops._p(ws), nbytes, B, D, H, W, cin, cout,
                           ops._stream())
    t = timeit(fn)
    gf = 2.0 * B * D * H * W * 27 * cin * cout / 1e9
    print('PW_WG_DEBUG=%s %dx%dx%d %d->%d: %.1f us (%.0f TF direct-form)' % (os.environ.get('PW_WG_DEBUG', '0'), D, H, W, cin, cout, t, gf / t * 1e3),
          flush=True)
