"""Barrier arrival / departure times of the eight waves of one block of k_conv3d_wino_h2 (development aid, PW_CONV_PROBE): for the
k-th workgroup barrier of the block's second tile, when each wave ARRIVED and when it LEFT, in cycles relative to the first barrier's
release.  Waves 0-3 = GEMM role, 4-7 = transform + DMA role.  CIN / COUT select the layer (default 64 -> 64)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
torch.manual_seed(0)
cin = int(os.environ.get('CIN', 64)); cout = int(os.environ.get('COUT', 64))
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
os.environ['PW_CONV_PROBE'] = str(buf.data_ptr())
from preworld_amd import ops  # noqa: E402
x = ops.f32_to_h2(torch.relu(torch.randn(1, 16, 200, 200, cin, device=dev)))
uw, mul = ops.pack_conv_weight_wino_h2(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
y = ops.H2(torch.empty(1, 16, 200, 200, cout, device=dev), ops.new_slot(dev))
for _ in range(20):
    ops.conv3d_wino_h2(x, uw, mul, out0=y, out_h2=(True, True))
torch.cuda.synchronize()
buf.zero_()
ops.conv3d_wino_h2(x, uw, mul, out0=y, out_h2=(True, True))
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(8, 32, 2).astype(np.float64)
nb = int((t[0, :, 0] > 0).sum())
t0 = t[:, 0, 1].max()
print('layer %d -> %d: %d barriers in the tile; cycles from barrier 0 to the last barrier: %.0f' % (cin, cout, nb, t[:, nb - 1, 1].max() - t0))
print('barrier | arrival of waves G0 G1 G2 G3 | T0 T1 T2 T3 (cycles before the barrier opened) | interval since the previous barrier')
prev = t0
for k in range(1, nb):
    rel = t[:, k, 1].min()
    arr = rel - t[:, k, 0]
    print('%3d  | %s | %s | %6.0f' % (k, ' '.join('%6.0f' % v for v in arr[:4]), ' '.join('%6.0f' % v for v in arr[4:]), rel - prev))
    prev = rel
