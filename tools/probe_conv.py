"""Per-block phase timestamps of the tiled conv kernel (development aid).
PW_CONV_PROBE=<device ptr> makes k_conv3d_k3s1 write {start, staged, taps done, end} cycle
counters for every wave; this script launches one conv and prints phase statistics."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
torch.manual_seed(0)
x32 = torch.randn(1, 16, 200, 200, 32, device=dev)
nblk = 2500
buf = torch.zeros(nblk * 4 * 4, dtype=torch.int64, device=dev)
os.environ['PW_CONV_PROBE'] = str(buf.data_ptr())
from preworld_amd import ops  # noqa: E402
w = ops.pack_conv_weight(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05)
for _ in range(3):
    ops.conv3d_ndhwc(x32, w, ksize=3, algo=1)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nblk, 4, 4).astype(np.float64)
for k, name in enumerate(['stage (incl. B0 load)', 'taps', 'epilogue']):
    a = t[..., k + 1] - t[..., k]
    print('%-22s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f' %
          (name, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()))
life = t[..., 3] - t[..., 0]
print('block life mean %.0f' % life.mean())
t0 = t[..., 0].min()
starts = np.sort(t[:, 0, 0] - t0)
ends = np.sort(t[:, 0, 3] - t0)
gaps = []
# idle time between a wave ending and the next wave starting in the same slot is not visible here;
# report the kernel span and the sum of lives per slot instead
print('kernel span %.0f  sum(life)/2048 slots %.0f' % (ends[-1], life.sum() / 2048.0))
