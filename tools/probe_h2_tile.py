"""Phase timestamps of the tile-per-block split-fp16 conv kernel (PW_H2_TILE=1): per block and wave
{start, halo landed, taps done, epilogue done}."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
cin = int(os.environ.get('CIN', 32)); cout = int(os.environ.get('COUT', 32))
x = torch.randn(1, 16, 200, 200, cin, device=dev)
buf = torch.zeros(4096 * 4 * 4, dtype=torch.int64, device=dev)
os.environ['PW_CONV_PROBE'] = str(buf.data_ptr())
os.environ['PW_H2_TILE'] = '1'
from preworld_amd import ops
wh, inv = ops.pack_conv_weight_h2(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
xh = ops.f32_to_h2(x)
for _ in range(3):
    buf.zero_()
    ops.conv3d_h2(xh, wh, inv)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(4096, 4, 4).astype(np.float64)
v = t[..., 0] > 0
print('blocks recorded', int(v[:, 0].sum()))
for k, name in enumerate(['halo DMA wait', 'taps', 'epilogue']):
    d = (t[..., k + 1] - t[..., k])[v]
    print('%-14s mean %8.0f p10 %8.0f p50 %8.0f p90 %8.0f' % (name, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
life = (t[..., 3] - t[..., 0])[v]
print('block lifetime mean %.0f; kernel span %.0f cycles; sum of lifetimes / (span * 512 slots) = %.2f' % (
    life.mean(), t[..., 3][v].max() - t[..., 0][v].min(), life.sum() / 4 / ((t[..., 3][v].max() - t[..., 0][v].min()) * 512)))
