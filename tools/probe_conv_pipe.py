"""Per-stage phase timestamps of the persistent DMA-pipelined conv kernel (development aid).
PW_CONV_PROBE=<device ptr> makes k_conv3d_k3s1_pipe write, for each wave and each of its first 16
stages, the cycle counter at {stage start, taps done, barrier passed, epilogue done}."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
torch.manual_seed(0)
cin = int(os.environ.get('CIN', 32)); cout = int(os.environ.get('COUT', 32))
dims = [int(v) for v in os.environ.get('DIMS', '16,200,200').split(',')]          # DIMS=4,50,50 CIN=128 COUT=128: a small-grid layer
x = torch.randn(1, dims[0], dims[1], dims[2], cin, device=dev)
nblk = 256
buf = torch.zeros(nblk * 8 * 16 * 4 + 8 * 27, dtype=torch.int64, device=dev)
os.environ['PW_CONV_PROBE'] = str(buf.data_ptr())
from preworld_amd import ops  # noqa: E402
if os.environ.get('OCC', '0') == '1':         # k_occ_head_h2 (same probe layout; no per-tap table)
    import numpy as _np
    rs = _np.random.RandomState(0)
    T = lambda a: torch.from_numpy(_np.ascontiguousarray(a)).to(dev)
    w0 = T((rs.standard_normal((16, 32, 3, 3, 3)) * 0.05).astype(_np.float32))
    wpk, inv = ops.pack_occ_weight_h2(w0)
    hargs = (inv.contiguous(), T(_np.zeros(16, _np.float32))) + ops.pack_occ_tail_h2(
        T(rs.standard_normal((8, 16)).astype(_np.float32)), T(_np.ones(8, _np.float32)), T(_np.zeros(8, _np.float32)),
        T(rs.standard_normal((18, 8)).astype(_np.float32))) + ((30.0, 0.0, 30.0, 0.0),)
    xh = ops.f32_to_h2(x)
    for _ in range(3):
        buf.zero_()
        ops.occ_head_h2(xh, wpk, *hargs, want_geo=True)
elif os.environ.get('H2', '0') == '1':        # the split-fp16 kernel k_conv3d_h2 (same probe layout)
    wh, inv = ops.pack_conv_weight_h2(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    xh = ops.f32_to_h2(x)
    for _ in range(3):
        buf.zero_()
        ops.conv3d_h2(xh, wh, inv)
else:
    w = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    for _ in range(3):
        buf.zero_()
        ops.conv3d_ndhwc(x, w, ksize=3, algo=4)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
t = raw[:nblk * 8 * 16 * 4].reshape(nblk, 8, 16, 4).astype(np.float64)
taps = raw[nblk * 8 * 16 * 4:].reshape(8, 27).astype(np.float64)
valid = t[..., 0] > 0
print('stages recorded per wave: mean %.2f' % valid.sum(-1).mean())
for k, name in enumerate(['taps', 'wait+barrier', 'epilogue']):
    a = (t[..., k + 1] - t[..., k])[valid]
    print('%-14s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f' %
          (name, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()))
# stage-to-stage period of a wave
per = (t[:, :, 1:, 0] - t[:, :, :-1, 0])[valid[:, :, 1:] & valid[:, :, :-1]]
print('stage period    mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f' % (per.mean(), np.percentile(per, 10), np.percentile(per, 50), np.percentile(per, 90)))
gap = (t[:, :, 1:, 0] - t[:, :, :-1, 3])[valid[:, :, 1:] & valid[:, :, :-1]]
print('next-stage setup mean %9.0f' % gap.mean())
# per-stage index means (first stages include cold misses)
for st in range(0, 10):
    m = valid[:, :, st]
    if m.any():
        print('stage %2d: taps %8.0f  wait %7.0f  epi %7.0f' % (st, (t[:, :, st, 1] - t[:, :, st, 0])[m].mean(),
              (t[:, :, st, 2] - t[:, :, st, 1])[m].mean(), (t[:, :, st, 3] - t[:, :, st, 2])[m].mean()))
# per block (the counters of different XCDs are not synchronised): first stage start -> last stage end
spans = [t[b, :, :, 3][valid[b]].max() - t[b, :, :, 0][valid[b]].min() for b in range(nblk) if valid[b].any()]
print('block span (cycles): mean %.0f  max %.0f   (x stages/16 recorded)' % (np.mean(spans), np.max(spans)))
for w in range(8):
    m = valid[:, w, 1:9]
    if not m.any():
        continue
    print('wave %d: taps mean %8.0f  p10 %8.0f p90 %8.0f   wait mean %7.0f' % (
        w, (t[:, w, 1:9, 1] - t[:, w, 1:9, 0])[m].mean(), np.percentile((t[:, w, 1:9, 1] - t[:, w, 1:9, 0])[m], 10),
        np.percentile((t[:, w, 1:9, 1] - t[:, w, 1:9, 0])[m], 90), (t[:, w, 1:9, 2] - t[:, w, 1:9, 1])[m].mean()))
# one block, one stage: per-wave absolute times relative to stage start of wave 0
b = 17
if os.environ.get('OCC', '0') == '1':
    sys.exit(0)
for st in (3, 4):
    base = t[b, :, st, 0].min()
    print('block %d stage %d:' % (b, st), ' '.join('w%d[%.0f..%.0f]' % (w, t[b, w, st, 0] - base, t[b, w, st, 1] - base) for w in range(8)))

print('per-tap start times (block 17, stage 3), relative to the earliest wave:')
base = taps[:, 0].min()
for w in range(8):
    print('w%d' % w, ' '.join('%5.0f' % (v - base) for v in taps[w]))
print('tap durations:')
for w in (0, 4):
    print('w%d' % w, ' '.join('%5.0f' % v for v in np.diff(taps[w])))
