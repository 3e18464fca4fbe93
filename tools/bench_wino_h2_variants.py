"""Sustained time per launch of pw_conv3d_wino_h2 (h2 in, h2 out) at the three full-resolution layer shapes for the library selected
by PW_LIB_PATH (tools/build_variant.py: -DWX_X_* timing-only ablations); the direct kernel next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from preworld_amd import ops

DEV = 'cuda:0'


def timeit(fn, secs=0.3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 40 * 1e3


out = []
for cin, cout in ((32, 32), (32, 64), (64, 64)):
    torch.manual_seed(0)
    x = ops.f32_to_h2(torch.relu(torch.randn(1, 16, 200, 200, cin, device=DEV)) * 1.7)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    uw, mul = ops.pack_conv_weight_wino_h2(w)
    y = ops.H2(torch.empty(1, 16, 200, 200, cout, device=DEV), ops.new_slot(DEV))
    t = timeit(lambda: ops.conv3d_wino_h2(x, uw, mul, out0=y, out_h2=(True, True)))
    s = '%d->%d wino %.1f' % (cin, cout, t)
    if os.environ.get('DIRECT', '0') == '1':
        wpk, inv = ops.pack_conv_weight_h2(w)
        s += ' direct %.1f' % timeit(lambda: ops.conv3d_h2(x, wpk, inv, out0=y, out_h2=(True, True)))
    out.append(s)
print(os.path.basename(os.environ.get('PW_LIB_PATH', 'real')), ' | '.join(out), flush=True)
