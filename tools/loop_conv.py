"""Loop one conv layer for LOOP_S seconds (power / clock sampling target)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops  # noqa: E402
dev = 'cuda:0'
cin = int(os.environ.get('CIN', 32)); cout = int(os.environ.get('COUT', 32))
x = torch.randn(1, 16, 200, 200, cin, device=dev)
w = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
if os.environ.get('H2') == '1':                     # the split-fp16 kernel (PW_LIB_PATH picks a variant library)
    wh, inv = ops.pack_conv_weight_h2(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05)
    xh = ops.f32_to_h2(torch.zeros_like(x) if os.environ.get('ZERO') == '1' else x)
    y = torch.empty(1, 16, 200, 200, cout, device=dev)
    run = lambda: ops.conv3d_h2(xh, wh, inv, out0=y, out_h2=(True, True))
else:
    run = lambda: ops.conv3d_ndhwc(x, w, ksize=3, algo=1)
t_end = time.time() + float(os.environ.get('LOOP_S', 8))
n = 0
torch.cuda.synchronize()
t0 = time.time()
while time.time() < t_end:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
print('cin %d cout %d pipe %s: %.1f us/launch, %.1f TFLOP/s' % (cin, cout, os.environ.get('ALGO', 'auto'), dt / n * 1e6,
      640000 * 27 * cin * cout * 2 / (dt / n) * 1e-12))
