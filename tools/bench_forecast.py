"""Sustained time of pw_forecast_steps_h2 (6 steps, 640 000 voxels, h2 in / h2 out) for the library selected by PW_LIB_PATH."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from preworld_amd import ops

dev = 'cuda:0'
torch.manual_seed(0)
fw1 = torch.randn(128, 64, device=dev) * 0.1
fw2 = torch.randn(32, 128, device=dev) * 0.1
fb2 = torch.randn(32, device=dev)
packed = ops.forecast_pack_h2(fw1, fw2)
x = ops.f32_to_h2(torch.randn(1, 16, 200, 200, 32, device=dev))
c1p = torch.randn(1, 128, device=dev) * 0.3
st = ops.H2(torch.empty(6, 1, 16, 200, 200, 32, device=dev), ops.new_slot(dev))
fn = lambda: ops.forecast_steps_h2(x, 1, packed, c1p, fb2, 6, states=st, out_h2=True)   # noqa: E731
for _ in range(5):
    fn()
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    fn()
e.record()
torch.cuda.synchronize()
print(os.path.basename(os.environ.get('PW_LIB_PATH', 'real')), 'forecast 6 steps: %.1f us' % (s.elapsed_time(e) / 50 * 1e3), flush=True)
