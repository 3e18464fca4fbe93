"""k_fpn3d_fuse at the C3 shape, split-fp16 in / out (what the captured step runs): time and achieved HBM rate of the 163.8 MB.
Round 5: 69.6 us = 2.36 TB/s; a variant without the low-resolution staging (tools/build_variant.py fpn_nostage --only=pw_fpn3d.hip
-DPW_X_FPN_NOSTAGE, results garbage) 50.9 us = 3.22 TB/s: the staging costs 19 us, the streaming part itself sits at 0.40 of HBM peak.
Staging with 16-byte loads (20 instead of 64 load instructions per wave, same arithmetic) measured 70.3 us: the cost is the dependent
latency chain per 32-voxel wave, not the instruction count; not adopted."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops  # noqa: E402
from bench_h2 import timeit  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
x = torch.randn(1, 16, 200, 200, 32, device=dev)
y16 = torch.randn(1, 8, 100, 100, 32, device=dev)
y32 = torch.randn(1, 4, 50, 50, 32, device=dev)
w = torch.randn(32, 32, 1, 1, 1, device=dev) * 0.2
sc = torch.rand(32, device=dev) + 0.5
bi = torch.randn(32, device=dev) * 0.1
wp, inv = ops.pack_conv_weight_h2(w)
xh = ops.f32_to_h2(x)
t = timeit(lambda: ops.fpn3d_fuse(xh, wp, y16, y32, (sc * inv).contiguous(), bi, out_h2=True))
print('%s: k_fpn3d_fuse h2 -> h2 %.1f us = %.2f TB/s of 163.8 MB (timing only; parity: tests/test_gpu_encoder.py)' % (
    os.environ.get('PW_LIB_PATH', 'default library'), t, 163.84e6 / t * 1e-6))
