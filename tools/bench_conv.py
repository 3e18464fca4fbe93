"""Run the dominant conv kernels a few times (target for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops  # noqa: E402

dev = 'cuda:0'
torch.manual_seed(0)
x32 = torch.randn(1, 16, 200, 200, 32, device=dev)
x64 = torch.randn(1, 16, 200, 200, 64, device=dev)
w = lambda co, ci: ops.pack_conv_weight(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
w3232, w3264, w6464 = w(32, 32), w(64, 32), w(64, 64)
if os.environ.get('WINO'):         # the Winograd kernel on the same three full-resolution shapes
    uw = lambda co, ci: ops.pack_conv_weight_wino(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
    u3232, u3264, u6464 = uw(32, 32), uw(64, 32), uw(64, 64)
    uocc = ops.pack_conv_weight_wino(torch.randn(16, 32, 3, 3, 3, device=dev) * 0.05, cout_total=16)
    sc, bi = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    w1 = torch.randn(8, 16, device=dev); s1 = torch.ones(8, device=dev); b1 = torch.zeros(8, device=dev)
    w2 = torch.randn(18, 8, device=dev)
    for _ in range(int(os.environ.get('N', 3))):
        ops.conv3d_wino(x32, u3232)
        ops.conv3d_wino(x32, u3264)
        ops.conv3d_wino(x64, u6464)
        ops.occ_head_fused(x32, uocc, sc, bi, w1, s1, b1, w2, want_geo=True)
    torch.cuda.synchronize()
    print('done')
    sys.exit(0)
for _ in range(int(os.environ.get('N', 3))):
    ops.conv3d_ndhwc(x32, w3232, ksize=3, algo=1)
    ops.conv3d_ndhwc(x32, w3264, ksize=3, algo=1)
    ops.conv3d_ndhwc(x64, w6464, ksize=3, algo=1)
torch.cuda.synchronize()
if os.environ.get('TIME'):
    import json
    def timeit(fn, iters=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3
    r = {'stagger': os.environ.get('PW_CONV_STAGGER', '1')}
    r['32_32_us'] = timeit(lambda: ops.conv3d_ndhwc(x32, w3232, ksize=3, algo=1))
    r['32_64_us'] = timeit(lambda: ops.conv3d_ndhwc(x32, w3264, ksize=3, algo=1))
    r['64_64_us'] = timeit(lambda: ops.conv3d_ndhwc(x64, w6464, ksize=3, algo=1))
    r['32_32_TF'] = 35.39e3 / r['32_32_us']; r['32_64_TF'] = 70.78e3 / r['32_64_us']; r['64_64_TF'] = 141.56e3 / r['64_64_us']
    print(json.dumps(r))
print('done')
