"""OccHead kernels at the C3 shapes: fp32 Winograd (k_occ_head_wino) vs split-fp16 (k_occ_head_h2); development aid."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from preworld_amd import ops, _lib
from bench_h2 import timeit

DEV = 'cuda:0'
rs = np.random.RandomState(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
w0 = T((rs.standard_normal((16, 32, 3, 3, 3)) * 0.05).astype(np.float32))
s0 = T(rs.uniform(0.5, 1.5, 16).astype(np.float32)); b0 = T((rs.standard_normal(16) * 0.3).astype(np.float32))
w1 = T((rs.standard_normal((8, 16)) * 0.4).astype(np.float32))
s1 = T(rs.uniform(0.5, 1.5, 8).astype(np.float32)); b1 = T((rs.standard_normal(8) * 0.3).astype(np.float32))
w2 = T((rs.standard_normal((18, 8)) * 0.5).astype(np.float32))
wpk, inv = ops.pack_occ_weight_h2(w0)
hargs = ((s0 * inv).contiguous(), b0) + ops.pack_occ_tail_h2(w1, s1, b1, w2) + (ops.occ_head_bounds(w0, s0, b0, w1, s1, b1),)
wino = ops.pack_conv_weight_wino(w0, cout_total=16)
fargs = (ops._pad32(s0, 1.0), ops._pad32(b0, 0.0), w1, s1, b1, w2)
for B in (1, 6):
    x = torch.randn(B, 16, 200, 200, 32, device=DEV)
    xh = ops.f32_to_h2(x)
    occ = torch.empty(B, 16, 200, 200, device=DEV, dtype=torch.uint8)
    gf = 2 * B * 640000 * (27 * 32 * 16 + 16 * 8 + 8 * 18) / 1e9
    tw = timeit(lambda: ops.occ_head_fused(x, wino, *fargs, occ=occ, want_geo=True))
    th = timeit(lambda: ops.occ_head_h2(xh, wpk, *hargs, occ=occ, want_geo=True))
    print('B=%d  %.1f GF   wino f32 %.1f us (%.0f TF)   h2 %.1f us (%.0f TF direct, %.0f executed)   %s' % (
        B, gf, tw, gf / tw * 1e-3, th, gf / th * 1e-3, 3 * gf / th * 1e-3, _lib.lib().pw_last_kernel().decode()), flush=True)
