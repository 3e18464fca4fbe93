"""Stride-2 3x3x3 layers of the C3 encoder on the split-fp16 gather kernel (conv1 + downsample fused: N = 2 x Cout); A/B of env knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from preworld_amd import ops, _lib
from bench_h2 import timeit
DEV = 'cuda:0'
for (D, H, W), cin, cout in (((16, 200, 200), 32, 64), ((8, 100, 100), 64, 128)):
    x = ops.f32_to_h2(torch.randn(1, D, H, W, cin, device=DEV))
    w1 = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    w2 = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    wpk, inv = ops.pack_conv_weights_h2_concat([w1, w2])
    t = timeit(lambda: ops.conv3d_h2(x, wpk, inv, cout0=cout, cout1=cout, relu0=True, ksize=3, stride=2))
    gf = 2.0 * (D // 2) * (H // 2) * (W // 2) * 27 * cin * 2 * cout / 1e9
    print('%dx%dx%d %d->2x%d s2  %.1f GF  %s  %.1f us (%.0f TF direct)' % (D, H, W, cin, cout, gf, _lib.lib().pw_last_kernel().decode(), t, gf / t * 1e3), flush=True)
    probe = torch.zeros(64, dtype=torch.int64, device=DEV)
    os.environ['PW_CONV_PROBE'] = str(probe.data_ptr())
    ops.conv3d_h2(x, wpk, inv, cout0=cout, cout1=cout, relu0=True, ksize=3, stride=2)
    torch.cuda.synchronize()
    del os.environ['PW_CONV_PROBE']
    pr = probe.view(8, 8)[:4, :5]
    d = (pr[:, 1:] - pr[:, :-1]).float().mean(0).tolist()
    print('   block 300 ticks: stage0 %.0f  taps0 %.0f  rest of passes %.0f  epilogue %.0f' % tuple(d), flush=True)
