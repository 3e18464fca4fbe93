import sys, time, torch
sys.path.insert(0, '/root/repo')
from preworld_amd import ops
torch.manual_seed(0)
dev = 'cuda:0'
for (B, D, H, W, ci, co) in [(1, 4, 8, 8, 32, 32), (1, 5, 9, 11, 32, 32), (2, 8, 16, 24, 64, 64), (1, 16, 200, 200, 32, 32)]:
    x = torch.randn(B, D, H, W, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
    sc = torch.rand(co, device=dev) + 0.5; bi = torch.randn(co, device=dev)
    ref = ops.conv3d_ndhwc(x, ops.pack_conv_weight(w), sc, bi, ksize=3, relu0=True)
    y = ops.conv3d_wino(x, ops.pack_conv_weight_wino(w), sc, bi, relu0=True)
    err = (y - ref).abs().max().item()
    print((B, D, H, W, ci, co), 'max abs diff %.3e  ref max %.2f' % (err, ref.abs().max().item()), flush=True)
x = torch.randn(1, 16, 200, 200, 32, device=dev)
for (ci, co) in [(32, 32), (32, 64), (64, 64)]:
    x = torch.randn(1, 16, 200, 200, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
    uw = ops.pack_conv_weight_wino(w); wp = ops.pack_conv_weight(w)
    for name, fn in (('wino', lambda: ops.conv3d_wino(x, uw)), ('direct', lambda: ops.conv3d_ndhwc(x, wp, ksize=3))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); n = 0; t0 = time.time()
        while time.time() - t0 < 1.0:
            for _ in range(20): fn()
            torch.cuda.synchronize(); n += 20
        print('%d->%d %s %.1f us' % (ci, co, name, (time.time() - t0) / n * 1e6), flush=True)
