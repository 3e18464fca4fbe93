"""The small-grid layers of the encoder (8x100x100 64->64, 4x50x50 128->128) on pw_conv3d_h2 with each kernel choice (algo 0 auto =
0 auto; 6 the persistent LDS-tiled kernel with one wave per SIMD, 5 its paired-wave form, 2 gather, 3 gather with the input-channel chunks
split over the 4 waves): time per launch.
Development aid for VERDICT r03 weak 4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench_h2 import timeit  # noqa: E402
from preworld_amd import _lib, ops  # noqa: E402

DEV = 'cuda:0'
for (B, D, H, W), cin, cout in (((1, 8, 100, 100), 64, 64), ((1, 4, 50, 50), 128, 128), ((1, 16, 200, 200), 64, 64)):
    torch.manual_seed(0)
    x = torch.randn(B, D, H, W, cin, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
    sc, bi = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    xh = ops.f32_to_h2(x)
    wpk, inv = ops.pack_conv_weight_h2(w)
    y = torch.empty(B, D, H, W, cout, device=DEV)
    gf = 2.0 * B * D * H * W * 27 * cin * cout / 1e9
    ref = None
    for algo in [int(v) for v in os.environ.get('ALGOS', '6,5,2,3').split(',')]:
        out = ops.conv3d_h2(xh, wpk, sc * inv, bi, relu0=True, out0=y, out_h2=(True, True), algo=algo)
        t = timeit(lambda: ops.conv3d_h2(xh, wpk, sc * inv, bi, relu0=True, out0=y, out_h2=(True, True), algo=algo))
        kern = _lib.lib().pw_last_kernel().decode()
        f = ops.h2_to_f32(out)
        same = '' if ref is None else '  max|diff| vs algo 6 %.2e' % float((f - ref).abs().max())
        ref = f if ref is None else ref
        print('%dx%dx%dx%d %d->%d %5.1f GF  algo %d  %-46s %.1f us (%.0f TF direct)%s' % (B, D, H, W, cin, cout, gf, algo, kern, t, gf / t * 1e3, same),
              flush=True)
