"""Sustained timing of the split-fp16 conv kernel (pw_conv3d_h2) on the C3 layer shapes next to the Winograd fp32 kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from preworld_amd import ops, _lib

DEV = 'cuda:0'
SHAPES = [((1, 16, 200, 200), 32, 32), ((1, 16, 200, 200), 32, 64), ((1, 16, 200, 200), 64, 64),
          ((1, 8, 100, 100), 64, 64), ((1, 4, 50, 50), 128, 128), ((2, 16, 200, 200), 32, 32), ((2, 4, 50, 50), 128, 128)]


def timeit(fn, secs=0.4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
  shapes = [SHAPES[int(os.environ['SHAPE'])]] if os.environ.get('SHAPE') else SHAPES
  for (B, D, H, W), cin, cout in shapes:
      torch.manual_seed(0)
      x = torch.randn(B, D, H, W, cin, device=DEV)
      w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
      sc = torch.ones(cout, device=DEV); bi = torch.zeros(cout, device=DEV)
      xh = ops.f32_to_h2(x)
      wpk, inv = ops.pack_conv_weight_h2(w)
      y = torch.empty(B, D, H, W, cout, device=DEV)
      uw = ops.pack_conv_weight_wino(w)
      gf = 2.0 * B * D * H * W * 27 * cin * cout / 1e9
      res = {}
      for fmt in (True, False):
          t = timeit(lambda: ops.conv3d_h2(xh, wpk, sc * inv, bi, relu0=True, out0=y, out_h2=(fmt, fmt)))
          res['h2->h2' if fmt else 'h2->f32'] = t
      kern = _lib.lib().pw_last_kernel().decode()
      if os.environ.get('QUICK'):          # pipelined kernel only (variant A/B runs)
          print('%dx%dx%dx%d %d->%d %6.1f GF  %s' % (B, D, H, W, cin, cout, gf, kern),
                '  '.join('%s %.1f us (%.0f TF direct)' % (k, v, gf / v * 1e3) for k, v in res.items()), flush=True)
          continue
      t = timeit(lambda: ops.conv3d_wino(x, uw, sc, bi, relu0=True, out0=y))
      res['wino f32'] = t
      print('%dx%dx%dx%d %d->%d %6.1f GF  %s' % (B, D, H, W, cin, cout, gf, kern),
            '  '.join('%s %.1f us (%.0f TF direct)' % (k, v, gf / v * 1e3) for k, v in res.items()), flush=True)


if __name__ == '__main__':
    main()
