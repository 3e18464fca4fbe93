"""Sustained timing of the fused OccHead kernel (1 s loop); honours PW_CONV_DMA_STAGE / PW_OCC_PIPE;
WINO=1 times the Winograd kernel (k_occ_head_wino) instead of the direct one."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops  # noqa: E402
dev = 'cuda:0'
x = torch.randn(1, 16, 200, 200, 32, device=dev)
sc, bi = torch.ones(32, device=dev), torch.zeros(32, device=dev)
w1 = torch.randn(8, 16, device=dev); s1 = torch.ones(8, device=dev); b1 = torch.zeros(8, device=dev)
w2 = torch.randn(18, 8, device=dev)
w0 = torch.randn(16, 32, 3, 3, 3, device=dev) * 0.05
wp16 = ops.pack_conv_weight_wino(w0, cout_total=16) if os.environ.get('WINO') else ops.pack_conv_weight16(w0)
fn = lambda: ops.occ_head_fused(x, wp16, sc, bi, w1, s1, b1, w2, want_geo=True)
for _ in range(5): fn()
torch.cuda.synchronize()
n, t0 = 0, time.time()
while time.time() - t0 < float(os.environ.get('LOOP_S', 1.0)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = (time.time() - t0) / n
print('occ_head %.1f us  %.1f TFLOP/s useful  (DMA_STAGE=%s PIPE=%s WINO=%s)' % (dt * 1e6, 640000 * 2 * (27 * 32 * 16 + 16 * 8 + 8 * 18) / dt * 1e-12,
      os.environ.get('PW_CONV_DMA_STAGE', 'default'), os.environ.get('PW_OCC_PIPE', 'default'), os.environ.get('WINO', '0')))
