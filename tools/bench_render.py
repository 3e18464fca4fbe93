"""C5 (BASELINE.json configs[4]) render head timing: forward and forward + backward, literal shape (3072 rays x 96 uniform
samples) and the reference shape (38 400 rays x 417 samples); fp32 grid.  Prints points/s and achieved HBM GB/s on the
algorithmic bytes of SURVEY 8d (grids 53.8 MB read once + rays / outputs; backward adds the 61 MB gradient grid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from preworld_amd import modules as M, ops, synth as S
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
head = M.NerfHead(point_cloud_range=[-40, -40, -1, 40, 40, 5.4], voxel_size=0.4, scene_center=[0, 0, 2.2], radius=39).to(DEV)
density, semantic, color = S.render_grids(41)
grid = M.pack_attribute_grid(T(density), T(semantic), T(color))
consts = head.consts(torch.eye(3))


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


# a scene-like grid beside the all-samples-survive random one: empty space (sigma = -8) with occupied voxels (sigma = +4) on a
# ground slab and a ring of "buildings" 15-25 m out, so that rays terminate after a few occupied samples
xs, ys, zs = np.meshgrid(np.arange(200), np.arange(200), np.arange(16), indexing='ij')
rr = np.hypot(xs - 100, ys - 100)
dens_scene = np.where((zs < 2) | ((rr > 40) & (rr < 60) & ((xs // 8 + ys // 8) % 2 == 0)), 4.0, -8.0).astype(np.float32)
grid_scene = M.pack_attribute_grid(T(dens_scene), T(semantic), T(color))

for name, R, t, grid in (('C5 literal 3072 x 96, random grid', 3072, None, grid), ('reference 38400 x 417, random grid (every sample kept)', 38400, head.t_table(DEV), grid),
                         ('C5 literal 3072 x 96, scene-like grid', 3072, None, grid_scene),
                         ('reference 38400 x 417, scene-like grid', 38400, head.t_table(DEV), grid_scene)):
    if t is None:
        b = torch.linspace(0, 2, 97)
        t = ((b[1:] + b[:-1]) * 0.5).to(DEV).contiguous()
    o, d = S.rays(7, R)
    ro, rd = T(o), T(d)
    S_ = t.numel()
    gd, gs, gc, gl = torch.randn(R, device=DEV), torch.randn(R, 17, device=DEV), torch.randn(R, 3, device=DEV), torch.randn(R, device=DEV)
    gg = torch.zeros_like(grid)
    fwd = timeit(lambda: ops.render_rays(ro, rd, t, grid, consts))
    g16 = grid.to(torch.bfloat16)
    fwd16 = timeit(lambda: ops.render_rays(ro, rd, t, g16, consts))
    bwd = timeit(lambda: ops.render_rays_backward(ro, rd, t, grid, consts, gd, gs, gc, gl, grad_grid=gg, algo='sorted'), n=10)
    bwd_at = timeit(lambda: ops.render_rays_backward(ro, rd, t, grid, consts, gd, gs, gc, gl, grad_grid=gg, algo='atomics'), n=5)
    zero = timeit(lambda: gg.zero_())
    pts = R * S_
    kept = int(ops.render_rays(ro, rd, t, grid, consts, want_debug=True)['counts'][:, 2].sum())
    print('[%d kept samples] ' % kept, end='')
    print('%s: forward %.1f us (%.2f G samples/s, %.0f GB/s on 57.8 MB) | backward (sorted, deterministic) %.1f us = %.1f x forward; '
          'backward (float atomics) %.1f us | + %.1f us zero-fill of the 61 MB gradient grid | fwd+bwd %.1f us (%.2f G samples/s) | '
          'forward with the grid stored as bf16: %.1f us' % (
              name, fwd, pts / fwd * 1e-3, 57.8e6 / fwd * 1e-3, bwd, bwd / fwd, bwd_at, zero, fwd + bwd + zero,
              pts / (fwd + bwd + zero) * 1e-3, fwd16), flush=True)
