"""a few sorted-backward calls of the render head for rocprofv3 --kernel-trace (development aid): R rays from argv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from preworld_amd import modules as M, ops, synth as S
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 38400
scene = len(sys.argv) > 2 and sys.argv[2] == 'scene'
head = M.NerfHead(point_cloud_range=[-40, -40, -1, 40, 40, 5.4], voxel_size=0.4, scene_center=[0, 0, 2.2], radius=39).to(DEV)
density, semantic, color = S.render_grids(41)
if scene:
    xs, ys, zs = np.meshgrid(np.arange(200), np.arange(200), np.arange(16), indexing='ij')
    rr = np.hypot(xs - 100, ys - 100)
    density = np.where((zs < 2) | ((rr > 40) & (rr < 60) & ((xs // 8 + ys // 8) % 2 == 0)), 4.0, -8.0).astype(np.float32)
grid = M.pack_attribute_grid(T(density), T(semantic), T(color))
consts = head.consts(torch.eye(3))
t = head.t_table(DEV)
o, d = S.rays(7, R)
ro, rd = T(o), T(d)
gd, gs, gc, gl = torch.randn(R, device=DEV), torch.randn(R, 17, device=DEV), torch.randn(R, 3, device=DEV), torch.randn(R, device=DEV)
gg = torch.zeros_like(grid)
for _ in range(4):
    ops.render_rays_backward(ro, rd, t, grid, consts, gd, gs, gc, gl, grad_grid=gg, algo='sorted')
torch.cuda.synchronize()
