"""Can the voxel-side training step run under ONE hipGraph?  pre_process -> encoder -> neck -> final_conv -> OccHead -> loss_voxel -> backward,
from the pooled BEV grids (the lift and its backward keep their host syncs outside).  Prints eager vs replay time and checks the gradients."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from preworld_amd import harness, synth as S  # noqa: E402
from preworld_amd.modules import to_channels_last_3d  # noqa: E402

dev = 'cuda:0'
cfg = harness.model_cfg(S.GRID_CONFIG_FULL, detector='PreWorld')
cfg.update(if_render=False, if_post_finetune=True, use_lss_depth_loss=False, weight_voxel_ce=1.0, weight_voxel_sem_scal=1.0,
           weight_voxel_geo_scal=1.0, weight_voxel_lovasz=1.0)
net = harness.build_model(cfg, S.synth_state_dict(0), dev).train()
frames = harness.lifted_frames(0, 6, dev, n_frames=2)
sem = torch.randint(0, 18, (1, 200, 200, 16), device=dev)
vt = net.img_view_transformer


def pooled(fr):
    d, f = fr['depth'], fr['tran_feat']
    B, N = fr['sensor2keyego'].shape[:2]
    inp = [d.new_empty(B, N, 1, d.shape[-2], d.shape[-1]), fr['sensor2keyego'], None, fr['intrin'], fr['post_rot'], fr['post_tran'], fr['bda']]
    with torch.no_grad():
        return to_channels_last_3d(vt.view_transform(inp, d, f)[0]).float().contiguous()


key_in = pooled(frames[0]).requires_grad_(True)
adj_in = pooled(frames[1])


def region():
    net.zero_grad(set_to_none=True)
    key_in.grad = None
    key = net.pre_process_net.forward_cl(key_in)[0]
    with torch.no_grad():
        adj = net.pre_process_net.forward_cl(adj_in)[0]
    losses = net.forward_train_from_feats(net.bev_encoder_cl(torch.cat([adj, key], -1)), voxel_semantics=sem)
    total = sum(losses.values())
    total.backward()
    return total


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('eager region: %.2f ms' % timeit(region))
ref_total = float(region().detach())
ref_grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
ref_key = key_in.grad.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        region()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    total = region()
torch.cuda.synchronize()
print('captured; replay: %.2f ms' % timeit(g.replay))
g.replay()
torch.cuda.synchronize()
print('loss eager %.6f  replay %.6f' % (ref_total, float(total)))
worst = max(float((p.grad - ref_grads[n]).abs().max() / (ref_grads[n].abs().max() + 1e-30)) for n, p in net.named_parameters() if n in ref_grads)
print('largest relative gradient difference replay vs eager: %.2e; d/d key_in %.2e' % (worst, float((key_in.grad - ref_key).abs().max() / ref_key.abs().max())))
