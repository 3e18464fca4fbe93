"""Training step of the voxel encoder at the C3 shape: CustomResNet3D [1,2,4] on (1,64,16,200,200), forward with batch-statistics
BatchNorm + backward (conv dgrad / wgrad, BN backward) through preworld_amd.train.  Development aid / profiles/r02_train.txt."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import _lib, modules as M, synth as S, train  # noqa: E402

dev = 'cuda:0'
sd = S.synth_state_dict(0)
enc = M.CustomResNet3D(numC_input=64, num_layer=[1, 2, 4], with_cp=False, num_channels=[32, 64, 128], stride=[1, 2, 2],
                       backbone_output_ids=[0, 1, 2])
own = enc.state_dict()
with torch.no_grad():
    for k in own:
        if 'num_batches_tracked' not in k:
            own[k].copy_(torch.from_numpy(sd['img_bev_encoder_backbone.' + k]))
enc = enc.to(dev).train()
x = torch.randn(1, 16, 200, 200, 64, device=dev).requires_grad_(True)


def step():
    for p in enc.parameters():
        p.grad = None
    x.grad = None
    feats = enc.forward_cl(x)
    loss = sum(f.sum() for f in feats)
    loss.backward()


def fwd():
    with torch.no_grad():
        enc.forward_cl(x)


ONLY_STEP = os.environ.get('ONLY_STEP', '0') == '1'       # rocprofv3 runs: just the whole training step
for fn, name in (() if ONLY_STEP else ((fwd, 'forward (train-mode BN)'), (step, 'forward + backward'))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-28s %.2f ms' % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)

# single layers
for (D, H, W), cin, cout, s in (() if ONLY_STEP else (((16, 200, 200), 32, 32, 1), ((16, 200, 200), 64, 32, 1), ((16, 200, 200), 32, 64, 2),
                                 ((8, 100, 100), 64, 64, 1), ((8, 100, 100), 64, 128, 2), ((4, 50, 50), 128, 128, 1))):
    xx = torch.randn(1, D, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    y = train.conv3d_raw(xx, w, s)
    g = torch.randn_like(y)
    gf = 2.0 * y.numel() // cout * 27 * cin * cout / 1e9

    def t(fn):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 5 * 1e3
    tw = t(lambda: train.conv3d_wgrad(xx, g, w.shape, s))
    td = t(lambda: train.conv3d_dgrad(g, w, xx.shape, s))
    print('%dx%dx%d %d->%d s%d  %.1f GF: wgrad %.0f us (%.0f TF)  dgrad %.0f us (%.0f TF)' % (D, H, W, cin, cout, s, gf, tw, gf / tw * 1e3,
                                                                                    td, gf / td * 1e3), flush=True)

# ---- the whole voxel side of PreWorld.forward_train at the C3 shape, from lifted inputs (image side excluded): pooling with
# backward -> pre_process (key frame under autograd, adjacent under no_grad) -> encoder -> neck -> final_conv -> OccHead -> loss_voxel
from preworld_amd import harness  # noqa: E402
from preworld_amd.modules import to_channels_last_3d  # noqa: E402
cfg = harness.model_cfg(S.GRID_CONFIG_FULL, detector='PreWorld')
cfg.update(if_render=False, if_post_finetune=True, use_lss_depth_loss=False, weight_voxel_ce=1.0, weight_voxel_sem_scal=1.0,
           weight_voxel_geo_scal=1.0, weight_voxel_lovasz=1.0)
net = harness.build_model(cfg, sd, dev).train()
frames = harness.lifted_frames(0, 6, dev, n_frames=2)
sem = torch.randint(0, 18, (1, 200, 200, 16), device=dev)
vt = net.img_view_transformer


def lift(fr, grad):
    d, f = fr['depth'].detach().requires_grad_(grad), fr['tran_feat'].detach().requires_grad_(grad)
    B, N = fr['sensor2keyego'].shape[:2]
    inp = [d.new_empty(B, N, 1, d.shape[-2], d.shape[-1]), fr['sensor2keyego'], None, fr['intrin'], fr['post_rot'], fr['post_tran'],
           fr['bda']]
    x = to_channels_last_3d(vt.view_transform(inp, d, f)[0]).float()
    return net.pre_process_net.forward_cl(x)[0], d


def train_step():
    net.zero_grad(set_to_none=True)
    key, d = lift(frames[0], True)
    with torch.no_grad():
        adj, _ = lift(frames[1], False)
    feat = net.bev_encoder_cl(torch.cat([adj, key], -1))
    losses = net.forward_train_from_feats(feat, voxel_semantics=sem)
    sum(losses.values()).backward()
    return losses


for _ in range(2):
    out = train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get('N_STEPS', '5'))
for _ in range(n):
    train_step()
torch.cuda.synchronize()
print('PreWorld.forward_train voxel side, C3 shape (6 cams, 2 frames, 200x200x16), forward + backward: %.1f ms per step   losses: %s' % (
    (time.perf_counter() - t0) / n * 1e3, ', '.join('%s %.3f' % (k, float(v.detach())) for k, v in out.items() if 'sup' not in k)), flush=True)

if os.environ.get('PW_TORCH_PROFILE'):           # which torch ops (glue around the HIP kernels) the step spends device time in
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            train_step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=28, max_name_column_width=60), flush=True)
