#!/usr/bin/env python
"""Headline benchmark: samples/s of PreWorld's camera->voxel occupancy hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2] [--no-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one sample through the whole hot path with every input already resident in HBM:
  C3 (default, BASELINE.json configs[2], the config north_star's target is quoted on):
     2 frames (key + adjacent) x [LSS geometry -> voxel sort -> voxel pooling -> pre_process],
     concat -> CustomResNet3D [1,2,4] -> LSSFPN3D -> final_conv -> 6-step state-conditioned
     forecast -> OccHead x 7 states -> 7 uint8 200x200x16 occupancy grids.
  C2 (configs[1]): key frame only (with_prev=False), 1 state.
Inputs are what LSSViewTransformerBEVDepth.forward hands to view_transform
(mmdet3d/models/necks/view_transformer.py:798-803): softmaxed depth (6,88,32,88) and context
features (6,32,32,88) per frame plus the camera tensors; the image backbone / DepthNet stay on
PyTorch and are outside the measured path.  Weights are random (seeded), data
synthetic -- there is no dataset or checkpoint in this environment.

N > 1: one process per GPU, each rank runs its own sample stream (samples are independent:
no data-path collective), barrier + synchronize on both sides, max over ranks -> weak scaling.

The JSON line also carries
  roofline     -- the dominant kernel (MFMA fp32 conv3d), FLOP/launch / HIP-event duration
  cpu_baseline -- the CPU oracle (a port: the reference has no CPU path) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from preworld_amd import modules as M  # noqa: E402
from preworld_amd import ops  # noqa: E402
from preworld_amd import synth as S  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_HBM_GBPS = 8000.0


def build_net(dev, with_prev=True):
    net = M.PreWorld4DTraj(
        img_view_transformer=dict(type='LSSViewTransformerBEVStereo', grid_config=S.GRID_CONFIG_FULL,
                                  input_size=S.INPUT_SIZE, in_channels=512, out_channels=32,
                                  sid=False, collapse_z=False, loss_depth_weight=0.05,
                                  depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=96, stereo=True,
                                                    bias=5.0), downsample=16),
        img_bev_encoder_backbone=dict(type='CustomResNet3D', numC_input=64, num_layer=[1, 2, 4],
                                      with_cp=False, num_channels=[32, 64, 128], stride=[1, 2, 2],
                                      backbone_output_ids=[0, 1, 2]),
        img_bev_encoder_neck=dict(type='LSSFPN3D', in_channels=224, out_channels=32),
        pre_process=dict(type='CustomResNet3D', numC_input=32, with_cp=False, num_layer=[1],
                         num_channels=[32], stride=[1], backbone_output_ids=[0]),
        occupancy_head=dict(type='OccHead', with_cp=False, use_deblock=False,
                            norm_cfg=dict(type='SyncBN', requires_grad=True), soft_weights=True,
                            final_occ_size=[200, 200, 16], empty_idx=17, num_level=1,
                            in_channels=[32], out_channel=18,
                            point_cloud_range=[-40, -40, -1, 40, 40, 5.4]),
        if_post_finetune=True, with_prev=with_prev)
    sd = S.synth_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return net.to(dev).eval(), sd


def make_inputs(dev, seed, n_frames):
    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    frames = []
    for f in range(n_frames):
        rig = S.synthetic_rig(6, dx=-2.5 * f)              # adjacent frame: ego moved 2.5 m
        depth, feat = S.lift_inputs(seed * 16 + f)
        frames.append(dict(depth=T(depth).view(6, 88, 32, 88), tran_feat=T(feat).view(6, 32, 32, 88),
                           sensor2keyego=T(rig['sensor2ego']), intrin=T(rig['intrin']),
                           post_rot=T(rig['post_rot']), post_tran=T(rig['post_tran']),
                           bda=T(rig['bda'])))
    ego = T(S.ego_state(seed))
    return frames, ego


# ------------------------------------------------------------------------------ roofline probe
class KernelProbe:
    """Times every conv-family launch of one eager step with HIP events recorded on the launch
    stream (torch's current stream == the stream passed through the C ABI)."""

    def __init__(self):
        self.records = []
        self._orig = {}

    def __enter__(self):
        for name in ('conv3d_ndhwc', 'conv3d_wino', 'occ_head_fused', 'forecast_steps', 'fpn3d_fuse',
                     'bev_pool_dense', 'segment_sort'):
            self._orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self._orig[name]))
        return self

    def __exit__(self, *a):
        for name, fn in self._orig.items():
            setattr(ops, name, fn)

    def _wrap(self, name, fn):
        def inner(*args, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*args, **kw)
            e.record()
            self.records.append((name, self._work(name, args, kw), s, e))
            return out
        return inner

    @staticmethod
    def _work(name, args, kw):
        """(kernel variant label, algorithmic flops, algorithmic HBM bytes) of one launch."""
        if name == 'conv3d_ndhwc':
            x, wpk = args[0], args[1]
            B, D, H, W, Cin = x.shape
            taps, nt = wpk.shape[1], wpk.shape[2]
            ks = kw.get('ksize', 3)
            st = kw.get('stride', 1)
            nv = B * (D // st) * (H // st) * (W // st)
            cout = (kw.get('cout0') or nt * 32) + (kw.get('cout1') or 0)
            tiled = ks == 3 and st == 1 and kw.get('algo', 0) != 2
            # one label per (kernel family, grid, channels): the library picks the tile/N-group variant
            # per grid size, and small grids fill the 256 CUs less well than the full-resolution ones
            label = '%s %dx%dx%d %d->%d' % ('conv3d_k3s1_mfma' if tiled else 'conv3d_gather_mfma<k%d,s%d>' % (ks, st),
                                           D, H, W, Cin, cout)
            byts = 4.0 * (x.numel() + nv * cout + wpk.numel())
            return label, 2.0 * nv * taps * Cin * cout, byts
        if name == 'conv3d_wino':
            # algorithmic work = the direct-form conv; the kernel itself executes 64 transform-
            # domain products per 2x2x2 outputs instead of 8*27: x8/27 of these flops on the matrix pipe
            x, uw = args[0], args[1]
            B, D, H, W, Cin = x.shape
            cout = (kw.get('cout0') or uw.shape[2] * 16) + (kw.get('cout1') or 0)
            nv = B * D * H * W
            label = 'conv3d_wino_mfma %dx%dx%d %d->%d' % (D, H, W, Cin, cout)
            return label, 2.0 * nv * 27 * Cin * cout, 4.0 * (x.numel() + nv * cout + uw.numel())
        if name == 'occ_head_fused':
            x = args[0]
            nv = x.numel() // x.shape[-1]
            # Winograd-domain weights are 5-d (pack_conv_weight_wino): same algorithmic work, 8/27 of the conv's
            # multiplies executed
            label = 'conv3d_wino_mfma<occ_head>' if args[1].dim() == 5 else 'conv3d_k3s1_mfma<occ_head>'
            # the six forecast states are decoded by ONE launch over a batch of 6: counted as 6 units of one state
            # each, so that per-launch figures (duration, algorithmic bytes, PMC traffic) stay per 200x200x16 state
            return label, 2.0 * nv * (27 * 32 * 16 + 16 * 8 + 8 * 18), 4.0 * x.numel() + nv, int(x.shape[0])
        if name == 'forecast_steps':
            v0, n_steps = args[0], args[6]
            nv = v0.numel() // 32
            return 'forecast_mfma', 2.0 * nv * n_steps * (32 * 128 + 128 * 32), \
                4.0 * v0.numel() * (1 + n_steps)
        if name == 'fpn3d_fuse':
            x = args[0]
            return 'fpn3d_fuse', 2.0 * (x.numel() // 32) * 32 * 32, 8.0 * x.numel()
        if name == 'bev_pool_dense':
            depth, feat, vs = args[0], args[1], args[2]
            return 'bev_pool_dense', 2.0 * 32 * vs.order.numel() * 0.6, \
                4.0 * (vs.n_keys * 32 + depth.numel() + feat.numel() + vs.n_keys + 2 * 879748)
        return name, 0.0, 0.0

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, work, s, e in self.records:
            label, fl, by = work[:3]
            a = agg.setdefault(label, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a['launches'] += work[3] if len(work) > 3 else 1
            a['ms'] += s.elapsed_time(e)
            a['flops'] += fl
            a['bytes'] += by
        return agg


# ------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(sd, budget_s=25.0):
    """The CPU oracle (oracle/ -- a port: the reference ships no CPU implementation of these ops,
    SURVEY.md 0) timed on the host cores for the C3 pipeline on a BOUNDED sample: the C1-sized
    grid (100x100x8 = 1/8 of the voxels, 1 camera), scaled to full-size samples/s by the voxel
    ratio (the conv/MLP stack, >99% of the CPU time, is linear in the voxel count)."""
    from oracle import oracle as O
    threads = O.num_threads()
    gc = S.GRID_CONFIG_C1
    scale = (200 * 200 * 16) / (100 * 100 * 8)
    times = []
    t_all = time.time()
    # repeat with fresh seeds until ~10 s of CPU work are in the sample (at most 6 runs, at least 2)
    while len(times) < 2 or (time.time() - t_all < 10.0 and len(times) < 6):
        seed = 100 + 16 * len(times)
        t0 = time.time()
        bevs = []
        for f in range(2):
            depth, feat = S.lift_inputs(seed + f, N=1)
            r = S.synthetic_rig(1, dx=-2.5 * f)
            bev = O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'],
                                       r['post_tran'], r['bda'], gc, S.INPUT_SIZE, S.DOWNSAMPLE)
            bevs.append(O.pre_process(bev, sd))
        x = O.encoder_forward(bevs[1], bevs[0], sd)
        vf = O.final_conv(x, sd)
        states, _ = O.preworld4d_decode(vf, S.ego_state(len(times)), sd, n_steps=6, post_finetune=True)
        times.append(time.time() - t0)
        assert len(states) == 7 and states[0].shape == (100, 100, 8)
    dt = float(np.mean(times))
    return dict(value=1.0 / (dt * scale), unit='samples/s', cores=threads, kind='port',
                sample='%d runs of the C3 pipeline on the C1-sized grid (1 cam, 100x100x8 = 1/8 of the voxels): '
                       'mean %.2f s (min %.2f, max %.2f; %.1f s of CPU work) on %d OpenMP threads, scaled x%d by voxel '
                       'count; the reference has no CPU path for these ops'
                       % (len(times), dt, min(times), max(times), sum(times), threads, int(scale)))


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--settle-s', type=float, default=1.0,
                    help='untimed run-in before the W warm-up steps: the GPU needs a few hundred ms of load to leave '
                         'its idle clock state (measured: 304 us vs 278 us for the same conv launch)')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='independent samples in flight per GPU (one hipGraph + HIP stream each); 1 = strictly serial')
    ap.add_argument('--config', default='C3', choices=['C3', 'C2'])
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
    dist = None
    n_dev = torch.cuda.device_count()
    # one rank per GPU over RCCL.  Fewer GPUs than ranks only happens when the launch contract is
    # exercised on a 1-GPU development box: ranks then share a device and rendezvous over gloo
    # (RCCL refuses two ranks on one GPU); the timing of such a run means nothing.
    oversubscribed = world > n_dev
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = 'cuda:%d' % dev_index
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo' if oversubscribed else 'nccl',       # 'nccl' == RCCL on ROCm
                                rank=rank, world_size=world)

    n_frames, n_steps_fc = (2, 6) if args.config == 'C3' else (1, 0)
    net, sd = build_net(dev, with_prev=args.config == 'C3')
    frames, ego = make_inputs(dev, seed=rank, n_frames=n_frames)

    def step():
        return net.simple_test_from_lift(frames, ego, n_steps=n_steps_fc)

    # eager warmup (also fills the packed-weight caches and sets kernel attributes)
    for _ in range(max(1, args.warmup)):
        out = step()
    torch.cuda.synchronize()

    # run-in: clocks settled before anything is measured
    t_settle = time.perf_counter() + args.settle_s
    while time.perf_counter() < t_settle:
        step()
        torch.cuda.synchronize()

    # live per-kernel timing for the roofline object (eager, HIP events on the launch stream)
    roofline = None
    if rank == 0:
        with KernelProbe() as probe:
            for _ in range(3):
                step()
        agg = probe.summary()
        dom = max(agg.items(), key=lambda kv: kv[1]['ms'])
        label, a = dom
        tf = a['flops'] / (a['ms'] * 1e-3) / 1e12
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this
        # process, so the figure comes from the committed rocprofv3 --pmc passes (same kernel, same
        # shape; profiles/r01_pmc_hbm_traffic.md says how it was collected and corrected)
        traffic = None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
        if os.path.exists(pmc):
            ent = json.load(open(pmc)).get(label)
            if ent:
                traffic = int(ent['fetch_bytes'] + ent['write_bytes'])
        roofline = dict(bound='mfma', kernel=label, achieved=round(tf, 2), peak=PEAK_FP32_MFMA_TFLOPS,
                        unit='TFLOP/s', frac=round(tf / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic,
                        traffic_unit='bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)',
                        algorithmic_bytes=int(a['bytes'] / a['launches']),
                        avg_launch_us=round(a['ms'] * 1e3 / a['launches'], 2),
                        launches_per_step=a['launches'] // 3,
                        **(dict(executed_tflops=round(tf * 8 / 27, 2), executed_frac=round(tf * 8 / 27 / PEAK_FP32_MFMA_TFLOPS, 4),
                                note='Winograd F(2x2x2,3x3x3): achieved = direct-form (algorithmic) FLOPs / time, which can '
                                     'exceed the matrix-pipe peak; executed_* counts the 8/27 of them the MFMAs really do'
                                     + ('; a launch here = one 200x200x16 state (states 1-6 share one kernel launch over a '
                                        'batch of 6)' if 'occ_head' in label else ''))
                           if label.startswith('conv3d_wino') else {}),
                        all_kernels={k: dict(us_per_step=round(v['ms'] * 1e3 / 3, 1),
                                             launches=v['launches'] // 3,
                                             tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2),
                                             alg_GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1))
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])})

    graph = None
    latency_ms = None
    if not args.no_graph:
        from preworld_amd.pipeline import CapturedSample
        # M independent samples in flight, each a hipGraph over its own static buffers on its own HIP
        # stream: while one sample sits in a stage that cannot fill 256 CUs (the 8x100x100 / 4x50x50
        # encoder levels, the last partial wave of tiles of every launch, the latency-bound sort), the other
        # one's kernels take the idle CUs.  A step is still one sample; steps alternate between the streams.
        M = max(1, args.in_flight)
        caps = [CapturedSample(net, *make_inputs(dev, seed=rank * 8 + k, n_frames=n_frames), n_steps=n_steps_fc)
                for k in range(M)]
        streams = [torch.cuda.Stream() for _ in range(M)]
        graph = caps[0]
        out = graph.out

        def run_steps(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % M]):
                    caps[i % M].replay()
    else:
        M = 1

        def run_steps(n):
            for _ in range(n):
                step()

    t_settle = time.perf_counter() + args.settle_s
    while time.perf_counter() < t_settle:
        run_steps(M)
        torch.cuda.synchronize()
    if graph is not None and M > 1 and rank == 0:        # single-sample latency, reported next to the throughput
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            caps[0].replay()
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - t0) / 20 * 1e3
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device='cpu' if oversubscribed else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on the produced states (cheap, outside the timed region)
    occ0 = out['semantic_occ_0s'][0]
    assert occ0.shape == (200, 200, 16) and occ0.dtype == torch.uint8
    n_states = sum(1 for k in out if k.startswith('semantic_occ_'))

    if rank == 0:
        samples = args.steps * world                 # one sample per step per rank
        res = {
            'metric': 'samples/sec (6-cam frame -> 200x200x16 occ)',
            'value': round(samples / elapsed, 3),
            'unit': 'samples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': ('C3: 7-state temporal, 6 cams, key+adjacent frame, 200x200x16, '
                             'LSS pooling x2 + pre_process x2 + CustomResNet3D + LSSFPN3D + final_conv '
                             '+ 6-step forecast + OccHead x7 -> 7 uint8 occupancy grids')
                if args.config == 'C3' else
                'C2: single frame (with_prev=False), 6 cams, 200x200x16, 1 state',
                'states_per_sample': n_states,
                'launch': ('hipGraph replay, %d independent sample(s) in flight on %d HIP stream(s)' % (M, M))
                if graph is not None else 'eager',
                'single_sample_latency_ms': round(latency_ms, 4) if latency_ms else None,
                'parallelism': 'replicas x%d (independent samples, no data-path collective)%s' % (
                    world, ' -- OVERSUBSCRIBED development run, %d GPU(s): not a measurement' % n_dev if oversubscribed else ''),
                'excluded': 'image backbone + DepthNet (stay on PyTorch, SURVEY 8a)',
            },
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(sd)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
