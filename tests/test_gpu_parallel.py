"""The N > 1 latency mode (frames sharded for the lift, states sharded for the decode) with the
REAL HIP modules: two processes share cuda:0 and rendezvous over gloo on 127.0.0.1 (RCCL refuses two ranks on one GPU;
the 0.64 MB occupancy grids go through host memory for the all_gather).  Every rank must end up with exactly the
single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, full=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from preworld_amd import harness, synth as S
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = 'cuda:0'
        torch.cuda.set_device(0)
        net = harness.build_model(harness.model_cfg(S.GRID_CONFIG_FULL if full else S.GRID_CONFIG_C1), S.synth_state_dict(0), dev)
        frames = harness.lifted_frames(5, 6 if full else 1, dev)
        ego = torch.from_numpy(S.ego_state(5)).to(dev)
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=6)
        st = {}
        got = harness.simple_test_sharded(net, frames, ego, n_steps=6, gather_on_host=True, timings=st)
        same = [int((got['semantic_occ_%ds' % k][0].cpu() != want['semantic_occ_%ds' % k][0].cpu()).sum()) for k in range(7)]
        # round 6: the same mode as three captured phases around the two collectives (pipeline.ShardedSample), calibrated once; first on
        # the calibration sample, then on ANOTHER sample copied into the static buffers
        from preworld_amd.pipeline import ShardedSample
        ss = ShardedSample(net, frames, ego, n_steps=6, gather_on_host=True)
        got = ss.run()
        same_g = [int((got['semantic_occ_%ds' % k][0].cpu() != want['semantic_occ_%ds' % k][0].cpu()).sum()) for k in range(7)]
        frames2 = harness.lifted_frames(6, 6 if full else 1, dev)
        ego2 = torch.from_numpy(S.ego_state(6)).to(dev)
        with torch.no_grad():
            want2 = net.simple_test_from_lift(frames2, ego2, n_steps=6)
        st2 = {}
        got = ss.run(frames2, ego2, timings=st2)
        same_g += [int((got['semantic_occ_%ds' % k][0].cpu() != want2['semantic_occ_%ds' % k][0].cpu()).sum()) for k in range(7)]
        bad = ss.bad_replays()
        assert bad[0] == 0 and bad[1] == 2, bad
        assert st2['frames_bytes_received'] == st['frames_bytes_received'] and st2['states_bytes_received'] == st['states_bytes_received']
        same += same_g
        del ss
        # with_prev=False (C2's frame handling): the adjacent frame is dropped, its channel slice is zeros
        net.with_prev = False
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=2)
        got = harness.simple_test_sharded(net, frames, ego, n_steps=2, gather_on_host=True)
        same += [int((got['semantic_occ_%ds' % k][0].cpu() != want['semantic_occ_%ds' % k][0].cpu()).sum()) for k in range(3)]
        q.put((rank, same, {k: v for k, v in st.items() if 'bytes' in k}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('full', [False, True], ids=['C1-grid', 'C4-full-size-200x200x16'])
def test_sharded_lift_and_decode_two_ranks_equal_single_process(full):
    """full: BASELINE.json configs[3] at its real size on what one GPU allows -- 6 cameras, 200x200x16, the 81.92 MB fp32 frame
    exchange and its re-split under one exponent, all 7 states, with and without the adjacent frame (bevdet_occ.py:266-267)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, full)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    # voxels that differ from the single-process result, per state (7 eager with the adjacent frame, 2 x 7 through the captured
    # phases of pipeline.ShardedSample -- calibration sample, then a second sample -- and 3 without the adjacent frame).  The sharded pass
    # exchanges fp32 values and re-splits the concatenated buffer under ONE exponent, the single process keeps each frame under the
    # exponent of its own slot: values can differ below 2^-38 of a tensor's maximum, i.e. a handful of exact ties among the 640 000
    # voxels of a full-size state (the same allowance as a separately calibrated eager pass, tests/test_gpu_range.py); both ranks
    # must hold the SAME grids
    assert res[0][1] == res[1][1], res
    frame_bytes = (16 * 200 * 200 if full else 8 * 100 * 100) * 32 * 4
    for rank, diff, st in res:
        print('[sharded, 2 ranks, %s] differing voxels per state: %s; bytes %s' % ('200x200x16' if full else 'C1 grid', diff, st))
        assert max(diff) <= (8 if full else 0), diff
        assert st['frames_bytes_received'] == frame_bytes            # 2 frames on 2 ranks: the other rank's frame, once


def test_sharded_lift_and_decode_eight_ranks_equal_single_process():
    """BASELINE.json configs[3]'s world size, functionally: EIGHT gloo ranks sharing the one GPU, C1 grid, the real
    HIP modules.  Frames 0 / 1 are lifted on ranks 0 / 1 and broadcast (each rank receives the frames it did not lift: 2 x 10.24 MB,
    1 x on the owners -- not the 8 padded slots of round 4), states 0..6 decode on ranks 0..6, rank 7 owns no state and decodes
    nothing; every rank ends with the single-process grids."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, False)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    frame_bytes = 8 * 100 * 100 * 32 * 4
    for rank, diff, st in res:
        assert diff == res[0][1], (rank, diff, res[0][1])               # every rank holds the same grids
        assert max(diff) == 0, diff
        assert st['frames_bytes_received'] == (1 if rank < 2 else 2) * frame_bytes, (rank, st)
        assert st['frames_bytes_sent'] == (7 * frame_bytes if rank < 2 else 0), (rank, st)
        assert st['states_bytes_received'] == 7 * 100 * 100 * 8, (rank, st)
    print('[sharded, 8 ranks, C1 grid] all ranks equal the single-process result; bytes received per rank: %s'
          % [r[2]['frames_bytes_received'] for r in res])


def _rccl_worker(port, q, full=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from preworld_amd import harness, parallel, synth as S
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)            # 'nccl' == RCCL on ROCm
    try:
        dev = 'cuda:0'
        net = harness.build_model(harness.model_cfg(S.GRID_CONFIG_FULL if full else S.GRID_CONFIG_C1), S.synth_state_dict(0), dev)
        frames = harness.lifted_frames(5, 6 if full else 1, dev)
        ego = torch.from_numpy(S.ego_state(5)).to(dev)
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=6)
        parallel.ALWAYS_COLLECTIVE = True                            # run both all_gathers even with one rank
        t = {}
        got = harness.simple_test_sharded(net, frames, ego, n_steps=6, timings=t)
        torch.cuda.synchronize()
        same = [int((got['semantic_occ_%ds' % k][0] != want['semantic_occ_%ds' % k][0]).sum()) for k in range(7)]
        # the captured form (pipeline.ShardedSample) over RCCL: device-tensor collectives between the three graphs
        from preworld_amd.pipeline import ShardedSample
        ss = ShardedSample(net, frames, ego, n_steps=6)
        t2 = {}
        got = ss.run(timings=t2)
        got = ss.run(frames, ego, timings=t2)
        torch.cuda.synchronize()
        same += [int((got['semantic_occ_%ds' % k][0] != want['semantic_occ_%ds' % k][0]).sum()) for k in range(7)]
        assert ss.bad_replays() == (0, 2)
        t['captured'] = {k: round(v, 3) for k, v in t2.items()}
        q.put((dist.get_backend(), same, bool(got['semantic_occ_0s'][0].is_cuda), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('full', [False, True], ids=['C1-grid', 'C4-full-size-200x200x16'])
def test_rccl_device_tensor_all_gathers_world_size_1(full):
    """RCCL really initialised (backend 'nccl') and both data-path collectives of the sharded mode -- the frame all_gather (10.24 MB
    on the C1 grid; full: 2 x 81.92 MB through all_gather_into_tensor, the size BASELINE.json configs[3] moves) and the uint8 state
    all_gather -- executed on DEVICE tensors; one rank is all a 1-GPU box allows.  The per-phase timings are printed."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q, full))
    p.start()
    backend, same, on_dev, t = q.get(timeout=600)
    p.join(60)
    assert p.exitcode == 0
    assert backend == 'nccl' and on_dev and max(same) <= (8 if full else 0), (backend, same)       # (ties: see the two-rank test)
    cap = t.pop('captured')
    assert {k for k in t if 'bytes' not in k} == {'lift', 'gather_frames', 'encoder', 'decode', 'gather_states'}, t
    assert {k for k in cap if 'bytes' not in k} == {'lift', 'gather_frames', 'encoder', 'decode', 'gather_states'}, cap
    print('[sharded, RCCL world 1, %s] eager phase ms / bytes: %s' % ('200x200x16' if full else 'C1 grid', {k: round(v, 3) for k, v in t.items()}))
    print('[sharded, RCCL world 1, %s] captured (pipeline.ShardedSample) phase ms / bytes: %s' % ('200x200x16' if full else 'C1 grid', cap))


def _syncbn_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from preworld_amd import train
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        rs = np.random.RandomState(3)
        x_all = torch.from_numpy(rs.standard_normal((2, 4, 6, 8, 16)).astype(np.float32)).cuda()
        dy_all = torch.from_numpy(rs.standard_normal((2, 4, 6, 8, 16)).astype(np.float32)).cuda()
        gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)).cuda().requires_grad_()
        beta = torch.from_numpy(rs.standard_normal(16).astype(np.float32)).cuda().requires_grad_()
        x = x_all[rank:rank + 1].clone().requires_grad_()
        y, mean, var, cnt = train.BatchNormCL.apply(x, gamma, beta, None, 1e-5, True, True)
        (y * dy_all[rank:rank + 1]).sum().backward()
        q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), gamma.grad.cpu().numpy(), beta.grad.cpu().numpy(),
               mean.cpu().numpy(), var.cpu().numpy(), float(cnt)))
    finally:
        dist.destroy_process_group()


def test_syncbn_two_ranks_equal_one_process_on_the_whole_batch():
    """OccHead is built with norm_cfg SyncBN in every PreWorld config (preworld-7frame-finetune.py:39) and trained on 8 GPUs:
    the reference all-reduces the batch statistics (and the backward sums) over the ranks.  train.BatchNormCL(sync=True) on two
    ranks holding one sample each must equal ordinary batch-statistics BatchNorm over the two-sample batch."""
    import torch.nn.functional as F
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rs = np.random.RandomState(3)
    x_all = torch.from_numpy(rs.standard_normal((2, 4, 6, 8, 16)).astype(np.float32)).double().requires_grad_()
    dy_all = torch.from_numpy(rs.standard_normal((2, 4, 6, 8, 16)).astype(np.float32)).double()
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 16).astype(np.float32)).double().requires_grad_()
    beta = torch.from_numpy(rs.standard_normal(16).astype(np.float32)).double().requires_grad_()
    xf = x_all.reshape(-1, 16)
    yf = torch.relu(F.batch_norm(xf, None, None, gamma, beta, training=True, eps=1e-5)).reshape(x_all.shape)
    (yf * dy_all).sum().backward()
    for r in range(2):
        _, y, dx, dg, db, mean, var, cnt = res[r]
        assert cnt == 2 * 4 * 6 * 8
        np.testing.assert_allclose(y, yf[r:r + 1].detach().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dx, x_all.grad[r:r + 1].numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(mean, xf.mean(0).detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var, xf.var(0, unbiased=False).detach().numpy(), rtol=1e-5, atol=1e-6)
    # parameter gradients stay per rank (DDP averages them): their sum is the whole batch's gradient
    np.testing.assert_allclose(res[0][3] + res[1][3], gamma.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(res[0][4] + res[1][4], beta.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_bench_py_two_ranks_through_the_driver_launch_line():
    """the driver's own command for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with N = 2 on this 1-GPU box: the ranks find the device
    oversubscribed, rendezvous over gloo, run the replicas mode (no data-path collective), reduce the time with MAX and rank 0
    prints ONE JSON line with the whole-job value (exercise bench.py itself, not only the harness)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--settle-s', '0.1']
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['steps'] == 4 and r['scaling'] == 'weak' and r['value'] > 0
    assert abs(r['value'] - 2 * 4 / (r['ms_per_step'] * 4e-3)) < 1e-2 * r['value']           # whole-job: both ranks' samples
    assert r['rccl_world'] is None and 'OVERSUBSCRIBED' in r['config']['note']
    assert 'cpu_baseline' not in r and r['roofline']['kernel'].startswith('k_conv3d')
