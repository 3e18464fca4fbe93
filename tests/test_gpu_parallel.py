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


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from preworld_amd import harness, synth as S
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dev = 'cuda:0'
        torch.cuda.set_device(0)
        net = harness.build_model(harness.model_cfg(S.GRID_CONFIG_C1), S.synth_state_dict(0), dev)
        frames = harness.lifted_frames(5, 1, dev)
        ego = torch.from_numpy(S.ego_state(5)).to(dev)
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=6)
        got = harness.simple_test_sharded(net, frames, ego, n_steps=6, gather_on_host=True)
        same = [bool(torch.equal(got['semantic_occ_%ds' % k][0].cpu(), want['semantic_occ_%ds' % k][0].cpu())) for k in range(7)]
        # with_prev=False (C2's frame handling): the adjacent frame is dropped, its channel slice is zeros
        net.with_prev = False
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=2)
        got = harness.simple_test_sharded(net, frames, ego, n_steps=2, gather_on_host=True)
        same += [bool(torch.equal(got['semantic_occ_%ds' % k][0].cpu(), want['semantic_occ_%ds' % k][0].cpu())) for k in range(3)]
        q.put((rank, same))
    finally:
        dist.destroy_process_group()


def test_sharded_lift_and_decode_two_ranks_equal_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1]
    for _, same in res:
        assert all(same), same


def _rccl_worker(port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from preworld_amd import harness, parallel, synth as S
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)            # 'nccl' == RCCL on ROCm
    try:
        dev = 'cuda:0'
        net = harness.build_model(harness.model_cfg(S.GRID_CONFIG_C1), S.synth_state_dict(0), dev)
        frames = harness.lifted_frames(5, 1, dev)
        ego = torch.from_numpy(S.ego_state(5)).to(dev)
        with torch.no_grad():
            want = net.simple_test_from_lift(frames, ego, n_steps=6)
        parallel.ALWAYS_COLLECTIVE = True                            # run both all_gathers even with one rank
        got = harness.simple_test_sharded(net, frames, ego, n_steps=6)
        torch.cuda.synchronize()
        same = [bool(torch.equal(got['semantic_occ_%ds' % k][0], want['semantic_occ_%ds' % k][0])) for k in range(7)]
        q.put((dist.get_backend(), same, bool(got['semantic_occ_0s'][0].is_cuda)))
    finally:
        dist.destroy_process_group()


def test_rccl_device_tensor_all_gathers_world_size_1():
    """RCCL really initialised (backend 'nccl') and both data-path collectives of the sharded mode -- the 10.24 MB frame
    all_gather (C1 grid) and the uint8 state all_gather -- executed on DEVICE tensors; one rank is all a 1-GPU box allows."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    backend, same, on_dev = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert backend == 'nccl' and on_dev and all(same), (backend, same)
