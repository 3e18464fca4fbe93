"""N>1 paths on CPU: world_size-2 and world_size-8 gloo process groups exercising the sharding/gather logic of
preworld_amd.parallel with stand-in compute (no GPU kernels are called here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from preworld_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _forecast(v0, k):
    v = v0
    for _ in range(k):
        v = v + torch.tanh(v * 0.5 + 0.1)
    return v


STEPS = [0]


def _forecast_states(v0, k):
    """the contract of decode_states_sharded's forecast_fn: states 1 .. k from ONE pass of the recursion"""
    out, v = [], v0
    for _ in range(k):
        v = v + torch.tanh(v * 0.5 + 0.1)
        STEPS[0] += 1
        out.append(v)
    return out


def _decode(f):
    return f.argmax(-1).to(torch.uint8)


def _worker(rank, world, port, n_states, n_frames, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        v0 = torch.randn(6, 5, 4, 18, generator=g)
        decoded = []

        def decode(f):
            decoded.append(1)
            return _decode(f)
        st = {}
        STEPS[0] = 0
        states = parallel.decode_states_sharded(v0, _forecast_states, decode, n_states, grid_like=((6, 5, 4), torch.uint8, 'cpu'), stats=st)
        mine = parallel.owned_states(n_states, rank, world)
        assert STEPS[0] == (max(mine) if mine else 0)          # ONE recursion up to the largest owned state (round 6), not one per state
        seq = [_decode(v0 if k == 0 else _forecast(v0, k)) for k in range(n_states)]
        ok1 = all(torch.equal(a, b) for a, b in zip(states, seq))
        # a rank decodes exactly its own states -- none at all if it owns none (rank 7 of 8 with 7 states)
        ok1 = ok1 and len(decoded) == len(parallel.owned_states(n_states, rank, world))
        frames = [torch.full((3,), float(f)) for f in range(n_frames)]
        lifts = []

        def lift(fr):
            lifts.append(1)
            return fr.repeat(4).view(4, 3) * 2.0
        lifted = parallel.lift_frames_sharded(frames, lift, (4, 3), torch.float32, 'cpu', stats=st)
        ok2 = all(torch.equal(l, frames[f].repeat(4).view(4, 3) * 2.0) for f, l in enumerate(lifted))
        n_mine = len(range(rank, n_frames, world))
        ok2 = ok2 and len(lifts) == n_mine
        # the exchange delivers every frame this rank did not lift exactly once: (F - own) x 48 bytes, no padding slots
        ok2 = ok2 and st['frames_bytes_received'] == (n_frames - n_mine) * 48 and st['frames_bytes_sent'] == n_mine * (world - 1) * 48
        q.put((rank, ok1, ok2, parallel.owned_states(n_states, rank, world)))
    finally:
        dist.destroy_process_group()


def _run(world, n_states, n_frames):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_states, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = []
    for rank, ok1, ok2, mine in res:
        assert ok1 and ok2, (rank, ok1, ok2)
        owned += mine
    assert sorted(owned) == list(range(n_states))       # every state decoded exactly once


@pytest.mark.parametrize('n_states,n_frames', [(7, 3), (2, 2)])
def test_state_sharded_decode_world2(n_states, n_frames):
    """3 frames on 2 ranks: one full round through all_gather_into_tensor + one broadcast"""
    _run(2, n_states, n_frames)


@pytest.mark.parametrize('n_frames', [2, 11])
def test_sharded_world8(n_frames):
    """BASELINE.json configs[3]'s world size: 7 states on 8 ranks -- rank 7 owns none, decodes nothing and still
    takes part in the all_gather -- and 2 frames on 8 ranks: two broadcasts, each rank receives the frames it did not lift and no
    padding (round 4 moved 8 slots per rank); 11 frames: one full round + a partial one."""
    _run(8, 7, n_frames)


def test_single_process_fallbacks():
    v0 = torch.randn(3, 3, 3, 18)
    STEPS[0] = 0
    out = parallel.decode_states_sharded(v0, _forecast_states, _decode, 4)
    assert len(out) == 4 and torch.equal(out[2], _decode(_forecast(v0, 2))) and STEPS[0] == 3        # 3 steps, not 0 + 1 + 2 + 3
    assert parallel.owned_states(7, 3, 8) == [3] and parallel.owned_states(7, 7, 8) == []


def _worker_no_grid_like(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        try:
            parallel.decode_states_sharded(torch.randn(2, 2, 2, 18), _forecast_states, _decode, 2)       # 3 ranks, 2 states, no grid_like
            q.put((rank, 'returned'))
        except ValueError:
            q.put((rank, 'raised'))
    finally:
        dist.destroy_process_group()


def test_missing_grid_like_fails_on_every_rank_before_the_collective():
    """with more ranks than states and no grid_like, EVERY rank raises before any compute or collective (round 5: only the
    rank without a state raised, after the others had entered all_gather -- they hung until the timeout)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_no_grid_like, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 'raised'), (1, 'raised'), (2, 'raised')]
