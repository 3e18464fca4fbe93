"""N>1 paths on CPU: world_size-2 gloo process groups exercising the sharding/gather logic of
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


def _decode(f):
    return f.argmax(-1).to(torch.uint8)


def _worker(rank, world, port, n_states, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        v0 = torch.randn(6, 5, 4, 18, generator=g)
        states = parallel.decode_states_sharded(v0, _forecast, _decode, n_states)
        seq = [_decode(v0 if k == 0 else _forecast(v0, k)) for k in range(n_states)]
        ok1 = all(torch.equal(a, b) for a, b in zip(states, seq))
        frames = [torch.full((3,), float(f)) for f in range(3)]
        lifted = parallel.lift_frames_sharded(frames, lambda fr: fr.repeat(4).view(4, 3) * 2.0 + rank * 0,
                                              (4, 3), torch.float32, 'cpu')
        ok2 = all(torch.equal(l, frames[f].repeat(4).view(4, 3) * 2.0) for f, l in enumerate(lifted))
        q.put((rank, ok1, ok2, parallel.owned_states(n_states, rank, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_states', [7, 2])
def test_state_sharded_decode_world2(n_states):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_states, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = []
    for rank, ok1, ok2, mine in res:
        assert ok1 and ok2, (rank, ok1, ok2)
        owned += mine
    assert sorted(owned) == list(range(n_states))       # every state decoded exactly once


def test_single_process_fallbacks():
    v0 = torch.randn(3, 3, 3, 18)
    out = parallel.decode_states_sharded(v0, _forecast, _decode, 4)
    assert len(out) == 4 and torch.equal(out[2], _decode(_forecast(v0, 2)))
    assert parallel.owned_states(7, 3, 8) == [3] and parallel.owned_states(7, 7, 8) == []
