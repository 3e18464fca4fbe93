"""Multi-GPU modes for the hot path (one process per GPU, torch.distributed; backend 'nccl' is
RCCL over xGMI on ROCm, 'gloo' in the CPU tests).

* replicas (throughput mode, what bench.py measures): samples are independent, so each rank
  runs its own sample stream; the data path has NO collective (the reference is plain DDP with
  one sample per rank at test time, tools/dist_test.sh + apis/test.py:198-223).
* state-sharded decode (latency mode for ONE sample, BASELINE.json configs[3]): every rank holds
  the encoder output v0; rank r owns the output states {r, r+W, ...}.  State k is k applications
  of the pointwise residual MLP (preworld_temporal_traj.py:335-342) followed by OccHead, so a
  rank runs the cheap recursion locally ONCE up to its largest owned state and decodes only its
  own states; ONE all_gather of uint8 grids (0.64 MB each) assembles the 7 states everywhere.
* frame-sharded lift (the literal north_star wording): input frame f (key / adjacent) is lifted,
  pooled and pre-processed on rank f % W and the (B,Z,Y,X,32) fp32 features (81.92 MB each) reach
  every rank before cat + bev_encoder (bevdet_occ.py:266-267): full rounds of W frames through one
  all_gather, the F % W frames of a partial round by broadcasts from their owners -- 2 frames on
  8 ranks = two broadcasts, 164 MB arriving per rank.  xGMI is point-to-point (7 links x ~153 GB/s
  per GPU): one 81.92 MB frame costs ~0.5 ms per link, i.e. comparable to the ~1 ms of
  lift + pre_process it saves -- which is why replicas, not this mode, is the throughput mode.
"""
import torch
import torch.distributed as dist

# tests set this to run the collectives even in a 1-rank group (RCCL init + device-tensor all_gather on one GPU)
ALWAYS_COLLECTIVE = False


def all_agree(ok, device, group=None, via_host=False):
    """True iff `ok` is true on every rank (one 4-byte MIN all-reduce; trivially `ok` without a process group)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cpu' if via_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def owned_states(n_states, rank, world):
    """round-robin ownership: cheap states (few recursion steps) and expensive ones interleave"""
    return list(range(rank, n_states, world))


def _no_mark(name):
    pass


def _global_rank(group, r):
    return dist.get_global_rank(group, r) if group is not None else r


def gather_states(local, n_states, group=None, grid_like=None, stats=None, send=None, recv=None):
    """the one all_gather of the state-sharded decode: `local` = {k: uint8 grid} of this rank's owned states -> list of all n_states grids.
    Every rank pads to the same number of slots (0.64 MB per slot at 200 x 200 x 16).  send / recv: optional preallocated buffers
    ((slots,) + shape and a list of `world` such tensors) so that a captured pass can write into / read from fixed addresses."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = owned_states(n_states, rank, world)
    slots = (n_states + world - 1) // world
    if send is None:
        if local:
            ref = next(iter(local.values()))
            shape, dtype, device = tuple(ref.shape), ref.dtype, ref.device
        else:
            shape, dtype, device = tuple(grid_like[0]), grid_like[1], torch.device(grid_like[2])
        send = torch.zeros((slots,) + shape, dtype=dtype, device=device)
        for i, k in enumerate(mine):
            send[i] = local[k]
    if recv is None:
        recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    if stats is not None:
        stats['states_bytes_received'] = (world - 1) * send.numel() * send.element_size()
    out = [None] * n_states
    for r in range(world):
        for i, k in enumerate(owned_states(n_states, r, world)):
            out[k] = recv[r][i]
    return out


def decode_states_sharded(v0, forecast_fn, decode_fn, n_states, group=None, mark=_no_mark, grid_like=None, stats=None):
    """v0: encoder output on every rank.  forecast_fn(v0, k) -> the features of states 1 .. k as a sequence `s` with s[j - 1] = state j
    (ONE pass of the recursion up to k: the forecast kernel writes every intermediate state anyway); decode_fn(features) -> uint8
    occupancy grid.  A rank runs the recursion once, up to its LARGEST owned state (round 6: it used to restart from v0 for every
    owned state -- 0 + 1 + ... + 6 = 21 steps at world 1 instead of 6).  Returns the list of all n_states grids (identical on every
    rank).  grid_like = (shape, dtype, device) of a decoded grid: a rank that owns no state (world > n_states: rank 7 of 8 with 7
    states) sizes its empty slot from it and decodes NOTHING; required on EVERY rank as soon as world > n_states, checked before any
    compute or collective so that all ranks fail together instead of one leaving the others inside all_gather.
    stats (dict): receives 'states_bytes_received' = bytes of other ranks' slots that reach this rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > n_states and grid_like is None:
        raise ValueError('decode_states_sharded: %d ranks for %d states -- some rank owns no state; pass grid_like on every rank'
                         % (world, n_states))
    mine = owned_states(n_states, rank, world)
    local = {}
    states = forecast_fn(v0, max(mine)) if mine and max(mine) > 0 else None
    for k in mine:
        local[k] = decode_fn(v0 if k == 0 else states[k - 1])
    mark('decode')
    if world == 1 and not ALWAYS_COLLECTIVE:
        return [local[k] for k in range(n_states)]
    out = gather_states(local, n_states, group, grid_like, stats)
    mark('gather_states')
    return out


def exchange_frames(lifted, F, out_shape, dtype, device, group=None, via_host=False, stats=None, static=None):
    """the frame exchange of the sharded lift: `lifted` = {f: this rank's lifted frames (f % W == rank)} -> list of all F frames on every
    rank.  Full rounds through ONE all_gather_into_tensor, the F % W frames of a partial round by broadcasts from their owners (see
    lift_frames_sharded).  static: optional dict the call keeps its receive buffers in ('recv' for the all_gather, 'bcast%d' per
    broadcast frame) -- a captured pass reads the frames at fixed addresses, so every call must deliver into the same tensors."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    out_shape = tuple(out_shape)
    n_full = F // world
    xdev = 'cpu' if via_host else device
    outs = [None] * F
    frame_bytes = dtype.itemsize
    for d in out_shape:
        frame_bytes *= d
    received = sent = 0
    static = static if static is not None else {}

    def buf(key, shape, dev):
        t = static.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != torch.device(dev):
            t = static[key] = torch.empty(shape, dtype=dtype, device=dev)
        return t

    def to_device(key, t):
        if not via_host:
            return t
        d = buf(key + '_dev', t.shape, device)
        d.copy_(t, non_blocking=True)
        return d
    if n_full:
        send = (lifted[rank][None] if n_full == 1 else torch.stack([lifted[j * world + rank] for j in range(n_full)], 0)).to(xdev)
        recv = buf('recv', (world * n_full,) + out_shape, xdev)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.view((world, n_full) + out_shape)
        for j in range(n_full):
            for r in range(world):
                outs[j * world + r] = lifted[j * world + r] if r == rank else to_device('recv%d_%d' % (r, j), recv[r, j])
        received += (world - 1) * n_full * frame_bytes
        sent += (world - 1) * n_full * frame_bytes
    works = []
    for f in range(n_full * world, F):
        owner = f % world
        if owner == rank:
            b = lifted[f].to(xdev) if via_host else lifted[f]
        else:
            b = buf('bcast%d' % f, out_shape, xdev)
        works.append((f, owner, b, dist.broadcast(b, src=_global_rank(group, owner), group=group, async_op=True)))
    for f, owner, b, w in works:
        w.wait()
        outs[f] = lifted[f] if owner == rank else to_device('bcast%d' % f, b)
        if owner == rank:
            sent += (world - 1) * frame_bytes
        else:
            received += frame_bytes
    if stats is not None:
        stats['frames_bytes_received'], stats['frames_bytes_sent'] = received, sent
    return outs


def lift_frames_sharded(frames, lift_fn, out_shape, dtype, device, group=None, via_host=False, mark=_no_mark, stats=None):
    """frames: list of per-frame inputs (present on every rank); frame f is lifted by rank f % W.  Returns the list of lifted
    features on every rank, in frame order.  The exchange moves every frame to every rank exactly once and nothing else:
      * the F // W full rounds (every rank owns one frame of the round) go through ONE all_gather_into_tensor;
      * the F % W frames of the last, partial round are BROADCAST by their owners -- with F = 2 frames on W = 8 ranks (BASELINE.json
        configs[3]) that is two broadcasts and 2 x 81.92 = 164 MB arriving per rank (82 MB on the two owners); round 4 padded the
        round to W slots and all-gathered 8 x 81.92 MB, six of them zeros.  On xGMI (point-to-point, 7 links x ~153 GB/s) an owner
        feeds its 7 peers over 7 different links, ~0.54 ms per frame if the links run concurrently.
    via_host=True stages the buffers through host memory (gloo cannot move device tensors; RCCL takes them as they are).
    stats (dict): receives 'frames_bytes_received' / 'frames_bytes_sent' of this rank (payload, excluding its own frames)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    F = len(frames)
    if world == 1 and not ALWAYS_COLLECTIVE:
        outs = [lift_fn(fr).contiguous() for fr in frames]
        assert all(tuple(o.shape) == tuple(out_shape) for o in outs)
        mark('lift')
        return outs
    out_shape = tuple(out_shape)
    n_full, tail = F // world, F % world
    mine = list(range(rank, F, world))
    lifted = {}
    for f in mine:
        buf = lift_fn(frames[f])
        assert tuple(buf.shape) == out_shape
        lifted[f] = buf.contiguous()
    mark('lift')
    outs = exchange_frames(lifted, F, out_shape, dtype, device, group, via_host, stats)
    mark('gather_frames')
    return outs
