"""Multi-GPU modes for the hot path (one process per GPU, torch.distributed; backend 'nccl' is
RCCL over xGMI on ROCm, 'gloo' in the CPU tests).

* replicas (throughput mode, what bench.py measures): samples are independent, so each rank
  runs its own sample stream; the data path has NO collective (the reference is plain DDP with
  one sample per rank at test time, tools/dist_test.sh + apis/test.py:198-223).
* state-sharded decode (latency mode for ONE sample, BASELINE.json configs[3]): every rank holds
  the encoder output v0; rank r owns the output states {r, r+W, ...}.  State k is k applications
  of the pointwise residual MLP (preworld_temporal_traj.py:335-342) followed by OccHead, so a
  rank recomputes the cheap recursion locally up to its largest owned state and decodes only its
  own states; ONE all_gather of uint8 grids (0.64 MB each) assembles the 7 states everywhere.
* frame-sharded lift (the literal north_star wording): input frame f (key / adjacent) is lifted,
  pooled and pre-processed on rank f % W and the (B,Z,Y,X,32) fp32 features (81.92 MB each) are
  exchanged with ONE all_gather before cat + bev_encoder (bevdet_occ.py:266-267).  xGMI is point-to-point
  (7 links x ~153 GB/s per GPU): an all-gather of one 81.92 MB shard costs ~0.5 ms
  direct vs ~3.7 ms around a ring, i.e. comparable to the ~1 ms of lift+pre_process it saves --
  which is why replicas, not this mode, is the throughput mode.
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


def decode_states_sharded(v0, forecast_fn, decode_fn, n_states, group=None, mark=_no_mark):
    """v0: encoder output on every rank.  forecast_fn(v0, k) -> state-k features (k >= 1 applications
    of the recursion); decode_fn(features) -> uint8 occupancy grid.  Returns the list of all
    n_states grids (identical on every rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = owned_states(n_states, rank, world)
    local = {}
    for k in mine:
        feats = v0 if k == 0 else forecast_fn(v0, k)
        local[k] = decode_fn(feats)
    mark('decode')
    if world == 1 and not ALWAYS_COLLECTIVE:
        return [local[k] for k in range(n_states)]
    # pad every rank to the same number of slots so one all_gather moves everything
    slots = (n_states + world - 1) // world
    ref = next(iter(local.values())) if local else decode_fn(v0)
    send = torch.zeros((slots,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
    for i, k in enumerate(mine):
        send[i] = local[k]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    mark('gather_states')
    out = [None] * n_states
    for r in range(world):
        for i, k in enumerate(owned_states(n_states, r, world)):
            out[k] = recv[r][i]
    return out


def lift_frames_sharded(frames, lift_fn, out_shape, dtype, device, group=None, via_host=False, mark=_no_mark):
    """frames: list of per-frame inputs (present on every rank); frame f is lifted by rank f % W.  The features reach
    every rank through ONE all_gather of a (slots, *out_shape) buffer per rank (slots = ceil(F / W); 81.92 MB per frame at
    C3) -- on xGMI's point-to-point links every rank's shard travels its own link.  Returns the list of lifted features
    on every rank, in frame order.  via_host=True stages the buffers through host memory (gloo cannot all_gather device
    tensors; RCCL takes them as they are)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    F = len(frames)
    if world == 1 and not ALWAYS_COLLECTIVE:
        outs = [lift_fn(fr).contiguous() for fr in frames]
        assert all(tuple(o.shape) == tuple(out_shape) for o in outs)
        mark('lift')
        return outs
    slots = (F + world - 1) // world
    send = torch.zeros((slots,) + tuple(out_shape), dtype=dtype, device=device)
    for i, f in enumerate(range(rank, F, world)):
        buf = lift_fn(frames[f])
        assert tuple(buf.shape) == tuple(out_shape)
        send[i].copy_(buf)
    mark('lift')
    if via_host:
        send = send.cpu()
    recv = torch.empty((world * slots,) + tuple(out_shape), dtype=dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.to(device).view((world, slots) + tuple(out_shape))
    mark('gather_frames')
    return [recv[f % world, f // world] for f in range(F)]
