"""Serving-side composition of the hot path: synthetic or real lifted inputs in HBM -> LSS voxel
pooling -> voxel encoder -> forecast decode -> occupancy heads, mirroring the call order of
PreWorld4DTraj.simple_test (mmdet3d/models/detectors/preworld_temporal_traj.py:212-370).

`CapturedSample` records one sample's ~60 kernel launches into a hipGraph (torch.cuda.CUDAGraph) over
static input/output buffers: a replay has no launch gaps (kernel time == wall time in
profiles/r01_bench_kernel_stats_v5.md) and no per-launch host work, which is what the C3 step needs
once the kernels themselves run in 10-1000 us."""
import numpy as np
import torch

from . import ops


def to_dev(a, dev='cuda:0'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


class CapturedSample:
    """net: modules.PreWorld4DTraj (eval).  frames / ego: example inputs defining the shapes (list of
    dicts as for simple_test_from_lift, (B,1,21) ego states).  run(frames, ego) copies new inputs into
    the static buffers, replays the graph and returns the (static) result dict: consume or clone the
    outputs before the next run()."""

    fused_input_copy = True          # run(): one pw_copy_many launch for the inputs (False: one torch copy per tensor; A/B)

    def __init__(self, net, frames, ego, n_steps=6, d2h=False):
        """d2h=True: the replay also delivers the reference's host payload (preworld_temporal_traj.py:311-366: every
        semantic_occ / geo_occ grid as a contiguous uint8 (X,Y,Z) host array) -- one device-side gather of the (X,Y,Z)
        views into a (n_grids, X, Y, Z) buffer and ONE async copy into pinned host memory (`self.host`, rows in the
        order of `self.host_keys`) instead of the reference's 14 synchronous .cpu() calls per sample."""
        self.net, self.n_steps, self.d2h = net, n_steps, d2h
        self.host = self.host_keys = None
        self.frames = [{k: v.clone() for k, v in f.items()} for f in frames]
        self.ego = ego.clone()
        # Activation ranges of the split-fp16 path (ops.RangeCtx): this sample's own table of per-tensor exponents.  The
        # warm-up calibrates it (repeats the pass until every h2 tensor's maximum sits in the window); the captured kernels
        # read the exponents from the table, clear and re-record the maxima on every replay, and the table travels to pinned
        # host memory with the results: ranges_ok() is the per-replay check, recalibrate() the (rare) repair.
        self.rctx = ops.RangeCtx(self.ego.device)
        self.host_rng = torch.zeros(tuple(self.rctx.compact.shape), dtype=torch.int32, pin_memory=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():        # warm-up off the capture stream: fills the packed-weight
            ops.ranged(self._step, self.rctx)                  # caches, sets the kernels' LDS attributes, calibrates the ranges
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a HIP call from another thread of the process (the RCCL watchdog of a multi-GPU run polls
        # events) must not invalidate this thread's capture
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'), torch.no_grad(), ops.use_range(self.rctx):
            self.rctx.begin()
            self.out = self._step()
            self.rctx.fold()
            self.rctx.audit()                                   # sticky device counters: EVERY replay is range-checked (bad_replays())
            self.host_rng.copy_(self.rctx.compact, non_blocking=True)

    def _step(self):
        kw = dict(n_steps=self.n_steps) if hasattr(self.net, 'forecast_cl') else {}
        args = (self.frames, self.ego) if hasattr(self.net, 'forecast_cl') else (self.frames,)
        out = self.net.simple_test_from_lift(*args, **kw)
        if self.d2h:
            keys = [k for k in out if k.startswith(('semantic_occ', 'geo_occ'))]
            g = out.get('grids')
            if g is not None and g.numel() == len(keys) * out[keys[0]][0].numel() and all(
                    out[k][0].data_ptr() == g.data_ptr() + i * out[k][0].numel() for i, k in enumerate(keys)):
                dev = g.view((len(keys),) + tuple(out[keys[0]][0].shape))  # the kernels wrote the payload rows in place
            else:
                dev = torch.stack([out[k][0] for k in keys])              # (n_grids, X, Y, Z) contiguous uint8
            if self.host is None:
                self.host = torch.empty(dev.shape, dtype=torch.uint8, pin_memory=True)
                self.host_keys = keys
            self.host.copy_(dev, non_blocking=True)
        return out

    def replay(self):
        self.graph.replay()
        return self.out

    def eager(self):
        """the same pass as a replay -- same static buffers, same exponent table, same slot order -- launched eagerly on the
        current stream: results are bit-identical to replay() (all kernels are deterministic)"""
        with torch.no_grad(), ops.use_range(self.rctx):
            self.rctx.begin()
            out = self._step()
            self.rctx.fold()
            return out

    def ranges_ok(self):
        """after the replay has completed (stream / device synchronised): did every h2 tensor of it stay inside the
        representable window under the exponents it ran with?  False -> recalibrate() and replay that sample again."""
        return not self.rctx.check(self.host_rng)

    def bad_replays(self):
        """(replays that left their calibrated activation ranges, replays audited) since capture (synchronises): the device-side
        tally of pw_rng_audit, which every replay feeds -- unlike ranges_ok(), which only sees the LAST replay's table"""
        torch.cuda.synchronize()
        return self.rctx.audited()

    def recalibrate(self):
        """re-derive the exponents from the inputs currently in the static buffers (eager passes; the graph keeps reading
        the same table).  Waits for everything in flight first: a replay must not see the table change under it."""
        torch.cuda.synchronize()
        with torch.no_grad():
            ops.ranged(self._step, self.rctx)
        torch.cuda.synchronize()

    def run_checked(self, frames, ego):
        """run() + wait + range check, repaired and replayed once if the sample left the calibrated window"""
        out = self.run(frames, ego)
        torch.cuda.current_stream().synchronize()
        if not self.ranges_ok():
            self.recalibrate()
            out = self.replay()
            torch.cuda.current_stream().synchronize()
            if not self.ranges_ok():
                raise ops._lib.PreworldHipError('activation ranges still outside the window after recalibration: slots %s'
                                                % self.rctx.check(self.host_rng))
        return out

    def run(self, frames, ego):
        dsts, srcs = [self.ego], [ego]
        for dst, src in zip(self.frames, frames):
            for k, v in src.items():
                dsts.append(dst[k])
                srcs.append(v)
        if self.fused_input_copy and all(s.is_cuda and s.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape for d, s in zip(dsts, srcs)):
            ops.copy_many(dsts, srcs)                      # one launch for the ~15 input tensors
        else:
            for d, s in zip(dsts, srcs):
                d.copy_(s, non_blocking=True)
        return self.replay()
