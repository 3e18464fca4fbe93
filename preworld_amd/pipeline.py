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


def clone_frame(f):
    """static copy of one frame's inputs.  depthnet_tail hands the context over channels-last and says so through an attribute of the
    tensor object (modules.LSSViewTransformer.depthnet_tail): a clone must keep it, or the lift reads the buffer channel-first"""
    out = {}
    for k, v in f.items():
        out[k] = v.clone()
        if getattr(v, '_pw_channels_last', False):
            out[k]._pw_channels_last = True
    return out


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
        self.frames = [clone_frame(f) for f in frames]
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


class ShardedSample:
    """BASELINE.json configs[3] -- ONE sample across the ranks of a node (frames lifted on different ranks, RCCL exchange of the
    per-frame voxel features before the encoder, states forecast + decoded round-robin, all_gather of the uint8 grids; the reference's
    join point is bevdet_occ.py:266-267, its result gather apis/test.py:198-223) -- as three captured compute phases between the two
    collectives (round 6; harness.simple_test_sharded is the eager form: ~60 launches + a calibration all-reduce and a host sync per
    call):
        graph A   this rank's frames: LSS lift + pooling + pre_process -> fp32 features in static buffers   (nothing on other ranks)
        exchange  parallel.exchange_frames: all_gather_into_tensor (full rounds) / broadcasts from the owners (partial round)
        graph B   cat [adjacent ..., key] -> one h2 split -> CustomResNet3D -> LSSFPN3D -> final_conv
        graph C   ONE pass of the forecast recursion up to this rank's largest owned state, OccHead on its owned states -> uint8 slots
        gather    parallel.gather_states: one all_gather of the 0.64 MB slots
    Activation ranges: calibrated ONCE at construction (ops.ranged over eager passes; the ranks agree through a MIN all-reduce per
    calibration pass, as the passes contain collectives), then every replay re-records its maxima and the device-side audit tallies
    the replays that left the window (bad_replays(), like CapturedSample) -- no host sync and no collective for the ranges per sample.
    run(frames, ego) copies a sample's inputs (every rank holds all of them) into the static buffers and returns the reference's
    result dict; timings (dict) receives the phases' milliseconds from HIP events and the payload bytes."""

    def __init__(self, net, frames, ego, n_steps=6, group=None, gather_on_host=False):
        import torch.distributed as dist
        from . import parallel
        from .modules import as_f32, precision
        self.net, self.n_steps, self.group, self.via_host = net, n_steps, group, gather_on_host
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or parallel.ALWAYS_COLLECTIVE
        vt = net.img_view_transformer
        _, _, size = vt._grid()
        self.size = [int(v) for v in size]
        n = net.num_adj + 1
        self.frames = [clone_frame(f) for f in (frames[:n] if net.with_prev else frames[:1])]
        self.ego = ego.clone()
        self.dev = self.ego.device
        self.F = len(self.frames)
        f0 = self.frames[0]
        self.B, self.C, self.n = f0['sensor2keyego'].shape[0], vt.out_channels, n
        self.h2 = precision() == 'h2' and self.C % 32 == 0
        self.n_states = n_steps + 1
        self.mine_f = list(range(self.rank, self.F, self.world))
        self.mine_s = parallel.owned_states(self.n_states, self.rank, self.world)
        self.shape = (self.B, self.size[2], self.size[1], self.size[0], self.C)
        self._as_f32, self._par = as_f32, parallel
        self.static = {}                                        # receive buffers of the frame exchange (fixed addresses)
        self.rctx = ops.RangeCtx(self.dev)
        self.host_rng = torch.zeros(tuple(self.rctx.compact.shape), dtype=torch.int32, pin_memory=True)
        slots = (self.n_states + self.world - 1) // self.world
        gdev = 'cpu' if gather_on_host else self.dev
        self.send = torch.zeros((slots, self.size[0], self.size[1], self.size[2]), dtype=torch.uint8, device=self.dev)
        self.send_x = self.send if not gather_on_host else torch.zeros(self.send.shape, dtype=torch.uint8, pin_memory=True)
        self.recv = [torch.empty(self.send.shape, dtype=torch.uint8, device=gdev) for _ in range(self.world)]
        self.lifted, self.v0, self.all_frames = {}, None, None
        # calibration: eager passes with the collectives inside, all ranks agree on whether another pass runs
        with torch.no_grad():
            ops.ranged(self._eager_pass, self.rctx,
                       agree=lambda ok: parallel.all_agree(ok, self.dev, group, gather_on_host))
        torch.cuda.synchronize()
        # capture: the same pass cut at the collectives; slots are handed out in call order, so A -> B -> C under ONE begin()
        self.gA, self.gB, self.gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        kw = dict(capture_error_mode='thread_local')
        with torch.no_grad(), ops.use_range(self.rctx):
            with torch.cuda.graph(self.gA, **kw):
                self.rctx.begin()
                self._phase_lift()
            pool = self.gA.pool()
            self.all_frames = self._exchange()                  # eager, now: graph B is captured on the buffers it delivers into
            with torch.cuda.graph(self.gB, pool=pool, **kw):
                self._phase_encoder()
            with torch.cuda.graph(self.gC, pool=pool, **kw):
                self._phase_decode()
                self.rctx.fold()
                self.rctx.audit()
                self.host_rng.copy_(self.rctx.compact, non_blocking=True)
        torch.cuda.synchronize()

    # ---- the three compute phases (shared by the eager calibration pass and the captures)
    def _phase_lift(self):
        for f in self.mine_f:
            # fp32 values travel: an h2 buffer means nothing without its rank-local exponent ((hi + lo) * 2^e is exact)
            self.lifted[f] = self._as_f32(self.net.lift_frame_cl(out_h2=self.h2, **self.frames[f])).contiguous()

    def _exchange(self, stats=None):
        if not self.collective:
            return [self.lifted[f] for f in range(self.F)]
        return self._par.exchange_frames(self.lifted, self.F, self.shape, torch.float32, self.dev, self.group, self.via_host, stats,
                                         static=self.static)

    def _phase_encoder(self):
        lifted, C, n = self.all_frames, self.C, self.n
        net, h2 = self.net, self.h2
        # [adjacent ..., key] (bevdet_occ.py:266): every frame goes straight into its channel slice of ONE buffer -- on the split-fp16
        # path as ONE h2 tensor under ONE range slot of this pass's table (the same data, hence the same exponent, on every rank)
        x = torch.empty(self.shape[:-1] + (n * C,), device=self.dev, dtype=torch.float32)
        slot = ops.new_slot(self.dev) if h2 else None
        for j in range(n):                                                      # frame j sits at channel block n - 1 - j
            lo, hi = (n - 1 - j) * C, (n - j) * C
            if j < len(lifted):
                if h2:
                    ops.f32_to_h2(lifted[j], out=ops.H2(x[..., lo:hi], slot))
                else:
                    x[..., lo:hi].copy_(lifted[j])
            else:
                x[..., lo:hi].zero_()                                           # with_prev=False: zeros (bevdet_occ.py:243-258)
        self.v0 = net.final_conv.forward_cl(net.bev_encoder_cl(ops.H2(x, slot) if h2 else x, out_h2=h2), out_h2=h2)

    def _phase_decode(self):
        net = self.net
        if not self.mine_s:
            return
        kmax = max(self.mine_s)
        states = net.forecast_cl(self.v0, self.ego, kmax, out_h2=self.h2)[0] if kmax > 0 else None
        inplace = self.h2 and self.B == 1       # the kernel writes a (Z,Y,X) result as the (X,Y,Z)-contiguous payload grid, in place
        i = 0
        while i < len(self.mine_s):
            k = self.mine_s[i]
            run = 1                              # consecutive owned forecast states decode in ONE launch (world 1: states 1 .. 6)
            while k > 0 and i + run < len(self.mine_s) and self.mine_s[i + run] == k + run:
                run += 1
            if k == 0:
                feats = self.v0
            elif run == 1:
                feats = states[k - 1]
            else:
                feats = states.view((states.shape[0] * self.B,) + tuple(self.v0.shape[1:]))[(k - 1) * self.B:(k - 1 + run) * self.B]
            dst = self.send[i:i + run]
            if inplace:
                net.occupancy_head.decode_cl(feats, transposed=True, occ_out=dst.permute(0, 3, 2, 1))
            else:
                occ = net.occupancy_head.decode_cl(feats, transposed=True)
                dst.copy_(occ.permute(0, 3, 2, 1).reshape(run, self.B, *dst.shape[1:])[:, 0])      # batch element 0 (:306)
            i += run

    def _gather(self, stats=None):
        if not self.collective:
            return [self.send[i] for i in range(self.n_states)]
        if self.via_host:
            self.send_x.copy_(self.send, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return self._par.gather_states(None, self.n_states, self.group, stats=stats, send=self.send_x, recv=self.recv)

    def _eager_pass(self):
        self._phase_lift()
        self.all_frames = self._exchange()
        self._phase_encoder()
        self._phase_decode()
        return self._gather()

    def run(self, frames=None, ego=None, timings=None):
        """one sample: optional new inputs into the static buffers, A -> exchange -> B -> C -> gather.  Returns
        {'semantic_occ_%ds': [(X,Y,Z) uint8]} (static buffers: consume before the next run)."""
        if frames is not None:
            dsts, srcs = [self.ego], [ego]
            for dst, src in zip(self.frames, frames):
                for k, v in src.items():
                    dsts.append(dst[k])
                    srcs.append(v)
            ops.copy_many(dsts, srcs)
        ev = []

        def mark(name):
            if timings is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))
        mark('start')
        self.gA.replay()
        mark('lift')
        got = self._exchange(timings)
        assert all(a.data_ptr() == b.data_ptr() for a, b in zip(got, self.all_frames)), 'frame exchange left its static buffers'
        mark('gather_frames')
        self.gB.replay()
        mark('encoder')
        self.gC.replay()
        mark('decode')
        grids = self._gather(timings)
        mark('gather_states')
        if timings is not None:
            torch.cuda.synchronize()
            for (_, a), (name, b) in zip(ev[:-1], ev[1:]):
                timings[name] = a.elapsed_time(b)
        return {'semantic_occ_%ds' % k: [g] for k, g in enumerate(grids)}

    def bad_replays(self):
        """(passes that left their calibrated activation ranges, passes audited) since capture (synchronises)"""
        torch.cuda.synchronize()
        return self.rctx.audited()
