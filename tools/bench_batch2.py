"""VERDICT r02 item 6 as an experiment: ONE captured step carrying B = 2 samples (every launch sees batch 2: the 8x100x100 / 4x50x50
levels, the forecast and the OccHead get twice the work items) against two B = 1 samples in flight on two streams (bench.py's
default).  Prints samples/s of both on the same box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from preworld_amd import harness  # noqa: E402
from preworld_amd.pipeline import CapturedSample  # noqa: E402

dev = 'cuda:0'
net, _ = bench.build_net(dev, 'C3')


def timed(fn, n=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def stack2(a, b):
    fr = [{k: torch.cat([fa[k], fb[k]], 0) for k in fa} for fa, fb in zip(a[0], b[0])]
    return fr, torch.cat([a[1], b[1]], 0)


ins = [bench.make_inputs(dev, seed=k, n_frames=2) for k in range(4)]
caps1 = [CapturedSample(net, *ins[k], n_steps=6, d2h=False) for k in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def two_in_flight():
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            caps1[i].replay()


t1 = timed(two_in_flight)
print('2 x (B = 1) in flight: %.3f ms per pair -> %.1f samples/s' % (t1 * 1e3, 2 / t1), flush=True)
cap2 = CapturedSample(net, *stack2(ins[0], ins[1]), n_steps=6, d2h=False)
t2 = timed(cap2.replay)
print('1 x (B = 2):           %.3f ms per pair -> %.1f samples/s' % (t2 * 1e3, 2 / t2), flush=True)
cap2b = CapturedSample(net, *stack2(ins[2], ins[3]), n_steps=6, d2h=False)
caps2 = [cap2, cap2b]


def two_b2():
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            caps2[i].replay()


t3 = timed(two_b2)
print('2 x (B = 2) in flight: %.3f ms per four -> %.1f samples/s' % (t3 * 1e3, 4 / t3), flush=True)
