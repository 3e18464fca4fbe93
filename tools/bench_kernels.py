"""Per-kernel micro-benchmarks (HIP events on the launch stream).  Development aid only."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops, synth as S  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3   # us


def bench_lss():
    dev = 'cuda:0'
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rig = S.synthetic_rig(6)
    gc = S.GRID_CONFIG_FULL
    fr = torch.stack(torch.meshgrid(torch.arange(88.), torch.arange(32.), torch.arange(88.),
                                    indexing='ij'), -1)
    from preworld_amd.modules import create_frustum
    fr = create_frustum(gc['depth'], S.INPUT_SIZE, S.DOWNSAMPLE).to(dev)
    lower = [-40., -40., -1.]
    interval = [0.4, 0.4, 0.4]
    size = [200, 200, 16]
    s2e, K, pr, pt, bda = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
    depth, feat = S.lift_inputs(0)
    d_t = T(depth)
    f_t = T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
    res = {}
    ipr, comb, tr = ops.lss_camera_matrices(s2e, K, pr)
    res['camera_matrices_us'] = timeit(lambda: ops.lss_camera_matrices(s2e, K, pr))
    vox = ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6)
    res['voxel_index_us'] = timeit(lambda: ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6))
    seg_start, order = ops.segment_sort(vox, 640000)
    res['segment_sort_us'] = timeit(lambda: ops.segment_sort(vox, 640000))
    out = torch.empty(640000, 32, device=dev)
    res['pool_dense_us'] = timeit(lambda: ops.bev_pool_dense(d_t, f_t, seg_start, order, 640000, 88, 32 * 88, out=out))
    kept = int(seg_start[-1])
    alg = 82e6 + 5.95e6 + 2.16e6 + kept * 4 + 2.56e6
    res['pool_dense_GBps'] = alg / res['pool_dense_us'] / 1e3
    rb, rd, rf, st, ln = ops.lss_ranks(seg_start, order, 640000, 88, 32 * 88)
    res['lss_ranks_us'] = timeit(lambda: ops.lss_ranks(seg_start, order, 640000, 88, 32 * 88))
    o2 = torch.zeros(1, 16, 200, 200, 32, device=dev)
    res['pool_intervals_us'] = timeit(lambda: ops.bev_pool_v2_forward(d_t, f_t, o2, rd, rf, rb, ln, st))
    res['memset_82MB_us'] = timeit(lambda: o2.zero_())
    res['kept'] = kept
    res['n_intervals'] = int(st.numel())
    return res


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='lss')
    a = ap.parse_args()
    out = {}
    if 'lss' in a.what:
        out['lss'] = bench_lss()
    print(json.dumps(out, indent=1))
