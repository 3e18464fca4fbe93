"""LSS stage of one frame at the C3 shape (6 cams, 88x32x88 frustum, 200x200x16 grid): voxel index, sort, pooling (fp32 and h2
output).  Development aid; """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops, synth as S  # noqa: E402
from preworld_amd.modules import create_frustum  # noqa: E402
from bench_h2 import timeit  # noqa: E402

dev = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rig = S.synthetic_rig(6)
gc = S.GRID_CONFIG_FULL
fr = create_frustum(gc['depth'], S.INPUT_SIZE, S.DOWNSAMPLE).to(dev)
lower, interval, size = [-40., -40., -1.], [0.4, 0.4, 0.4], [200, 200, 16]
s2e, K, pr, pt, bda = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
depth, feat = S.lift_inputs(0)
d_t = T(depth)
f_t = T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
ipr, comb, tr = ops.lss_camera_matrices(s2e, K, pr)
t_idx = timeit(lambda: ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6))
vox = ops.lss_voxel_index(fr, ipr, pt, comb, tr, bda, lower, interval, size, 1, 6)
srt = lambda: ops.segment_sort(vox, 640000, aux_div=88 * 32 * 88, aux_mod=32 * 88, long_threshold=ops.LONG_SEGMENT)
vs = srt()
t_sort = timeit(srt)
out = torch.empty(640000, 32, device=dev)
t_pool = timeit(lambda: ops.bev_pool_dense(d_t, f_t, vs, out=out))
t_pool_h2 = timeit(lambda: ops.bev_pool_dense(d_t, f_t, vs, out=out, out_h2=True))
kept = int(vs.seg_start[-1])
alg = 4.0 * (640000 * 32 + d_t.numel() + f_t.numel() + 640001 + 2 * kept)
print('%s: voxel_index %.1f us | sort %.1f us | pool fp32 %.1f us (%.2f TB/s) | pool h2 %.1f us (%.2f TB/s) | frame total %.1f us' % (
    ' '.join('%s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('PW_')) or 'default',
    t_idx, t_sort, t_pool, alg / t_pool * 1e-6, t_pool_h2, alg / t_pool_h2 * 1e-6, t_idx + t_sort + t_pool_h2), flush=True)
# where does the pooling time go: the dense sweep alone (long segments walked by their lane group) vs with the long-segment blocks
vs0 = ops.VoxelSort(vs.seg_start, vs.order, vs.order_feat, None, None, vs.n_keys)
t0 = timeit(lambda: ops.bev_pool_dense(d_t, f_t, vs0, out=out))
print('pool fp32 without the long-segment blocks (same result, long segments inside the sweep): %.1f us; long segments: %d' % (
    t0, int(vs.n_long.item())), flush=True)
# the slot-based single-call form (ops.lss_lift_pool): camera matrices + index + lists + pooling in 5 launches
cams = (s2e, K, pr, pt, bda)
d5 = d_t.view(1, 6, 88, 32, 88)
t_fused = timeit(lambda: ops.lss_lift_pool(fr, *cams, lower, interval, size, d5, f_t, out=out))
t_fused_h2 = timeit(lambda: ops.lss_lift_pool(fr, *cams, lower, interval, size, d5, f_t, out=out, out_h2=True))
ref = ops.bev_pool_dense(d_t, f_t, vs)
got = ops.lss_lift_pool(fr, *cams, lower, interval, size, d5, f_t)
print('lss_lift_pool (5 launches, whole frame): fp32 %.1f us | h2 %.1f us | same bits as the sort path: %s' % (
    t_fused, t_fused_h2, bool(torch.equal(ref, got))), flush=True)
