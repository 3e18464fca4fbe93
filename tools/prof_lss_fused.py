"""Loop ops.lss_lift_pool at the C3 frame shape (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from preworld_amd import ops, synth as S  # noqa: E402
from preworld_amd.modules import create_frustum  # noqa: E402

dev = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rig = S.synthetic_rig(6)
fr = create_frustum(S.GRID_CONFIG_FULL['depth'], S.INPUT_SIZE, S.DOWNSAMPLE).to(dev)
lower, interval, size = [-40., -40., -1.], [0.4, 0.4, 0.4], [200, 200, 16]
cams = [T(rig[k]) for k in ('sensor2ego', 'intrin', 'post_rot', 'post_tran', 'bda')]
depth, feat = S.lift_inputs(0)
d_t = T(depth).view(1, 6, 88, 32, 88)
f_t = T(np.ascontiguousarray(feat.transpose(0, 1, 3, 4, 2)))
out = torch.empty(640000, 32, device=dev)
h2 = os.environ.get('H2', '1') == '1'
for _ in range(int(os.environ.get('N', '200'))):
    ops.lss_lift_pool(fr, *cams, lower, interval, size, d_t, f_t, out=out, out_h2=h2)
torch.cuda.synchronize()
