"""N eager pre-train (render) steps of bench.py's extra.c5.pretrain_step, for rocprofv3 --kernel-trace --stats.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

step = bench.pretrain_step_ms('cuda:0', 'step')
for _ in range(int(os.environ.get('N_STEPS', '8'))):
    step()
torch.cuda.synchronize()
