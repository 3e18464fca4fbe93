"""Which Python lines of the training step launch the torch glue kernels (copy_ / fill_ / add_ / ...) on large tensors: a
TorchDispatchMode logs every aten op with its element count and the innermost preworld_amd frame.  Development aid."""
import collections
import os
import runpy
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['ONLY_STEP'] = '1'
os.environ['N_STEPS'] = '1'
import torch
from torch.utils._python_dispatch import TorchDispatchMode

if os.environ.get('WHICH', 'finetune') == 'pretrain':              # the pre-train (render) step of bench.py's extra.c5
    import bench
    step = bench.pretrain_step_ms('cuda:0', 'step')
    for _ in range(2):
        step()
else:
    g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_train.py'), run_name='__main__')
    step = g['train_step']
log = collections.defaultdict(lambda: [0, 0])
MIN_ELEMS = int(os.environ.get('MIN_ELEMS', '200000'))          # MIN_ELEMS=0 BY_COUNT=1: every launch, busiest source lines first
SKIP = ('aten.view', 'aten.permute', 'aten.slice.', 'aten.select', 'aten.detach', 'aten.alias', 'aten.t.', 'aten.reshape', 'aten._unsafe_view',
        'aten.unsqueeze', 'aten.squeeze', 'aten.expand', 'aten.as_strided', 'aten.transpose', 'aten.empty', 'aten.is_', 'aten.sym_', 'aten.stride')


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            n = 0
            for a in list(args) + [out]:
                if isinstance(a, torch.Tensor):
                    n = max(n, a.numel())
                elif isinstance(a, (list, tuple)):
                    n = max([n] + [t.numel() for t in a if isinstance(t, torch.Tensor)])
            if n >= MIN_ELEMS:
                where = 'autograd engine'
                for f in reversed(traceback.extract_stack()[:-1]):
                    if 'preworld_amd' in f.filename or 'bench_train' in f.filename:
                        where = '%s:%d %s' % (os.path.basename(f.filename), f.lineno, (f.line or '').strip()[:80])
                        break
                k = (name, where)
                log[k][0] += n
                log[k][1] += 1
        return out


with Log():
    step()
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -kv[1][1 if os.environ.get('BY_COUNT') else 0])
print('aten ops on tensors of >= %d elements in ONE training step (x4 bytes, x2-3 for read+write):' % MIN_ELEMS)
for (name, where), (n, c) in rows[:int(os.environ.get('ROWS', '60'))]:
    print('%8.1f M elems  x%-3d %-32s %s' % (n / 1e6, c, name, where))
