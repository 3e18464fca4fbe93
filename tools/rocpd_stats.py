"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md
(rocprofv3 --kernel-trace --stats in ROCm 7.2 writes this database by default.)"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '')
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |' %
              (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == '__main__':
    main(sys.argv[1])
