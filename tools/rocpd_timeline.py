"""One training step as a timeline: the kernels between the last two launches of a marker kernel (default k_voxel_loss_finish: one per
step) of a rocprofv3 --kernel-trace database, in launch order, with their duration and the idle gap before each.  Development aid:
shows which torch glue kernels sit between which library kernels.     python tools/rocpd_timeline.py results.db [marker] [min_us]"""
import sqlite3
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from rocpd_stats import short  # noqa: E402


def main(db, marker='k_voxel_loss_finish', min_us=0.0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, start, end from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = (marks[-2] + 1, marks[-1] + 1) if len(marks) >= 2 else (0, len(rows))
    step = rows[a:b]
    busy = sum(e - s for _, s, e in step) / 1e3
    print('%d kernels, %.1f us busy, %.1f us wall' % (len(step), busy, (step[-1][2] - step[0][1]) / 1e3))
    prev_end = step[0][1]
    for name, s, e in step:
        d = (e - s) / 1e3
        if d >= min_us:
            print('%9.1f us  gap %7.1f  %s' % (d, (s - prev_end) / 1e3, short(name)[:110]))
        prev_end = e


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:3]), *(float(v) for v in sys.argv[3:4]))
