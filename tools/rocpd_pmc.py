"""Per-kernel mean of one PMC counter from a rocprofv3 rocpd database (view counters_collection).

    python tools/rocpd_pmc.py gpurun_out/pmc_FETCH_SIZE/p_results.db
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '')[:70]


def main(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    need = {'kernel_name': None, 'counter_name': None, 'value': None, 'dispatch_id': None}
    for c in cols:
        for k in need:
            if c == k or (need[k] is None and k.split('_')[0] in c and k.split('_')[-1] in c):
                need[k] = c
    q = "select %s, %s, %s, %s from counters_collection" % (need['kernel_name'], need['counter_name'], need['value'], need['dispatch_id'])
    agg = {}
    for kname, cname, val, did in cur.execute(q):
        d = agg.setdefault((short(kname), cname), {})
        d[did] = d.get(did, 0.0) + float(val)          # sum over XCD / instance rows of one dispatch
    print('| kernel | counter | dispatches | mean per dispatch | min | max |')
    print('|---|---|---|---|---|---|')
    for (k, c), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        v = list(d.values())
        print('| `%s` | %s | %d | %.4g | %.4g | %.4g |' % (k, c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == '__main__':
    main(sys.argv[1])
