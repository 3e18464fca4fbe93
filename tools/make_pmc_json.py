"""gpurun_out/<V>/pmc_{FETCH,WRITE}_SIZE.md (tools/rocpd_pmc.py tables, KB per dispatch) -> profiles/<round>_pmc_traffic.json
{'by_kernel': {rocprof kernel name: {fetch_bytes, write_bytes, fetch_reported_bytes, dispatches}}} -- what bench.py reads for
`roofline.traffic`.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide (16 B/lane)
streaming reads, so kernels whose reads are dwordx4 / `buffer_load ... lds` are doubled (WIDE below); WRITE_SIZE is exact on
this box (81.92 MB fill = 80 000 KB)."""
import json
import re
import sys

WIDE = ('k_conv3d_h2', 'k_conv3d_gather', 'k_occ_head', 'k_forecast', 'k_conv3d_wino', 'k_conv3d_k3s1')


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r'\| `(.+?)` \| (\w+) \| (\d+) \| ([\d.e+-]+) \|', line)
        if m:
            out[m.group(1)] = (int(m.group(3)), float(m.group(4)) * 1024.0)
    return out


def main(vdir, dst):
    import os
    f, w = table(vdir + '/pmc_FETCH_SIZE.md'), table(vdir + '/pmc_WRITE_SIZE.md')
    if os.path.exists(vdir + '/pmc_C5_FETCH_SIZE.md'):        # the render head's own passes (bench.py --config C5): k_render_* only
        for into, extra in ((f, table(vdir + '/pmc_C5_FETCH_SIZE.md')), (w, table(vdir + '/pmc_C5_WRITE_SIZE.md'))):
            into.update({k: v for k, v in extra.items() if k.startswith(('k_render', 'k_rb_', 'k_attr'))})
    by = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith('k_'):
            continue
        fr = f.get(k, (0, 0.0))[1]
        by[k] = dict(fetch_reported_bytes=int(fr), fetch_bytes=int(fr * (2 if k.startswith(WIDE) else 1)),
                     write_bytes=int(w.get(k, (0, 0.0))[1]), dispatches=f.get(k, w.get(k))[0],
                     fetch_doubled=bool(k.startswith(WIDE)))
    bid = vdir + '/build_id.txt'       # written on the GPU box by the profile script: hash of the sources the profiled library was built from
    json.dump({'_note': __doc__, 'source': vdir, 'build_id': open(bid).read().strip() if os.path.exists(bid) else None, 'by_kernel': by},
              open(dst, 'w'), indent=1)
    print('wrote', dst, len(by), 'kernels')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
