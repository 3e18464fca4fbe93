"""gpurun_out/<V>/ (written by tools/run_profiles_r04.sh) -> the committed summaries under profiles/:
  <V>_bench_kernel_stats.md  bench lines + rocprofv3 --kernel-trace --stats tables (serial and 2-in-flight)
  <V>_pmc.md                 the four PMC passes
  <V>_layers_h2.txt          per-layer conv timings (tools/bench_h2.py)
  <V>_bench.json             the default bench line
  r02_pmc_traffic.json       what bench.py reads for roofline.traffic (tools/make_pmc_json.py)
usage: python tools/collect_profiles.py r02b "one-line description of the build"
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(v, what):
    src, dst = os.path.join(ROOT, 'gpurun_out', v), os.path.join(ROOT, 'profiles')
    rd = lambda f: open(os.path.join(src, f)).read()
    lines = []
    for f in ('bench', 'bench_serial', 'bench_if3', 'bench_if4', 'bench_c2', 'bench_sharded_w1'):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'show_bench.py'), os.path.join(src, f + '.json')],
                             capture_output=True, text=True).stdout.splitlines()
        lines.append('* %s: %s' % (f, out[0] if out else 'n/a'))
    with open(os.path.join(dst, v + '_bench_kernel_stats.md'), 'w') as o:
        o.write('# %s — rocprofv3 --kernel-trace --stats of bench.py (C3, 1x MI355X), %s\n\n' % (v, what))
        o.write('Command set: `V=%s bash tools/run_profiles_%s.sh` (one gpurun call).  bench lines of the same build:\n\n' % (v, v[:3]))
        o.write('\n'.join(lines) + '\n\n')
        o.write('## strictly serial (--in-flight 1): kernel time per step == wall time per step\n\n' + rd('kernel_stats_serial.md') + '\n')
        o.write('## default (2 samples in flight): kernels of the two streams overlap, so the sum exceeds wall time\n\n' + rd('kernel_stats.md') + '\n')
    with open(os.path.join(dst, v + '_pmc.md'), 'w') as o:
        o.write('# %s — PMC passes over one eager, strictly serial C3 step (separate passes: FETCH_SIZE | WRITE_SIZE | MFMA busy + GUI '
                'active | SQ mix)\n\nCommand: `rocprofv3 --pmc <counters> --kernel-trace -d ... -- python bench.py --no-graph --in-flight 1 '
                '--steps 4 --warmup 1 --settle-s 0 --no-cpu-baseline`; tables by tools/rocpd_pmc.py (KB per dispatch for the TCC counters). '
                'FETCH_SIZE is reported at half the bytes for 16 B/lane reads on gfx950 (doubled in profiles/rNN_pmc_traffic.json for the '
                'MFMA kernels); WRITE_SIZE exact. Matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).\n\n' % v)
        for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA", "SQ", "C5_FETCH_SIZE", "C5_WRITE_SIZE"):
            if os.path.exists(os.path.join(src, 'pmc_%s.md' % c)):
                o.write('## %s%s\n%s\n' % (c, ' (bench.py --config C5: the render head)' if c.startswith('C5') else '', rd('pmc_%s.md' % c)))
    shutil.copy(os.path.join(src, 'layers_h2.txt'), os.path.join(dst, v + '_layers_h2.txt'))
    shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, v + '_bench.json'))
    if os.path.exists(os.path.join(src, 'bench_c5.json')):
        shutil.copy(os.path.join(src, 'bench_c5.json'), os.path.join(dst, v + '_bench_c5.json'))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'make_pmc_json.py'), src, os.path.join(dst, v[:3] + '_pmc_traffic.json')])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
