"""Pretty-print the per-kernel table of a bench.py JSON line (stdin or file)."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in txt.splitlines() if l.startswith('{')][-1]
r = json.loads(line)
print('value %.1f %s  ms/step %.3f  latency %s' % (r['value'], r['unit'], r['ms_per_step'], r['config'].get('single_sample_latency_ms')))
rf = r['roofline']
print('dominant', rf['kernel'], 'bound', rf['bound'], 'achieved', rf['achieved'], rf['unit'], 'frac', rf['frac'])
tot = 0
for k, v in rf['all_kernels'].items():
    tot += v['us_per_step']
    print('%-58s %8.1f us  x%-2d  %7.1f TF  %7.1f GB/s' % (k, v['us_per_step'], v['launches'], v['tflops'], v['alg_GBps']))
print('sum of probed kernels %.1f us' % tot)
if 'cpu_baseline' in r:
    print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['sample'][:200])
