cd /root/repo
for v in 0 1 2 0 1 2; do
  PW_CONV_WINO_H2=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('wino_h2=$v', d['value'], d['ms_per_step'], d['config'].get('single_sample_latency_ms'))"
done
PW_CONV_WINO_H2=2 timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_range.py -x -q -m gpu 2>&1 | tail -5
