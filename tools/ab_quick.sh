# serial + default bench of the working tree and of _ab_old on ONE box
for i in 1 2; do
  for d in . _ab_old; do
    (cd $d && PW_LIFT_STREAMS=0 python bench.py --in-flight 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$d serial', d['value'], d['ms_per_step'])")
    (cd $d && python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$d default', d['value'], d['ms_per_step'])")
  done
done
