# serial bench (in-flight 1) of the real library and of each experiment variant on ONE box
R=$PWD
for v in "" both; do
  if [ -z "$v" ]; then unset PW_LIB_PATH; tag=real; else export PW_LIB_PATH=$R/preworld_amd/csrc/variants/libpreworld_hip_$v.so; tag=$v; fi
  for i in 1 2; do
    PW_LIFT_STREAMS=0 python bench.py --in-flight 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$tag', d['value'], d['ms_per_step'])"
  done
done
(cd _ab_old && for i in 1 2; do PW_LIFT_STREAMS=0 python bench.py --in-flight 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('old', d['value'], d['ms_per_step'])"; done)
