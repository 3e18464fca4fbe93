# Which kernels does the PIPELINED step time actually depend on?  Builds of the library in which one kernel family returns at once
# (tools/build_variant.py abl_X --only=file -DPW_X_SKIP_X; results are garbage, timing only) next to the real library on ONE box:
#   for n in smallconv bigconv occ fc lss; do python tools/build_variant.py abl_$n --only=<file> -DPW_X_SKIP_<N>; done
#   gpurun -- 'bash tools/ablate_step.sh'          -> profiles/r04_step_ablation.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PW_BENCH_ABLATION=1
for v in base abl_smallconv abl_bigconv abl_occ abl_fc abl_lss; do
  if [ $v = base ]; then L=""; else L="PW_LIB_PATH=$PWD/preworld_amd/csrc/variants/libpreworld_hip_$v.so"; fi
  for f in 2 1; do
    r=$(env $L timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 100 --in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms per step  %.1f samples/s' % (d['ms_per_step'], d['value']))" 2>&1 | tail -1)
    echo "$v in_flight=$f: $r"
  done
done
