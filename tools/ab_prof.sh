set -x
R=$PWD
O=$R/gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for w in old new; do
  if [ $w = old ]; then D=$R/_ab_old; else D=$R; fi
  PW_LIFT_STREAMS=0 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r1 -- python $D/bench.py --in-flight 1 --steps 20 --warmup 2 --settle-s 0.5 --no-cpu-baseline > $O/bench_$w.json 2> $O/err_$w.txt
  python $R/tools/rocpd_stats.py $(find $O/prof_$w -name '*.db' | head -1) > $O/stats_$w.md 2>&1
  rm -rf $O/prof_$w
done
