R=$PWD; O=$R/gpurun_out/prb; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for cfg in "38400" "38400 scene" "3072"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/p_$tag -o r -- python $R/tools/prof_render_bwd.py $cfg > /dev/null 2> $O/err_$tag.txt
  python $R/tools/rocpd_stats.py $(find $O/p_$tag -name '*.db' | head -1) > $O/stats_$tag.md 2>&1
  rm -rf $O/p_$tag
done
