#!/bin/bash
# round 5: kernel times (rocprofv3 --kernel-trace --stats) and HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of one frame's
# lift + pooling (ops.lss_lift_pool, C3 frame shape, h2 output) -> gpurun_out/lss_r05.md
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/lss_r05; mkdir -p $O
cd /tmp
N=200 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/tools/prof_lss_fused.py > $O/kt.log 2>&1
N=20 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o p -- python $R/tools/prof_lss_fused.py > $O/f.log 2>&1
N=20 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o p -- python $R/tools/prof_lss_fused.py > $O/w.log 2>&1
cd $R
{ echo "## kernel times, 200 calls"; python tools/rocpd_stats.py $(find $O/kt -name '*.db' | head -1) | grep -i "k_lss\|Name\|---";
  echo; echo "## FETCH_SIZE (KB per dispatch, as reported)"; python tools/rocpd_pmc.py $(find $O/f -name '*.db' | head -1) | grep -i "k_lss\|kernel\|---";
  echo; echo "## WRITE_SIZE (KB per dispatch)"; python tools/rocpd_pmc.py $(find $O/w -name '*.db' | head -1) | grep -i "k_lss\|kernel\|---"; } > gpurun_out/lss_r05.md 2>&1
rm -rf $O
cat gpurun_out/lss_r05.md
