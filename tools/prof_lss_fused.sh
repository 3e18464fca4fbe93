#!/bin/bash
# kernel times of ops.lss_lift_pool under the PW_LSS_DEBUG experiments -> gpurun_out/lss_fused_<tag>.md
export TMPDIR=/tmp
for dbg in ${DBGS:-0 1 3 7 8}; do
  PW_LSS_DEBUG=$dbg timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/lssf_$dbg -o r -- python tools/prof_lss_fused.py > gpurun_out/lssf_$dbg.log 2>&1
  echo "== PW_LSS_DEBUG=$dbg"; python tools/rocpd_stats.py gpurun_out/lssf_$dbg/r_results.db | grep k_lss
done
