# cache-level counters of the stereo cost volume at the reference shape (one gpurun call)
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/stereo_pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "FETCH_SIZE TCP_TOTAL_ACCESSES_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $O/$n -o p -- python $R/tools/bench_kernels.py --what stereo > /dev/null 2>> $O/err.txt
  python $R/tools/rocpd_pmc.py $(find $O/$n -name '*.db' | head -1) 2>&1 | grep -i "stereo\|kernel" > $O/$n.md
  rm -rf $O/$n
done
cat $O/*.md
