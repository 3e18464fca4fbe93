cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; mkdir -p gpurun_out/pw; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $R/gpurun_out/pw/counters.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES" ; do
  i=$((i+1))
  WINO=1 N=1 timeout 120 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pw/p$i -o p -- python $R/tools/bench_conv.py > /dev/null 2> $R/gpurun_out/pw/p$i.err
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pw/p$i -name '*.db' | head -1) 2>&1 | grep -i "kernel\|---\|wino" > $R/gpurun_out/pw/p$i.md
  rm -rf $R/gpurun_out/pw/p$i
done
