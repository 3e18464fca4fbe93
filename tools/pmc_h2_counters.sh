# SQ / LDS / TCP counters of the pipelined split-fp16 conv on one layer shape (SHAPE = index into tools/bench_h2.py SHAPES)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/pmc_h2${TAG:-}; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TD_BUSY_avr" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" ; do
  i=$((i+1))
  QUICK=1 SHAPE=${SHAPE:-2} timeout 180 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p -- python $R/tools/bench_h2.py > /dev/null 2> $O/p$i.err
  python $R/tools/rocpd_pmc.py $(find $O/p$i -name '*.db' | head -1) 2>&1 | grep -i "kernel\|---\|conv3d_h2" > $O/p$i.md
  rm -rf $O/p$i
done
