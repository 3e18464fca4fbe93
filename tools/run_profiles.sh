set -x
V=${V:-v7}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$V
python bench.py > gpurun_out/$V/bench.json 2> gpurun_out/$V/bench.err
python bench.py --in-flight 1 --no-cpu-baseline > gpurun_out/$V/bench_serial.json 2>> gpurun_out/$V/bench.err
python bench.py --in-flight 3 --no-cpu-baseline > gpurun_out/$V/bench_if3.json 2>> gpurun_out/$V/bench.err
python bench.py --in-flight 4 --no-cpu-baseline > gpurun_out/$V/bench_if4.json 2>> gpurun_out/$V/bench.err
python tools/loop_occ.py > gpurun_out/$V/occ.txt 2>&1; WINO=1 python tools/loop_occ.py >> gpurun_out/$V/occ.txt 2>&1
ALGO=wino python tools/bench_layers.py > gpurun_out/$V/layers_wino.txt 2>&1
python tools/bench_layers.py > gpurun_out/$V/layers_direct.txt 2>&1
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$V/prof -o r1 -- python $R/bench.py --steps 6 --warmup 2 --settle-s 0.2 --no-cpu-baseline > $R/gpurun_out/$V/prof_bench.json 2> $R/gpurun_out/$V/prof.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$V/prof_serial -o r1 -- python $R/bench.py --in-flight 1 --steps 20 --warmup 2 --settle-s 0.5 --no-cpu-baseline > $R/gpurun_out/$V/prof_bench_serial.json 2>> $R/gpurun_out/$V/prof.err
WINO=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/$V/pmc_FETCH_SIZE -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/$V/prof.err
WINO=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/$V/pmc_WRITE_SIZE -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/$V/prof.err
WINO=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/$V/pmc_MFMA -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/$V/prof.err
WINO=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/$V/pmc_SQ -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/$V/prof.err
cd $R
python tools/rocpd_stats.py gpurun_out/$V/prof/*/r1_results.db > gpurun_out/$V/kernel_stats.md 2>&1 || python tools/rocpd_stats.py $(find gpurun_out/$V/prof -name '*.db' | head -1) > gpurun_out/$V/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find gpurun_out/$V/prof_serial -name '*.db' | head -1) > gpurun_out/$V/kernel_stats_serial.md 2>&1
for c in FETCH_SIZE WRITE_SIZE MFMA SQ; do python tools/rocpd_pmc.py $(find gpurun_out/$V/pmc_$c -name '*.db' | head -1) > gpurun_out/$V/pmc_$c.md 2>&1; done
find gpurun_out/$V -name '*.db' -delete
ls -la gpurun_out/$V
