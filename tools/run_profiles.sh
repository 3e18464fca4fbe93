set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/v6
python bench.py > gpurun_out/v6/bench.json 2> gpurun_out/v6/bench.err
python bench.py --in-flight 1 --no-cpu-baseline > gpurun_out/v6/bench_serial.json 2>> gpurun_out/v6/bench.err
ALGO=wino python tools/bench_layers.py > gpurun_out/v6/layers_wino.txt 2>&1
python tools/bench_layers.py > gpurun_out/v6/layers_direct.txt 2>&1
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/v6/prof -o r1 -- python $R/bench.py --steps 6 --warmup 2 --settle-s 0.2 --no-cpu-baseline > $R/gpurun_out/v6/prof_bench.json 2> $R/gpurun_out/v6/prof.err
WINO=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/v6/pmc_FETCH_SIZE -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/v6/prof.err
WINO=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/v6/pmc_WRITE_SIZE -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/v6/prof.err
WINO=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/v6/pmc_MFMA -o p -- python $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/v6/prof.err
cd $R
python tools/rocpd_stats.py gpurun_out/v6/prof/*/r1_results.db > gpurun_out/v6/kernel_stats.md 2>&1 || python tools/rocpd_stats.py $(find gpurun_out/v6/prof -name '*.db' | head -1) > gpurun_out/v6/kernel_stats.md 2>&1
for c in FETCH_SIZE WRITE_SIZE MFMA; do python tools/rocpd_pmc.py $(find gpurun_out/v6/pmc_$c -name '*.db' | head -1) > gpurun_out/v6/pmc_$c.md 2>&1; done
find gpurun_out/v6 -name '*.db' -delete
ls -la gpurun_out/v6
