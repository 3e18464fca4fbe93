# round-6 profile set in ONE gpurun call:  V=r06a bash tools/run_profiles_r06.sh
#   bench lines (default, serial, C2, C5, sharded world 1), rocprofv3 --kernel-trace --stats of the bench (default + serial),
#   separate PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy | SQ mix) over an eager, strictly serial run of the same step,
#   and FETCH / WRITE passes over the C5 render head
set -x
V=${V:-r06a}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/$V
mkdir -p $O
python -c "from preworld_amd import build; print(build.source_hash())" > $O/build_id.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --in-flight 1 --no-cpu-baseline --no-extra > $O/bench_serial.json 2>> $O/bench.err
python bench.py --in-flight 3 --no-cpu-baseline --no-extra > $O/bench_if3.json 2>> $O/bench.err
python bench.py --config C2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err
python bench.py --mode sharded --steps 30 --no-cpu-baseline > $O/bench_sharded_w1.json 2>> $O/bench.err
python bench.py --config C5 > $O/bench_c5.json 2>> $O/bench.err
python tools/bench_h2.py > $O/layers_h2.txt 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o r1 -- python $R/bench.py --steps 6 --warmup 2 --settle-s 0.2 --no-cpu-baseline --no-extra > $O/prof_bench.json 2> $O/prof.err
PW_LIFT_STREAMS=0 rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r1 -- python $R/bench.py --in-flight 1 --steps 20 --warmup 2 --settle-s 0.5 --no-cpu-baseline --no-extra > $O/prof_bench_serial.json 2>> $O/prof.err
EAGER="env PW_LIFT_STREAMS=0 python $R/bench.py --no-graph --in-flight 1 --steps 4 --warmup 1 --settle-s 0.0 --no-cpu-baseline --no-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH_SIZE -o p -- $EAGER > /dev/null 2>> $O/prof.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_WRITE_SIZE -o p -- $EAGER > /dev/null 2>> $O/prof.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_MFMA -o p -- $EAGER > /dev/null 2>> $O/prof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_SQ -o p -- $EAGER > /dev/null 2>> $O/prof.err
C5="python $R/bench.py --config C5 --steps 20 --warmup 2 --settle-s 0.2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_C5_FETCH_SIZE -o p -- $C5 > /dev/null 2>> $O/prof.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_C5_WRITE_SIZE -o p -- $C5 > /dev/null 2>> $O/prof.err
cd $R
python tools/rocpd_stats.py $(find $O/prof -name '*.db' | head -1) > $O/kernel_stats.md 2>&1
python tools/rocpd_stats.py $(find $O/prof_serial -name '*.db' | head -1) > $O/kernel_stats_serial.md 2>&1
for c in FETCH_SIZE WRITE_SIZE MFMA SQ C5_FETCH_SIZE C5_WRITE_SIZE; do python tools/rocpd_pmc.py $(find $O/pmc_$c -name '*.db' | head -1) > $O/pmc_$c.md 2>&1; done
find $O -name '*.db' -delete
rm -rf $O/prof $O/prof_serial $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA $O/pmc_SQ $O/pmc_C5_FETCH_SIZE $O/pmc_C5_WRITE_SIZE
ls -la $O
