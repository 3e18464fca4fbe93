cd /root/repo; export TMPDIR=/tmp
ONLY_STEP=1 PW_TORCH_PROFILE=1 timeout 300 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/train_tp.log
R=$PWD; cd /tmp; ONLY_STEP=1 N_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trk -o r -- python $R/tools/bench_train.py > $R/gpurun_out/train_rk.log 2>&1; cd $R
python tools/rocpd_stats.py $(find gpurun_out/trk -name '*.db' | head -1) | head -60 > gpurun_out/train_kernels.md; rm -rf gpurun_out/trk
cat gpurun_out/train_tp.log | head -50; cat gpurun_out/train_kernels.md
