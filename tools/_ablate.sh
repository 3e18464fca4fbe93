QUICK=1 python tools/bench_h2.py 2>/dev/null > gpurun_out/h2_quick.txt
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_range.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -15 > gpurun_out/h2_tests.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/h2_bench.json
