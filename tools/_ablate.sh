H2=1 CIN=64 COUT=64 python tools/probe_conv_pipe.py 2>/dev/null > gpurun_out/probe_h2_64_new.txt
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r03_gpu_all4.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
