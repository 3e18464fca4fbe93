( ./tools/probes/mfma_power 6 > gpurun_out/mfma_power.txt 2>&1 ) &
pid=$!
sleep 3
for i in $(seq 1 22); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/mfma_power_smi.txt
  echo >> gpurun_out/mfma_power_smi.txt
  sleep 0.8
done
wait $pid
