#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while a conv kernel loops on the GPU.
# usage: tools/power_probe.sh <label> <env assignments...>
label=$1; shift
env "$@" LOOP_S=8 python tools/loop_conv.py > /tmp/loop_$label.log 2>&1 &
pid=$!
sleep 5
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '
  echo
  sleep 0.7
done
wait $pid
echo "$label: $(tail -1 /tmp/loop_$label.log)"
