#!/bin/bash
# Package power and shader clock of the GPU while a command runs (rocm-smi sampled twice a second):
#   bash tools/power_sample.sh <seconds> <command ...>      e.g.  bash tools/power_sample.sh 40 python3 tools/bench_conv.py --batch 700 --iters 400 --only L3.0.conv2
S=$1; shift
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
echo "idle:"; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | head -2
"$@" > /tmp/power_cmd.log 2>&1 &
PID=$!
sleep 4
N=0
while kill -0 $PID 2>/dev/null && [ $N -lt $((2 * S)) ]; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo
  sleep 0.5; N=$((N + 1))
done
wait $PID
tail -2 /tmp/power_cmd.log
