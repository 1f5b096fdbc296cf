#!/bin/bash
# clocks and socket power while the headline loop runs (is the search kernel power-limited?)
mkdir -p gpurun_out
python bench.py --steps 4000 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | head -8
  echo "--"
  sleep 0.7
done
wait $BP
tail -c 600 gpurun_out/power_bench.json | cut -c1-400
