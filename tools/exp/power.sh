#!/bin/bash
# samples rocm-smi power / clocks while the per-layer bench loops (is the f16x3 stack power-limited?)
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower --showperflevel < /dev/null > gpurun_out/power_idle.txt 2>&1
python tools/bench_layers.py --iters 2500 < /dev/null > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks < /dev/null 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' ; echo
  sleep 0.5
done > gpurun_out/power_samples.txt
wait $BP
tail -3 gpurun_out/power_bench.log
cat gpurun_out/power_samples.txt | cut -c1-400
grep -iE "max|power" gpurun_out/power_idle.txt | head
