#!/bin/bash
# full GPU validation of the current build: parity suite, fuzz, per-layer times, bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu < /dev/null > gpurun_out/fc_pytest.log 2>&1 || { tail -30 gpurun_out/fc_pytest.log; exit 1; }
tail -2 gpurun_out/fc_pytest.log
if grep -q "Memory access fault" gpurun_out/fc_pytest.log; then exit 9; fi
timeout -k 10 300 python tests/fuzz/fuzz_parity.py 300 < /dev/null > gpurun_out/fc_fuzz.log 2>&1 || { tail -30 gpurun_out/fc_fuzz.log; exit 2; }
tail -2 gpurun_out/fc_fuzz.log
timeout -k 10 200 python tools/bench_layers.py < /dev/null > gpurun_out/fc_layers.log 2>&1 || { tail -30 gpurun_out/fc_layers.log; exit 3; }
grep -E "conv|deconv|pmap|sum|total" gpurun_out/fc_layers.log | tail -30
timeout -k 10 400 python bench.py < /dev/null > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err || { tail -30 gpurun_out/fc_bench.err; exit 4; }
cat gpurun_out/fc_bench.json
if grep -q "Memory access fault" gpurun_out/fc_*.log gpurun_out/fc_bench.err; then exit 9; fi
