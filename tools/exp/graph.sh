#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_train.py tests/test_dist.py -x -q < /dev/null > gpurun_out/graph_test.log 2>&1; rc=$?
tail -40 gpurun_out/graph_test.log
if grep -q "Memory access fault" gpurun_out/graph_test.log; then exit 9; fi
[ $rc = 0 ] || exit $rc
for mode in eager graph; do
  timeout -k 10 300 python tools/bench_train.py 16 40 256 $mode < /dev/null > gpurun_out/train_$mode.json 2> gpurun_out/train_$mode.err || { tail -20 gpurun_out/train_$mode.err; exit 3; }
  cat gpurun_out/train_$mode.json
done
