#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wider or gdn_layer or variant" < /dev/null > gpurun_out/wide.log 2>&1; rc=$?
tail -25 gpurun_out/wide.log
if grep -q "Memory access fault" gpurun_out/wide.log; then exit 9; fi
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tests/fuzz/fuzz_parity.py 300 77 < /dev/null > gpurun_out/wide_fuzz.log 2>&1 || { tail -30 gpurun_out/wide_fuzz.log; exit 2; }
tail -2 gpurun_out/wide_fuzz.log
