#!/bin/bash
# final validation of the round: GPU suite, smoke(), training bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu < /dev/null > gpurun_out/final_pytest.log 2>&1 || { tail -30 gpurun_out/final_pytest.log; exit 1; }
tail -2 gpurun_out/final_pytest.log
if grep -q "Memory access fault" gpurun_out/final_pytest.log; then exit 9; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > gpurun_out/final_smoke.log 2>&1 || { tail -20 gpurun_out/final_smoke.log; exit 2; }
tail -3 gpurun_out/final_smoke.log
timeout -k 10 300 python tools/bench_train.py 16 40 256 < /dev/null > gpurun_out/final_train.json 2> gpurun_out/final_train.err || { tail gpurun_out/final_train.err; exit 3; }
cat gpurun_out/final_train.json
