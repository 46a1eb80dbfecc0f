#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "slide or pipelin or range or bench or driver or round" < /dev/null > gpurun_out/b2_pytest.log 2>&1 || { tail -30 gpurun_out/b2_pytest.log; exit 1; }
tail -2 gpurun_out/b2_pytest.log
for i in 1 2; do
timeout -k 10 400 python bench.py --no-cpu-baseline --no-sub-runs < /dev/null > gpurun_out/b2_bench$i.json 2> gpurun_out/b2_bench$i.err || { tail -30 gpurun_out/b2_bench$i.err; exit 4; }
python - <<PY
import json
d=json.load(open('gpurun_out/b2_bench$i.json'))
print(round(d['value'],1), round(d['ms_per_step'],3), d['gpu_ms_per_step'], {k: round(v,2) for k,v in d['host_ms_per_step'].items()})
PY
done
