#!/bin/bash
# Coder pool sizes / lockstep width under a CPU budget: tools/sweep_host_budget.sh [cpus]  (one bench line per setting)
CPUS=${1:-8}
cd "$(dirname "$0")/.."
for cfg in "4 8 4" "3 5 2" "4 4 2" "2 6 2" "4 8 2" "8 8 2" "8 8 4" "2 6 4" "3 8 4" "6 8 1"; do
  set -- $cfg
  CAE_ENC_THREADS=$1 CAE_DEC_THREADS=$2 CAE_CODER_LOCKSTEP=$3 python bench.py --cpus $CPUS --steps 32 --warmup 2 --no-sub-runs --no-cpu-baseline 2>/dev/null \
    | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('enc dec lock = $cfg:', round(d['value']), 'tiles/s', round(d['ms_per_step'], 2), 'ms/step', round(d['host_use']['cpus_busy'], 2), 'CPUs busy', {k: round(v, 1) for k, v in d['host_ms_per_step'].items()})"
done
