# pipelined bench with different encode / decode pool sizes (CAE_ENC_THREADS / CAE_DEC_THREADS override the split)
for cfg in "16 16" "0 0" "8 12" "10 14"; do
  set -- $cfg
  CAE_ENC_THREADS=$1 CAE_DEC_THREADS=$2 python bench.py --steps 64 --no-cpu-baseline --no-sub-runs > gpurun_out/bench_split_$1_$2.json 2> gpurun_out/bench_split_$1_$2.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_split_$1_$2.json'))
print('enc/dec', '$1/$2', d['host_coder_threads'], 'tiles/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['host_ms_per_step'].items()}, d['host_use'])
PY
done
