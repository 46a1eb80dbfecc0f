for cfg in "3 8 16" "4 8 16" "5 8 16" "4 16 16" "5 16 16"; do
  set -- $cfg
  CAE_PIPELINE_DEPTH=$1 CAE_ENC_THREADS=$2 CAE_DEC_THREADS=$3 python bench.py --steps 96 --no-cpu-baseline --no-sub-runs > gpurun_out/bench_d$1_$2_$3.json 2> gpurun_out/bench_d$1_$2_$3.err < /dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_d$1_$2_$3.json'))
print('depth/enc/dec', '$1/$2/$3', 'tiles/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['host_ms_per_step'].items()}, round(d['host_use']['cpus_busy'],1))
PY
done
