lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Flags" | cut -c1-600 > gpurun_out/lscpu.txt
nproc >> gpurun_out/lscpu.txt
for t in 16 8 12; do
  CAE_CODER_THREADS=$t python bench.py --steps 64 --no-cpu-baseline --no-sub-runs > gpurun_out/bench_thr$t.json 2> gpurun_out/bench_thr$t.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_thr$t.json'))
print('threads', $t, 'tiles/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), d['host_ms_per_step'])
PY
done
