#!/bin/bash
# gpurun_out/r03 (tools/collect_profiles.sh) -> profiles/r03_* (the committed, judged copies)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03
HEAD=$(git rev-parse --short HEAD)
cp $O/bench_f16x3.json profiles/r03_bench_f16x3.json
cp $O/stats/p_kernel_stats.csv profiles/r03_f16x3_kernel_stats.csv
cp $O/pmc_busy.txt profiles/r03_f16x3_pmc_busy.txt
[ -f $O/train_stats/p_kernel_stats.csv ] && cp $O/train_stats/p_kernel_stats.csv profiles/r03_train_kernel_stats.csv
python3 tools/gpu_idle.py $O/stats > profiles/r03_f16x3_gpu_idle.txt
python3 tools/hbm_traffic.py $O/fetch $O/write > /tmp/hbm_r03.json
python3 - <<PY
import json
t = json.load(open('/tmp/hbm_r03.json'))
tot = sum(v['hbm_mb'] for k, v in t.items() if any(x in k for x in ('conv', 'pmap', 'nchw')))
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-runs, reduced by tools/hbm_traffic.py (tools/collect_profiles.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte requests as 64 bytes); per launch, batch 32 tiles of 1024x1024x3. Check: nchw_to_c8s (fused dequantiser) reads and writes exactly its 100.7 MB. deconv_s2_f16<3,4,8,1,true>@4194304 is deconv3+IGDN3 in its product-map form (writes 1.07 GB instead of 4.29 GB); pmap_gather replaces the last layer.',
       'commit': '$HEAD', 'per_step_gb_main_kernels': tot / 1e3, 'f16x3': t}
json.dump(out, open('profiles/r03_hbm_traffic.json', 'w'), indent=1)
b = json.load(open('profiles/r03_bench_f16x3.json'))
print('bench', round(b['value'], 1), 'tiles/s', round(b['ms_per_step'], 3), 'ms/step; dominant', b['roofline']['kernel'], round(b['roofline']['ms_per_launch'], 3), 'ms frac', round(b['roofline']['frac'], 3), '; HBM GB/step', round(tot / 1e3, 2))
PY
{ echo "# Round 3 -- f16x3 path: rocprofv3 kernel trace of \`python3 bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-sub-runs\`"; echo
  echo "Command (tools/collect_profiles.sh): \`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/stats -o p -- python3 bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-sub-runs\` (build $HEAD)."
  echo "Raw: \`r03_f16x3_kernel_stats.csv\` (--stats).  Grouped by (kernel, grid); first call of each group dropped; batch 32 tiles of 1024x1024x3."
  echo "Un-profiled bench of the same build: \`r03_bench_f16x3.json\` (HIP-event time of the dominant kernel, issued f16 MFMA rate against the 2500 TFLOP/s dense peak)."
  echo "Steady state of the profiled run: \`r03_f16x3_gpu_idle.txt\`; HBM bytes per launch: \`r03_hbm_traffic.json\`; clock / MFMA busy / waits: \`r03_f16x3_pmc_busy.txt\`."; echo
  echo "| kernel | grid (threads) | calls | avg us | min us | max us |"; echo "|---|---|---|---|---|---|"; grep -E "cae::" $O/kernel_summary.md; } > profiles/r03_f16x3_kernel_summary.md
head -2 profiles/r03_f16x3_gpu_idle.txt
