#!/bin/bash
# Collects the round's profile artifacts on the GPU box into gpurun_out/r03/ (copy what is to be judged into profiles/).
#   1. un-profiled bench line (default command)
#   2. rocprofv3 --kernel-trace --stats of a shorter run            -> kernel summary
#   3. separate --pmc passes: FETCH_SIZE | WRITE_SIZE | busy / wait  -> HBM traffic per launch, clock, MFMA busy
# (counters are collected in runs of their own, with --kernel-trace only: gpurun refuses other trace domains with --pmc)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err < /dev/null
echo "bench done" >> $O/progress.txt
ARGS="--steps 24 --warmup 2 --no-cpu-baseline --no-sub-runs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py $ARGS > $O/stats_bench.json 2> $O/stats.err < /dev/null
echo "stats done" >> $O/progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-runs > /dev/null 2> $O/fetch.err < /dev/null
echo "fetch done" >> $O/progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-runs > /dev/null 2> $O/write.err < /dev/null
echo "write done" >> $O/progress.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/busy -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sub-runs > /dev/null 2> $O/busy.err < /dev/null
echo "busy done" >> $O/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o p -- python3 $R/tools/bench_train.py 128 8 256 > $O/train_bench.json 2> $O/train_stats.err < /dev/null
echo "train stats done" >> $O/progress.txt
cd $R
python3 tools/summarize_trace.py $(ls $O/stats/*kernel_trace.csv | head -1) 1 > $O/kernel_summary.md 2> $O/summ.err < /dev/null
python3 tools/hbm_traffic.py $O/fetch $O/write > $O/hbm_traffic.json 2> $O/hbm.err < /dev/null
python3 tools/pmc_summary.py $O/busy > $O/pmc_busy.txt 2> $O/pmc.err < /dev/null
python3 tools/gpu_idle.py $O/stats > $O/gpu_idle.txt 2>&1 < /dev/null
ls -la $O | head -30
