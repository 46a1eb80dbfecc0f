#!/bin/bash
# rocprofv3 kernel stats of the batch-128 training step -> gpurun_out/train_prof/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/tools/bench_train.py 128 8 256 > $O/train_bench.json 2> $O/err.log < /dev/null
cd $R
f=$(ls $O/stats/*kernel_stats.csv | head -1)
head -14 $f | cut -c1-150
tail -1 $O/train_bench.json
