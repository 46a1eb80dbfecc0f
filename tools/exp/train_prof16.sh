#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof16
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/tools/bench_train.py 16 20 256 > $O/train_bench.json 2> $O/err.log < /dev/null
cd $R
f=$(ls $O/stats/*kernel_stats.csv | head -1)
head -22 $f | cut -c1-130
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot/1e6, "calls", calls)
PY
tail -1 $O/train_bench.json
