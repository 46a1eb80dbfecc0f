#!/bin/bash
cd $GRAFT_REPO_ROOT
for bs in 16 128; do
for b in 32 64 128 192 256 384; do
  echo "blocks=$b b$bs: $(CAE_WG8_BLOCKS=$b timeout -k 10 100 python tools/bench_gg_train.py $bs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v['ms'] for k,v in d.items() if 'wgrad' in k})")"
done
done
