#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 512 256 128; do
  echo "old-kernel blocks=$b b128: $(CAE_WG_BLOCKS=$b timeout -k 10 100 python tools/bench_train.py 128 8 256 2>/dev/null | tail -1 | cut -c1-80)"
  echo "old-kernel blocks=$b b16: $(CAE_WG_BLOCKS=$b timeout -k 10 100 python tools/bench_train.py 16 20 256 2>/dev/null | tail -1 | cut -c1-80)"
done
