#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for b in 512 256 128 64; do
  echo "rep $rep old-kernel blocks=$b: $(CAE_WG_BLOCKS=$b timeout -k 10 100 python tools/bench_train.py 128 12 256 2>/dev/null | tail -1 | cut -c36-60)"
done
done
