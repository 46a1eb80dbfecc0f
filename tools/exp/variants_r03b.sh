#!/bin/bash
# f16x3 vs fp32 kernels for residual / multiscale variants of the canonical shape (batch 16, 1024^2 tiles)
cd $GRAFT_REPO_ROOT
for a in "--act GDN --residual" "--act LeakyReLU --residual" "--act GDN --multiscale" "--act none --residual"; do
  timeout -k 10 200 python tools/bench_variants.py $a --batch 16 || exit 1
done
