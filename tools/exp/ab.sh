#!/bin/bash
# A/B of library variants on one box: tools/exp/ab.sh <variant names...>  (base = in-tree library)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/tools/exp/var/lib_$v.so; [ $v = base ] && lib=$PWD/cnn_autoencoder_amd/libcae_hip.so
  CAE_LIB=$lib timeout -k 10 120 python tools/bench_layers.py < /dev/null > gpurun_out/ab_${v}.log 2>&1 || { tail -5 gpurun_out/ab_${v}.log; exit 1; }
  echo "$v: $(grep -E 'analysis|synthesis' gpurun_out/ab_${v}.log | awk '{printf "%s ", $2}') $(grep total gpurun_out/ab_${v}.log | sed 's/.*total main kernels//')"
done
done
if [ -n "$PARITY" ]; then
  CAE_LIB=$PWD/tools/exp/var/lib_$PARITY.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q < /dev/null > gpurun_out/ab_parity.log 2>&1; tail -3 gpurun_out/ab_parity.log
fi
