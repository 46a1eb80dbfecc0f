#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in base nodma fewreads nostore all3 base; do
  lib=$PWD/tools/exp/var/lib_$v.so; [ $v = base ] && lib=$PWD/cnn_autoencoder_amd/libcae_hip.so
  for act in GDN none; do
    CAE_LIB=$lib timeout -k 10 120 python tools/bench_layers.py --act $act < /dev/null > gpurun_out/abl_${v}_$act.log 2>&1 || { tail -5 gpurun_out/abl_${v}_$act.log; exit 1; }
    echo "$v $act: $(grep -E 'synthesis.[012]' gpurun_out/abl_${v}_$act.log | awk '{printf "%s ", $2}')"
  done
done
