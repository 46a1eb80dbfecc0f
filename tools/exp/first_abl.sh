#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in first nostore; do
  for act in GDN none; do
    CAE_LIB=$PWD/tools/exp/var/lib_$v.so timeout -k 10 120 python tools/bench_layers.py --act $act < /dev/null > gpurun_out/fabl_${v}_$act.log 2>&1 || { tail -5 gpurun_out/fabl_${v}_$act.log; exit 1; }
    echo "$v $act: $(grep -E 'analysis.[01]' gpurun_out/fabl_${v}_$act.log | awk '{printf "%s ", $2}')"
  done
done
