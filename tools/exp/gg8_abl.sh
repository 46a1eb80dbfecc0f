#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base nomfma nodma both; do
  lib=$PWD/tools/exp/var/lib_$v.so; [ $v = base ] && lib=$PWD/cnn_autoencoder_amd/libcae_hip.so
  echo "$v: $(CAE_LIB=$lib timeout -k 10 100 python tools/bench_gg_train.py 128 2>/dev/null)"
done
for npb in 1 2 8 16; do
  echo "npb=$npb: $(CAE_GG8_NPB=$npb timeout -k 10 100 python tools/bench_gg_train.py 128 2>/dev/null)"
done
