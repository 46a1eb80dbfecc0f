#!/bin/bash
# build_train_variant.sh NAME "-DFLAG=.. ..."  -> tools/exp/var/lib_NAME.so : the product library with cae_train.o rebuilt
# under extra flags (kernel A/B experiments; loaded through CAE_LIB)
set -e
cd "$(dirname "$0")/../../cnn_autoencoder_amd/csrc"
mkdir -p ../../tools/exp/var _obj_var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include $2 --offload-arch=gfx950 -c cae_train.hip -o _obj_var/cae_train_$1.o
objs=$(ls _obj/*.o | grep -v cae_train.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs _obj_var/cae_train_$1.o -L/opt/rocm/lib -lhsa-runtime64 -o ../../tools/exp/var/lib_$1.so
echo built tools/exp/var/lib_$1.so
