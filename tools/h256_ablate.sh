#!/bin/bash
# (ABL_EXTRA: further compiler flags; ABL_TAG: suffix of the library name)
# Diagnostic builds of the H = 256 recurrent kernels (LOB_ABL_H256, see lstm_rec_h256_bf16.hip): ab/liblob_abl<k>.so =
# the current objects with that one source recompiled.  Run here (hipcc cross-compiles), then on the GPU box:
#   LOB_LIB_PATH=ab/liblob_abl1.so python tools/rec_bench.py 256
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab
OBJS=$(ls $R/lstm_ode_bci_amd/build/*.o | grep -v lstm_rec_h256_bf16.o)
SRC=${ABL_SRC:-lstm_rec_h256_bf16}
DEF=${ABL_DEF:-LOB_ABL_H256}
OBJS=$(ls $R/lstm_ode_bci_amd/build/*.o | grep -v $SRC.o)
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -D$DEF=$k '-DLOB_BUILD_ID="ablation"' -I $R/include \
      $ABL_EXTRA -I $R/lstm_ode_bci_amd/csrc -c $R/lstm_ode_bci_amd/csrc/$SRC.hip -o $R/ab/h256_abl$k.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/liblob_abl$k$ABL_TAG.so $OBJS $R/ab/h256_abl$k.o
  rm $R/ab/h256_abl$k.o
  echo built ab/liblob_abl$k$ABL_TAG.so
done
