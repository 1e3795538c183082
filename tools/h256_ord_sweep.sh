#!/bin/bash
# On the GPU box: time the H = 256 recurrent kernels of every ab/liblob_abl<k>.so (tools/h256_ablate.sh) interleaved twice (box drift), and run the H = 256 parity tests against each variant first.
# usage: tools/h256_ord_sweep.sh <outfile> <k> [<k> ...]
set -o pipefail
OUT=$1; shift
: > $OUT
for k in "$@"; do
  LOB_LIB_PATH=ab/liblob_abl$k.so timeout -k 10 300 python -m pytest tests/test_gpu_twins.py tests/test_gpu_training.py -m gpu -q -x -k "h256" > gpurun_out/ord_test_$k.log 2>&1
  echo "variant $k tests rc=$? $(tail -1 gpurun_out/ord_test_$k.log)" >> $OUT
done
for rep in 1 2; do
  echo "== rep $rep" >> $OUT
  for k in "$@"; do
    LOB_LIB_PATH=ab/liblob_abl$k.so timeout -k 10 120 python tools/rec_bench.py 256 >> $OUT 2>&1
  done
done
cat $OUT
