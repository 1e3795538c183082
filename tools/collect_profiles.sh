#!/bin/bash
# Collects the judged profile set of the default bench (run on the GPU box through gpurun):
#   kernel stats, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel-trace only), bench JSON lines.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 7 --warmup 2 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
cd $R
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > $O/b_train.json 2> $O/b_train.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --batch 1024 --no-cpu-baseline > $O/b_train_b1024.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --batch 8192 --no-cpu-baseline > $O/b_train_b8192.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --hidden 256 --no-cpu-baseline > $O/b_train_h256.json 2> /dev/null
echo done > $O/done.txt
