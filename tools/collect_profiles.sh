#!/bin/bash
# Collects the judged profile set of the default bench (run on the GPU box through gpurun):
#   kernel stats, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel-trace only), SQ counters, bench JSON lines.
# usage: tools/collect_profiles.sh [tag]      (outputs under gpurun_out/<tag>/)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-final}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-extra --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 7 --warmup 2 $B > $O/stats.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- python3 $R/bench.py --steps 3 --warmup 1 $B > $O/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- python3 $R/bench.py --steps 3 --warmup 1 $B > $O/write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -o run -- python3 $R/bench.py --steps 3 --warmup 1 $B > $O/sq.log 2>&1
echo sq done
# the fp32 forward (configs[1] arithmetic) for its own kernel table
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fwd32 -o run -- python3 $R/bench.py --mode fwd --precision fp32 --steps 7 --warmup 2 $B > $O/stats_fwd32.log 2>&1
cd $R
# keep only the small summaries (the traces are tens of MB)
for d in stats fetch write sq stats_fwd32; do find $O/$d -name "*kernel_trace.csv" -delete; done
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > $O/b_default.json 2> $O/b_default.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --batch 1024 --no-cpu-baseline --mode train > $O/b_train_b1024.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --batch 8192 --no-cpu-baseline --mode train > $O/b_train_b8192.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --mode train --precision fp32 --no-cpu-baseline > $O/b_train_fp32.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --mode fwd --precision fp32 --no-extra --no-cpu-baseline > $O/b_fwd_fp32.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --mode fwd --precision mixed --no-extra --no-cpu-baseline > $O/b_fwd_mixed.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --mode coupled --precision fp32 --no-extra --no-cpu-baseline > $O/b_coupled_fp32.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --hidden 256 --no-extra --no-cpu-baseline > $O/b_train_h256.json 2> /dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_h256 -o run -- python3 $R/bench.py --hidden 256 --steps 5 --warmup 2 $B > $O/stats_h256.log 2>&1
find $O/stats_h256 -name "*kernel_trace.csv" -delete
cd $R
python3 tools/latency_probe.py > $O/latency_probe.txt 2>&1
echo done > $O/done.txt
