#!/bin/bash
# Collects the judged profile set (run on the GPU box through gpurun):
#   kernel stats of the default bench command, FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel-trace only) over
#   runs that contain NOTHING but training steps (so that the totals divide into bytes per step), SQ counters, the same for
#   the H = 256 step, and the bench JSON lines.
# usage: tools/collect_profiles.sh [tag]      (outputs under gpurun_out/<tag>/; tools/pmc_traffic.py / sq_counters.py turn
#                                              them into profiles/<tag>_*.csv here, after the call)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS="--no-extra --no-cpu-baseline --no-roofline"      # nothing but the steps under the profiler
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
# the default bench command under the tracer: its per-kernel averages are what roofline.sec_per_launch is compared with
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 7 --warmup 2 --no-extra --no-cpu-baseline > $O/stats.log 2>&1
echo stats done
for H in 128 256; do
  S=$([ $H = 128 ] && echo "" || echo "_h256")
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps$S -o run -- python3 $R/bench.py --hidden $H --steps 6 --warmup 2 $STEPS > $O/steps$S.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$S -o run -- python3 $R/bench.py --hidden $H --steps 3 --warmup 1 $STEPS > $O/fetch$S.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write$S -o run -- python3 $R/bench.py --hidden $H --steps 3 --warmup 1 $STEPS > $O/write$S.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq$S -o run -- python3 $R/bench.py --hidden $H --steps 3 --warmup 1 $STEPS > $O/sq$S.log 2>&1
  echo "H=$H done"
done
# the fp32 forward (configs[1] arithmetic) for its own kernel table
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fwd32 -o run -- python3 $R/bench.py --mode fwd --precision fp32 --steps 7 --warmup 2 --no-extra --no-cpu-baseline > $O/stats_fwd32.log 2>&1
# the fp32 training step (the fp16-split backward kernels of round 4)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train32 -o run -- python3 $R/bench.py --mode train --precision fp32 --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > $O/stats_train32.log 2>&1
# the mixed inference forward (the reference's GPU inference arithmetic, 06:349): what stops it -- kernel times and where its waves' cycles go
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fwdmix -o run -- python3 $R/bench.py --mode fwd --precision mixed --steps 8 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > $O/stats_fwdmix.log 2>&1
timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq_fwdmix -o run -- python3 $R/bench.py --mode fwd --precision mixed --steps 3 --warmup 1 $STEPS > $O/sq_fwdmix.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/sq2_fwdmix -o run -- python3 $R/bench.py --mode fwd --precision mixed --steps 3 --warmup 1 $STEPS > $O/sq2_fwdmix.log 2>&1 || true
cd $R
# keep only the small summaries (the traces are tens of MB)
find $O -name "*kernel_trace.csv" -delete
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --detail $O/b_default_detail.json > $O/b_default.json 2> $O/b_default.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --batch 1024 --no-cpu-baseline --no-extra --mode train > $O/b_train_b1024.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --batch 8192 --no-cpu-baseline --no-extra --mode train > $O/b_train_b8192.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --mode train --precision fp32 --no-cpu-baseline --no-extra > $O/b_train_fp32.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --mode fwd --precision mixed --no-extra --no-cpu-baseline > $O/b_fwd_mixed.json 2> /dev/null
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --hidden 256 --no-extra --no-cpu-baseline > $O/b_train_h256.json 2> /dev/null
python3 tools/latency_probe.py > $O/latency_probe.txt 2>&1
{ python3 tools/api_probe.py 3 300; python3 tools/api_probe.py 8 300; } 2>&1 | grep -v amdgpu.ids > $O/api_probe.txt
echo done > $O/done.txt
