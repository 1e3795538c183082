#!/bin/bash
# After tools/collect_profiles.sh <tag> ran on the GPU box (outputs merged into gpurun_out/<tag>/): turn them into the
# tracked files under profiles/ -- kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE passes -> profiles/pmc_traffic.json,
# stamped with this tree's build id: run BEFORE touching csrc/ again), SQ counters, the bench lines.
# usage: tools/postprocess_profiles.sh <tag>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:?tag}
O=$R/gpurun_out/$TAG
P=$R/profiles
cp $O/stats/run_kernel_stats.csv $P/${TAG}_default_bench_kernel_stats.csv
cp $O/steps/run_kernel_stats.csv $P/${TAG}_train_mixed_B4096_kernel_stats.csv
cp $O/steps_h256/run_kernel_stats.csv $P/${TAG}_train_mixed_B4096_H256_kernel_stats.csv
cp $O/stats_fwd32/run_kernel_stats.csv $P/${TAG}_fwd_fp32_B4096_kernel_stats.csv
[ -f $O/stats_train32/run_kernel_stats.csv ] && cp $O/stats_train32/run_kernel_stats.csv $P/${TAG}_train_fp32_B4096_kernel_stats.csv
python3 $R/tools/pmc_traffic.py $O/fetch/run_counter_collection.csv $O/write/run_counter_collection.csv ${TAG}_train_mixed_B4096 4096 128 4 mixed > /dev/null
python3 $R/tools/pmc_traffic.py $O/fetch_h256/run_counter_collection.csv $O/write_h256/run_counter_collection.csv ${TAG}_train_mixed_B4096_H256 4096 256 4 mixed > /dev/null
python3 $R/tools/sq_counters.py $O/sq/run_counter_collection.csv $P/${TAG}_train_mixed_B4096_sq_counters.csv > /dev/null
python3 $R/tools/sq_counters.py $O/sq_h256/run_counter_collection.csv $P/${TAG}_train_mixed_B4096_H256_sq_counters.csv > /dev/null
[ -f $O/stats_fwdmix/run_kernel_stats.csv ] && cp $O/stats_fwdmix/run_kernel_stats.csv $P/${TAG}_fwd_mixed_B4096_kernel_stats.csv
[ -f $O/sq_fwdmix/run_counter_collection.csv ] && python3 $R/tools/sq_counters.py $O/sq_fwdmix/run_counter_collection.csv $P/${TAG}_fwd_mixed_B4096_sq_counters.csv > /dev/null
cp $O/b_default.json $P/${TAG}_default_bench.json
cp $O/b_default_detail.json $P/${TAG}_default_bench_detail.json
cp $O/b_train_b1024.json $P/${TAG}_train_mixed_B1024_bench.json
cp $O/b_train_b8192.json $P/${TAG}_train_mixed_B8192_bench.json
cp $O/b_train_fp32.json $P/${TAG}_train_fp32_B4096_bench.json
cp $O/b_fwd_mixed.json $P/${TAG}_fwd_mixed_B4096_bench.json
cp $O/b_train_h256.json $P/${TAG}_train_mixed_B4096_H256_bench.json
cp $O/latency_probe.txt $P/${TAG}_latency_probe.txt
[ -f $O/api_probe.txt ] && cp $O/api_probe.txt $P/${TAG}_api_probe.txt
echo "profiles/${TAG}_* written; pmc_traffic.json:"; cat $P/pmc_traffic.json
