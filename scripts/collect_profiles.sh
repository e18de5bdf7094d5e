#!/bin/bash
# copies the judged summaries of a scripts/profile_round.sh run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked)
# usage: HALART_COMMIT=<label> bash scripts/collect_profiles.sh <tag>
set -eu
TAG=$1
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG
cp $S/bench.json profiles/${TAG}_bench.json
cp $S/stats/bench_kernel_stats.csv profiles/${TAG}_kernel_stats_bench.csv
cp $S/pmc_config4/summary.txt profiles/${TAG}_pmc_config4.txt
[ -f $S/pmc_config2/summary.txt ] && cp $S/pmc_config2/summary.txt profiles/${TAG}_pmc_config2.txt
[ -f $S/scene_configs.txt ] && cp $S/scene_configs.txt profiles/${TAG}_scene_configs.txt
python3 scripts/pmc_traffic.py $S/pmc_config4 profiles/r03_traffic_config3.json "profiles/${TAG}_pmc_config4.txt" > /dev/null
ls -la profiles | grep $TAG
