#!/bin/bash
# One call = the round's evidence: bench.py line, rocprofv3 kernel stats of the same command, PMC passes for configs[1]
# and the 1 M-triangle scene, per-config timings.  usage (on the GPU box): bash scripts/profile_round.sh <tag>
set -u
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "stats done"
cd $ROOT
bash scripts/pmc_collect.sh 2 4 $TAG/pmc_config2 > $OUT/pmc_config2.log 2>&1; echo "pmc2 done"
bash scripts/pmc_collect.sh 4 2 $TAG/pmc_config4 > $OUT/pmc_config4.log 2>&1; echo "pmc4 done"
python3 scripts/pmc_traffic.py $OUT/pmc_config2 $OUT/traffic_closest.json "profiles/${TAG}_pmc_config2.txt"
python3 scripts/bench_scenes.py --configs 2,3,4,5 > $OUT/scene_configs.txt 2> $OUT/scene_configs.err; echo "scenes done"
ls $OUT
