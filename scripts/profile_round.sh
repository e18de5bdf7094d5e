#!/bin/bash
# One call = the round's evidence: bench.py line, rocprofv3 kernel stats of the same command, PMC passes for configs[3]
# (bench.py's headline) and configs[1], per-config timings.
# usage (on the GPU box): HALART_COMMIT=<label> bash scripts/profile_round.sh <tag> [quick]
set -u
TAG=${1:-r03_x}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench done: $(head -c 300 $OUT/bench.json)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "stats done"
cd $ROOT
bash scripts/pmc_collect.sh 4 2 $TAG/pmc_config4 > $OUT/pmc_config4.log 2>&1; echo "pmc config4 (configs[3]) done"
python3 scripts/pmc_traffic.py $OUT/pmc_config4 $OUT/traffic_config3.json "profiles/${TAG}_pmc_config4.txt"
if [ -z "$QUICK" ]; then
  bash scripts/pmc_collect.sh 2 4 $TAG/pmc_config2 > $OUT/pmc_config2.log 2>&1; echo "pmc config2 (configs[1]) done"
  python3 scripts/bench_scenes.py --configs 2,3,4,5 > $OUT/scene_configs.txt 2> $OUT/scene_configs.err; echo "scenes done"
fi
ls $OUT
