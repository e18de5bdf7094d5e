#!/bin/bash
# Compile-time / env knob sweep on the GPU box (the box has hipcc and rebuilds libhalart.so in ~10 s):
#   gpurun -- 'bash scripts/variant_sweep.sh "-DRT_WAVES_PER_SIMD_STAGED=5" "-DRT_WORK_SHARDS=32" ...'
# prints Mrays/s and ms/frame of bench.py (40 steps) for the default build and for each EXTRA flag; the default is rebuilt at the end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'])"; }
build() { touch hala-renderer_amd/csrc/integrator.hip hala-renderer_amd/csrc/renderer.hip hala-renderer_amd/csrc/bvh_build.hip; make -C hala-renderer_amd/csrc -j8 EXTRA="$1" > gpurun_out/variant_make.log 2>&1 || { echo "build failed: $1"; tail -n 5 gpurun_out/variant_make.log; }; }
mkdir -p gpurun_out
echo "default"; run
for v in "$@"; do build "$v"; echo "$v"; run; done
build ""
