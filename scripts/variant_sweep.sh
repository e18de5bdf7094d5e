#!/bin/bash
# Compile-time / env knob sweep on the GPU box (the box has hipcc and rebuilds libhalart.so in ~20 s):
#   gpurun -- 'bash scripts/variant_sweep.sh "-DRT_STACK_LDS=6" "ENV:HALART_LEAF_MAX=4" ...'
# prints Mrays/s, ms/frame and the per-kernel split of bench.py (configs[3]) for the default build and for each variant
# (a compile flag rebuilds the library; "ENV:NAME=VALUE" runs the current build with that environment variable; "BOTH:<flags>|<NAME=VALUE ...>"
# does both); the default is rebuilt at the end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
STEPS=${SWEEP_STEPS:-20}
SEC=${SWEEP_SECONDARY:---no-secondary}
run() { env $1 timeout -k 10 300 python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline $SEC 2>/dev/null | python3 -c "
import sys,json
b=json.loads(sys.stdin.read()); r=b['roofline']; k=r['ms_per_frame_by_kernel']; u=k['one_launch_per_pass']; f=k['fused_launches']; s=r['simt']; ub=r['unfused_kernels']['batch']
print(b['value'], b['ms_per_step'], '| per pass: closest', u['closest'], 'shade', u['shade'], 'shadow', u['shadow'], '| fused frames: primary', f['primary'], 'shade', f['shade'], 'fused', f['fused_traversal'],
      '| nodes/tris per bounce ray', ub['nodes_per_ray'], ub['tris_per_ray'],
      '| leaf passes/step @ lanes', s['closest']['leaf_passes_per_wave_step'], s['closest']['leaf_path_lanes_of_64'], s['shadow']['leaf_passes_per_wave_step'], s['shadow']['leaf_path_lanes_of_64'],
      '| node lanes', s['closest']['node_path_lanes_of_64'], s['shadow']['node_path_lanes_of_64'])
c=b.get('secondary',{}).get('configs1')
if c: print('   cornell', c['value'], c['ms_per_frame'], 'batch', c['batch_kernel']['avg_launch_ms'], 'shadow', c['shadow_kernel']['avg_launch_ms'], 'shade', c['shade_kernel']['avg_launch_ms'], '| 4K on 1 GPU', b['secondary']['configs4_on_1_gpu']['ms_per_frame'])"; }
# the HALART_* environment knobs only exist in -DHALART_TUNING builds (kernels.h: tune_env): every build of the sweep is one, the release
# build is restored at the end
build() { touch hala-renderer_amd/csrc/integrator.hip hala-renderer_amd/csrc/renderer.hip hala-renderer_amd/csrc/bvh_build.hip; make -C hala-renderer_amd/csrc -j16 EXTRA="$1" > gpurun_out/variant_make.log 2>&1 || { echo "build failed: $1"; tail -n 5 gpurun_out/variant_make.log; }; }
mkdir -p gpurun_out
build "-DHALART_TUNING"
echo "default"; run ""
rebuilt=1
for v in "$@"; do
  case "$v" in
    ENV:*) echo "$v"; run "${v#ENV:}";;
    BOTH:*) w="${v#BOTH:}"; build "-DHALART_TUNING ${w%%|*}"; echo "$v"; run "${w#*|}";;
    *) build "-DHALART_TUNING $v"; echo "$v"; run "";;
  esac
done
if [ $rebuilt = 1 ]; then build ""; fi
