#!/bin/bash
# Kernel timeline of untimed frames (fused launches) of configs[3]: rocprofv3 --kernel-trace of a short bench run with the per-launch timing
# events off, then the launches of the LAST frame in order (start offset, duration).   usage: bash scripts/frame_timeline.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-timeline}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH_TIMING_PERIOD=1000000 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
lo, hi = idx[-3] + 1, idx[-2] + 1   # a frame from the middle of the timed region (the very last frames are the counting ones)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
print(f"frame: {(prev_end - t0) / 1e3:.1f} us")
PY
