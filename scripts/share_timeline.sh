#!/bin/bash
# Kernel timeline of one frame of rank 0's share of configs[4] (3840x2160, 4 spp) for a given world size, fused launches, no timing events:
# where a strong-scaled rank's frame goes (launch durations and the gaps between them).   usage: bash scripts/share_timeline.sh <world> <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WORLD=${1:-8}
OUT=$ROOT/gpurun_out/${2:-share_timeline}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PROBE_WORLDS=$WORLD rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/scripts/shard_scaling_probe.py > $OUT/probe.txt 2> $OUT/probe.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
lo, hi = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:80]}")
    busy += e - s
    prev_end = e
print(f"frame: {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(prev_end - t0 - busy) / 1e3:.1f} us")
PY
