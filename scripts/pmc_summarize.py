#!/usr/bin/env python3
"""Aggregates the counter_collection CSVs of scripts/pmc_collect.sh: per kernel, per counter: mean per launch and total."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        short = name.split("(")[0].replace("void ", "").replace("rt::", "").replace("(anonymous namespace)::", "")
        c = row.get("Counter_Name")
        v = float(row.get("Counter_Value", 0) or 0)
        a = agg[short][c]
        a[0] += v
        a[1] += 1
out = {}
for k in sorted(agg):
    if not (k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_raygen") or k.startswith("k_resolve")):
        continue
    out[k] = {c: {"mean_per_launch": a[0] / max(a[1], 1), "launches": a[1]} for c, a in sorted(agg[k].items())}
    print(f"== {k}")
    for c, a in sorted(agg[k].items()):
        print(f"   {c:40s} mean/launch {a[0] / max(a[1], 1):16.1f}   launches {a[1]}")
    d = {c: a[0] / max(a[1], 1) for c, a in agg[k].items()}
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
        wc = d["SQ_WAVE_CYCLES"]
        print(f"   -> wave-cycle shares: wait_any {d.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst {d.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  "
              f"active_valu {d.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  active_vmem {d.get('SQ_ACTIVE_INST_VMEM', 0) / wc:.3f}  active_lds {d.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}")
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        pass
    if "FETCH_SIZE" in d:
        print(f"   -> FETCH_SIZE mean/launch {d['FETCH_SIZE']:.1f} KB (x2 for wide streaming reads on gfx950: guide §HBM)")
    if "TCC_HIT_sum" in d:
        print(f"   -> L2 hit rate {d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d.get('TCC_MISS_sum', 0), 1):.3f}")
    if d.get("TCP_TCC_READ_REQ_sum") and d.get("TCP_TCC_READ_REQ_LATENCY_sum"):
        print(f"   -> mean TCP->TCC read latency {d['TCP_TCC_READ_REQ_LATENCY_sum'] / d['TCP_TCC_READ_REQ_sum']:.1f} cycles")
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
