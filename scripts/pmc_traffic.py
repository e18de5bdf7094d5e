#!/usr/bin/env python3
"""profiles/traffic_closest.json from a scripts/pmc_collect.sh summary: HBM bytes per launch of the kernel bench.py's
`roofline` prices (k_trace_batch<false, false, true> on configs[1]), with the guide's gfx950 correction.
usage: pmc_traffic.py <gpurun_out/pmc_dir> <out.json> <source label>"""
import json
import sys

d = json.load(open(sys.argv[1] + "/summary.json"))
key = next(k for k in d if k.startswith("k_trace_batch<false, false"))
v = d[key]
fetch_kb = v["FETCH_SIZE"]["mean_per_launch"]
write_kb = v["WRITE_SIZE"]["mean_per_launch"]
out = {
    "kernel": "rt::" + key,
    "workload": "scripts/pmc_run.py 2 (bench.py configs[1]: Cornell 1920x1080, update_batch(4)); bounce-ray launches only (depth >= 1)",
    "launches_averaged": v["FETCH_SIZE"]["launches"],
    "FETCH_SIZE_KB_per_launch": fetch_kb,
    "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md §HBM); "
                  "WRITE_SIZE taken as is; separate --pmc passes (scripts/pmc_collect.sh)",
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    # what actually bounds the kernel (the BVH of this scene lives in LDS): VALU issue slots.  One SIMD issues one VALU
    # instruction per 4 cycles; 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8
    "valu_issue_utilisation": round(v["SQ_INSTS_VALU"]["mean_per_launch"] / (1024 * (v["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8) / 4), 3),
    "active_lanes_per_valu_instruction": round(v["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / v["SQ_INSTS_VALU"]["mean_per_launch"], 1),
    "source": sys.argv[3],
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
