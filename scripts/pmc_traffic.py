#!/usr/bin/env python3
"""profiles/r03_traffic_config3.json from a scripts/pmc_collect.sh summary: HBM-side bytes per launch of the kernel bench.py's
`roofline` prices (the fused launch k_trace_shadow_then_batch<false, false, false> on configs[3]), with the guide's gfx950 correction, stamped with the hash
of the sources it was collected from (bench.py quotes it only while that hash matches) and the commit label.
usage: pmc_traffic.py <gpurun_out/pmc_dir> <out.json> <source label> [kernel prefix]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402

prefix = sys.argv[4] if len(sys.argv) > 4 else "k_trace_shadow_then_batch<false, false, false>"
d = json.load(open(sys.argv[1] + "/summary.json"))
key = next(k for k in d if k.startswith(prefix))
v = d[key]
m = lambda c: v[c]["mean_per_launch"]  # noqa: E731
fetch_kb, write_kb = m("FETCH_SIZE"), m("WRITE_SIZE")
out = {
    "kernel": "rt::" + key,
    "workload": "scripts/pmc_run.py 4 (bench.py's configs[3]: 1 M-triangle atrium, 1920x1080, update_batch(4)); the fused launches (both shadow passes of bounce d + the closest-hit pass of bounce d + 1), averaged over the five of a frame",
    "launches_averaged": v["FETCH_SIZE"]["launches"],
    "FETCH_SIZE_KB_per_launch": fetch_kb,
    "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md §HBM; an upper bound here: "
                  "64-B node gathers are not wide reads); WRITE_SIZE taken as is; separate --pmc passes (scripts/pmc_collect.sh); both counters sit on the "
                  "L2's fabric side, i.e. Infinity-Cache hits are included",
    "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "l2_hit_rate": round(m("TCC_HIT_sum") / max(m("TCC_HIT_sum") + m("TCC_MISS_sum"), 1), 3),
    # one SIMD issues one VALU instruction per 4 cycles; 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8
    "valu_issue_utilisation": round(m("SQ_INSTS_VALU") / (1024 * (m("GRBM_GUI_ACTIVE") / 8) / 4), 3),
    "active_lanes_per_valu_instruction": round(m("SQ_THREAD_CYCLES_VALU") / m("SQ_INSTS_VALU"), 1),
    "wait_any_share_of_wave_cycles": round(m("SQ_WAIT_ANY") / max(m("SQ_WAVE_CYCLES"), 1), 3),
    "source": sys.argv[3],
    "source_hash": source_hash(),
    "commit": os.environ.get("HALART_COMMIT", "unknown"),
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
