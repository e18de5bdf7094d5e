#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: rank 0's share of configs[4] (3840x2160, 4 spp, 1 M triangles) for world = 1, 2, 4, 8 — what
each rank of an N-GPU run computes per frame, without the collective.  ideal = t(1) / N; the shortfall is per-launch fixed cost
(launch gaps, the latency-bound tails of small persistent launches, load imbalance between tile sets)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import workloads  # noqa: E402

cfg = workloads.baseline_config(4)
period = int(os.environ.get("PROBE_TIMING_PERIOD", "0"))
base = None
for world in [int(w) for w in os.environ.get("PROBE_WORLDS", "1,2,4,8").split(",")]:
    per_rank = []
    for rank in ([0] if world == 1 else [0, world - 1]):
        r = H.HalaRenderer("probe", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0)
        if world > 1:
            r.set_tile_shard(rank, world, int(os.environ.get("PROBE_TILE", "32")))
        r.set_envmap(cfg["env"], 0.0)
        r.set_scene(cfg["scene"])
        r.commit()
        r.set_launch_timing_period(period)
        for _ in range(2):
            r.reset_accumulation(); r.update_batch(cfg["spp"]); r.render()
        r.wait_idle()
        steps = 6
        s0 = r.statistics()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.reset_accumulation(); r.update_batch(cfg["spp"]); r.render()
        r.wait_idle()
        dt = (time.perf_counter() - t0) / steps
        s1 = r.statistics()
        per_rank.append((dt * 1e3, (s1.rays_total - s0.rays_total) / steps))
        r.close()
    ms = max(p[0] for p in per_rank)
    if base is None:
        base = ms
    print(json.dumps({"world": world, "ms_per_frame_slowest_probed_rank": round(ms, 3), "ideal_ms": round(base / world, 3),
                      "compute_scaling_efficiency": round(base / world / ms, 3), "rays_per_rank": [int(p[1]) for p in per_rank]}), flush=True)
