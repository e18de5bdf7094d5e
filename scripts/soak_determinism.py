#!/usr/bin/env python3
"""Determinism soak (run by hand on the GPU box): BASELINE configs[3] rendered `--frames` times — every third frame with per-launch timing
events (one launch per pass), the others with the fused shadow + closest-hit launches; every frame's accumulated image must hash the same.
The wave-cooperative leaf pass merges hits through LDS atomics and relies on a wave's LDS operations executing in program order: a race
there would show as a rare differing pixel among the ~1.9 G rays of 40 frames."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--two-level", action="store_true", help="hala_rt_build_options::instancing = 2 (RENDER_SPEC 4.5)")
args = ap.parse_args()
cfg = workloads.baseline_config(args.config)
r = H.HalaRenderer("soak", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0)
if cfg["env"] is not None:
    r.set_envmap(cfg["env"], 0.0)
r.set_scene(cfg["scene"])
if args.two_level:
    r.set_build_options(instancing=True)
r.commit()
r.set_launch_timing_period(3)
hashes = {}
t0 = time.perf_counter()
rays = 0
for k in range(args.frames):
    s0 = r.statistics().rays_total
    r.reset_accumulation()
    if k % 5 == 4:  # the same 4 samples as single updates (no batching)
        for _ in range(cfg["spp"]):
            r.update()
    else:
        r.update_batch(cfg["spp"])
    r.render()
    h = hashlib.sha256(r.read_image(0).tobytes()).hexdigest()
    hashes[h] = hashes.get(h, 0) + 1
    rays += r.statistics().rays_total - s0
    if k % 10 == 9:
        print(json.dumps({"frames": k + 1, "distinct_images": len(hashes), "rays": int(rays), "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
r.close()
print(json.dumps({"config": cfg["name"], "frames": args.frames, "distinct_images": len(hashes), "rays": int(rays), "hashes": {k[:16]: v for k, v in hashes.items()}}))
sys.exit(0 if len(hashes) == 1 else 1)
