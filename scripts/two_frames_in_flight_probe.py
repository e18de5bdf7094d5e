#!/usr/bin/env python3
"""What would two frames in flight buy?  Two renderers (own streams, own wavefront buffers) on the same scene render their frames
alternately, so that frame k + 1 is enqueued while frame k runs and the tails of one's persistent launches can be filled by the other's
kernels; against one renderer rendering the same number of frames.  usage: two_frames_in_flight_probe.py [config 3|4] [world] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import workloads  # noqa: E402

index = int(sys.argv[1]) if len(sys.argv) > 1 else 3
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = workloads.baseline_config(index)


def make():
    r = H.HalaRenderer("probe", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0)
    if world > 1:
        r.set_tile_shard(0, world, 32)
    r.set_envmap(cfg["env"], 0.0)
    r.set_scene(cfg["scene"])
    r.commit()
    return r


def run(rs, n):
    for r in rs:
        for _ in range(2):
            r.reset_accumulation(); r.update_batch(cfg["spp"])
        r.wait_idle()
    t0 = time.perf_counter()
    for k in range(n):
        r = rs[k % len(rs)]
        r.reset_accumulation(); r.update_batch(cfg["spp"]); r.render()
    for r in rs:
        r.wait_idle()
    return (time.perf_counter() - t0) / n * 1e3


a, b = make(), make()
one = run([a], frames)
two = run([a, b], frames)
print(f"configs[{index}] world {world}: one renderer {one:.3f} ms per frame | two renderers alternating {two:.3f} ms per frame ({one / two:.3f}x)")
