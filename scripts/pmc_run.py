#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: renders N frames of one config (no warm-up games) so that per-kernel
counter rows can be averaged.  usage: pmc_run.py <config:2|3|4> <frames>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import scenes  # noqa: E402

cfg, frames = sys.argv[1], int(sys.argv[2])
env = None
if cfg == "2":
    s, w, h = scenes.cornell_box(aspect=16 / 9), 1920, 1080
elif cfg == "3":
    s, w, h, env = scenes.bunny_class(subdivisions=6, disney=True), 1920, 1080, scenes.sky_sun_envmap(2048, 1024)
else:
    s, w, h, env = scenes.sponza_class(target_triangles=1_000_000), 1920, 1080, scenes.sky_sun_envmap(1024, 512, sun_gain=50.0)
r = H.HalaRenderer("pmc", w, h, 5, 3, False, False, False, 0)
if env is not None:
    r.set_envmap(env, 0.0)
r.set_scene(s)
r.commit()
spp = 16 if cfg == "3" else 4
for _ in range(frames):  # same launch shape as bench.py: one wavefront pass per frame
    r.reset_accumulation()
    r.update_batch(spp)
r.wait_idle()
print("rays", r.statistics().rays_total)
r.close()
