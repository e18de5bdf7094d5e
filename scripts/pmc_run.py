#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: renders N frames of one BASELINE config (no warm-up games) so that per-kernel
counter rows can be averaged.  usage: pmc_run.py <config:2|3|4|5 = BASELINE configs[1..4]> <frames>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import workloads  # noqa: E402

cfg = workloads.baseline_config(int(sys.argv[1]) - 1)
frames = int(sys.argv[2])
r = H.HalaRenderer("pmc", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0)
if cfg["env"] is not None:
    r.set_envmap(cfg["env"], 0.0)
r.set_scene(cfg["scene"])
r.commit()
for _ in range(frames):  # same launch shape as bench.py: one wavefront pass per frame
    r.reset_accumulation()
    r.update_batch(cfg["spp"])
r.wait_idle()
print("rays", r.statistics().rays_total)
r.close()
