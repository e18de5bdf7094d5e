#!/usr/bin/env python3
"""Parity soak (run by hand on the GPU box): random scenes (tests/random_scenes.py) rendered by libhalart.so and by the CPU oracle; prints the
seeds whose images or ray counts differ.   usage: python scripts/soak_random_scenes.py [--seeds N] [--first K]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hala_renderer_amd as H  # noqa: E402
import oracle_lib as O  # noqa: E402
from test_gpu_parity import render_random_scene_both  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--first", type=int, default=100)
ap.add_argument("--instances", action="store_true", help="objects referenced by several nodes + two-level trees on both sides (RENDER_SPEC 4.5)")
a = ap.parse_args()
failed, dark = [], 0
for seed in range(a.first, a.first + a.seeds):
    bad, rays_ok, lit = render_random_scene_both(H, O, seed, big=seed % 6 == 5, instances=a.instances)
    dark += lit < 1e-4
    if bad != [0, 0, 0, 0] or not rays_ok:
        failed.append(seed)
        print("seed", seed, "differs: pixels per image", bad, "ray counts equal:", rays_ok, flush=True)
    if (seed - a.first + 1) % 25 == 0:
        print(f"{seed - a.first + 1} scenes, {len(failed)} differing, {dark} black", flush=True)
print("differing seeds:", failed)
