#!/usr/bin/env python3
"""Generates the frozen fixtures of tests/golden/ from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference repository holds no golden vectors, tests or fixtures
(SURVEY.md §4, §8c), so these are authored here; they pin the ORACLE against drift (the oracle itself is pinned to
the reference's source lines by the independent emulations in tests/test_oracle_host.py).

Fixtures are data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from hala_renderer_amd import scenes  # noqa: E402
from test_oracle_host import kat_images  # noqa: E402


def main():
    # (i) env-map KATs
    out = {}
    for name, img in kat_images().items():
        t, m, c = O.envmap_build_distribution(img)
        out[f"{name}_img"] = img
        out[f"{name}_total"] = np.float32(t)
        out[f"{name}_marginal"] = m
        out[f"{name}_conditional"] = c
    np.savez_compressed(os.path.join(HERE, "envmap_kat.npz"), **out)

    # (vi) Cornell-box primary-ray hit table, 16x16 grid of frame 0
    cb = O.OracleScene(scenes.cornell_box())
    rays = cb.camera_rays(16, 16, 0)
    hits = cb.trace(rays, 0)
    np.savez_compressed(os.path.join(HERE, "cornell_primary_hits_16x16.npz"), rays=rays, hits=hits)

    # (vii) oracle renders at 64x64: config 1/2 (Cornell, 2 spp), config 3 style (blob + env map IS, 2 spp)
    imgs, st = cb.render(64, 64, frames=2, max_depth=5, rr_depth=3)
    np.savez_compressed(os.path.join(HERE, "cornell_64x64_2spp.npz"), accum=imgs[0], albedo=imgs[1], normal=imgs[2],
                        rays=np.array([st.rays_closest, st.rays_shadow], dtype=np.uint64))
    env = scenes.sky_sun_envmap(64, 32)
    bs = O.OracleScene(scenes.bunny_class(subdivisions=2), envmap=env)
    imgs, st = bs.render(64, 36, frames=2, max_depth=4, rr_depth=2, env_rotation=30.0, env_intensity=1.5)
    np.savez_compressed(os.path.join(HERE, "blob_env_64x36_2spp.npz"), accum=imgs[0], albedo=imgs[1], normal=imgs[2], env=env,
                        rays=np.array([st.rays_closest, st.rays_shadow], dtype=np.uint64))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
