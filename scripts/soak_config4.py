#!/usr/bin/env python3
"""One-off soak (run by hand on the GPU box): BASELINE configs[3] at 1920x1080, seven frames in mixed batch sizes, ACES tonemap,
rotated env map, against the oracle walking (a) the product's tree and (b) its own; prints the number of differing pixels."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hala_renderer_amd as H
from hala_renderer_amd import scenes
import oracle_lib as O
s = scenes.sponza_class(target_triangles=1_000_000)
scenes.attach_textures(s, sets=6, size=1024)
env = scenes.sky_sun_envmap(1024, 512, sun_gain=50.0)
r = H.HalaRenderer("soak", 1920, 1080, 5, 3, True, True, False, 0)
r.set_envmap(env, 40.0); r.set_scene(s); r.commit()
osc = O.OracleScene(s, envmap=env)
for own_tree in (False, True):
    if not own_tree:
        osc.use_bvh(*r.download_bvh())
    else:
        osc.use_bvh(None)
    r.reset_accumulation()
    r.update_batch(4); r.update(); r.update_batch(2); r.render()
    imgs, st = osc.render(1920, 1080, frames=7, tonemap=(True, True, False), env_rotation=40.0)
    for k, name in enumerate(("accum", "albedo", "normal", "final")):
        got = r.read_image(k)
        bad = np.any(got != imgs[k], axis=-1)
        print("oracle tree" if own_tree else "shared tree", name, "differing pixels:", int(bad.sum()), flush=True)
    stg = r.statistics()
    print("rays", stg.rays_closest_total, st.rays_closest, stg.rays_shadow_total, st.rays_shadow)
r.close()
