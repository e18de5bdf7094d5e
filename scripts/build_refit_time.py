#!/usr/bin/env python3
"""Times commit() (upload + BVH build) and refit() (node transforms -> instance records -> leaf boxes -> fit -> pack) on the
1M-triangle atrium; run by hand on the GPU box.  The numbers quoted in DESIGN.md section 5 come from here."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hala_renderer_amd as H
from hala_renderer_amd import scenes

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
s = scenes.sponza_class(target_triangles=tris)
r = H.HalaRenderer("time", 1920, 1080, 5, 3, True, True, False, 0)
r.set_scene(s)
t0 = time.perf_counter(); r.commit(); t1 = time.perf_counter()
print("triangles", r.bvh_info().triangle_count, "first commit ms %.2f" % ((t1 - t0) * 1e3), flush=True)
t0 = time.perf_counter(); r.commit(); t1 = time.perf_counter()
print("second commit ms %.2f" % ((t1 - t0) * 1e3), flush=True)
node = next(k for k, n in enumerate(s.nodes) if getattr(n, "mesh_index", 0xffffffff) != 0xffffffff)
m = np.array(s.nodes[node].local_transform, dtype=np.float32).reshape(4, 4).copy()
best = 1e9
for k in range(6):
    m[0, 3] += 0.01
    r.update_node_transform(node, m)
    t0 = time.perf_counter(); r.refit(); t1 = time.perf_counter()
    best = min(best, t1 - t0)
print("refit ms (best of 6) %.3f" % (best * 1e3))
r.update(); r.render()
r.close()
