import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, ctypes as C
import hala_renderer_amd as H
from hala_renderer_amd import scenes
import oracle_lib as O
s = scenes.bunny_class(subdivisions=6, disney=True)
r = H.HalaRenderer("t", 64, 64, 5, 3, False, False, False, 0)
r.set_scene(s); r.commit()
nodes, tris = r.download_bvh()
osc = O.OracleScene(s)
rays = osc.camera_rays(480, 270, 0)
# per-ray counts via the oracle on the GPU-built BVH, one ray at a time for a sample
idx = np.random.RandomState(0).choice(len(rays), 4000, replace=False)
cnt = []
for i in idx:
    h, c = O.trace_on_bvh(nodes, tris, rays[i:i+1], 0)
    cnt.append(c[0])
cnt = np.array(cnt)
print("camera rays: mean nodes", cnt.mean(), "p50", np.percentile(cnt,50), "p99", np.percentile(cnt,99), "max", cnt.max())
# shadow-like rays: from ground/blob hit points toward the sun
hits = osc.trace(rays, 0)
ok = hits["prim"] != 0xFFFFFFFF
P = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok][:, None]
sun = np.array([0.4, 0.6, 0.35]); sun /= np.linalg.norm(sun)
sr = np.zeros(len(P), dtype=H._abi.RAY_DTYPE)
sr["origin"] = P + sun * 1e-3; sr["direction"] = sun; sr["tmax"] = 3e38
idx = np.random.RandomState(1).choice(len(sr), 4000, replace=False)
cnt = []
for i in idx:
    h, c = O.trace_on_bvh(nodes, tris, sr[i:i+1], 1)
    cnt.append(c[0])
cnt = np.array(cnt)
print("sun shadow rays: mean nodes", cnt.mean(), "p50", np.percentile(cnt,50), "p99", np.percentile(cnt,99), "max", cnt.max())
w = np.argsort(cnt)[-5:]
for k in w: print("  worst", cnt[k], sr[idx[k]]["origin"])
