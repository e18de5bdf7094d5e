#!/usr/bin/env python3
"""Looks for rays whose closest hit differs between the GPU traversal (compressed BVH4 built on the GPU) and the oracle's
own BVH2 on the 1 M-triangle atrium, and says which one brute force agrees with."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hala_renderer_amd as H
from hala_renderer_amd import scenes
import oracle_lib as O

which = sys.argv[2] if len(sys.argv) > 2 else "atrium"
s = {"atrium": lambda: scenes.sponza_class(target_triangles=1_000_000, disney=False), "blob": lambda: scenes.bunny_class(subdivisions=6),
     "cornell": lambda: scenes.cornell_box(aspect=16 / 9), "sheets": lambda: scenes.stacked_sheets(4096)}[which]()
r = H.HalaRenderer("hunt", 64, 64, 5, 3, False, False, False, 0)
r.set_scene(s); r.commit()
osc = O.OracleScene(s)
nodes, tris = r.download_bvh()
rng = np.random.RandomState(7)
total = bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rays = osc.camera_rays(1920, 1080, it)
    hits = osc.trace(rays, 0)
    ok = hits["prim"] != 0xFFFFFFFF
    P = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok][:, None]
    d = rng.randn(len(P), 3).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.zeros(len(P), dtype=H._abi.RAY_DTYPE)
    sec["origin"] = P + d * np.float32(1e-3); sec["direction"] = d; sec["tmax"] = 3.0e38
    for label, rr, want in (("primary", rays, hits), ("secondary", sec, None)):
        if want is None:
            want = osc.trace(rr, 0)
        got = r.trace_rays_host(rr, 0)
        diff = np.nonzero((got["prim"] != want["prim"]) | (got["t"] != want["t"]))[0]
        total += len(rr); bad += len(diff)
        # any-hit: segments that end just short of / just beyond the closest hit must be free / occluded
        hitm = want["prim"] != 0xFFFFFFFF
        seg = rr[hitm].copy()
        seg["tmax"] = want["t"][hitm] * np.float32(1.001)
        occ = r.trace_rays_host(seg, 1)["t"] > 0
        ooc = osc.trace(seg, 1)["t"] > 0
        bad_any = int((occ != ooc).sum())
        total += len(seg); bad += bad_any
        if bad_any:
            print(label, "any-hit disagreements:", bad_any, "of", len(seg), "(gpu occluded", int(occ.sum()), "oracle", int(ooc.sum()), ")", flush=True)
        for i in diff[:5]:
            bf = osc.trace(rr[i:i + 1], 0, brute=True)[0]
            on_gpu_tree = O.trace_on_bvh(nodes, tris, rr[i:i + 1], 0)[0][0]
            print(label, "ray", i, "gpu", got[i], "oracle", want[i], "brute", bf, "oracle-on-gpu-tree", on_gpu_tree, "ray:", rr[i], flush=True)
print("rays", total, "mismatches", bad)
r.close()
