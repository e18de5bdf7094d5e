#!/usr/bin/env python3
"""GPU builder (argv[1]: sah | ploc | lbvh; default: the library's choice) vs binned-SAH (oracle's CPU builder), both as compressed 4-wide trees: traversal steps per ray on
the same rays, and an instruction-weighted cost (a node visit ~ 170 VALU, a triangle test ~ 60)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hala_renderer_amd as H
from hala_renderer_amd import scenes
import oracle_lib as O
for name, s in [("blob82k", scenes.bunny_class(subdivisions=6)), ("atrium250k", scenes.sponza_class(target_triangles=250000, disney=False))]:
    r = H.HalaRenderer("q", 64, 64, 5, 3, False, False, False, 0)
    if len(sys.argv) > 1:
        r.set_build_options(builder=sys.argv[1])
    r.set_scene(s); r.commit()
    nodes, tris = r.download_bvh()
    osc = O.OracleScene(s)
    snodes, stris = osc.export_bvh4()
    rays = osc.camera_rays(480, 270, 0)
    hits = osc.trace(rays, 0)
    ok = hits["prim"] != 0xFFFFFFFF
    # incoherent secondary rays: from the hit points into random directions
    rng = np.random.RandomState(0)
    P = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok][:, None]
    d = rng.randn(len(P), 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.zeros(len(P), dtype=H._abi.RAY_DTYPE)
    sec["origin"] = P + d * 1e-3; sec["direction"] = d; sec["tmax"] = 3e38
    for label, rr in (("primary", rays), ("secondary", sec)):
        _, c_sah = O.trace_on_bvh(snodes, stris, rr, 0)
        _, c_lbvh = O.trace_on_bvh(nodes, tris, rr, 0)
        n = len(rr)
        cost = lambda c: (170 * c[0] + 60 * c[1]) / n
        print(f"{name} {label}: cost SAH {cost(c_sah):.0f} GPU {cost(c_lbvh):.0f} ratio {cost(c_lbvh) / cost(c_sah):.2f}")
        print(f"{name} {label}: SAH nodes/ray {c_sah[0]/n:.2f} tris/ray {c_sah[1]/n:.2f} | GPU-built nodes/ray {c_lbvh[0]/n:.2f} tris/ray {c_lbvh[1]/n:.2f} | oracle nodes {len(snodes)} gpu nodes {len(nodes)}")
    r.close()
