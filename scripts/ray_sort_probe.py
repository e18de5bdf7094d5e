#!/usr/bin/env python3
"""How much would ordering the bounce rays buy?  Traces the same set of incoherent secondary rays (diffuse bounce off the
primary hits of the 1 M-triangle atrium) through hala_rt_trace_rays in (a) path order, (b) sorted by direction octant and a
Morton code of the origin, (c) random order; prints Grays/s in kernel (torch events on the stream the kernel runs on)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hala_renderer_amd as H
from hala_renderer_amd import scenes

s = scenes.sponza_class(target_triangles=1_000_000, disney=False)
r = H.HalaRenderer("probe", 64, 64, 5, 3, False, False, False, 0)
r.set_scene(s); r.commit()
info = r.bvh_info()
W, Hh = 1920, 1080
# primary rays on the host from the packed camera: simple pinhole through pixel centres (the exact camera model does not matter here)
cam = r.packed_cameras()[0]
pos, right, up, fwd = (np.array(list(getattr(cam, k)), dtype=np.float32) for k in ("position", "right", "up", "forward"))
th = np.tan(0.5 * cam.yfov)
ys, xs = np.meshgrid(np.arange(Hh), np.arange(W), indexing="ij")
u = ((xs + 0.5) / W * 2 - 1) * th * (W / Hh); v = (1 - (ys + 0.5) / Hh * 2) * th
d = fwd[None, None] + right[None, None] * u[..., None] + up[None, None] * v[..., None]
d /= np.linalg.norm(d, axis=-1, keepdims=True)
rays = np.zeros(W * Hh, dtype=H._abi.RAY_DTYPE)
rays["origin"] = pos; rays["direction"] = d.reshape(-1, 3); rays["tmax"] = 3e38
hits = r.trace_rays_host(rays, 0)
ok = hits["prim"] != 0xFFFFFFFF
P = rays["origin"][ok] + rays["direction"][ok] * hits["t"][ok][:, None]
rng = np.random.RandomState(1)
dd = rng.randn(len(P), 3).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
sec = np.zeros(len(P), dtype=H._abi.RAY_DTYPE)
sec["origin"] = P + dd * 1e-2; sec["direction"] = dd; sec["tmax"] = 3e38
sec = np.concatenate([sec] * 3)  # ~6 M rays, like one bounce launch of a 4-spp batch
mn, mx = np.array(list(info.scene_min)), np.array(list(info.scene_max))
q = np.clip(((sec["origin"] - mn) / (mx - mn) * 1024).astype(np.int64), 0, 1023)
def spread(x):
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
    return x
morton = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
octant = (sec["direction"][:, 0] < 0) * 4 + (sec["direction"][:, 1] < 0) * 2 + (sec["direction"][:, 2] < 0)
orders = {"path order": np.arange(len(sec)), "octant + origin morton(30b)": np.lexsort((morton, octant)),
          "octant + origin morton(15b)": np.lexsort((morton >> 15, octant)), "origin morton only": np.argsort(morton, kind="stable"),
          "random": rng.permutation(len(sec))}
d_hits = torch.zeros(len(sec) * 16, dtype=torch.uint8, device="cuda")
for name, order in orders.items():
    d_rays = torch.from_numpy(sec[order].view(np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()  # a real stream handle: handle 0 would select the renderer's own stream
    torch.cuda.set_stream(side)
    st = side.cuda_stream
    for _ in range(2):
        r.trace_rays(d_rays.data_ptr(), d_hits.data_ptr(), len(sec), 0, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r.trace_rays(d_rays.data_ptr(), d_hits.data_ptr(), len(sec), 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:30s} {len(sec) / ms / 1e6:7.3f} Grays/s  ({ms:.3f} ms for {len(sec)} rays)", flush=True)
r.close()
