#!/usr/bin/env python3
"""Lifetime / leak / limits check (run by hand on the GPU box): many create-render-destroy cycles, max_depth at its
limit, ragged sizes, sharded and batched forms; device memory in use must return to where it started."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hala_renderer_amd as H
from hala_renderer_amd import scenes

def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

torch.zeros(1, device="cuda")
s_small, s_big = scenes.cornell_box(aspect=1.3), scenes.sponza_class(target_triangles=120000)
base = used()
for it in range(40):
    w, h = 97 + 13 * (it % 5), 61 + 7 * (it % 3)
    r = H.HalaRenderer("stress", w, h, 64 if it % 7 == 0 else 5, 3, False, False, False, 0)
    if it % 4 == 1:
        r.set_tile_shard(it % 3, 3, 16 if it % 2 else 32)
    r.set_scene(s_big if it % 5 == 0 else s_small)
    r.commit()
    if it % 3 == 0:
        r.update_batch(1 + it % 6)
    else:
        for _ in range(2):
            r.update()
    r.render()
    r.wait_idle()
    st = r.statistics()
    assert st.rays_total > 0 and np.isfinite(st.last_gpu_ms)
    r.close()
    if it % 10 == 9:
        print(f"cycle {it + 1}: device memory in use {used():.0f} MiB (start {base:.0f})", flush=True)
        if it == 9:
            steady = used()  # code objects, stream and event pools are in place after the first cycles
leak = used() - steady
print("growth over the last 30 cycles, MiB:", round(leak, 1))
assert leak < 8, "device memory keeps growing"
print("OK")
