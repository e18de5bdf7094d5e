#!/usr/bin/env python3
"""Per-config timing of the hot path on one GPU (BASELINE.json configs 2-4 + the 4K frame of config 5 on one rank).
Prints one JSON line per config: Mrays/s, ms/frame, traversal-kernel time share, nodes/triangles per ray, BVH build ms."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import hala_renderer_amd as H  # noqa: E402
from hala_renderer_amd import scenes  # noqa: E402


def run(name, scene, w, h, spp, env=None, max_depth=5, rr_depth=3, steps=8):
    t0 = time.perf_counter()
    r = H.HalaRenderer(name, w, h, max_depth, rr_depth, False, False, False, 0)
    if env is not None:
        r.set_envmap(env, 0.0)
    r.set_scene(scene)
    t1 = time.perf_counter()
    r.commit()
    t2 = time.perf_counter()
    r.set_launch_timing_period(1)  # every update timed: one launch per pass (the per-kernel columns below)
    info = r.bvh_info()
    for _ in range(spp):  # warm-up frame
        r.update()
    r.wait_idle()
    s0 = r.statistics()
    t3 = time.perf_counter()
    for _ in range(steps):
        r.reset_accumulation()
        if os.environ.get("HALART_NO_BATCH"):
            for _ in range(spp):
                r.update()
        else:
            r.update_batch(spp)
    r.wait_idle()
    dt = time.perf_counter() - t3
    s1 = r.statistics()
    r.set_counting(True)
    c0 = r.statistics()
    r.reset_accumulation()
    r.update()
    r.wait_idle()
    c1 = r.statistics()
    r.set_counting(False)
    rays = s1.rays_total - s0.rays_total
    nc = max(c1.rays_closest_counted - c0.rays_closest_counted, 1)
    ns = max(c1.rays_shadow_counted - c0.rays_shadow_counted, 1)
    out = {
        "config": name, "triangles": info.triangle_count, "nodes": info.node_count, "bvh_depth": info.max_depth, "lds_nodes": info.lds_node_count,
        "resolution": [w, h], "spp": spp, "mrays_per_s": round(rays / dt / 1e6, 1), "ms_per_frame": round(dt / steps * 1e3, 3),
        "rays_per_frame": int(rays / steps), "set_scene_ms": round((t1 - t0) * 1e3, 1), "commit_ms(bvh build)": round((t2 - t1) * 1e3, 1),
        "closest_ms_per_frame": round((s1.traverse_closest_ms_total - s0.traverse_closest_ms_total) / steps, 3),
        "shadow_ms_per_frame": round((s1.traverse_shadow_ms_total - s0.traverse_shadow_ms_total) / steps, 3),
        "gpu_ms_per_frame": round((s1.gpu_ms_total - s0.gpu_ms_total) / steps, 3),
        "closest_grays_in_kernel": round((s1.rays_closest_total - s0.rays_closest_total) / max(s1.traverse_closest_ms_total - s0.traverse_closest_ms_total, 1e-9) / 1e6, 3),
        "shadow_grays_in_kernel": round((s1.rays_shadow_total - s0.rays_shadow_total) / max(s1.traverse_shadow_ms_total - s0.traverse_shadow_ms_total, 1e-9) / 1e6, 3),
        "closest_nodes_per_ray": round((c1.nodes_closest_total - c0.nodes_closest_total) / nc, 2),
        "closest_tris_per_ray": round((c1.tris_closest_total - c0.tris_closest_total) / nc, 2),
        "shadow_nodes_per_ray": round((c1.nodes_shadow_total - c0.nodes_shadow_total) / ns, 2),
        "shadow_tris_per_ray": round((c1.tris_shadow_total - c0.tris_shadow_total) / ns, 2),
    }
    for kind in ("closest", "shadow"):  # SIMT utilisation of the two code paths of the traversal kernels (counting pass)
        ws = getattr(c1, f"wave_steps_{kind}_total") - getattr(c0, f"wave_steps_{kind}_total")
        lp = getattr(c1, f"leaf_passes_{kind}_total") - getattr(c0, f"leaf_passes_{kind}_total")
        ll = getattr(c1, f"leaf_lanes_{kind}_total") - getattr(c0, f"leaf_lanes_{kind}_total")
        nn = getattr(c1, f"nodes_{kind}_total") - getattr(c0, f"nodes_{kind}_total")
        out[f"{kind}_simt"] = {"node_path_lanes": round(nn / max(ws, 1), 1), "leaf_passes_per_wave_step": round(lp / max(ws, 1), 2),
                               "leaf_path_lanes": round(ll / max(lp, 1), 1)}
    out["closest_alg_GBps"] = round((48 + 64 * out["closest_nodes_per_ray"] + 48 * out["closest_tris_per_ray"]) * out["closest_grays_in_kernel"], 1)
    img = r.read_image(0)
    out["mean_radiance"] = round(float(img[..., :3].mean()), 4)
    r.close()
    print(json.dumps(out), flush=True)
    return img


def main():
    from hala_renderer_amd import workloads
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4", help="1-based: 2..5 = BASELINE configs[1..4]")
    ap.add_argument("--save", default="")
    args = ap.parse_args()
    want = set(args.configs.split(","))
    names = {"2": "config2_cornell_1080p_4spp", "3": "config3_blob82k_env_1080p_16spp", "4": "config4_atrium1M_1080p_4spp", "5": "config5_atrium1M_4K_4spp_1gpu"}
    steps = {"2": 8, "3": 2, "4": 2, "5": 1}
    shared = None
    for k in ("2", "3", "4", "5"):
        if k not in want:
            continue
        c = workloads.baseline_config(int(k) - 1)
        if k in ("4", "5"):  # the same scene objects for both
            if shared is None:
                shared = (c["scene"], c["env"])
            c["scene"], c["env"] = shared
        img = run(names[k], c["scene"], c["width"], c["height"], c["spp"], env=c["env"], steps=steps[k])
        if args.save and k in ("3", "4"):
            np.save(os.path.join(args.save, f"config{k}.npy"), img[::2, ::2, :3].astype(np.float16))


if __name__ == "__main__":
    main()
