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


def media_scene():
    """RENDER_SPEC 7.1c-f in one frame: a glass blob (TIR-aware transmission) filled with an absorbing medium, a second, invisible
    (opacity 0) ball of forward-scattering fog, a cut-out (opacity 0.5) ground, one quad light, env map"""
    import hala_renderer_amd as H
    s = scenes.bunny_class(subdivisions=2, aspect=48 / 32, disney=True)
    s.materials[0] = H.HalaMaterial(type=1, base_color=(1.0, 1.0, 1.0), metallic=0.0, roughness=0.1, specular_transmission=1.0, ior=1.45,
                                    medium=H.HalaMedium(1, (0.9, 0.5, 0.3), 1.5, 0.0))
    fog = scenes.blob_mesh(subdivisions=2, amplitude=0.0)
    fog.material_index = len(s.materials)
    s.materials.append(H.HalaMaterial(type=0, base_color=(1.0, 1.0, 1.0), roughness=0.5, opacity=0.0, medium=H.HalaMedium(2, (0.95, 0.9, 0.8), 2.5, 0.6)))
    s.meshes.append(H.HalaMesh([fog]))
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] *= 0.6
    m[:3, 3] = (1.4, 0.2, 0.4)
    s.nodes.append(H.HalaNode(name="fog", mesh_index=len(s.meshes) - 1, local_transform=m))
    for mat in s.materials[1:2]:
        mat.opacity = 0.5
    return s


def media(write=True):
    env = scenes.sky_sun_envmap(64, 32, sun_gain=100.0)
    sc = O.OracleScene(media_scene(), envmap=env)
    imgs, st = sc.render(48, 32, frames=2, max_depth=12, rr_depth=3, env_rotation=15.0)
    if write:
        np.savez_compressed(os.path.join(HERE, "media_glass_48x32_2spp.npz"), accum=imgs[0], albedo=imgs[1], normal=imgs[2], env=env,
                            rays=np.array([st.rays_closest, st.rays_shadow], dtype=np.uint64))
    return imgs, st


def instanced_scene():
    """RENDER_SPEC 4.5: the Cornell box with the short block's mesh referenced three more times — rotated + non-uniformly scaled, mirrored,
    sheared by its parent — so that its ten triangles are intersected in object space when instancing is on"""
    import hala_renderer_amd as H
    s = scenes.cornell_box(aspect=64 / 48)

    def xf(t, scale, cs):
        c, sn = cs  # exact rationals (3-4-5 and 5-12-13 triangles): no libm in the fixture's scene
        m = np.eye(4)
        m[:3, :3] = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]]) @ np.diag(scale)
        m[:3, 3] = t
        return m.astype(np.float32)

    shear = np.eye(4, dtype=np.float32); shear[0, 1] = 0.35; shear[:3, 3] = (60.0, 200.0, 120.0)
    s.nodes.append(H.HalaNode(name="copy0", mesh_index=1, local_transform=xf((250, 330, 230), (0.6, 1.4, 0.5), (0.8, 0.6))))
    s.nodes.append(H.HalaNode(name="copy1", mesh_index=1, local_transform=xf((520, 60, 90), (-0.8, 0.9, 0.7), (12.0 / 13.0, -5.0 / 13.0))))
    s.nodes.append(H.HalaNode(name="shear", local_transform=shear))
    s.nodes.append(H.HalaNode(name="copy2", mesh_index=1, parent=len(s.nodes) - 1, local_transform=xf((-20, 150, 200), (0.5, 0.5, 0.5), (0.6, 0.8))))
    return s


def instanced(write=True):
    O.set_instancing(True)
    try:
        sc = O.OracleScene(instanced_scene())
        imgs, st = sc.render(64, 48, frames=2, max_depth=5, rr_depth=3)
    finally:
        O.set_instancing(False)
    if write:
        np.savez_compressed(os.path.join(HERE, "instanced_cornell_64x48_2spp.npz"), accum=imgs[0], albedo=imgs[1], normal=imgs[2],
                            rays=np.array([st.rays_closest, st.rays_shadow], dtype=np.uint64))
    return imgs, st


def config0(write=True):
    """BASELINE configs[0] (Cornell 512x512, 1 spp): hash of the colour PFM the reference's save_images would write"""
    import hashlib
    import json
    from hala_renderer_amd import workloads
    cfg = workloads.baseline_config(0)
    imgs, st = O.OracleScene(cfg["scene"]).render(512, 512, frames=1, max_depth=cfg["max_depth"], rr_depth=cfg["rr_depth"])
    out = {"what": "BASELINE configs[0]: Cornell box 512x512, 1 spp, frame_index 0, max_depth 5, rr_depth 3 — sha256 of the _color.pfm bytes (oracle render; tests/test_baseline_configs.py regenerates and compares)",
           "color_pfm_sha256": hashlib.sha256(O.pfm_bytes(imgs[0])).hexdigest(), "rays": [int(st.rays_closest), int(st.rays_shadow)],
           "mean_radiance": float(imgs[0][..., :3].mean())}
    if write:
        json.dump(out, open(os.path.join(HERE, "config0_cornell_512x512_1spp.json"), "w"), indent=1)
    return out


def main():
    if "--only-config0" in sys.argv:
        print(config0())
        return
    if "--only-instanced" in sys.argv:  # round 3 (RENDER_SPEC 4.5): written alone so that the older files keep their bytes
        instanced()
        print("instanced fixture written to", HERE)
        return
    if "--only-media" in sys.argv:  # added later than the others: written alone so that their files keep their bytes
        media()
        print("media fixture written to", HERE)
        return
    media()
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
