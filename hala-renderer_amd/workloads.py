"""The five BASELINE.json configs as (scene, environment, frame) bundles, so that bench.py, the parity tests and the
profiling scripts all render exactly the same thing.  No assets exist in the container: every scene is
procedural (`scenes.py`), built as the `cpu::HalaScene` the reference's loader would hand to `set_scene`."""
import os

from . import scenes

MAX_DEPTH, RR_DEPTH = 5, 3

_NAMES = [
    "configs[0]: Cornell box (32 triangles), 512x512, 1 spp",
    "configs[1]: Cornell box (32 triangles), 1920x1080, 4 spp, diffuse-only closest hit",
    "configs[2]: bunny-class blob (81 920 triangles) + 2048x1024 env-map importance sampling, 1920x1080, 16 spp",
    "configs[3]: Sponza-class atrium (1 000 210 triangles, 54 instances, 24 materials incl. Disney glass / metal / clearcoat, "
    "18 procedural 1024^2 textures with mips, 2 quad lights + 1024x512 env map), 1920x1080, 4 spp",
    "configs[4]: the same atrium, 3840x2160, 4 spp, pixel-tile shard",
]
_FRAMES = [(512, 512, 1), (1920, 1080, 4), (1920, 1080, 16), (1920, 1080, 4), (3840, 2160, 4)]


def atrium(target_triangles=1_000_000, aspect=16.0 / 9.0, textures=True, texture_size=1024):
    """configs[3]/[4]'s scene + env map (the 18 textures: 6 sets of base colour + normal + metallic-roughness)"""
    s = scenes.sponza_class(target_triangles=target_triangles, aspect=aspect)
    if textures and not os.environ.get("HALART_NO_TEXTURES"):  # (experiment knob: what the texture fetches cost)
        scenes.attach_textures(s, sets=6, size=int(os.environ.get("HALART_TEXTURE_SIZE", texture_size)))  # (experiment knob)
    return s, scenes.sky_sun_envmap(1024, 512, sun_gain=50.0)


def baseline_config(index, width=None, height=None):
    """-> dict(name, scene, env (ndarray or None), width, height, spp, max_depth, rr_depth)"""
    w, h, spp = _FRAMES[index]
    w, h = width or w, height or h
    env = None
    if index in (0, 1):
        scene = scenes.cornell_box(aspect=w / h)
    elif index == 2:
        scene, env = scenes.bunny_class(subdivisions=6, disney=True, aspect=w / h), scenes.sky_sun_envmap(2048, 1024)
    else:
        scene, env = atrium(aspect=w / h)
    return {"name": _NAMES[index], "scene": scene, "env": env, "width": w, "height": h, "spp": spp,
            "max_depth": MAX_DEPTH, "rr_depth": RR_DEPTH}
