"""Random small scenes for the GPU-vs-oracle parity soak (tests/test_gpu_parity.py::test_random_scenes_bit_exact,
scripts/soak_random_scenes.py): every feature of docs/RENDER_SPEC.md drawn at random and combined — Disney parameters, transmission,
opacity, media, texture maps, all light types, env map on / off, perspective / thin-lens / orthographic cameras, node hierarchies with
rotation and non-uniform (also mirrored) scale, render depth and tonemap settings."""
import math

import numpy as np

import hala_renderer_amd as H
from hala_renderer_amd import scenes

f32 = np.float32


def _rot_scale(rs, mirror_ok=True):
    a, b = rs.uniform(-math.pi, math.pi, 2)
    ca, sa, cb, sb = math.cos(a), math.sin(a), math.cos(b), math.sin(b)
    ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
    rx = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
    sc = rs.uniform(0.5, 1.6, 3)
    if mirror_ok and rs.rand() < 0.25:
        sc[rs.randint(3)] *= -1.0
    return (ry @ rx) * sc[None, :]


def _xform(rs, t, mirror_ok=True):
    m = np.eye(4, dtype=f32)
    m[:3, :3] = _rot_scale(rs, mirror_ok).astype(f32)
    m[:3, 3] = t
    return m


def random_material(rs):
    M = H.HalaMaterial()
    M.type = H.HalaMaterialType.DISNEY if rs.rand() < 0.7 else H.HalaMaterialType.DIFFUSE
    M.base_color = tuple(rs.uniform(0.05, 1.0, 3))
    M.roughness = float(rs.choice([0.0, rs.uniform(0.02, 1.0)], p=[0.1, 0.9]))
    M.metallic = float(rs.choice([0.0, 1.0, rs.uniform()], p=[0.4, 0.2, 0.4]))
    M.anisotropic = float(rs.choice([0.0, rs.uniform()], p=[0.6, 0.4]))
    M.subsurface = float(rs.choice([0.0, rs.uniform()], p=[0.7, 0.3]))
    M.specular_tint = float(rs.uniform()) if rs.rand() < 0.3 else 0.0
    M.sheen = float(rs.uniform()) if rs.rand() < 0.3 else 0.0
    M.sheen_tint = float(rs.uniform())
    M.clearcoat = float(rs.uniform()) if rs.rand() < 0.3 else 0.0
    M.clearcoat_roughness = float(rs.uniform())
    M.ior = float(rs.uniform(1.05, 2.2))
    if M.type == H.HalaMaterialType.DISNEY and rs.rand() < 0.3:
        M.specular_transmission = float(rs.choice([1.0, rs.uniform(0.2, 1.0)]))
    if rs.rand() < 0.2:
        M.opacity = float(rs.choice([0.0, rs.uniform(0.1, 0.9)]))
    if rs.rand() < 0.15:
        M.emission = tuple(rs.uniform(0.0, 3.0, 3))
    if rs.rand() < 0.3:
        kind = rs.choice([H.HalaMediumType.ABSORB, H.HalaMediumType.SCATTER, H.HalaMediumType.EMISSIVE])
        M.medium = H.HalaMedium(type=int(kind), color=tuple(rs.uniform(0.1, 1.0, 3)), density=float(rs.uniform(0.1, 3.0)),
                                anisotropy=float(rs.uniform(-0.8, 0.8)))
    return M


def random_scene(seed, big=False, instances=False):
    """-> (scene, env or None, render kwargs); instances: some objects are referenced by further nodes with transforms of their own
    (RENDER_SPEC 4.5: such primitives are intersected in object space when the tree is two-level) — drawn from a stream of their own, so
    the scene is otherwise the one the seed gives without them"""
    rs = np.random.RandomState(seed)
    s = H.HalaScene()
    n_obj = rs.randint(2, 5)
    s.nodes.append(H.HalaNode(name="root", local_transform=_xform(rs, rs.uniform(-0.3, 0.3, 3), mirror_ok=False)))
    for k in range(n_obj):
        sub = (5 if k == 0 else 3) if big else int(rs.randint(1, 4))
        prim = scenes.blob_mesh(subdivisions=sub, seed=int(rs.randint(1 << 20)), radius=float(rs.uniform(0.4, 0.9)), amplitude=float(rs.uniform(0.0, 0.3)))
        prim.material_index = len(s.materials)
        s.materials.append(random_material(rs))
        s.meshes.append(H.HalaMesh([prim]))
        parent = 0 if rs.rand() < 0.7 or k == 0 else int(rs.randint(1, len(s.nodes)))
        s.nodes.append(H.HalaNode(name=f"obj{k}", parent=parent, mesh_index=len(s.meshes) - 1,
                                  local_transform=_xform(rs, rs.uniform(-1.6, 1.6, 3) * np.array([1.0, 0.5, 1.0]))))
    ground = scenes._merge_quads([((-6, -1.4, 6), (6, -1.4, 6), (6, -1.4, -6), (-6, -1.4, -6))])
    ground.material_index = len(s.materials)
    gm = random_material(rs); gm.opacity = 1.0 if rs.rand() < 0.8 else gm.opacity; gm.specular_transmission = 0.0; gm.medium = H.HalaMedium()
    s.materials.append(gm)
    s.meshes.append(H.HalaMesh([ground]))
    s.nodes.append(H.HalaNode(name="ground", mesh_index=len(s.meshes) - 1))
    if rs.rand() < 0.5:
        scenes.attach_textures(s, sets=int(rs.randint(1, 3)), size=int(rs.choice([8, 16, 33])), seed=int(rs.randint(1000)), every=int(rs.randint(1, 3)))
        if rs.rand() < 0.5:  # an emission map too
            s.materials[0].emission_map_index = 0
            s.materials[0].emission = (1.0, 0.8, 0.6)
    types = [H.HalaLightType.POINT, H.HalaLightType.DIRECTIONAL, H.HalaLightType.SPOT, H.HalaLightType.QUAD, H.HalaLightType.SPHERE]
    for k in range(rs.randint(0, 4)):
        lt = types[rs.randint(len(types))]
        params = {H.HalaLightType.POINT: (0.0, 0.0), H.HalaLightType.DIRECTIONAL: (float(rs.uniform(0.0, 0.3)), 0.0),
                  H.HalaLightType.SPOT: (float(rs.uniform(0.1, 0.5)), float(rs.uniform(0.5, 1.0))),
                  H.HalaLightType.QUAD: (float(rs.uniform(0.3, 1.5)), float(rs.uniform(0.3, 1.5))), H.HalaLightType.SPHERE: (float(rs.uniform(0.1, 0.5)), 0.0)}[lt]
        s.lights.append(H.HalaLight(tuple(rs.uniform(0.2, 1.0, 3)), float(rs.uniform(2.0, 30.0)), lt, params))
        pos = rs.uniform(-2.5, 2.5, 3); pos[1] = rs.uniform(1.0, 3.5)
        s.nodes.append(H.HalaNode(name=f"light{k}", light_index=k, local_transform=scenes.look_at_node_transform(tuple(pos), tuple(rs.uniform(-0.5, 0.5, 3)))))
    w, h = [(64, 40), (48, 48), (57, 31)][rs.randint(3)]
    eye = rs.uniform(-1.0, 1.0, 3) + np.array([0.0, 0.8, 4.0])
    if rs.rand() < 0.75:
        s.cameras = [H.HalaPerspectiveCamera(aspect=w / h, yfov=float(rs.uniform(0.4, 1.1)), focal_distance=float(rs.uniform(2.0, 6.0)),
                                             aperture=float(rs.choice([0.0, rs.uniform(0.02, 0.3)])))]
    else:
        s.cameras = [H.HalaOrthographicCamera(xmag=float(rs.uniform(1.5, 3.0)), ymag=float(rs.uniform(1.0, 2.5)))]
    s.nodes.append(H.HalaNode(name="camera", camera_index=0, local_transform=scenes.look_at_node_transform(tuple(eye), (0.0, 0.0, 0.0))))
    env = None
    if rs.rand() < 0.6 or not s.lights:
        ew, eh = [(32, 16), (64, 32), (37, 19)][rs.randint(3)]
        env = scenes.sky_sun_envmap(ew, eh, sun_gain=float(rs.choice([1.0, 30.0, 1e3])))
        if rs.rand() < 0.3:
            env[..., :3] *= rs.uniform(0.0, 2.0, (eh, ew, 3)).astype(f32)
    if instances:
        ri = np.random.RandomState(seed + 7919)
        n_meshes = len(s.meshes) - 1  # not the ground
        for k in range(int(ri.randint(2, 5))):
            parent = 0 if ri.rand() < 0.6 else int(ri.randint(1, 1 + n_meshes))  # under the root or under one of the objects
            s.nodes.append(H.HalaNode(name=f"inst{k}", parent=parent, mesh_index=int(ri.randint(0, n_meshes)),
                                      local_transform=_xform(ri, ri.uniform(-1.8, 1.8, 3) * np.array([1.0, 0.5, 1.0]))))
    kw = dict(width=w, height=h, frames=int(rs.randint(1, 4)), max_depth=int(rs.randint(1, 9)), rr_depth=int(rs.randint(1, 5)),
              tonemap=[(False, False, False), (True, False, False), (True, True, False), (True, True, True)][rs.randint(4)],
              env_rotation=float(rs.choice([0.0, rs.uniform(0.0, 360.0)])), env_intensity=float(rs.choice([1.0, rs.uniform(0.2, 3.0)])),
              exposure=float(rs.choice([1.0, rs.uniform(0.3, 3.0)])))
    return s, env, kw
