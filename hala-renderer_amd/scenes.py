"""Procedural scenes for the BASELINE.json configs: no scene or HDR assets exist in the
container, so Cornell box, the "bunny-class" mesh, the "Sponza-class" atrium and the env maps are generated
from closed forms / integer hashes.  Everything is built as a `scene.HalaScene`, i.e. exactly what
cpu::HalaScene::new would hand to set_scene (src/scene/cpu/scene.rs:40-55).
"""
import math

import numpy as np

from . import _abi as A
from .scene import (HalaLight, HalaLightType, HalaMaterial, HalaMaterialType, HalaMesh, HalaNode,
                    HalaPerspectiveCamera, HalaPrimitive, HalaScene)


def _vertices(pos, nrm, uv=None, tan=None):
    v = np.zeros(len(pos), dtype=A.VERTEX_DTYPE)
    v["position"] = np.asarray(pos, dtype=np.float32)
    v["normal"] = np.asarray(nrm, dtype=np.float32)
    v["tangent"] = np.asarray(tan if tan is not None else np.tile([1.0, 0.0, 0.0], (len(pos), 1)), dtype=np.float32)
    v["tex_coord"] = np.asarray(uv if uv is not None else np.zeros((len(pos), 2)), dtype=np.float32)
    return v


def _quad(p0, p1, p2, p3):
    """two triangles (p0,p1,p2), (p0,p2,p3) with a flat normal = normalize((p1-p0) x (p3-p0))"""
    p = np.array([p0, p1, p2, p3], dtype=np.float64)
    n = np.cross(p[1] - p[0], p[3] - p[0])
    n = n / np.linalg.norm(n)
    uv = [[0, 0], [1, 0], [1, 1], [0, 1]]
    return p, np.tile(n, (4, 1)), uv, [0, 1, 2, 0, 2, 3]


def _merge_quads(quads):
    pos, nrm, uv, idx = [], [], [], []
    for q in quads:
        p, n, t, i = _quad(*q)
        base = len(pos)
        pos += p.tolist(); nrm += n.tolist(); uv += t; idx += [base + k for k in i]
    return HalaPrimitive(indices=np.array(idx, dtype=np.uint32), vertices=_vertices(pos, nrm, uv))


def look_at_node_transform(eye, target, up=(0.0, 1.0, 0.0)):
    """glTF camera convention: the camera looks down its local -Z (src/scene/gpu/camera.rs:29-32)."""
    eye = np.asarray(eye, dtype=np.float64); target = np.asarray(target, dtype=np.float64)
    f = target - eye; f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64)); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.eye(4)
    m[:3, 0] = r; m[:3, 1] = u; m[:3, 2] = -f; m[:3, 3] = eye
    return m.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------
# Config 1/2: Cornell box, 32 triangles
# ---------------------------------------------------------------------------------------------------------
def cornell_box(aspect=1.0, analytic_light=True) -> HalaScene:
    """Classic 555-unit Cornell box: 5 walls (10 tris) + short and tall block (5 faces each, 20 tris) +
    a 2-triangle light fixture = 32 triangles; one QUAD light (type 3) just below the fixture."""
    white, red, green, fixture, block = 0, 1, 2, 3, 4
    s = HalaScene()
    s.materials = [
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.73, 0.73, 0.73), roughness=0.0),
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.65, 0.05, 0.05), roughness=0.0),
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.12, 0.45, 0.15), roughness=0.0),
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.0, 0.0, 0.0), roughness=0.0, emission=(1.0, 0.7, 0.2)),
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.73, 0.73, 0.73), roughness=0.6),
    ]
    walls_white = _merge_quads([
        ((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)),              # floor
        ((556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)),  # ceiling
        ((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)),  # back
    ]); walls_white.material_index = white
    wall_green = _merge_quads([((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2))]); wall_green.material_index = green
    wall_red = _merge_quads([((552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0))]); wall_red.material_index = red
    fix = _merge_quads([((348, 548.6, 222), (348, 548.6, 337), (208, 548.6, 337), (208, 548.6, 222))]); fix.material_index = fixture
    short = _merge_quads([
        ((130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)),
        ((290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)),
        ((130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)),
        ((82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)),
        ((240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)),
    ]); short.material_index = block
    tall = _merge_quads([
        ((423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)),
        ((423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)),
        ((472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)),
        ((314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)),
        ((265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)),
    ]); tall.material_index = block
    s.meshes = [HalaMesh([walls_white, wall_green, wall_red, fix]), HalaMesh([short]), HalaMesh([tall])]
    s.nodes = [
        HalaNode(name="room", mesh_index=0),
        HalaNode(name="short_block", mesh_index=1),
        HalaNode(name="tall_block", mesh_index=2),
        HalaNode(name="camera", camera_index=0, local_transform=look_at_node_transform((278, 273, -800), (278, 273, 0))),
    ]
    s.cameras = [HalaPerspectiveCamera(aspect=aspect, yfov=2.0 * math.atan(0.0125 / 0.035), znear=1.0)]
    if analytic_light:
        m = np.eye(4, dtype=np.float32)
        m[:3, 0] = (1, 0, 0); m[:3, 1] = (0, 0, 1); m[:3, 2] = (0, -1, 0); m[:3, 3] = (278.0, 548.3, 279.5)
        s.nodes.append(HalaNode(name="light", light_index=0, local_transform=m))
        s.lights = [HalaLight(color=(1.0, 12.0 / 17.0, 4.0 / 17.0), intensity=17.0, light_type=HalaLightType.QUAD, params=(130.0, 105.0))]
    return s


# ---------------------------------------------------------------------------------------------------------
# Config 3: "bunny-class" closed mesh — subdivided icosphere displaced by hash noise
# ---------------------------------------------------------------------------------------------------------
def _icosphere(subdivisions):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdivisions):
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        edges = np.concatenate([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)])
        key = np.sort(edges, axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        ab, bc, ca = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v, f


def _hash_noise(p, seed):
    """smooth 3-D value noise from an integer hash; p float64 [N,3] -> [N] in [-1,1]"""
    def h(ix, iy, iz):
        x = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) ^ \
            (iz.astype(np.uint64) * np.uint64(83492791)) ^ np.uint64(seed * 2654435761 & 0xFFFFFFFF)
        x = (x ^ (x >> np.uint64(13))) * np.uint64(0x5BD1E995) & np.uint64(0xFFFFFFFF)
        x = x ^ (x >> np.uint64(15))
        return (x & np.uint64(0xFFFFFF)).astype(np.float64) / float(0xFFFFFF) * 2.0 - 1.0
    pf = np.floor(p); fr = p - pf
    w = fr * fr * (3.0 - 2.0 * fr)
    i = pf.astype(np.int64) + 1024
    out = np.zeros(len(p))
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                wx = w[:, 0] if dx else 1.0 - w[:, 0]
                wy = w[:, 1] if dy else 1.0 - w[:, 1]
                wz = w[:, 2] if dz else 1.0 - w[:, 2]
                out += wx * wy * wz * h(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz)
    return out


def _smooth_normals(pos, faces):
    n = np.zeros_like(pos)
    fn = np.cross(pos[faces[:, 1]] - pos[faces[:, 0]], pos[faces[:, 2]] - pos[faces[:, 0]])
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return n / ln


def blob_mesh(subdivisions=6, seed=1234, radius=1.0, amplitude=0.25) -> HalaPrimitive:
    """icosphere (20 * 4^s triangles; s=6 -> 81 920) displaced along the normal by 3 octaves of hash noise"""
    v, f = _icosphere(subdivisions)
    d = np.zeros(len(v))
    for o, (freq, amp) in enumerate([(1.7, 1.0), (3.9, 0.45), (8.3, 0.2)]):
        d += amp * _hash_noise(v * freq + 17.0 * o, seed + o)
    pos = v * (radius * (1.0 + amplitude * d))[:, None]
    nrm = _smooth_normals(pos, f)
    uv = np.stack([0.5 + np.arctan2(v[:, 2], v[:, 0]) / (2 * math.pi), np.arccos(np.clip(v[:, 1], -1, 1)) / math.pi], 1)
    return HalaPrimitive(indices=f.astype(np.uint32).reshape(-1), vertices=_vertices(pos, nrm, uv))


def bunny_class(subdivisions=6, seed=1234, aspect=16.0 / 9.0, disney=False) -> HalaScene:
    """Config 3: ~82 k-triangle closed blob on a ground quad, lit by an env map (set separately)."""
    s = HalaScene()
    s.materials = [
        HalaMaterial(type=HalaMaterialType.DISNEY if disney else HalaMaterialType.DIFFUSE, base_color=(0.8, 0.6, 0.4),
                     roughness=0.35 if disney else 0.4, metallic=0.0),
        HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.5, 0.5, 0.5), roughness=0.0),
    ]
    blob = blob_mesh(subdivisions, seed); blob.material_index = 0
    ground = _merge_quads([((-8, -1.3, 8), (8, -1.3, 8), (8, -1.3, -8), (-8, -1.3, -8))]); ground.material_index = 1
    s.meshes = [HalaMesh([blob]), HalaMesh([ground])]
    s.nodes = [HalaNode(name="blob", mesh_index=0), HalaNode(name="ground", mesh_index=1),
               HalaNode(name="camera", camera_index=0, local_transform=look_at_node_transform((0.0, 0.6, 4.2), (0.0, 0.0, 0.0)))]
    s.cameras = [HalaPerspectiveCamera(aspect=aspect, yfov=math.radians(40.0), znear=0.1)]
    return s


def stacked_sheets(count=4096, seed=3, aspect=1.0) -> HalaScene:
    """Traversal-stack stress case: `count` large, slightly jittered quads piled inside one small slab, all overlapping
    in x/y.  Every node's child boxes overlap almost completely, so a ray through the pile hits every child of every node:
    closest-hit rays visit most of the tree and the per-lane stack grows by up to three entries per level (past the
    entries kept in LDS, into the global spill area)."""
    rng = np.random.RandomState(seed)
    quads = []
    for i in range(count):
        z = float(rng.uniform(-0.05, 0.05))
        dx, dy = rng.uniform(-0.02, 0.02, 2)
        tilt = float(rng.uniform(-0.01, 0.01))
        quads.append(((-1 + dx, -1 + dy, z - tilt), (1 + dx, -1 + dy, z + tilt), (1 + dx, 1 + dy, z + tilt), (-1 + dx, 1 + dy, z - tilt)))
    s = HalaScene()
    s.materials = [HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=(0.7, 0.7, 0.7))]
    prim = _merge_quads(quads); prim.material_index = 0
    s.meshes = [HalaMesh([prim])]
    s.nodes = [HalaNode(name="sheets", mesh_index=0),
               HalaNode(name="camera", camera_index=0, local_transform=look_at_node_transform((0.3, 0.2, 3.0), (0.0, 0.0, 0.0)))]
    s.cameras = [HalaPerspectiveCamera(aspect=aspect, yfov=math.radians(45.0), znear=0.1)]
    return s


def sky_sun_envmap(width=2048, height=1024, sun_dir=(0.4, 0.6, 0.35), sun_radius_deg=2.0, sun_gain=1.0e4):
    """Config 3 env map: analytic sky gradient + a sun disc 1e4x brighter (exercises the A1 table tails).
    Returns RGBA32F [H, W, 4] with row 0 = top (v = 0 <-> +Y), matching RENDER_SPEC §7.3's (u, v) mapping."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    theta = v * math.pi
    phi = u * 2.0 * math.pi - math.pi
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    d = np.stack([st * np.cos(phi)[None, :], np.broadcast_to(ct, (height, width)), st * np.sin(phi)[None, :]], -1)
    t = np.clip(0.5 * (d[..., 1] + 1.0), 0, 1)
    horizon = np.array([0.9, 0.9, 0.95]); zenith = np.array([0.25, 0.45, 0.9]); ground = np.array([0.2, 0.18, 0.15])
    up = np.clip(d[..., 1], 0, 1)[..., None]
    sky = horizon * (1 - up) + zenith * up
    img = np.where(d[..., 1:2] >= 0, sky, ground * (0.3 + 0.7 * t[..., None]))
    sd = np.asarray(sun_dir, dtype=np.float64); sd /= np.linalg.norm(sd)
    cosang = d @ sd
    sun = cosang >= math.cos(math.radians(sun_radius_deg))
    img = np.where(sun[..., None], np.array([1.0, 0.95, 0.85]) * sun_gain, img)
    out = np.ones((height, width, 4), dtype=np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------
# procedural textures (config 4: ">= 16 procedural 1024^2 textures (base colour + normal + MR) with mips")
# ---------------------------------------------------------------------------------------------------------
def _tile_noise(size, cells, seed):
    """tileable value noise in [0,1], [size,size] float64"""
    rng = np.random.RandomState(seed)
    g = rng.rand(cells, cells)
    t = np.arange(size) * (cells / size)
    i0 = np.floor(t).astype(int) % cells
    i1 = (i0 + 1) % cells
    f = t - np.floor(t)
    w = f * f * (3 - 2 * f)
    a = g[i0][:, i0] * (1 - w)[None, :] + g[i0][:, i1] * w[None, :]
    b = g[i1][:, i0] * (1 - w)[None, :] + g[i1][:, i1] * w[None, :]
    return a * (1 - w)[:, None] + b * w[:, None]


def procedural_texture_set(size=1024, seed=0, tint=(0.8, 0.7, 0.6)):
    """returns (base_color sRGB8, normal UNORM8, metallic_roughness UNORM8) as HalaImageData"""
    from .scene import HalaImageData
    n = 0.55 * _tile_noise(size, 8, seed) + 0.3 * _tile_noise(size, 32, seed + 1) + 0.15 * _tile_noise(size, 128, seed + 2)
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    tiles = (((xx * 8 // size) + (yy * 8 // size)) % 2).astype(np.float64)
    base = np.clip((0.55 + 0.45 * n)[..., None] * np.asarray(tint)[None, None, :] * (0.75 + 0.25 * tiles[..., None]), 0, 1)
    base8 = np.concatenate([np.round(base * 255), np.full((size, size, 1), 255.0)], -1).astype(np.uint8)
    hgt = n + 0.08 * tiles
    dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
    dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
    nrm = np.stack([-dx * size / 64.0, -dy * size / 64.0, np.ones_like(dx)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm8 = np.concatenate([np.round((nrm * 0.5 + 0.5) * 255), np.full((size, size, 1), 255.0)], -1).astype(np.uint8)
    rough = np.clip(0.35 + 0.6 * _tile_noise(size, 16, seed + 3), 0, 1)
    metal = (tiles > 0.5).astype(np.float64) * (_tile_noise(size, 4, seed + 4) > 0.5)
    mr8 = np.stack([np.zeros_like(rough), np.round(rough * 255), np.round(metal * 255), np.full_like(rough, 255.0)], -1).astype(np.uint8)
    return (HalaImageData(A_FORMAT_SRGB, size, size, base8), HalaImageData(A_FORMAT_UNORM, size, size, nrm8), HalaImageData(A_FORMAT_UNORM, size, size, mr8))


A_FORMAT_UNORM, A_FORMAT_SRGB, A_FORMAT_FLOAT, A_FORMAT_BGRA_TAG = 0, 1, 2, 3  # HALA_FORMAT_* of include/halart.h


def attach_textures(scene: HalaScene, sets, size=1024, seed=100, every=1):
    """generates `sets` texture triples and binds them round-robin to the scene's materials (every `every`-th material)"""
    for k in range(sets):
        tint = [(0.85, 0.8, 0.7), (0.7, 0.35, 0.3), (0.35, 0.5, 0.75), (0.45, 0.65, 0.4), (0.85, 0.75, 0.4)][k % 5]
        for img in procedural_texture_set(size, seed + 10 * k, tint):
            idx = len(scene.image_data)
            scene.image_data.append(img)
            scene.image2data_mapping[idx] = idx
            scene.texture2image_mapping[idx] = idx
    for i, m in enumerate(scene.materials):
        if i % every:
            continue
        k = (i // every) % sets
        m.base_color_map_index = 3 * k
        m.normal_map_index = 3 * k + 1
        if m.type == HalaMaterialType.DISNEY:
            m.metallic_roughness_map_index = 3 * k + 2
    return scene


# ---------------------------------------------------------------------------------------------------------
# Config 4/5: "Sponza-class" atrium, ~1 M triangles
# ---------------------------------------------------------------------------------------------------------
def _grid_mesh(nx, nz, fn):
    """(nx x nz) quad grid -> 2*nx*nz triangles; fn(u, v) -> positions [.,3] for u,v in [0,1]"""
    u, v = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, nz + 1), indexing="ij")
    pos = fn(u.reshape(-1), v.reshape(-1))
    i, j = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    a = (i * (nz + 1) + j).reshape(-1); b = a + (nz + 1); c = b + 1; d = a + 1
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    nrm = _smooth_normals(pos, faces)
    uv = np.stack([u.reshape(-1), v.reshape(-1)], 1)
    return HalaPrimitive(indices=faces.astype(np.uint32).reshape(-1), vertices=_vertices(pos, nrm, uv))


def _column(segments, rings, flutes=12):
    def fn(u, v):
        ang = u * 2 * math.pi
        r = 0.45 * (1.0 - 0.12 * v) * (1.0 + 0.06 * np.cos(ang * flutes)) * (1.0 + 0.25 * np.exp(-((v - 0.02) / 0.03) ** 2) + 0.25 * np.exp(-((v - 0.98) / 0.03) ** 2))
        return np.stack([r * np.cos(ang), v * 6.0, -r * np.sin(ang)], 1)
    return _grid_mesh(segments, rings, fn)


def _drape(nx, nz, seed):
    def fn(u, v):
        p = np.stack([u * 6.0, v * 3.0, np.zeros_like(u)], 1)
        wave = 0.25 * np.sin(u * 9.0 * math.pi) * (0.3 + 0.7 * v) + 0.08 * _hash_noise(p * 2.3, seed)
        return np.stack([u * 3.0 - 1.5, 5.5 - v * 3.5 - 0.4 * np.sin(u * math.pi), wave], 1)
    return _grid_mesh(nx, nz, fn)


def _arch(nx, nz):
    def fn(u, v):
        ang = u * math.pi
        r = 2.0 + 0.35 * v
        return np.stack([r * np.cos(ang), 6.0 + r * np.sin(ang) * 0.8, v * 0.6 - 0.3], 1)
    return _grid_mesh(nx, nz, fn)


def sponza_class(target_triangles=1_000_000, aspect=16.0 / 9.0, seed=7, disney=True) -> HalaScene:
    """Atrium of instanced fluted columns, arches, draped cloth and a tessellated floor.  Instancing follows the
    reference data model: several nodes reference one mesh (gpu_uploader.rs:843-875 emits one instance per
    node x primitive).  Triangle count is steered by the tessellation level to ~target_triangles."""
    s = HalaScene()
    rng = np.random.RandomState(seed)
    T = HalaMaterialType.DISNEY if disney else HalaMaterialType.DIFFUSE
    palette = [(0.78, 0.74, 0.66), (0.65, 0.62, 0.58), (0.7, 0.25, 0.2), (0.2, 0.35, 0.6), (0.25, 0.5, 0.3), (0.8, 0.7, 0.3)]
    s.materials = []
    for i in range(24):
        base = palette[i % len(palette)]
        if i % 4 == 0:
            s.materials.append(HalaMaterial(type=HalaMaterialType.DIFFUSE, base_color=base, roughness=0.2 + 0.03 * i))
        else:
            s.materials.append(HalaMaterial(type=T, base_color=base, roughness=0.15 + 0.035 * (i % 12),
                                            metallic=1.0 if i % 6 == 1 else 0.0,
                                            clearcoat=1.0 if i % 8 == 3 else 0.0, clearcoat_roughness=0.1,
                                            specular_transmission=0.9 if i % 8 == 5 else 0.0, ior=1.5))  # 5: the second column mesh and one drape are glass
    n_col, n_arch, n_drape = 28, 14, 10
    # budget: columns 55 %, drapes 25 %, arches 8 %, floor+walls 12 %
    k = math.sqrt(max(target_triangles, 2000) / 1_000_000.0)
    cs, cr = max(8, int(96 * k)), max(8, int(102 * k))       # 2*cs*cr tris per column
    dn, dm = max(8, int(112 * k)), max(8, int(112 * k))
    an, am = max(8, int(170 * k)), max(4, int(17 * k))
    fl = max(8, int(245 * k))
    col_mesh = HalaMesh([_column(cs, cr)]); col_mesh.primitives[0].material_index = 1
    col_mesh2 = HalaMesh([_column(cs, cr, flutes=16)]); col_mesh2.primitives[0].material_index = 5
    arch_mesh = HalaMesh([_arch(an, am)]); arch_mesh.primitives[0].material_index = 4
    drape_meshes = []
    for i in range(n_drape):
        m = HalaMesh([_drape(dn, dm, seed + i)]); m.primitives[0].material_index = 2 + (i % 20)
        drape_meshes.append(m)
    floor = _grid_mesh(fl, fl, lambda u, v: np.stack([u * 40.0 - 20.0, 0.02 * np.sin(u * 60) * np.sin(v * 60), v * 24.0 - 12.0], 1))
    floor.material_index = 0
    walls = _merge_quads([((-20, 0, -12), (20, 0, -12), (20, 12, -12), (-20, 12, -12)),
                          ((20, 0, 12), (-20, 0, 12), (-20, 12, 12), (20, 12, 12)),
                          ((-20, 0, 12), (-20, 0, -12), (-20, 12, -12), (-20, 12, 12)),
                          ((20, 0, -12), (20, 0, 12), (20, 12, 12), (20, 12, -12))]); walls.material_index = 8
    s.meshes = [col_mesh, col_mesh2, arch_mesh] + drape_meshes + [HalaMesh([floor, walls])]
    room_mesh = len(s.meshes) - 1
    s.nodes = [HalaNode(name="atrium")]  # parent of everything: exercises update_node_hierarchies
    s.nodes.append(HalaNode(name="room", parent=0, mesh_index=room_mesh))

    def xf(tx, ty, tz, ry=0.0, sc=1.0):
        m = np.eye(4)
        c, sn = math.cos(ry), math.sin(ry)
        m[:3, :3] = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]]) * sc
        m[:3, 3] = (tx, ty, tz)
        return m.astype(np.float32)

    for i in range(n_col):
        row = i % 2
        x = -17.5 + (i // 2) * (35.0 / (n_col // 2 - 1))
        z = -7.0 if row == 0 else 7.0
        s.nodes.append(HalaNode(name=f"column_{i}", parent=0, mesh_index=i % 2, local_transform=xf(x, 0.0, z, ry=0.37 * i)))
    for i in range(n_arch):
        x = -17.5 + (i + 0.5) * (35.0 / n_arch)
        s.nodes.append(HalaNode(name=f"arch_{i}", parent=0, mesh_index=2, local_transform=xf(x, 0.0, -7.0 if i % 2 else 7.0, sc=0.62)))
    for i in range(n_drape):
        x = -16.0 + i * 3.5
        s.nodes.append(HalaNode(name=f"drape_{i}", parent=0, mesh_index=3 + i, local_transform=xf(x, 3.0, -10.5 + (i % 3) * 0.4, ry=0.05 * i)))
    s.nodes.append(HalaNode(name="camera", parent=0, camera_index=0,
                            local_transform=look_at_node_transform((-15.0, 3.2, 0.5), (6.0, 3.5, -1.0))))
    s.cameras = [HalaPerspectiveCamera(aspect=aspect, yfov=math.radians(55.0), znear=0.1)]
    for i, (x, z) in enumerate([(-8.0, 0.0), (9.0, 0.0)]):
        m = np.eye(4, dtype=np.float32)
        m[:3, 0] = (1, 0, 0); m[:3, 1] = (0, 0, 1); m[:3, 2] = (0, -1, 0); m[:3, 3] = (x, 11.5, z)
        s.nodes.append(HalaNode(name=f"light_{i}", parent=0, light_index=i, local_transform=m))
        s.lights.append(HalaLight(color=(1.0, 0.95, 0.85), intensity=40.0, light_type=HalaLightType.QUAD, params=(4.0, 3.0)))
    _ = rng
    return s
