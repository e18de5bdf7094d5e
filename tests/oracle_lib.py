"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE (the checker), never the product.

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import hala_renderer_amd as H
from hala_renderer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32),
                ("ground_color", C.c_float * 4), ("sky_color", C.c_float * 4), ("env_rotation_degrees", C.c_float),
                ("env_intensity", C.c_float), ("exposure_value", C.c_float), ("enable_tonemap", C.c_int),
                ("enable_aces", C.c_int), ("use_simple_aces", C.c_int), ("num_threads", C.c_int)]


class RenderStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("triangles_tested", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_scene_triangle_count.restype = C.c_uint32
        _lib.orc_scene_node_count.restype = C.c_uint32
        _lib.orc_pfm_bytes.restype = C.c_size_t
    return _lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def envmap_build_distribution(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    total = C.c_float()
    marg = np.empty(h, dtype=np.float32)
    cond = np.empty((h, w), dtype=np.float32)
    lib().orc_envmap_build_distribution(fptr(rgba), C.c_uint32(w), C.c_uint32(h), C.byref(total), fptr(marg), fptr(cond))
    return np.float32(total.value), marg, cond


def envmap_validate(px):
    px = np.ascontiguousarray(px, dtype=np.float32)
    h, w, ch = px.shape
    return lib().orc_envmap_validate(fptr(px), C.c_uint32(ch), C.c_uint32(w), C.c_uint32(h))


def world_transforms(scene: H.HalaScene):
    holder = scene.to_desc()
    out = np.empty((len(scene.nodes), 16), dtype=np.float32)
    lib().orc_update_node_hierarchies(holder.ptr(), fptr(out))
    return out


def pack_material(m: H.HalaMaterial) -> A.GpuMaterial:
    s = H.HalaScene(materials=[m])
    holder = s.to_desc()
    out = A.GpuMaterial()
    lib().orc_pack_material(holder.desc.materials, C.byref(out))
    return out


def pack_cameras(scene):
    holder = scene.to_desc()
    out = (A.GpuCamera * 8)()
    n = lib().orc_pack_cameras(holder.ptr(), out)
    return None if n < 0 else list(out[:n])


def pack_lights(scene):
    holder = scene.to_desc()
    out = (A.GpuLight * 32)()
    bb = (A.Aabb * 32)()
    n = lib().orc_pack_lights(holder.ptr(), out, bb)
    return list(out[:n]), list(bb[:n])


def pack_instances(scene, capacity=65536):
    holder = scene.to_desc()
    t = np.zeros((capacity, 12), dtype=np.float32)
    md = (A.GpuMeshData * capacity)()
    n = lib().orc_pack_instances(holder.ptr(), fptr(t), md, C.c_uint32(capacity))
    return t[:n], list(md[:n])


def primitive_bounds(vertices):
    v = np.ascontiguousarray(vertices, dtype=A.VERTEX_DTYPE)
    c = (C.c_float * 3)()
    e = (C.c_float * 3)()
    lib().orc_primitive_bounds(C.c_void_p(v.ctypes.data), C.c_uint32(v.size), c, e)
    return np.array(c[:], dtype=np.float32), np.array(e[:], dtype=np.float32)


def tonemap_pixels(rgba, enable_tonemap, enable_aces, use_simple_aces):
    out = np.ascontiguousarray(rgba, dtype=np.float32).copy()
    lib().orc_tonemap_pixels(fptr(out), C.c_size_t(out.size // 4), C.c_int(enable_tonemap), C.c_int(enable_aces), C.c_int(use_simple_aces))
    return out


def pfm_bytes(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    buf = (C.c_uint8 * (64 + 12 * w * h))()
    n = lib().orc_pfm_bytes(fptr(rgba), C.c_uint32(w), C.c_uint32(h), buf, C.c_size_t(len(buf)))
    return bytes(buf[:n])


class OracleScene:
    def __init__(self, scene, envmap=None):
        """scene: a hala_renderer_amd.HalaScene, or a NativeScene (what the library's own glTF loader produced)"""
        if hasattr(scene, "desc_ptr"):
            self._h = C.c_void_p(lib().orc_scene_create(scene.desc_ptr()))
        else:
            holder = scene.to_desc()
            self._h = C.c_void_p(lib().orc_scene_create(holder.ptr()))
        if not self._h:
            raise RuntimeError("orc_scene_create failed")
        self.has_env = False
        if envmap is not None:
            self.set_envmap(envmap)

    def set_envmap(self, rgba):
        px = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w, ch = px.shape
        if ch == 3:
            px = np.concatenate([px, np.ones((h, w, 1), np.float32)], -1)
        lib().orc_scene_set_envmap(self._h, fptr(px), C.c_uint32(w), C.c_uint32(h))
        self.has_env = True

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def triangle_count(self):
        return lib().orc_scene_triangle_count(self._h)

    @property
    def node_count(self):
        return lib().orc_scene_node_count(self._h)

    def bounds(self):
        mn = (C.c_float * 3)(); mx = (C.c_float * 3)()
        lib().orc_scene_bounds(self._h, mn, mx)
        return np.array(mn[:], np.float32), np.array(mx[:], np.float32)

    def triangles(self):
        out = np.empty((self.triangle_count, 9), dtype=np.float32)
        lib().orc_scene_get_triangles(self._h, fptr(out))
        return out

    def use_bvh(self, nodes_u32=None, tris_u32=None, refs_u32=None):
        """render() then traverses this product-built tree (None: the oracle's own again); refs_u32: the instance references of a
        two-level tree (renderer.download_instance_refs())"""
        if nodes_u32 is None:
            lib().orc_scene_use_bvh4(self._h, C.c_void_p(0), C.c_uint32(0), C.c_void_p(0), C.c_uint32(0))
            return
        self._ext = (np.ascontiguousarray(nodes_u32), np.ascontiguousarray(tris_u32), None if refs_u32 is None else np.ascontiguousarray(refs_u32))
        if refs_u32 is not None and len(refs_u32):
            lib().orc_scene_use_bvh4_two_level(self._h, C.c_void_p(self._ext[0].ctypes.data), C.c_uint32(self._ext[0].size // 16),
                                               C.c_void_p(self._ext[1].ctypes.data), C.c_uint32(self._ext[1].size // 12),
                                               C.c_void_p(self._ext[2].ctypes.data), C.c_uint32(self._ext[2].size // 16))
            return
        lib().orc_scene_use_bvh4(self._h, C.c_void_p(self._ext[0].ctypes.data), C.c_uint32(self._ext[0].size // 16),
                                 C.c_void_p(self._ext[1].ctypes.data), C.c_uint32(self._ext[1].size // 12))

    def export_bvh4(self):
        """the oracle's SAH tree in the product's compressed 4-wide format: (nodes [n,16] u32, triangles [m,12] u32)"""
        fn = lib().orc_scene_export_bvh4
        fn.restype = C.c_uint32
        n = fn(self._h, C.c_void_p(0), C.c_uint32(0))
        assert n > 0
        nodes = np.zeros((n, 16), dtype=np.uint32)
        assert fn(self._h, C.c_void_p(nodes.ctypes.data), C.c_uint32(n)) == n
        tris = np.zeros((self.triangle_count, 12), dtype=np.uint32)
        lib().orc_scene_get_bvh_triangles(self._h, C.c_void_p(tris.ctypes.data))
        return nodes, tris

    def trace(self, rays, mode=0, count_steps=False, brute=False):
        rays = np.ascontiguousarray(rays, dtype=A.RAY_DTYPE)
        hits = np.empty(rays.shape[0], dtype=A.HIT_DTYPE)
        if brute:
            lib().orc_trace_rays_brute(self._h, C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_uint32(rays.shape[0]), C.c_int(mode))
            return hits
        ctr = (C.c_uint64 * 2)(0, 0)
        lib().orc_trace_rays(self._h, C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_uint32(rays.shape[0]), C.c_int(mode), ctr)
        return (hits, (ctr[0], ctr[1])) if count_steps else hits

    def texture_info(self, tex):
        w, h, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert lib().orc_scene_texture_info(self._h, C.c_uint32(tex), C.byref(w), C.byref(h), C.byref(m)) == 0
        return w.value, h.value, m.value

    def texture_level(self, tex, level):
        w, h, _ = self.texture_info(tex)
        out = np.empty((max(1, h >> level), max(1, w >> level), 4), dtype=np.float32)
        lib().orc_scene_texture_level(self._h, C.c_uint32(tex), C.c_uint32(level), fptr(out))
        return out

    def sample_texture(self, tex, uv_lod):
        q = np.ascontiguousarray(uv_lod, dtype=np.float32).reshape(-1, 3)
        out = np.empty((q.shape[0], 4), dtype=np.float32)
        lib().orc_scene_sample_texture(self._h, C.c_uint32(tex), fptr(q), C.c_uint32(q.shape[0]), fptr(out))
        return out

    def camera_rays(self, width, height, frame_index=0):
        rays = np.empty(width * height, dtype=A.RAY_DTYPE)
        lib().orc_generate_camera_rays(self._h, C.c_uint32(width), C.c_uint32(height), C.c_uint32(frame_index), C.c_void_p(rays.ctypes.data))
        return rays

    def render(self, width, height, frames=1, first_frame=0, max_depth=5, rr_depth=3, images=None, rect=None,
               ground=(1, 1, 1, 1), sky=(0.5, 0.7, 1.0, 1.0), env_rotation=0.0, env_intensity=1.0, exposure=1.0,
               tonemap=(False, False, False), threads=0):
        p = RenderParams()
        p.width, p.height, p.max_depth, p.rr_depth = width, height, max_depth, rr_depth
        p.ground_color = (C.c_float * 4)(*ground); p.sky_color = (C.c_float * 4)(*sky)
        p.env_rotation_degrees, p.env_intensity, p.exposure_value = env_rotation, env_intensity, exposure
        p.enable_tonemap, p.enable_aces, p.use_simple_aces = [int(x) for x in tonemap]
        p.num_threads = threads
        if images is None:
            images = [np.zeros((height, width, 4), dtype=np.float32) for _ in range(4)]
        accum, albedo, normal, final = images
        x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
        st = RenderStats()
        lib().orc_render(self._h, C.byref(p), C.c_uint32(first_frame), C.c_uint32(frames), C.c_uint32(x0), C.c_uint32(y0),
                         C.c_uint32(x1), C.c_uint32(y1), fptr(accum), fptr(albedo), fptr(normal), fptr(final), C.byref(st))
        return images, st


def set_instancing(on=False):
    """scenes created from now on: on = RENDER_SPEC 4.5 (instanced primitives intersected in object space: what the product does with
    build option instancing = True), off (the default) = everything flattened"""
    lib().orc_set_instancing_off(C.c_int(0 if on else 1))


def trace_on_bvh(nodes_u32, tris_u32, rays, mode=0, refs_u32=None):
    rays = np.ascontiguousarray(rays, dtype=A.RAY_DTYPE)
    hits = np.empty(rays.shape[0], dtype=A.HIT_DTYPE)
    ctr = (C.c_uint64 * 2)(0, 0)
    if refs_u32 is not None and len(refs_u32):  # two-level tree (RENDER_SPEC 4.5)
        refs_u32 = np.ascontiguousarray(refs_u32)
        lib().orc_trace_rays_on_bvh4_two_level(C.c_void_p(nodes_u32.ctypes.data), C.c_uint32(nodes_u32.size // 16), C.c_void_p(tris_u32.ctypes.data),
                                               C.c_uint32(tris_u32.size // 12), C.c_void_p(refs_u32.ctypes.data), C.c_void_p(rays.ctypes.data),
                                               C.c_void_p(hits.ctypes.data), C.c_uint32(rays.shape[0]), C.c_int(mode), ctr)
        return hits, (ctr[0], ctr[1])
    fn = lib().orc_trace_rays_on_bvh4
    fn(C.c_void_p(nodes_u32.ctypes.data), C.c_uint32(nodes_u32.size // 16), C.c_void_p(tris_u32.ctypes.data),
       C.c_uint32(tris_u32.size // 12), C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data),
       C.c_uint32(rays.shape[0]), C.c_int(mode), ctr)
    return hits, (ctr[0], ctr[1])


def validate_bvh(nodes_u32, tris_u32, ref_triangles9):
    md = C.c_uint32()
    # ref_triangles9 None: containment is checked on v0, v0 + e1, v0 + e2 of the 48-B records instead of the exact vertices
    ref = np.ascontiguousarray(ref_triangles9, dtype=np.float32) if ref_triangles9 is not None else None
    fn = lib().orc_validate_bvh4
    fn.restype = C.c_int
    rc = fn(C.c_void_p(nodes_u32.ctypes.data), C.c_uint32(nodes_u32.size // 16), C.c_void_p(tris_u32.ctypes.data),
            C.c_uint32(tris_u32.size // 12), fptr(ref) if ref is not None else C.c_void_p(0), C.byref(md))
    return rc, md.value


def validate_bvh_two_level(scene, nodes_u32, tris_u32, refs_u32):
    """structural check of a product-built two-level tree (RENDER_SPEC 4.5) against the scene `scene` (an OracleScene): (code, levels)"""
    md = C.c_uint32()
    fn = lib().orc_validate_bvh4_two_level
    fn.restype = C.c_int
    refs_u32 = np.ascontiguousarray(refs_u32)
    rc = fn(scene._h, C.c_void_p(nodes_u32.ctypes.data), C.c_uint32(nodes_u32.size // 16), C.c_void_p(tris_u32.ctypes.data),
            C.c_uint32(tris_u32.size // 12), C.c_void_p(refs_u32.ctypes.data), C.c_uint32(refs_u32.size // 16), C.byref(md))
    return rc, md.value


def tile_assignment(tiles_x, tiles_y, world):
    n = tiles_x * tiles_y
    owner = np.empty(n, dtype=np.uint32); slot = np.empty(n, dtype=np.uint32)
    lib().orc_tile_assignment(C.c_uint32(tiles_x), C.c_uint32(tiles_y), C.c_uint32(world), C.c_void_p(owner.ctypes.data), C.c_void_p(slot.ctypes.data))
    return owner, slot


def probe_sincos_2pi(u):
    u = np.ascontiguousarray(u, dtype=np.float32); s = np.empty_like(u); c = np.empty_like(u)
    lib().orc_probe_sincos_2pi(fptr(u), fptr(s), fptr(c), C.c_size_t(u.size))
    return s, c


def probe_acos(x):
    x = np.ascontiguousarray(x, dtype=np.float32); o = np.empty_like(x)
    lib().orc_probe_acos(fptr(x), fptr(o), C.c_size_t(x.size))
    return o


def probe_atan2(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32); x = np.ascontiguousarray(x, dtype=np.float32); o = np.empty_like(x)
    lib().orc_probe_atan2(fptr(y), fptr(x), fptr(o), C.c_size_t(x.size))
    return o


def probe_rng(pixel_id, frame_index, n):
    o = np.empty(n, dtype=np.float32)
    lib().orc_probe_rng(C.c_uint32(pixel_id), C.c_uint32(frame_index), fptr(o), C.c_size_t(n))
    return o


def probe_log(x):
    x = np.ascontiguousarray(x, dtype=np.float32); o = np.empty_like(x)
    lib().orc_probe_log(fptr(x), fptr(o), C.c_size_t(x.size))
    return o


def probe_hg(d, g, u1, u2):
    d = np.ascontiguousarray(d, dtype=np.float32)
    u1 = np.ascontiguousarray(u1, dtype=np.float32); u2 = np.ascontiguousarray(u2, dtype=np.float32)
    o = np.empty((u1.size, 3), dtype=np.float32)
    lib().orc_probe_hg(fptr(d), C.c_float(g), fptr(u1), fptr(u2), fptr(o), C.c_size_t(u1.size))
    return o


def probe_exp_neg(x):
    x = np.ascontiguousarray(x, dtype=np.float32); o = np.empty_like(x)
    lib().orc_probe_exp_neg(fptr(x), fptr(o), C.c_size_t(x.size))
    return o
