"""HalaRenderer — host mirror of the reference's ray-tracing renderer (src/rt_renderer.rs:568-1353) over the
C ABI of libhalart.so.  Method names, argument meaning, call order and error behaviour follow the reference;
every method cites the lines it mirrors.  All compute happens in the HIP library.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A


class HalaRenderer:
    """reference: `pub struct HalaRenderer` src/rt_renderer.rs:568-617."""

    RAYGEN, MISS, CALLABLE = 0, 1, 2  # shader stages accepted by push_general_shader

    def __init__(self, name, width, height, max_depth, rr_depth, enable_tonemap, enable_aces, use_simple_aces,
                 max_frames, device_ordinal=0):
        """HalaRenderer::new (src/rt_renderer.rs:650-813). `width`/`height` stand for gpu_req.{width,height}
        (:661-662); the winit window is replaced by `device_ordinal` (headless)."""
        from . import check, load_library
        self._lib = load_library()
        self._check = check
        self._h = C.c_void_p()
        check(self._lib.hala_rt_create(name.encode(), C.c_uint32(width), C.c_uint32(height), C.c_int(device_ordinal),
                                       C.c_uint32(max_depth), C.c_uint32(rr_depth), C.c_int(bool(enable_tonemap)),
                                       C.c_int(bool(enable_aces)), C.c_int(bool(use_simple_aces)),
                                       C.c_uint64(max_frames), C.byref(self._h)))
        self.width, self.height = width, height

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self):
        """Drop order of src/rt_renderer.rs:620-633."""
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hala_rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- shaders (accepted, validated, recorded; SPIR-V has no meaning for the HIP integrator) -----------
    def push_general_shader(self, code: bytes, stage, group_type=None, debug_name=""):
        """src/rt_renderer.rs:925-957"""
        self._check(self._lib.hala_rt_push_general_shader(self._h, code, C.c_size_t(len(code)), C.c_int(stage), debug_name.encode()))

    def push_general_shader_with_file(self, file_path, stage, group_type=None, debug_name=""):
        """src/rt_renderer.rs:965-995"""
        self._check(self._lib.hala_rt_push_general_shader_with_file(self._h, os.fsencode(file_path), C.c_int(stage), debug_name.encode()))

    def push_hit_shaders(self, closest_code=None, any_hit_code=None, intersection_code=None, debug_name=""):
        """src/rt_renderer.rs:1003-1046"""
        def arg(b):
            return (b, C.c_size_t(len(b))) if b else (None, C.c_size_t(0))
        c, a, i = arg(closest_code), arg(any_hit_code), arg(intersection_code)
        self._check(self._lib.hala_rt_push_hit_shaders(self._h, c[0], c[1], a[0], a[1], i[0], i[1], debug_name.encode()))

    def push_hit_shaders_with_file(self, closest_hit_path=None, any_hit_path=None, intersection_path=None, debug_name=""):
        """src/rt_renderer.rs:1056-1112"""
        enc = lambda p: os.fsencode(p) if p else None  # noqa: E731
        self._check(self._lib.hala_rt_push_hit_shaders_with_file(self._h, enc(closest_hit_path), enc(any_hit_path), enc(intersection_path), debug_name.encode()))

    def load_blue_noise_texture(self, path_or_rgba8):
        """src/rt_renderer.rs:1117-1156: a PNG / JPEG path like the reference, or decoded RGBA8 pixels [H,W,4]; optional in this integrator"""
        if isinstance(path_or_rgba8, (str, bytes, os.PathLike)):
            self._check(self._lib.hala_rt_load_blue_noise_texture(self._h, os.fsencode(path_or_rgba8)))
            return
        px = np.ascontiguousarray(path_or_rgba8, dtype=np.uint8)
        self._check(self._lib.hala_rt_load_blue_noise_pixels(self._h, px.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint32(px.shape[1]), C.c_uint32(px.shape[0])))

    # -- scene / environment ------------------------------------------------------------------------------
    def set_scene(self, scene_in_cpu):
        """src/rt_renderer.rs:1161-1178"""
        if hasattr(scene_in_cpu, "desc_ptr"):  # a scene the library loaded itself (NativeScene: hala_scene_load_gltf)
            self._check(self._lib.hala_rt_set_scene(self._h, scene_in_cpu.desc_ptr()))
            return
        holder = scene_in_cpu.to_desc()
        self._check(self._lib.hala_rt_set_scene(self._h, holder.ptr()))

    def set_envmap(self, path_or_pixels, rotation=0.0):
        """src/rt_renderer.rs:1184-1195. A str/PathLike goes through the library's decoder (.hdr / .pfm);
        an ndarray [H,W,3|4] float32 is the already decoded image."""
        if isinstance(path_or_pixels, (str, bytes, os.PathLike)):
            self._check(self._lib.hala_rt_set_envmap_file(self._h, os.fsencode(path_or_pixels), C.c_float(rotation)))
            return
        px = np.ascontiguousarray(path_or_pixels, dtype=np.float32)
        h, w, ch = px.shape
        self._check(self._lib.hala_rt_set_envmap_pixels(self._h, px.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(ch), C.c_uint32(w), C.c_uint32(h), C.c_float(rotation)))

    def set_ground_color(self, color):
        """src/rt_renderer.rs:1199-1201"""
        self._lib.hala_rt_set_ground_color(self._h, (C.c_float * 4)(*color))

    def set_sky_color(self, color):
        """src/rt_renderer.rs:1205-1207"""
        self._lib.hala_rt_set_sky_color(self._h, (C.c_float * 4)(*color))

    def set_env_intensity(self, intensity):
        """src/rt_renderer.rs:1211-1213"""
        self._lib.hala_rt_set_env_intensity(self._h, C.c_float(intensity))

    def set_exposure_value(self, exposure_value):
        """src/rt_renderer.rs:1217-1219"""
        self._lib.hala_rt_set_exposure_value(self._h, C.c_float(exposure_value))

    # -- HalaRendererTrait (src/renderer.rs:210-324) ------------------------------------------------------------
    def commit(self):
        """src/rt_renderer.rs:136-379"""
        self._check(self._lib.hala_rt_commit(self._h))

    BUILDERS = {None: 0, "auto": 0, "sah": 1, "ploc": 2, "lbvh": 3}

    def set_build_options(self, builder=None, ploc_tail=0, ploc_look_every=0, collapse_look_every=0, instancing=None):
        """how the next commit() builds the acceleration structure (hala_rt_set_build_options): builder = None | "sah" | "ploc" | "lbvh";
        instancing = True: two-level tree (RENDER_SPEC 4.5: primitives referenced by several instances are stored once), False: every
        instance flattened to world space (one tree), None: automatic (flattened up to 2^26 triangles); the other fields only change how the
        host drives the build rounds (same tree)"""
        o = A.BuildOptions(builder=self.BUILDERS[builder], ploc_tail=ploc_tail, ploc_look_every=ploc_look_every, collapse_look_every=collapse_look_every,
                           instancing=0 if instancing is None else (2 if instancing else 1))
        self._check(self._lib.hala_rt_set_build_options(self._h, C.byref(o)))

    def update(self, delta_time=0.0, width=None, height=None, ui_fn=None):
        """src/rt_renderer.rs:387-471 — one sample per pixel; `ui_fn` is dropped."""
        self._check(self._lib.hala_rt_update(self._h, C.c_double(delta_time), C.c_uint32(width or self.width), C.c_uint32(height or self.height)))

    def update_batch(self, frames: int):
        """`frames` update()s as one wavefront pass (bit-identical result; fewer, larger launches)"""
        self._check(self._lib.hala_rt_update_batch(self._h, C.c_uint32(frames)))

    def render(self):
        """src/rt_renderer.rs:475-502: nothing to present; bounds the updates in flight to two (it does not flush: read_image,
        save_images, statistics and wait_idle wait for the stream themselves)"""
        self._check(self._lib.hala_rt_render(self._h))

    def wait_idle(self):
        """src/renderer.rs:251-256"""
        self._check(self._lib.hala_rt_wait_idle(self._h))

    def save_images(self, path):
        """src/rt_renderer.rs:1224-1352"""
        self._check(self._lib.hala_rt_save_images(self._h, os.fsencode(path)))

    def info(self):
        """src/renderer.rs:212"""
        i = A.RtInfo()
        self._check(self._lib.hala_rt_get_info(self._h, C.byref(i)))
        return i

    def statistics(self):
        """src/renderer.rs:218, :135-207"""
        s = A.RtStatistics()
        self._check(self._lib.hala_rt_get_statistics(self._h, C.byref(s)))
        return s

    def reset_accumulation(self):
        """HalaRendererStatistics::reset (src/renderer.rs:168-174): the next update() renders frame_index 0"""
        self._check(self._lib.hala_rt_reset_accumulation(self._h))

    def set_counting(self, enable: bool):
        """count BVH nodes visited / triangles tested in update() (inputs of the algorithmic-bytes figure)"""
        self._check(self._lib.hala_rt_set_counting(self._h, C.c_int(bool(enable))))

    def set_pass_fusion(self, mode):
        """0: one launch per pass; 1 (default): fused launches except in timed updates; 2: always (timed updates fill traverse_fused_*)"""
        self._check(self._lib.hala_rt_set_pass_fusion(self._h, C.c_uint32(mode)))

    def set_launch_timing_period(self, period):
        """per-launch HIP events on every `period`-th update (1: all, the default; 0: none) — see include/halart.h"""
        self._check(self._lib.hala_rt_set_launch_timing_period(self._h, C.c_uint32(period)))

    # -- read-back used by tests and bench (what save_images downloads, :1239-1254) -------------------------------
    ACCUM, ALBEDO, NORMAL, FINAL = 0, 1, 2, 3

    def read_image(self, which=0) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.hala_rt_read_image(self._h, C.c_int(which), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def global_uniform(self) -> A.GlobalUniform:
        u = A.GlobalUniform()
        self._check(self._lib.hala_rt_get_global_uniform(self._h, C.byref(u)))
        return u

    def packed_cameras(self):
        buf = (A.GpuCamera * A.MAX_CAMERA_COUNT)(); n = C.c_uint32()
        self._check(self._lib.hala_rt_get_packed_cameras(self._h, buf, C.c_uint32(A.MAX_CAMERA_COUNT), C.byref(n)))
        return list(buf[:n.value])

    def packed_lights(self):
        buf = (A.GpuLight * A.MAX_LIGHT_COUNT)(); bb = (A.Aabb * A.MAX_LIGHT_COUNT)(); n = C.c_uint32()
        self._check(self._lib.hala_rt_get_packed_lights(self._h, buf, bb, C.c_uint32(A.MAX_LIGHT_COUNT), C.byref(n)))
        return list(buf[:n.value]), list(bb[:n.value])

    def packed_materials(self, capacity=4096):
        buf = (A.GpuMaterial * capacity)(); n = C.c_uint32()
        self._check(self._lib.hala_rt_get_packed_materials(self._h, buf, C.c_uint32(capacity), C.byref(n)))
        return list(buf[:min(n.value, capacity)])

    def packed_primitives(self, capacity=65536):
        buf = (A.GpuMeshData * capacity)(); t = np.zeros((capacity, 12), dtype=np.float32); n = C.c_uint32()
        self._check(self._lib.hala_rt_get_packed_primitives(self._h, buf, t.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(capacity), C.byref(n)))
        k = min(n.value, capacity)
        return list(buf[:k]), t[:k]

    def env_distribution(self, width, height):
        total = C.c_float(); marg = np.empty(height, dtype=np.float32); cond = np.empty((height, width), dtype=np.float32)
        self._check(self._lib.hala_rt_get_env_distribution(self._h, C.byref(total), marg.ctypes.data_as(C.POINTER(C.c_float)), cond.ctypes.data_as(C.POINTER(C.c_float))))
        return total.value, marg, cond

    # -- textures (set 2 binding 0) ---------------------------------------------------------------------------------------
    def texture_info(self, texture):
        w, h, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.hala_rt_get_texture_info(self._h, C.c_uint32(texture), C.byref(w), C.byref(h), C.byref(m)))
        return w.value, h.value, m.value

    def read_texture_level(self, texture, level):
        w, h, _ = self.texture_info(texture)
        lw, lh = max(1, w >> level), max(1, h >> level)
        out = np.empty((lh, lw, 4), dtype=np.float32)
        self._check(self._lib.hala_rt_read_texture_level(self._h, C.c_uint32(texture), C.c_uint32(level), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def sample_texture(self, texture, uv_lod):
        q = np.ascontiguousarray(uv_lod, dtype=np.float32).reshape(-1, 3)
        out = np.empty((q.shape[0], 4), dtype=np.float32)
        self._check(self._lib.hala_rt_sample_texture_host(self._h, C.c_uint32(texture), q.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(q.shape[0]),
                                                          out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- ray-batch operator / BVH introspection ----------------------------------------------------------------------
    def trace_rays_host(self, rays: np.ndarray, mode=0, count_steps=False):
        rays = np.ascontiguousarray(rays, dtype=A.RAY_DTYPE)
        hits = np.empty(rays.shape[0], dtype=A.HIT_DTYPE)
        ctr = (C.c_uint64 * 2)(0, 0)
        self._check(self._lib.hala_rt_trace_rays_host(self._h, C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_uint32(rays.shape[0]), C.c_int(mode), ctr if count_steps else None))
        return (hits, (ctr[0], ctr[1])) if count_steps else hits

    def trace_rays(self, d_rays: int, d_hits: int, count: int, mode=0, d_counters: int = 0, stream: int = 0):
        """device-pointer form: the vkCmdTraceRaysKHR analogue for one ray batch (src/rt_renderer.rs:458-464)"""
        self._check(self._lib.hala_rt_trace_rays(self._h, C.c_void_p(d_rays), C.c_void_p(d_hits), C.c_uint32(count), C.c_int(mode), C.c_void_p(d_counters), C.c_void_p(stream)))

    def bvh_info(self) -> A.BvhInfo:
        i = A.BvhInfo()
        self._check(self._lib.hala_rt_get_bvh_info(self._h, C.byref(i)))
        return i

    def download_bvh(self):
        i = self.bvh_info()
        nodes = np.empty(i.node_count * 16, dtype=np.uint32)
        tris = np.empty(max(i.stored_triangle_count, 1) * 12, dtype=np.uint32)
        self._check(self._lib.hala_rt_download_bvh(self._h, C.c_void_p(nodes.ctypes.data), C.c_void_p(tris.ctypes.data)))
        return nodes, tris[: i.stored_triangle_count * 12]

    def download_instance_refs(self):
        """two-level trees: the 64-B records the instance leaves index ([n, 16] uint32: 12 floats world -> object, root node, global id
        of the first triangle, first shading record, instance index); empty for one-level trees"""
        n = C.c_uint32(0)
        self._check(self._lib.hala_rt_download_instance_refs(self._h, None, C.c_uint32(0), C.byref(n)))
        refs = np.empty((max(n.value, 1), 16), dtype=np.uint32)
        self._check(self._lib.hala_rt_download_instance_refs(self._h, C.c_void_p(refs.ctypes.data), C.c_uint32(n.value), C.byref(n)))
        return refs[: n.value]

    def update_node_transform(self, node_index, local_transform):
        m = np.asarray(local_transform, dtype=np.float32)
        self._check(self._lib.hala_rt_update_node_transform(self._h, C.c_uint32(node_index), (C.c_float * 16)(*m.T.reshape(-1).tolist())))

    def update_vertices(self, mesh_index, primitive_index, vertices):
        """deforming geometry: new vertices (VERTEX_DTYPE records, same count) for one primitive; applied by the next refit()"""
        v = np.ascontiguousarray(vertices, dtype=A.VERTEX_DTYPE)
        self._check(self._lib.hala_rt_update_vertices(self._h, C.c_uint32(mesh_index), C.c_uint32(primitive_index), C.c_void_p(v.ctypes.data), C.c_uint32(v.shape[0])))

    def update_material(self, material_index, material):
        """replace one cpu::HalaMaterial of the scene (interactive material edits); applied by the next refit()"""
        from .scene import fill_material_desc
        d = A.MaterialDesc()
        fill_material_desc(d, material)
        self._check(self._lib.hala_rt_update_material(self._h, C.c_uint32(material_index), C.byref(d)))

    def refit(self):
        self._check(self._lib.hala_rt_refit(self._h))

    # -- multi-GPU tile sharding ---------------------------------------------------------------------------------------
    def set_tile_shard(self, rank, world, tile_size=32):
        self._check(self._lib.hala_rt_set_tile_shard(self._h, C.c_uint32(rank), C.c_uint32(world), C.c_uint32(tile_size)))

    def stream_handle(self):
        """the renderer's hipStream_t as an integer (torch.cuda.ExternalStream(handle) wraps it)"""
        p = C.c_void_p()
        self._check(self._lib.hala_rt_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def tile_buffer(self, which=0):
        p = C.c_void_p(); n = C.c_size_t()
        self._check(self._lib.hala_rt_tile_buffer(self._h, C.c_int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- the exchange step inside the library (RCCL; include/halart.h "hala_rt_comm_*", "hala_rt_tile_allgather*") --------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: made on rank 0, handed to the other ranks over any host channel"""
        from . import check, load_library
        buf = (C.c_uint8 * 128)()
        check(load_library().hala_rt_comm_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, unique_id: bytes, rank: int, world: int):
        self._check(self._lib.hala_rt_comm_init_rank(self._h, (C.c_uint8 * 128)(*unique_id), C.c_uint32(rank), C.c_uint32(world)))

    def comm_attach(self, nccl_comm: int):
        self._check(self._lib.hala_rt_comm_attach(self._h, C.c_void_p(nccl_comm)))

    def comm_destroy(self):
        self._check(self._lib.hala_rt_comm_destroy(self._h))

    def tile_allgather(self, aovs=(0,)):
        self._check(self._lib.hala_rt_tile_allgather(self._h, C.c_uint32(sum(1 << a for a in aovs))))

    def tile_allgather_begin(self, aovs=(0,)):
        self._check(self._lib.hala_rt_tile_allgather_begin(self._h, C.c_uint32(sum(1 << a for a in aovs))))

    def tile_allgather_finish(self):
        self._check(self._lib.hala_rt_tile_allgather_finish(self._h))

    def tile_allgather_begin_external(self, aovs=(0,)):
        """the pipeline of tile_allgather_begin with the exchange left to the caller (exchange_buffers); no communicator needed"""
        self._check(self._lib.hala_rt_tile_allgather_begin_external(self._h, C.c_uint32(sum(1 << a for a in aovs))))

    def exchange_buffers(self, which=0):
        """-> (staged ptr, staged bytes, receive ptr, receive bytes, exchange hipStream_t) of the exchange in flight"""
        a = C.c_void_p(); na = C.c_size_t(); b = C.c_void_p(); nb = C.c_size_t(); st = C.c_void_p()
        self._check(self._lib.hala_rt_get_exchange_buffers(self._h, C.c_int(which), C.byref(a), C.byref(na), C.byref(b), C.byref(nb), C.byref(st)))
        return a.value, na.value, b.value, nb.value, st.value

    def gathered_buffer(self, which=0):
        p = C.c_void_p(); n = C.c_size_t()
        self._check(self._lib.hala_rt_get_gathered_buffer(self._h, C.c_int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    def scatter_gathered_tiles(self, which, d_gathered: int, nbytes: int, stream: int = 0):
        """de-interleave a gathered buffer into this renderer's row-major image; stream = a hipStream_t of the caller's (0: the renderer's)"""
        self._check(self._lib.hala_rt_scatter_gathered_tiles_on_stream(self._h, C.c_int(which), C.c_void_p(d_gathered), C.c_size_t(nbytes), C.c_void_p(stream or None)))
