"""HalaRayTracingProgram — host mirror of src/raytracing_program.rs:25-341: the generic "RT pass" object.

In the reference a program is {SPIR-V shader groups, pipeline, SBT} and `trace_rays(w, h, d)` launches
w*h*d ray-gen invocations against whatever acceleration structure the bound descriptor sets reference.  Here the
shader groups are the library's HIP traversal kernels, the "descriptor sets" are the device buffers of one ray
batch (hala_ray[] in, hala_hit[] out) and the acceleration structure is the one owned by a committed HalaRenderer.
"""
import ctypes as C
import json
from dataclasses import dataclass, field
from typing import List, Optional

from . import _abi as A


@dataclass
class HalaRayTracingHitShaderDesc:
    """src/raytracing_program.rs:25-30"""
    closest_hit_shader_file_path: Optional[str] = None
    any_hit_shader_file_path: Optional[str] = None
    intersection_shader_file_path: Optional[str] = None


@dataclass
class HalaRayTracingProgramDesc:
    """src/raytracing_program.rs:33-55 (serde field names and defaults)"""
    raygen_shader_file_paths: List[str] = field(default_factory=list)
    miss_shader_file_paths: List[str] = field(default_factory=list)
    hit_shader_file_paths: List[HalaRayTracingHitShaderDesc] = field(default_factory=list)
    callable_shader_file_paths: List[str] = field(default_factory=list)
    push_constant_size: int = 0
    bindings: List[str] = field(default_factory=list)
    ray_recursion_depth: int = 1

    def to_json(self) -> str:
        """the serde form (:33-47)"""
        return json.dumps({
            "raygen_shader_file_paths": self.raygen_shader_file_paths, "miss_shader_file_paths": self.miss_shader_file_paths,
            "hit_shader_file_paths": [{"closest_hit_shader_file_path": h.closest_hit_shader_file_path, "any_hit_shader_file_path": h.any_hit_shader_file_path,
                                       "intersection_shader_file_path": h.intersection_shader_file_path} for h in self.hit_shader_file_paths],
            "callable_shader_file_paths": self.callable_shader_file_paths, "push_constant_size": self.push_constant_size,
            "bindings": self.bindings, "ray_recursion_depth": self.ray_recursion_depth})

    @staticmethod
    def from_json(text: str) -> "HalaRayTracingProgramDesc":
        """Parses with the library's parser (hala_rtprog_parse_desc) for validation — required keys are
        `raygen_shader_file_paths` and `hit_shader_file_paths`, everything else has a serde default."""
        from . import check, load_library
        info = A.RtProgDescInfo()
        check(load_library().hala_rtprog_parse_desc(text.encode(), C.byref(info)))
        d = json.loads(text)
        desc = HalaRayTracingProgramDesc(
            raygen_shader_file_paths=list(d["raygen_shader_file_paths"]),
            miss_shader_file_paths=list(d.get("miss_shader_file_paths", [])),
            hit_shader_file_paths=[HalaRayTracingHitShaderDesc(**h) for h in d["hit_shader_file_paths"]],
            callable_shader_file_paths=list(d.get("callable_shader_file_paths", [])),
            push_constant_size=int(d.get("push_constant_size", 0)),
            bindings=list(d.get("bindings", [])),
            ray_recursion_depth=int(d.get("ray_recursion_depth", 1)),
        )
        assert len(desc.raygen_shader_file_paths) == info.raygen_count and desc.ray_recursion_depth == info.ray_recursion_depth
        return desc


class HalaRayTracingProgram:
    """src/raytracing_program.rs:70-341 — a thin ctypes twin of the library's `hala_rtprog_*` object (include/halart.h)"""

    CLOSEST_HIT, ANY_HIT = 0, 1

    def __init__(self, renderer, desc, debug_name: str = ""):
        """HalaRayTracingProgram::new (:85-252): `renderer` supplies the device and acceleration structure (logical_device +
        descriptor_set_layouts in the reference); `desc` is a HalaRayTracingProgramDesc or its serde JSON text."""
        from . import check, load_library
        self._lib = load_library()
        self._check = check
        self.renderer = renderer
        self.desc = desc if isinstance(desc, HalaRayTracingProgramDesc) else HalaRayTracingProgramDesc.from_json(desc)
        self.debug_name = debug_name
        self._h = C.c_void_p()
        text = desc if isinstance(desc, str) else self.desc.to_json()
        check(self._lib.hala_rtprog_create(renderer._h, text.encode(), debug_name.encode(), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hala_rtprog_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_pso(self):
        """:256 — the 'pipeline' is the traversal kernel pair of the library"""
        return ("halart::traverse_closest", "halart::traverse_any")

    def bind(self, d_rays: int, d_hits: int):
        """:264-278 — bind the ray batch (device addresses) in place of descriptor sets"""
        self._check(self._lib.hala_rtprog_bind(self._h, C.c_void_p(int(d_rays)), C.c_void_p(int(d_hits))))

    def push_constants(self, offset: int, data: bytes):
        """:285-300 — byte 0..3 of the constant block selects the hit mode (0 closest, 1 any)"""
        self._check(self._lib.hala_rtprog_push_constants(self._h, C.c_uint32(offset), bytes(data), C.c_size_t(len(data))))

    def push_constants_f32(self, offset: int, data):
        """:307-322"""
        arr = (C.c_float * len(data))(*data)
        self._check(self._lib.hala_rtprog_push_constants_f32(self._h, C.c_uint32(offset), arr, C.c_size_t(len(data))))

    def trace_rays(self, width: int, height: int, depth: int = 1, stream: int = 0):
        """:330-332"""
        self._check(self._lib.hala_rtprog_trace_rays(self._h, C.c_uint32(width), C.c_uint32(height), C.c_uint32(depth), C.c_void_p(stream or None)))

    def trace_rays_indirect(self, indirect_device_address: int, stream: int = 0):
        """:338-340"""
        self._check(self._lib.hala_rtprog_trace_rays_indirect(self._h, C.c_void_p(indirect_device_address), C.c_void_p(stream or None)))
