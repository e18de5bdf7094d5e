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
    """src/raytracing_program.rs:70-341"""

    CLOSEST_HIT, ANY_HIT = 0, 1

    def __init__(self, renderer, desc: HalaRayTracingProgramDesc, debug_name: str = ""):
        """HalaRayTracingProgram::new (:85-252): `renderer` supplies the device and acceleration structure
        (logical_device + descriptor_set_layouts in the reference)."""
        from . import HalaRendererError
        if not desc.raygen_shader_file_paths:
            raise HalaRendererError("The raygen shader list is empty!")
        self.renderer = renderer
        self.desc = desc
        self.debug_name = debug_name
        self._rays = self._hits = 0
        self._constants = bytearray(max(desc.push_constant_size, 4))

    def get_pso(self):
        """:256 — the 'pipeline' is the traversal kernel pair of the library"""
        return ("halart::traverse_closest", "halart::traverse_any")

    def bind(self, d_rays: int, d_hits: int):
        """:264-278 — bind the ray batch (device addresses) in place of descriptor sets"""
        self._rays, self._hits = int(d_rays), int(d_hits)

    def push_constants(self, offset: int, data: bytes):
        """:285-300 — byte 0..3 of the constant block selects the hit mode (0 closest, 1 any)"""
        from . import HalaRendererError
        if offset + len(data) > len(self._constants):
            raise HalaRendererError("push constant range exceeds push_constant_size")
        self._constants[offset:offset + len(data)] = data

    def push_constants_f32(self, offset: int, data):
        """:307-322"""
        import struct
        self.push_constants(offset, struct.pack(f"<{len(data)}f", *data))

    def _mode(self):
        return int.from_bytes(self._constants[0:4], "little") & 1

    def trace_rays(self, width: int, height: int, depth: int = 1, stream: int = 0):
        """:330-332"""
        self.renderer.trace_rays(self._rays, self._hits, width * height * depth, self._mode(), 0, stream)

    def trace_rays_indirect(self, indirect_device_address: int, stream: int = 0):
        """:338-340"""
        from . import check
        r = self.renderer
        check(r._lib.hala_rt_trace_rays_indirect(r._h, C.c_void_p(self._rays), C.c_void_p(self._hits), C.c_void_p(indirect_device_address), C.c_int(self._mode()), C.c_void_p(stream)))
