"""CPU tier: record layouts (SURVEY.md §8a sizes/offsets) and the C-ABI surface of libhalart.so.
No compute call is made here — the library is only dlopen'ed and its symbol table checked."""
import ctypes as C
import os
import re

import pytest

from hala_renderer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_record_sizes():
    # sizes verified against glam's x86-64 alignments in SURVEY.md §8a
    assert C.sizeof(A.Vertex) == 44             # src/scene/vertex.rs:2-9
    assert C.sizeof(A.GpuCamera) == 80          # src/scene/gpu/camera.rs:10-20
    assert C.sizeof(A.GpuLight) == 80           # src/scene/gpu/light.rs:7-32
    assert C.sizeof(A.Aabb) == 24
    assert C.sizeof(A.GpuMaterial) == 144       # src/scene/gpu/material.rs:6-48
    assert C.sizeof(A.GpuMeshData) == 96        # src/scene/gpu/mesh.rs:32-39
    assert C.sizeof(A.GlobalUniform) == 112     # src/rt_renderer.rs:44-65
    assert C.sizeof(A.Ray) == 32 and C.sizeof(A.Hit) == 16


def test_reference_record_offsets():
    cam = A.GpuCamera
    assert (cam.position.offset, cam.right.offset, cam.up.offset, cam.forward.offset) == (0, 16, 32, 48)
    assert (cam.yfov.offset, cam.focal_distance_or_xmag.offset, cam.aperture_or_ymag.offset, cam.type.offset) == (60, 64, 68, 72)
    li = A.GpuLight
    assert (li.intensity.offset, li.position.offset, li.u.offset, li.v.offset, li.radius.offset, li.area.offset, li.type.offset) == (0, 16, 32, 48, 60, 64, 68)
    m = A.GpuMaterial
    expect = dict(medium_color=0, medium_density=12, medium_anisotropy=16, medium_type=20, base_color=32, opacity=44, emission=48,
                  anisotropic=60, metallic=64, roughness=68, subsurface=72, specular_tint=76, sheen=80, sheen_tint=84,
                  clearcoat=88, clearcoat_roughness=92, clearcoat_tint=96, specular_transmission=108, ior=112, ax=116, ay=120,
                  base_color_map_index=124, normal_map_index=128, metallic_roughness_map_index=132, emission_map_index=136, type=140)
    for k, v in expect.items():
        assert getattr(m, k).offset == v, k
    md = A.GpuMeshData
    assert (md.transform.offset, md.material_index.offset, md.vertices.offset, md.indices.offset) == (0, 64, 72, 80)
    u = A.GlobalUniform
    expect = dict(ground_color=0, sky_color=16, resolution=32, max_depth=40, rr_depth=44, frame_index=48, camera_index=52, env_type=56,
                  env_map_width=60, env_map_height=64, env_total_sum=68, env_rotation=72, env_intensity=76, exposure_value=80,
                  enable_tonemap=84, enable_aces=88, use_simple_aces=92, num_of_lights=96)
    for k, v in expect.items():
        assert getattr(u, k).offset == v, k


def test_numpy_dtypes_match_ctypes():
    assert A.VERTEX_DTYPE.itemsize == C.sizeof(A.Vertex)
    assert A.RAY_DTYPE.itemsize == C.sizeof(A.Ray)
    assert A.HIT_DTYPE.itemsize == C.sizeof(A.Hit)


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "halart.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(hala_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_header_and_export_list_agree():
    assert _declared_functions() == sorted(A.EXPORTS)


def test_library_exports_every_declared_symbol(halart):
    lib = C.CDLL(halart.LIB_PATH)
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert b"halart" in halart.load_library().hala_version()


def test_oracle_layouts_match(oracle):
    """the oracle declares its own copies of the records; they must agree with the ABI (same ctypes objects are passed to both)"""
    src = open(os.path.join(ROOT, "oracle", "oracle_api.h")).read()
    for name, size in [("orc_vertex", 44), ("orc_gpu_camera", 80), ("orc_gpu_light", 80), ("orc_aabb", 24), ("orc_gpu_material", 144),
                       ("orc_gpu_mesh_data", 96), ("orc_global_uniform", 112)]:
        assert re.search(rf"\}}\s*{name};\s*//\s*{size} B", src), name


def test_no_compute_without_gpu_fails_loudly(halart):
    """On a box without a HIP device the product must refuse to work instead of falling back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(halart.HalaRendererError):
        halart.HalaRenderer("x", 8, 8, 2, 1, False, False, False, 0)


def test_rtprog_desc_parser(halart):
    """serde field names/defaults of HalaRayTracingProgramDesc (src/raytracing_program.rs:33-55); host-only parsing"""
    d = halart.HalaRayTracingProgramDesc.from_json(
        '{"raygen_shader_file_paths": ["a.rgen.spv"], "hit_shader_file_paths": [{"closest_hit_shader_file_path": "a.rchit.spv", '
        '"any_hit_shader_file_path": null}]}')
    assert d.ray_recursion_depth == 1 and d.push_constant_size == 0 and d.miss_shader_file_paths == [] and len(d.hit_shader_file_paths) == 1
    d = halart.HalaRayTracingProgramDesc.from_json(
        '{"raygen_shader_file_paths": ["r"], "miss_shader_file_paths": ["m1", "m2"], "hit_shader_file_paths": [], '
        '"callable_shader_file_paths": ["c"], "push_constant_size": 16, "bindings": ["UNIFORM_BUFFER"], "ray_recursion_depth": 2}')
    assert (len(d.miss_shader_file_paths), d.push_constant_size, d.ray_recursion_depth) == (2, 16, 2)
    with pytest.raises(halart.HalaRendererError):  # raygen_shader_file_paths has no serde default
        halart.HalaRayTracingProgramDesc.from_json('{"hit_shader_file_paths": []}')
    with pytest.raises(halart.HalaRendererError):
        halart.HalaRayTracingProgramDesc.from_json('{"raygen_shader_file_paths": ["r"]}')
    with pytest.raises(halart.HalaRendererError):
        halart.HalaRayTracingProgramDesc.from_json('{"raygen_shader_file_paths": ["r"], "hit_shader_file_paths": [] trailing')


def test_library_has_no_load_time_dependency_on_rccl_or_roctx():
    """RCCL and the profiler's marker library are resolved on first use (csrc/dyn_api.h): a one-GPU host loads libhalart.so without
    either installed"""
    import subprocess
    import hala_renderer_amd as H
    out = subprocess.run(["readelf", "-d", H.LIB_PATH], capture_output=True, text=True, check=True).stdout
    needed = [l.split("[")[1].split("]")[0] for l in out.splitlines() if "(NEEDED)" in l]
    assert needed and not [n for n in needed if "rccl" in n or "roctx" in n or "rocprofiler" in n], needed
