"""ctypes mirror of include/halart.h (the C-ABI boundary of libhalart.so).

Every Structure here is checked against the byte sizes of the reference's #[repr(C)] records
 in tests/test_layouts.py.  Nothing in this module touches the GPU.
"""
import ctypes as C

INVALID_INDEX = 0xFFFFFFFF  # u32::MAX, reference: src/scene/cpu/node.rs:23-25
MAX_CAMERA_COUNT = 8        # reference: src/scene/loader/gpu_uploader.rs:39
MAX_LIGHT_COUNT = 32        # reference: src/scene/loader/gpu_uploader.rs:40


class Vertex(C.Structure):  # src/scene/vertex.rs:2-9, 44 B
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3),
                ("tangent", C.c_float * 3), ("tex_coord", C.c_float * 2)]


class GpuCamera(C.Structure):  # src/scene/gpu/camera.rs:10-20, 80 B
    _fields_ = [("position", C.c_float * 3), ("_pad0", C.c_float),
                ("right", C.c_float * 3), ("_pad1", C.c_float),
                ("up", C.c_float * 3), ("_pad2", C.c_float),
                ("forward", C.c_float * 3), ("yfov", C.c_float),
                ("focal_distance_or_xmag", C.c_float), ("aperture_or_ymag", C.c_float),
                ("type", C.c_uint32), ("_pad3", C.c_uint32)]


class GpuLight(C.Structure):  # src/scene/gpu/light.rs:7-32, 80 B
    _fields_ = [("intensity", C.c_float * 3), ("_pad0", C.c_float),
                ("position", C.c_float * 3), ("_pad1", C.c_float),
                ("u", C.c_float * 3), ("_pad2", C.c_float),
                ("v", C.c_float * 3), ("radius", C.c_float), ("area", C.c_float),
                ("type", C.c_uint32), ("_pad3", C.c_uint32 * 2)]


class Aabb(C.Structure):  # HalaAABB, gpu_uploader.rs:169-180, 24 B
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3)]


class GpuMaterial(C.Structure):  # src/scene/gpu/material.rs:6-48, 144 B
    _fields_ = [("medium_color", C.c_float * 3), ("medium_density", C.c_float),
                ("medium_anisotropy", C.c_float), ("medium_type", C.c_uint32),
                ("_medium_padding", C.c_float * 2),
                ("base_color", C.c_float * 3), ("opacity", C.c_float),
                ("emission", C.c_float * 3), ("anisotropic", C.c_float),
                ("metallic", C.c_float), ("roughness", C.c_float),
                ("subsurface", C.c_float), ("specular_tint", C.c_float),
                ("sheen", C.c_float), ("sheen_tint", C.c_float),
                ("clearcoat", C.c_float), ("clearcoat_roughness", C.c_float),
                ("clearcoat_tint", C.c_float * 3), ("specular_transmission", C.c_float),
                ("ior", C.c_float), ("ax", C.c_float), ("ay", C.c_float),
                ("base_color_map_index", C.c_uint32), ("normal_map_index", C.c_uint32),
                ("metallic_roughness_map_index", C.c_uint32), ("emission_map_index", C.c_uint32),
                ("type", C.c_uint32)]


class GpuMeshData(C.Structure):  # src/scene/gpu/mesh.rs:32-39, 96 B
    _fields_ = [("transform", C.c_float * 16), ("material_index", C.c_uint32), ("_pad0", C.c_uint32),
                ("vertices", C.c_uint64), ("indices", C.c_uint64), ("_pad1", C.c_uint64)]


class GlobalUniform(C.Structure):  # src/rt_renderer.rs:44-65, 112 B
    _fields_ = [("ground_color", C.c_float * 4), ("sky_color", C.c_float * 4),
                ("resolution", C.c_float * 2), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32),
                ("frame_index", C.c_uint32), ("camera_index", C.c_uint32), ("env_type", C.c_uint32),
                ("env_map_width", C.c_uint32), ("env_map_height", C.c_uint32),
                ("env_total_sum", C.c_float), ("env_rotation", C.c_float), ("env_intensity", C.c_float),
                ("exposure_value", C.c_float), ("enable_tonemap", C.c_uint32), ("enable_aces", C.c_uint32),
                ("use_simple_aces", C.c_uint32), ("num_of_lights", C.c_uint32), ("_pad", C.c_uint32 * 3)]


class NodeDesc(C.Structure):  # src/scene/cpu/node.rs:2-12
    _fields_ = [("name", C.c_char_p), ("parent", C.c_int32), ("local_transform", C.c_float * 16),
                ("mesh_index", C.c_uint32), ("camera_index", C.c_uint32), ("light_index", C.c_uint32)]


class PrimitiveDesc(C.Structure):  # src/scene/cpu/mesh.rs:6-13
    _fields_ = [("indices", C.POINTER(C.c_uint32)), ("index_count", C.c_uint32),
                ("vertices", C.POINTER(Vertex)), ("vertex_count", C.c_uint32),
                ("material_index", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("primitives", C.POINTER(PrimitiveDesc)), ("primitive_count", C.c_uint32)]


class MaterialDesc(C.Structure):  # src/scene/cpu/material.rs:24-50, :75-80
    _fields_ = [("type", C.c_uint32), ("base_color", C.c_float * 3), ("opacity", C.c_float),
                ("emission", C.c_float * 3), ("anisotropic", C.c_float), ("metallic", C.c_float),
                ("roughness", C.c_float), ("subsurface", C.c_float), ("specular_tint", C.c_float),
                ("sheen", C.c_float), ("sheen_tint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoat_roughness", C.c_float), ("clearcoat_tint", C.c_float * 3),
                ("specular_transmission", C.c_float), ("ior", C.c_float),
                ("medium_type", C.c_uint32), ("medium_color", C.c_float * 3),
                ("medium_density", C.c_float), ("medium_anisotropy", C.c_float),
                ("base_color_map_index", C.c_uint32), ("emission_map_index", C.c_uint32),
                ("normal_map_index", C.c_uint32), ("metallic_roughness_map_index", C.c_uint32)]


class LightDesc(C.Structure):  # src/scene/cpu/light.rs:30-39
    _fields_ = [("color", C.c_float * 3), ("intensity", C.c_float), ("light_type", C.c_uint32),
                ("param0", C.c_float), ("param1", C.c_float)]


class CameraDesc(C.Structure):  # src/scene/cpu/camera.rs:4-29
    _fields_ = [("type", C.c_uint32), ("aspect", C.c_float), ("yfov", C.c_float), ("znear", C.c_float),
                ("zfar", C.c_float), ("focal_distance", C.c_float), ("aperture", C.c_float),
                ("xmag", C.c_float), ("ymag", C.c_float)]


class ImageDesc(C.Structure):  # src/scene/cpu/image_data.rs:14-20
    _fields_ = [("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("data", C.c_void_p), ("num_of_bytes", C.c_size_t)]


class IndexPair(C.Structure):
    _fields_ = [("key", C.c_uint32), ("value", C.c_uint32)]


class SceneDesc(C.Structure):  # src/scene/cpu/scene.rs:17-26
    _fields_ = [("nodes", C.POINTER(NodeDesc)), ("node_count", C.c_uint32),
                ("meshes", C.POINTER(MeshDesc)), ("mesh_count", C.c_uint32),
                ("materials", C.POINTER(MaterialDesc)), ("material_count", C.c_uint32),
                ("lights", C.POINTER(LightDesc)), ("light_count", C.c_uint32),
                ("cameras", C.POINTER(CameraDesc)), ("camera_count", C.c_uint32),
                ("texture2image_mapping", C.POINTER(IndexPair)), ("texture_count", C.c_uint32),
                ("image2data_mapping", C.POINTER(IndexPair)), ("image_count", C.c_uint32),
                ("image_data", C.POINTER(ImageDesc)), ("image_data_count", C.c_uint32)]


class Ray(C.Structure):  # 32 B
    _fields_ = [("origin", C.c_float * 3), ("tmin", C.c_float), ("direction", C.c_float * 3), ("tmax", C.c_float)]


class Hit(C.Structure):  # 16 B
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("prim", C.c_uint32)]


class RtInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class RtStatistics(C.Structure):
    _fields_ = [("total_frames", C.c_uint64), ("last_gpu_ms", C.c_double), ("rays_last_update", C.c_uint64),
                ("rays_total", C.c_uint64), ("traverse_ms_last_update", C.c_double),
                ("gpu_ms_total", C.c_double), ("traverse_closest_ms_total", C.c_double),
                ("traverse_shadow_ms_total", C.c_double), ("traverse_closest_launches", C.c_uint64),
                ("traverse_shadow_launches", C.c_uint64), ("updates_rendered", C.c_uint64),
                ("rays_closest_total", C.c_uint64), ("rays_shadow_total", C.c_uint64),
                ("nodes_closest_total", C.c_uint64), ("tris_closest_total", C.c_uint64),
                ("nodes_shadow_total", C.c_uint64), ("tris_shadow_total", C.c_uint64),
                ("rays_closest_counted", C.c_uint64), ("rays_shadow_counted", C.c_uint64),
                ("wave_steps_closest_total", C.c_uint64), ("leaf_passes_closest_total", C.c_uint64), ("leaf_lanes_closest_total", C.c_uint64),
                ("wave_steps_shadow_total", C.c_uint64), ("leaf_passes_shadow_total", C.c_uint64), ("leaf_lanes_shadow_total", C.c_uint64),
                ("traverse_primary_ms_total", C.c_double), ("traverse_primary_launches", C.c_uint64),
                ("nodes_primary_total", C.c_uint64), ("tris_primary_total", C.c_uint64), ("rays_primary_counted", C.c_uint64),
                ("rays_primary_total", C.c_uint64),
                ("rays_closest_timed", C.c_uint64), ("rays_primary_timed", C.c_uint64), ("rays_shadow_timed", C.c_uint64),
                ("shade_ms_total", C.c_double), ("shade_launches", C.c_uint64),
                ("traverse_fused_ms_total", C.c_double), ("traverse_fused_launches", C.c_uint64),
                ("rays_fused_closest_timed", C.c_uint64), ("rays_fused_shadow_timed", C.c_uint64)]


class BuildOptions(C.Structure):  # hala_rt_build_options
    _fields_ = [("builder", C.c_uint32), ("ploc_tail", C.c_uint32), ("ploc_look_every", C.c_uint32),
                ("collapse_look_every", C.c_uint32), ("instancing", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class BvhInfo(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("triangle_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("lds_node_count", C.c_uint32), ("scene_min", C.c_float * 3), ("scene_max", C.c_float * 3),
                ("node_width", C.c_uint32), ("stored_triangle_count", C.c_uint32), ("instance_node_count", C.c_uint32),
                ("instance_ref_count", C.c_uint32), ("tree_bytes", C.c_uint64)]


class RtProgDescInfo(C.Structure):
    _fields_ = [("raygen_count", C.c_uint32), ("miss_count", C.c_uint32), ("hit_count", C.c_uint32),
                ("callable_count", C.c_uint32), ("push_constant_size", C.c_uint32),
                ("binding_count", C.c_uint32), ("ray_recursion_depth", C.c_uint32)]


# numpy dtypes of the batch records
import numpy as _np

RAY_DTYPE = _np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])
HIT_DTYPE = _np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4")])
VERTEX_DTYPE = _np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3), ("tex_coord", "<f4", 2)])

# every symbol include/halart.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = [
    "hala_last_error_message", "hala_rt_create", "hala_rt_destroy",
    "hala_rt_push_general_shader", "hala_rt_push_general_shader_with_file",
    "hala_rt_push_hit_shaders", "hala_rt_push_hit_shaders_with_file",
    "hala_rt_load_blue_noise_texture", "hala_rt_load_blue_noise_pixels", "hala_rt_set_scene", "hala_rt_set_envmap_pixels",
    "hala_rt_set_envmap_file", "hala_rt_set_ground_color", "hala_rt_set_sky_color",
    "hala_rt_set_env_intensity", "hala_rt_set_exposure_value", "hala_rt_commit", "hala_rt_set_build_options", "hala_rt_update", "hala_rt_update_batch",
    "hala_rt_render", "hala_rt_wait_idle", "hala_rt_save_images", "hala_rt_read_image",
    "hala_rt_get_info", "hala_rt_get_statistics", "hala_rt_set_counting", "hala_rt_set_launch_timing_period", "hala_rt_set_pass_fusion", "hala_rt_reset_accumulation", "hala_rt_get_global_uniform",
    "hala_rt_get_packed_cameras", "hala_rt_get_packed_lights", "hala_rt_get_packed_materials",
    "hala_rt_get_packed_primitives", "hala_rt_get_env_distribution", "hala_rt_get_texture_info",
    "hala_rt_read_texture_level", "hala_rt_sample_texture_host", "hala_rt_set_tile_shard",
    "hala_rt_tile_buffer", "hala_rt_get_stream", "hala_rt_scatter_gathered_tiles", "hala_rt_scatter_gathered_tiles_on_stream", "hala_rt_trace_rays",
    "hala_rt_trace_rays_host", "hala_rt_trace_rays_indirect", "hala_rt_get_bvh_info", "hala_rt_download_bvh", "hala_rt_download_instance_refs",
    "hala_rt_update_node_transform", "hala_rt_update_vertices", "hala_rt_update_material", "hala_rt_refit", "hala_envmap_build_distribution",
    "hala_tonemap_pixels", "hala_write_pfm", "hala_rtprog_parse_desc", "hala_version",
    "hala_scene_load_gltf", "hala_scene_get_desc", "hala_scene_free", "hala_load_float_image",
    "hala_rtprog_create", "hala_rtprog_destroy", "hala_rtprog_get_desc_info", "hala_rtprog_bind", "hala_rtprog_push_constants",
    "hala_rtprog_push_constants_f32", "hala_rtprog_trace_rays", "hala_rtprog_trace_rays_indirect",
    "hala_rt_comm_unique_id", "hala_rt_comm_init_rank", "hala_rt_comm_attach", "hala_rt_comm_destroy",
    "hala_rt_tile_allgather", "hala_rt_tile_allgather_begin", "hala_rt_tile_allgather_finish", "hala_rt_get_gathered_buffer",
    "hala_rt_tile_allgather_begin_external", "hala_rt_get_exchange_buffers",
]
