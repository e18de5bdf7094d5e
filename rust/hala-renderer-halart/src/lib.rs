//! `HalaRenderer` / `HalaRayTracingProgram` over libhalart.so — the same public names, argument meaning and error type
//! as hala-renderer's ray-tracing renderer (`src/rt_renderer.rs`, `src/raytracing_program.rs`, `src/error.rs`), so an
//! application written against the reference switches by changing one `use`.
//!
//! SOURCE ONLY — never compiled: the build container has no Rust toolchain.  Every `extern "C"`
//! declaration below is a line-for-line transcription of `include/halart.h`; the ctypes twin of this file
//! (`hala-renderer_amd/renderer.py`) is what the test-suite exercises against the same library.
#![allow(non_camel_case_types)]

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int, c_void};
use std::path::Path;

pub mod sys {
  use super::*;

  #[repr(C)] pub struct hala_rt_renderer { _private: [u8; 0] }

  /// src/scene/vertex.rs:2-9 (44 B)
  #[repr(C)] #[derive(Clone, Copy)]
  pub struct hala_vertex { pub position: [f32; 3], pub normal: [f32; 3], pub tangent: [f32; 3], pub tex_coord: [f32; 2] }

  #[repr(C)] pub struct hala_node_desc {
    pub name: *const c_char, pub parent: i32, pub local_transform: [f32; 16],
    pub mesh_index: u32, pub camera_index: u32, pub light_index: u32,
  }
  #[repr(C)] pub struct hala_primitive_desc {
    pub indices: *const u32, pub index_count: u32, pub vertices: *const hala_vertex, pub vertex_count: u32, pub material_index: u32,
  }
  #[repr(C)] pub struct hala_mesh_desc { pub primitives: *const hala_primitive_desc, pub primitive_count: u32 }
  #[repr(C)] pub struct hala_material_desc {
    pub type_: u32, pub base_color: [f32; 3], pub opacity: f32, pub emission: [f32; 3], pub anisotropic: f32, pub metallic: f32,
    pub roughness: f32, pub subsurface: f32, pub specular_tint: f32, pub sheen: f32, pub sheen_tint: f32, pub clearcoat: f32,
    pub clearcoat_roughness: f32, pub clearcoat_tint: [f32; 3], pub specular_transmission: f32, pub ior: f32,
    pub medium_type: u32, pub medium_color: [f32; 3], pub medium_density: f32, pub medium_anisotropy: f32,
    pub base_color_map_index: u32, pub emission_map_index: u32, pub normal_map_index: u32, pub metallic_roughness_map_index: u32,
  }
  #[repr(C)] pub struct hala_light_desc { pub color: [f32; 3], pub intensity: f32, pub light_type: u32, pub param0: f32, pub param1: f32 }
  #[repr(C)] pub struct hala_camera_desc {
    pub type_: u32, pub aspect: f32, pub yfov: f32, pub znear: f32, pub zfar: f32, pub focal_distance: f32, pub aperture: f32, pub xmag: f32, pub ymag: f32,
  }
  #[repr(C)] pub struct hala_image_desc { pub format: u32, pub width: u32, pub height: u32, pub data: *const c_void, pub num_of_bytes: usize }
  #[repr(C)] pub struct hala_index_pair { pub key: u32, pub value: u32 }
  #[repr(C)] pub struct hala_scene_desc {
    pub nodes: *const hala_node_desc, pub node_count: u32,
    pub meshes: *const hala_mesh_desc, pub mesh_count: u32,
    pub materials: *const hala_material_desc, pub material_count: u32,
    pub lights: *const hala_light_desc, pub light_count: u32,
    pub cameras: *const hala_camera_desc, pub camera_count: u32,
    pub texture2image_mapping: *const hala_index_pair, pub texture_count: u32,
    pub image2data_mapping: *const hala_index_pair, pub image_count: u32,
    pub image_data: *const hala_image_desc, pub image_data_count: u32,
  }
  #[repr(C)] #[derive(Clone, Copy)] pub struct hala_ray { pub origin: [f32; 3], pub tmin: f32, pub direction: [f32; 3], pub tmax: f32 }
  #[repr(C)] #[derive(Clone, Copy)] pub struct hala_hit { pub t: f32, pub u: f32, pub v: f32, pub prim: u32 }
  #[repr(C)] #[derive(Default)] pub struct hala_rt_info { pub width: u32, pub height: u32 }

  /// hala_rt_build_options (include/halart.h): every field 0 = the default
  #[repr(C)] #[derive(Default, Clone, Copy)]
  pub struct hala_rt_build_options { pub builder: u32, pub ploc_tail: u32, pub ploc_look_every: u32, pub collapse_look_every: u32, pub instancing: u32, pub reserved: [u32; 3] }
  #[repr(C)] pub struct hala_scene { _private: [u8; 0] }
  #[repr(C)] pub struct hala_rtprog { _private: [u8; 0] }
  extern "C" {
    pub fn hala_last_error_message() -> *const c_char;
    pub fn hala_rt_create(name: *const c_char, width: u32, height: u32, device_ordinal: c_int, max_depth: u32, rr_depth: u32,
                          enable_tonemap: c_int, enable_aces: c_int, use_simple_aces: c_int, max_frames: u64, out: *mut *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_destroy(r: *mut hala_rt_renderer);
    pub fn hala_rt_push_general_shader(r: *mut hala_rt_renderer, code: *const c_void, code_size: usize, stage: c_int, debug_name: *const c_char) -> c_int;
    pub fn hala_rt_push_general_shader_with_file(r: *mut hala_rt_renderer, file_path: *const c_char, stage: c_int, debug_name: *const c_char) -> c_int;
    pub fn hala_rt_push_hit_shaders_with_file(r: *mut hala_rt_renderer, closest: *const c_char, any: *const c_char, isect: *const c_char, debug_name: *const c_char) -> c_int;
    pub fn hala_rt_load_blue_noise_pixels(r: *mut hala_rt_renderer, rgba8: *const u8, width: u32, height: u32) -> c_int;
    pub fn hala_rt_set_scene(r: *mut hala_rt_renderer, scene: *const hala_scene_desc) -> c_int;
    pub fn hala_rt_set_envmap_file(r: *mut hala_rt_renderer, path: *const c_char, rotation_degrees: f32) -> c_int;
    pub fn hala_rt_set_envmap_pixels(r: *mut hala_rt_renderer, pixels: *const f32, channels: u32, width: u32, height: u32, rotation_degrees: f32) -> c_int;
    pub fn hala_rt_set_ground_color(r: *mut hala_rt_renderer, rgba: *const f32);
    pub fn hala_rt_set_sky_color(r: *mut hala_rt_renderer, rgba: *const f32);
    pub fn hala_rt_set_env_intensity(r: *mut hala_rt_renderer, intensity: f32);
    pub fn hala_rt_set_exposure_value(r: *mut hala_rt_renderer, exposure_value: f32);
    pub fn hala_rt_commit(r: *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_update(r: *mut hala_rt_renderer, delta_time: f64, width: u32, height: u32) -> c_int;
    pub fn hala_rt_update_batch(r: *mut hala_rt_renderer, frames: u32) -> c_int;
    pub fn hala_rt_render(r: *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_wait_idle(r: *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_save_images(r: *mut hala_rt_renderer, path: *const c_char) -> c_int;
    pub fn hala_rt_get_info(r: *mut hala_rt_renderer, out: *mut hala_rt_info) -> c_int;
    pub fn hala_rt_trace_rays(r: *mut hala_rt_renderer, d_rays: *const hala_ray, d_hits: *mut hala_hit, count: u32, mode: c_int,
                              d_counters: *mut u64, hip_stream: *mut c_void) -> c_int;
    pub fn hala_rt_trace_rays_indirect(r: *mut hala_rt_renderer, d_rays: *const hala_ray, d_hits: *mut hala_hit, d_indirect: *const u32,
                                       mode: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn hala_rt_update_node_transform(r: *mut hala_rt_renderer, node_index: u32, local_transform: *const f32) -> c_int;
    pub fn hala_rt_update_material(r: *mut hala_rt_renderer, material_index: u32, material: *const hala_material_desc) -> c_int;
    pub fn hala_rt_update_vertices(r: *mut hala_rt_renderer, mesh_index: u32, primitive_index: u32, vertices: *const hala_vertex, vertex_count: u32) -> c_int;
    pub fn hala_rt_refit(r: *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_load_blue_noise_texture(r: *mut hala_rt_renderer, path: *const c_char) -> c_int;
    pub fn hala_rt_set_tile_shard(r: *mut hala_rt_renderer, rank: u32, world: u32, tile_size: u32) -> c_int;
    // the exchange step (RCCL all-gather + de-interleave) inside the library
    pub fn hala_rt_comm_unique_id(out_128_bytes: *mut u8) -> c_int;
    pub fn hala_rt_comm_init_rank(r: *mut hala_rt_renderer, unique_id_128_bytes: *const u8, rank: u32, world: u32) -> c_int;
    pub fn hala_rt_comm_attach(r: *mut hala_rt_renderer, nccl_comm: *mut c_void) -> c_int;
    pub fn hala_rt_comm_destroy(r: *mut hala_rt_renderer) -> c_int;
    pub fn hala_rt_tile_allgather(r: *mut hala_rt_renderer, aov_mask: u32) -> c_int;
    pub fn hala_rt_tile_allgather_begin(r: *mut hala_rt_renderer, aov_mask: u32) -> c_int;
    pub fn hala_rt_tile_allgather_finish(r: *mut hala_rt_renderer) -> c_int;
    // HalaRayTracingProgram as a C object
    pub fn hala_rtprog_create(r: *mut hala_rt_renderer, desc_json: *const c_char, debug_name: *const c_char, out: *mut *mut hala_rtprog) -> c_int;
    pub fn hala_rtprog_destroy(p: *mut hala_rtprog);
    pub fn hala_rtprog_bind(p: *mut hala_rtprog, d_rays: *const hala_ray, d_hits: *mut hala_hit) -> c_int;
    pub fn hala_rtprog_push_constants(p: *mut hala_rtprog, offset: u32, data: *const c_void, len: usize) -> c_int;
    pub fn hala_rtprog_push_constants_f32(p: *mut hala_rtprog, offset: u32, data: *const f32, count: usize) -> c_int;
    pub fn hala_rtprog_trace_rays(p: *mut hala_rtprog, width: u32, height: u32, depth: u32, hip_stream: *mut c_void) -> c_int;
    pub fn hala_rtprog_trace_rays_indirect(p: *mut hala_rtprog, d_indirect: *const u32, hip_stream: *mut c_void) -> c_int;
    pub fn hala_rt_tile_buffer(r: *mut hala_rt_renderer, which: c_int, d_ptr: *mut *mut c_void, bytes: *mut usize) -> c_int;
    pub fn hala_rt_scatter_gathered_tiles(r: *mut hala_rt_renderer, which: c_int, d_gathered: *const c_void, bytes: usize) -> c_int;
    pub fn hala_rt_scatter_gathered_tiles_on_stream(r: *mut hala_rt_renderer, which: c_int, d_gathered: *const c_void, bytes: usize, hip_stream: *mut c_void) -> c_int;
    pub fn hala_rt_get_stream(r: *mut hala_rt_renderer, hip_stream: *mut *mut c_void) -> c_int;
    pub fn hala_rt_set_launch_timing_period(r: *mut hala_rt_renderer, period: u32) -> c_int;
    pub fn hala_rt_set_pass_fusion(r: *mut hala_rt_renderer, mode: u32) -> c_int;
    pub fn hala_rt_set_build_options(r: *mut hala_rt_renderer, options: *const hala_rt_build_options) -> c_int;
    pub fn hala_rt_tile_allgather_begin_external(r: *mut hala_rt_renderer, aov_mask: u32) -> c_int;
    pub fn hala_rt_get_exchange_buffers(r: *mut hala_rt_renderer, which: c_int, d_staged: *mut *mut c_void, staged_bytes: *mut usize,
                                        d_receive: *mut *mut c_void, receive_bytes: *mut usize, hip_stream: *mut *mut c_void) -> c_int;
    // cpu::HalaScene::new inside the library (for hosts without the Rust `src/scene` module)
    pub fn hala_scene_load_gltf(path: *const c_char, out: *mut *mut hala_scene) -> c_int;
    pub fn hala_scene_get_desc(scene: *const hala_scene) -> *const hala_scene_desc;
    pub fn hala_scene_free(scene: *mut hala_scene);
    pub fn hala_load_float_image(path: *const c_char, width: *mut u32, height: *mut u32, channels: *mut u32, dst: *mut f32, capacity_floats: usize) -> c_int;
  }
}

/// src/error.rs:5-22
#[derive(thiserror::Error, Debug)]
#[error("{msg}")]
pub struct HalaRendererError { msg: String }
impl HalaRendererError {
  pub fn new(msg: &str) -> Self { Self { msg: msg.to_string() } }
  pub fn message(&self) -> &str { &self.msg }
}
fn check(rc: c_int) -> Result<(), HalaRendererError> {
  if rc == 0 { return Ok(()); }
  let msg = unsafe { CStr::from_ptr(sys::hala_last_error_message()) }.to_string_lossy().into_owned();
  Err(HalaRendererError { msg })
}
fn cpath<P: AsRef<Path>>(p: P) -> CString { CString::new(p.as_ref().to_string_lossy().as_bytes()).unwrap() }

/// CPU scene model the application already owns (`hala_renderer::scene::cpu`); only the parts read here are listed.
pub mod cpu {
  pub struct HalaNode { pub name: String, pub parent: Option<u32>, pub local_transform: glam::Mat4, pub mesh_index: u32, pub camera_index: u32, pub light_index: u32 }
  pub struct HalaPrimitive { pub indices: Vec<u32>, pub vertices: Vec<super::sys::hala_vertex>, pub material_index: u32 }
  pub struct HalaMesh { pub primitives: Vec<HalaPrimitive> }
  pub struct HalaScene {
    pub nodes: Vec<HalaNode>, pub meshes: Vec<HalaMesh>, pub materials: Vec<super::sys::hala_material_desc>,
    pub lights: Vec<super::sys::hala_light_desc>, pub cameras: Vec<super::sys::hala_camera_desc>,
    pub texture2image_mapping: std::collections::BTreeMap<u32, u32>, pub image2data_mapping: std::collections::BTreeMap<u32, u32>,
    pub image_data: Vec<(u32, u32, u32, Vec<u8>)>,  // (format, width, height, bytes)
  }
}

/// src/renderer.rs:11-15
pub struct HalaRendererInfo { pub name: String, pub width: u32, pub height: u32 }

/// src/rt_renderer.rs:568-617 — same constructor arguments minus the window (headless).
pub struct HalaRenderer { h: *mut sys::hala_rt_renderer, info: HalaRendererInfo }

impl HalaRenderer {
  /// src/rt_renderer.rs:650-813
  #[allow(clippy::too_many_arguments)]
  pub fn new(name: &str, width: u32, height: u32, device_ordinal: i32, max_depth: u32, rr_depth: u32, enable_tonemap: bool,
             enable_aces: bool, use_simple_aces: bool, max_frames: u64) -> Result<Self, HalaRendererError> {
    let mut h = std::ptr::null_mut();
    let cname = CString::new(name).unwrap();
    check(unsafe { sys::hala_rt_create(cname.as_ptr(), width, height, device_ordinal, max_depth, rr_depth, enable_tonemap as c_int,
                                       enable_aces as c_int, use_simple_aces as c_int, max_frames, &mut h) })?;
    Ok(Self { h, info: HalaRendererInfo { name: name.to_string(), width, height } })
  }
  pub fn info(&self) -> &HalaRendererInfo { &self.info }

  /// src/rt_renderer.rs:965-995 (accepted, recorded, ignored: the integrator is compiled into the library)
  pub fn push_general_shader_with_file(&mut self, file_path: &str, stage: i32, debug_name: &str) -> Result<(), HalaRendererError> {
    let (p, n) = (CString::new(file_path).unwrap(), CString::new(debug_name).unwrap());
    check(unsafe { sys::hala_rt_push_general_shader_with_file(self.h, p.as_ptr(), stage, n.as_ptr()) })
  }
  /// src/rt_renderer.rs:1161-1178 — the scene is borrowed for the call only
  pub fn set_scene(&mut self, scene: &mut cpu::HalaScene) -> Result<(), HalaRendererError> {
    let names: Vec<CString> = scene.nodes.iter().map(|n| CString::new(n.name.as_str()).unwrap()).collect();
    let nodes: Vec<sys::hala_node_desc> = scene.nodes.iter().zip(&names).map(|(n, name)| sys::hala_node_desc {
      name: name.as_ptr(), parent: n.parent.map_or(-1, |p| p as i32), local_transform: n.local_transform.to_cols_array(),
      mesh_index: n.mesh_index, camera_index: n.camera_index, light_index: n.light_index }).collect();
    let prims: Vec<Vec<sys::hala_primitive_desc>> = scene.meshes.iter().map(|m| m.primitives.iter().map(|p| sys::hala_primitive_desc {
      indices: p.indices.as_ptr(), index_count: p.indices.len() as u32, vertices: p.vertices.as_ptr(), vertex_count: p.vertices.len() as u32,
      material_index: p.material_index }).collect()).collect();
    let meshes: Vec<sys::hala_mesh_desc> = prims.iter().map(|p| sys::hala_mesh_desc { primitives: p.as_ptr(), primitive_count: p.len() as u32 }).collect();
    let t2i: Vec<sys::hala_index_pair> = scene.texture2image_mapping.iter().map(|(k, v)| sys::hala_index_pair { key: *k, value: *v }).collect();
    let i2d: Vec<sys::hala_index_pair> = scene.image2data_mapping.iter().map(|(k, v)| sys::hala_index_pair { key: *k, value: *v }).collect();
    let images: Vec<sys::hala_image_desc> = scene.image_data.iter().map(|(f, w, h, d)| sys::hala_image_desc {
      format: *f, width: *w, height: *h, data: d.as_ptr() as *const c_void, num_of_bytes: d.len() }).collect();
    let desc = sys::hala_scene_desc {
      nodes: nodes.as_ptr(), node_count: nodes.len() as u32, meshes: meshes.as_ptr(), mesh_count: meshes.len() as u32,
      materials: scene.materials.as_ptr(), material_count: scene.materials.len() as u32, lights: scene.lights.as_ptr(), light_count: scene.lights.len() as u32,
      cameras: scene.cameras.as_ptr(), camera_count: scene.cameras.len() as u32, texture2image_mapping: t2i.as_ptr(), texture_count: t2i.len() as u32,
      image2data_mapping: i2d.as_ptr(), image_count: i2d.len() as u32, image_data: images.as_ptr(), image_data_count: images.len() as u32 };
    check(unsafe { sys::hala_rt_set_scene(self.h, &desc) })
  }
  /// src/rt_renderer.rs:1117-1156
  pub fn load_blue_noise_texture<P: AsRef<Path>>(&mut self, path: P) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rt_load_blue_noise_texture(self.h, cpath(path).as_ptr()) })
  }
  /// src/rt_renderer.rs:1184-1195
  pub fn set_envmap<P: AsRef<Path>>(&mut self, path: P, rotation: f32) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rt_set_envmap_file(self.h, cpath(path).as_ptr(), rotation) })
  }
  /// src/rt_renderer.rs:1199-1219
  pub fn set_ground_color(&mut self, color: glam::Vec4) { unsafe { sys::hala_rt_set_ground_color(self.h, color.to_array().as_ptr()) } }
  pub fn set_sky_color(&mut self, color: glam::Vec4) { unsafe { sys::hala_rt_set_sky_color(self.h, color.to_array().as_ptr()) } }
  pub fn set_env_intensity(&mut self, intensity: f32) { unsafe { sys::hala_rt_set_env_intensity(self.h, intensity) } }
  pub fn set_exposure_value(&mut self, exposure_value: f32) { unsafe { sys::hala_rt_set_exposure_value(self.h, exposure_value) } }
  /// src/rt_renderer.rs:1224-1352
  pub fn save_images<P: AsRef<Path>>(&self, path: P) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_save_images(self.h, cpath(path).as_ptr()) }) }

  // HalaRendererTrait (src/renderer.rs:210-324)
  pub fn commit(&mut self) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_commit(self.h) }) }
  pub fn update(&mut self, delta_time: f64, width: u32, height: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_update(self.h, delta_time, width, height) }) }
  pub fn update_batch(&mut self, frames: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_update_batch(self.h, frames) }) }
  pub fn render(&mut self) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_render(self.h) }) }
  pub fn wait_idle(&self) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_wait_idle(self.h) }) }

  // beyond the reference: refit + tile sharding (BASELINE.json north_star)
  pub fn update_node_transform(&mut self, node_index: u32, local: glam::Mat4) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rt_update_node_transform(self.h, node_index, local.to_cols_array().as_ptr()) })
  }
  pub fn update_vertices(&mut self, mesh_index: u32, primitive_index: u32, vertices: &[sys::hala_vertex]) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rt_update_vertices(self.h, mesh_index, primitive_index, vertices.as_ptr(), vertices.len() as u32) })
  }
  pub fn refit(&mut self) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_refit(self.h) }) }
  /// the hipStream_t of this renderer, for stream-ordered hand-overs of the tile buffer (multi-GPU gather)
  pub fn stream(&self) -> Result<*mut c_void, HalaRendererError> {
    let mut s: *mut c_void = std::ptr::null_mut();
    check(unsafe { sys::hala_rt_get_stream(self.h, &mut s) })?;
    Ok(s)
  }
  /// per-launch timing events on every `period`-th update (1: all, 0: none)
  pub fn set_launch_timing_period(&mut self, period: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_set_launch_timing_period(self.h, period) }) }
  /// 0: one launch per pass, 1 (default): fused launches except in timed updates, 2: always
  pub fn set_pass_fusion(&mut self, mode: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_set_pass_fusion(self.h, mode) }) }
  /// How the next commit() builds the acceleration structure: builder 0 auto | 1 SAH | 2 PLOC | 3 LBVH; instancing 0 auto | 1 flattened |
  /// 2 two-level (the BLAS / TLAS split of gpu_uploader.rs:782-815, :937-959)
  pub fn set_build_options(&mut self, builder: u32, instancing: u32) -> Result<(), HalaRendererError> {
    let o = sys::hala_rt_build_options { builder, instancing, ..Default::default() };
    check(unsafe { sys::hala_rt_set_build_options(self.h, &o) })
  }
  pub fn set_tile_shard(&mut self, rank: u32, world: u32, tile_size: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_set_tile_shard(self.h, rank, world, tile_size) }) }
}
impl Drop for HalaRenderer { fn drop(&mut self) { unsafe { sys::hala_rt_destroy(self.h) } } }

/// A scene loaded by the library's own glTF reader (csrc/gltf_loader.cpp): what `cpu::HalaScene::new(path)` returns in the
/// reference (src/scene/cpu/scene.rs:40-55), kept on the C side and handed to `set_scene` by reference.
pub struct HalaNativeScene { h: *mut sys::hala_scene }
impl HalaNativeScene {
  pub fn new<P: AsRef<Path>>(path: P) -> Result<Self, HalaRendererError> {
    let mut h = std::ptr::null_mut();
    check(unsafe { sys::hala_scene_load_gltf(cpath(path).as_ptr(), &mut h) })?;
    Ok(Self { h })
  }
}
impl Drop for HalaNativeScene { fn drop(&mut self) { unsafe { sys::hala_scene_free(self.h) } } }
impl HalaRenderer {
  pub fn set_native_scene(&mut self, scene: &HalaNativeScene) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rt_set_scene(self.h, sys::hala_scene_get_desc(scene.h)) })
  }
}

/// src/raytracing_program.rs:70-341 — the generic "RT pass" object, over the library's `hala_rtprog_*` exports: a ray batch traced
/// against a committed renderer's acceleration structure.  `desc` is serialised with serde exactly as the reference's
/// HalaRayTracingProgramDesc (:33-47), so an application's existing JSON descriptions load unchanged.
pub struct HalaRayTracingProgram<'a> { h: *mut sys::hala_rtprog, _renderer: std::marker::PhantomData<&'a HalaRenderer> }
impl<'a> HalaRayTracingProgram<'a> {
  /// :85-252 — the renderer stands for logical_device + descriptor_set_layouts
  pub fn new(renderer: &'a HalaRenderer, desc_json: &str, debug_name: &str) -> Result<Self, HalaRendererError> {
    let (d, n) = (CString::new(desc_json).unwrap(), CString::new(debug_name).unwrap());
    let mut h = std::ptr::null_mut();
    check(unsafe { sys::hala_rtprog_create(renderer.h, d.as_ptr(), n.as_ptr(), &mut h) })?;
    Ok(Self { h, _renderer: std::marker::PhantomData })
  }
  /// :264-278 — device addresses of the batch stand in for descriptor sets
  pub fn bind(&mut self, d_rays: *const sys::hala_ray, d_hits: *mut sys::hala_hit) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rtprog_bind(self.h, d_rays, d_hits) })
  }
  /// :285-300 — bytes 0..3 select the hit group: 0 closest hit, 1 any hit
  pub fn push_constants(&mut self, offset: u32, data: &[u8]) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rtprog_push_constants(self.h, offset, data.as_ptr() as *const c_void, data.len()) })
  }
  /// :307-322
  pub fn push_constants_f32(&mut self, offset: u32, data: &[f32]) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rtprog_push_constants_f32(self.h, offset, data.as_ptr(), data.len()) })
  }
  /// :330-332
  pub fn trace_rays(&self, width: u32, height: u32, depth: u32) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rtprog_trace_rays(self.h, width, height, depth, std::ptr::null_mut()) })
  }
  /// :338-340
  pub fn trace_rays_indirect(&self, indirect_device_address: u64) -> Result<(), HalaRendererError> {
    check(unsafe { sys::hala_rtprog_trace_rays_indirect(self.h, indirect_device_address as *const u32, std::ptr::null_mut()) })
  }
}
impl<'a> Drop for HalaRayTracingProgram<'a> { fn drop(&mut self) { unsafe { sys::hala_rtprog_destroy(self.h) } } }

/// multi-GPU: one process (or thread) per GPU; rank 0 makes the id, the application hands it to the other ranks
impl HalaRenderer {
  pub fn comm_unique_id() -> Result<[u8; 128], HalaRendererError> {
    let mut id = [0u8; 128];
    check(unsafe { sys::hala_rt_comm_unique_id(id.as_mut_ptr()) })?;
    Ok(id)
  }
  pub fn comm_init_rank(&mut self, id: &[u8; 128], rank: u32, world: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_comm_init_rank(self.h, id.as_ptr(), rank, world) }) }
  /// aov_mask bit k = AOV k (0 accum, 1 albedo, 2 normal, 3 final): RCCL all-gather of the rank's tiles + de-interleave, stream-ordered
  pub fn tile_allgather(&mut self, aov_mask: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_tile_allgather(self.h, aov_mask) }) }
  pub fn tile_allgather_begin(&mut self, aov_mask: u32) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_tile_allgather_begin(self.h, aov_mask) }) }
  pub fn tile_allgather_finish(&mut self) -> Result<(), HalaRendererError> { check(unsafe { sys::hala_rt_tile_allgather_finish(self.h) }) }
}
