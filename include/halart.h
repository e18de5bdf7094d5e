/*
 * halart.h — C ABI of libhalart.so, the MI355X-native replacement for the hot path of
 * hala-renderer's ray-tracing renderer (reference: src/rt_renderer.rs, src/raytracing_program.rs,
 * src/envmap.rs, src/scene/loader/gpu_uploader.rs).
 *
 * The reference has no FFI boundary of its own: its "operator API" is the public Rust surface
 * `HalaRenderer` / `HalaRayTracingProgram` / `cpu::HalaScene`.  Every export below is what an
 * `extern "C"` Rust shim with those names would bind to; the reference item each entry point
 * replaces is cited as file:line (relative to the reference checkout).
 *
 * Conventions (reference: src/error.rs:5-22 — every fallible method returns Result<_, HalaRendererError>):
 *   - every fallible function returns an int status: 0 = Ok, non-zero = Err; the message of the last
 *     error on the calling thread is returned by hala_last_error_message() (== HalaRendererError::message()).
 *   - nothing aborts or throws across the boundary.
 *   - one handle <-> one host thread <-> one GPU (the reference is !Send/!Sync: src/renderer.rs:44).
 *   - all pointers are plain host pointers unless the name says `_device`/`d_` (then a HIP device pointer).
 *   - the library REQUIRES a HIP device: there is no CPU execution path in it.
 */
#ifndef HALART_H
#define HALART_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALA_OK 0
#define HALA_ERR 1

#define HALA_INVALID_INDEX 0xffffffffu /* u32::MAX "none" marker (src/scene/cpu/node.rs:23-25) */
#define HALA_MAX_CAMERA_COUNT 8        /* src/scene/loader/gpu_uploader.rs:39 */
#define HALA_MAX_LIGHT_COUNT 32        /* src/scene/loader/gpu_uploader.rs:40 */

/* ------------------------------------------------------------------------------------------------
 * Byte-exact device records (what the reference's shaders read).  Sizes/offsets are static_asserted
 * in hala-renderer_amd/csrc/hala_types.h and checked from Python in tests/test_layouts.py.
 * ---------------------------------------------------------------------------------------------- */

/* src/scene/vertex.rs:2-9 — 44 B */
typedef struct hala_vertex {
  float position[3];
  float normal[3];
  float tangent[3];
  float tex_coord[2];
} hala_vertex;

/* src/scene/gpu/camera.rs:10-20 — 80 B, align 16 */
typedef struct hala_gpu_camera {
  float position[3]; float _pad0;
  float right[3];    float _pad1;
  float up[3];       float _pad2;
  float forward[3];
  float yfov;
  float focal_distance_or_xmag;
  float aperture_or_ymag;
  uint32_t type; /* 0 perspective, 1 orthographic */
  uint32_t _pad3;
} hala_gpu_camera;

/* src/scene/gpu/light.rs:7-32 — 80 B, align 16 */
typedef struct hala_gpu_light {
  float intensity[3]; float _pad0;
  float position[3];  float _pad1;
  float u[3];         float _pad2;
  float v[3];
  float radius;
  float area;
  uint32_t type; /* 0 point, 1 directional, 2 spot, 3 quad, 4 sphere (src/scene/cpu/light.rs:7-12) */
  uint32_t _pad3[2];
} hala_gpu_light;

/* light AABB, `HalaAABB` of the absent hala-gfx crate: fields min/max per gpu_uploader.rs:169-180 — 24 B */
typedef struct hala_aabb {
  float min[3];
  float max[3];
} hala_aabb;

/* src/scene/gpu/material.rs:6-48 — 144 B, align 16 */
typedef struct hala_gpu_material {
  /* HalaMedium, 32 B */
  float medium_color[3];
  float medium_density;
  float medium_anisotropy;
  uint32_t medium_type;
  float _medium_padding[2];

  float base_color[3];
  float opacity;
  float emission[3];
  float anisotropic;
  float metallic;
  float roughness;
  float subsurface;
  float specular_tint;
  float sheen;
  float sheen_tint;
  float clearcoat;
  float clearcoat_roughness;
  float clearcoat_tint[3];
  float specular_transmission;
  float ior;
  float ax;
  float ay;
  uint32_t base_color_map_index;
  uint32_t normal_map_index;
  uint32_t metallic_roughness_map_index;
  uint32_t emission_map_index;
  uint32_t type; /* 0 diffuse, 1 disney (src/scene/cpu/material.rs:7-9) */
} hala_gpu_material;

/* src/scene/gpu/mesh.rs:32-39 — 96 B (glam::Mat4 is 16-aligned) */
typedef struct hala_gpu_mesh_data {
  float transform[16]; /* column-major object->world */
  uint32_t material_index;
  uint32_t _pad0;
  uint64_t vertices; /* device address of this primitive's hala_vertex[] */
  uint64_t indices;  /* device address of this primitive's uint32_t[]    */
  uint64_t _pad1;
} hala_gpu_mesh_data;

/* src/rt_renderer.rs:44-65 — 112 B, filled per frame at :408-427 */
typedef struct hala_global_uniform {
  float ground_color[4];
  float sky_color[4];
  float resolution[2];
  uint32_t max_depth;
  uint32_t rr_depth;
  uint32_t frame_index;
  uint32_t camera_index;
  uint32_t env_type; /* 0 SKY, 1 MAP (src/rt_renderer.rs:24-28) */
  uint32_t env_map_width;
  uint32_t env_map_height;
  float env_total_sum;
  float env_rotation; /* degrees / 360 (src/rt_renderer.rs:420) */
  float env_intensity;
  float exposure_value;
  uint32_t enable_tonemap;
  uint32_t enable_aces;
  uint32_t use_simple_aces;
  uint32_t num_of_lights;
  uint32_t _pad[3];
} hala_global_uniform;

/* ------------------------------------------------------------------------------------------------
 * Borrowed view of cpu::HalaScene (src/scene/cpu/scene.rs:17-26).  The renderer copies what it
 * needs during hala_rt_set_scene; the caller keeps ownership (as with `&mut cpu::HalaScene`).
 * ---------------------------------------------------------------------------------------------- */

/* src/scene/cpu/node.rs:2-12.  world transforms are recomputed by the library exactly as
 * update_node_hierarchies does (src/scene/cpu/scene.rs:99-114): parents must precede children. */
typedef struct hala_node_desc {
  const char* name;
  int32_t parent;            /* -1 == None */
  float local_transform[16]; /* column-major glam::Mat4 */
  uint32_t mesh_index;       /* HALA_INVALID_INDEX == none */
  uint32_t camera_index;
  uint32_t light_index;
} hala_node_desc;

/* src/scene/cpu/mesh.rs:6-13 (meshlet fields are rasterizer-only and omitted) */
typedef struct hala_primitive_desc {
  const uint32_t* indices;
  uint32_t index_count;
  const hala_vertex* vertices;
  uint32_t vertex_count;
  uint32_t material_index;
} hala_primitive_desc;

typedef struct hala_mesh_desc {
  const hala_primitive_desc* primitives;
  uint32_t primitive_count;
} hala_mesh_desc;

/* src/scene/cpu/material.rs:24-50, :75-80 */
typedef struct hala_material_desc {
  uint32_t type; /* 0 DIFFUSE, 1 DISNEY; anything else is an error (from_u8 panics in the reference) */
  float base_color[3];
  float opacity;
  float emission[3];
  float anisotropic;
  float metallic;
  float roughness;
  float subsurface;
  float specular_tint;
  float sheen;
  float sheen_tint;
  float clearcoat;
  float clearcoat_roughness;
  float clearcoat_tint[3];
  float specular_transmission;
  float ior;
  uint32_t medium_type; /* 0 NONE, 1 ABSORB, 2 SCATTER, 3 EMISSIVE */
  float medium_color[3];
  float medium_density;
  float medium_anisotropy;
  uint32_t base_color_map_index;
  uint32_t emission_map_index;
  uint32_t normal_map_index;
  uint32_t metallic_roughness_map_index;
} hala_material_desc;

/* src/scene/cpu/light.rs:30-39 */
typedef struct hala_light_desc {
  float color[3];
  float intensity;
  uint32_t light_type; /* 0..4 */
  float param0;
  float param1;
} hala_light_desc;

/* src/scene/cpu/camera.rs:4-29 */
typedef struct hala_camera_desc {
  uint32_t type; /* 0 perspective, 1 orthographic */
  float aspect;
  float yfov;
  float znear;
  float zfar;
  float focal_distance;
  float aperture;
  float xmag;
  float ymag;
} hala_camera_desc;

/* src/scene/cpu/image_data.rs:14-20; format codes are this library's own small enum */
#define HALA_FORMAT_R8G8B8A8_UNORM 0
#define HALA_FORMAT_R8G8B8A8_SRGB 1        /* glTF 8-bit RGB(A) images (src/scene/loader/gltf_loader.rs:395-396) */
#define HALA_FORMAT_R32G32B32A32_SFLOAT 2
#define HALA_FORMAT_B8G8R8A8_UNORM 3       /* files loaded through cpu::HalaImageData: RGBA bytes tagged BGRA without
                                              a swizzle (src/scene/cpu/image_data.rs:39-43) — red and blue swap */
typedef struct hala_image_desc {
  uint32_t format;
  uint32_t width;
  uint32_t height;
  const void* data;
  size_t num_of_bytes;
} hala_image_desc;

typedef struct hala_index_pair {
  uint32_t key;
  uint32_t value;
} hala_index_pair;

typedef struct hala_scene_desc {
  const hala_node_desc* nodes;         uint32_t node_count;
  const hala_mesh_desc* meshes;        uint32_t mesh_count;
  const hala_material_desc* materials; uint32_t material_count;
  const hala_light_desc* lights;       uint32_t light_count;
  const hala_camera_desc* cameras;     uint32_t camera_count;
  const hala_index_pair* texture2image_mapping; uint32_t texture_count; /* BTreeMap<u32,u32>, ascending key */
  const hala_index_pair* image2data_mapping;    uint32_t image_count;
  const hala_image_desc* image_data;   uint32_t image_data_count;
} hala_scene_desc;

/* cpu::HalaScene::new (src/scene/cpu/scene.rs:40-55) + HalaGltfLoader::load (src/scene/loader/gltf_loader.rs:121-227): loads
 * a `.gltf` file (external or base64 buffers; PNG images) into an owned scene whose borrowed description is what
 * hala_rt_set_scene takes.  Error messages are the reference's ("Unsupported file ...", "No scene in glTF file ...",
 * "Read indices from mesh ... failed.", "Invalid material type.", "Unsupported image format.", ...). */
typedef struct hala_scene hala_scene;
int hala_scene_load_gltf(const char* path, hala_scene** out);
const hala_scene_desc* hala_scene_get_desc(const hala_scene* scene);
void hala_scene_free(hala_scene* scene);

/* ------------------------------------------------------------------------------------------------
 * Errors
 * ---------------------------------------------------------------------------------------------- */
/* HalaRendererError::message() (src/error.rs:23) of the last failed call on this thread. */
const char* hala_last_error_message(void);

/* ------------------------------------------------------------------------------------------------
 * HalaRenderer (src/rt_renderer.rs:568-1353, trait src/renderer.rs:210-324)
 * ---------------------------------------------------------------------------------------------- */
typedef struct hala_rt_renderer hala_rt_renderer;

/* HalaRenderer::new (src/rt_renderer.rs:650-813).  Headless: width/height replace gpu_req.{width,height}
 * (:661-662), device_ordinal replaces the winit window/swapchain.  max_frames == 0 => u64::MAX (:774). */
int hala_rt_create(const char* name, uint32_t width, uint32_t height, int device_ordinal,
                   uint32_t max_depth, uint32_t rr_depth, int enable_tonemap, int enable_aces,
                   int use_simple_aces, uint64_t max_frames, hala_rt_renderer** out);
/* Drop order of src/rt_renderer.rs:620-633: images, then everything else. */
void hala_rt_destroy(hala_rt_renderer* r);

/* push_general_shader{,_with_file} (src/rt_renderer.rs:925-995) and push_hit_shaders{,_with_file}
 * (:1003-1112).  SPIR-V has no meaning for HIP kernels: the call is validated, recorded (so that
 * commit can enforce "at least one raygen shader was pushed" like the reference pipeline would) and
 * otherwise ignored.  stage: 0 raygen, 1 miss, 2 callable. */
int hala_rt_push_general_shader(hala_rt_renderer* r, const void* code, size_t code_size, int stage,
                                const char* debug_name);
int hala_rt_push_general_shader_with_file(hala_rt_renderer* r, const char* file_path, int stage,
                                          const char* debug_name);
int hala_rt_push_hit_shaders(hala_rt_renderer* r, const void* closest_hit, size_t closest_hit_size,
                             const void* any_hit, size_t any_hit_size, const void* intersection,
                             size_t intersection_size, const char* debug_name);
int hala_rt_push_hit_shaders_with_file(hala_rt_renderer* r, const char* closest_hit_path,
                                       const char* any_hit_path, const char* intersection_path,
                                       const char* debug_name);

/* load_blue_noise_texture (src/rt_renderer.rs:1117-1156).  Optional here (mandatory at commit in the
 * reference, :319).  The image is decoded, validated and uploaded like the reference does, and then IGNORED:
 * the built-in integrator draws every sample from a counter-based hash RNG keyed by (pixel id, frame index)
 * (docs/RENDER_SPEC.md 2.3), which is what makes a pixel's value independent of tiling, batching and ranks. */
int hala_rt_load_blue_noise_texture(hala_rt_renderer* r, const char* path); /* PNG / JPEG, like HalaImageData::new_with_file */
int hala_rt_load_blue_noise_pixels(hala_rt_renderer* r, const uint8_t* rgba8, uint32_t width,
                                   uint32_t height);

/* set_scene (src/rt_renderer.rs:1161-1178) -> HalaSceneGPUUploader::upload(.., false, false, true)
 * (src/scene/loader/gpu_uploader.rs:63-545, :774-967).  Refused (the reference would hand them to the driver unchecked): primitives
 * without a material, indices out of range, vertex positions that are not finite. */
int hala_rt_set_scene(hala_rt_renderer* r, const hala_scene_desc* scene);

/* set_envmap (src/rt_renderer.rs:1184-1195) -> EnvMap::new_with_file (src/envmap.rs:38-232).
 * _pixels takes the already decoded image (RGB or RGBA f32, row 0 = top) and applies the same
 * validation (NaN/Inf rejection :63-71), alpha := 1 repack (:72-89) and table build (:239-388);
 * _file decodes Radiance .hdr (RGBE), .pfm or OpenEXR (scanline or single-level tiled; NONE / RLE / ZIPS / ZIP / PIZ; half, float) itself and honours
 * ./out/<stem>.dist_cache (:90-142). */
int hala_rt_set_envmap_pixels(hala_rt_renderer* r, const float* pixels, uint32_t channels,
                              uint32_t width, uint32_t height, float rotation_degrees);
int hala_rt_set_envmap_file(hala_rt_renderer* r, const char* path, float rotation_degrees);

/* src/rt_renderer.rs:1199-1219 */
void hala_rt_set_ground_color(hala_rt_renderer* r, const float rgba[4]);
void hala_rt_set_sky_color(hala_rt_renderer* r, const float rgba[4]);
void hala_rt_set_env_intensity(hala_rt_renderer* r, float intensity);
void hala_rt_set_exposure_value(hala_rt_renderer* r, float exposure_value);

/* commit (src/rt_renderer.rs:136-379): fails with "The scene in GPU is none!" without a scene (:138).
 * Here it also flattens the instances to world space, builds the BVH on the GPU (the work the
 * reference hands to vkCmdBuildAccelerationStructuresKHR: gpu_uploader.rs:784-811, :937-959) and
 * allocates the wavefront queues. */
int hala_rt_commit(hala_rt_renderer* r);
/* How commit() builds the acceleration structure — the HalaAccelerationStructure build flags of gpu_uploader.rs:784-811 /
 * :937-959 (PREFER_FAST_TRACE there).  Every field: 0 = the default.  The *_look_every / ploc_tail fields only change how the host
 * drives the rounds of a build (test hooks: the tree is the same for every value).  Takes effect at the next commit. */
typedef struct hala_rt_build_options {
  uint32_t builder;             /* 0 auto: full-sweep SAH from 4096 triangles, LBVH below | 1 SAH (fast trace) | 2 PLOC (fast build) | 3 LBVH */
  uint32_t ploc_tail;           /* 0 / 1: the last PLOC rounds in one workgroup | 2: every round its own launch */
  uint32_t ploc_look_every;     /* PLOC rounds between two host looks at the device counters (default 6) */
  uint32_t collapse_look_every; /* levels of the 4-wide collapse between two host looks (default 8) */
  uint32_t instancing;          /* what becomes of a primitive that several instances reference (RENDER_SPEC 4.5; the reference's BLAS /
                                 * TLAS split, gpu_uploader.rs:782-815, :843-885, :937-959):
                                 *  2: two-level tree — the primitive gets ONE object-space tree, its instances are leaves of instance
                                 *     levels that are rebuilt on the host when a node moves; every instanced primitive is stored once;
                                 *  1: every instance is flattened to world space, one tree over all triangles: the faster tree on this
                                 *     hardware (configs[3]: 9.7 instead of 11.8 ms per frame), at the full triangle count in memory;
                                 *  0: automatic — 1 unless the flattened scene holds more than 2^26 triangles (about 15 GB of tree), then 2.
                                 * The two forms intersect instanced geometry in different spaces: images agree to rounding, not bit for bit;
                                 * hala_bvh_info::instance_ref_count tells which one a commit chose. */
  uint32_t reserved[3];         /* must be 0 */
} hala_rt_build_options;
int hala_rt_set_build_options(hala_rt_renderer* r, const hala_rt_build_options* options);

/* update (src/rt_renderer.rs:387-471): pre_update bookkeeping (src/renderer.rs:266-281), the
 * `total_frames > max_frames` early-out (:394-396), the HalaGlobalUniform fill (:408-427) and one
 * trace_rays(width, height, 1) (:458-464) == one sample per pixel.  ui_fn is dropped. */
int hala_rt_update(hala_rt_renderer* r, double delta_time, uint32_t width, uint32_t height);
/* `frames` consecutive update() calls executed as ONE wavefront pass with `frames` paths per pixel in flight (in
 * chunks of at most 16).  Result, frame bookkeeping and max_frames behaviour are bit-identical to calling
 * hala_rt_update `frames` times; what changes is the launch count and the size of each launch (288 GB of HBM hold
 * the extra path state; the per-launch tail of the longest ray is amortised over `frames` times more rays). */
int hala_rt_update_batch(hala_rt_renderer* r, uint32_t frames);
/* render (src/rt_renderer.rs:475-502): no swapchain to present to.  Like submit_and_present_frame, which blocks only on the fence of
 * the swapchain image it is about to reuse, it bounds the updates in flight to two: it returns once the update BEFORE the latest has
 * finished.  Readers (read_image, save_images, get_statistics, wait_idle) wait for everything themselves. */
int hala_rt_render(hala_rt_renderer* r);
/* wait_idle (src/renderer.rs:251-256) */
int hala_rt_wait_idle(hala_rt_renderer* r);

/* save_images (src/rt_renderer.rs:1224-1352): <stem>_color.pfm (accum, tonemapped on the host exactly
 * as :1256-1316), <stem>_albedo.pfm, <stem>_normal.pfm; PFM layout of :1318-1334. */
int hala_rt_save_images(hala_rt_renderer* r, const char* path);

/* Test/bench access to what save_images reads back (:1239-1254): which = 0 accum, 1 albedo, 2 normal
 * (RGBA32F, 4*W*H floats, row 0 = top), 3 final (tonemapped RGBA32F the raygen stage writes, :688). */
int hala_rt_read_image(hala_rt_renderer* r, int which, float* dst_rgba32f);

/* info()/statistics() (src/renderer.rs:212-218, :135-207) */
typedef struct hala_rt_info {
  uint32_t width;
  uint32_t height;
} hala_rt_info;
typedef struct hala_rt_statistics {
  uint64_t total_frames;       /* HalaRendererStatistics::total_frames */
  double last_gpu_ms;          /* GPU time of the last update (get_gpu_frame_time, renderer.rs:275) */
  uint64_t rays_last_update;   /* closest-hit + shadow traversals launched by the last update */
  uint64_t rays_total;
  double traverse_ms_last_update; /* time inside the traversal kernels only (HIP events on the renderer's stream) */
  /* totals since create / the last accumulation reset */
  double gpu_ms_total;
  double traverse_closest_ms_total;   /* k_trace_batch<closest> launches, HIP events on the renderer's stream */
  double traverse_shadow_ms_total;    /* k_trace_shadow launches (one event pair per depth brackets the light and the
                                       * environment launch; traverse_shadow_launches counts the launches as issued) */
  uint64_t traverse_closest_launches;
  uint64_t traverse_shadow_launches;
  uint64_t updates_rendered;
  uint64_t rays_closest_total;
  uint64_t rays_shadow_total;
  /* only counted while hala_rt_set_counting(r, 1): BVH nodes visited / triangles tested per kernel, and the
   * rays those counts belong to */
  uint64_t nodes_closest_total, tris_closest_total, nodes_shadow_total, tris_shadow_total;
  uint64_t rays_closest_counted, rays_shadow_counted;
  /* counting launches, wave level (SIMT utilisation): wave steps = node visits issued by a 64-lane wave (lane
   * utilisation of the node path = nodes / (64 * wave_steps)); leaf passes = executions of a leaf-test copy by a
   * wave, leaf lanes = lanes taking part in them (utilisation of the leaf path = leaf_lanes / (64 * leaf_passes)) */
  uint64_t wave_steps_closest_total, leaf_passes_closest_total, leaf_lanes_closest_total;
  uint64_t wave_steps_shadow_total, leaf_passes_shadow_total, leaf_lanes_shadow_total;
  /* the depth-0 share of traverse_closest_*: launches of the kernel that generates the camera rays it traces
   * (k_trace_primary); the rest are k_trace_batch launches over the bounce-ray queues */
  double traverse_primary_ms_total;
  uint64_t traverse_primary_launches;
  uint64_t nodes_primary_total, tris_primary_total, rays_primary_counted; /* counting launches, camera rays only */
  uint64_t rays_primary_total; /* camera rays of all updates (part of rays_closest_total) */
  /* rays of the updates whose launches carried timing events (hala_rt_set_launch_timing_period): the rays the
   * traverse_*_ms_total / *_launches figures belong to.  Equal to the *_total fields while every update is timed. */
  uint64_t rays_closest_timed, rays_primary_timed, rays_shadow_timed;
  /* the shade launches between the closest-hit and the shadow launches of the timed updates (k_shade: closest-hit shading,
   * miss, NEE set-up, BSDF sampling, queue compaction) */
  double shade_ms_total;
  uint64_t shade_launches;
  /* timed updates that ran with fused passes (hala_rt_set_pass_fusion(r, 2)): the launches of k_trace_shadow_then_batch — the shadow
   * passes of bounce d and the closest-hit pass of bounce d + 1 in one persistent launch — and the rays they traced from each queue
   * (bounce rays / connections).  Such updates add nothing to traverse_shadow_* and only their depth-0 launch to traverse_closest_*. */
  double traverse_fused_ms_total;
  uint64_t traverse_fused_launches;
  uint64_t rays_fused_closest_timed, rays_fused_shadow_timed;
} hala_rt_statistics;
int hala_rt_get_info(hala_rt_renderer* r, hala_rt_info* out);
int hala_rt_get_statistics(hala_rt_renderer* r, hala_rt_statistics* out);
/* HalaRendererStatistics::reset (src/renderer.rs:168-174), which the reference calls on the device-lost path
 * (src/rt_renderer.rs:557): total_frames := 0, so the next update() renders frame_index 0 and the running
 * means restart. */
int hala_rt_reset_accumulation(hala_rt_renderer* r);
/* enable = 1: update() launches the counting variants of the traversal kernels (BVH nodes visited and
 * triangles tested per ray — the inputs of the algorithmic-bytes figure). Slower; off by default. */
int hala_rt_set_counting(hala_rt_renderer* r, int enable);
/* Per-launch timing (the traverse_*_ms_total statistics) brackets every traversal launch with two HIP events, i.e. ~22 barrier
 * packets per update (70 us of a 2.25 ms frame on MI355X).  period = 0 (default): none; 1: every update is timed; n > 1: every n-th.
 * The frame-level figures (last_gpu_ms, gpu_ms_total, ray counts) are always collected. */
int hala_rt_set_launch_timing_period(hala_rt_renderer* r, uint32_t period);
/* Pass fusion: the shadow passes of bounce d and the closest-hit pass of bounce d + 1 are independent and can run as ONE persistent
 * launch (one tail of long rays per bounce instead of three).  mode 0: never; 1 (default): every update except the timed ones, which
 * keep one launch per pass so that every measured launch is one kernel symbol with the chip to itself; 2: always — timed updates then
 * fill the traverse_fused_* statistics.  Counting updates (hala_rt_set_counting) never fuse.  Images do not depend on the mode. */
int hala_rt_set_pass_fusion(hala_rt_renderer* r, uint32_t mode);
/* the 112-B record the last update uploaded (src/rt_renderer.rs:408-427) */
int hala_rt_get_global_uniform(hala_rt_renderer* r, hala_global_uniform* out);

/* What upload() packed (gpu_uploader.rs:99-122 cameras, :148-303 lights + AABBs, :306-331 materials,
 * :843-885 primitives/instances) — read back from the device for parity tests. Each call copies
 * min(capacity, count) records and returns the count through *count. */
int hala_rt_get_packed_cameras(hala_rt_renderer* r, hala_gpu_camera* dst, uint32_t capacity, uint32_t* count);
int hala_rt_get_packed_lights(hala_rt_renderer* r, hala_gpu_light* dst, hala_aabb* dst_aabbs, uint32_t capacity, uint32_t* count);
int hala_rt_get_packed_materials(hala_rt_renderer* r, hala_gpu_material* dst, uint32_t capacity, uint32_t* count);
int hala_rt_get_packed_primitives(hala_rt_renderer* r, hala_gpu_mesh_data* dst, float* dst_instance_3x4, uint32_t capacity, uint32_t* count);
/* env tables of set_envmap (src/envmap.rs:239-388): total_sum, marginal[H], conditional[W*H] */
int hala_rt_get_env_distribution(hala_rt_renderer* r, float* total_sum, float* marginal, float* conditional);

/* Textures of set 2 binding 0 (src/rt_renderer.rs:197-226) as uploaded by gpu_uploader.rs:334-403: every texture is a
 * full mip chain (gen_mipmaps, :400); the sampler is linear / linear-mip / REPEAT (:341-353).  8-bit images stay 8-bit at
 * every level (RGBA8 in HBM, decoded at fetch: docs/RENDER_SPEC.md 7.4), float images are RGBA32F.  Introspection + a stand-alone
 * fetch for parity tests: read_texture_level returns the level's texel VALUES (decoded to linear RGBA32F), uv_lod holds (u, v, lod)
 * triples. */
int hala_rt_get_texture_info(hala_rt_renderer* r, uint32_t texture, uint32_t* width, uint32_t* height, uint32_t* mips);
int hala_rt_read_texture_level(hala_rt_renderer* r, uint32_t texture, uint32_t level, float* dst_rgba32f);
int hala_rt_sample_texture_host(hala_rt_renderer* r, uint32_t texture, const float* uv_lod, uint32_t count, float* dst_rgba32f);

/* Multi-GPU pixel-tile sharding (no reference equivalent; BASELINE.json north_star).  The frame is cut
 * into tile_size x tile_size tiles; tile t belongs to rank perm(t) % world (perm = fixed bijective
 * scramble).  After this call update() renders only this rank's tiles into a tile-major buffer;
 * hala_rt_tile_buffer gives its device address + byte size (per AOV) for the RCCL all-gather, and
 * hala_rt_scatter_gathered_tiles de-interleaves the gathered [world][tiles_per_rank][ts*ts][4] buffer (inside a tile the pixels
 * come in 8 x 8 blocks when ts is a multiple of 8: docs/RENDER_SPEC.md 9) into the row-major images of this renderer.  The renderer works on its own HIP stream: wait (hala_rt_wait_idle, or a stream
 * dependency on hala_rt_get_stream) before another stream reads the tile buffer — hala_rt_render does not flush. */
int hala_rt_set_tile_shard(hala_rt_renderer* r, uint32_t rank, uint32_t world, uint32_t tile_size);
int hala_rt_tile_buffer(hala_rt_renderer* r, int which, void** d_ptr, size_t* bytes);
/* the hipStream_t every launch of this renderer goes to (for stream-ordered hand-overs: hipStreamWaitEvent both ways) */
int hala_rt_get_stream(hala_rt_renderer* r, void** hip_stream);
int hala_rt_scatter_gathered_tiles(hala_rt_renderer* r, int which, const void* d_gathered, size_t bytes);

/* The exchange step itself (ncclAllGather over xGMI), inside the library so that a Rust / C host has a multi-GPU
 * path without any Python.  librccl is resolved on the first hala_rt_comm_* call (dlopen; the instance already in the process is
 * preferred, so a communicator handed to hala_rt_comm_attach meets the RCCL it came from): a host that renders on one GPU loads
 * libhalart.so without RCCL installed.  One process (or thread) per GPU; every rank calls the same sequence.
 *   hala_rt_comm_unique_id   : ncclGetUniqueId — rank 0 makes the 128-byte id and hands it to the other ranks over the host
 *                               application's own channel (MPI, a socket, torch.distributed.broadcast ...)
 *   hala_rt_comm_init_rank   : ncclCommInitRank on the renderer's device; rank / world must equal hala_rt_set_tile_shard's.
 *   hala_rt_comm_attach      : use a communicator the caller owns (an ncclComm_t) instead; it is not destroyed by the library.
 *   hala_rt_tile_allgather   : aov_mask bit k = AOV k (0 accum, 1 albedo, 2 normal, 3 final).  Gathers the rank's tile buffers and
 *                               de-interleaves them into this renderer's row-major images (read_image / save_images then work on
 *                               every rank).  Stream-ordered, never blocks the host: the collective runs on a side stream behind
 *                               the updates enqueued so far, and the renderer's stream waits for it.
 *   _begin / _finish         : the pipelined form: begin(k) snapshots the tile buffers (frame k + 1 may then overwrite them) and
 *                               starts the collective; it runs beside the rendering of frame k + 1 until finish() — called by the
 *                               next begin(), or explicitly — de-interleaves.  xGMI is point-to-point: a ring all-gather of
 *                               N x 33 MB is bound by one link per hop and can take as long as rendering a rank's share.
 *   hala_rt_get_gathered_buffer: the [world][tiles_per_rank][ts][ts][4] receive buffer of the last collective (tests).
 *   hala_rt_tile_allgather_begin_external + hala_rt_get_exchange_buffers: the same pipeline with the exchange done by the CALLER —
 *                               another transport (MPI, a CPU rehearsal of N ranks on one GPU), no communicator needed: begin_external
 *                               snapshots the tile buffers exactly like _begin; the caller then moves every rank's staging buffer
 *                               (rank k's at offset k x staged_bytes) into the receive buffer, stream-ordered on the exchange stream
 *                               the call returns, and calls _finish.
 * hala_rt_set_tile_shard completes a collective in flight and refuses to change rank / world while a communicator is attached. */
#define HALA_COMM_UNIQUE_ID_BYTES 128
int hala_rt_comm_unique_id(void* out_128_bytes);
int hala_rt_comm_init_rank(hala_rt_renderer* r, const void* unique_id_128_bytes, uint32_t rank, uint32_t world);
int hala_rt_comm_attach(hala_rt_renderer* r, void* nccl_comm);
int hala_rt_comm_destroy(hala_rt_renderer* r);
int hala_rt_tile_allgather(hala_rt_renderer* r, uint32_t aov_mask);
int hala_rt_tile_allgather_begin(hala_rt_renderer* r, uint32_t aov_mask);
int hala_rt_tile_allgather_finish(hala_rt_renderer* r);
int hala_rt_get_gathered_buffer(hala_rt_renderer* r, int which, void** d_ptr, size_t* bytes);
int hala_rt_tile_allgather_begin_external(hala_rt_renderer* r, uint32_t aov_mask);
int hala_rt_get_exchange_buffers(hala_rt_renderer* r, int which, void** d_staged, size_t* staged_bytes, void** d_receive, size_t* receive_bytes,
                                 void** hip_stream);
/* the same launched on a stream of the caller's (NULL: the renderer's): the de-interleave of frame k can then run beside the rendering
 * of frame k + 1.  The caller orders it against the renderer's stream (hala_rt_get_stream) before anything reads the images. */
int hala_rt_scatter_gathered_tiles_on_stream(hala_rt_renderer* r, int which, const void* d_gathered, size_t bytes, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * The ray-batch operator under the renderer: what vkCmdTraceRaysKHR + the closest-hit stage do for
 * one batch (src/rt_renderer.rs:458-464, src/raytracing_program.rs:330-340).
 * ---------------------------------------------------------------------------------------------- */
typedef struct hala_ray {
  float origin[3];
  float tmin;        /* a negative tmin is clamped to +0 when the ray is set up (RENDER_SPEC 4.2: rays start at or after their origin) —
                      * in every traversal variant, whatever the scene's size */
  float direction[3];
  float tmax;
} hala_ray; /* 32 B */

typedef struct hala_hit {
  float t;       /* < 0 => miss */
  float u;
  float v;
  uint32_t prim; /* global triangle id (instance order, then triangle order); HALA_INVALID_INDEX on miss */
} hala_hit; /* 16 B */

/* mode 0: closest hit; mode 1: any hit (shadow): t = 1 if occluded else -1. Rays/hits are DEVICE
 * pointers (coalesced 32-B / 16-B records). If d_counters != NULL (device, 2 x uint64) the kernel
 * also adds the number of BVH nodes visited and triangles tested (for the algorithmic-bytes figure).
 * Streams: hip_stream = NULL launches on the renderer's stream.  The launch uses per-renderer scratch (work counters,
 * step counters, the traversal-stack spill area) that update() uses too, so all update / trace_rays launches of ONE
 * renderer are serialised on the device: a call on another stream first makes that stream wait (hipStreamWaitEvent)
 * for the previous such launch, wherever it ran, and records an event behind its own.  Calls may come from one host
 * thread at a time (the reference's renderer is !Send / !Sync too). */
int hala_rt_trace_rays(hala_rt_renderer* r, const hala_ray* d_rays, hala_hit* d_hits, uint32_t count,
                       int mode, uint64_t* d_counters, void* hip_stream);
/* trace_rays_indirect (src/raytracing_program.rs:338-340): d_indirect points at a device-resident
 * VkTraceRaysIndirectCommandKHR {uint32 width, height, depth}; width*height*depth rays are traced. */
int hala_rt_trace_rays_indirect(hala_rt_renderer* r, const hala_ray* d_rays, hala_hit* d_hits,
                                const uint32_t* d_indirect, int mode, void* hip_stream);
/* Host-pointer convenience wrapper used by the tests (copies in/out around the same kernel). */
int hala_rt_trace_rays_host(hala_rt_renderer* r, const hala_ray* rays, hala_hit* hits, uint32_t count,
                            int mode, uint64_t counters[2]);

/* BVH introspection for the oracle cross-check: 64-B nodes + 48-B triangles as laid out in HBM.
 * node_width is 4: compressed 4-wide nodes (docs/RENDER_SPEC.md §4.1b). */
typedef struct hala_bvh_info {
  uint32_t node_count;
  uint32_t triangle_count;
  uint32_t max_depth;
  uint32_t lds_node_count; /* nodes staged in LDS by the traversal kernel */
  float scene_min[3];
  float scene_max[3];
  uint32_t node_width;
  /* two-level trees (RENDER_SPEC 4.5; scenes in which several instances reference one primitive): the triangles the trees store (every
   * instanced primitive once; == triangle_count otherwise), the nodes of the instance levels (the first nodes of the array; 0: one-level
   * tree) and the instance references their leaves index */
  uint32_t stored_triangle_count;
  uint32_t instance_node_count;
  uint32_t instance_ref_count;
  uint64_t tree_bytes; /* device bytes of nodes + triangles (+ any-hit copy) + shading records + instance tables */
} hala_bvh_info;
int hala_rt_get_bvh_info(hala_rt_renderer* r, hala_bvh_info* out);
/* node_count nodes and stored_triangle_count triangles.  Two-level trees: child references are absolute; a reference with bits 31..28 = 0xF
 * is an instance leaf whose low 28 bits index the records of hala_rt_download_instance_refs (64 B each: 3 rows of world -> object and the
 * translation as 12 floats, then root node, global id of the instance's first triangle, first shading record, instance index); the
 * triangles of an instanced primitive are in object space and carry ids local to the primitive. */
int hala_rt_download_bvh(hala_rt_renderer* r, void* nodes_64B, void* triangles_48B);
int hala_rt_download_instance_refs(hala_rt_renderer* r, void* refs_64B, uint32_t capacity, uint32_t* count);
/* Refit after vertex/transform edits (north_star "BVH build/refit"; the reference rebuilds only):
 * re-flattens instances with the given node local transforms and refits AABBs bottom-up on the GPU. */
int hala_rt_update_node_transform(hala_rt_renderer* r, uint32_t node_index, const float local_transform[16]);
/* Deforming geometry: replaces the vertices of primitive `primitive_index` of mesh `mesh_index` (indices into the scene handed
 * to hala_rt_set_scene, cpu/mesh.rs: HalaMesh::primitives).  The vertex count must be the primitive's own (the topology, i.e. the
 * index buffer, stays); host pointer, copied before the call returns.  Takes effect at the next hala_rt_refit. */
int hala_rt_update_vertices(hala_rt_renderer* r, uint32_t mesh_index, uint32_t primitive_index, const hala_vertex* vertices,
                            uint32_t vertex_count);
/* Replaces material `material_index` of the scene (cpu::HalaMaterial, the record hala_rt_set_scene took); texture indices must stay
 * within the scene's textures.  Takes effect at the next hala_rt_refit (which re-publishes the packed 144-B records; the geometry
 * is untouched, so is the tree). */
int hala_rt_update_material(hala_rt_renderer* r, uint32_t material_index, const hala_material_desc* material);
/* Applies the edits: node hierarchies, materials, camera / light / instance records, and — if an instance's transform or a primitive's vertices
 * changed — the tree (topology kept, boxes re-derived).  A move of camera or light nodes alone leaves the tree untouched.  The
 * accumulation restarts either way. */
int hala_rt_refit(hala_rt_renderer* r);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone pieces of the path (usable without a renderer)
 * ---------------------------------------------------------------------------------------------- */
/* EnvMap::build_distribution_maps (src/envmap.rs:239-388) on the GPU. pixels: RGBA32F host, W*H*4. */
int hala_envmap_build_distribution(int device_ordinal, const float* rgba32f, uint32_t width,
                                   uint32_t height, float* total_sum, float* marginal,
                                   float* conditional);
/* save_images' host tonemap (src/rt_renderer.rs:1256-1316) applied in place to RGBA32F pixels. */
void hala_tonemap_pixels(float* rgba32f, size_t pixel_count, int enable_tonemap, int enable_aces,
                         int use_simple_aces);
/* The decoder behind hala_rt_set_envmap_file (`image::open(path)` of src/envmap.rs:48-53 for the float formats): fills
 * width / height / channels (3 or 4); copies the row-0-is-top float pixels into dst when dst != NULL and capacity_floats
 * is large enough.  No GPU involved. */
int hala_load_float_image(const char* path, uint32_t* width, uint32_t* height, uint32_t* channels, float* dst,
                          size_t capacity_floats);
/* save_images' PFM writer (src/rt_renderer.rs:1318-1334). */
int hala_write_pfm(const char* path, const float* rgba32f, uint32_t width, uint32_t height);

/* HalaRayTracingProgramDesc (src/raytracing_program.rs:25-55): parses the serde JSON field names and
 * defaults; returns the parsed counts (used by the host mirror of HalaRayTracingProgram::new). */
typedef struct hala_rtprog_desc_info {
  uint32_t raygen_count;
  uint32_t miss_count;
  uint32_t hit_count;
  uint32_t callable_count;
  uint32_t push_constant_size;
  uint32_t binding_count;
  uint32_t ray_recursion_depth;
} hala_rtprog_desc_info;
int hala_rtprog_parse_desc(const char* desc_json, hala_rtprog_desc_info* out);

/* HalaRayTracingProgram (src/raytracing_program.rs:70-341), one export per method: the generic "RT pass" object an application
 * builds beside the renderer.  The shader paths of the description are recorded (SPIR-V has no meaning for the HIP kernels), the
 * pipeline is the library's traversal kernel pair, bind() takes the device buffers of one ray batch where the reference takes
 * descriptor sets, and bytes 0..3 of the push-constant block select the hit group: 0 closest hit, 1 any hit.
 *   hala_rtprog_create              <- HalaRayTracingProgram::new (:85-252): desc_json = the serde form of HalaRayTracingProgramDesc
 *                                      (:33-55); the renderer supplies device + acceleration structure (logical_device and
 *                                      descriptor_set_layouts there); fails on a parse error or an empty raygen list
 *   hala_rtprog_bind                <- bind (:264-278)
 *   hala_rtprog_push_constants{,_f32} <- push_constants / push_constants_f32 (:285-322): offset + length must lie inside
 *                                      push_constant_size (at least 4 bytes are kept for the mode word)
 *   hala_rtprog_trace_rays          <- trace_rays(index, command_buffers, width, height, depth) (:330-332): width*height*depth rays of
 *                                      the bound batch; hip_stream as in hala_rt_trace_rays (the command buffer of the reference)
 *   hala_rtprog_trace_rays_indirect <- trace_rays_indirect (:338-340): device-resident {width, height, depth}
 * The program borrows the renderer: destroy it before the renderer. */
typedef struct hala_rtprog hala_rtprog;
int hala_rtprog_create(hala_rt_renderer* r, const char* desc_json, const char* debug_name, hala_rtprog** out);
void hala_rtprog_destroy(hala_rtprog* p);
int hala_rtprog_get_desc_info(const hala_rtprog* p, hala_rtprog_desc_info* out);
int hala_rtprog_bind(hala_rtprog* p, const hala_ray* d_rays, hala_hit* d_hits);
int hala_rtprog_push_constants(hala_rtprog* p, uint32_t offset, const void* data, size_t len);
int hala_rtprog_push_constants_f32(hala_rtprog* p, uint32_t offset, const float* data, size_t count);
int hala_rtprog_trace_rays(hala_rtprog* p, uint32_t width, uint32_t height, uint32_t depth, void* hip_stream);
int hala_rtprog_trace_rays_indirect(hala_rtprog* p, const uint32_t* d_indirect, void* hip_stream);

const char* hala_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HALART_H */
