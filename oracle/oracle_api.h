// ORACLE — TEST INFRASTRUCTURE ONLY (see spec_math.h).  C entry points of liboracle.so, loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product.
//
// The record layouts below restate the reference's #[repr(C)] device structs and cpu::HalaScene from the
// reference sources (file:line given per struct); they are declared here independently of include/halart.h so
// that the oracle stands alone.  tests/test_layouts.py checks both against the same ctypes definitions.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NONE 0xffffffffu

// src/scene/vertex.rs:2-9
typedef struct { float position[3], normal[3], tangent[3], tex_coord[2]; } orc_vertex;  // 44 B
// src/scene/gpu/camera.rs:10-20
typedef struct {
  float position[3], _p0, right[3], _p1, up[3], _p2, forward[3];
  float yfov, focal_distance_or_xmag, aperture_or_ymag;
  uint32_t type, _p3;
} orc_gpu_camera;  // 80 B
// src/scene/gpu/light.rs:7-32
typedef struct {
  float intensity[3], _p0, position[3], _p1, u[3], _p2, v[3];
  float radius, area;
  uint32_t type, _p3[2];
} orc_gpu_light;  // 80 B
typedef struct { float min[3], max[3]; } orc_aabb;  // 24 B (gpu_uploader.rs:169-180)
// src/scene/gpu/material.rs:6-48
typedef struct {
  float medium_color[3], medium_density, medium_anisotropy;
  uint32_t medium_type;
  float _medium_padding[2];
  float base_color[3], opacity, emission[3], anisotropic;
  float metallic, roughness, subsurface, specular_tint;
  float sheen, sheen_tint, clearcoat, clearcoat_roughness;
  float clearcoat_tint[3], specular_transmission;
  float ior, ax, ay;
  uint32_t base_color_map_index, normal_map_index, metallic_roughness_map_index, emission_map_index, type;
} orc_gpu_material;  // 144 B
// src/scene/gpu/mesh.rs:32-39
typedef struct {
  float transform[16];
  uint32_t material_index, _p0;
  uint64_t vertices, indices, _p1;
} orc_gpu_mesh_data;  // 96 B
// src/rt_renderer.rs:44-65
typedef struct {
  float ground_color[4], sky_color[4], resolution[2];
  uint32_t max_depth, rr_depth, frame_index, camera_index, env_type, env_map_width, env_map_height;
  float env_total_sum, env_rotation, env_intensity, exposure_value;
  uint32_t enable_tonemap, enable_aces, use_simple_aces, num_of_lights, _pad[3];
} orc_global_uniform;  // 112 B

// cpu::HalaScene view (src/scene/cpu/*.rs) — same field order as include/halart.h's hala_*_desc
typedef struct {
  const char* name;
  int32_t parent;
  float local_transform[16];
  uint32_t mesh_index, camera_index, light_index;
} orc_node_desc;
typedef struct {
  const uint32_t* indices; uint32_t index_count;
  const orc_vertex* vertices; uint32_t vertex_count;
  uint32_t material_index;
} orc_primitive_desc;
typedef struct { const orc_primitive_desc* primitives; uint32_t primitive_count; } orc_mesh_desc;
typedef struct {
  uint32_t type;
  float base_color[3], opacity, emission[3], anisotropic, metallic, roughness, subsurface, specular_tint;
  float sheen, sheen_tint, clearcoat, clearcoat_roughness, clearcoat_tint[3], specular_transmission, ior;
  uint32_t medium_type;
  float medium_color[3], medium_density, medium_anisotropy;
  uint32_t base_color_map_index, emission_map_index, normal_map_index, metallic_roughness_map_index;
} orc_material_desc;
typedef struct { float color[3], intensity; uint32_t light_type; float param0, param1; } orc_light_desc;
typedef struct { uint32_t type; float aspect, yfov, znear, zfar, focal_distance, aperture, xmag, ymag; } orc_camera_desc;
typedef struct { uint32_t format, width, height; const void* data; size_t num_of_bytes; } orc_image_desc;
typedef struct { uint32_t key, value; } orc_index_pair;
typedef struct {
  const orc_node_desc* nodes; uint32_t node_count;
  const orc_mesh_desc* meshes; uint32_t mesh_count;
  const orc_material_desc* materials; uint32_t material_count;
  const orc_light_desc* lights; uint32_t light_count;
  const orc_camera_desc* cameras; uint32_t camera_count;
  const orc_index_pair* texture2image_mapping; uint32_t texture_count;
  const orc_index_pair* image2data_mapping; uint32_t image_count;
  const orc_image_desc* image_data; uint32_t image_data_count;
} orc_scene_desc;

// ---- A1: EnvMap::build_distribution_maps (src/envmap.rs:239-388) -------------------------------------------
// pixels: RGBA32F, W*H*4 floats, row-major.  Sequential f32 sums in the reference's order.
void orc_envmap_build_distribution(const float* rgba, uint32_t width, uint32_t height, float* total_sum,
                                   float* marginal, float* conditional);
// src/envmap.rs:63-89: returns 0 if all finite, 1 if a NaN is present, 2 if an infinity is present
// (first offending channel in scan order decides, as in the reference).
int orc_envmap_validate(const float* pixels, uint32_t channels, uint32_t width, uint32_t height);

// ---- A8/A9/A10/A11/A13: what HalaSceneGPUUploader::upload packs --------------------------------------------
// src/scene/cpu/scene.rs:99-114
void orc_update_node_hierarchies(const orc_scene_desc* scene, float* world_transforms /* node_count*16 */);
// src/scene/gpu/material.rs:51-110
void orc_pack_material(const orc_material_desc* in, orc_gpu_material* out);
// src/scene/loader/gpu_uploader.rs:99-122 + src/scene/gpu/camera.rs:28-61; returns number packed or -1
int orc_pack_cameras(const orc_scene_desc* scene, orc_gpu_camera* out /* [8] */);
// src/scene/loader/gpu_uploader.rs:148-293; returns number packed
int orc_pack_lights(const orc_scene_desc* scene, orc_gpu_light* out /* [32] */, orc_aabb* aabbs /* [32] */);
// src/scene/loader/gpu_uploader.rs:843-885: instance list in node order then primitive order.
// transforms3x4: row-major 3x4 per instance (:854-858); mesh_data: transform + material index (addresses 0).
// returns the number of instances (excluding the trailing light instance).
int orc_pack_instances(const orc_scene_desc* scene, float* transforms3x4, orc_gpu_mesh_data* mesh_data,
                       uint32_t capacity);
// src/scene/loader/gpu_uploader.rs:460-467 + src/scene/bounds.rs:78-108 (center, extents)
void orc_primitive_bounds(const orc_vertex* vertices, uint32_t count, float center[3], float extents[3]);

// ---- A16: save_images (src/rt_renderer.rs:1256-1334) ---------------------------------------------------------
void orc_tonemap_pixels(float* rgba, size_t pixel_count, int enable_tonemap, int enable_aces, int use_simple_aces);
// writes the exact byte string of the PFM file into out (capacity >= 32 + 12*w*h); returns its length
size_t orc_pfm_bytes(const float* rgba, uint32_t width, uint32_t height, uint8_t* out, size_t capacity);

// ---- the scene as the integrator sees it ------------------------------------------------------------------
typedef struct orc_scene orc_scene;
// Flattens instances to world space (RENDER_SPEC §3), builds a binned-SAH BVH2 on the CPU.
orc_scene* orc_scene_create(const orc_scene_desc* desc);
void orc_scene_destroy(orc_scene* s);
// env map for the scene (RGBA32F); builds the A1 tables with orc_envmap_build_distribution
void orc_scene_set_envmap(orc_scene* s, const float* rgba, uint32_t width, uint32_t height);
uint32_t orc_scene_triangle_count(const orc_scene* s);
uint32_t orc_scene_node_count(const orc_scene* s);
void orc_scene_bounds(const orc_scene* s, float mn[3], float mx[3]);
// world-space triangles in global-id order: 9 floats each (v0, v1, v2)
void orc_scene_get_triangles(const orc_scene* s, float* out9);

// ---- ray-batch operator (RENDER_SPEC §4) ---------------------------------------------------------------------
typedef struct { float origin[3], tmin, direction[3], tmax; } orc_ray;
typedef struct { float t, u, v; uint32_t prim; } orc_hit;
// mode 0 closest / 1 any. counters (may be NULL): [0] += nodes visited, [1] += triangles tested.
void orc_trace_rays(const orc_scene* s, const orc_ray* rays, orc_hit* hits, uint32_t count, int mode,
                    uint64_t* counters);
// Brute force over all triangles (no BVH): the BVH-independent truth for closest hits.
void orc_trace_rays_brute(const orc_scene* s, const orc_ray* rays, orc_hit* hits, uint32_t count, int mode);
// The traversal rule of RENDER_SPEC §4.4b over an externally supplied BVH in the product's HBM layout (64-B compressed
// 4-wide nodes §4.1b, 48-B triangles) — used to check the GPU-built BVH and to count nodes/triangles on it — and the
// structural validation of such a BVH: every triangle id appears exactly once, the dequantised child boxes contain
// their triangles, no cycles.  Returns 0 if valid, else a non-zero code; *max_depth receives the tree depth.
void orc_trace_rays_on_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                            const orc_ray* rays, orc_hit* hits, uint32_t count, int mode, uint64_t* counters);
int orc_validate_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                      const float* ref9, uint32_t* max_depth);

// Makes orc_render traverse a tree the product built (same layout as above; node_count 0 switches back to the oracle's own).
void orc_scene_use_bvh4(orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count);
/* two-level trees (RENDER_SPEC 4.5): + the 64-B instance references (include/halart.h: hala_rt_download_instance_refs) */
void orc_scene_use_bvh4_two_level(orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                                  const void* refs64, uint32_t ref_count);
void orc_trace_rays_on_bvh4_two_level(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count, const void* refs64,
                                      const orc_ray* rays, orc_hit* hits, uint32_t count, int mode, uint64_t* counters);
int orc_validate_bvh4_two_level(const orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                                const void* refs64, uint32_t ref_count, uint32_t* max_depth);
/* scenes created from now on: 0 = instanced primitives are intersected in object space (RENDER_SPEC 4.5; hala_rt_build_options::instancing
 * = 2), 1 (the default) = every instance is flattened to world space */
void orc_set_instancing_off(int off);
// The oracle's own SAH tree in the product's 4-wide format (greedy surface-area collapse, conservative quantisation) and
// its triangles in that tree's order: a CPU-built tree to pin the two functions above on, and a quality yardstick for
// the product's builder.  Returns the node count (query with nodes64_out == NULL); 0 on failure.
uint32_t orc_scene_export_bvh4(const orc_scene* s, void* nodes64_out, uint32_t capacity);
void orc_scene_get_bvh_triangles(const orc_scene* s, void* tris48_out);

// ---- the integrator (RENDER_SPEC §5-§8) ------------------------------------------------------------------------
typedef struct {
  uint32_t width, height;
  uint32_t max_depth, rr_depth;
  float ground_color[4], sky_color[4];
  float env_rotation_degrees, env_intensity, exposure_value;
  int enable_tonemap, enable_aces, use_simple_aces;
  int num_threads;  // <= 0: all hardware threads
} orc_render_params;
typedef struct {
  uint64_t rays_closest, rays_shadow;
  uint64_t nodes_visited, triangles_tested;  // over closest + shadow traversals
} orc_render_stats;
// Renders frames [first_frame, first_frame + frame_count) for the pixel rectangle [x0,x1) x [y0,y1) and folds
// them into accum/albedo/normal (RGBA32F, full W*H*4 images, row 0 = top) as the running mean of RENDER_SPEC §8
// (the images must hold the mean of frames [0, first_frame) on entry; first_frame == 0 ignores their content).
// final_rgba (may be NULL) receives the tonemapped image of RENDER_SPEC §8.
void orc_render(const orc_scene* s, const orc_render_params* p, uint32_t first_frame, uint32_t frame_count,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float* accum, float* albedo, float* normal,
                float* final_rgba, orc_render_stats* stats);
// The camera rays of frame `frame_index` (RENDER_SPEC §5), one per pixel, for ray-batch tests.
void orc_generate_camera_rays(const orc_scene* s, uint32_t width, uint32_t height, uint32_t frame_index,
                              orc_ray* rays);

// textures (RENDER_SPEC §7.4): mip chain + trilinear REPEAT fetch; uvl = (u, v, lod) triples
int orc_scene_texture_info(const orc_scene* s, uint32_t tex, uint32_t* w, uint32_t* h, uint32_t* mips);
void orc_scene_texture_level(const orc_scene* s, uint32_t tex, uint32_t level, float* out_rgba);
void orc_scene_sample_texture(const orc_scene* s, uint32_t tex, const float* uvl, uint32_t n, float* out_rgba);

// spec_math probes for tests (vectorised over n)
void orc_probe_sincos_2pi(const float* u, float* s, float* c, size_t n);
void orc_probe_acos(const float* x, float* out, size_t n);
void orc_probe_exp_neg(const float* x, float* out, size_t n);
void orc_probe_log(const float* x, float* out, size_t n);
void orc_probe_hg(const float* d3, float g, const float* u1, const float* u2, float* out3, size_t n);
void orc_probe_atan2(const float* y, const float* x, float* out, size_t n);
void orc_probe_rng(uint32_t pixel_id, uint32_t frame_index, float* out, size_t n);

// multi-GPU tile permutation (RENDER_SPEC §9): owner rank of tile t, and its slot in that rank's buffer
void orc_tile_assignment(uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t* owner, uint32_t* slot);

#ifdef __cplusplus
}
#endif
