// hala_types.h — record layouts shared by the host code and the HIP kernels of libhalart.so.
//
// The hala_* records are the reference's #[repr(C)] device structs (cited in include/halart.h); the
// static_asserts pin the byte sizes SURVEY.md §8a verified.  The rt:: records are this library's own
// HBM formats (DESIGN.md "Data layout in HBM").
#pragma once
#include <hip/hip_vector_types.h>

#include <cstddef>
#include <cstdint>

#include "../../include/halart.h"

static_assert(sizeof(hala_vertex) == 44, "HalaVertex is 44 B (src/scene/vertex.rs:2-9)");
static_assert(sizeof(hala_gpu_camera) == 80, "gpu::HalaCamera is 80 B (src/scene/gpu/camera.rs:10-20)");
static_assert(offsetof(hala_gpu_camera, forward) == 48 && offsetof(hala_gpu_camera, yfov) == 60 && offsetof(hala_gpu_camera, type) == 72, "HalaCamera offsets");
static_assert(sizeof(hala_gpu_light) == 80, "gpu::HalaLight is 80 B (src/scene/gpu/light.rs:7-32)");
static_assert(offsetof(hala_gpu_light, v) == 48 && offsetof(hala_gpu_light, radius) == 60 && offsetof(hala_gpu_light, type) == 68, "HalaLight offsets");
static_assert(sizeof(hala_aabb) == 24, "HalaAABB is 24 B");
static_assert(sizeof(hala_gpu_material) == 144, "gpu::HalaMaterial is 144 B (src/scene/gpu/material.rs:6-48)");
static_assert(offsetof(hala_gpu_material, base_color) == 32 && offsetof(hala_gpu_material, ior) == 112 && offsetof(hala_gpu_material, type) == 140, "HalaMaterial offsets");
static_assert(sizeof(hala_gpu_mesh_data) == 96, "gpu::HalaMeshData is 96 B (src/scene/gpu/mesh.rs:32-39)");
static_assert(offsetof(hala_gpu_mesh_data, material_index) == 64 && offsetof(hala_gpu_mesh_data, vertices) == 72 && offsetof(hala_gpu_mesh_data, indices) == 80, "HalaMeshData offsets");
static_assert(sizeof(hala_global_uniform) == 112, "HalaGlobalUniform is 112 B (src/rt_renderer.rs:44-65)");
static_assert(offsetof(hala_global_uniform, resolution) == 32 && offsetof(hala_global_uniform, frame_index) == 48 && offsetof(hala_global_uniform, env_total_sum) == 68 && offsetof(hala_global_uniform, num_of_lights) == 96, "HalaGlobalUniform offsets");
static_assert(sizeof(hala_ray) == 32 && sizeof(hala_hit) == 16, "ray batch records");

namespace rt {

constexpr uint32_t kAbsent = 0xffffffffu;
constexpr float kTMax = 3.402823466e+38f;
constexpr float kNoNeePdf = 1e18f;  // RENDER_SPEC 7.1f: power_heuristic(kNoNeePdf, b) == 1 for every pdf b a light or the env map reports

// 64-B compressed BVH4 node (RENDER_SPEC §4.1b): up to four children in the bytes a plain BVH2 node would take.  Child boxes are
// 8-bit quantised against the node's own box: lo = pmin + qlo * 2^e, hi = pmin + qhi * 2^e per axis (lo rounded down,
// hi rounded up, so the quantised box always contains the true one).  Halves the dependent fetches per ray and the
// node bytes per ray of a BVH2 — measured +25-34 % rays/s on the 82 k and 1 M triangle scenes (profiles/r01_h_experiments.txt).
struct alignas(16) BvhNode4 {
  float pmin[3];
  uint32_t exps;     // byte a = biased float exponent of the quantum of axis a: quantum = uint_as_float(byte << 23)
  uint32_t qlo[3];   // byte c of qlo[a] = quantised low plane of child c on axis a (v_cvt_f32_ubyte{c} unpacks it)
  uint32_t qhi[3];
  uint32_t pad[2];
  uint32_t ref[4];   // 0xffffffff absent | bit31: leaf, bits 30..28 = count-1, bits 27..0 = first triangle | node index
};
static_assert(sizeof(BvhNode4) == 64, "BVH4 node is 64 B");
constexpr uint32_t kLeafRef = 0x80000000u;
// Two-level trees (RENDER_SPEC 4.5): the top levels of the node array are a tree over INSTANCES.  A child reference with bit 31 set and
// count bits 7 (large-scene builds emit leaves of <= 4 triangles, so the pattern is free) is an instance leaf: bits 27..0 index the
// InstRef table; the traversal moves the ray into the instance's object space and goes on at the root of its primitive's own tree,
// whose triangles are stored ONCE however many instances reference the primitive.
constexpr uint32_t kInstLeafTag = 0xF0000000u;
constexpr uint32_t kExitRef = 0xfffffffeu;  // stack sentinel: "leaving the instance" (the three entries below it hold the world-space ray)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool is_inst_leaf(uint32_t ref) { return (ref & kInstLeafTag) == kInstLeafTag && ref < kExitRef; }
// one instance of an instanced primitive: world -> object (rows of the inverse of the upper 3x3 of object -> world, and its translation:
// p' = rows * (p - tr); RENDER_SPEC 4.5), where its primitive's tree starts, and what turns the tree's local triangle numbers into
// global ids (instance order, RENDER_SPEC 3) and shading-record indices
struct alignas(64) InstRef {
  float r0[3], r1[3], r2[3], tr[3];
  uint32_t root;        // node index of the primitive's tree
  uint32_t gid_base;    // global id of the instance's first triangle
  uint32_t shade_base;  // index of the primitive's first shading record
  uint32_t inst;        // index into SceneView::primitives
};
static_assert(sizeof(InstRef) == 64, "instance reference is 64 B");
// per instance, in instance order (= ascending first_tri): how a global triangle id finds its shading record
struct InstInfo {
  uint32_t first_tri;   // global id of the instance's first triangle
  uint32_t shade_base;  // shading record of that triangle (shared by all instances of an instanced primitive)
  uint32_t instanced;   // 1: intersected in object space; its shading records hold object-space geometry
  uint32_t pad;
};

// 48-B triangle (RENDER_SPEC §4.1): v0|global id, e1 = v1-v0, e2 = v2-v0 in world space.
struct alignas(16) Tri {
  float v0[3]; uint32_t id;
  float e1[3]; uint32_t pad1;  // any-hit copy: bit 0 translucent, bit 1 boundary of a medium (RENDER_SPEC 7.1d / 7.1g)
  float e2[3]; uint32_t pad2;  // BVH-order copies: the shading kind of the triangle's material (kShadeKind*, 1..7) — see hit_encode
};
static_assert(sizeof(Tri) == 48, "triangle is 48 B");

// The renderer's own hit queue carries the shading kind of the hit triangle's material next to its id: prim word = id << 3 | kind.  The
// closest-hit traversal has the triangle's three 16-B words in registers when it accepts a hit (the kind travels in word 11), so the
// bounce shade can group its paths by kind without touching a shading record first.  id < 2^28 (leaf references), ids are distinct, so
// encoded words order exactly like ids: the closest-hit tie rule (t, then lower id) is unchanged.  kAbsent (no hit) stays all ones.
// Ray batches of the C ABI (hala_rt_trace_rays) get the plain id.
constexpr uint32_t kHitKindBits = 3;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t hit_encode(uint32_t id, uint32_t kind) { return (id << kHitKindBits) | (kind & 7u); }

// 128-B shading record of one triangle (global-id order), 128-B aligned: what the closest-hit stage interpolates, gathered once at
// build time from the vertex / index arenas so that shading a hit costs ONE dependent 64-B line (untextured materials) or two
// (textured) instead of a chain triangle -> instance -> indices -> three 44-B vertices.  The bounce-shade launches are bound by
// random 64-B fabric requests (profiles/r02_*): the 112-B record + the 48-B triangle it replaces straddled four lines on average.
// Vertex attributes are the LOCAL ones, untouched: the instance transform is applied to the interpolated normal exactly as
// RENDER_SPEC §6 says.  gcross = cross(e1, e2) of the world-space edges (the same fma form the shading would evaluate).
struct alignas(128) ShadeTri {
  float gcross[3]; uint32_t inst;      // line 1: geometric normal (unnormalised; its length is twice the area) | instance (node x primitive)
  float n0[3];     uint32_t material;  //         vertex normals | material index
  float n1[3];     uint32_t pad0;
  float n2[3];     uint32_t pad1;
  float uv[3][2];                      // line 2 (textured materials only): texture coordinates
  float tg[3][3];                      //         tangents (normal mapping)
  uint32_t pad2;
};
static_assert(sizeof(ShadeTri) == 128, "shading record is 128 B");

// One texture of set 2 binding 0 (src/rt_renderer.rs:197-226): a full mip chain of linear RGBA32F texels in the
// texture arena (8-bit sources are decoded once at upload; gen_mipmaps of gpu_uploader.rs:366-400 is a 2x2 box filter
// kernel).  mip_offset[l] = first texel of level l, in float4 units from the arena base.
constexpr uint32_t kMaxMips = 16;
// Texel storage (RENDER_SPEC 7.4).  Float images: linear RGBA32F, row-major, in the float arena.  8-bit images stay 8-bit — 4 B per
// texel in the byte arena, tiled 4x4 (a 64-B line holds a 4x4 block: a bilinear footprint touches 1.6 lines on average instead of 2.5,
// and the whole set of textures is a quarter of the RGBA32F size: the bounce shade is bound by random 64-B requests) — and the sampler
// decodes them: sRGB bytes through a 256-entry table (what the *_SRGB sampler of the reference does), UNORM bytes / 255.
constexpr uint32_t kTexFloat = 0, kTexSrgb8 = 1, kTexUnorm8 = 2;
struct TexDesc {
  uint32_t width, height, mips, format;
  uint32_t mip_offset[kMaxMips];  // first texel of level l in its arena (float4 units / tiled 4-B texels: level l holds ceil(w/4) x ceil(h/4) tiles of 16)
};
// index of texel (x, y) inside a tiled level of width w (x < w, y < h)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t tex_tiled_index(uint32_t x, uint32_t y, uint32_t w) { return (((y >> 2) * ((w + 3u) >> 2) + (x >> 2)) << 4) + ((y & 3u) << 2) + (x & 3u); }
inline uint32_t tex_tiled_size(uint32_t w, uint32_t h) { return ((w + 3u) >> 2) * ((h + 3u) >> 2) * 16u; }
static_assert(sizeof(TexDesc) == 80, "texture descriptor is 80 B");

// What every kernel of one update() sees (the "descriptor sets" of src/rt_renderer.rs:141-209, :671-745 as
// plain device pointers).
struct SceneView {
  const TexDesc* textures;   // set 2 binding 0, indexed by the material's *_map_index
  const float4* tex_arena;
  const uint32_t* tex_arena8;  // 8-bit images (RGBA bytes, tiled 4x4)
  const float* tex_lut;        // 512 floats: the sRGB EOTF, then b / 255 (shading.h::tex8_fetch)
  uint32_t texture_count;
  uint32_t shade_sort;  // 1: the scene's materials span several shading kinds — the bounce shade kernel regroups its paths by kind
  const BvhNode4* nodes;
  const Tri* tris;             // BVH order
  const Tri* tris_any;         // what the any-hit launches traverse (RENDER_SPEC 7.1d): == tris unless the scene has opacity-0 materials,
                               // whose triangles are degenerate (never hit) in this copy — no test in the kernels, the launcher swaps the pointer
  const Tri* tris_by_id;       // global-id order (for shading)
  const ShadeTri* shade_tris;  // global-id order: per-vertex attributes of the hit triangle in one record
  const uint32_t* inst_first_tri;
  const hala_gpu_mesh_data* primitives;  // set 1 binding 4
  const hala_gpu_material* materials;    // set 1 binding 3
  const uint8_t* material_kind;          // per material: the shading-kind bin (kShadeKind*) the bounce shade kernel groups paths by
  const hala_gpu_light* lights;          // set 1 binding 2
  const hala_gpu_camera* cameras;        // set 1 binding 1
  const float* env_pixels;       // RGBA32F (set 0 binding 6)
  const float* env_marginal;     // set 0 binding 7[0]
  const float* env_conditional;  // set 0 binding 7[1]
  uint32_t node_count, tri_count, lds_nodes, lds_tris;
  // two-level trees (RENDER_SPEC 4.5): 0 for scenes without instanced primitives — shading records are then indexed by global id
  const InstRef* inst_refs;
  const InstInfo* inst_info;
  uint32_t instance_count;
  uint32_t two_level;
  uint32_t any_translucent;   // 1: some material is translucent or bounds a medium (RENDER_SPEC 7.1d / 7.1g): the any-hit launches run their ALPHA variants
  uint32_t scatter_media;     // 1: some material holds a scattering medium (RENDER_SPEC 7.1f): the SCATTER shade kernels apply
  uint32_t simple_materials;  // 1: every material is an untextured, opaque DIFFUSE one without a medium (the SIMPLE shade kernels apply)
  float ray_eps;
  uint32_t staged;  // 1: the whole BVH fits the LDS budget (lds_nodes == node_count, lds_tris == tri_count) and is staged per workgroup
};

// Shading kinds: what decides most of k_shade's control flow.  Bounce paths reach the shade kernel in arbitrary order; inside a
// workgroup they are regrouped by kind so that a wave runs one flavour of the BSDF code (RENDER_SPEC is untouched: which lane
// shades which path is not observable).
constexpr uint32_t kShadeKindMiss = 0;       // no surface: environment / sky (or an analytic light in front of everything)
constexpr uint32_t kShadeKindFirst = 1;      // 1 + (DISNEY ? 2 : 0) + (textured ? 1 : 0); DISNEY with transmission: 5 + textured
constexpr uint32_t kShadeKindSpecial = 7;    // opacity < 1 or a participating medium
constexpr uint32_t kShadeKinds = 9;          // + 8: no path in this lane
inline uint8_t shade_kind_of(const hala_gpu_material& m, uint32_t texture_count) {
  if (m.opacity < 1.0f || m.medium_type != 0u) return (uint8_t)kShadeKindSpecial;
  const bool tex = m.base_color_map_index < texture_count || m.normal_map_index < texture_count ||
                   m.metallic_roughness_map_index < texture_count || m.emission_map_index < texture_count;
  if (m.type == 1u && m.specular_transmission > 0.0f) return (uint8_t)(5u + (tex ? 1u : 0u));
  return (uint8_t)(kShadeKindFirst + (m.type == 1u ? 2u : 0u) + (tex ? 1u : 0u));
}

// RENDER_SPEC 7.1d / 7.1g: how an any-hit ray treats the triangles of a material: 0 = every hit blocks, 1 = invisible (opacity exactly 0,
// no medium behind it), 2 = translucent (blocks with probability opacity x base-colour-map alpha, decided per (ray key, triangle)),
// 3 = invisible boundary of a medium (never blocks, adds to the ray's optical depth), 4 = translucent boundary of a medium
inline uint8_t any_class_of(const hala_gpu_material& m, bool base_map_has_alpha) {
  const bool medium = m.medium_type == 1u || m.medium_type == 2u;
  if (m.opacity == 0.0f) return medium ? 3 : 1;
  if (m.opacity < 1.0f || base_map_has_alpha) return medium ? 4 : 2;
  return 0;
}
constexpr uint32_t kAnyKeyLight = 0xA511E9B3u, kAnyKeyEnv = 0x63D83595u, kAnyKeyBatch = 0x5BD1E995u;

// Unsharded frames deal the pixels to path slots in 8 x 8 blocks (one block = the 64 lanes of a wave): camera rays of a square patch
// share far more of the tree (and of the textures at their hits) than those of a 64 x 1 strip.  Which slot renders which pixel is not
// observable (the RNG is keyed by the pixel id, images are written by pixel).
#ifndef RT_PIXEL_BLOCK
#define RT_PIXEL_BLOCK 8
#endif
constexpr uint32_t kPixelBlock = RT_PIXEL_BLOCK;  // 8: 64 pixels per block; 0: row-major slots (A/B)
// per-update constants derived on the host from HalaGlobalUniform + camera 0 (RENDER_SPEC §5)
struct FrameConst {
  hala_global_uniform u;  // the 112-B record itself (src/rt_renderer.rs:408-427)
  float aspect, tan_half;
  float pixel_spread;  // angular size of one pixel: 2*tan_half / height (texture LOD, RENDER_SPEC §7.4)
  uint32_t width, height;
  // pixel-tile sharding (RENDER_SPEC §9)
  uint32_t tile_size, tiles_x, tiles_y, world, rank, tiles_per_rank, perm_a, perm_b;
  uint32_t blocks_x;     // world == 1: pixel blocks (kPixelBlock x kPixelBlock, one per wave) per row of blocks
  uint32_t pixel_slots;  // number of pixel slots this rank renders
  // sample batching: `samples` consecutive frames (frame_index .. frame_index+samples-1) travel through the wavefront
  // together; path slot = sample * pixel_slots + pixel slot.  The resolve kernel folds them in frame order, so the
  // result is bit-identical to `samples` single-sample updates.
  uint32_t samples;
  uint32_t slot_count;  // pixel_slots * samples
};

// device control block: queue sizes and work counters of the wavefront loop.  One slot per bounce, so a single
// hipMemsetAsync per update() resets everything and no kernel has to clear a counter another one still reads.
constexpr uint32_t kMaxDepth = 64;
// Work distribution of the persistent traversal kernels: one returning atomic on a single word saturates at
// ~88 dequeues/us on MI355X (guide: "dequeue" row), which a 2 M-ray launch of 64-ray batches would hit.  The ray range
// is therefore cut into kWorkShards contiguous shards, each with its own counter on its own 128-B line; a wave starts
// on shard (blockIdx & 15) — blocks b and b+8 share an XCD — and moves on to the next shard when its own is dry.
#ifndef RT_WORK_SHARDS
#define RT_WORK_SHARDS 16
#endif
constexpr uint32_t kWorkShards = RT_WORK_SHARDS;  // power of two, <= 64 (the dry mask is one 64-bit word)
constexpr uint32_t kWorkStride = 32;  // uint32 words between shard counters (128 B)
constexpr uint32_t kWorkBatch = 128;  // rays handed out per dequeue (two 64-lane passes)
struct WorkCounters {
  uint32_t c[kWorkShards * kWorkStride];
  unsigned long long dry[kWorkStride / 2];  // dry[0]: bit s set once shard s has handed out all of its batches (own 128-B line)
};
struct Control {
  uint32_t n_active[kMaxDepth + 1];  // ray-queue size entering bounce d
  uint32_t n_shadow[2][kMaxDepth];   // connections produced by bounce d: [0] light, [1] environment
  uint32_t pad[3];
  WorkCounters work_closest;    // re-zeroed by the shade kernel of every bounce (it runs between two uses)
  WorkCounters work_shadow[2];
  unsigned long long rays_closest, rays_shadow;  // totals of this update
  // only filled by counting launches: steps[kind] = {nodes visited, triangles tested}; kind 0 closest-hit
  // kernel, kind 1 shadow / any-hit kernel
  unsigned long long steps[2][2];
  // counting launches, wave level: probe[kind] = {wave steps (one node visit by every lane that holds a ray), leaf passes
  // (executions of one leaf-test copy by a wave), lanes taking part in those passes}: SIMT utilisation of the two code paths
  unsigned long long probe[2][3];
  unsigned long long primary_steps[2];  // counting launches: steps[0] as it stood after the depth-0 launch (camera rays only)
};

// per-path records indexed by path slot (everything a live path needs from bounce to bounce travels in its queue entry)
struct P3 { float x, y, z; };  // 12-B per-path record: one dwordx3 load / store, no padding word to move
struct PathState {
  P3* radiance;            // L of the path: set by the depth-0 shade, added to by later shades (rarely: light hit,
                           // environment, emission) and by the shadow passes
  P3* radiance_env;        // Le of the path: the unoccluded environment connections (their own sum, RENDER_SPEC 6: the two shadow
                           // passes of a bounce then never touch the same word and may share a launch)
  P3* albedo;              // first-hit AOVs of this sample
  P3* normal;
};

struct ShadowEntry {  // 48 B: one NEE connection = shadow ray + the contribution it carries if unoccluded
  hala_ray ray;
  float contrib[3];  // throughput * f * Le * cos * weight / pdf, already multiplied out (RENDER_SPEC §6.5-6.6)
  uint32_t slot;     // pixel slot whose radiance receives it
};
static_assert(sizeof(ShadowEntry) == 48, "shadow entry is 48 B");

struct Queues {
  hala_ray* rays[2];  // bounce rays; tmin / tmax are implied (0 / FLT_MAX): their fields carry the path slot / the RNG counter
  float4* state[2];  // throughput.xyz | pdf of the last BSDF sample, next to the ray of the same queue entry
  hala_hit* hits;
  uint32_t* perm;  // bounce launches of multi-kind scenes: the order in which k_shade takes the queue's entries (k_shade_sort: by shading kind inside windows)
  // compact connection queues of the current bounce: [0] light NEE, [1] environment NEE.  A path owns at most one entry
  // per queue, the two queues are traced by consecutive launches, so contributions land in spec order without atomics.
  ShadowEntry* shadow[2];
};

}  // namespace rt
