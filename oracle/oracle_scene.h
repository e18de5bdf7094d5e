// ORACLE — TEST INFRASTRUCTURE ONLY (see spec_math.h).
// oracle_scene.h — the flattened world-space scene + BVH2 the CPU integrator works on (RENDER_SPEC §3, §4).
#pragma once
#include <cstdint>
#include <vector>

#include "oracle_api.h"
#include "spec_math.h"

namespace orc {

// RENDER_SPEC §4.1 — 64-B node: two child boxes, two child refs, two counts.
// count == 0: `child` is a node index; count > 0: `child` is the first triangle (BVH order) of a leaf.
// An absent child has an inverted box (min = +inf, max = -inf), child = 0, count = 0 and is never entered.
struct Node {
  float c0min[3], c0max[3], c1min[3], c1max[3];
  uint32_t child0, child1, count0, count1;
};
static_assert(sizeof(Node) == 64, "node is 64 B");

// RENDER_SPEC §4.1b — 64-B compressed 4-wide node as the product lays it out in HBM
struct Node4 {
  float pmin[3];
  uint32_t exps;    // byte a: biased exponent of the quantum of axis a
  uint32_t qlo[3];  // byte c: child c's quantised low plane on axis a
  uint32_t qhi[3];
  uint32_t pad[2];
  uint32_t ref[4];  // 0xffffffff absent | bit31 leaf, bits 30..28 count-1, bits 27..0 first triangle | node index
};
static_assert(sizeof(Node4) == 64, "4-wide node is 64 B");

// RENDER_SPEC §4.1 — 48-B triangle: v0 | global id, e1 = v1 - v0 | 0, e2 = v2 - v0 | 0
struct Tri {
  float v0[3]; uint32_t id;
  float e1[3]; uint32_t pad1;
  float e2[3]; uint32_t pad2;
};
static_assert(sizeof(Tri) == 48, "triangle is 48 B");

struct Instance {
  float transform[16];
  uint32_t material_index;
  uint32_t first_triangle;  // global id of its first triangle
  const orc_vertex* vertices;
  const uint32_t* indices;
  // RENDER_SPEC 4.5: an instance of a primitive that several instances reference is intersected in OBJECT space — its triangles keep
  // their local positions, the ray is moved: p' = rows * (p - tr) with rows = the inverse of the upper 3x3 of `transform`
  bool instanced = false;
  float r0[3], r1[3], r2[3], tr[3];
};

// a product tree's instance reference (RENDER_SPEC 4.5; include/halart.h: hala_rt_download_instance_refs), 64 B
struct InstRef {
  float r0[3], r1[3], r2[3], tr[3];
  uint32_t root, gid_base, shade_base, inst;
};
static_assert(sizeof(InstRef) == 64, "instance reference is 64 B");

// RENDER_SPEC §7.4: one image decoded to linear RGBA32F with its full 2x2-box mip chain
struct Image {
  uint32_t width = 0, height = 0, mips = 0;
  std::vector<std::vector<float>> levels;  // RGBA32F per level
  bool has_alpha = false;                  // some texel of level 0 has alpha < 1 (RENDER_SPEC 7.1d: cut-out materials)
};

struct EnvMap {
  uint32_t width = 0, height = 0;
  std::vector<float> pixels;  // RGBA32F
  float total_sum = 0.0f;
  std::vector<float> marginal, conditional;
};

}  // namespace orc

struct orc_scene {
  std::vector<orc::Instance> instances;
  std::vector<uint32_t> tri_instance;  // global triangle id -> instance
  std::vector<orc::Tri> tris_by_id;    // indexed by global id; world space, except the triangles of instanced instances (object space: RENDER_SPEC 4.5)
  std::vector<float> tri_verts9;       // world-space v0,v1,v2 per global id (RENDER_SPEC §3)
  std::vector<orc::Tri> tris;          // BVH order
  std::vector<orc::Tri> tris_any;      // RENDER_SPEC 7.1d: what the any-hit traversals see — the triangles of opacity-0 materials made
                                       // degenerate (e1 = e2 = 0: never hit), those of translucent materials flagged (pad1 = 1);
                                       // empty when the scene has neither (then `tris` serves both)
  std::vector<orc::Node> nodes;
  float bounds_min[3], bounds_max[3];
  float ray_eps;
  std::vector<orc_gpu_material> materials;
  orc_gpu_camera cameras[8]; int camera_count = 0;
  orc_gpu_light lights[32]; int light_count = 0;
  // owned copies of the caller's vertex / index arrays
  std::vector<std::vector<orc_vertex>> owned_vertices;
  std::vector<std::vector<uint32_t>> owned_indices;
  orc::EnvMap env;
  std::vector<orc::Image> images;       // per cpu::HalaImageData
  std::vector<uint32_t> texture_image;  // texture index -> image (gpu_uploader.rs:336-338)
  // optional: a tree handed over by the product (orc_scene_use_bvh4); the integrator then traverses IT (RENDER_SPEC §4.4b)
  std::vector<orc::Node4> ext_nodes;
  std::vector<orc::Tri> ext_tris, ext_tris_any;
  std::vector<orc::InstRef> ext_refs;   // ... and the instance references of its two-level form
};

namespace orc {
struct Counters { uint64_t nodes = 0, tris = 0; };
struct Hit { float t, u, v; uint32_t prim; };
// RENDER_SPEC §4: closest / any traversal over (nodes, tris); s (may be null: no instancing) tells which triangles are intersected in object space
Hit trace_closest(const orc_scene* s, const Node* nodes, const Tri* tris, V3 o, V3 d, float tmin, float tmax, Counters* c);
// RENDER_SPEC 7.1d / 7.1g, any-hit rays: 0 = every hit blocks, 1 = invisible (opacity exactly 0, no medium behind it), 2 = translucent
// (blocks with probability opacity x base-colour-map alpha, decided per (ray key, triangle)), 3 = invisible boundary of a medium (never
// blocks, adds to the ray's optical depth), 4 = translucent boundary of a medium (2, and 3 when it lets the ray through)
int any_class(const orc_scene* s, uint32_t material_index);
bool invisible(const orc_scene* s, uint32_t tri_id);
// `in` with the triangles of invisible materials made degenerate and those of translucent ones flagged (empty when the scene has neither)
void make_any_triangles(const orc_scene* s, const std::vector<Tri>& in, std::vector<Tri>* out);
// what an any-hit ray needs to decide whether a flagged triangle blocks it
// tau: the ray's optical depth per channel in 2^-16 units, summed with 32-bit wrap-around arithmetic (order-independent; 7.1g)
struct AnyCtx { const orc_scene* s; uint32_t key; uint32_t tau[3]; };
float hit_alpha(const orc_scene* s, uint32_t prim, float u, float v);  // opacity x base-colour-map alpha (bilinear, level 0) at a hit
// a triangle of the any-hit copy was hit inside (tmin, tmax) at distance t with Moeller-Trumbore determinant det: true = it blocks the ray
bool any_hit_event(AnyCtx* ax, const Tri& tr, float t, float det, float u, float v);
bool trace_any(const orc_scene* s, const Node* nodes, const Tri* tris, V3 o, V3 d, float tmin, float tmax, AnyCtx* ax, Counters* c);
V3 any_transmittance(const AnyCtx& ax);  // exp_neg(-max(tau, 0) / 65536) per channel
// the same on the scene's tree of choice: the product's 4-wide tree if one was handed over, else the oracle's own BVH2
Hit scene_trace_closest(const orc_scene* s, V3 o, V3 d, float tmin, float tmax, Counters* c);
// trans (may be null): what an unblocked ray keeps after the media it crossed (7.1g)
bool scene_trace_any(const orc_scene* s, V3 o, V3 d, float tmin, float tmax, uint32_t key, Counters* c, V3* trans = nullptr);
constexpr uint32_t kAnyKeyLight = 0xA511E9B3u, kAnyKeyEnv = 0x63D83595u, kAnyKeyBatch = 0x5BD1E995u;
V3 tonemap_select(V3 color, int enable_tonemap, int enable_aces, int use_simple_aces);
}  // namespace orc
