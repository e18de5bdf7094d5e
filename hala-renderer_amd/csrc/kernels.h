// kernels.h — host-callable launchers of the HIP kernels of libhalart.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "hala_types.h"

namespace rt {

struct LaunchCfg {
  uint32_t persistent_blocks;  // grid of the persistent traversal kernels
  uint2* spill;                // global stack spill area or nullptr (stack need <= kStackLds)
  uint32_t refill;             // a wave refills its idle lanes once this many are idle (64 = whole-wave batches)
};

// integrator.hip
// `tree`: 0 = a large one-level tree, 1 = a tree staged whole in LDS, 2 = a two-level tree (RENDER_SPEC 4.5): each has its kernel variants
size_t traverse_fixed_lds_bytes(int tree);  // LDS bytes of one workgroup besides a staged BVH: per-lane stacks (+ leaf work lists)
uint32_t traverse_stack_lds_levels(int tree);  // stack entries kept in LDS
uint32_t traverse_max_leaf(bool staged);  // largest leaf (triangles) the traversal variant accepts
uint32_t traverse_stack_spill_levels();  // deeper entries spilled to global scratch (8 B each, per lane)
uint32_t traverse_blocks_per_cu(size_t dynamic_lds_bytes, int tree);  // resident workgroups per CU (occupancy query)
void launch_trace_batch(const LaunchCfg& lc, const SceneView& sv, const hala_ray* rays, hala_hit* hits, const uint32_t* n_ptr,
                        uint32_t n_imm, WorkCounters* work, Control* ctl, bool any, bool count, bool account, hipStream_t s);
void launch_trace_shadow(const LaunchCfg& lc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth,
                         uint32_t kind /* 0 light, 1 environment */, bool count, hipStream_t s);
// the last shadow pass of bounce `depth` and the closest-hit traversal of bounce depth + 1 as ONE persistent launch (false: not fusable)
bool launch_trace_shadow_then_batch(const LaunchCfg& lc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth,
                                    uint32_t kinds /* bit 0 light, bit 1 environment connections */, bool with_closest, hipStream_t s);
void launch_trace_primary(const LaunchCfg& lc, const SceneView& sv, const FrameConst& fc, hala_hit* hits, WorkCounters* work, Control* ctl,
                          uint32_t n_account /* real paths among fc.slot_count */, bool count, hipStream_t s);
void launch_shade(const FrameConst& fc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth, hipStream_t s);
void launch_resolve(const FrameConst& fc, const PathState& ps, float4* accum, float4* albedo, float4* normal, float4* final_img, hipStream_t s);
void launch_scatter_tiles(const FrameConst& fc, const float4* gathered, float4* full, hipStream_t s);
void launch_sample_texture(const SceneView& sv, uint32_t tex, const float* uvl, uint32_t n, float4* out, hipStream_t s);

// texture.hip — K8: one level of the mip chain (2x2 box filter over linear RGBA32F texels)
void launch_mip_downsample(const float4* src, uint32_t sw, uint32_t sh, float4* dst, uint32_t dw, uint32_t dh, hipStream_t s);
// 8-bit images: row-major bytes -> 4x4-tiled level 0; tiled level l - 1 -> tiled level l (decode, box filter, encode)
void launch_tile8(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t s);
void launch_mip_downsample8(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh, uint32_t format, const float* lut,
                            const float* thr, hipStream_t s);

// Tuning knobs read from the environment exist only in builds made with -DHALART_TUNING (scripts/variant_sweep.sh): the results and
// the speed of the release library do not depend on the caller's environment.
#ifdef HALART_TUNING
inline const char* tune_env(const char* name) { return getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif

// bvh_build.hip — K1/K3/K4: flatten instances to world space, LBVH build, refit
// How commit() builds the hierarchy (hala_rt_set_build_options; 0 = the default everywhere).  The driver fields only change HOW the
// host walks the rounds of a build, never the tree (tests/test_gpu_parity.py::test_ploc_drivers_build_the_same_tree).
struct BuildOptions {
  uint32_t builder = 0;              // 0 auto (full-sweep SAH from 4096 triangles, LBVH below) | 1 SAH | 2 PLOC | 3 LBVH
  uint32_t ploc_tail = 0;            // 0 / 1: the last PLOC rounds (<= 512 clusters) inside one workgroup | 2: every round its own launch
  uint32_t ploc_look_every = 0;      // PLOC rounds between two looks of the host at the device's counters (default 6)
  uint32_t collapse_look_every = 0;  // levels of the 4-wide collapse between two looks (default 8)
};
struct BvhBuffers {
  // inputs (device): the instances whose triangles this tree holds, in instance order
  const hala_gpu_mesh_data* primitives;  // per listed instance
  const uint32_t* inst_first_tri;        // [instance_count + 1] prefix of triangle counts WITHIN this tree
  uint32_t instance_count;
  uint32_t tri_count;
  // trees over a subset of the scene (RENDER_SPEC 4.5): null = the whole scene, flattened (triangle k of the tree has global id k)
  const uint32_t* gid_first = nullptr;   // per listed instance: global id of its first triangle
  const uint32_t* inst_index = nullptr;  // per listed instance: its index in the scene's instance list (kAbsent: shared by several)
  bool object_space = false;             // the tree of an instanced primitive: vertex positions as they are, triangle ids local
  // outputs (device, allocated by the caller)
  Tri* tris_by_id;         // [tri_count]
  Tri* tris;               // [tri_count] BVH order
  Tri* tris_any = nullptr; // [tri_count] BVH order: triangles of invisible materials degenerate, of translucent ones flagged (RENDER_SPEC 7.1d);
                           // null: the scene has neither
  const uint8_t* material_any_class = nullptr;  // per material: 0 blocks always, 1 invisible, 2 translucent (for tris_any)
  uint32_t material_count = 0;
  const uint8_t* material_kind = nullptr;       // per material: shading kind (hala_types.h: shade_kind_of), stamped into word 11 of the BVH-order triangles
  ShadeTri* shade_tris;    // [tri_count] global-id order
  BvhNode4* nodes;         // capacity >= max(tri_count - 1, 1)
  void* topology = nullptr;  // builder state kept for refit (freed with bvh_free_topology)
  BuildOptions opt;
  // results
  uint32_t node_count;
  uint32_t max_depth;    // levels of the emitted tree
  uint32_t stack_need;   // traversal stack entries a ray can need
  float scene_min[3], scene_max[3];
};
// Builds everything; returns "" on success or an error message.  Synchronises the stream before returning.
std::string bvh_build(BvhBuffers& b, uint32_t leaf_max, hipStream_t s);
// Re-flattens (instance transforms / vertices may have changed) and refits the existing topology bottom-up.
std::string bvh_refit(BvhBuffers& b, hipStream_t s);
// A tree built into a sub-range of the scene's node / triangle arrays numbers its nodes and triangles from 0: make its child references
// absolute (inner: += node_offset, leaf: first triangle += tri_offset).  After every bvh_build / bvh_refit of such a tree.
std::string bvh_relocate(BvhBuffers& b, uint32_t node_offset, uint32_t tri_offset, hipStream_t s);
void bvh_free_topology(void* topology);

// bvh_tlas.cpp — the instance levels of a two-level tree (RENDER_SPEC 4.5), built on the host.  An item is a world-space box with the
// child reference a node slot gets for it: an instance leaf (kInstLeafTag | InstRef index) or the root node of a subtree that needs no
// transform; `need` = traversal stack entries a ray can need behind it.  Returns the node count (nodes in breadth-first order, root 0).
struct TlasItem { float mn[3], mx[3]; uint32_t ref, need; };
uint32_t tlas_build(const std::vector<TlasItem>& items, std::vector<BvhNode4>& out, uint32_t* levels, uint32_t* stack_need);

// envmap.hip — A1: EnvMap::build_distribution_maps (src/envmap.rs:239-388) on the GPU
// d_rgba: W*H float4; outputs device pointers: total_sum[1], marginal[H], conditional[W*H]
std::string envmap_build_distribution(const float4* d_rgba, uint32_t width, uint32_t height, float* d_total_sum, float* d_marginal,
                                      float* d_conditional, hipStream_t s);

}  // namespace rt
