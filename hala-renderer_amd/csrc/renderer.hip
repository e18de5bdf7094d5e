// renderer.hip — HalaRenderer (src/rt_renderer.rs:568-1353) re-designed for one MI355X: the C++ object behind the
// C ABI of include/halart.h.  Descriptor sets become a struct of device pointers (rt::SceneView), trace_rays becomes the
// wavefront kernel sequence of integrator.hip, timestamp queries become HIP events, staging buffers become
// hipMemcpyAsync from the caller's memory.  Everything that computes runs on the GPU; this file only orchestrates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "dyn_api.h"
#include "hala_types.h"
#include "host_image.h"
#include "host_scene.h"
#include "host_util.h"
#include "kernels.h"

namespace rt { std::string decode_image_file_rgba8(const char* path, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgba); }  // gltf_loader.cpp

using namespace rt;

namespace {

constexpr uint32_t kLeafMax = 4;  // triangles per leaf, large scenes: the wave tests a leaf's triangles side by side (traverse.h), so fewer, fuller leaves win (profiles/r02_experiments.txt; 2 while each lane tested its own leaves)
constexpr uint32_t kLeafMaxStaged = 4;  // ... unless the whole tree sits in LDS: leaves of two packed pairs, fewer node steps (+2 % on Cornell)
constexpr size_t kLdsStageBudget = 40 * 1024;  // a BVH up to this size is staged whole in LDS (next to the 24-KB stack)
constexpr uint32_t kRefillThreshold = 24;      // idle lanes that trigger a refill of the wave (persistent_trace; profiles/r02_experiments.txt)
constexpr int kStatRing = 16;
constexpr uint32_t kMaxSampleBatch = 16;  // frames per wavefront pass in hala_rt_update_batch (~250 B of state per path)

// ---- RENDER_SPEC §2.2 on the host (for tan(yfov/2); same polynomials as rt_math.h) ---------------------------
float h_sin_poly(float a) {
  float a2 = a * a;
  float p = -2.50521083854417187751e-8f;
  p = std::fmaf(p, a2, 2.75573192239858906526e-6f);
  p = std::fmaf(p, a2, -1.98412698412698412698e-4f);
  p = std::fmaf(p, a2, 8.33333333333333333333e-3f);
  p = std::fmaf(p, a2, -1.66666666666666666667e-1f);
  p = std::fmaf(p, a2, 1.0f);
  return a * p;
}
float h_cos_poly(float a) {
  float a2 = a * a;
  float p = 2.08767569878680989792e-9f;
  p = std::fmaf(p, a2, -2.75573192239858906526e-7f);
  p = std::fmaf(p, a2, 2.48015873015873015873e-5f);
  p = std::fmaf(p, a2, -1.38888888888888888889e-3f);
  p = std::fmaf(p, a2, 4.16666666666666666667e-2f);
  p = std::fmaf(p, a2, -0.5f);
  p = std::fmaf(p, a2, 1.0f);
  return p;
}
void h_sincos_rad(float a, float* s, float* c) {
  float t = a * 0.15915494309189533577f;
  t = t - std::floor(t);
  if (t >= 1.0f) t = 0.0f;
  float x = t * 4.0f;
  int q = (int)x;
  float f = x - (float)q;
  float ang = f * 1.57079632679489661923f;
  float sa = h_sin_poly(ang), ca = h_cos_poly(ang);
  switch (q & 3) {
    case 0: *s = sa; *c = ca; break;
    case 1: *s = ca; *c = -sa; break;
    case 2: *s = -sa; *c = -ca; break;
    default: *s = -ca; *c = sa; break;
  }
}

uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }
uint32_t mod_inverse(uint32_t a, uint32_t n) {  // a^-1 mod n (a, n coprime); n == 1 -> 0
  long long t = 0, nt = 1, r = n, nr = a % n;
  while (nr != 0) { long long q = r / nr; long long tmp = t - q * nt; t = nt; nt = tmp; tmp = r - q * nr; r = nr; nr = tmp; }
  if (t < 0) t += n;
  return (uint32_t)t;
}

struct TraceEvents {
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  hipEvent_t frame_begin = nullptr, frame_end = nullptr;
  bool pending = false, counted = false;
  uint32_t samples = 1;  // frames rendered by this wavefront pass
  uint32_t shadow_launches = 0;  // k_trace_shadow launches inside the timed brackets of this pass (0, 1 or 2 per depth)
  // timed passes: bit d of fused_mask = the third bracket of depth d holds a fused launch (k_trace_shadow_then_batch); bit d of
  // traced_mask = the closest-hit pass of depth d ran inside depth d - 1's fused launch (its own bracket is empty)
  unsigned long long fused_mask = 0, traced_mask = 0;
  uint32_t* host_queue_sizes = nullptr;  // pinned: the head of the control block (n_active | n_shadow) as the pass left it
  unsigned long long* host_counts = nullptr;  // pinned: rays_closest, rays_shadow, steps[2][2], probe[2][3]
};

}  // namespace

struct hala_rt_renderer {
  std::string name;
  uint32_t width = 0, height = 0;
  int device = 0;
  uint32_t max_depth = 0, rr_depth = 0;
  bool enable_tonemap = false, enable_aces = false, use_simple_aces = false;
  uint64_t max_frames = 0;
  hipStream_t stream = nullptr;
  uint32_t cu_count = 256;

  float ground[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // src/rt_renderer.rs:799
  float sky[4] = {0.5f, 0.7f, 1.0f, 1.0f};     // :800
  float env_intensity = 1.0f, exposure = 1.0f, env_rotation = 0.0f;  // :798-803

  uint32_t n_raygen = 0, n_miss = 0, n_callable = 0, n_hit = 0;
  DeviceArray<uint8_t> blue_noise;
  uint32_t blue_w = 0, blue_h = 0;

  bool has_scene = false, committed = false;
  HostScene hs;
  DeviceArray<hala_vertex> d_vertices;
  DeviceArray<uint32_t> d_indices;
  std::vector<size_t> prim_vertex_offset, prim_index_offset;
  DeviceArray<hala_gpu_camera> d_cameras;
  DeviceArray<hala_gpu_light> d_lights;
  DeviceArray<hala_gpu_material> d_materials;
  DeviceArray<uint8_t> d_material_kind;
  std::vector<uint8_t> material_kind;  // host copy: a refit restamps the triangles when an edit changed a material's shading kind
  bool shade_sort = false, simple_materials = false, scatter_media = false;
  DeviceArray<hala_gpu_mesh_data> d_instances;
  DeviceArray<uint32_t> d_inst_first_tri;
  DeviceArray<float4> d_tex_arena;
  DeviceArray<uint32_t> d_tex_arena8;  // 8-bit images: RGBA bytes, tiled 4x4 (RENDER_SPEC 7.4)
  DeviceArray<float> d_srgb_lut, d_srgb_thr;
  DeviceArray<TexDesc> d_textures;
  std::vector<TexDesc> host_textures;

  BvhBuffers bvh{};
  // two-level trees (RENDER_SPEC 4.5): scenes in which some primitive is referenced by several instances.  `bvh` then only carries the
  // totals; the trees live in `blas` — [0] the world-space tree over the triangles of all instances that are NOT instanced (if any), then
  // one object-space tree per instanced primitive — as sub-ranges of the node / triangle / shading-record arrays, behind the instance
  // levels (the first tlas_capacity nodes), which are rebuilt on the host whenever a node moves.
  struct Blas {
    BvhBuffers b{};
    uint32_t node_off = 0, tri_off = 0, node_cap = 0;
    bool object_space = false;
    uint32_t prim = 0;                      // object_space: the primitive (index into hs.prims)
    std::vector<uint32_t> insts;            // world tree: the instances it holds, in instance order
    DeviceArray<hala_gpu_mesh_data> d_md;
    DeviceArray<uint32_t> d_first, d_gid, d_inst;
    ~Blas() { if (b.topology) bvh_free_topology(b.topology); }
  };
  std::vector<std::unique_ptr<Blas>> blas;
  bool two_level = false;
  uint32_t instancing_mode = 0;            // hala_rt_build_options::instancing: 0 automatic (by size), 1 never (everything flattened), 2 by the rule of RENDER_SPEC 4.5
  std::vector<uint8_t> inst_instanced;     // per instance: intersected in object space
  std::vector<int32_t> prim_blas;          // per primitive: index into blas, -1
  uint32_t tlas_capacity = 0, tlas_nodes = 0, stored_tris = 0;
  std::vector<InstRef> inst_refs;
  DeviceArray<InstRef> d_inst_refs;
  DeviceArray<InstInfo> d_inst_info;
  DeviceArray<Tri> d_tris_by_id, d_tris;
  DeviceArray<Tri> d_tris_any;
  DeviceArray<ShadeTri> d_shade_tris;
  DeviceArray<BvhNode4> d_nodes;
  uint32_t lds_nodes = 0, lds_tris = 0;
  bool staged = false;  // whole BVH staged in LDS by the traversal kernels
  uint32_t leaf_max_built = 0;
  float ray_eps = 0.0f;
  DeviceArray<uint2> d_spill;
  LaunchCfg lcfg{};
  uint32_t fuse_mode = 1;  // hala_rt_set_pass_fusion: 0 never, 1 untimed updates, 2 always (shadow passes of bounce d + closest-hit pass of bounce d + 1 in one launch)

  bool has_env = false;
  uint32_t env_w = 0, env_h = 0;
  DeviceArray<float4> d_env;
  DeviceArray<float> d_env_total, d_marginal, d_conditional;
  float env_total_sum = 0.0f;

  // tile shard (RENDER_SPEC §9)
  uint32_t real_pixels = 0;  // pixels among the rank's slot_count slots that exist in the frame
  uint32_t rank = 0, world = 1, tile_size = 32, tiles_x = 0, tiles_y = 0, tiles_per_rank = 0, perm_a_inv = 0, perm_b = 7;
  uint32_t slot_count = 0;      // pixel slots of this rank
  uint32_t blocks_x = 0;        // world == 1: 8 x 8 pixel blocks per row of blocks (hala_types.h: kPixelBlock)
  // pixels of this rank's image buffers: its tile slots when sharded, the row-major frame otherwise (whose path slots may hold padding)
  size_t image_pixels() const { return world <= 1 ? (size_t)width * height : (size_t)slot_count; }
  uint32_t batch_capacity = 1;  // samples the wavefront buffers can hold in flight (hala_rt_update_batch)

  DeviceArray<float4> img_local[4];  // accum, albedo, normal, final (slot order)
  DeviceArray<float4> img_full[4];   // row-major, only after scatter_gathered_tiles (world > 1)
  bool full_valid[4] = {false, false, false, false};
  DeviceArray<P3> ps_lr, ps_le, ps_alb, ps_nrm;
  DeviceArray<hala_ray> q_rays[2];
  DeviceArray<float4> q_state[2];
  DeviceArray<hala_hit> q_hits;
  DeviceArray<uint32_t> q_perm;
  DeviceArray<ShadowEntry> q_shadow[2];
  DeviceArray<Control> d_ctl;
  DeviceArray<WorkCounters> d_batch_work;

  uint64_t total_frames = 0;
  bool counting = false;
  hala_global_uniform last_uniform{};
  TraceEvents ring[kStatRing];
  int ring_pos = 0;
  bool vertices_dirty = false;  // hala_rt_update_vertices since the last refit
  bool materials_dirty_any = false;  // a material edit touched an opacity-0 material (old or new)
  bool materials_dirty_any_refit = false;  // ... as hala_rt_refit found it
  bool any_invisible = false;   // the scene has invisible or translucent materials: the any-hit launches traverse d_tris_any (RENDER_SPEC 7.1d)
  bool any_translucent = false; // ... translucent ones: the ALPHA variants of the any-hit kernels
  DeviceArray<uint8_t> d_material_any_class;
  std::vector<uint8_t> material_any_class;
  uint32_t launch_event_period = 0;  // per-launch timing events on every n-th update (hala_rt_set_launch_timing_period; 0: none)
  unsigned long long update_counter = 0;
  hala_rt_statistics stats{};
  // update() and trace_rays() share per-renderer scratch (work counters, step counters, the stack spill area): launches that use it
  // are ordered across streams by an event — the last user records one, a user on another stream waits for it first
  hipEvent_t scratch_event = nullptr;       // not owned: a ring slot's frame_end or batch_done
  hipStream_t scratch_stream = nullptr;
  hipEvent_t batch_done = nullptr;
  // multi-GPU exchange (C1 of SURVEY 2.1): one RCCL all-gather of the rank's tile buffer per AOV and frame, inside the library
  ncclComm_t comm = nullptr;
  bool comm_owned = false;
  int comm_rank = 0, comm_world = 1;
  hipStream_t gather_stream = nullptr;
  DeviceArray<float4> gather_stage[4], gather_recv[4];
  hipEvent_t ev_rendered = nullptr, ev_staged = nullptr, ev_gathered = nullptr;
  uint32_t gather_pending = 0;  // AOV mask of the collective in flight (hala_rt_tile_allgather_begin)
  int scratch_acquire(hipStream_t s) {
    if (scratch_event && scratch_stream != s) RT_HIP(hipStreamWaitEvent(s, scratch_event, 0));
    return HALA_OK;
  }

  ~hala_rt_renderer() {
    if (device >= 0) (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& t : ring) {
      for (auto e : t.ev) (void)hipEventDestroy(e);
      if (t.frame_begin) (void)hipEventDestroy(t.frame_begin);
      if (t.frame_end) (void)hipEventDestroy(t.frame_end);
      if (t.host_counts) (void)hipHostFree(t.host_counts);
      if (t.host_queue_sizes) (void)hipHostFree(t.host_queue_sizes);
    }
    if (batch_done) (void)hipEventDestroy(batch_done);
    if (gather_stream) { (void)hipStreamSynchronize(gather_stream); (void)hipStreamDestroy(gather_stream); }
    for (hipEvent_t e : {ev_rendered, ev_staged, ev_gathered}) if (e) (void)hipEventDestroy(e);
    if (comm && comm_owned) { if (const RcclApi* api = rccl_api(nullptr)) (void)api->CommDestroy(comm); }
    // images first, then everything else (src/rt_renderer.rs:620-633)
    for (auto& i : img_local) i.release();
    for (auto& i : img_full) i.release();
    if (bvh.topology) bvh_free_topology(bvh.topology);
    blas.clear();
    if (stream) (void)hipStreamDestroy(stream);
  }

  SceneView view() const {
    SceneView sv{};
    sv.nodes = d_nodes.ptr; sv.tris = d_tris.ptr; sv.tris_any = any_invisible ? d_tris_any.ptr : d_tris.ptr; sv.tris_by_id = d_tris_by_id.ptr; sv.shade_tris = d_shade_tris.ptr;
    sv.inst_first_tri = d_inst_first_tri.ptr; sv.primitives = d_instances.ptr; sv.materials = d_materials.ptr; sv.material_kind = d_material_kind.ptr;
    sv.lights = d_lights.ptr; sv.cameras = d_cameras.ptr;
    sv.textures = d_textures.ptr; sv.tex_arena = d_tex_arena.ptr; sv.tex_arena8 = d_tex_arena8.ptr; sv.tex_lut = d_srgb_lut.ptr; sv.texture_count = (uint32_t)host_textures.size(); sv.shade_sort = shade_sort ? 1u : 0u; sv.simple_materials = simple_materials ? 1u : 0u; sv.scatter_media = scatter_media ? 1u : 0u; sv.any_translucent = any_translucent ? 1u : 0u;
    sv.env_pixels = reinterpret_cast<const float*>(d_env.ptr); sv.env_marginal = d_marginal.ptr; sv.env_conditional = d_conditional.ptr;
    sv.node_count = bvh.node_count; sv.tri_count = bvh.tri_count; sv.lds_nodes = lds_nodes; sv.lds_tris = lds_tris;
    sv.inst_refs = d_inst_refs.ptr; sv.inst_info = d_inst_info.ptr; sv.instance_count = (uint32_t)hs.instances.size(); sv.two_level = two_level ? 1u : 0u;
    sv.ray_eps = ray_eps;
    sv.staged = staged ? 1u : 0u;
    return sv;
  }
  Queues queues() const {
    Queues q{};
    q.rays[0] = q_rays[0].ptr; q.rays[1] = q_rays[1].ptr; q.state[0] = q_state[0].ptr; q.state[1] = q_state[1].ptr;
    q.hits = q_hits.ptr; q.perm = q_perm.ptr; q.shadow[0] = q_shadow[0].ptr; q.shadow[1] = q_shadow[1].ptr;
    return q;
  }
  PathState path_state() const { return PathState{ps_lr.ptr, ps_le.ptr, ps_alb.ptr, ps_nrm.ptr}; }

  FrameConst frame_const(const hala_global_uniform& u, uint32_t samples = 1) const {
    FrameConst fc{};
    fc.u = u;
    fc.aspect = u.resolution[0] / u.resolution[1];
    float sn = 0.0f, cs = 1.0f;
    if (!hs.cameras.empty()) h_sincos_rad(0.5f * hs.cameras[0].yfov, &sn, &cs);
    fc.tan_half = sn / cs;
    fc.pixel_spread = 2.0f * fc.tan_half / u.resolution[1];
    fc.width = width; fc.height = height;
    fc.tile_size = tile_size; fc.tiles_x = tiles_x; fc.tiles_y = tiles_y; fc.world = world; fc.rank = rank; fc.blocks_x = blocks_x;
    fc.tiles_per_rank = tiles_per_rank; fc.perm_a = perm_a_inv; fc.perm_b = perm_b;
    fc.pixel_slots = slot_count; fc.samples = samples; fc.slot_count = slot_count * samples;
    return fc;
  }

  void reset_accumulation() {  // statistics.reset() of the device-lost path (src/rt_renderer.rs:557)
    total_frames = 0;
    for (bool& v : full_valid) v = false;
  }

  // resolve one ring slot's events into the totals (the slot's work must have completed)
  void resolve_slot(TraceEvents& t) {
    if (!t.pending) return;
    (void)hipEventSynchronize(t.frame_end);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, t.frame_begin, t.frame_end) == hipSuccess) { stats.last_gpu_ms = ms; stats.gpu_ms_total += ms; }
    // four events per depth: a | closest-hit launch | b | shade launch | c | shadow launch(es) or the fused launch | d
    double tr[3] = {0.0, 0.0, 0.0}, sh = 0.0;  // closest-hit launches, shadow launches, fused launches
    unsigned long long rays[3] = {0, 0, 0}, launches[3] = {0, 0, 0};
    const uint32_t* qn = t.host_queue_sizes;  // Control::n_active[kMaxDepth + 1] | n_shadow[2][kMaxDepth] of this pass (timed passes only)
    for (size_t k = 0, depth = 0; k + 3 < t.used; k += 4, ++depth) {
      float m = 0.0f;
      const bool own_closest = depth == 0 || !((t.traced_mask >> depth) & 1ull);  // else: it ran inside the previous depth's fused launch
      if (own_closest && hipEventElapsedTime(&m, t.ev[k], t.ev[k + 1]) == hipSuccess) {
        const unsigned long long n = depth == 0 ? (unsigned long long)real_pixels * t.samples : qn[depth];
        tr[0] += m; rays[0] += n; launches[0] += 1;
        if (depth == 0) { stats.traverse_primary_ms_total += m; stats.traverse_primary_launches += 1; stats.rays_primary_timed += n; }  // k_trace_primary
      }
      if (hipEventElapsedTime(&m, t.ev[k + 1], t.ev[k + 2]) == hipSuccess) sh += m;
      if (hipEventElapsedTime(&m, t.ev[k + 2], t.ev[k + 3]) == hipSuccess) {
        const unsigned long long ns = (unsigned long long)qn[kMaxDepth + 1 + depth] + qn[kMaxDepth + 1 + kMaxDepth + depth];
        if ((t.fused_mask >> depth) & 1ull) {
          tr[2] += m; launches[2] += 1;
          stats.rays_fused_shadow_timed += ns;
          if ((t.traced_mask >> (depth + 1)) & 1ull) stats.rays_fused_closest_timed += qn[depth + 1];
        } else { tr[1] += m; rays[1] += ns; }
      }
    }
    stats.traverse_ms_last_update = tr[0] + tr[1] + tr[2];
    stats.traverse_closest_ms_total += tr[0];
    stats.traverse_shadow_ms_total += tr[1];
    stats.traverse_fused_ms_total += tr[2];
    stats.shade_ms_total += sh;
    stats.traverse_closest_launches += launches[0];
    stats.traverse_fused_launches += launches[2];
    stats.shade_launches += t.used / 4;
    stats.traverse_shadow_launches += t.used ? t.shadow_launches : 0;  // as issued: one per connection kind the scene has, per depth
    stats.rays_closest_timed += rays[0]; stats.rays_shadow_timed += rays[1];
    stats.updates_rendered += t.samples;
    const unsigned long long rc = t.host_counts[0], rs = t.host_counts[1];
    stats.rays_last_update = rc + rs;
    stats.rays_total += rc + rs;
    stats.rays_closest_total += rc;
    stats.rays_primary_total += (unsigned long long)real_pixels * t.samples;
    stats.rays_shadow_total += rs;
    if (t.counted) {
      stats.nodes_closest_total += t.host_counts[2]; stats.tris_closest_total += t.host_counts[3];
      stats.nodes_shadow_total += t.host_counts[4]; stats.tris_shadow_total += t.host_counts[5];
      stats.rays_closest_counted += rc; stats.rays_shadow_counted += rs;
      stats.wave_steps_closest_total += t.host_counts[6]; stats.leaf_passes_closest_total += t.host_counts[7]; stats.leaf_lanes_closest_total += t.host_counts[8];
      stats.nodes_primary_total += t.host_counts[12]; stats.tris_primary_total += t.host_counts[13];
      stats.rays_primary_counted += (unsigned long long)real_pixels * t.samples;
      stats.wave_steps_shadow_total += t.host_counts[9]; stats.leaf_passes_shadow_total += t.host_counts[10]; stats.leaf_lanes_shadow_total += t.host_counts[11];
    }
    t.pending = false;
  }
  hipEvent_t next_event(TraceEvents& t) {
    if (t.used == t.ev.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); t.ev.push_back(e); }
    return t.ev[t.used++];
  }
};

namespace {

int ensure_device(hala_rt_renderer* r) {
  if (!r) RT_FAIL("The renderer handle is null!");
  RT_HIP(hipSetDevice(r->device));
  return HALA_OK;
}

void compute_tiling(hala_rt_renderer* r) {
  if (r->world <= 1) {
    r->real_pixels = r->width * r->height;
    r->slot_count = r->real_pixels;
    r->blocks_x = 0;
    if (kPixelBlock) {  // whole blocks: the border blocks of a frame that is not a multiple of the block size hold padding slots
      r->blocks_x = (r->width + kPixelBlock - 1) / kPixelBlock;
      r->slot_count = r->blocks_x * ((r->height + kPixelBlock - 1) / kPixelBlock) * kPixelBlock * kPixelBlock;
    }
    r->tiles_x = r->tiles_y = r->tiles_per_rank = 0;
    return;
  }
  r->tiles_x = (r->width + r->tile_size - 1) / r->tile_size;
  r->tiles_y = (r->height + r->tile_size - 1) / r->tile_size;
  const uint32_t n = r->tiles_x * r->tiles_y;
  r->tiles_per_rank = (n + r->world - 1) / r->world;
  uint32_t A = 0x9E3779B1u % n;  // RENDER_SPEC §9: perm(t) = (t*A + B) mod n, A coprime to n
  if (A == 0) A = 1;
  while (gcd_u32(A, n) != 1) ++A;
  r->perm_a_inv = mod_inverse(A, n);
  r->perm_b = 7;
  r->slot_count = r->tiles_per_rank * r->tile_size * r->tile_size;
  // pixels this rank really owns (its padding tiles and the out-of-frame part of border tiles hold no paths)
  uint64_t real = 0;
  for (uint32_t t = 0; t < n; ++t) {
    const uint32_t k = (uint32_t)(((uint64_t)t * A + r->perm_b) % n);
    if (k % r->world != r->rank) continue;
    const uint32_t tx = t % r->tiles_x, ty = t / r->tiles_x;
    const uint32_t w = std::min(r->tile_size, r->width - tx * r->tile_size), h = std::min(r->tile_size, r->height - ty * r->tile_size);
    real += (uint64_t)w * h;
  }
  r->real_pixels = (uint32_t)real;
}

// wavefront state for `samples` frames in flight (hala_rt_update_batch): everything indexed by path slot
int alloc_wavefront(hala_rt_renderer* r, uint32_t samples) {
  const size_t n = (size_t)r->slot_count * samples;
  if (n > 0xfffffff0ull) RT_FAIL("The sample batch is too large for 32-bit path slots.");
  RT_HIP(r->ps_lr.resize(n)); RT_HIP(r->ps_le.resize(n)); RT_HIP(r->ps_alb.resize(n)); RT_HIP(r->ps_nrm.resize(n));
  RT_HIP(r->q_rays[0].resize(n)); RT_HIP(r->q_rays[1].resize(n)); RT_HIP(r->q_state[0].resize(n)); RT_HIP(r->q_state[1].resize(n));
  RT_HIP(r->q_hits.resize(n)); RT_HIP(r->q_perm.resize(n)); RT_HIP(r->q_shadow[0].resize(n)); RT_HIP(r->q_shadow[1].resize(n));
  r->batch_capacity = samples;
  return HALA_OK;
}

int alloc_frame_buffers(hala_rt_renderer* r) {
  const size_t n = r->slot_count;
  for (auto& i : r->img_local) { RT_HIP(i.resize(n)); RT_HIP(hipMemsetAsync(i.ptr, 0, n * sizeof(float4), r->stream)); }
  if (alloc_wavefront(r, 1) != HALA_OK) return HALA_ERR;
  RT_HIP(r->d_ctl.resize(1));
  RT_HIP(hipMemsetAsync(r->d_ctl.ptr, 0, sizeof(Control), r->stream));
  RT_HIP(r->d_batch_work.resize(1));
  return HALA_OK;
}

// geometry = false re-publishes only the small records (cameras, lights, materials, instances): what a refit needs, since
// node transforms move instances, cameras and lights but leave the vertex / index arenas untouched.
int upload_packed(hala_rt_renderer* r, bool geometry = true) {
  HostScene& hs = r->hs;
  if (geometry) {
  // one arena each for all vertex / index buffers (the reference creates one buffer pair per primitive,
  // gpu_uploader.rs:421-456; device addresses per primitive are what matters to the shaders, :869-870)
  size_t nv = 0, ni = 0;
  r->prim_vertex_offset.clear(); r->prim_index_offset.clear();
  for (const auto& p : hs.prims) {
    r->prim_vertex_offset.push_back(nv); r->prim_index_offset.push_back(ni);
    nv += p.vertices.size();
    ni += (p.indices.size() + 3) & ~size_t(3);  // keep every index buffer 16-B aligned
  }
  RT_HIP(r->d_vertices.resize(nv)); RT_HIP(r->d_indices.resize(ni));
  for (size_t k = 0; k < hs.prims.size(); ++k) {
    const auto& p = hs.prims[k];
    if (!p.vertices.empty()) RT_HIP(hipMemcpyAsync(r->d_vertices.ptr + r->prim_vertex_offset[k], p.vertices.data(), p.vertices.size() * sizeof(hala_vertex), hipMemcpyHostToDevice, r->stream));
    if (!p.indices.empty()) RT_HIP(hipMemcpyAsync(r->d_indices.ptr + r->prim_index_offset[k], p.indices.data(), p.indices.size() * 4, hipMemcpyHostToDevice, r->stream));
  }
  }
  for (size_t i = 0; i < hs.instances.size(); ++i) {
    const uint32_t p = hs.instance_prim[i];
    hs.instances[i].vertices = reinterpret_cast<uint64_t>(r->d_vertices.ptr + r->prim_vertex_offset[p]);  // get_device_address (:869)
    hs.instances[i].indices = reinterpret_cast<uint64_t>(r->d_indices.ptr + r->prim_index_offset[p]);     // (:870)
  }
  RT_HIP(r->d_cameras.upload(hs.cameras.data(), hs.cameras.size(), r->stream));
  RT_HIP(r->d_lights.upload(hs.lights.data(), hs.lights.size(), r->stream));
  RT_HIP(r->d_materials.upload(hs.gpu_materials.data(), hs.gpu_materials.size(), r->stream));
  {
    std::vector<uint8_t> kind(hs.gpu_materials.size());
    for (size_t i = 0; i < kind.size(); ++i) kind[i] = shade_kind_of(hs.gpu_materials[i], (uint32_t)hs.texture_image.size());
    RT_HIP(r->d_material_kind.upload(kind.data(), kind.size(), r->stream));
    r->material_kind = kind;
    uint32_t seen = 0;
    for (uint8_t k : kind) seen |= 1u << k;
    r->shade_sort = (seen & (seen - 1u)) != 0u;  // two kinds or more (a one-kind scene like the Cornell box only pays for the sort)
    r->scatter_media = false;
    for (const auto& m : hs.gpu_materials) r->scatter_media = r->scatter_media || m.medium_type == 2u;
    r->simple_materials = seen == (1u << kShadeKindFirst);  // nothing but untextured opaque DIFFUSE: the SIMPLE shade kernels (configs[1])
    if (tune_env("HALART_NO_SIMPLE_SHADE")) r->simple_materials = false;
  }
  RT_HIP(r->d_instances.upload(hs.instances.data(), hs.instances.size(), r->stream));
  RT_HIP(r->d_inst_first_tri.upload(hs.inst_first_tri.data(), hs.inst_first_tri.size(), r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));
  return HALA_OK;
}

// textures: upload level 0 of every image, build the mip chains on the GPU (gen_mipmaps, gpu_uploader.rs:400), publish
// one TexDesc per texture.  mip count = ceil(log2(max(w,h))) + 1 (gpu_uploader.rs:366), capped at kMaxMips.
int upload_textures(hala_rt_renderer* r) {
  const HostScene& hs = r->hs;
  std::vector<TexDesc> img_desc(hs.images.size());
  size_t total_f = 0, total_8 = 0, largest_8 = 0;  // float4 texels / tiled 4-B texels / largest level 0 among the 8-bit images
  for (size_t k = 0; k < hs.images.size(); ++k) {
    TexDesc& td = img_desc[k];
    memset(&td, 0, sizeof(td));
    td.width = hs.images[k].width; td.height = hs.images[k].height; td.format = hs.images[k].format;
    uint32_t m = std::max(td.width, td.height), p2 = 1, lg = 0;
    while (p2 < m) { p2 <<= 1; ++lg; }
    td.mips = std::min<uint32_t>(lg + 1, kMaxMips);
    size_t& total = td.format == kTexFloat ? total_f : total_8;
    for (uint32_t l = 0; l < td.mips; ++l) {
      if (total > 0xffffffffull) RT_FAIL("The texture arena exceeds 2^32 texels.");
      td.mip_offset[l] = (uint32_t)total;
      const uint32_t lw = std::max(1u, td.width >> l), lh = std::max(1u, td.height >> l);
      total += td.format == kTexFloat ? (size_t)lw * lh : (size_t)tex_tiled_size(lw, lh);
    }
    if (td.format != kTexFloat) largest_8 = std::max(largest_8, (size_t)td.width * td.height);
  }
  RT_HIP(r->d_tex_arena.resize(total_f));
  RT_HIP(r->d_tex_arena8.resize(total_8));
  if (total_8) RT_HIP(hipMemsetAsync(r->d_tex_arena8.ptr, 0, total_8 * 4, r->stream));  // the padding texels of partial tiles
  // the sRGB decode table and the midpoints between its entries (the encoder of the 8-bit mip chain bisects them)
  const float* lut = srgb_decode_lut();
  float thr[256];
  for (int k = 0; k < 255; ++k) thr[k] = (lut[k] + lut[k + 1]) * 0.5f;
  thr[255] = 3.402823466e+38f;
  float lut512[512];  // shading.h::tex8_fetch: the sRGB EOTF, then b / 255
  for (int k = 0; k < 256; ++k) { lut512[k] = lut[k]; lut512[256 + k] = (float)k / 255.0f; }
  RT_HIP(r->d_srgb_lut.upload(lut512, 512, r->stream));
  RT_HIP(r->d_srgb_thr.upload(thr, 256, r->stream));
  DeviceArray<uint32_t> staging;  // row-major level 0 of one 8-bit image at a time
  RT_HIP(staging.resize(largest_8));
  for (size_t k = 0; k < hs.images.size(); ++k) {
    const TexDesc& td = img_desc[k];
    if (td.format == kTexFloat) {
      RT_HIP(hipMemcpyAsync(r->d_tex_arena.ptr + td.mip_offset[0], hs.images[k].rgba.data(), (size_t)td.width * td.height * 16, hipMemcpyHostToDevice, r->stream));
      for (uint32_t l = 1; l < td.mips; ++l)
        launch_mip_downsample(r->d_tex_arena.ptr + td.mip_offset[l - 1], std::max(1u, td.width >> (l - 1)), std::max(1u, td.height >> (l - 1)),
                              r->d_tex_arena.ptr + td.mip_offset[l], std::max(1u, td.width >> l), std::max(1u, td.height >> l), r->stream);
    } else {
      RT_HIP(hipMemcpyAsync(staging.ptr, hs.images[k].rgba8.data(), (size_t)td.width * td.height * 4, hipMemcpyHostToDevice, r->stream));
      launch_tile8(staging.ptr, td.width, td.height, r->d_tex_arena8.ptr + td.mip_offset[0], r->stream);
      for (uint32_t l = 1; l < td.mips; ++l)
        launch_mip_downsample8(r->d_tex_arena8.ptr + td.mip_offset[l - 1], std::max(1u, td.width >> (l - 1)), std::max(1u, td.height >> (l - 1)),
                               r->d_tex_arena8.ptr + td.mip_offset[l], std::max(1u, td.width >> l), std::max(1u, td.height >> l), td.format,
                               r->d_srgb_lut.ptr, r->d_srgb_thr.ptr, r->stream);
    }
  }
  std::vector<TexDesc> tex(hs.texture_image.size());
  for (size_t i = 0; i < tex.size(); ++i) tex[i] = img_desc[hs.texture_image[i]];
  RT_HIP(r->d_textures.upload(tex.data(), tex.size(), r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));  // (also: `staging` and the host images may go)
  RT_HIP(hipGetLastError());
  r->host_textures = tex;
  return HALA_OK;
}

int configure_traversal(hala_rt_renderer* r) {
  const size_t nb = (size_t)r->bvh.node_count * 64, tb = (size_t)r->bvh.tri_count * 48;
  // Whole BVH in LDS when it fits the budget (the STAGED kernel variants read it with ds_read only); otherwise nothing
  // is staged: a top-of-tree slice measured no gain (profiles/r01_h_experiments.txt), the caches already hold it.
  size_t budget = kLdsStageBudget;
  if (const char* e = tune_env("HALART_LDS_STAGE_BYTES")) budget = (size_t)strtoul(e, nullptr, 10);  // tuning knob
  r->staged = !r->two_level && nb + tb <= budget;  // (the LDS-staged kernel variants know no instances)
  r->lds_nodes = r->staged ? r->bvh.node_count : 0u;
  r->lds_tris = r->staged ? r->bvh.tri_count : 0u;
  if (r->leaf_max_built > traverse_max_leaf(r->staged)) RT_FAIL("The BVH was built with larger leaves than the traversal variant for its size accepts.");
  const int tree = r->staged ? 1 : (r->two_level ? 2 : 0);
  const size_t smem = (size_t)r->lds_nodes * 64 + (size_t)r->lds_tris * 48 + traverse_fixed_lds_bytes(tree);
  uint32_t per_cu = traverse_blocks_per_cu(smem, tree);
  if (per_cu == 0) RT_FAIL("The traversal kernel does not fit on a compute unit with the requested LDS staging.");
  per_cu = std::min(per_cu, 8u);
  if (const char* e = tune_env("HALART_BLOCKS_PER_CU")) per_cu = std::min(per_cu, std::max(1u, (uint32_t)atoi(e)));  // tuning knob
  r->lcfg.persistent_blocks = r->cu_count * per_cu;
  r->lcfg.spill = nullptr;
  // measured (profiles/r01_c_refill_sweep.txt): whole-wave refills are best when the BVH lives in LDS (uniform, cheap rays);
  // refilling once half the wave is idle is best when node fetches go to L2 / Infinity Cache
  r->lcfg.refill = r->staged ? 64u : kRefillThreshold;
  if (const char* e = tune_env("HALART_REFILL")) r->lcfg.refill = std::min(64u, std::max(1u, (uint32_t)strtoul(e, nullptr, 10)));  // tuning knob
  if (r->two_level || r->bvh.stack_need > traverse_stack_lds_levels(tree)) {
    if (r->two_level || r->bvh.stack_need > traverse_stack_lds_levels(tree) + traverse_stack_spill_levels()) {
      // 3 x levels is a loose bound (every node on the path deferring three siblings).  Before refusing the tree, take the exact
      // one: need(node) = (inner children - 1) + max need(inner child) — the worst order visits the child with the deepest
      // need first while all its siblings wait.  Nodes are in breadth-first order (children behind their parent): one reverse sweep.
      std::vector<BvhNode4> nodes(r->bvh.node_count);
      RT_HIP(hipMemcpy(nodes.data(), r->d_nodes.ptr, nodes.size() * sizeof(BvhNode4), hipMemcpyDeviceToHost));
      std::vector<uint32_t> need(nodes.size(), 0u);
      for (size_t i = nodes.size(); i-- > 0;) {
        uint32_t inner = 0, deepest = 0;
        for (uint32_t ref : nodes[i].ref) {
          if (ref == kAbsent) continue;
          if (is_inst_leaf(ref)) {  // RENDER_SPEC 4.5: the world-space ray (3 entries) and the exit mark wait below the instance's own entries
            const uint32_t k = ref & 0x0fffffffu;
            ++inner;
            if (k < r->inst_refs.size() && r->inst_refs[k].root < need.size()) deepest = std::max(deepest, 4u + need[r->inst_refs[k].root]);
            continue;
          }
          if (ref & kLeafRef) continue;
          ++inner;
          if (ref < need.size()) deepest = std::max(deepest, need[ref]);
        }
        need[i] = inner ? inner - 1u + deepest : 0u;
      }
      r->bvh.stack_need = need.empty() ? 1u : std::max(1u, need[0]);
    }
    if (r->bvh.stack_need > traverse_stack_lds_levels(tree) + traverse_stack_spill_levels())
      RT_FAIL("The BVH is deeper than the traversal stack supports (" + std::to_string(r->bvh.max_depth) + " levels, " + std::to_string(r->bvh.stack_need) + " stack entries).");
    RT_HIP(r->d_spill.resize((size_t)r->lcfg.persistent_blocks * 256 * traverse_stack_spill_levels()));
    r->lcfg.spill = r->d_spill.ptr;
  }
  const float ex = r->bvh.scene_max[0] - r->bvh.scene_min[0], ey = r->bvh.scene_max[1] - r->bvh.scene_min[1], ez = r->bvh.scene_max[2] - r->bvh.scene_min[2];
  r->ray_eps = std::sqrt(std::fmaf(ez, ez, std::fmaf(ey, ey, ex * ex))) * 1e-5f;  // RENDER_SPEC §3
  return HALA_OK;
}

// RENDER_SPEC 7.1d: scenes with opacity-0 materials get a second copy of the BVH-order triangles for the any-hit launches
int attach_any_triangles(hala_rt_renderer* r) {
  const HostScene& hs = r->hs;
  std::vector<uint8_t> cls(hs.gpu_materials.size());
  r->any_invisible = false; r->any_translucent = false;
  for (size_t i = 0; i < cls.size(); ++i) {
    const hala_gpu_material& m = hs.gpu_materials[i];
    const bool cutout = m.base_color_map_index < hs.texture_image.size() && hs.images[hs.texture_image[m.base_color_map_index]].has_alpha;
    cls[i] = any_class_of(m, cutout);
    r->any_invisible = r->any_invisible || cls[i] != 0;
    r->any_translucent = r->any_translucent || cls[i] >= 2;
  }
  r->material_any_class = cls;
  RT_HIP(r->d_material_any_class.upload(cls.data(), cls.size(), r->stream));
  if (r->any_invisible) RT_HIP(r->d_tris_any.resize(r->two_level ? r->stored_tris : r->hs.triangle_count));
  r->bvh.tris_any = r->any_invisible ? r->d_tris_any.ptr : nullptr;
  r->bvh.material_any_class = r->d_material_any_class.ptr;
  r->bvh.material_kind = r->d_material_kind.ptr;
  r->bvh.material_count = (uint32_t)r->hs.gpu_materials.size();
  return HALA_OK;
}

// ---- two-level trees (RENDER_SPEC 4.5) ---------------------------------------------------------------------------------------------
// world -> object of one instance: rows of the inverse of the upper 3x3 (cross products of its columns over the determinant) and the
// translation; false: not invertible in float (the instance is flattened to world space like a primitive that is referenced once)
static float h_dot3(const float* a, const float* b) { return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); }
static void h_cross3(const float* a, const float* b, float* o) {
  o[0] = std::fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = std::fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = std::fmaf(a[0], b[1], -(a[1] * b[0]));
}
static bool world_to_object(const float* m, InstRef* o) {
  const float c0[3] = {m[0], m[1], m[2]}, c1[3] = {m[4], m[5], m[6]}, c2[3] = {m[8], m[9], m[10]};
  float k0[3], k1[3], k2[3];
  h_cross3(c1, c2, k0); h_cross3(c2, c0, k1); h_cross3(c0, c1, k2);
  const float det = h_dot3(c0, k0);
  if (!(det != 0.0f) || !std::isfinite(det)) return false;
  const float inv = 1.0f / det;
  for (int k = 0; k < 3; ++k) { o->r0[k] = k0[k] * inv; o->r1[k] = k1[k] * inv; o->r2[k] = k2[k] * inv; o->tr[k] = m[12 + k]; }
  for (int k = 0; k < 3; ++k)
    if (!std::isfinite(o->r0[k]) || !std::isfinite(o->r1[k]) || !std::isfinite(o->r2[k]) || !std::isfinite(o->tr[k])) return false;
  return true;
}
static void h_transform_point(const float* m, const float* p, float* o) {  // RENDER_SPEC 3
  for (int k = 0; k < 3; ++k) o[k] = std::fmaf(m[8 + k], p[2], std::fmaf(m[4 + k], p[1], m[k] * p[0])) + m[12 + k];
}
// which instances are intersected in object space: those of a primitive that several instances reference, if their transform can be inverted
static void classify_instances(hala_rt_renderer* r, std::vector<uint8_t>* flags) {
  const HostScene& hs = r->hs;
  flags->assign(hs.instances.size(), 0);
  // automatic: the flattened tree is the faster one (no moves into object space, no instance levels to walk) while it fits comfortably
  constexpr uint32_t kFlattenLimit = 1u << 26;  // triangles: ~15 GB of nodes, triangles and shading records
  if (r->instancing_mode == 1u || (r->instancing_mode == 0u && hs.triangle_count <= kFlattenLimit)) return;
  std::vector<uint32_t> refs(hs.prims.size(), 0u);
  for (uint32_t p : hs.instance_prim) refs[p]++;
  for (size_t i = 0; i < hs.instances.size(); ++i) {
    InstRef tmp;
    (*flags)[i] = refs[hs.instance_prim[i]] >= 2u && hs.prims[hs.instance_prim[i]].indices.size() >= 3 && world_to_object(hs.instances[i].transform, &tmp) ? 1 : 0;
  }
}

// the instance levels: InstRef per instanced instance, one item per instanced instance + one for the world tree, the host build, the upload;
// also the scene bounds (RENDER_SPEC 4.5: world tree's exact bounds + the boxes of the transformed corners of the instanced primitives' bounds)
int build_instance_levels(hala_rt_renderer* r) {
  const HostScene& hs = r->hs;
  std::vector<TlasItem> items;
  r->inst_refs.clear();
  std::vector<InstInfo> info(hs.instances.size());
  float smin[3] = {INFINITY, INFINITY, INFINITY}, smax[3] = {-INFINITY, -INFINITY, -INFINITY};
  uint32_t deepest = 0;
  // shading records: the world tree's triangles in its own order (instance order), then every instanced primitive's
  std::vector<uint32_t> flat_base(hs.instances.size(), 0u);
  if (!r->blas.empty() && !r->blas[0]->object_space) {
    const hala_rt_renderer::Blas& w = *r->blas[0];
    uint32_t at = w.tri_off;
    for (uint32_t i : w.insts) { flat_base[i] = at; at += hs.inst_first_tri[i + 1] - hs.inst_first_tri[i]; }
    if (w.b.tri_count) {
      TlasItem it{};
      const float pad = std::max({std::fabs(w.b.scene_min[0]), std::fabs(w.b.scene_min[1]), std::fabs(w.b.scene_min[2]), std::fabs(w.b.scene_max[0]),
                                  std::fabs(w.b.scene_max[1]), std::fabs(w.b.scene_max[2])}) * 1.9073486328125e-06f * 2.0f;
      for (int k = 0; k < 3; ++k) { it.mn[k] = w.b.scene_min[k] - pad; it.mx[k] = w.b.scene_max[k] + pad; smin[k] = std::min(smin[k], w.b.scene_min[k]); smax[k] = std::max(smax[k], w.b.scene_max[k]); }
      it.ref = w.node_off;  // its root: an inner child, no transform
      it.need = w.b.stack_need;
      deepest = std::max(deepest, w.b.max_depth);
      items.push_back(it);
    }
  }
  for (size_t i = 0; i < hs.instances.size(); ++i) {
    info[i].first_tri = hs.inst_first_tri[i];
    info[i].instanced = r->inst_instanced[i];
    info[i].pad = 0;
    if (!r->inst_instanced[i]) { info[i].shade_base = flat_base[i]; continue; }
    const hala_rt_renderer::Blas& bl = *r->blas[(size_t)r->prim_blas[hs.instance_prim[i]]];
    info[i].shade_base = bl.tri_off;
    InstRef ref{};
    if (!world_to_object(hs.instances[i].transform, &ref)) RT_FAIL("An instanced node's transform stopped being invertible: commit() again.");
    ref.root = bl.node_off; ref.gid_base = hs.inst_first_tri[i]; ref.shade_base = bl.tri_off; ref.inst = (uint32_t)i;
    // world box of the instance: the eight corners of its primitive's exact object-space bounds, moved to world space
    TlasItem it{};
    float wmn[3] = {INFINITY, INFINITY, INFINITY}, wmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 8; ++c) {
      const float p[3] = {(c & 1) ? bl.b.scene_max[0] : bl.b.scene_min[0], (c & 2) ? bl.b.scene_max[1] : bl.b.scene_min[1], (c & 4) ? bl.b.scene_max[2] : bl.b.scene_min[2]};
      float q[3];
      h_transform_point(hs.instances[i].transform, p, q);
      for (int k = 0; k < 3; ++k) { wmn[k] = std::min(wmn[k], q[k]); wmx[k] = std::max(wmx[k], q[k]); }
    }
    // padded like every box (RENDER_SPEC 4.1b), twice: once for the rounding of the move to world space, once for the object-space pad of the leaves below
    const float amax = std::max({std::fabs(wmn[0]), std::fabs(wmn[1]), std::fabs(wmn[2]), std::fabs(wmx[0]), std::fabs(wmx[1]), std::fabs(wmx[2])});
    const float ext = std::max({wmx[0] - wmn[0], wmx[1] - wmn[1], wmx[2] - wmn[2]});
    const float pad = (amax + ext) * 1.9073486328125e-06f * 2.0f;
    for (int k = 0; k < 3; ++k) { it.mn[k] = wmn[k] - pad; it.mx[k] = wmx[k] + pad; smin[k] = std::min(smin[k], wmn[k]); smax[k] = std::max(smax[k], wmx[k]); }
    it.ref = kInstLeafTag | (uint32_t)r->inst_refs.size();
    it.need = 4u + bl.b.stack_need;  // the world-space ray (3 entries) and the exit mark wait below the instance's own entries
    deepest = std::max(deepest, bl.b.max_depth);
    if (r->inst_refs.size() >= 0x0ffffff0u) RT_FAIL("Too many instances.");
    r->inst_refs.push_back(ref);
    items.push_back(it);
  }
  if (items.size() > r->tlas_capacity) RT_FAIL("internal: instance levels larger than reserved");
  std::vector<BvhNode4> nodes;
  uint32_t levels = 0, need = 0;
  r->tlas_nodes = tlas_build(items, nodes, &levels, &need);
  RT_HIP(hipMemcpyAsync(r->d_nodes.ptr, nodes.data(), nodes.size() * sizeof(BvhNode4), hipMemcpyHostToDevice, r->stream));
  RT_HIP(r->d_inst_refs.upload(r->inst_refs.data(), r->inst_refs.size(), r->stream));
  RT_HIP(r->d_inst_info.upload(info.data(), info.size(), r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));  // `nodes`, `info` go out of scope
  r->bvh.max_depth = levels + deepest;
  r->bvh.stack_need = need;
  for (int k = 0; k < 3; ++k) { r->bvh.scene_min[k] = items.empty() ? 0.0f : smin[k]; r->bvh.scene_max[k] = items.empty() ? 0.0f : smax[k]; }
  return HALA_OK;
}

// builds / refits one tree of a two-level scene into its sub-ranges and makes its references absolute
static int blas_build_or_refit(hala_rt_renderer* r, hala_rt_renderer::Blas& bl, bool refit) {
  const HostScene& hs = r->hs;
  std::vector<hala_gpu_mesh_data> md;
  std::vector<uint32_t> first{0u}, gid, inst;
  if (bl.object_space) {
    hala_gpu_mesh_data m{};
    uint32_t any = 0;
    while (hs.instance_prim[any] != bl.prim) ++any;  // any instance of the primitive: material and buffer addresses are the primitive's
    m = hs.instances[any];
    const Mat4 id = Mat4::identity();
    memcpy(m.transform, id.m, 64);
    md.push_back(m); gid.push_back(0u); inst.push_back(kAbsent);
    first.push_back((uint32_t)(hs.prims[bl.prim].indices.size() / 3));
  } else {
    for (uint32_t i : bl.insts) {
      md.push_back(hs.instances[i]); gid.push_back(hs.inst_first_tri[i]); inst.push_back(i);
      first.push_back(first.back() + (hs.inst_first_tri[i + 1] - hs.inst_first_tri[i]));
    }
  }
  RT_HIP(bl.d_md.upload(md.data(), md.size(), r->stream));
  RT_HIP(bl.d_first.upload(first.data(), first.size(), r->stream));
  RT_HIP(bl.d_gid.upload(gid.data(), gid.size(), r->stream));
  RT_HIP(bl.d_inst.upload(inst.data(), inst.size(), r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));
  BvhBuffers& b = bl.b;
  b.primitives = bl.d_md.ptr; b.inst_first_tri = bl.d_first.ptr; b.instance_count = (uint32_t)md.size(); b.tri_count = first.back();
  b.gid_first = bl.d_gid.ptr; b.inst_index = bl.d_inst.ptr; b.object_space = bl.object_space;
  b.tris_by_id = r->d_tris_by_id.ptr + bl.tri_off; b.tris = r->d_tris.ptr + bl.tri_off; b.shade_tris = r->d_shade_tris.ptr + bl.tri_off;
  b.tris_any = r->any_invisible ? r->d_tris_any.ptr + bl.tri_off : nullptr;
  b.material_any_class = r->d_material_any_class.ptr; b.material_kind = r->d_material_kind.ptr; b.material_count = (uint32_t)hs.gpu_materials.size();
  b.nodes = r->d_nodes.ptr + bl.node_off;
  b.opt = r->bvh.opt;
  const std::string e = refit ? bvh_refit(b, r->stream) : bvh_build(b, kLeafMax, r->stream);
  if (!e.empty()) RT_FAIL(e);
  const std::string e2 = bvh_relocate(b, bl.node_off, bl.tri_off, r->stream);
  if (!e2.empty()) RT_FAIL(e2);
  return HALA_OK;
}

int build_two_level(hala_rt_renderer* r) {
  const HostScene& hs = r->hs;
  r->blas.clear();
  r->prim_blas.assign(hs.prims.size(), -1);
  // the trees: [0] the world tree over the instances that stay flattened (if any), then one per instanced primitive in order of first use
  std::unique_ptr<hala_rt_renderer::Blas> world(new hala_rt_renderer::Blas());
  uint32_t n_items = 0;
  for (size_t i = 0; i < hs.instances.size(); ++i) {
    if (!r->inst_instanced[i]) { world->insts.push_back((uint32_t)i); continue; }
    ++n_items;
    const uint32_t p = hs.instance_prim[i];
    if (r->prim_blas[p] < 0) r->prim_blas[p] = -2;  // marked; numbered below
  }
  uint32_t tri_at = 0;
  if (!world->insts.empty()) {
    for (uint32_t i : world->insts) tri_at += hs.inst_first_tri[i + 1] - hs.inst_first_tri[i];
    world->tri_off = 0; world->b.tri_count = tri_at;
    ++n_items;
    r->blas.push_back(std::move(world));
  }
  for (size_t i = 0; i < hs.instances.size(); ++i) {
    const uint32_t p = hs.instance_prim[i];
    if (!r->inst_instanced[i] || r->prim_blas[p] != -2) continue;
    std::unique_ptr<hala_rt_renderer::Blas> bl(new hala_rt_renderer::Blas());
    bl->object_space = true; bl->prim = p; bl->tri_off = tri_at;
    bl->b.tri_count = (uint32_t)(hs.prims[p].indices.size() / 3);
    tri_at += bl->b.tri_count;
    r->prim_blas[p] = (int32_t)r->blas.size();
    r->blas.push_back(std::move(bl));
  }
  if (tri_at >= (1u << 28)) RT_FAIL("The scene stores 2^28 triangles or more.");
  r->stored_tris = tri_at;
  r->tlas_capacity = std::max(1u, n_items);
  uint32_t node_at = r->tlas_capacity;
  for (auto& bl : r->blas) {
    bl->node_off = node_at;
    bl->node_cap = std::max<uint32_t>(bl->b.tri_count, 2) - 1;
    node_at += bl->node_cap;
  }
  RT_HIP(r->d_tris_by_id.resize(tri_at)); RT_HIP(r->d_tris.resize(tri_at)); RT_HIP(r->d_shade_tris.resize(tri_at));
  RT_HIP(r->d_nodes.resize(node_at));
  RT_HIP(hipMemsetAsync(r->d_nodes.ptr, 0xff, (size_t)node_at * sizeof(BvhNode4), r->stream));  // unused slots of the reserved ranges: absent children
  if (attach_any_triangles(r) != HALA_OK) return HALA_ERR;
  uint32_t nodes_used = r->tlas_capacity;
  for (auto& bl : r->blas) {
    if (blas_build_or_refit(r, *bl, false) != HALA_OK) return HALA_ERR;
    nodes_used = std::max(nodes_used, bl->node_off + bl->b.node_count);
  }
  r->bvh.tri_count = tri_at;
  r->bvh.node_count = node_at;  // the node array as a whole (reserved ranges included: hala_rt_download_bvh)
  r->bvh.tris_any = r->any_invisible ? r->d_tris_any.ptr : nullptr;
  r->leaf_max_built = kLeafMax;
  if (build_instance_levels(r) != HALA_OK) return HALA_ERR;
  return configure_traversal(r);
}

int build_bvh(hala_rt_renderer* r) {
  classify_instances(r, &r->inst_instanced);
  r->two_level = false;
  for (uint8_t f : r->inst_instanced) r->two_level = r->two_level || f != 0;
  if (r->bvh.topology) { bvh_free_topology(r->bvh.topology); r->bvh.topology = nullptr; }
  r->blas.clear();
  r->bvh.gid_first = nullptr; r->bvh.inst_index = nullptr; r->bvh.object_space = false;
  if (r->two_level) return build_two_level(r);
  const uint32_t n = r->hs.triangle_count;
  r->stored_tris = n; r->tlas_nodes = 0; r->tlas_capacity = 0;
  RT_HIP(r->d_tris_by_id.resize(n)); RT_HIP(r->d_tris.resize(n)); RT_HIP(r->d_shade_tris.resize(n));
  RT_HIP(r->d_nodes.resize(std::max<uint32_t>(n, 2) - 1));
  r->bvh.primitives = r->d_instances.ptr; r->bvh.inst_first_tri = r->d_inst_first_tri.ptr;
  r->bvh.instance_count = (uint32_t)r->hs.instances.size(); r->bvh.tri_count = n;
  r->bvh.tris_by_id = r->d_tris_by_id.ptr; r->bvh.shade_tris = r->d_shade_tris.ptr; r->bvh.tris = r->d_tris.ptr; r->bvh.nodes = r->d_nodes.ptr;
  if (attach_any_triangles(r) != HALA_OK) return HALA_ERR;
  // a scene this small will be staged in LDS (configure_traversal: 48 B per triangle + at most ~32 B of nodes per triangle)
  uint32_t leaf_max = (size_t)n * 80 <= kLdsStageBudget ? kLeafMaxStaged : kLeafMax;
  if (const char* ev = tune_env("HALART_LEAF_MAX")) leaf_max = std::min(8u, std::max(1u, (uint32_t)atoi(ev)));  // tuning knob
  if ((size_t)n * 80 > kLdsStageBudget) leaf_max = std::min(leaf_max, traverse_max_leaf(false));  // one consumer lane per triangle of a leaf item
  const std::string e = bvh_build(r->bvh, leaf_max, r->stream);
  if (!e.empty()) RT_FAIL(e);
  r->leaf_max_built = leaf_max;
  return configure_traversal(r);
}

std::string file_stem(const char* path) {
  std::string p(path);
  const size_t slash = p.find_last_of("/\\");
  std::string base = slash == std::string::npos ? p : p.substr(slash + 1);
  const size_t dot = base.find_last_of('.');
  if (dot != std::string::npos && dot != 0) base = base.substr(0, dot);
  return base;
}

int install_envmap(hala_rt_renderer* r, const float* pixels, uint32_t channels, uint32_t w, uint32_t h, float rotation,
                   const float* cached_total, const float* cached_marginal, const float* cached_conditional) {
  if (!pixels || w == 0 || h == 0 || (channels != 3 && channels != 4)) RT_FAIL("Unsupported color type for environment map.");  // src/envmap.rs:57-60
  std::vector<float> data((size_t)w * h * 4);
  for (size_t i = 0; i < (size_t)w * h; ++i) {  // src/envmap.rs:63-89
    for (uint32_t c = 0; c < 3; ++c) {
      const float v = pixels[i * channels + c];
      if (std::isnan(v)) RT_FAIL("The pixel value is NaN!");
      if (std::isinf(v)) RT_FAIL("The pixel value is infinite!");
      data[4 * i + c] = v;
    }
    data[4 * i + 3] = 1.0f;  // :87
  }
  RT_HIP(r->d_env.upload(reinterpret_cast<const float4*>(data.data()), (size_t)w * h, r->stream));
  RT_HIP(r->d_env_total.resize(1)); RT_HIP(r->d_marginal.resize(h)); RT_HIP(r->d_conditional.resize((size_t)w * h));
  if (cached_total) {  // ./out/<stem>.dist_cache hit (src/envmap.rs:91-117)
    RT_HIP(hipMemcpyAsync(r->d_env_total.ptr, cached_total, 4, hipMemcpyHostToDevice, r->stream));
    RT_HIP(hipMemcpyAsync(r->d_marginal.ptr, cached_marginal, (size_t)h * 4, hipMemcpyHostToDevice, r->stream));
    RT_HIP(hipMemcpyAsync(r->d_conditional.ptr, cached_conditional, (size_t)w * h * 4, hipMemcpyHostToDevice, r->stream));
    RT_HIP(hipStreamSynchronize(r->stream));
    r->env_total_sum = *cached_total;
  } else {
    const std::string e = envmap_build_distribution(r->d_env.ptr, w, h, r->d_env_total.ptr, r->d_marginal.ptr, r->d_conditional.ptr, r->stream);
    if (!e.empty()) RT_FAIL(e);
    RT_HIP(hipMemcpy(&r->env_total_sum, r->d_env_total.ptr, 4, hipMemcpyDeviceToHost));
  }
  r->env_w = w; r->env_h = h; r->has_env = true; r->env_rotation = rotation;  // src/rt_renderer.rs:1192
  return HALA_OK;
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* hala_last_error_message(void) { return get_last_error(); }
const char* hala_version(void) { return "halart 0.1 (gfx950)"; }

int hala_rt_create(const char* name, uint32_t width, uint32_t height, int device_ordinal, uint32_t max_depth, uint32_t rr_depth,
                   int enable_tonemap, int enable_aces, int use_simple_aces, uint64_t max_frames, hala_rt_renderer** out) {
  if (!out) RT_FAIL("The output handle is null!");
  *out = nullptr;
  if (width == 0 || height == 0) RT_FAIL("The renderer resolution is zero!");
  if (max_depth == 0 || max_depth > kMaxDepth) RT_FAIL("max_depth must be in 1.." + std::to_string(kMaxDepth) + ".");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) RT_FAIL("No HIP device is available: libhalart has no CPU path.");
  if (device_ordinal < 0 || device_ordinal >= count) RT_FAIL("The requested device ordinal does not exist.");
  RT_HIP(hipSetDevice(device_ordinal));
  std::unique_ptr<hala_rt_renderer> r(new hala_rt_renderer());
  if (const char* ev = tune_env("HALART_EVENT_PERIOD")) r->launch_event_period = (uint32_t)std::max(0, atoi(ev));  // tuning knob
  r->name = name ? name : "";
  r->width = width; r->height = height; r->device = device_ordinal;
  r->max_depth = max_depth; r->rr_depth = rr_depth;
  r->enable_tonemap = enable_tonemap != 0; r->enable_aces = enable_aces != 0; r->use_simple_aces = use_simple_aces != 0;
  r->max_frames = max_frames == 0 ? UINT64_MAX : max_frames;  // src/rt_renderer.rs:774
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  r->cu_count = (uint32_t)prop.multiProcessorCount;
  RT_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
  if (const char* ev = tune_env("HALART_FUSE")) r->fuse_mode = (uint32_t)std::max(0, std::min(2, atoi(ev)));
  compute_tiling(r.get());
  // create_storage_images (src/rt_renderer.rs:818-917): final, accum, albedo, normal
  if (alloc_frame_buffers(r.get()) != HALA_OK) return HALA_ERR;
  RT_HIP(hipStreamSynchronize(r->stream));
  *out = r.release();
  return HALA_OK;
}

void hala_rt_destroy(hala_rt_renderer* r) { delete r; }

int hala_rt_push_general_shader(hala_rt_renderer* r, const void* code, size_t code_size, int stage, const char*) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if (!code || code_size == 0) RT_FAIL("The shader code is empty!");
  if (stage == 0) r->n_raygen++; else if (stage == 1) r->n_miss++; else if (stage == 2) r->n_callable++; else RT_FAIL("Invalid general shader stage.");
  return HALA_OK;
}
int hala_rt_push_general_shader_with_file(hala_rt_renderer* r, const char* file_path, int stage, const char* debug_name) {
  if (!r) RT_FAIL("The renderer handle is null!");
  struct stat st;
  if (!file_path || stat(file_path, &st) != 0) RT_FAIL(std::string("Failed to load shader file \"") + (file_path ? file_path : "") + "\".");
  static const char dummy = 0;
  return hala_rt_push_general_shader(r, &dummy, 1, stage, debug_name);
}
int hala_rt_push_hit_shaders(hala_rt_renderer* r, const void* ch, size_t chs, const void* ah, size_t ahs, const void* is, size_t iss, const char*) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if ((!ch || !chs) && (!ah || !ahs) && (!is || !iss)) RT_FAIL("The hit shader group is empty!");
  r->n_hit++;
  return HALA_OK;
}
int hala_rt_push_hit_shaders_with_file(hala_rt_renderer* r, const char* ch, const char* ah, const char* is, const char*) {
  if (!r) RT_FAIL("The renderer handle is null!");
  struct stat st;
  for (const char* p : {ch, ah, is}) if (p && stat(p, &st) != 0) RT_FAIL(std::string("Failed to load shader file \"") + p + "\".");
  if (!ch && !ah && !is) RT_FAIL("The hit shader group is empty!");
  r->n_hit++;
  return HALA_OK;
}

int hala_rt_load_blue_noise_pixels(hala_rt_renderer* r, const uint8_t* rgba8, uint32_t width, uint32_t height);
int hala_rt_load_blue_noise_texture(hala_rt_renderer* r, const char* path) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if (!path || !*path || file_stem(path).empty()) RT_FAIL("The file name is none!");  // src/rt_renderer.rs:1120
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> px;
  std::string e;
  try { e = rt::decode_image_file_rgba8(path, &w, &h, &px); }
  catch (const std::exception& ex) { e = std::string("Failed to open image \"") + path + "\": " + ex.what(); }  // nothing is thrown across the C ABI
  if (!e.empty()) RT_FAIL(e);
  return hala_rt_load_blue_noise_pixels(r, px.data(), w, h);
}
int hala_rt_load_blue_noise_pixels(hala_rt_renderer* r, const uint8_t* rgba8, uint32_t width, uint32_t height) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!rgba8 || !width || !height) RT_FAIL("The blue noise texture is empty!");
  RT_HIP(r->blue_noise.upload(rgba8, (size_t)width * height * 4, r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));
  r->blue_w = width; r->blue_h = height;
  return HALA_OK;
}

int hala_rt_set_scene(hala_rt_renderer* r, const hala_scene_desc* scene) {
  RtRange range("halart::set_scene");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  RT_HIP(hipStreamSynchronize(r->stream));
  r->has_scene = false; r->committed = false;  // "Release the old scene in the GPU." (src/rt_renderer.rs:1164)
  const std::string e = r->hs.assign(scene);
  if (!e.empty()) RT_FAIL(e);
  if (upload_packed(r) != HALA_OK) return HALA_ERR;
  if (upload_textures(r) != HALA_OK) return HALA_ERR;
  r->has_scene = true;
  return HALA_OK;
}

int hala_rt_set_envmap_pixels(hala_rt_renderer* r, const float* pixels, uint32_t channels, uint32_t width, uint32_t height, float rotation_degrees) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  RT_HIP(hipStreamSynchronize(r->stream));
  return install_envmap(r, pixels, channels, width, height, rotation_degrees, nullptr, nullptr, nullptr);
}

int hala_rt_set_envmap_file(hala_rt_renderer* r, const char* path, float rotation_degrees) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!path || !*path) RT_FAIL("The file name is none!");  // src/envmap.rs:45
  HostImage img;
  const std::string e = load_float_image(path, &img);
  if (!e.empty()) RT_FAIL(e);
  RT_HIP(hipStreamSynchronize(r->stream));
  // ./out/<stem>.dist_cache: [f32 total_sum][f32 x H][f32 x W*H], native endian, no header (src/envmap.rs:90-142)
  const std::string cache = "./out/" + file_stem(path) + ".dist_cache";
  const size_t W = img.width, H = img.height;
  std::vector<float> blob(1 + H + W * H);
  FILE* f = fopen(cache.c_str(), "rb");
  if (f) {
    const size_t got = fread(blob.data(), 4, blob.size(), f);
    fclose(f);
    if (got != blob.size()) RT_FAIL("Failed to read from file.");  // :101, :107, :114
    return install_envmap(r, img.pixels.data(), img.channels, img.width, img.height, rotation_degrees, &blob[0], &blob[1], &blob[1 + H]);
  }
  if (install_envmap(r, img.pixels.data(), img.channels, img.width, img.height, rotation_degrees, nullptr, nullptr, nullptr) != HALA_OK) return HALA_ERR;
  blob[0] = r->env_total_sum;
  RT_HIP(hipMemcpy(&blob[1], r->d_marginal.ptr, H * 4, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(&blob[1 + H], r->d_conditional.ptr, W * H * 4, hipMemcpyDeviceToHost));
  f = fopen(cache.c_str(), "wb");
  if (!f) RT_FAIL("Failed to create file \"" + cache + "\".");  // :125 (the reference does not create ./out either)
  const size_t put = fwrite(blob.data(), 4, blob.size(), f);
  if (fclose(f) != 0 || put != blob.size()) RT_FAIL("Failed to write to file.");
  return HALA_OK;
}

void hala_rt_set_ground_color(hala_rt_renderer* r, const float rgba[4]) { if (r && rgba) memcpy(r->ground, rgba, 16); }
void hala_rt_set_sky_color(hala_rt_renderer* r, const float rgba[4]) { if (r && rgba) memcpy(r->sky, rgba, 16); }
void hala_rt_set_env_intensity(hala_rt_renderer* r, float v) { if (r) r->env_intensity = v; }
void hala_rt_set_exposure_value(hala_rt_renderer* r, float v) { if (r) r->exposure = v; }

int hala_rt_commit(hala_rt_renderer* r) {
  RtRange range("halart::commit");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene) RT_FAIL("The scene in GPU is none!");  // src/rt_renderer.rs:138
  if (r->hs.cameras.empty()) RT_FAIL("The scene has no camera.");
  if (r->hs.instances.empty()) RT_FAIL("The scene has no mesh primitive.");  // `primitives[0]` panics in the reference (gpu_uploader.rs:888)
  if (build_bvh(r) != HALA_OK) return HALA_ERR;
  r->committed = true;
  r->reset_accumulation();
  return HALA_OK;
}

int hala_rt_set_build_options(hala_rt_renderer* r, const hala_rt_build_options* o) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if (!o) RT_FAIL("The build options are null!");
  if (o->builder > 3u || o->ploc_tail > 2u || o->instancing > 2u) RT_FAIL("Invalid build options.");
  for (uint32_t v : o->reserved) if (v != 0u) RT_FAIL("Invalid build options (reserved fields must be 0).");
  r->instancing_mode = o->instancing;
  r->bvh.opt.builder = o->builder; r->bvh.opt.ploc_tail = o->ploc_tail;
  r->bvh.opt.ploc_look_every = o->ploc_look_every; r->bvh.opt.collapse_look_every = o->collapse_look_every;
  return HALA_OK;
}

// `frames` consecutive update()s in one wavefront pass.  Bookkeeping per frame as in the reference: total_frames is
// incremented first (pre_update, src/renderer.rs:278) and a frame whose number exceeds max_frames is skipped
// (src/rt_renderer.rs:394-396); the frames that do render share one kernel sequence with `samples` paths per pixel.
static int update_impl(hala_rt_renderer* r, uint32_t frames) {
  RtRange range("halart::update");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->committed) RT_FAIL("The pipeline is none!");  // src/rt_renderer.rs:443
  const uint64_t first = r->total_frames;  // frame_index of the first frame of this batch = total_frames - 1 after its increment
  r->total_frames += frames;
  if (first >= r->max_frames) return HALA_OK;
  const uint32_t samples = (uint32_t)std::min<uint64_t>(frames, r->max_frames - first);
  if (samples > r->batch_capacity) {
    RT_HIP(hipStreamSynchronize(r->stream));
    if (alloc_wavefront(r, samples) != HALA_OK) return HALA_ERR;
  }
  hala_global_uniform u{};                                 // :408-427
  memcpy(u.ground_color, r->ground, 16); memcpy(u.sky_color, r->sky, 16);
  u.resolution[0] = (float)r->width; u.resolution[1] = (float)r->height;
  u.max_depth = r->max_depth; u.rr_depth = r->rr_depth;
  u.frame_index = (uint32_t)first;  // == total_frames - 1 for a single-frame update (:414)
  u.camera_index = 0;
  u.env_type = r->has_env ? 1u : 0u;
  u.env_map_width = r->has_env ? r->env_w : 0; u.env_map_height = r->has_env ? r->env_h : 0;
  u.env_total_sum = r->has_env ? r->env_total_sum : 0.0f;
  u.env_rotation = r->env_rotation / 360.0f;
  u.env_intensity = r->env_intensity; u.exposure_value = r->exposure;
  u.enable_tonemap = r->enable_tonemap; u.enable_aces = r->enable_aces; u.use_simple_aces = r->use_simple_aces;
  u.num_of_lights = (uint32_t)r->hs.lights.size();
  r->last_uniform = u;
  r->last_uniform.frame_index = (uint32_t)(first + samples - 1);  // what the last frame of the batch would have uploaded

  TraceEvents& te = r->ring[r->ring_pos];
  r->ring_pos = (r->ring_pos + 1) % kStatRing;
  r->resolve_slot(te);
  if (!te.frame_begin) { RT_HIP(hipEventCreate(&te.frame_begin)); RT_HIP(hipEventCreate(&te.frame_end)); RT_HIP(hipHostMalloc(reinterpret_cast<void**>(&te.host_counts), 14 * sizeof(unsigned long long), hipHostMallocDefault)); }
  te.used = 0; te.counted = r->counting; te.shadow_launches = 0; te.fused_mask = 0; te.traced_mask = 0;

  te.samples = samples;
  const FrameConst fc = r->frame_const(u, samples);
  const SceneView sv = r->view();
  const Queues q = r->queues();
  const PathState ps = r->path_state();
  Control* ctl = r->d_ctl.ptr;
  hipStream_t s = r->stream;
  if (r->scratch_acquire(s) != HALA_OK) return HALA_ERR;
  RT_HIP(hipEventRecord(te.frame_begin, s));
  RT_HIP(hipMemsetAsync(ctl, 0, sizeof(Control), s));
  // per-launch HIP events (statistics: traverse_*_ms_total) on every launch_event_period-th update; each record is a barrier
  // packet on the stream, i.e. a few microseconds between two launches
  const bool timed = r->launch_event_period == 1u || (r->launch_event_period > 1u && (r->update_counter % r->launch_event_period) == 0u);
  r->update_counter++;
  // The shadow passes of bounce d and the closest-hit traversal of bounce d + 1 are independent: untimed updates issue them as ONE
  // persistent launch (k_trace_shadow_then_batch: one tail of long rays instead of three).  Updates that carry per-launch timing events or
  // counting kernels keep one launch per pass, so that every measured launch is one kernel symbol with the chip to itself.
  const bool fuse = (r->fuse_mode == 2u || (r->fuse_mode == 1u && !timed)) && !r->counting && (u.num_of_lights > 0 || u.env_type == 1u);
  constexpr size_t kQueueSizeWords = (kMaxDepth + 1) + 2 * kMaxDepth;
  if (timed && !te.host_queue_sizes) RT_HIP(hipHostMalloc(reinterpret_cast<void**>(&te.host_queue_sizes), kQueueSizeWords * sizeof(uint32_t), hipHostMallocDefault));
  bool traced = false;  // the closest-hit pass of this depth already ran inside the previous depth's fused launch
  for (uint32_t depth = 0; depth < r->max_depth; ++depth) {
    if (timed) { hipEvent_t a = r->next_event(te); RT_HIP(hipEventRecord(a, s)); }
    // depth 0: the camera rays are generated inside the traversal kernel, there is no ray-generation pass
    if (depth == 0) {
      launch_trace_primary(r->lcfg, sv, fc, q.hits, &ctl->work_closest, ctl, r->real_pixels * samples, r->counting, s);
      if (r->counting) RT_HIP(hipMemcpyAsync(ctl->primary_steps, ctl->steps[0], 16, hipMemcpyDeviceToDevice, s));
    }
    else if (!traced) launch_trace_batch(r->lcfg, sv, q.rays[depth & 1u], q.hits, &ctl->n_active[depth], 0, &ctl->work_closest, ctl, false, r->counting, true, s);
    traced = false;
    if (timed) { hipEvent_t b = r->next_event(te); RT_HIP(hipEventRecord(b, s)); }
    launch_shade(fc, sv, q, ps, ctl, depth, s);
    if (timed) { hipEvent_t c = r->next_event(te); RT_HIP(hipEventRecord(c, s)); }
    // light connections add to the path's L, environment connections to its Le (RENDER_SPEC §6): the two passes are independent of each
    // other and of the next bounce's closest-hit pass
    const uint32_t kinds = (u.num_of_lights > 0 ? 1u : 0u) | (u.env_type == 1u ? 2u : 0u);
    const bool last = depth + 1u >= r->max_depth;  // no closest-hit pass follows: only worth one launch when there are two shadow passes
    if (fuse && kinds && (!last || kinds == 3u) && launch_trace_shadow_then_batch(r->lcfg, sv, q, ps, ctl, depth, kinds, !last, s)) {
      traced = !last;
      if (timed) { te.fused_mask |= 1ull << depth; if (traced) te.traced_mask |= 1ull << (depth + 1u); }
    }
    else
      for (uint32_t kind = 0; kind < 2u; ++kind) {
        if (!((kinds >> kind) & 1u)) continue;
        launch_trace_shadow(r->lcfg, sv, q, ps, ctl, depth, kind, r->counting, s);
        te.shadow_launches += timed ? 1u : 0u;
      }
    if (timed) { hipEvent_t d = r->next_event(te); RT_HIP(hipEventRecord(d, s)); }
  }
  launch_resolve(fc, ps, r->img_local[0].ptr, r->img_local[1].ptr, r->img_local[2].ptr, r->img_local[3].ptr, s);
  RT_HIP(hipMemcpyAsync(te.host_counts, &ctl->rays_closest, 14 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  if (timed) RT_HIP(hipMemcpyAsync(te.host_queue_sizes, ctl->n_active, kQueueSizeWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RT_HIP(hipEventRecord(te.frame_end, s));
  r->scratch_event = te.frame_end; r->scratch_stream = s;
  RT_HIP(hipGetLastError());
  te.pending = true;
  for (bool& v : r->full_valid) v = false;
  return HALA_OK;
}

int hala_rt_update(hala_rt_renderer* r, double, uint32_t, uint32_t) { return update_impl(r, 1); }

int hala_rt_update_batch(hala_rt_renderer* r, uint32_t frames) {
  if (!r) RT_FAIL("The renderer handle is null!");
  while (frames > 0) {
    const uint32_t chunk = std::min(frames, kMaxSampleBatch);
    if (update_impl(r, chunk) != HALA_OK) return HALA_ERR;
    frames -= chunk;
  }
  return HALA_OK;
}

int hala_rt_render(hala_rt_renderer* r) {  // src/rt_renderer.rs:475-502: nothing to present; make the frame's work visible
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (r->total_frames > r->max_frames) return HALA_OK;  // :484-486
  // submit_and_present_frame hands the frame to the queue and only blocks on the fence of the swapchain image it reuses: with no
  // swapchain here, render() bounds the updates in flight to two (it waits for the update before the latest), so the host can
  // enqueue the next frame while this one runs.  Everything that reads results (read_image, save_images, statistics, wait_idle,
  // tile_buffer users via wait_idle) synchronises on its own.
  const TraceEvents& before_last = r->ring[(r->ring_pos + kStatRing - 2) % kStatRing];
  if (before_last.pending && before_last.frame_end) RT_HIP(hipEventSynchronize(before_last.frame_end));
  return HALA_OK;
}
int hala_rt_wait_idle(hala_rt_renderer* r) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  RT_HIP(hipStreamSynchronize(r->stream));
  return HALA_OK;
}

int hala_rt_read_image(hala_rt_renderer* r, int which, float* dst) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (which < 0 || which > 3 || !dst) RT_FAIL("Invalid image selector.");
  RT_HIP(hipStreamSynchronize(r->stream));  // wait_idle (src/rt_renderer.rs:1242)
  const size_t bytes = (size_t)r->width * r->height * sizeof(float4);
  if (r->world <= 1) { RT_HIP(hipMemcpy(dst, r->img_local[which].ptr, bytes, hipMemcpyDeviceToHost)); return HALA_OK; }
  if (!r->full_valid[which]) RT_FAIL("The frame is sharded across ranks: gather the tiles (hala_rt_scatter_gathered_tiles) before reading the image.");
  RT_HIP(hipMemcpy(dst, r->img_full[which].ptr, bytes, hipMemcpyDeviceToHost));
  return HALA_OK;
}

int hala_rt_save_images(hala_rt_renderer* r, const char* path) {
  RtRange range("halart::save_images");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!path || !*path) RT_FAIL("The file name is none!");  // src/rt_renderer.rs:1234
  std::string p(path);
  const size_t slash = p.find_last_of("/\\");
  const std::string dir = slash == std::string::npos ? "" : p.substr(0, slash + 1);
  const std::string stem = file_stem(path);
  std::vector<float> px((size_t)r->width * r->height * 4);
  static const char* suffix[3] = {"_color.pfm", "_albedo.pfm", "_normal.pfm"};  // :1235-1237
  for (int which = 0; which < 3; ++which) {
    if (hala_rt_read_image(r, which, px.data()) != HALA_OK) return HALA_ERR;
    if (which == 0) tonemap_pixels(px.data(), (size_t)r->width * r->height, r->enable_tonemap, r->enable_aces, r->use_simple_aces);  // :1256-1316
    const std::string e = write_pfm((dir + stem + suffix[which]).c_str(), px.data(), r->width, r->height);
    if (!e.empty()) RT_FAIL(e);
  }
  return HALA_OK;
}

int hala_rt_get_info(hala_rt_renderer* r, hala_rt_info* out) {
  if (!r || !out) RT_FAIL("The renderer handle is null!");
  out->width = r->width; out->height = r->height;
  return HALA_OK;
}
int hala_rt_get_statistics(hala_rt_renderer* r, hala_rt_statistics* out) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!out) RT_FAIL("The output pointer is null!");
  RT_HIP(hipStreamSynchronize(r->stream));
  for (int k = 0; k < kStatRing; ++k) r->resolve_slot(r->ring[(r->ring_pos + k) % kStatRing]);  // oldest first
  r->stats.total_frames = r->total_frames;
  *out = r->stats;
  return HALA_OK;
}
int hala_rt_reset_accumulation(hala_rt_renderer* r) {
  if (!r) RT_FAIL("The renderer handle is null!");
  r->reset_accumulation();
  return HALA_OK;
}
int hala_rt_set_launch_timing_period(hala_rt_renderer* r, uint32_t period) {
  if (!r) RT_FAIL("The renderer handle is null!");
  r->launch_event_period = period;
  r->update_counter = 0;
  return HALA_OK;
}
int hala_rt_set_pass_fusion(hala_rt_renderer* r, uint32_t mode) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if (mode > 2u) RT_FAIL("Invalid pass fusion mode.");
  r->fuse_mode = mode;
  return HALA_OK;
}
int hala_rt_set_counting(hala_rt_renderer* r, int enable) {
  if (!r) RT_FAIL("The renderer handle is null!");
  r->counting = enable != 0;
  return HALA_OK;
}
int hala_rt_get_global_uniform(hala_rt_renderer* r, hala_global_uniform* out) {
  if (!r || !out) RT_FAIL("The renderer handle is null!");
  *out = r->last_uniform;
  return HALA_OK;
}

#define RT_READBACK(dev, T, dst, cap, cnt)                                                               \
  do {                                                                                                     \
    *(cnt) = (uint32_t)(dev).count;                                                                        \
    const size_t _n = std::min<size_t>((cap), (dev).count);                                                \
    if ((dst) && _n) RT_HIP(hipMemcpy((dst), (dev).ptr, _n * sizeof(T), hipMemcpyDeviceToHost));           \
  } while (0)

int hala_rt_get_packed_cameras(hala_rt_renderer* r, hala_gpu_camera* dst, uint32_t capacity, uint32_t* count) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || !count) RT_FAIL("The scene in GPU is none!");
  RT_READBACK(r->d_cameras, hala_gpu_camera, dst, capacity, count);
  return HALA_OK;
}
int hala_rt_get_packed_lights(hala_rt_renderer* r, hala_gpu_light* dst, hala_aabb* bb, uint32_t capacity, uint32_t* count) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || !count) RT_FAIL("The scene in GPU is none!");
  RT_READBACK(r->d_lights, hala_gpu_light, dst, capacity, count);
  if (bb) memcpy(bb, r->hs.light_aabbs.data(), std::min<size_t>(capacity, r->hs.light_aabbs.size()) * sizeof(hala_aabb));
  return HALA_OK;
}
int hala_rt_get_packed_materials(hala_rt_renderer* r, hala_gpu_material* dst, uint32_t capacity, uint32_t* count) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || !count) RT_FAIL("The scene in GPU is none!");
  RT_READBACK(r->d_materials, hala_gpu_material, dst, capacity, count);
  return HALA_OK;
}
int hala_rt_get_packed_primitives(hala_rt_renderer* r, hala_gpu_mesh_data* dst, float* t3x4, uint32_t capacity, uint32_t* count) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || !count) RT_FAIL("The scene in GPU is none!");
  RT_READBACK(r->d_instances, hala_gpu_mesh_data, dst, capacity, count);
  if (t3x4) memcpy(t3x4, r->hs.instance_3x4.data(), std::min<size_t>(capacity, r->hs.instances.size()) * 12 * sizeof(float));
  return HALA_OK;
}
int hala_rt_get_env_distribution(hala_rt_renderer* r, float* total_sum, float* marginal, float* conditional) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_env) RT_FAIL("The environment map is none!");
  if (total_sum) *total_sum = r->env_total_sum;
  if (marginal) RT_HIP(hipMemcpy(marginal, r->d_marginal.ptr, (size_t)r->env_h * 4, hipMemcpyDeviceToHost));
  if (conditional) RT_HIP(hipMemcpy(conditional, r->d_conditional.ptr, (size_t)r->env_w * r->env_h * 4, hipMemcpyDeviceToHost));
  return HALA_OK;
}

// ---- textures (set 2 binding 0) ---------------------------------------------------------------------------------------
int hala_rt_get_texture_info(hala_rt_renderer* r, uint32_t texture, uint32_t* width, uint32_t* height, uint32_t* mips) {
  if (!r) RT_FAIL("The renderer handle is null!");
  if (!r->has_scene || texture >= r->host_textures.size()) RT_FAIL("The texture does not exist.");
  const TexDesc& td = r->host_textures[texture];
  if (width) *width = td.width;
  if (height) *height = td.height;
  if (mips) *mips = td.mips;
  return HALA_OK;
}
int hala_rt_read_texture_level(hala_rt_renderer* r, uint32_t texture, uint32_t level, float* dst_rgba32f) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || texture >= r->host_textures.size() || !dst_rgba32f) RT_FAIL("The texture does not exist.");
  const TexDesc& td = r->host_textures[texture];
  if (level >= td.mips) RT_FAIL("The mip level does not exist.");
  const uint32_t lw = std::max(1u, td.width >> level), lh = std::max(1u, td.height >> level);
  const size_t n = (size_t)lw * lh;
  if (td.format == kTexFloat) {
    RT_HIP(hipMemcpy(dst_rgba32f, r->d_tex_arena.ptr + td.mip_offset[level], n * 16, hipMemcpyDeviceToHost));
    return HALA_OK;
  }
  // 8-bit texels: de-tile and decode on the host exactly like the sampler does on the device
  std::vector<uint32_t> tiled(tex_tiled_size(lw, lh));
  RT_HIP(hipMemcpy(tiled.data(), r->d_tex_arena8.ptr + td.mip_offset[level], tiled.size() * 4, hipMemcpyDeviceToHost));
  const float* lut = srgb_decode_lut();
  for (uint32_t y = 0; y < lh; ++y)
    for (uint32_t x = 0; x < lw; ++x) {
      const uint32_t t = tiled[tex_tiled_index(x, y, lw)];
      float* o = dst_rgba32f + ((size_t)y * lw + x) * 4;
      for (int c = 0; c < 3; ++c) { const uint32_t b = (t >> (8 * c)) & 0xffu; o[c] = td.format == kTexSrgb8 ? lut[b] : (float)b / 255.0f; }
      o[3] = (float)(t >> 24) / 255.0f;
    }
  return HALA_OK;
}
int hala_rt_sample_texture_host(hala_rt_renderer* r, uint32_t texture, const float* uv_lod, uint32_t count, float* dst_rgba32f) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->has_scene || texture >= r->host_textures.size()) RT_FAIL("The texture does not exist.");
  if (!count) return HALA_OK;
  if (!uv_lod || !dst_rgba32f) RT_FAIL("Invalid argument.");
  DeviceArray<float> d_in;
  DeviceArray<float4> d_out;
  RT_HIP(d_in.upload(uv_lod, (size_t)count * 3, r->stream));
  RT_HIP(d_out.resize(count));
  launch_sample_texture(r->view(), texture, d_in.ptr, count, d_out.ptr, r->stream);
  RT_HIP(hipMemcpyAsync(dst_rgba32f, d_out.ptr, (size_t)count * 16, hipMemcpyDeviceToHost, r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));
  RT_HIP(hipGetLastError());
  return HALA_OK;
}

// ---- multi-GPU tiles ------------------------------------------------------------------------------------------------
int hala_rt_tile_allgather_finish(hala_rt_renderer* r);
int hala_rt_set_tile_shard(hala_rt_renderer* r, uint32_t rank, uint32_t world, uint32_t tile_size) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (world == 0 || rank >= world) RT_FAIL("Invalid rank / world size.");
  if (tile_size == 0 || tile_size > 256) RT_FAIL("Invalid tile size.");
  // a collective in flight belongs to the old shard: complete it (its receive buffer is laid out for the old world size)
  if (r->gather_pending && hala_rt_tile_allgather_finish(r) != HALA_OK) return HALA_ERR;
  // a communicator is bound to (rank, world): gather_recv is sized by it and the de-interleave indexes it by the shard's world
  if (r->comm && ((uint32_t)r->comm_rank != rank || (uint32_t)r->comm_world != world))
    RT_FAIL("The renderer holds a communicator for rank " + std::to_string(r->comm_rank) + " of " + std::to_string(r->comm_world) +
            ": call hala_rt_comm_destroy before changing the tile shard.");
  RT_HIP(hipStreamSynchronize(r->stream));
  if (r->gather_stream) RT_HIP(hipStreamSynchronize(r->gather_stream));
  r->rank = rank; r->world = world; r->tile_size = tile_size;
  compute_tiling(r);
  if (alloc_frame_buffers(r) != HALA_OK) return HALA_ERR;
  RT_HIP(hipStreamSynchronize(r->stream));
  r->reset_accumulation();
  return HALA_OK;
}
int hala_rt_tile_buffer(hala_rt_renderer* r, int which, void** d_ptr, size_t* bytes) {
  if (!r || which < 0 || which > 3 || !d_ptr || !bytes) RT_FAIL("Invalid argument.");
  *d_ptr = r->img_local[which].ptr;
  *bytes = r->image_pixels() * sizeof(float4);
  return HALA_OK;
}
int hala_rt_get_stream(hala_rt_renderer* r, void** hip_stream) {
  if (!hip_stream) RT_FAIL("Invalid argument.");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  *hip_stream = static_cast<void*>(r->stream);
  return HALA_OK;
}
int hala_rt_scatter_gathered_tiles_on_stream(hala_rt_renderer* r, int which, const void* d_gathered, size_t bytes, void* hip_stream) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (which < 0 || which > 3 || !d_gathered) RT_FAIL("Invalid argument.");
  if (r->world <= 1) RT_FAIL("The renderer is not sharded.");
  if (bytes != r->image_pixels() * r->world * sizeof(float4)) RT_FAIL("The gathered buffer has the wrong size.");
  RT_HIP(r->img_full[which].resize((size_t)r->width * r->height));
  hala_global_uniform u = r->last_uniform;
  const FrameConst fc = r->frame_const(u);
  launch_scatter_tiles(fc, static_cast<const float4*>(d_gathered), r->img_full[which].ptr, hip_stream ? static_cast<hipStream_t>(hip_stream) : r->stream);  // stream ordered: readers wait themselves
  RT_HIP(hipGetLastError());
  r->full_valid[which] = true;
  return HALA_OK;
}
int hala_rt_scatter_gathered_tiles(hala_rt_renderer* r, int which, const void* d_gathered, size_t bytes) {
  return hala_rt_scatter_gathered_tiles_on_stream(r, which, d_gathered, bytes, nullptr);
}


// ---- RCCL tile all-gather (BASELINE.json north_star: "RCCL all-gather of tiles over xGMI") -----------------------------------------
// librccl is resolved on first use (dyn_api.h): a one-GPU host loads libhalart.so without it.
#define RT_RCCL_API(api)                                   \
  std::string _rccl_err;                                   \
  const RcclApi* api = rccl_api(&_rccl_err);               \
  if (!api) RT_FAIL(_rccl_err)
#define RT_NCCL(api, expr)                                                                                         \
  do {                                                                                                             \
    const ncclResult_t _r = (expr);                                                                                \
    if (_r != ncclSuccess) RT_FAIL(std::string("RCCL: ") + (api)->GetErrorString(_r) + " (" #expr ")");            \
  } while (0)

int hala_rt_comm_unique_id(void* out_128_bytes) {
  if (!out_128_bytes) RT_FAIL("Invalid argument.");
  static_assert(sizeof(ncclUniqueId) == HALA_COMM_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  RT_RCCL_API(api);
  ncclUniqueId id;
  RT_NCCL(api, api->GetUniqueId(&id));
  memcpy(out_128_bytes, &id, sizeof(id));
  return HALA_OK;
}
// the side stream and the three hand-over events of the exchange (with or without a communicator)
static int ensure_gather_resources(hala_rt_renderer* r) {
  if (!r->gather_stream) RT_HIP(hipStreamCreateWithFlags(&r->gather_stream, hipStreamNonBlocking));
  for (hipEvent_t* e : {&r->ev_rendered, &r->ev_staged, &r->ev_gathered}) if (!*e) RT_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  return HALA_OK;
}
static int comm_common(hala_rt_renderer* r, int rank, int world) {
  if ((uint32_t)world != r->world || (uint32_t)rank != r->rank)
    RT_FAIL("The communicator's rank / size (" + std::to_string(rank) + " / " + std::to_string(world) + ") differ from the renderer's tile shard (" +
            std::to_string(r->rank) + " / " + std::to_string(r->world) + "): call hala_rt_set_tile_shard first.");
  r->comm_rank = rank; r->comm_world = world;
  return ensure_gather_resources(r);
}
int hala_rt_comm_init_rank(hala_rt_renderer* r, const void* unique_id_128_bytes, uint32_t rank, uint32_t world) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!unique_id_128_bytes || world == 0 || rank >= world) RT_FAIL("Invalid argument.");
  if (r->comm) RT_FAIL("The renderer already has a communicator.");
  RT_RCCL_API(api);
  if (comm_common(r, (int)rank, (int)world) != HALA_OK) return HALA_ERR;
  ncclUniqueId id;
  memcpy(&id, unique_id_128_bytes, sizeof(id));
  RT_NCCL(api, api->CommInitRank(&r->comm, (int)world, id, (int)rank));
  r->comm_owned = true;
  return HALA_OK;
}
int hala_rt_comm_attach(hala_rt_renderer* r, void* nccl_comm) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!nccl_comm) RT_FAIL("Invalid argument.");
  if (r->comm) RT_FAIL("The renderer already has a communicator.");
  RT_RCCL_API(api);
  int rank = 0, world = 0;
  RT_NCCL(api, api->CommUserRank(static_cast<ncclComm_t>(nccl_comm), &rank));
  RT_NCCL(api, api->CommCount(static_cast<ncclComm_t>(nccl_comm), &world));
  if (comm_common(r, rank, world) != HALA_OK) return HALA_ERR;
  r->comm = static_cast<ncclComm_t>(nccl_comm);
  r->comm_owned = false;
  return HALA_OK;
}
int hala_rt_comm_destroy(hala_rt_renderer* r) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (r->gather_stream) RT_HIP(hipStreamSynchronize(r->gather_stream));
  if (r->comm && r->comm_owned) {
    RT_RCCL_API(api);
    RT_NCCL(api, api->CommDestroy(r->comm));
  }
  r->comm = nullptr; r->comm_owned = false; r->gather_pending = 0;
  return HALA_OK;
}

// finish(k - 1) -> [side stream waits for the renderer's stream: frame k is complete] -> staging <- tiles -> [renderer's stream waits
// for that copy: frame k + 1 may overwrite the tiles] -> the exchange (receive <- every rank's staging) on the side stream.  Nothing
// blocks the host.  external = false: the exchange is ncclAllGather on the renderer's communicator.  external = true
// (hala_rt_tile_allgather_begin_external): the CALLER performs it — a host with another transport (MPI, a gloo rehearsal on one GPU,
// tests that emulate the ranks) reads the staging buffer and fills the receive buffer on the exchange stream (hala_rt_get_exchange_buffers) —
// everything else (staging copy, event order, de-interleave in finish) is this very code.
static int allgather_begin(hala_rt_renderer* r, uint32_t aov_mask, bool external) {
  if (aov_mask == 0u || aov_mask > 15u) RT_FAIL("Invalid AOV mask.");
  if (hala_rt_tile_allgather_finish(r) != HALA_OK) return HALA_ERR;
  if (ensure_gather_resources(r) != HALA_OK) return HALA_ERR;
  const uint32_t world = external ? r->world : (uint32_t)r->comm_world;
  if (world != r->world) RT_FAIL("The communicator's size differs from the renderer's tile shard.");  // (set_tile_shard refuses the change)
  const size_t n = r->image_pixels();
  hipStream_t g = r->gather_stream;
  RT_HIP(hipEventRecord(r->ev_rendered, r->stream));
  RT_HIP(hipStreamWaitEvent(g, r->ev_rendered, 0));
  for (int which = 0; which < 4; ++which) {
    if (!(aov_mask & (1u << which))) continue;
    RT_HIP(r->gather_stage[which].resize(n));
    RT_HIP(r->gather_recv[which].resize(n * (size_t)world));
    RT_HIP(hipMemcpyAsync(r->gather_stage[which].ptr, r->img_local[which].ptr, n * sizeof(float4), hipMemcpyDeviceToDevice, g));
  }
  RT_HIP(hipEventRecord(r->ev_staged, g));
  RT_HIP(hipStreamWaitEvent(r->stream, r->ev_staged, 0));
  if (!external) {
    RT_RCCL_API(api);
    for (int which = 0; which < 4; ++which)
      if (aov_mask & (1u << which))
        RT_NCCL(api, api->AllGather(r->gather_stage[which].ptr, r->gather_recv[which].ptr, n * 4, ncclFloat, r->comm, g));
  }
  r->gather_pending = aov_mask;
  return HALA_OK;
}
int hala_rt_tile_allgather_begin(hala_rt_renderer* r, uint32_t aov_mask) {
  RtRange range("halart::tile_allgather_begin");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->comm) RT_FAIL("The renderer has no communicator: call hala_rt_comm_init_rank or hala_rt_comm_attach first.");
  return allgather_begin(r, aov_mask, false);
}
int hala_rt_tile_allgather_begin_external(hala_rt_renderer* r, uint32_t aov_mask) {
  RtRange range("halart::tile_allgather_begin_external");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  return allgather_begin(r, aov_mask, true);
}
int hala_rt_get_exchange_buffers(hala_rt_renderer* r, int which, void** d_staged, size_t* staged_bytes, void** d_receive, size_t* receive_bytes, void** hip_stream) {
  if (!r || which < 0 || which > 3) RT_FAIL("Invalid argument.");
  if (!(r->gather_pending & (1u << which))) RT_FAIL("No exchange of this image is in flight: call hala_rt_tile_allgather_begin_external first.");
  if (d_staged) *d_staged = r->gather_stage[which].ptr;
  if (staged_bytes) *staged_bytes = r->gather_stage[which].bytes();
  if (d_receive) *d_receive = r->gather_recv[which].ptr;
  if (receive_bytes) *receive_bytes = r->gather_recv[which].bytes();
  if (hip_stream) *hip_stream = static_cast<void*>(r->gather_stream);
  return HALA_OK;
}
// de-interleave on the side stream (beside the rendering of the next frame), then whatever the renderer's stream does next — and
// whoever waits for it — sees the row-major images complete
int hala_rt_tile_allgather_finish(hala_rt_renderer* r) {
  RtRange range("halart::tile_allgather_finish");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->gather_pending) return HALA_OK;
  const uint32_t mask = r->gather_pending;
  r->gather_pending = 0;
  hipStream_t g = r->gather_stream;
  if (r->world > 1) {
    const FrameConst fc = r->frame_const(r->last_uniform);
    for (int which = 0; which < 4; ++which) {
      if (!(mask & (1u << which))) continue;
      if (r->gather_recv[which].count != r->image_pixels() * (size_t)r->world) RT_FAIL("The receive buffer does not match the tile shard.");
      RT_HIP(r->img_full[which].resize((size_t)r->width * r->height));
      launch_scatter_tiles(fc, r->gather_recv[which].ptr, r->img_full[which].ptr, g);
      r->full_valid[which] = true;
    }
  }
  RT_HIP(hipEventRecord(r->ev_gathered, g));
  RT_HIP(hipStreamWaitEvent(r->stream, r->ev_gathered, 0));
  RT_HIP(hipGetLastError());
  return HALA_OK;
}
int hala_rt_tile_allgather(hala_rt_renderer* r, uint32_t aov_mask) {
  if (hala_rt_tile_allgather_begin(r, aov_mask) != HALA_OK) return HALA_ERR;
  return hala_rt_tile_allgather_finish(r);
}
int hala_rt_get_gathered_buffer(hala_rt_renderer* r, int which, void** d_ptr, size_t* bytes) {
  if (!r || which < 0 || which > 3 || !d_ptr || !bytes) RT_FAIL("Invalid argument.");
  *d_ptr = r->gather_recv[which].ptr;
  *bytes = r->gather_recv[which].bytes();
  return HALA_OK;
}

// ---- ray-batch operator ------------------------------------------------------------------------------------------------
int hala_rt_trace_rays(hala_rt_renderer* r, const hala_ray* d_rays, hala_hit* d_hits, uint32_t count, int mode, uint64_t* d_counters, void* hip_stream) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");  // src/rt_renderer.rs:284
  if (mode != 0 && mode != 1) RT_FAIL("Invalid trace mode.");
  if (count == 0) return HALA_OK;
  if (!d_rays || !d_hits) RT_FAIL("The ray batch is null!");
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : r->stream;
  if (r->scratch_acquire(s) != HALA_OK) return HALA_ERR;  // stream-ordered behind the previous user of the renderer's scratch (include/halart.h)
  RT_HIP(hipMemsetAsync(r->d_batch_work.ptr, 0, sizeof(WorkCounters), s));
  // counters: the kernel accumulates into the control block's 64-bit fields; copy them out if requested
  if (d_counters) RT_HIP(hipMemsetAsync(&r->d_ctl.ptr->steps[mode][0], 0, 16, s));
  launch_trace_batch(r->lcfg, r->view(), d_rays, d_hits, nullptr, count, r->d_batch_work.ptr, r->d_ctl.ptr, mode == 1, d_counters != nullptr, false, s);
  if (d_counters) RT_HIP(hipMemcpyAsync(d_counters, &r->d_ctl.ptr->steps[mode][0], 16, hipMemcpyDeviceToDevice, s));
  if (!r->batch_done) RT_HIP(hipEventCreateWithFlags(&r->batch_done, hipEventDisableTiming));
  RT_HIP(hipEventRecord(r->batch_done, s));
  r->scratch_event = r->batch_done; r->scratch_stream = s;
  RT_HIP(hipGetLastError());
  return HALA_OK;
}
int hala_rt_trace_rays_indirect(hala_rt_renderer* r, const hala_ray* d_rays, hala_hit* d_hits, const uint32_t* d_indirect, int mode, void* hip_stream) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!d_indirect) RT_FAIL("The indirect command address is null!");
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : r->stream;
  uint32_t whd[3] = {0, 0, 0};
  RT_HIP(hipMemcpyAsync(whd, d_indirect, 12, hipMemcpyDeviceToHost, s));
  RT_HIP(hipStreamSynchronize(s));
  const uint64_t n = (uint64_t)whd[0] * whd[1] * whd[2];
  if (n > 0xffffffffull) RT_FAIL("The indirect launch is too large.");
  return hala_rt_trace_rays(r, d_rays, d_hits, (uint32_t)n, mode, nullptr, hip_stream);
}
int hala_rt_trace_rays_host(hala_rt_renderer* r, const hala_ray* rays, hala_hit* hits, uint32_t count, int mode, uint64_t counters[2]) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (count == 0) return HALA_OK;
  DeviceArray<hala_ray> d_rays;
  DeviceArray<hala_hit> d_hits;
  DeviceArray<uint64_t> d_ctr;
  RT_HIP(d_rays.upload(rays, count, r->stream));
  RT_HIP(d_hits.resize(count));
  if (counters) RT_HIP(d_ctr.resize(2));
  if (hala_rt_trace_rays(r, d_rays.ptr, d_hits.ptr, count, mode, counters ? d_ctr.ptr : nullptr, r->stream) != HALA_OK) return HALA_ERR;
  RT_HIP(hipMemcpyAsync(hits, d_hits.ptr, (size_t)count * sizeof(hala_hit), hipMemcpyDeviceToHost, r->stream));
  if (counters) RT_HIP(hipMemcpyAsync(counters, d_ctr.ptr, 16, hipMemcpyDeviceToHost, r->stream));
  RT_HIP(hipStreamSynchronize(r->stream));
  return HALA_OK;
}

int hala_rt_get_bvh_info(hala_rt_renderer* r, hala_bvh_info* out) {
  if (!r || !out) RT_FAIL("The renderer handle is null!");
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");
  out->node_count = r->bvh.node_count; out->triangle_count = r->hs.triangle_count; out->max_depth = r->bvh.max_depth; out->lds_node_count = r->lds_nodes;
  out->node_width = 4u;
  out->stored_triangle_count = r->stored_tris; out->instance_node_count = r->two_level ? r->tlas_nodes : 0u;
  out->instance_ref_count = r->two_level ? (uint32_t)r->inst_refs.size() : 0u;
  out->tree_bytes = (uint64_t)r->d_nodes.bytes() + r->d_tris.bytes() + r->d_tris_any.bytes() + r->d_shade_tris.bytes() + r->d_inst_refs.bytes() + r->d_inst_info.bytes();
  memcpy(out->scene_min, r->bvh.scene_min, 12); memcpy(out->scene_max, r->bvh.scene_max, 12);
  return HALA_OK;
}
int hala_rt_download_bvh(hala_rt_renderer* r, void* nodes_64B, void* triangles_48B) {
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");
  RT_HIP(hipStreamSynchronize(r->stream));
  if (nodes_64B) RT_HIP(hipMemcpy(nodes_64B, r->d_nodes.ptr, (size_t)r->bvh.node_count * 64, hipMemcpyDeviceToHost));
  if (triangles_48B && r->bvh.tri_count) {
    RT_HIP(hipMemcpy(triangles_48B, r->d_tris.ptr, (size_t)r->bvh.tri_count * 48, hipMemcpyDeviceToHost));
    Tri* t = static_cast<Tri*>(triangles_48B);
    for (uint32_t i = 0; i < r->bvh.tri_count; ++i) t[i].pad2 = 0u;  // word 11 is the library's own (shading kind for the hit queue): not part of the 48-B format
  }
  return HALA_OK;
}

int hala_rt_download_instance_refs(hala_rt_renderer* r, void* refs_64B, uint32_t capacity, uint32_t* count) {
  if (!r || !count) RT_FAIL("Invalid argument.");
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");
  *count = r->two_level ? (uint32_t)r->inst_refs.size() : 0u;
  if (refs_64B && r->two_level) memcpy(refs_64B, r->inst_refs.data(), std::min<size_t>(capacity, r->inst_refs.size()) * sizeof(InstRef));
  return HALA_OK;
}
int hala_rt_update_node_transform(hala_rt_renderer* r, uint32_t node_index, const float local_transform[16]) {
  if (!r || !local_transform) RT_FAIL("Invalid argument.");
  if (!r->has_scene || node_index >= r->hs.nodes.size()) RT_FAIL("The node does not exist.");
  memcpy(r->hs.nodes[node_index].local.m, local_transform, 64);
  return HALA_OK;
}
int hala_rt_update_vertices(hala_rt_renderer* r, uint32_t mesh_index, uint32_t primitive_index, const hala_vertex* vertices, uint32_t vertex_count) {
  if (!r || !vertices) RT_FAIL("Invalid argument.");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");
  if (mesh_index + 1u >= r->hs.mesh_first_prim.size()) RT_FAIL("The mesh does not exist.");
  const uint32_t first = r->hs.mesh_first_prim[mesh_index], end = r->hs.mesh_first_prim[mesh_index + 1u];
  if (primitive_index >= end - first) RT_FAIL("The primitive does not exist.");
  HostPrimitive& p = r->hs.prims[first + primitive_index];
  if (vertex_count != p.vertices.size()) RT_FAIL("The vertex count differs from the primitive's (" + std::to_string(p.vertices.size()) + "): refit keeps the topology, use set_scene + commit.");
  for (uint32_t k = 0; k < vertex_count; ++k)
    if (!std::isfinite(vertices[k].position[0]) || !std::isfinite(vertices[k].position[1]) || !std::isfinite(vertices[k].position[2])) RT_FAIL("Vertex position is not finite.");
  memcpy(p.vertices.data(), vertices, (size_t)vertex_count * sizeof(hala_vertex));
  // the copy below reads the renderer's own host copy, which outlives it; earlier frames still read the arena: wait for them
  RT_HIP(hipStreamSynchronize(r->stream));
  r->vertices_dirty = true;
  if (vertex_count) RT_HIP(hipMemcpyAsync(r->d_vertices.ptr + r->prim_vertex_offset[first + primitive_index], p.vertices.data(), (size_t)vertex_count * sizeof(hala_vertex), hipMemcpyHostToDevice, r->stream));
  return HALA_OK;
}
int hala_rt_update_material(hala_rt_renderer* r, uint32_t material_index, const hala_material_desc* material) {
  if (!r || !material) RT_FAIL("Invalid argument.");
  if (!r->has_scene || material_index >= r->hs.materials.size()) RT_FAIL("The material does not exist.");
  if (material->type > 1u) RT_FAIL("Invalid material type.");  // cpu/material.rs:14
  if (r->hs.materials[material_index].opacity == 0.0f || material->opacity == 0.0f) r->materials_dirty_any = true;
  r->hs.materials[material_index] = *material;
  return HALA_OK;
}
int hala_rt_refit(hala_rt_renderer* r) {
  RtRange range("halart::refit");
  if (ensure_device(r) != HALA_OK) return HALA_ERR;
  if (!r->committed) RT_FAIL("The top level acceleration structure is none!");
  RT_HIP(hipStreamSynchronize(r->stream));
  const std::vector<hala_gpu_mesh_data> before = r->hs.instances;  // object -> world of every instance as the tree was fitted to it
  const std::vector<uint8_t> kinds_before = r->material_kind;
  r->hs.update_node_hierarchies();
  const std::string e = r->hs.pack();
  if (!e.empty()) RT_FAIL(e);
  if (upload_packed(r, false) != HALA_OK) return HALA_ERR;
  r->bvh.primitives = r->d_instances.ptr; r->bvh.inst_first_tri = r->d_inst_first_tri.ptr;
  // only cameras / lights moved (the interactive case: a camera node): the geometry and its tree stand as they are
  const bool had_invisible = r->any_invisible;
  const std::vector<uint8_t> classes_before = r->material_any_class;
  if (attach_any_triangles(r) != HALA_OK) return HALA_ERR;
  if (classes_before != r->material_any_class) r->materials_dirty_any = true;  // the any-hit copy of the triangles must be rewritten
  // (a material edit can change which triangles the shadow rays see: their copy is rewritten by the refit pass)
  // (the BVH-order triangles carry their material's shading kind: rewritten by the refit pass as well)
  bool geometry_moved = r->vertices_dirty || r->materials_dirty_any || had_invisible != r->any_invisible || before.size() != r->hs.instances.size() ||
                        kinds_before != r->material_kind;
  r->materials_dirty_any_refit = r->materials_dirty_any;
  r->materials_dirty_any = false;
  for (size_t i = 0; i < before.size() && !geometry_moved; ++i) geometry_moved = memcmp(before[i].transform, r->hs.instances[i].transform, 64) != 0;
  // which instances are intersected in object space may have changed (a transform that is no longer invertible, or is again): rebuild
  std::vector<uint8_t> flags;
  classify_instances(r, &flags);
  if (flags != r->inst_instanced) {
    if (build_bvh(r) != HALA_OK) return HALA_ERR;
    r->vertices_dirty = false;
  } else if (r->two_level) {
    // RENDER_SPEC 4.5: a node that moves an instanced primitive only touches the instance levels (rebuilt on the host below).  The trees
    // underneath are refitted when what THEY hold changed: vertices or materials (any tree), the transform of a flattened instance (the world tree)
    const bool content = r->vertices_dirty || r->materials_dirty_any_refit || had_invisible != r->any_invisible || kinds_before != r->material_kind;
    r->bvh.tris_any = r->any_invisible ? r->d_tris_any.ptr : nullptr;
    for (auto& bl : r->blas) {
      bool moved = content;
      if (!bl->object_space)
        for (uint32_t i : bl->insts) moved = moved || memcmp(before[i].transform, r->hs.instances[i].transform, 64) != 0;
      if (moved && blas_build_or_refit(r, *bl, true) != HALA_OK) return HALA_ERR;
    }
    if (build_instance_levels(r) != HALA_OK) return HALA_ERR;
    if (configure_traversal(r) != HALA_OK) return HALA_ERR;
    r->vertices_dirty = false;
  } else if (geometry_moved) {
    const std::string e2 = bvh_refit(r->bvh, r->stream);
    if (!e2.empty()) RT_FAIL(e2);
    if (configure_traversal(r) != HALA_OK) return HALA_ERR;
    r->vertices_dirty = false;
  }
  r->reset_accumulation();  // like the device-lost path: accumulation restarts (src/rt_renderer.rs:557)
  return HALA_OK;
}

// ---- stand-alone pieces ---------------------------------------------------------------------------------------------------
int hala_envmap_build_distribution(int device_ordinal, const float* rgba32f, uint32_t width, uint32_t height, float* total_sum, float* marginal, float* conditional) {
  if (!rgba32f || !total_sum || !marginal || !conditional || !width || !height) RT_FAIL("Invalid argument.");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) RT_FAIL("No HIP device is available: libhalart has no CPU path.");
  RT_HIP(hipSetDevice(device_ordinal));
  DeviceArray<float4> d_px;
  DeviceArray<float> d_total, d_m, d_c;
  const size_t n = (size_t)width * height;
  RT_HIP(d_px.upload(reinterpret_cast<const float4*>(rgba32f), n, nullptr));
  RT_HIP(d_total.resize(1)); RT_HIP(d_m.resize(height)); RT_HIP(d_c.resize(n));
  const std::string e = envmap_build_distribution(d_px.ptr, width, height, d_total.ptr, d_m.ptr, d_c.ptr, nullptr);
  if (!e.empty()) RT_FAIL(e);
  RT_HIP(hipMemcpy(total_sum, d_total.ptr, 4, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(marginal, d_m.ptr, (size_t)height * 4, hipMemcpyDeviceToHost));
  RT_HIP(hipMemcpy(conditional, d_c.ptr, n * 4, hipMemcpyDeviceToHost));
  return HALA_OK;
}

void hala_tonemap_pixels(float* rgba32f, size_t pixel_count, int enable_tonemap, int enable_aces, int use_simple_aces) {
  if (rgba32f) tonemap_pixels(rgba32f, pixel_count, enable_tonemap, enable_aces, use_simple_aces);
}
int hala_write_pfm(const char* path, const float* rgba32f, uint32_t width, uint32_t height) {
  if (!path || !rgba32f) RT_FAIL("Invalid argument.");
  const std::string e = write_pfm(path, rgba32f, width, height);
  if (!e.empty()) RT_FAIL(e);
  return HALA_OK;
}

int hala_load_float_image(const char* path, uint32_t* width, uint32_t* height, uint32_t* channels, float* dst, size_t capacity_floats) {
  if (!path || !width || !height || !channels) RT_FAIL("Invalid argument.");
  HostImage img;
  const std::string e = load_float_image(path, &img);
  if (!e.empty()) RT_FAIL(e);
  *width = img.width; *height = img.height; *channels = img.channels;
  if (dst) {
    if (capacity_floats < img.pixels.size()) RT_FAIL("The destination buffer is too small.");
    memcpy(dst, img.pixels.data(), img.pixels.size() * sizeof(float));
  }
  return HALA_OK;
}

int hala_rtprog_parse_desc(const char* desc_json, hala_rtprog_desc_info* out) {
  // serde field names and defaults of HalaRayTracingProgramDesc (src/raytracing_program.rs:33-55)
  if (!desc_json || !out) RT_FAIL("Invalid argument.");
  JsonValue root;
  const std::string e = json_parse(desc_json, &root);
  if (!e.empty()) RT_FAIL("Failed to parse the ray tracing program description: " + e);
  if (root.kind != JsonValue::Object) RT_FAIL("The ray tracing program description is not an object.");
  auto string_array = [&](const char* key, bool required, uint32_t* n) -> int {
    const JsonValue* v = root.find(key);
    if (!v) { if (required) RT_FAIL(std::string("missing field `") + key + "`"); *n = 0; return HALA_OK; }
    if (v->kind != JsonValue::Array) RT_FAIL(std::string("field `") + key + "` is not an array");
    for (const auto& it : v->items) if (it.kind != JsonValue::String) RT_FAIL(std::string("field `") + key + "` must hold strings");
    *n = (uint32_t)v->items.size();
    return HALA_OK;
  };
  memset(out, 0, sizeof(*out));
  if (string_array("raygen_shader_file_paths", true, &out->raygen_count) != HALA_OK) return HALA_ERR;
  if (string_array("miss_shader_file_paths", false, &out->miss_count) != HALA_OK) return HALA_ERR;
  if (string_array("callable_shader_file_paths", false, &out->callable_count) != HALA_OK) return HALA_ERR;
  if (string_array("bindings", false, &out->binding_count) != HALA_OK) return HALA_ERR;
  const JsonValue* hits = root.find("hit_shader_file_paths");
  if (!hits) RT_FAIL("missing field `hit_shader_file_paths`");
  if (hits->kind != JsonValue::Array) RT_FAIL("field `hit_shader_file_paths` is not an array");
  for (const auto& h : hits->items) {
    if (h.kind != JsonValue::Object) RT_FAIL("a hit shader description is not an object");
    for (const auto& m : h.members) {
      if (m.first != "closest_hit_shader_file_path" && m.first != "any_hit_shader_file_path" && m.first != "intersection_shader_file_path") continue;
      if (m.second.kind != JsonValue::String && m.second.kind != JsonValue::Null) RT_FAIL("field `" + m.first + "` must be a string or null");
    }
  }
  out->hit_count = (uint32_t)hits->items.size();
  auto u32_field = [&](const char* key, uint32_t def, uint32_t* dst) -> int {
    const JsonValue* v = root.find(key);
    if (!v) { *dst = def; return HALA_OK; }
    if (v->kind != JsonValue::Number || v->num < 0 || v->num > 4294967295.0 || v->num != std::floor(v->num)) RT_FAIL(std::string("field `") + key + "` is not a u32");
    *dst = (uint32_t)v->num;
    return HALA_OK;
  };
  if (u32_field("push_constant_size", 0, &out->push_constant_size) != HALA_OK) return HALA_ERR;
  if (u32_field("ray_recursion_depth", 1, &out->ray_recursion_depth) != HALA_OK) return HALA_ERR;  // default_ray_recursion_depth :53-55
  return HALA_OK;
}


// ---- HalaRayTracingProgram (src/raytracing_program.rs:70-341) as an object of the C ABI -------------------------------------------------
// Reference: {shader groups, pipeline, SBT}; bind() attaches descriptor sets, push_constants() writes the constant block, trace_rays(w, h, d)
// launches w*h*d ray-gen invocations against the acceleration structure the descriptor sets name.  Here the shader groups are the library's
// traversal kernels (the SPIR-V paths of the description are recorded, as hala_rt_push_*_shader does), the "descriptor sets" are the device
// buffers of one ray batch, and the acceleration structure is the committed renderer's.  Bytes 0..3 of the constant block select the
// hit-group behaviour: 0 = closest hit, 1 = any hit.
struct hala_rtprog {
  hala_rt_renderer* renderer = nullptr;
  hala_rtprog_desc_info info{};
  std::string debug_name;
  std::vector<uint8_t> constants;
  const hala_ray* d_rays = nullptr;
  hala_hit* d_hits = nullptr;
};

int hala_rtprog_create(hala_rt_renderer* r, const char* desc_json, const char* debug_name, hala_rtprog** out) {
  if (!out) RT_FAIL("The output handle is null!");
  *out = nullptr;
  if (!r) RT_FAIL("The renderer handle is null!");
  hala_rtprog_desc_info info;
  if (hala_rtprog_parse_desc(desc_json, &info) != HALA_OK) return HALA_ERR;
  if (info.raygen_count == 0) RT_FAIL("The raygen shader list is empty!");  // a pipeline without a ray generation group cannot be built (:85-106)
  if (info.push_constant_size % 4u != 0u) RT_FAIL("push_constant_size must be a multiple of 4.");  // VkPushConstantRange.size
  std::unique_ptr<hala_rtprog> p(new hala_rtprog());
  p->renderer = r; p->info = info; p->debug_name = debug_name ? debug_name : "";
  p->constants.assign(std::max<uint32_t>(info.push_constant_size, 4u), 0);
  *out = p.release();
  return HALA_OK;
}
void hala_rtprog_destroy(hala_rtprog* p) { delete p; }
int hala_rtprog_get_desc_info(const hala_rtprog* p, hala_rtprog_desc_info* out) {
  if (!p || !out) RT_FAIL("Invalid argument.");
  *out = p->info;
  return HALA_OK;
}
int hala_rtprog_bind(hala_rtprog* p, const hala_ray* d_rays, hala_hit* d_hits) {  // :264-278
  if (!p) RT_FAIL("The program handle is null!");
  if (!d_rays || !d_hits) RT_FAIL("The ray batch is null!");
  p->d_rays = d_rays; p->d_hits = d_hits;
  return HALA_OK;
}
int hala_rtprog_push_constants(hala_rtprog* p, uint32_t offset, const void* data, size_t len) {  // :285-300
  if (!p) RT_FAIL("The program handle is null!");
  if (!data && len) RT_FAIL("Invalid argument.");
  if ((size_t)offset + len > p->constants.size()) RT_FAIL("The push constant range exceeds push_constant_size.");
  if (len) memcpy(p->constants.data() + offset, data, len);
  return HALA_OK;
}
int hala_rtprog_push_constants_f32(hala_rtprog* p, uint32_t offset, const float* data, size_t count) {  // :307-322
  return hala_rtprog_push_constants(p, offset, data, count * sizeof(float));
}
static int rtprog_mode(const hala_rtprog* p) {
  uint32_t m = 0;
  memcpy(&m, p->constants.data(), 4);
  return (int)(m & 1u);
}
int hala_rtprog_trace_rays(hala_rtprog* p, uint32_t width, uint32_t height, uint32_t depth, void* hip_stream) {  // :330-332
  if (!p) RT_FAIL("The program handle is null!");
  if (!p->d_rays || !p->d_hits) RT_FAIL("The program is not bound to a ray batch.");
  const uint64_t n = (uint64_t)width * height * depth;
  if (n > 0xffffffffull) RT_FAIL("The launch is too large.");
  return hala_rt_trace_rays(p->renderer, p->d_rays, p->d_hits, (uint32_t)n, rtprog_mode(p), nullptr, hip_stream);
}
int hala_rtprog_trace_rays_indirect(hala_rtprog* p, const uint32_t* d_indirect, void* hip_stream) {  // :338-340
  if (!p) RT_FAIL("The program handle is null!");
  if (!p->d_rays || !p->d_hits) RT_FAIL("The program is not bound to a ray batch.");
  return hala_rt_trace_rays_indirect(p->renderer, p->d_rays, p->d_hits, d_indirect, rtprog_mode(p), hip_stream);
}

}  // extern "C"
