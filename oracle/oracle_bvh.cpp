// ORACLE — TEST INFRASTRUCTURE ONLY (see spec_math.h).
// oracle_bvh.cpp — scene flattening (RENDER_SPEC §3), a CPU binned-SAH BVH2 builder (the oracle's own; the
// product builds its BVH on the GPU and the two are compared through traversal RESULTS, never topology), the
// traversal rule of RENDER_SPEC §4, a brute-force intersector and a structural validator for product BVHs.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle_api.h"
#include "oracle_scene.h"

using namespace orc;

static constexpr uint32_t kAbsent = 0xffffffffu;

// RENDER_SPEC §3: p' = (fma(m8,z, fma(m4,y, m0*x))) + m12  (column-major 4x4, affine)
static inline V3 transform_point(const float* m, V3 p) {
  return V3{fmaf(m[8], p.z, fmaf(m[4], p.y, m[0] * p.x)) + m[12], fmaf(m[9], p.z, fmaf(m[5], p.y, m[1] * p.x)) + m[13],
            fmaf(m[10], p.z, fmaf(m[6], p.y, m[2] * p.x)) + m[14]};
}

// ---------------------------------------------------------------------------------------------------------
// builder
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Box {
  float mn[3], mx[3];
  void reset() { for (int i = 0; i < 3; ++i) { mn[i] = std::numeric_limits<float>::infinity(); mx[i] = -mn[i]; } }
  void grow(const float* p) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
  void grow(const Box& b) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); } }
  float half_area() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (dx < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};
struct BuildRef { Box box; float c[3]; uint32_t id; };
struct ChildRef { uint32_t child, count; Box box; };

struct Builder {
  std::vector<BuildRef>& refs;
  std::vector<Node>& nodes;
  uint32_t max_leaf = 4;
  Builder(std::vector<BuildRef>& r, std::vector<Node>& n) : refs(r), nodes(n) {}

  ChildRef build(uint32_t first, uint32_t count) {
    Box bounds, cb;
    bounds.reset(); cb.reset();
    for (uint32_t i = first; i < first + count; ++i) { bounds.grow(refs[i].box); cb.grow(refs[i].c); }
    auto make_leaf = [&]() { return ChildRef{first, count, bounds}; };
    if (count == 1) return make_leaf();
    // binned SAH over 3 axes, 16 bins
    const int NB = 16;
    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = -1, best_split = -1;
    for (int axis = 0; axis < 3; ++axis) {
      float lo = cb.mn[axis], hi = cb.mx[axis];
      if (!(hi > lo)) continue;
      Box bb[NB]; uint32_t bc[NB];
      for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
      float scale = (float)NB / (hi - lo);
      for (uint32_t i = first; i < first + count; ++i) {
        int b = std::min(NB - 1, std::max(0, (int)((refs[i].c[axis] - lo) * scale)));
        bb[b].grow(refs[i].box); bc[b]++;
      }
      float right_area[NB]; uint32_t right_cnt[NB];
      Box acc; acc.reset(); uint32_t cnt = 0;
      for (int b = NB - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
      acc.reset(); cnt = 0;
      for (int b = 0; b < NB - 1; ++b) {
        acc.grow(bb[b]); cnt += bc[b];
        if (cnt == 0 || right_cnt[b + 1] == 0) continue;
        float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
      }
    }
    uint32_t mid;
    if (best_axis < 0) {
      if (count <= max_leaf) return make_leaf();
      mid = first + count / 2;  // all centroids coincide: split by index
    } else {
      float leaf_cost = (float)count * bounds.half_area();
      float split_cost = 0.5f * bounds.half_area() + best_cost;
      if (count <= max_leaf && leaf_cost <= split_cost) return make_leaf();
      float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
      float scale = (float)NB / (hi - lo);
      auto it = std::partition(refs.begin() + first, refs.begin() + first + count, [&](const BuildRef& r) {
        int b = std::min(NB - 1, std::max(0, (int)((r.c[best_axis] - lo) * scale)));
        return b <= best_split;
      });
      mid = (uint32_t)(it - refs.begin());
      if (mid == first || mid == first + count) mid = first + count / 2;
    }
    uint32_t my = (uint32_t)nodes.size();
    nodes.push_back(Node{});
    ChildRef a = build(first, mid - first);
    ChildRef b = build(mid, first + count - mid);
    Node& n = nodes[my];
    memcpy(n.c0min, a.box.mn, 12); memcpy(n.c0max, a.box.mx, 12);
    memcpy(n.c1min, b.box.mn, 12); memcpy(n.c1max, b.box.mx, 12);
    n.child0 = a.child; n.count0 = a.count; n.child1 = b.child; n.count1 = b.count;
    return ChildRef{my, 0, bounds};
  }
};
}  // namespace

static void build_bvh(orc_scene* s) {
  const uint32_t N = (uint32_t)s->tris_by_id.size();
  std::vector<BuildRef> refs(N);
  float amax = 0.0f;
  for (int k = 0; k < 3; ++k) amax = std::max(amax, std::max(std::fabs(s->bounds_min[k]), std::fabs(s->bounds_max[k])));
  const float pad = amax * 1.9073486328125e-06f;  // 2^-19
  for (uint32_t i = 0; i < N; ++i) {
    // world-space vertices (the triangles of instanced instances are stored in object space: RENDER_SPEC 4.5; the oracle's own tree is
    // one world-space tree over ALL triangles either way — only the ray/triangle test follows the rule)
    const float* w9 = &s->tri_verts9[(size_t)i * 9];
    refs[i].box.reset(); refs[i].box.grow(w9); refs[i].box.grow(w9 + 3); refs[i].box.grow(w9 + 6);
    if (s->instances[s->tri_instance[i]].instanced)  // moved there by other arithmetic than the ray's way into object space: twice the pad
      for (int k = 0; k < 3; ++k) { refs[i].box.mn[k] -= pad; refs[i].box.mx[k] += pad; }
    // The slab test works on t = plane*idir - o*idir, whose rounding error is a few ulp of the largest coordinate involved — as
    // is the triangle test's.  A hit exactly on a box face can therefore be culled (seen once in 2*10^7 rays on the 1 M-triangle
    // scene, where brute force and the product's quantised boxes kept it).  Padding every box by 2^-19 of the scene's largest
    // coordinate puts such hits safely inside.
    for (int k = 0; k < 3; ++k) { refs[i].box.mn[k] -= pad; refs[i].box.mx[k] += pad; }
    for (int k = 0; k < 3; ++k) refs[i].c[k] = 0.5f * (refs[i].box.mn[k] + refs[i].box.mx[k]);
    refs[i].id = i;
  }
  s->nodes.clear();
  if (N == 0) {
    Node n{}; n.child0 = kAbsent; n.child1 = kAbsent; s->nodes.push_back(n);
  } else {
    s->nodes.reserve(N);
    Builder b(refs, s->nodes);
    ChildRef root = b.build(0, N);
    if (root.count > 0) {  // the whole scene is one leaf: root node with one child
      Node n{};
      memcpy(n.c0min, root.box.mn, 12); memcpy(n.c0max, root.box.mx, 12);
      n.child0 = root.child; n.count0 = root.count; n.child1 = kAbsent; n.count1 = 0;
      s->nodes.push_back(n);
    }
  }
  s->tris.resize(N);
  for (uint32_t i = 0; i < N; ++i) s->tris[i] = s->tris_by_id[refs[i].id];
}

// ---------------------------------------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------------------------------------
// RENDER_SPEC 4.5: world -> object of an instance — rows of the inverse of the upper 3x3 (cross products of its columns over the
// determinant) and the translation; false: not invertible in float (the instance is flattened like one that is referenced once)
static bool world_to_object(const float* m, float* r0, float* r1, float* r2, float* tr) {
  const V3 c0 = v3(m[0], m[1], m[2]), c1 = v3(m[4], m[5], m[6]), c2 = v3(m[8], m[9], m[10]);
  const V3 k0 = cross3(c1, c2), k1 = cross3(c2, c0), k2 = cross3(c0, c1);
  const float det = dot3(c0, k0);
  if (!(det != 0.0f) || !std::isfinite(det)) return false;
  const float inv = 1.0f / det;
  const V3 a = k0 * inv, b = k1 * inv, c = k2 * inv;
  r0[0] = a.x; r0[1] = a.y; r0[2] = a.z; r1[0] = b.x; r1[1] = b.y; r1[2] = b.z; r2[0] = c.x; r2[1] = c.y; r2[2] = c.z;
  for (int k = 0; k < 3; ++k) tr[k] = m[12 + k];
  for (int k = 0; k < 3; ++k) if (!std::isfinite(r0[k]) || !std::isfinite(r1[k]) || !std::isfinite(r2[k]) || !std::isfinite(tr[k])) return false;
  return true;
}
// RENDER_SPEC 4.5 is a per-scene choice (hala_rt_build_options::instancing; hala_bvh_info::instance_ref_count tells what the product chose).
// Default: everything flattened — what the product's automatic mode picks for every scene of up to 2^26 triangles.
static int g_instancing_off = 1;
extern "C" void orc_set_instancing_off(int off) { g_instancing_off = off; }

extern "C" orc_scene* orc_scene_create(const orc_scene_desc* desc) {
  orc_scene* s = new orc_scene();
  std::vector<float> world((size_t)desc->node_count * 16);
  orc_update_node_hierarchies(desc, world.data());
  // own the geometry
  s->owned_vertices.resize(desc->mesh_count ? 0 : 0);
  std::vector<std::vector<std::pair<size_t, size_t>>> slot(desc->mesh_count);
  for (uint32_t m = 0; m < desc->mesh_count; ++m) {
    for (uint32_t p = 0; p < desc->meshes[m].primitive_count; ++p) {
      const orc_primitive_desc& pr = desc->meshes[m].primitives[p];
      s->owned_vertices.emplace_back(pr.vertices, pr.vertices + pr.vertex_count);
      s->owned_indices.emplace_back(pr.indices, pr.indices + pr.index_count);
      slot[m].push_back({s->owned_vertices.size() - 1, s->owned_indices.size() - 1});
    }
  }
  // RENDER_SPEC 4.5: how many instances reference each primitive
  std::vector<std::vector<uint32_t>> prim_refs(desc->mesh_count);
  for (uint32_t m = 0; m < desc->mesh_count; ++m) prim_refs[m].assign(desc->meshes[m].primitive_count, 0u);
  for (uint32_t k = 0; k < desc->node_count; ++k)
    if (desc->nodes[k].mesh_index != ORC_NONE)
      for (uint32_t p = 0; p < desc->meshes[desc->nodes[k].mesh_index].primitive_count; ++p) prim_refs[desc->nodes[k].mesh_index][p]++;
  // RENDER_SPEC §3 / gpu_uploader.rs:843-875: instances in node order, then primitive order
  for (int i = 0; i < 3; ++i) { s->bounds_min[i] = std::numeric_limits<float>::infinity(); s->bounds_max[i] = -s->bounds_min[i]; }
  for (uint32_t k = 0; k < desc->node_count; ++k) {
    const orc_node_desc& node = desc->nodes[k];
    if (node.mesh_index == ORC_NONE) continue;
    const float* w = world.data() + 16 * k;
    for (uint32_t p = 0; p < desc->meshes[node.mesh_index].primitive_count; ++p) {
      const orc_primitive_desc& pr = desc->meshes[node.mesh_index].primitives[p];
      Instance inst;
      memcpy(inst.transform, w, 64);
      inst.material_index = pr.material_index;
      inst.first_triangle = (uint32_t)s->tris_by_id.size();
      inst.vertices = s->owned_vertices[slot[node.mesh_index][p].first].data();
      inst.indices = s->owned_indices[slot[node.mesh_index][p].second].data();
      inst.instanced = !g_instancing_off && prim_refs[node.mesh_index][p] >= 2u && pr.index_count >= 3u && world_to_object(w, inst.r0, inst.r1, inst.r2, inst.tr);
      uint32_t inst_id = (uint32_t)s->instances.size();
      s->instances.push_back(inst);
      float omn[3] = {INFINITY, INFINITY, INFINITY}, omx[3] = {-INFINITY, -INFINITY, -INFINITY};  // instanced: exact object-space bounds of the primitive's triangles
      for (uint32_t t = 0; t < pr.index_count / 3; ++t) {  // primitive_count = index_count / 3 (gpu_uploader.rs:804)
        V3 v[3], lv[3];
        for (int c = 0; c < 3; ++c) {
          const float* pp = inst.vertices[inst.indices[3 * t + c]].position;
          lv[c] = v3(pp[0], pp[1], pp[2]);
          v[c] = transform_point(w, lv[c]);
          const float a[3] = {v[c].x, v[c].y, v[c].z};
          if (!inst.instanced) for (int i = 0; i < 3; ++i) { s->bounds_min[i] = std::min(s->bounds_min[i], a[i]); s->bounds_max[i] = std::max(s->bounds_max[i], a[i]); }
          else for (int i = 0; i < 3; ++i) { omn[i] = std::min(omn[i], pp[i]); omx[i] = std::max(omx[i], pp[i]); }
        }
        Tri tr;
        // RENDER_SPEC 4.5: an instanced instance's triangles stay in object space (local positions as they are)
        V3 e1 = inst.instanced ? lv[1] - lv[0] : v[1] - v[0], e2 = inst.instanced ? lv[2] - lv[0] : v[2] - v[0];
        if (inst.instanced) v[0] = lv[0];
        tr.v0[0] = v[0].x; tr.v0[1] = v[0].y; tr.v0[2] = v[0].z; tr.id = (uint32_t)s->tris_by_id.size();
        tr.e1[0] = e1.x; tr.e1[1] = e1.y; tr.e1[2] = e1.z; tr.pad1 = 0;
        tr.e2[0] = e2.x; tr.e2[1] = e2.y; tr.e2[2] = e2.z; tr.pad2 = 0;
        s->tris_by_id.push_back(tr);
        s->tri_instance.push_back(inst_id);
        for (int c = 0; c < 3; ++c) {
          const V3 wv = inst.instanced ? transform_point(w, lv[c]) : v[c];
          s->tri_verts9.push_back(wv.x); s->tri_verts9.push_back(wv.y); s->tri_verts9.push_back(wv.z);
        }
      }
      if (inst.instanced)  // RENDER_SPEC 4.5: its share of the scene bounds = the box of the eight corners of the primitive's bounds, moved to world space
        for (int c = 0; c < 8; ++c) {
          const V3 q = transform_point(w, v3((c & 1) ? omx[0] : omn[0], (c & 2) ? omx[1] : omn[1], (c & 4) ? omx[2] : omn[2]));
          const float a[3] = {q.x, q.y, q.z};
          for (int i = 0; i < 3; ++i) { s->bounds_min[i] = std::min(s->bounds_min[i], a[i]); s->bounds_max[i] = std::max(s->bounds_max[i], a[i]); }
        }
    }
  }
  if (s->tris_by_id.empty()) for (int i = 0; i < 3; ++i) { s->bounds_min[i] = 0.0f; s->bounds_max[i] = 0.0f; }
  V3 ext = v3(s->bounds_max[0] - s->bounds_min[0], s->bounds_max[1] - s->bounds_min[1], s->bounds_max[2] - s->bounds_min[2]);
  s->ray_eps = sqrtf(dot3(ext, ext)) * 1e-5f;  // RENDER_SPEC §3
  build_bvh(s);
  // textures (RENDER_SPEC §7.4; gpu_uploader.rs:334-403): decode to linear RGBA32F, 2x2 box mips, texture -> image map
  for (uint32_t k = 0; k < desc->image_data_count; ++k) {
    const orc_image_desc& im = desc->image_data[k];
    Image img;
    img.width = im.width; img.height = im.height;
    const size_t n = (size_t)im.width * im.height;
    std::vector<float> l0(n * 4);
    // RENDER_SPEC 7.4: float images are linear RGBA32F texels; 8-bit images STAY 8-bit at every mip level and are decoded by the sampler
    // (sRGB bytes through the 256-entry table, UNORM bytes / 255; alpha always UNORM).  The oracle keeps the decoded values.
    float lut[256], thr[256];
    for (int i = 0; i < 256; ++i) { const double x = i / 255.0; lut[i] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4)); }
    for (int k = 0; k < 255; ++k) thr[k] = (lut[k] + lut[k + 1]) * 0.5f;
    thr[255] = 3.402823466e+38f;
    const bool bytes = im.format != 2, srgb = im.format == 1;
    if (!bytes) memcpy(l0.data(), im.data, n * 16);
    else {
      const uint8_t* p = static_cast<const uint8_t*>(im.data);
      for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 4; ++c) {
          int src = c;
          if (im.format == 3 && c < 3) src = 2 - c;  // RGBA bytes tagged BGRA (cpu/image_data.rs:39-43)
          const uint8_t b = p[4 * i + src];
          l0[4 * i + c] = (srgb && c < 3) ? lut[b] : (float)b / 255.0f;
        }
      }
    }
    // a mip texel of an 8-bit image goes back to a byte: UNORM floor(x * 255 + 0.5) clamped; sRGB the code whose decoded value is
    // nearest = the number of midpoints thr[k] = (lut[k] + lut[k + 1]) / 2 below x (bisection) — and is stored as that byte decodes
    auto requantise = [&](float x, int c) -> float {
      if (srgb && c < 3) {
        uint32_t lo = 0, hi = 255;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (thr[mid] < x) lo = mid + 1u; else hi = mid; }
        return lut[lo];
      }
      float v = floorf(x * 255.0f + 0.5f);
      v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
      return (float)(uint32_t)v / 255.0f;
    };
    for (size_t i = 0; i < n; ++i) img.has_alpha = img.has_alpha || l0[4 * i + 3] < 1.0f;  // RENDER_SPEC 7.1d (NaN: not a cut-out)
    uint32_t m = std::max(im.width, im.height), p2 = 1, lg = 0;
    while (p2 < m) { p2 <<= 1; ++lg; }
    img.mips = std::min<uint32_t>(lg + 1, 16);  // gpu_uploader.rs:366
    img.levels.push_back(std::move(l0));
    for (uint32_t l = 1; l < img.mips; ++l) {
      const uint32_t sw = std::max(1u, im.width >> (l - 1)), sh = std::max(1u, im.height >> (l - 1));
      const uint32_t dw = std::max(1u, im.width >> l), dh = std::max(1u, im.height >> l);
      const std::vector<float>& src = img.levels[l - 1];
      std::vector<float> dst((size_t)dw * dh * 4);
      for (uint32_t y = 0; y < dh; ++y)
        for (uint32_t x = 0; x < dw; ++x) {
          const uint32_t x0 = std::min(2 * x, sw - 1), x1 = std::min(2 * x + 1, sw - 1), y0 = std::min(2 * y, sh - 1), y1 = std::min(2 * y + 1, sh - 1);
          for (int c = 0; c < 4; ++c) {
            const float a = src[((size_t)y0 * sw + x0) * 4 + c], b = src[((size_t)y0 * sw + x1) * 4 + c];
            const float cc = src[((size_t)y1 * sw + x0) * 4 + c], d = src[((size_t)y1 * sw + x1) * 4 + c];
            const float box = ((a + b) + (cc + d)) * 0.25f;
            dst[((size_t)y * dw + x) * 4 + c] = bytes ? requantise(box, c) : box;
          }
        }
      img.levels.push_back(std::move(dst));
    }
    s->images.push_back(std::move(img));
  }
  for (uint32_t i = 0; i < desc->texture_count; ++i) {
    uint32_t data = ORC_NONE;
    for (uint32_t j = 0; j < desc->image_count; ++j)
      if (desc->image2data_mapping[j].key == desc->texture2image_mapping[i].value) { data = desc->image2data_mapping[j].value; break; }
    s->texture_image.push_back(data);
  }
  s->materials.resize(desc->material_count);
  for (uint32_t i = 0; i < desc->material_count; ++i) orc_pack_material(&desc->materials[i], &s->materials[i]);
  orc::make_any_triangles(s, s->tris, &s->tris_any);  // RENDER_SPEC 7.1d (needs the materials)
  s->camera_count = orc_pack_cameras(desc, s->cameras);
  if (s->camera_count < 0) s->camera_count = 0;
  s->light_count = orc_pack_lights(desc, s->lights, std::vector<orc_aabb>(32).data());
  if (s->light_count < 0) s->light_count = 0;
  return s;
}
extern "C" void orc_scene_destroy(orc_scene* s) { delete s; }
extern "C" void orc_scene_set_envmap(orc_scene* s, const float* rgba, uint32_t width, uint32_t height) {
  s->env.width = width; s->env.height = height;
  s->env.pixels.assign(rgba, rgba + (size_t)width * height * 4);
  for (size_t i = 0; i < (size_t)width * height; ++i) s->env.pixels[4 * i + 3] = 1.0f;  // src/envmap.rs:87
  s->env.marginal.resize(height); s->env.conditional.resize((size_t)width * height);
  orc_envmap_build_distribution(s->env.pixels.data(), width, height, &s->env.total_sum, s->env.marginal.data(), s->env.conditional.data());
}
extern "C" uint32_t orc_scene_triangle_count(const orc_scene* s) { return (uint32_t)s->tris_by_id.size(); }
extern "C" uint32_t orc_scene_node_count(const orc_scene* s) { return (uint32_t)s->nodes.size(); }
extern "C" void orc_scene_bounds(const orc_scene* s, float mn[3], float mx[3]) { memcpy(mn, s->bounds_min, 12); memcpy(mx, s->bounds_max, 12); }
extern "C" void orc_scene_get_triangles(const orc_scene* s, float* out9) {
  memcpy(out9, s->tri_verts9.data(), s->tri_verts9.size() * sizeof(float));
}

// ---------------------------------------------------------------------------------------------------------
// RENDER_SPEC §4.2 ray/triangle, §4.3 ray/box, §4.4 traversal order
// ---------------------------------------------------------------------------------------------------------
namespace orc {

struct RayPre { V3 o, d, idir, ood; float tmin; };

static inline float safe_inv(float d) {
  float dd = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
  return 1.0f / dd;
}
static inline RayPre make_ray(V3 o, V3 d, float tmin) {
  RayPre r; r.o = o; r.d = d; r.tmin = maxf(tmin, 0.0f);  // RENDER_SPEC §4.2: rays start at or after their origin
  r.idir = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  r.ood = r.o * r.idir;
  return r;
}
// returns true and the entry distance if the (padded) slab interval is non-empty against [tmin, tlimit]
static inline bool box_test(const RayPre& r, const float* mn, const float* mx, float tlimit, float* tnear) {
  float x0 = fmaf(mn[0], r.idir.x, -r.ood.x), x1 = fmaf(mx[0], r.idir.x, -r.ood.x);
  float y0 = fmaf(mn[1], r.idir.y, -r.ood.y), y1 = fmaf(mx[1], r.idir.y, -r.ood.y);
  float z0 = fmaf(mn[2], r.idir.z, -r.ood.z), z1 = fmaf(mx[2], r.idir.z, -r.ood.z);
  float tn = maxf(maxf(minf(x0, x1), minf(y0, y1)), maxf(minf(z0, z1), r.tmin));
  float tf = minf(minf(maxf(x0, x1), maxf(y0, y1)), minf(maxf(z0, z1), tlimit));
  *tnear = tn;
  return tn <= tf * 1.0000004f;
}
// Möller–Trumbore with the fma placement of spec_math.h. Returns true if (u,v) is inside and fills t,u,v.
static inline bool tri_test(const RayPre& r, const Tri& tr, float* t, float* u, float* v, float* det_out = nullptr) {
  V3 e1 = v3(tr.e1[0], tr.e1[1], tr.e1[2]), e2 = v3(tr.e2[0], tr.e2[1], tr.e2[2]);
  V3 p = cross3(r.d, e2);
  float det = dot3(e1, p);
  if (det_out) *det_out = det;
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  V3 tv = r.o - v3(tr.v0[0], tr.v0[1], tr.v0[2]);
  float uu = dot3(tv, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return false;
  V3 q = cross3(tv, e1);
  float vv = dot3(r.d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
  *t = dot3(e2, q) * inv; *u = uu; *v = vv;
  return true;
}

// RENDER_SPEC 4.5: the ray a triangle is tested with — the world-space ray, or, for a triangle of an instanced instance, the ray moved
// into that instance's object space (t is kept: the direction is not normalised); cached per instance
struct RayCtx {
  const orc_scene* s;
  RayPre world, obj;
  uint32_t cur = 0xffffffffu;
  const RayPre& for_tri(uint32_t gid) {
    if (!s) return world;
    const uint32_t ii = s->tri_instance[gid];
    const Instance& in = s->instances[ii];
    if (!in.instanced) return world;
    if (ii != cur) {
      obj = object_ray(world, in.r0, in.r1, in.r2, in.tr);
      cur = ii;
    }
    return obj;
  }
  static RayPre object_ray(const RayPre& w, const float* r0, const float* r1, const float* r2, const float* tr) {
    const V3 a = v3(r0[0], r0[1], r0[2]), b = v3(r1[0], r1[1], r1[2]), c = v3(r2[0], r2[1], r2[2]);
    const V3 tv = w.o - v3(tr[0], tr[1], tr[2]);
    return make_ray(v3(dot3(a, tv), dot3(b, tv), dot3(c, tv)), v3(dot3(a, w.d), dot3(b, w.d), dot3(c, w.d)), w.tmin);
  }
};

// RENDER_SPEC 7.1d / 7.1g: what a triangle of the any-hit copy does to an any-hit ray that hits it inside (tmin, tmax).  Flags (word 7 of the
// record): 0 blocks; 1 translucent: blocks iff hash(key, triangle) < opacity x alpha at the hit; 2 invisible boundary of a medium: never
// blocks, adds to the optical depth; 3 translucent boundary of a medium: 1, then 2 when the ray gets through.
bool any_hit_event(AnyCtx* ax, const Tri& tr, float t, float det, float u, float v) {
  if (!tr.pad1 || !ax) return true;
  if (tr.pad1 & 1u) {
    const float x = (float)(pcg_hash(ax->key + tr.id * 0x9E3779B1u) >> 8) * (1.0f / 16777216.0f);
    if (x < hit_alpha(ax->s, tr.id, u, v)) return true;
  }
  if (tr.pad1 & 2u) {  // 7.1g: +sigma t where the ray leaves the object (hit from behind: det < 0), -sigma t where it enters
    const orc_gpu_material& m = ax->s->materials[ax->s->instances[ax->s->tri_instance[tr.id]].material_index];
    for (int c = 0; c < 3; ++c) {
      const float sigma = m.medium_type == 1u ? m.medium_density * (1.0f - m.medium_color[c]) : m.medium_density;
      const uint32_t q = (uint32_t)(int32_t)floorf(minf(t * sigma, 4096.0f) * 65536.0f + 0.5f);
      ax->tau[c] = det < 0.0f ? ax->tau[c] + q : ax->tau[c] - q;
    }
  }
  return false;
}
V3 any_transmittance(const AnyCtx& ax) {
  float r[3];
  for (int c = 0; c < 3; ++c) {
    const int32_t q = (int32_t)ax.tau[c];
    r[c] = q > 0 ? exp_neg_poly(-((float)q * (1.0f / 65536.0f))) : 1.0f;
  }
  return v3(r[0], r[1], r[2]);
}
template <bool ANY>
static inline bool traverse(const orc_scene* s, const Node* nodes, const Tri* tris, const RayPre& r, float tmax, AnyCtx* ax, Hit* best, Counters* c) {
  best->t = tmax; best->prim = ORC_NONE; best->u = 0.0f; best->v = 0.0f;
  RayCtx rc{s, r, r};
  uint32_t stack[1024]; int sp = 0;
  uint32_t cur = 0;
  for (;;) {
    const Node& n = nodes[cur];
    if (c) c->nodes++;
    float tn0 = 0.0f, tn1 = 0.0f;
    bool valid0 = !(n.count0 == 0 && n.child0 == kAbsent), valid1 = !(n.count1 == 0 && n.child1 == kAbsent);
    bool h0 = valid0 && box_test(r, n.c0min, n.c0max, best->t, &tn0);
    bool h1 = valid1 && box_test(r, n.c1min, n.c1max, best->t, &tn1);
    // order: nearer entry first, ties -> child 0
    int order[2] = {0, 1};
    if (h0 && h1 && tn1 < tn0) { order[0] = 1; order[1] = 0; }
    uint32_t next = kAbsent;
    for (int k = 0; k < 2; ++k) {
      int ci = order[k];
      bool h = ci == 0 ? h0 : h1;
      if (!h) continue;
      uint32_t child = ci == 0 ? n.child0 : n.child1, count = ci == 0 ? n.count0 : n.count1;
      float tn = ci == 0 ? tn0 : tn1;
      if (count > 0) {
        if (!(tn <= best->t)) continue;  // best may have shrunk since the box test
        if (c) c->tris += count;
        for (uint32_t i = 0; i < count; ++i) {
          const Tri& tr = tris[child + i];
          float t, u, v, det;
          if (!tri_test(rc.for_tri(tr.id), tr, &t, &u, &v, &det)) continue;
          if (ANY) {
            if (t > r.tmin && t < tmax && any_hit_event(ax, tr, t, det, u, v)) { best->t = t; best->prim = tr.id; best->u = u; best->v = v; return true; }
          } else {
            if (t > r.tmin && (t < best->t || (t == best->t && tr.id < best->prim))) { best->t = t; best->u = u; best->v = v; best->prim = tr.id; }
          }
        }
      } else {
        if (next == kAbsent) next = child; else stack[sp++] = child;
      }
    }
    if (next == kAbsent) {
      if (sp == 0) break;
      next = stack[--sp];
    }
    cur = next;
  }
  return best->prim != ORC_NONE;
}

// RENDER_SPEC §4.4b: the same traversal over compressed 4-wide nodes.
static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

template <bool ANY>
static inline bool leaf_test(const Tri* tris, const RayPre& r, float tmax, AnyCtx* ax, Hit* best, uint32_t first, uint32_t count, uint32_t gid_base = 0) {
  for (uint32_t i = 0; i < count; ++i) {
    Tri tr = tris[first + i];
    tr.id += gid_base;  // inside an instance of a two-level tree the triangles carry ids local to the primitive
    float t, u, v, det;
    if (!tri_test(r, tr, &t, &u, &v, &det)) continue;
    if (ANY) {
      if (t > r.tmin && t < tmax && any_hit_event(ax, tr, t, det, u, v)) { best->t = t; best->prim = tr.id; best->u = u; best->v = v; return true; }
    } else if (t > r.tmin && (t < best->t || (t == best->t && tr.id < best->prim))) {
      best->t = t; best->u = u; best->v = v; best->prim = tr.id;
    }
  }
  return false;
}

// IEEE-754 minNum / maxNum with -0 < +0: what v_min_f32 / v_max_f32 compute (RENDER_SPEC §4.3b)
static inline float hw_minf(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return std::signbit(a) ? a : b;
  return a < b ? a : b;
}
static inline float hw_maxf(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return std::signbit(a) ? b : a;
  return a > b ? a : b;
}
static inline float key_tn(uint32_t key) { return bits_f(key & ~3u); }

constexpr size_t kSmallTreeBytes = 40 * 1024;  // RENDER_SPEC 4.4b: node_count * 64 + triangle_count * 48 <= this -> sequential leaf culling
static inline bool is_small_tree(size_t node_count, size_t tri_count) { return node_count * 64 + tri_count * 48 <= kSmallTreeBytes; }
// refs (may be null): the instance references of a two-level tree (RENDER_SPEC 4.5).  An instance leaf (reference bits 31..28 = 0xF)
// sorts and waits like an inner child; entering it moves the ray into the instance's object space and goes on at the root of the
// primitive's tree (whose triangles carry local ids: + gid_base); an exit mark on the stack brings the world-space ray back.
static inline bool is_inst_leaf(uint32_t ref) { return (ref & 0xF0000000u) == 0xF0000000u && ref < 0xfffffffeu; }
template <bool ANY>
static inline bool traverse4(const Node4* nodes, const Tri* tris, bool small_tree, const RayPre& r_world, float tmax, AnyCtx* ax, Hit* best, Counters* c,
                             const InstRef* refs = nullptr) {
  best->t = tmax; best->prim = ORC_NONE; best->u = 0.0f; best->v = 0.0f;
  struct Entry { uint32_t key, ref; };
  constexpr uint32_t kExit = 0xfffffffeu;
  Entry stack[1024]; int sp = 0;
  uint32_t cur = 0;
  RayPre r = r_world;
  uint32_t gid_base = 0;
  const bool slot_order = ANY;
  for (;;) {
    const Node4& n = nodes[cur];
    if (c) c->nodes++;
    const float idir[3] = {r.idir.x, r.idir.y, r.idir.z}, ood[3] = {r.ood.x, r.ood.y, r.ood.z};
    float k[3], adj[3];
    for (int a = 0; a < 3; ++a) {
      k[a] = bits_f(((n.exps >> (8 * a)) & 0xffu) << 23) * idir[a];
      adj[a] = fmaf(n.pmin[a], idir[a], -ood[a]);
    }
    Entry e[4];
    bool innerish[4];
    for (int ci = 0; ci < 4; ++ci) {
      float t0[3], t1[3];
      for (int a = 0; a < 3; ++a) {
        t0[a] = fmaf((float)((n.qlo[a] >> (8 * ci)) & 0xffu), k[a], adj[a]);
        t1[a] = fmaf((float)((n.qhi[a] >> (8 * ci)) & 0xffu), k[a], adj[a]);
      }
      float tn = hw_maxf(hw_maxf(hw_minf(t0[0], t1[0]), hw_minf(t0[1], t1[1])), hw_maxf(hw_minf(t0[2], t1[2]), r.tmin));
      float tf = hw_minf(hw_minf(hw_maxf(t0[0], t1[0]), hw_maxf(t0[1], t1[1])), hw_minf(hw_maxf(t0[2], t1[2]), best->t));
      bool hit = n.ref[ci] != kAbsent && tn <= tf * 1.0000004f;
      e[ci].key = hit ? ((f_bits(tn) & ~3u) | (uint32_t)ci) : 0xffffffffu;
      e[ci].ref = n.ref[ci];
    }
    // RENDER_SPEC 4.4c: an any-hit ray takes the children in slot order — nothing it finds moves its limit, so no order
    // spares it a visit, and a child that passed the slab test is never looked at again; every other ray: nearest first
    if (!slot_order) std::sort(e, e + 4, [](const Entry& a, const Entry& b) { return a.key < b.key; });  // keys of hits are distinct (slot bits)
    for (int i = 0; i < 4; ++i) innerish[i] = !(e[i].ref & 0x80000000u) || (refs && is_inst_leaf(e[i].ref));
    // inner children (and instance leaves): nearest next, the others stacked farthest first with their keys
    uint32_t next = kAbsent, next_key = 0;
    for (int i = 3; i >= 0; --i) {
      if (e[i].key == 0xffffffffu || !innerish[i]) continue;
      if (next != kAbsent) stack[sp++] = Entry{next_key, next};
      next = e[i].ref; next_key = e[i].key;
    }
    // leaves (RENDER_SPEC 4.4b).  Large trees: every leaf in reach AS THE NODE IS ENTERED is tested in full and counted — a hit in one
    // does not cull its siblings (the kernels test them side by side), and an any-hit ray still counts all of them before it stops.
    // Small trees (<= 40 KB: the ones the kernels hold in LDS): nearest first, each leaf only while it is still in reach, an any-hit
    // ray stops at the first accepted triangle.
    const float reach = best->t;
    bool occluded = false;
    for (int i = 0; i < 4; ++i) {
      if (e[i].key == 0xffffffffu) continue;
      uint32_t rf = e[i].ref;
      if (innerish[i]) continue;
      if (!slot_order && !(key_tn(e[i].key) <= (small_tree ? best->t : reach))) continue;  // 4.4c: a child that passed the slab test is not looked at again
      uint32_t count = ((rf >> 28) & 7u) + 1u;
      if (c) c->tris += count;
      if (!occluded && leaf_test<ANY>(tris, r, tmax, ax, best, rf & 0x0fffffffu, count, gid_base)) occluded = true;
      if (occluded && small_tree) return true;
    }
    if (occluded) return true;
    if (!slot_order && next != kAbsent && !(key_tn(next_key) <= best->t)) next = kAbsent;
    for (;;) {
      if (next == kAbsent) {
        if (sp == 0) return best->prim != ORC_NONE;
        Entry t = stack[--sp];
        if (t.ref == kExit) { r = r_world; gid_base = 0; continue; }
        if (slot_order || key_tn(t.key) <= best->t) next = t.ref;
        continue;
      }
      if (refs && is_inst_leaf(next)) {
        const InstRef& ir = refs[next & 0x0fffffffu];
        stack[sp++] = Entry{0u, kExit};
        r = RayCtx::object_ray(r_world, ir.r0, ir.r1, ir.r2, ir.tr);
        gid_base = ir.gid_base;
        next = ir.root;
      }
      break;
    }
    cur = next;
  }
}

Hit trace_closest(const orc_scene* s, const Node* nodes, const Tri* tris, V3 o, V3 d, float tmin, float tmax, Counters* c) {
  RayPre r = make_ray(o, d, tmin);
  Hit h;
  if (!traverse<false>(s, nodes, tris, r, tmax, nullptr, &h, c)) { h.t = -1.0f; h.u = 0.0f; h.v = 0.0f; h.prim = ORC_NONE; }
  return h;
}
bool trace_any(const orc_scene* s, const Node* nodes, const Tri* tris, V3 o, V3 d, float tmin, float tmax, AnyCtx* ax, Counters* c) {
  RayPre r = make_ray(o, d, tmin);
  Hit h;
  return traverse<true>(s, nodes, tris, r, tmax, ax, &h, c);
}
// a tree handed over by the product is "small" (RENDER_SPEC 4.4b) by ITS size; two-level trees are never LDS-staged
static inline bool ext_small(const orc_scene* s) { return s->ext_refs.empty() && is_small_tree(s->ext_nodes.size(), s->ext_tris.size()); }
Hit scene_trace_closest(const orc_scene* s, V3 o, V3 d, float tmin, float tmax, Counters* c) {
  if (s->ext_nodes.empty()) return trace_closest(s, s->nodes.data(), s->tris.data(), o, d, tmin, tmax, c);
  RayPre r = make_ray(o, d, tmin);
  Hit h;
  if (!traverse4<false>(s->ext_nodes.data(), s->ext_tris.data(), ext_small(s), r, tmax, nullptr, &h, c, s->ext_refs.empty() ? nullptr : s->ext_refs.data())) { h.t = -1.0f; h.u = 0.0f; h.v = 0.0f; h.prim = ORC_NONE; }
  return h;
}
// RENDER_SPEC 7.1d / 7.1g: how an any-hit ray treats the triangles of a material
int any_class(const orc_scene* s, uint32_t mi) {
  if (mi >= s->materials.size()) return 0;
  const orc_gpu_material& m = s->materials[mi];
  const bool medium = m.medium_type == 1u || m.medium_type == 2u;
  if (m.opacity == 0.0f) return medium ? 3 : 1;
  const bool cutout = m.base_color_map_index < s->texture_image.size() && s->images[s->texture_image[m.base_color_map_index]].has_alpha;
  if (m.opacity < 1.0f || cutout) return medium ? 4 : 2;
  return 0;
}
bool invisible(const orc_scene* s, uint32_t tri_id) { return any_class(s, s->instances[s->tri_instance[tri_id]].material_index) == 1; }
void make_any_triangles(const orc_scene* s, const std::vector<Tri>& in, std::vector<Tri>* out) {
  out->clear();
  bool some = false;
  for (uint32_t mi = 0; mi < s->materials.size(); ++mi) some = some || any_class(s, mi) != 0;
  if (!some) return;
  *out = in;
  for (Tri& t : *out) {
    const int k = any_class(s, s->instances[s->tri_instance[t.id]].material_index);
    if (k == 1) { t.e1[0] = t.e1[1] = t.e1[2] = 0.0f; t.e2[0] = t.e2[1] = t.e2[2] = 0.0f; }
    t.pad1 = k == 2 ? 1u : (k == 3 ? 2u : (k == 4 ? 3u : 0u));
  }
}
bool scene_trace_any(const orc_scene* s, V3 o, V3 d, float tmin, float tmax, uint32_t key, Counters* c, V3* trans) {
  AnyCtx ax{s, key, {0u, 0u, 0u}};
  bool occ;
  if (s->ext_nodes.empty()) occ = trace_any(s, s->nodes.data(), (s->tris_any.empty() ? s->tris : s->tris_any).data(), o, d, tmin, tmax, &ax, c);
  else {
    RayPre r = make_ray(o, d, tmin);
    Hit h;
    occ = traverse4<true>(s->ext_nodes.data(), (s->ext_tris_any.empty() ? s->ext_tris : s->ext_tris_any).data(), ext_small(s), r, tmax, &ax, &h, c,
                          s->ext_refs.empty() ? nullptr : s->ext_refs.data());
  }
  if (trans) *trans = any_transmittance(ax);
  return occ;
}
}  // namespace orc

// hands the integrator a tree built by the product (64-B compressed 4-wide nodes + 48-B triangles in its order); count 0: back
// to the oracle's own tree.  Results only depend on the tree where a hit lies within rounding error of a box face (~1 ray in
// 10^7 on the 1 M-triangle scene: scripts/hit_mismatch_hunt.py); sharing the tree takes those cases out of image comparisons.
extern "C" void orc_scene_use_bvh4(orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count) {
  s->ext_nodes.assign((const Node4*)nodes64, (const Node4*)nodes64 + node_count);
  s->ext_tris.assign((const Tri*)tris48, (const Tri*)tris48 + (node_count ? tri_count : 0));
  s->ext_refs.clear();
  orc::make_any_triangles(s, s->ext_tris, &s->ext_tris_any);
}
// ... the two-level form (RENDER_SPEC 4.5): + the instance references its instance leaves index.  The stored triangles of an instanced
// primitive carry local ids; their any-hit class is the primitive's (every instance of a primitive has its material)
extern "C" void orc_scene_use_bvh4_two_level(orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                                             const void* refs64, uint32_t ref_count) {
  s->ext_nodes.assign((const Node4*)nodes64, (const Node4*)nodes64 + node_count);
  s->ext_tris.assign((const Tri*)tris48, (const Tri*)tris48 + (node_count ? tri_count : 0));
  s->ext_refs.assign((const InstRef*)refs64, (const InstRef*)refs64 + ref_count);
  // global id of every stored triangle (for the material lookup): world-tree triangles carry it; a primitive's triangles get the ids of
  // the first instance that references them
  std::vector<Tri> tmp = s->ext_tris;
  std::vector<uint8_t> seen(tmp.size(), 0);
  for (const InstRef& ir : s->ext_refs) {
    const Instance& in = s->instances[ir.inst];
    const uint32_t n = (ir.inst + 1 < s->instances.size() ? s->instances[ir.inst + 1].first_triangle : (uint32_t)s->tris_by_id.size()) - in.first_triangle;
    // the primitive's triangles sit at [shade_base, shade_base + n) of the stored array, in the tree's order: ids are local
    for (uint32_t k = 0; k < n && ir.shade_base + k < tmp.size(); ++k)
      if (!seen[ir.shade_base + k]) { tmp[ir.shade_base + k].id += in.first_triangle; seen[ir.shade_base + k] = 1; }
  }
  std::vector<Tri> any;
  orc::make_any_triangles(s, tmp, &any);
  for (size_t k = 0; k < any.size(); ++k) any[k].id = s->ext_tris[k].id;  // back to the stored (local) ids
  s->ext_tris_any = any;
}

static void trace_batch(const orc_scene* s, const Node* nodes, const Tri* tris, const orc_ray* rays, orc_hit* hits, uint32_t count, int mode, uint64_t* counters) {
  uint64_t cn = 0, ct = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : cn, ct)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    const orc_ray& r = rays[i];
    Counters c;
    V3 o = v3(r.origin[0], r.origin[1], r.origin[2]), d = v3(r.direction[0], r.direction[1], r.direction[2]);
    if (mode == 0) {
      Hit h = trace_closest(s, nodes, tris, o, d, r.tmin, r.tmax, &c);
      hits[i].t = h.t; hits[i].u = h.u; hits[i].v = h.v; hits[i].prim = h.prim;
    } else {
      AnyCtx ax{s, pcg_hash((uint32_t)i ^ kAnyKeyBatch), {0u, 0u, 0u}};  // RENDER_SPEC 7.1d: the key of ray i of a batch
      bool occ = trace_any(s, nodes, tris, o, d, r.tmin, r.tmax, &ax, &c);
      hits[i].t = occ ? 1.0f : -1.0f; hits[i].u = 0.0f; hits[i].v = 0.0f; hits[i].prim = ORC_NONE;
    }
    cn += c.nodes; ct += c.tris;
  }
  if (counters) { counters[0] += cn; counters[1] += ct; }
}

extern "C" void orc_trace_rays(const orc_scene* s, const orc_ray* rays, orc_hit* hits, uint32_t count, int mode, uint64_t* counters) {
  trace_batch(s, s->nodes.data(), (mode == 1 && !s->tris_any.empty() ? s->tris_any : s->tris).data(), rays, hits, count, mode, counters);
}
static void trace_rays_on_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count, const InstRef* refs, const orc_ray* rays,
                               orc_hit* hits, uint32_t count, int mode, uint64_t* counters);
extern "C" void orc_trace_rays_on_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count, const orc_ray* rays,
                                       orc_hit* hits, uint32_t count, int mode, uint64_t* counters) {
  trace_rays_on_bvh4(nodes64, node_count, tris48, tri_count, nullptr, rays, hits, count, mode, counters);
}
extern "C" void orc_trace_rays_on_bvh4_two_level(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count, const void* refs64,
                                                 const orc_ray* rays, orc_hit* hits, uint32_t count, int mode, uint64_t* counters) {
  trace_rays_on_bvh4(nodes64, node_count, tris48, tri_count, (const InstRef*)refs64, rays, hits, count, mode, counters);
}
static void trace_rays_on_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count, const InstRef* refs, const orc_ray* rays,
                               orc_hit* hits, uint32_t count, int mode, uint64_t* counters) {
  const Node4* nodes = (const Node4*)nodes64;
  const Tri* tris = (const Tri*)tris48;
  const bool small_tree = !refs && orc::is_small_tree(node_count, tri_count);
  uint64_t cn = 0, ct = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : cn, ct)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    const orc_ray& ry = rays[i];
    Counters c;
    RayPre r = make_ray(v3(ry.origin[0], ry.origin[1], ry.origin[2]), v3(ry.direction[0], ry.direction[1], ry.direction[2]), ry.tmin);
    Hit h;
    if (mode == 0) {
      if (traverse4<false>(nodes, tris, small_tree, r, ry.tmax, nullptr, &h, &c, refs)) hits[i] = orc_hit{h.t, h.u, h.v, h.prim};
      else hits[i] = orc_hit{-1.0f, 0.0f, 0.0f, ORC_NONE};
    } else {
      hits[i] = orc_hit{traverse4<true>(nodes, tris, small_tree, r, ry.tmax, nullptr, &h, &c, refs) ? 1.0f : -1.0f, 0.0f, 0.0f, ORC_NONE};
    }
    cn += c.nodes; ct += c.tris;
  }
  if (counters) { counters[0] += cn; counters[1] += ct; }
}

extern "C" void orc_trace_rays_brute(const orc_scene* s, const orc_ray* rays, orc_hit* hits, uint32_t count, int mode) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    const orc_ray& ry = rays[i];
    RayPre r = make_ray(v3(ry.origin[0], ry.origin[1], ry.origin[2]), v3(ry.direction[0], ry.direction[1], ry.direction[2]), ry.tmin);
    Hit best{ry.tmax, 0.0f, 0.0f, ORC_NONE};
    bool any = false;
    RayCtx rc{s, r, r};
    for (const Tri& tr : s->tris_by_id) {
      float t, u, v;
      if (!tri_test(rc.for_tri(tr.id), tr, &t, &u, &v)) continue;
      if (mode == 1) {  // RENDER_SPEC 7.1d
        const int k = orc::any_class(s, s->instances[s->tri_instance[tr.id]].material_index);
        Tri flagged = tr; flagged.pad1 = k == 2 ? 1u : (k == 3 ? 2u : (k == 4 ? 3u : 0u));
        AnyCtx ax{s, pcg_hash((uint32_t)i ^ kAnyKeyBatch), {0u, 0u, 0u}};
        if (t > r.tmin && t < ry.tmax && k != 1 && any_hit_event(&ax, flagged, t, 1.0f, u, v)) { any = true; break; }
      }
      else if (t > r.tmin && (t < best.t || (t == best.t && tr.id < best.prim))) best = Hit{t, u, v, tr.id};
    }
    if (mode == 1) hits[i] = orc_hit{any ? 1.0f : -1.0f, 0.0f, 0.0f, ORC_NONE};
    else if (best.prim == ORC_NONE) hits[i] = orc_hit{-1.0f, 0.0f, 0.0f, ORC_NONE};
    else hits[i] = orc_hit{best.t, best.u, best.v, best.prim};
  }
}

// ---------------------------------------------------------------------------------------------------------
// structural validation of a product-built BVH (compressed 4-wide nodes): every child's DEQUANTISED box (evaluated in
// double, i.e. exactly) must contain what hangs below it
// ---------------------------------------------------------------------------------------------------------
extern "C" int orc_validate_bvh4(const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                                 const float* ref9, uint32_t* max_depth) {
  const Node4* nodes = (const Node4*)nodes64;
  const Tri* tris = (const Tri*)tris48;
  if (max_depth) *max_depth = 0;
  if (node_count == 0) return 1;
  auto child_box = [&](const Node4& n, int ci, double* mn, double* mx) {
    for (int a = 0; a < 3; ++a) {
      const int e = (int)((n.exps >> (8 * a)) & 0xffu) - 127;
      const double s = std::ldexp(1.0, e);
      mn[a] = (double)n.pmin[a] + (double)((n.qlo[a] >> (8 * ci)) & 0xffu) * s;
      mx[a] = (double)n.pmin[a] + (double)((n.qhi[a] >> (8 * ci)) & 0xffu) * s;
    }
  };
  std::vector<uint8_t> seen_tri(tri_count, 0), seen_node(node_count, 0), seen_slot(tri_count, 0);
  struct Item { uint32_t node, depth; };
  std::vector<Item> st{{0, 1}};
  uint32_t md = 0;
  while (!st.empty()) {
    Item it = st.back(); st.pop_back();
    if (it.node >= node_count) return 2;
    if (seen_node[it.node]) return 3;
    seen_node[it.node] = 1;
    md = std::max(md, it.depth);
    const Node4& n = nodes[it.node];
    for (int ci = 0; ci < 4; ++ci) {
      const uint32_t rf = n.ref[ci];
      if (rf == kAbsent) continue;
      double mn[3], mx[3];
      child_box(n, ci, mn, mx);
      if (!(rf & 0x80000000u)) {
        if (rf >= node_count) return 2;
        const Node4& cn = nodes[rf];
        for (int cj = 0; cj < 4; ++cj) {
          if (cn.ref[cj] == kAbsent) continue;
          double gmn[3], gmx[3];
          child_box(cn, cj, gmn, gmx);
          // a grandchild box is quantised against ITS parent's origin, so it may stick out of this (also quantised) box
          // by less than one quantum of the child node; what must hold exactly is containment of the geometry (below)
          for (int a = 0; a < 3; ++a) {
            const double q = std::ldexp(1.0, (int)((cn.exps >> (8 * a)) & 0xffu) - 127);
            if (gmn[a] < mn[a] - q || gmx[a] > mx[a] + q) return 4;
          }
        }
        st.push_back({rf, it.depth + 1});
      } else {
        const uint32_t first = rf & 0x0fffffffu, count = ((rf >> 28) & 7u) + 1u;
        if ((uint64_t)first + count > tri_count) return 5;
        for (uint32_t i = 0; i < count; ++i) {
          if (seen_slot[first + i]) return 6;
          seen_slot[first + i] = 1;
          const Tri& tr = tris[first + i];
          if (tr.id >= tri_count || seen_tri[tr.id]) return 7;
          seen_tri[tr.id] = 1;
          float v[3][3];
          for (int k = 0; k < 3; ++k) { v[0][k] = tr.v0[k]; v[1][k] = tr.v0[k] + tr.e1[k]; v[2][k] = tr.v0[k] + tr.e2[k]; }
          if (ref9) {
            const float* rv = ref9 + 9 * (size_t)tr.id;
            for (int k = 0; k < 3; ++k) {
              if (tr.v0[k] != rv[k]) return 8;
              if (tr.e1[k] != rv[3 + k] - rv[k]) return 8;
              if (tr.e2[k] != rv[6 + k] - rv[k]) return 8;
            }
          }
          for (int cc = 0; cc < 3; ++cc) {
            const float* pv = ref9 ? ref9 + 9 * (size_t)tr.id + 3 * cc : v[cc];
            for (int k = 0; k < 3; ++k) if ((double)pv[k] < mn[k] || (double)pv[k] > mx[k]) return 9;
          }
        }
      }
    }
  }
  for (uint32_t i = 0; i < tri_count; ++i) if (!seen_tri[i]) return 10;
  if (max_depth) *max_depth = md;
  return 0;
}

// ... of a two-level tree (RENDER_SPEC 4.5).  World-space part (instance levels + the tree over the triangles that are not instanced):
// every child box contains the exact world-space vertices below it — for an instance leaf: all triangles of that instance, moved to world
// space by RENDER_SPEC 3's arithmetic; the stored triangles are the scene's, bit for bit (object space for instanced primitives, ids
// local); every primitive's tree is valid in object space; every global triangle id is reachable exactly once.  0 = valid.
extern "C" int orc_validate_bvh4_two_level(const orc_scene* s, const void* nodes64, uint32_t node_count, const void* tris48, uint32_t tri_count,
                                           const void* refs64, uint32_t ref_count, uint32_t* max_depth) {
  const Node4* nodes = (const Node4*)nodes64;
  const Tri* tris = (const Tri*)tris48;
  const InstRef* refs = (const InstRef*)refs64;
  if (max_depth) *max_depth = 0;
  if (node_count == 0) return 1;
  auto child_box = [&](const Node4& n, int ci, double* mn, double* mx) {
    for (int a = 0; a < 3; ++a) {
      const int e = (int)((n.exps >> (8 * a)) & 0xffu) - 127;
      const double q = std::ldexp(1.0, e);
      mn[a] = (double)n.pmin[a] + (double)((n.qlo[a] >> (8 * ci)) & 0xffu) * q;
      mx[a] = (double)n.pmin[a] + (double)((n.qhi[a] >> (8 * ci)) & 0xffu) * q;
    }
  };
  const uint32_t total = (uint32_t)s->tris_by_id.size();
  std::vector<uint8_t> seen_gid(total, 0), seen_node(node_count, 0), seen_slot(tri_count, 0), seen_ref(ref_count, 0), queued_root(node_count, 0);
  struct Item { uint32_t node, depth; int ref; };    // ref >= 0: inside the tree of instance reference `ref` (object space)
  std::vector<Item> st{{0, 1, -1}};
  uint32_t md = 0;
  auto inst_tris = [&](uint32_t inst) {
    return (inst + 1 < s->instances.size() ? s->instances[inst + 1].first_triangle : total) - s->instances[inst].first_triangle;
  };
  while (!st.empty()) {
    Item it = st.back(); st.pop_back();
    if (it.node >= node_count) return 2;
    if (seen_node[it.node]) return 3;
    seen_node[it.node] = 1;
    md = std::max(md, it.depth);
    const Node4& n = nodes[it.node];
    for (int ci = 0; ci < 4; ++ci) {
      const uint32_t rf = n.ref[ci];
      if (rf == kAbsent) continue;
      double mn[3], mx[3];
      child_box(n, ci, mn, mx);
      if (orc::is_inst_leaf(rf)) {
        if (it.ref >= 0) return 20;  // no instances inside instances
        const uint32_t k = rf & 0x0fffffffu;
        if (k >= ref_count || seen_ref[k]) return 21;
        seen_ref[k] = 1;
        const InstRef& ir = refs[k];
        if (ir.inst >= s->instances.size() || !s->instances[ir.inst].instanced) return 22;
        const Instance& in = s->instances[ir.inst];
        if (ir.gid_base != in.first_triangle || ir.root >= node_count) return 23;
        if (memcmp(ir.r0, in.r0, 12) || memcmp(ir.r1, in.r1, 12) || memcmp(ir.r2, in.r2, 12) || memcmp(ir.tr, in.tr, 12)) return 24;  // RENDER_SPEC 4.5 arithmetic
        const uint32_t cnt = inst_tris(ir.inst);
        for (uint32_t t = 0; t < cnt; ++t) {
          const uint32_t gid = in.first_triangle + t;
          if (seen_gid[gid]) return 7;
          seen_gid[gid] = 1;
          const float* pv = &s->tri_verts9[(size_t)gid * 9];
          for (int cc = 0; cc < 3; ++cc)
            for (int a = 0; a < 3; ++a) if ((double)pv[3 * cc + a] < mn[a] || (double)pv[3 * cc + a] > mx[a]) return 25;
        }
        if (!queued_root[ir.root]) { queued_root[ir.root] = 1; st.push_back({ir.root, it.depth + 1, (int)k}); }  // a primitive's tree is checked once
        continue;
      }
      if (!(rf & 0x80000000u)) {
        if (rf >= node_count) return 2;
        const Node4& cn = nodes[rf];
        for (int cj = 0; cj < 4; ++cj) {
          if (cn.ref[cj] == kAbsent) continue;
          double gmn[3], gmx[3];
          child_box(cn, cj, gmn, gmx);
          for (int a = 0; a < 3; ++a) {
            const double q = std::ldexp(1.0, (int)((cn.exps >> (8 * a)) & 0xffu) - 127);
            if (gmn[a] < mn[a] - q || gmx[a] > mx[a] + q) return 4;
          }
        }
        st.push_back({rf, it.depth + 1, it.ref});
        continue;
      }
      const uint32_t first = rf & 0x0fffffffu, count = ((rf >> 28) & 7u) + 1u;
      if ((uint64_t)first + count > tri_count) return 5;
      for (uint32_t i = 0; i < count; ++i) {
        if (seen_slot[first + i]) return 6;
        seen_slot[first + i] = 1;
        const Tri& tr = tris[first + i];
        uint32_t gid = tr.id;
        const Instance* in = nullptr;
        if (it.ref >= 0) {  // object space: ids are local to the primitive
          const InstRef& ir = refs[it.ref];
          in = &s->instances[ir.inst];
          if (tr.id >= inst_tris(ir.inst)) return 7;
          gid = ir.gid_base + tr.id;
        } else {
          if (gid >= total || seen_gid[gid]) return 7;
          if (s->instances[s->tri_instance[gid]].instanced) return 26;  // an instanced triangle in the world tree
          seen_gid[gid] = 1;
        }
        const Tri& want = s->tris_by_id[gid];
        for (int k2 = 0; k2 < 3; ++k2) if (tr.v0[k2] != want.v0[k2] || tr.e1[k2] != want.e1[k2] || tr.e2[k2] != want.e2[k2]) return 8;
        for (int cc = 0; cc < 3; ++cc) {
          float pv[3];
          if (in) { const float* pp = in->vertices[in->indices[3 * tr.id + cc]].position; pv[0] = pp[0]; pv[1] = pp[1]; pv[2] = pp[2]; }
          else memcpy(pv, &s->tri_verts9[(size_t)gid * 9 + 3 * cc], 12);
          for (int a = 0; a < 3; ++a) if ((double)pv[a] < mn[a] || (double)pv[a] > mx[a]) return 9;
        }
      }
    }
  }
  for (uint32_t i = 0; i < total; ++i) if (!seen_gid[i]) return 10;
  if (max_depth) *max_depth = md;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The oracle's own (binned SAH) tree re-expressed in the product's compressed 4-wide format: greedy top-down collapse
// by surface area, breadth-first numbering, conservative 8-bit quantisation in double (RENDER_SPEC §4.1b).  This is NOT
// a mirror of the product's builder (which is validated through results and structure only); it gives the CPU tier a
// 4-wide tree to pin traverse4 / orc_validate_bvh4 on, and the quality probe a SAH tree in the same format.
// Returns the number of nodes (nodes64_out may be NULL to query); 0 if a leaf holds more than 8 triangles.
// ---------------------------------------------------------------------------------------------------------
extern "C" uint32_t orc_scene_export_bvh4(const orc_scene* s, void* nodes64_out, uint32_t capacity) {
  struct Child { Box box; uint32_t child, count; };  // count > 0: leaf [child, child + count); else binary node index
  auto children_of = [&](uint32_t bi, Child* out) {
    const Node& n = s->nodes[bi];
    int k = 0;
    if (!(n.count0 == 0 && n.child0 == kAbsent)) { memcpy(out[k].box.mn, n.c0min, 12); memcpy(out[k].box.mx, n.c0max, 12); out[k].child = n.child0; out[k].count = n.count0; ++k; }
    if (!(n.count1 == 0 && n.child1 == kAbsent)) { memcpy(out[k].box.mn, n.c1min, 12); memcpy(out[k].box.mx, n.c1max, 12); out[k].child = n.child1; out[k].count = n.count1; ++k; }
    return k;
  };
  std::vector<uint32_t> order{0};           // binary root of every 4-node, breadth-first
  std::vector<std::array<Child, 4>> kids;   // its children
  std::vector<int> nkids;
  for (size_t head = 0; head < order.size(); ++head) {
    std::array<Child, 4> c{};
    int n = children_of(order[head], c.data());
    while (n > 0 && n < 4) {
      int pick = -1; float best = -1.0f;
      for (int k = 0; k < n; ++k) if (c[k].count == 0 && c[k].box.half_area() > best) { best = c[k].box.half_area(); pick = k; }
      if (pick < 0) break;
      Child sub[2];
      int m = children_of(c[pick].child, sub);
      if (m == 2) { for (int k = n; k > pick + 1; --k) c[k] = c[k - 1]; c[pick] = sub[0]; c[pick + 1] = sub[1]; ++n; }
      else if (m == 1) c[pick] = sub[0];
      else break;
    }
    for (int k = 0; k < n; ++k) {
      if (c[k].count > 8) return 0;
      if (c[k].count == 0) { order.push_back(c[k].child); c[k].child = (uint32_t)order.size() - 1; }  // now a 4-node index
    }
    kids.push_back(c); nkids.push_back(n);
  }
  const uint32_t total = (uint32_t)order.size();
  if (!nodes64_out) return total;
  if (capacity < total) return 0;
  Node4* out = (Node4*)nodes64_out;
  for (uint32_t i = 0; i < total; ++i) {
    Node4 nd{};
    const int n = nkids[i];
    Box all; all.reset();
    for (int k = 0; k < n; ++k) all.grow(kids[i][k].box);
    if (n == 0) { for (int a = 0; a < 3; ++a) { all.mn[a] = 0.0f; all.mx[a] = 0.0f; } }
    int e[3];
    for (int a = 0; a < 3; ++a) {
      nd.pmin[a] = all.mn[a];
      const double q = ((double)all.mx[a] - (double)all.mn[a]) / 255.0;
      int ex = -100;
      if (q > 0.0) { (void)std::frexp(q, &ex); ex = std::min(100, std::max(-100, ex)); }  // q = m * 2^ex, m in [0.5, 1)
      e[a] = ex;
      nd.exps |= (uint32_t)(ex + 127) << (8 * a);
    }
    for (int k = 0; k < 4; ++k) {
      if (k >= n) { nd.ref[k] = kAbsent; continue; }
      const Child& ch = kids[i][k];
      nd.ref[k] = ch.count ? (0x80000000u | ((ch.count - 1u) << 28) | ch.child) : ch.child;
      for (int a = 0; a < 3; ++a) {
        const double sc = std::ldexp(1.0, e[a]), base = (double)all.mn[a];
        double lo = std::floor(((double)ch.box.mn[a] - base) / sc), hi = std::ceil(((double)ch.box.mx[a] - base) / sc);
        if (base + lo * sc > (double)ch.box.mn[a]) lo -= 1.0;
        if (base + hi * sc < (double)ch.box.mx[a]) hi += 1.0;
        lo = std::min(255.0, std::max(0.0, lo)); hi = std::min(255.0, std::max(0.0, hi));
        nd.qlo[a] |= (uint32_t)lo << (8 * k);
        nd.qhi[a] |= (uint32_t)hi << (8 * k);
      }
    }
    out[i] = nd;
  }
  return total;
}
// the oracle's triangles in ITS BVH order (48-B records), i.e. what the leaves of orc_scene_export_bvh4 index
extern "C" void orc_scene_get_bvh_triangles(const orc_scene* s, void* tris48_out) {
  memcpy(tris48_out, s->tris.data(), s->tris.size() * sizeof(Tri));
}
