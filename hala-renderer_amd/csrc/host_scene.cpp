// host_scene.cpp — host half of HalaSceneGPUUploader::upload (src/scene/loader/gpu_uploader.rs:63-545, :774-967)
// and of cpu::HalaScene's hierarchy update (src/scene/cpu/scene.rs:99-114): turns the borrowed hala_scene_desc into
// the packed records the kernels read.  Pure host arithmetic, float32, evaluated in the reference's order
// (compiled with -ffp-contract=off: Rust never fuses).
#include "host_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt {

const float* srgb_decode_lut() {  // sRGB EOTF (the *_SRGB formats of gltf_loader.rs:395-396 are decoded by the sampler in the reference)
  static float lut[256];
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 256; ++i) {
      const double c = i / 255.0;
      lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    ready = true;
  }
  return lut;
}

// glam::Mat4 * Mat4 (column-major): column c of the product = ((A.x*b.x + A.y*b.y) + A.z*b.z) + A.w*b.w
static Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int c = 0; c < 4; ++c) {
    for (int row = 0; row < 4; ++row) {
      float acc = a.m[0 + row] * b.m[4 * c + 0];
      acc = acc + a.m[4 + row] * b.m[4 * c + 1];
      acc = acc + a.m[8 + row] * b.m[4 * c + 2];
      acc = acc + a.m[12 + row] * b.m[4 * c + 3];
      r.m[4 * c + row] = acc;
    }
  }
  return r;
}

void HostScene::update_node_hierarchies() {
  // src/scene/cpu/scene.rs:99-114 — single pass in node order, parent's world transform must already be final
  const Mat4 identity = Mat4::identity();
  for (auto& n : nodes) n.world = identity;
  for (auto& n : nodes) {
    if (n.parent >= 0) n.world = mul(nodes[(size_t)n.parent].world, n.local);
    else n.world = n.local;
  }
}

std::string HostScene::assign(const hala_scene_desc* d) {
  if (!d) return "The scene is null!";
  nodes.clear(); prims.clear(); mesh_first_prim.clear(); materials.clear(); lights_cpu.clear(); cameras_cpu.clear();
  nodes.reserve(d->node_count);
  for (uint32_t i = 0; i < d->node_count; ++i) {
    const hala_node_desc& nd = d->nodes[i];
    HostNode n;
    n.parent = nd.parent;
    if (nd.parent >= (int32_t)i) return "Node hierarchy is not in parent-before-child order (src/scene/cpu/scene.rs:102-109).";
    memcpy(n.local.m, nd.local_transform, 64);
    n.mesh_index = nd.mesh_index; n.camera_index = nd.camera_index; n.light_index = nd.light_index;
    if (n.mesh_index != HALA_INVALID_INDEX && n.mesh_index >= d->mesh_count) return "Node references a mesh that does not exist.";
    if (n.light_index != HALA_INVALID_INDEX && n.light_index >= d->light_count) return "Node references a light that does not exist.";
    nodes.push_back(n);
  }
  for (uint32_t m = 0; m < d->mesh_count; ++m) {
    mesh_first_prim.push_back((uint32_t)prims.size());
    for (uint32_t p = 0; p < d->meshes[m].primitive_count; ++p) {
      const hala_primitive_desc& pd = d->meshes[m].primitives[p];
      HostPrimitive hp;
      hp.vertices.assign(pd.vertices, pd.vertices + pd.vertex_count);
      hp.indices.assign(pd.indices, pd.indices + pd.index_count);
      for (uint32_t idx : hp.indices) if (idx >= pd.vertex_count) return "Primitive index out of range.";
      // Vulkan makes a triangle with a non-finite position inactive; here it would poison the scene bounds every box is quantised
      // against (and the SAH costs of the build): refused, like the other inputs the reference would pass to the driver unchecked
      for (const hala_vertex& v : hp.vertices)
        if (!std::isfinite(v.position[0]) || !std::isfinite(v.position[1]) || !std::isfinite(v.position[2])) return "Vertex position is not finite.";
      hp.material_index = pd.material_index;
      // primitives without a material carry u32::MAX and the reference passes it through unchecked
      // (gltf_loader.rs:298, gpu_uploader.rs:868); a device-side out-of-bounds read is not acceptable here.
      if (hp.material_index >= d->material_count) return "Primitive references a material that does not exist.";
      prims.push_back(std::move(hp));
    }
  }
  mesh_first_prim.push_back((uint32_t)prims.size());
  materials.assign(d->materials, d->materials + d->material_count);
  for (const auto& m : materials) {
    if (m.type > 1u) return "Invalid material type.";      // HalaMaterialType::from_u8 panics (cpu/material.rs:14)
    if (m.medium_type > 3u) return "Invalid medium type.";  // HalaMediumType::from_u8 panics (cpu/material.rs:65)
  }
  // textures: gpu_uploader.rs:334-403.  textures[i] = image2data[texture2image[i-th key]]; every image_data entry
  // becomes one linear RGBA32F level-0 image (8-bit data decoded here once; the mip chain is built on the GPU).
  images.clear(); texture_image.clear();
  {
    for (uint32_t k = 0; k < d->image_data_count; ++k) {
      const hala_image_desc& im = d->image_data[k];
      if (!im.data || !im.width || !im.height) return "The image data is empty.";
      HostImage32F out;
      out.width = im.width; out.height = im.height;
      const size_t n = (size_t)im.width * im.height;
      if (im.format == HALA_FORMAT_R32G32B32A32_SFLOAT) {
        if (im.num_of_bytes < n * 16) return "The image data is too small.";
        out.format = kTexFloat;
        out.rgba.resize(n * 4);
        memcpy(out.rgba.data(), im.data, n * 16);
        for (size_t i = 0; i < n && !out.has_alpha; ++i) out.has_alpha = out.rgba[4 * i + 3] < 1.0f;
      } else if (im.format == HALA_FORMAT_R8G8B8A8_UNORM || im.format == HALA_FORMAT_R8G8B8A8_SRGB || im.format == HALA_FORMAT_B8G8R8A8_UNORM) {
        if (im.num_of_bytes < n * 4) return "The image data is too small.";
        out.format = im.format == HALA_FORMAT_R8G8B8A8_SRGB ? kTexSrgb8 : kTexUnorm8;
        out.rgba8.assign(static_cast<const uint8_t*>(im.data), static_cast<const uint8_t*>(im.data) + n * 4);
        if (im.format == HALA_FORMAT_B8G8R8A8_UNORM)  // RGBA bytes tagged BGRA without a swizzle: red and blue swap (cpu/image_data.rs:39-43)
          for (size_t i = 0; i < n; ++i) std::swap(out.rgba8[4 * i], out.rgba8[4 * i + 2]);
        for (size_t i = 0; i < n && !out.has_alpha; ++i) out.has_alpha = out.rgba8[4 * i + 3] < 255u;
      } else return "Unsupported image format.";
      images.push_back(std::move(out));
    }
    uint32_t prev_key = 0;
    for (uint32_t i = 0; i < d->texture_count; ++i) {
      const hala_index_pair& t = d->texture2image_mapping[i];
      if (i > 0 && t.key <= prev_key) return "texture2image_mapping is not in ascending key order (BTreeMap).";
      prev_key = t.key;
      const hala_index_pair* hit = nullptr;
      for (uint32_t j = 0; j < d->image_count; ++j) if (d->image2data_mapping[j].key == t.value) { hit = &d->image2data_mapping[j]; break; }
      if (!hit) return "The image " + std::to_string(t.value) + " is not found.";  // gpu_uploader.rs:337
      if (hit->value >= images.size()) return "The image data " + std::to_string(hit->value) + " is not found.";
      texture_image.push_back(hit->value);
    }
  }
  lights_cpu.assign(d->lights, d->lights + d->light_count);
  for (const auto& l : lights_cpu) if (l.light_type > 4u) return "Invalid light type.";  // cpu/light.rs:20
  cameras_cpu.assign(d->cameras, d->cameras + d->camera_count);
  update_node_hierarchies();
  return pack();
}

hala_gpu_material HostScene::pack_material(const hala_material_desc& m) {
  // src/scene/gpu/material.rs:51-110
  hala_gpu_material o;
  memset(&o, 0, sizeof(o));
  float roughness, ax, ay;
  if (m.type == 0u) {  // DIFFUSE: Oren–Nayar A/B (:53-60)
    const float sigma = m.roughness * 0.5f * 1.57079632679489661923f;
    const float sigma2 = sigma * sigma;
    roughness = m.roughness;
    ax = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    ay = 0.45f * sigma2 / (sigma2 + 0.09f);
  } else {  // (:61-69)
    roughness = m.roughness * m.roughness;
    const float clamped = std::min(std::max(m.anisotropic, 0.0f), 1.0f);
    const float aspect = std::sqrt(1.0f - clamped * 0.9f);
    ax = std::max(0.001f, roughness / aspect);
    ay = std::max(0.001f, roughness * aspect);
  }
  memcpy(o.medium_color, m.medium_color, 12);
  o.medium_density = m.medium_density; o.medium_anisotropy = m.medium_anisotropy; o.medium_type = m.medium_type;
  memcpy(o.base_color, m.base_color, 12); o.opacity = m.opacity;
  memcpy(o.emission, m.emission, 12); o.anisotropic = m.anisotropic;
  o.metallic = m.metallic; o.roughness = roughness; o.subsurface = m.subsurface; o.specular_tint = m.specular_tint;
  o.sheen = m.sheen; o.sheen_tint = m.sheen_tint; o.clearcoat = m.clearcoat; o.clearcoat_roughness = m.clearcoat_roughness;
  memcpy(o.clearcoat_tint, m.clearcoat_tint, 12); o.specular_transmission = m.specular_transmission;
  o.ior = m.ior; o.ax = ax; o.ay = ay;
  o.base_color_map_index = m.base_color_map_index; o.normal_map_index = m.normal_map_index;
  o.metallic_roughness_map_index = m.metallic_roughness_map_index; o.emission_map_index = m.emission_map_index;
  o.type = m.type;
  return o;
}

std::string HostScene::pack() {
  // ---- cameras: gpu_uploader.rs:99-122 + gpu/camera.rs:28-61 --------------------------------------------
  cameras.clear();
  for (size_t index = 0; index < cameras_cpu.size(); ++index) {
    if (index >= HALA_MAX_CAMERA_COUNT) break;
    const HostNode* cn = nullptr;
    for (const auto& n : nodes) if (n.camera_index == (uint32_t)index) { cn = &n; break; }
    if (!cn) return "The camera node of the camera " + std::to_string(index) + " is not found.";
    const hala_camera_desc& c = cameras_cpu[index];
    hala_gpu_camera g;
    memset(&g, 0, sizeof(g));
    const float* w = cn->world.m;
    for (int i = 0; i < 3; ++i) { g.position[i] = w[12 + i]; g.right[i] = w[i]; g.up[i] = w[4 + i]; g.forward[i] = -w[8 + i]; }
    if (c.type == 0u) { g.yfov = c.yfov; g.focal_distance_or_xmag = c.focal_distance; g.aperture_or_ymag = c.aperture; g.type = 0; }
    else { g.yfov = 0.0f; g.focal_distance_or_xmag = c.xmag; g.aperture_or_ymag = c.ymag; g.type = 1; }
    cameras.push_back(g);
  }
  // ---- lights: gpu_uploader.rs:148-293 (iterates nodes; order = node order) -------------------------------
  lights.clear(); light_aabbs.clear();
  for (const auto& node : nodes) {
    if (node.light_index == HALA_INVALID_INDEX) continue;
    const hala_light_desc& l = lights_cpu[node.light_index];
    const float* X = node.world.m; const float* Y = X + 4; const float* Z = X + 8; const float* P = X + 12;
    hala_gpu_light g;
    memset(&g, 0, sizeof(g));
    hala_aabb bb{};
    for (int i = 0; i < 3; ++i) g.intensity[i] = l.color[i] * l.intensity;
    g.type = l.light_type;
    switch (l.light_type) {
      case 0:
        for (int i = 0; i < 3; ++i) { g.position[i] = P[i]; bb.min[i] = P[i]; bb.max[i] = P[i]; }
        break;
      case 1:
        for (int i = 0; i < 3; ++i) g.u[i] = -Z[i];
        g.v[0] = std::cos(0.5f * l.param0);
        break;
      case 2:
        for (int i = 0; i < 3; ++i) { g.position[i] = P[i]; g.u[i] = -Z[i]; bb.min[i] = P[i]; bb.max[i] = P[i]; }
        g.v[0] = std::cos(l.param0);
        g.v[1] = std::cos(l.param1);
        break;
      case 3:
        for (int i = 0; i < 3; ++i) {
          float p = P[i];
          p -= X[i] * l.param0 * 0.5f;
          p -= Y[i] * l.param1 * 0.5f;
          const float another = p + X[i] * l.param0 + Y[i] * l.param1 + Z[i] * 0.01f;
          g.position[i] = p;
          g.u[i] = X[i] * l.param0;
          g.v[i] = Y[i] * l.param1;
          bb.min[i] = p; bb.max[i] = another;
        }
        g.area = l.param0 * l.param1;
        break;
      default:  // 4 SPHERE
        for (int i = 0; i < 3; ++i) { g.position[i] = P[i]; bb.min[i] = P[i] - l.param0; bb.max[i] = P[i] + l.param0; }
        g.radius = l.param0;
        g.area = 4.0f * 3.14159265358979323846f * l.param0 * l.param0;
        break;
    }
    hala_aabb sorted;
    for (int i = 0; i < 3; ++i) { sorted.min[i] = std::min(bb.min[i], bb.max[i]); sorted.max[i] = std::max(bb.min[i], bb.max[i]); }
    lights.push_back(g);
    light_aabbs.push_back(sorted);
    if (lights.size() >= HALA_MAX_LIGHT_COUNT) break;
  }
  // ---- materials: gpu_uploader.rs:306-331 ---------------------------------------------------------------------
  gpu_materials.clear();
  for (const auto& m : materials) gpu_materials.push_back(pack_material(m));
  // ---- instances: gpu_uploader.rs:843-875 (node order, then primitive order) ---------------------------------
  instances.clear(); instance_3x4.clear(); inst_first_tri.clear(); instance_node.clear(); instance_prim.clear();
  uint32_t tri_total = 0;
  for (size_t k = 0; k < nodes.size(); ++k) {
    const HostNode& node = nodes[k];
    if (node.mesh_index == HALA_INVALID_INDEX) continue;
    for (uint32_t p = mesh_first_prim[node.mesh_index]; p < mesh_first_prim[node.mesh_index + 1]; ++p) {
      hala_gpu_mesh_data md;
      memset(&md, 0, sizeof(md));
      memcpy(md.transform, node.world.m, 64);
      md.material_index = prims[p].material_index;
      instances.push_back(md);
      const float* w = node.world.m;
      for (int r = 0; r < 3; ++r) { instance_3x4.push_back(w[r]); instance_3x4.push_back(w[4 + r]); instance_3x4.push_back(w[8 + r]); instance_3x4.push_back(w[12 + r]); }
      inst_first_tri.push_back(tri_total);
      instance_node.push_back((uint32_t)k);
      instance_prim.push_back(p);
      const uint64_t nt = prims[p].indices.size() / 3;  // primitive_count = index_count / 3 (:804)
      if (tri_total + nt > 0xfffffff0ull) return "The scene has too many triangles.";
      tri_total += (uint32_t)nt;
    }
  }
  inst_first_tri.push_back(tri_total);
  triangle_count = tri_total;
  return "";
}

void HostScene::primitive_bounds(const HostPrimitive& p, float center[3], float extents[3]) {
  // gpu_uploader.rs:460-467 + src/scene/bounds.rs:46-108 (encapsulate_point per vertex, drift and all)
  for (int i = 0; i < 3; ++i) { center[i] = p.vertices.empty() ? 0.0f : p.vertices[0].position[i]; extents[i] = 0.0f; }
  for (const auto& v : p.vertices) {
    float mn[3], mx[3];
    for (int i = 0; i < 3; ++i) {
      mn[i] = std::min(center[i] - extents[i], v.position[i]);
      mx[i] = std::max(center[i] + extents[i], v.position[i]);
    }
    for (int i = 0; i < 3; ++i) { extents[i] = (mx[i] - mn[i]) * 0.5f; center[i] = mn[i] + extents[i]; }
  }
}

}  // namespace rt
