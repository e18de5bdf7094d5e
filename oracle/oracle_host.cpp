// ORACLE — TEST INFRASTRUCTURE ONLY (see spec_math.h).
// oracle_host.cpp — line-by-line CPU restatement of the host-side arithmetic of the reference's rt_renderer path:
// env-map distribution tables, cpu->gpu record packing, node hierarchy, instance list, tonemap and PFM writer.
// Each function cites the reference lines it follows.  Compile with -ffp-contract=off (Rust never fuses).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "oracle_api.h"
#include "spec_math.h"

using namespace orc;

// ------------------------------------------------------------------------------------------------------------
// A1 — src/envmap.rs:239-388
// ------------------------------------------------------------------------------------------------------------
static inline float env_luminance(float r, float g, float b) {
  // src/envmap.rs:249-251
  return 0.212671f * r + 0.715160f * g + 0.072169f * b;
}
static size_t lower_bound_f32(const float* array, size_t lower, size_t upper, float value) {
  // src/envmap.rs:252-265
  while (lower < upper) {
    size_t mid = (lower + upper) / 2;
    if (array[mid] < value) lower = mid + 1; else upper = mid;
  }
  return lower;
}

extern "C" void orc_envmap_build_distribution(const float* rgba, uint32_t width, uint32_t height, float* total_sum,
                                              float* marginal, float* conditional) {
  const size_t W = width, H = height;
  std::vector<float> cdf_2d(W * H), pdf_1d(H), cdf_1d(H);
  // :275 — sequential fold over all pixels, row-major
  float total = 0.0f;
  for (size_t i = 0; i < W * H; ++i) total = total + env_luminance(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]);
  *total_sum = total;
  // :277-299 — per row running sum, then divide by the row sum (0/0 = NaN for an all-black row, unguarded)
  for (size_t v = 0; v < H; ++v) {
    float row_weight_sum = 0.0f;
    for (size_t u = 0; u < W; ++u) {
      const float* p = rgba + 4 * (v * W + u);
      float weight = env_luminance(p[0], p[1], p[2]);
      row_weight_sum += weight;
      cdf_2d[v * W + u] = row_weight_sum;
    }
    for (size_t u = 0; u < W; ++u) cdf_2d[v * W + u] /= row_weight_sum;
    pdf_1d[v] = row_weight_sum;
  }
  // :300-308
  float col_weight_sum = 0.0f;
  for (size_t v = 0; v < H; ++v) {
    col_weight_sum = col_weight_sum + pdf_1d[v];
    cdf_1d[v] = col_weight_sum;
  }
  col_weight_sum = cdf_1d[H - 1];
  for (size_t v = 0; v < H; ++v) cdf_1d[v] /= col_weight_sum;
  // :311-319
  for (size_t v = 0; v < H; ++v) {
    float inv_height = 1.0f / (float)H;
    size_t row = lower_bound_f32(cdf_1d.data(), 0, H, (float)(v + 1) * inv_height);
    marginal[v] = (float)row * inv_height;
  }
  // :321-331
  for (size_t v = 0; v < H; ++v) {
    for (size_t u = 0; u < W; ++u) {
      float inv_width = 1.0f / (float)W;
      size_t col = lower_bound_f32(cdf_2d.data(), v * W, (v + 1) * W, (float)(u + 1) * inv_width) - v * W;
      conditional[v * W + u] = (float)col * inv_width;
    }
  }
}

extern "C" int orc_envmap_validate(const float* pixels, uint32_t channels, uint32_t width, uint32_t height) {
  // src/envmap.rs:63-89 — per pixel r, g, b in order; NaN test before infinity test
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    for (uint32_t c = 0; c < 3; ++c) {
      float v = pixels[i * channels + c];
      if (std::isnan(v)) return 1;
      if (std::isinf(v)) return 2;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// glam::Mat4 * Mat4 (column-major): each result column = ((A.c0*b.x + A.c1*b.y) + A.c2*b.z) + A.c3*b.w
// ------------------------------------------------------------------------------------------------------------
static void mat4_mul(const float* a, const float* b, float* out) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 4; ++i) {
      float acc = a[0 * 4 + i] * b[c * 4 + 0];
      acc = acc + a[1 * 4 + i] * b[c * 4 + 1];
      acc = acc + a[2 * 4 + i] * b[c * 4 + 2];
      acc = acc + a[3 * 4 + i] * b[c * 4 + 3];
      r[c * 4 + i] = acc;
    }
  memcpy(out, r, sizeof(r));
}

extern "C" void orc_update_node_hierarchies(const orc_scene_desc* scene, float* world) {
  // src/scene/cpu/scene.rs:99-114 — one pass in node order; parents precede children (BFS order of the loader)
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (uint32_t i = 0; i < scene->node_count; ++i) memcpy(world + 16 * i, ident, sizeof(ident));
  for (uint32_t i = 0; i < scene->node_count; ++i) {
    const orc_node_desc& n = scene->nodes[i];
    if (n.parent >= 0) mat4_mul(world + 16 * n.parent, n.local_transform, world + 16 * i);
    else memcpy(world + 16 * i, n.local_transform, sizeof(ident));
  }
}

extern "C" void orc_pack_material(const orc_material_desc* m, orc_gpu_material* o) {
  // src/scene/gpu/material.rs:51-110
  memset(o, 0, sizeof(*o));
  float roughness, ax, ay;
  if (m->type == 0) {  // DIFFUSE :53-60
    float sigma = m->roughness * 0.5f * 1.57079632679489661923f;
    float sigma2 = sigma * sigma;
    roughness = m->roughness;
    ax = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    ay = 0.45f * sigma2 / (sigma2 + 0.09f);
  } else {  // :61-69
    roughness = m->roughness * m->roughness;
    float aniso = m->anisotropic;
    aniso = aniso < 0.0f ? 0.0f : (aniso > 1.0f ? 1.0f : aniso);
    float aspect = sqrtf(1.0f - aniso * 0.9f);
    ax = std::max(0.001f, roughness / aspect);
    ay = std::max(0.001f, roughness * aspect);
  }
  memcpy(o->medium_color, m->medium_color, 12);
  o->medium_density = m->medium_density;
  o->medium_anisotropy = m->medium_anisotropy;
  o->medium_type = m->medium_type;
  memcpy(o->base_color, m->base_color, 12);
  o->opacity = m->opacity;
  memcpy(o->emission, m->emission, 12);
  o->anisotropic = m->anisotropic;
  o->metallic = m->metallic;
  o->roughness = roughness;
  o->subsurface = m->subsurface;
  o->specular_tint = m->specular_tint;
  o->sheen = m->sheen;
  o->sheen_tint = m->sheen_tint;
  o->clearcoat = m->clearcoat;
  o->clearcoat_roughness = m->clearcoat_roughness;
  memcpy(o->clearcoat_tint, m->clearcoat_tint, 12);
  o->specular_transmission = m->specular_transmission;
  o->ior = m->ior;
  o->ax = ax;
  o->ay = ay;
  o->base_color_map_index = m->base_color_map_index;
  o->normal_map_index = m->normal_map_index;
  o->metallic_roughness_map_index = m->metallic_roughness_map_index;
  o->emission_map_index = m->emission_map_index;
  o->type = m->type;
}

extern "C" int orc_pack_cameras(const orc_scene_desc* scene, orc_gpu_camera* out) {
  // src/scene/loader/gpu_uploader.rs:105-117 + src/scene/gpu/camera.rs:28-61
  std::vector<float> world((size_t)scene->node_count * 16);
  orc_update_node_hierarchies(scene, world.data());
  int n = 0;
  for (uint32_t index = 0; index < scene->camera_count; ++index) {
    if (index >= 8) break;  // MAX_CAMERA_COUNT :39, :109-111
    const float* w = nullptr;
    for (uint32_t k = 0; k < scene->node_count; ++k)
      if (scene->nodes[k].camera_index == index) { w = world.data() + 16 * k; break; }  // :112 first match
    if (!w) return -1;  // :113 "The camera node of the camera {} is not found."
    const orc_camera_desc& c = scene->cameras[index];
    orc_gpu_camera g;
    memset(&g, 0, sizeof(g));
    for (int i = 0; i < 3; ++i) {
      g.position[i] = w[12 + i];  // w_axis :29
      g.right[i] = w[0 + i];      // x_axis :30
      g.up[i] = w[4 + i];         // y_axis :31
      g.forward[i] = -w[8 + i];   // -z_axis :32
    }
    if (c.type == 0) { g.yfov = c.yfov; g.focal_distance_or_xmag = c.focal_distance; g.aperture_or_ymag = c.aperture; g.type = 0; }
    else { g.yfov = 0.0f; g.focal_distance_or_xmag = c.xmag; g.aperture_or_ymag = c.ymag; g.type = 1; }
    out[n++] = g;
  }
  return n;
}

extern "C" int orc_pack_lights(const orc_scene_desc* scene, orc_gpu_light* out, orc_aabb* aabbs) {
  // src/scene/loader/gpu_uploader.rs:148-293 — iterates NODES; a light referenced by two nodes appears twice
  std::vector<float> world((size_t)scene->node_count * 16);
  orc_update_node_hierarchies(scene, world.data());
  int n = 0;
  for (uint32_t k = 0; k < scene->node_count; ++k) {
    const orc_node_desc& node = scene->nodes[k];
    if (node.light_index == ORC_NONE) continue;
    const orc_light_desc& l = scene->lights[node.light_index];
    const float* w = world.data() + 16 * k;
    const float* X = w + 0; const float* Y = w + 4; const float* Z = w + 8; const float* Wp = w + 12;
    orc_gpu_light g;
    memset(&g, 0, sizeof(g));
    orc_aabb bb;
    for (int i = 0; i < 3; ++i) g.intensity[i] = l.color[i] * l.intensity;
    switch (l.light_type) {
      case 0:  // POINT :158-182
        for (int i = 0; i < 3; ++i) { g.position[i] = Wp[i]; bb.min[i] = Wp[i]; bb.max[i] = Wp[i]; }
        g.type = 0;
        break;
      case 1:  // DIRECTIONAL :183-199
        for (int i = 0; i < 3; ++i) { g.u[i] = -Z[i]; bb.min[i] = 0.0f; bb.max[i] = 0.0f; }
        g.v[0] = cosf(0.5f * l.param0);
        g.type = 1;
        break;
      case 2:  // SPOT :200-224
        for (int i = 0; i < 3; ++i) { g.position[i] = Wp[i]; g.u[i] = -Z[i]; bb.min[i] = Wp[i]; bb.max[i] = Wp[i]; }
        g.v[0] = cosf(l.param0);
        g.v[1] = cosf(l.param1);
        g.type = 2;
        break;
      case 3: {  // QUAD :225-253
        float pos[3], another[3];
        for (int i = 0; i < 3; ++i) {
          float p = Wp[i];
          p -= X[i] * l.param0 * 0.5f;
          p -= Y[i] * l.param1 * 0.5f;
          pos[i] = p;
          another[i] = p + X[i] * l.param0 + Y[i] * l.param1 + Z[i] * 0.01f;
          g.position[i] = p;
          g.u[i] = X[i] * l.param0;
          g.v[i] = Y[i] * l.param1;
          bb.min[i] = pos[i];
          bb.max[i] = another[i];
        }
        g.area = l.param0 * l.param1;
        g.type = 3;
        break;
      }
      case 4:  // SPHERE :254-272
        for (int i = 0; i < 3; ++i) { g.position[i] = Wp[i]; bb.min[i] = Wp[i] - l.param0; bb.max[i] = Wp[i] + l.param0; }
        g.radius = l.param0;
        g.area = 4.0f * 3.14159265358979323846f * l.param0 * l.param0;
        g.type = 4;
        break;
      default: return -1;  // :273 panic!("Invalid light type.")
    }
    orc_aabb sorted;  // :276-288
    for (int i = 0; i < 3; ++i) { sorted.min[i] = std::min(bb.min[i], bb.max[i]); sorted.max[i] = std::max(bb.min[i], bb.max[i]); }
    out[n] = g;
    aabbs[n] = sorted;
    ++n;
    if (n >= 32) break;  // :290-292
  }
  return n;
}

extern "C" int orc_pack_instances(const orc_scene_desc* scene, float* t3x4, orc_gpu_mesh_data* md, uint32_t capacity) {
  // src/scene/loader/gpu_uploader.rs:843-875
  std::vector<float> world((size_t)scene->node_count * 16);
  orc_update_node_hierarchies(scene, world.data());
  uint32_t n = 0;
  for (uint32_t k = 0; k < scene->node_count; ++k) {
    const orc_node_desc& node = scene->nodes[k];
    if (node.mesh_index == ORC_NONE) continue;
    const orc_mesh_desc& mesh = scene->meshes[node.mesh_index];
    const float* w = world.data() + 16 * k;
    for (uint32_t p = 0; p < mesh.primitive_count; ++p) {
      if (n < capacity) {
        float* t = t3x4 + 12 * n;  // :854-858 rows of the 3x4
        for (int r = 0; r < 3; ++r) { t[4 * r + 0] = w[0 + r]; t[4 * r + 1] = w[4 + r]; t[4 * r + 2] = w[8 + r]; t[4 * r + 3] = w[12 + r]; }
        memset(&md[n], 0, sizeof(md[n]));
        memcpy(md[n].transform, w, 64);  // :867
        md[n].material_index = mesh.primitives[p].material_index;  // :868
      }
      ++n;
    }
  }
  return (int)n;
}

extern "C" void orc_primitive_bounds(const orc_vertex* vertices, uint32_t count, float center[3], float extents[3]) {
  // src/scene/loader/gpu_uploader.rs:460-467 and src/scene/bounds.rs:45-108
  for (int i = 0; i < 3; ++i) { center[i] = count ? vertices[0].position[i] : 0.0f; extents[i] = 0.0f; }
  for (uint32_t k = 0; k < count; ++k) {
    float mn[3], mx[3];
    for (int i = 0; i < 3; ++i) {
      float lo = center[i] - extents[i];  // get_min :46-52
      float hi = center[i] + extents[i];  // get_max :62-68
      mn[i] = std::min(lo, vertices[k].position[i]);
      mx[i] = std::max(hi, vertices[k].position[i]);
    }
    for (int i = 0; i < 3; ++i) {  // set_min_max :80-91
      extents[i] = (mx[i] - mn[i]) * 0.5f;
      center[i] = mn[i] + extents[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// A16 — src/rt_renderer.rs:1256-1334
// ------------------------------------------------------------------------------------------------------------
static inline V3 rrt_odt_fit(V3 v) {  // :1260-1264
  V3 a = v * (v + v3s(0.0245786f)) - v3s(0.000090537f);
  V3 b = v * (v * 0.983729f + v3s(0.432951f)) + v3s(0.238081f);
  return V3{a.x / b.x, a.y / b.y, a.z / b.z};
}
static inline V3 mat3_mul(const float c0[3], const float c1[3], const float c2[3], V3 v) {
  // glam Mat3 * Vec3 = (c0*x + c1*y) + c2*z
  return V3{c0[0] * v.x + c1[0] * v.y + c2[0] * v.z, c0[1] * v.x + c1[1] * v.y + c2[1] * v.z,
            c0[2] * v.x + c1[2] * v.y + c2[2] * v.z};
}
static inline V3 clamp01(V3 c) { return V3{clampf(c.x, 0.0f, 1.0f), clampf(c.y, 0.0f, 1.0f), clampf(c.z, 0.0f, 1.0f)}; }
static inline V3 aces_fitted(V3 color) {  // :1265-1281
  static const float i0[3] = {0.59719f, 0.07600f, 0.02840f}, i1[3] = {0.35458f, 0.90834f, 0.13383f},
                     i2[3] = {0.04823f, 0.01566f, 0.83777f};
  static const float o0[3] = {1.60475f, -0.10208f, -0.00327f}, o1[3] = {-0.53108f, 1.10813f, -0.07276f},
                     o2[3] = {-0.07367f, -0.00605f, 1.07602f};
  color = mat3_mul(i0, i1, i2, color);
  color = rrt_odt_fit(color);
  color = mat3_mul(o0, o1, o2, color);
  return clamp01(color);
}
static inline V3 aces_simple(V3 c) {  // :1282-1291
  const float A = 2.51f, B = 0.03f, Y = 2.43f, D = 0.59f, E = 0.14f;
  V3 num = c * (c * A + v3s(B));
  V3 den = c * (c * Y + v3s(D)) + v3s(E);
  return clamp01(V3{num.x / den.x, num.y / den.y, num.z / den.z});
}
static inline V3 tonemap_limit(V3 c, float limit) {  // :1292-1294: c * 1.0 / (1.0 + luminance(c) / limit)
  float d = 1.0f + luminance(c) / limit;
  V3 n = c * 1.0f;
  return V3{n.x / d, n.y / d, n.z / d};
}
namespace orc {
V3 tonemap_select(V3 color, int enable_tonemap, int enable_aces, int use_simple_aces) {  // :1299-1311
  if (!enable_tonemap) return color;
  if (enable_aces) return use_simple_aces ? aces_simple(color) : aces_fitted(color);
  return tonemap_limit(color, 1.5f);
}
}  // namespace orc

extern "C" void orc_tonemap_pixels(float* rgba, size_t pixel_count, int enable_tonemap, int enable_aces, int use_simple_aces) {
  for (size_t i = 0; i < pixel_count; ++i) {
    V3 c = orc::tonemap_select(v3(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]), enable_tonemap, enable_aces, use_simple_aces);
    rgba[4 * i] = c.x; rgba[4 * i + 1] = c.y; rgba[4 * i + 2] = c.z;
  }
}

extern "C" size_t orc_pfm_bytes(const float* rgba, uint32_t width, uint32_t height, uint8_t* out, size_t capacity) {
  // :1321 writeln!("PF\n{} {}\n-1.0") then rows bottom-to-top (:1323 .rev()), RGB little-endian f32, alpha dropped
  char header[64];
  int hl = snprintf(header, sizeof(header), "PF\n%u %u\n-1.0\n", width, height);
  size_t need = (size_t)hl + (size_t)width * height * 12;
  if (need > capacity) return 0;
  memcpy(out, header, hl);
  uint8_t* p = out + hl;
  for (uint32_t row = height; row-- > 0;) {
    for (uint32_t x = 0; x < width; ++x) {
      memcpy(p, rgba + 4 * ((size_t)row * width + x), 12);  // x86-64 is little-endian
      p += 12;
    }
  }
  return need;
}

// spec_math probes
extern "C" void orc_probe_sincos_2pi(const float* u, float* s, float* c, size_t n) { for (size_t i = 0; i < n; ++i) sincos_2pi(u[i], &s[i], &c[i]); }
extern "C" void orc_probe_acos(const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = acos_poly(x[i]); }
extern "C" void orc_probe_exp_neg(const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = exp_neg_poly(x[i]); }
extern "C" void orc_probe_log(const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = log_poly(x[i]); }
extern "C" void orc_probe_hg(const float* d3, float g, const float* u1, const float* u2, float* out3, size_t n) {
  for (size_t i = 0; i < n; ++i) { V3 w = hg_sample(V3{d3[0], d3[1], d3[2]}, g, u1[i], u2[i]); out3[3 * i] = w.x; out3[3 * i + 1] = w.y; out3[3 * i + 2] = w.z; }
}
extern "C" void orc_probe_atan2(const float* y, const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = atan2_poly(y[i], x[i]); }
extern "C" void orc_probe_rng(uint32_t pixel_id, uint32_t frame_index, float* out, size_t n) {
  uint32_t s = rng_init(pixel_id, frame_index);
  for (size_t i = 0; i < n; ++i) out[i] = rng_next(&s);
}
