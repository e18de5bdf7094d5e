// ORACLE — TEST INFRASTRUCTURE ONLY (see spec_math.h).
// oracle_render.cpp — the scalar CPU statement of docs/RENDER_SPEC.md §5-§8: camera rays, the path loop,
// surface/material/light/environment evaluation, accumulation and the final-image tonemap.
//
// Parity status: the reference's integrator lives in SPIR-V shaders supplied by the embedding application
// (src/rt_renderer.rs:925-1112) which are NOT in the reference repository, so pixel parity with the reference is
// UNPINNED.  This file is the spec the HIP kernels are held to; it consumes exactly the records the
// reference binds for its shaders (HalaGlobalUniform src/rt_renderer.rs:44-65; cameras/lights/materials/primitives
// src/rt_renderer.rs:141-181; env tables src/envmap.rs:239-388).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_api.h"
#include "oracle_scene.h"

using namespace orc;

namespace {

constexpr float kTMax = 3.402823466e+38f;
constexpr float kTwoPiSq = 19.7392088021787172376f;  // 2*pi^2

struct Frame {
  const orc_scene* s;
  orc_render_params p;
  float res_w, res_h, aspect, tan_half, pixel_spread;
  float env_rotation;  // degrees / 360 (src/rt_renderer.rs:420)
  uint32_t env_type;
};

static inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---- RENDER_SPEC §5 camera ---------------------------------------------------------------------------------
static void camera_ray(const Frame& f, uint32_t px, uint32_t py, uint32_t* rng, V3* o, V3* d) {
  const orc_gpu_camera& cam = f.s->cameras[0];  // camera_index is always 0 (src/rt_renderer.rs:415)
  float r1 = rng_next(rng), r2 = rng_next(rng), r3 = rng_next(rng), r4 = rng_next(rng);
  float fx = ((float)px + r1) / f.res_w;
  float fy = ((float)py + r2) / f.res_h;
  float ndc_x = fx * 2.0f - 1.0f;
  float ndc_y = 1.0f - fy * 2.0f;
  V3 pos = ld3(cam.position), right = ld3(cam.right), up = ld3(cam.up), fwd = ld3(cam.forward);
  if (cam.type == 0) {
    float dx = ndc_x * f.aspect * f.tan_half;
    float dy = ndc_y * f.tan_half;
    V3 dir = normalize3(madd3(up, dy, madd3(right, dx, fwd)));
    float aperture = cam.aperture_or_ymag;
    if (aperture > 0.0f) {
      float ft = cam.focal_distance_or_xmag / dot3(dir, fwd);
      V3 focus = madd3(dir, ft, pos);
      float r = aperture * sqrtf(r3);
      float s, c;
      sincos_2pi(r4, &s, &c);
      V3 org = madd3(up, r * s, madd3(right, r * c, pos));
      *o = org;
      *d = normalize3(focus - org);
    } else {
      *o = pos;
      *d = dir;
    }
  } else {
    *o = madd3(up, ndc_y * cam.aperture_or_ymag, madd3(right, ndc_x * cam.focal_distance_or_xmag, pos));
    *d = normalize3(fwd);
  }
}

// ---- RENDER_SPEC §7.3 environment ----------------------------------------------------------------------------
static inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline V3 env_texel(const EnvMap& e, int x, int y) { const float* p = e.pixels.data() + 4 * ((size_t)y * e.width + x); return v3(p[0], p[1], p[2]); }

static inline void env_dir_to_uv(const Frame& f, V3 d, float* uu, float* vv) {
  float theta = acos_poly(clampf(d.y, -1.0f, 1.0f));
  float phi = atan2_poly(d.z, d.x);
  *uu = (kPi + phi) * kInvTwoPi + f.env_rotation;
  *vv = theta * kInvPi;
}
static V3 env_map_eval(const Frame& f, V3 d) {
  const EnvMap& e = f.s->env;
  float uu, vv;
  env_dir_to_uv(f, d, &uu, &vv);
  float x = uu * (float)e.width - 0.5f, y = vv * (float)e.height - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = wrapi((int)x0, (int)e.width), iy0 = wrapi((int)y0, (int)e.height);
  int ix1 = wrapi(ix0 + 1, (int)e.width), iy1 = wrapi(iy0 + 1, (int)e.height);
  V3 c00 = env_texel(e, ix0, iy0), c10 = env_texel(e, ix1, iy0), c01 = env_texel(e, ix0, iy1), c11 = env_texel(e, ix1, iy1);
  V3 top = c00 * (1.0f - fx) + c10 * fx;
  V3 bot = c01 * (1.0f - fx) + c11 * fx;
  return (top * (1.0f - fy) + bot * fy) * f.p.env_intensity;
}
static float env_map_pdf(const Frame& f, V3 d) {
  const EnvMap& e = f.s->env;
  float uu, vv;
  env_dir_to_uv(f, d, &uu, &vv);
  int ix = wrapi((int)floorf(uu * (float)e.width), (int)e.width);
  int iy = std::min((int)(vv * (float)e.height), (int)e.height - 1);
  float lum = luminance(env_texel(e, ix, iy));
  float st = sqrtf(maxf(0.0f, 1.0f - d.y * d.y));
  if (!(st > 0.0f) || !(lum > 0.0f)) return 0.0f;
  return (lum * (float)(e.width * e.height)) / (e.total_sum * kTwoPiSq * st);
}
static bool env_map_sample(const Frame& f, float r1, float r2, V3* wi, float* pdf) {
  const EnvMap& e = f.s->env;
  const int W = (int)e.width, H = (int)e.height;
  float fy = r1 * (float)H;
  int iy = std::min((int)fy, H - 1);
  float mv = e.marginal[iy];
  if (!(mv >= 0.0f)) mv = 0.0f;
  int row = std::min((int)(mv * (float)H + 0.5f), H - 1);
  float fxx = r2 * (float)W;
  int ix = std::min((int)fxx, W - 1);
  float cu = e.conditional[(size_t)row * W + ix];
  if (!(cu >= 0.0f)) cu = 0.0f;
  int col = std::min((int)(cu * (float)W + 0.5f), W - 1);
  float ju = fxx - (float)ix, jv = fy - (float)iy;
  float uu = ((float)col + ju) / (float)W;
  float vv = ((float)row + jv) / (float)H;
  float t = uu - f.env_rotation;
  t = t - floorf(t);
  if (t >= 1.0f) t = 0.0f;
  float sp, cp, st, ct;
  sincos_2pi(t, &sp, &cp);
  sincos_2pi(vv * 0.5f, &st, &ct);
  *wi = v3(-st * cp, ct, -st * sp);
  float lum = luminance(env_texel(e, col, row));
  if (!(st > 0.0f) || !(lum > 0.0f)) { *pdf = 0.0f; return false; }
  *pdf = (lum * (float)(e.width * e.height)) / (e.total_sum * kTwoPiSq * st);
  return true;
}
static V3 sky_eval(const Frame& f, V3 d) {
  float t = 0.5f * (d.y + 1.0f);
  V3 g = ld3(f.p.ground_color), s = ld3(f.p.sky_color);
  return (g * (1.0f - t) + s * t) * f.p.env_intensity;
}

// ---- RENDER_SPEC §7.1 materials --------------------------------------------------------------------------------
// DIFFUSE (type 0): Oren–Nayar with A = ax, B = ay as packed by src/scene/gpu/material.rs:53-60
// DISNEY (type 1), RENDER_SPEC §7.1b: diffuse (Disney retro-reflection + sheen) + anisotropic GGX specular with the
// alpha values packed by src/scene/gpu/material.rs:61-69 (roughness := r^2, ax, ay) + GGX clearcoat; VNDF sampling.
static inline float schlick5(float x) {
  float m = clampf(1.0f - x, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
static inline V3 mix3(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float ggx_d(V3 h, float ax, float ay) {  // h in the local frame
  float hx = h.x / ax, hy = h.y / ay;
  float k = hx * hx + hy * hy + h.z * h.z;
  return 1.0f / (kPi * ax * ay * k * k);
}
static inline float ggx_g1(V3 w, float ax, float ay) {  // Smith, w in the local frame, w.z > 0
  float a = ax * w.x, b = ay * w.y;
  float l = (a * a + b * b) / (w.z * w.z);
  return 2.0f / (1.0f + sqrtf(1.0f + l));
}
static inline V3 ggx_sample_vndf(V3 v, float ax, float ay, float r1, float r2) {  // Heitz 2018
  V3 vh = normalize3(v3(ax * v.x, ay * v.y, v.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 t1 = v3(1.0f, 0.0f, 0.0f);
  if (lensq > 0.0f) { float il = 1.0f / sqrtf(lensq); t1 = v3(-vh.y * il, vh.x * il, 0.0f); }
  V3 t2 = cross3(vh, t1);
  float r = sqrtf(r1);
  float s, c;
  sincos_2pi(r2, &s, &c);
  float a = r * c, b = r * s;
  float sn = 0.5f * (1.0f + vh.z);
  b = (1.0f - sn) * sqrtf(maxf(0.0f, 1.0f - a * a)) + sn * b;
  float cz = sqrtf(maxf(0.0f, 1.0f - a * a - b * b));
  V3 nh = t1 * a + t2 * b + vh * cz;
  return normalize3(v3(ax * nh.x, ay * nh.y, maxf(0.0f, nh.z)));
}
struct DisneyLobes { float pd, ps, pc, pt; V3 cspec0, csheen; float cc_alpha; };
// RENDER_SPEC §7.1c: what the refraction lobe needs beyond the packed material: trans = specular_transmission * (1 - metallic)
// (after the metallic map), eta = index of the far side relative to the side the path arrives from
struct Trans { float trans, eta; };
// exact unpolarised Fresnel reflectance of a dielectric interface; c = |cos| on the arriving side
static inline float fresnel_dielectric(float c, float eta) {
  float g2 = eta * eta - 1.0f + c * c;
  if (!(g2 > 0.0f)) return 1.0f;
  float g = sqrtf(g2);
  float a = (g - c) / (g + c);
  float b = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
  return 0.5f * a * a * (1.0f + b * b);
}
static inline DisneyLobes disney_lobes(const orc_gpu_material& m, V3 base, Trans tr, float nv) {
  DisneyLobes d;
  float lb = luminance(base);
  V3 tint = lb > 0.0f ? base * (1.0f / lb) : v3s(1.0f);
  float f0 = (m.ior - 1.0f) / (m.ior + 1.0f);
  f0 = f0 * f0;
  d.cspec0 = mix3(mix3(v3s(1.0f), tint, m.specular_tint) * f0, base, m.metallic);
  d.csheen = mix3(v3s(1.0f), tint, m.sheen_tint);
  d.cc_alpha = maxf(0.001f, m.clearcoat_roughness * m.clearcoat_roughness);
  float fv = schlick5(nv);
  float wd = (1.0f - m.metallic) * lb * (1.0f - tr.trans);
  float ws = luminance(mix3(d.cspec0, v3s(1.0f), fv));
  float wc = 0.25f * m.clearcoat * mixf(0.04f, 1.0f, fv);
  float wt = 0.0f;
  if (tr.trans > 0.0f) {  // §7.1c: the transmissive share reflects by the exact dielectric Fresnel term (1 beyond the critical angle)
    float fe = fresnel_dielectric(nv, tr.eta);
    ws = mixf(ws, fe, tr.trans);
    wt = tr.trans * (1.0f - fe);
  }
  float sum = wd + ws + wc + wt;
  if (!(sum > 0.0f)) { d.pd = d.ps = d.pc = d.pt = 0.0f; return d; }
  float inv = 1.0f / sum;
  d.pd = wd * inv; d.ps = ws * inv; d.pc = wc * inv; d.pt = wt * inv;
  return d;
}
static inline void disney_eval(const orc_gpu_material& m, V3 base, Trans tr, V3 wo, V3 wi, V3 n, V3* f, float* pdf) {
  float nl = dot3(n, wi), nv = dot3(n, wo);
  *f = v3s(0.0f); *pdf = 0.0f;
  if (!(nv > 0.0f)) return;
  V3 t, b;
  onb(n, &t, &b);
  V3 lo = v3(dot3(wo, t), dot3(wo, b), nv), li = v3(dot3(wi, t), dot3(wi, b), nl);
  if (nl < 0.0f) {  // §7.1c refraction
    if (!(tr.trans > 0.0f)) return;
    V3 h = lo + li * tr.eta;
    float h2 = dot3(h, h);
    if (!(h2 > 0.0f)) return;
    h = h * (1.0f / sqrtf(h2));
    if (h.z < 0.0f) h = -h;
    float odh = dot3(lo, h), idh = dot3(li, h);
    if (!(odh > 0.0f && idh < 0.0f)) return;
    DisneyLobes d = disney_lobes(m, base, tr, nv);
    float fr = fresnel_dielectric(odh, tr.eta);
    float ds = ggx_d(h, m.ax, m.ay);
    float g1o = ggx_g1(lo, m.ax, m.ay), g1i = ggx_g1(li, m.ax, m.ay);
    float den = odh + tr.eta * idh;
    float jac = tr.eta * tr.eta * (-idh) / (den * den);
    float w = tr.trans * (1.0f - fr) * ds * g1o * g1i * odh * jac / (-nl * nv);
    *f = v3(sqrtf(base.x), sqrtf(base.y), sqrtf(base.z)) * w;
    *pdf = d.pt * (g1o * odh * ds / nv) * jac;
    return;
  }
  if (!(nl > 0.0f)) return;
  V3 h = normalize3(lo + li);
  float ldh = dot3(li, h);
  DisneyLobes d = disney_lobes(m, base, tr, nv);
  float fl = schlick5(nl), fv = schlick5(nv), fh = schlick5(ldh);
  // diffuse + sheen
  float fd90 = 0.5f + 2.0f * sqrtf(m.roughness) * ldh * ldh;
  float fd = mixf(1.0f, fd90, fl) * mixf(1.0f, fd90, fv);
  float dw = (1.0f - m.metallic) * (1.0f - tr.trans);
  V3 fs = mix3(d.cspec0, v3s(1.0f), fh);  // specular Fresnel; the diffuse lobe only gets what it lets through
  if (tr.trans > 0.0f) fs = mix3(fs, v3s(fresnel_dielectric(ldh, tr.eta)), tr.trans);  // §7.1c: pairs with the (1 - F) of the refraction lobe
  V3 fr = base * (v3s(1.0f) - fs) * (kInvPi * fd * dw) + d.csheen * (m.sheen * fh * dw);
  // specular
  float ds = ggx_d(h, m.ax, m.ay);
  float g1o = ggx_g1(lo, m.ax, m.ay), g1i = ggx_g1(li, m.ax, m.ay);
  float denom = 4.0f * nl * nv;
  fr = fr + fs * (ds * g1o * g1i / denom);
  // clearcoat
  float dc = ggx_d(h, d.cc_alpha, d.cc_alpha);
  float c1o = ggx_g1(lo, d.cc_alpha, d.cc_alpha), c1i = ggx_g1(li, d.cc_alpha, d.cc_alpha);
  float fc = mixf(0.04f, 1.0f, fh);
  fr = fr + v3s(0.25f * m.clearcoat * fc * dc * c1o * c1i / denom);
  *f = fr;
  float inv4nv = 1.0f / (4.0f * nv);
  *pdf = d.pd * (nl * kInvPi) + d.ps * (g1o * ds * inv4nv) + d.pc * (c1o * dc * inv4nv);
}

static inline void bsdf_eval(const orc_gpu_material& m, V3 base, Trans tr, V3 wo, V3 wi, V3 n, V3* f, float* pdf) {
  if (m.type == 1u) { disney_eval(m, base, tr, wo, wi, n, f, pdf); return; }
  float nl = dot3(n, wi), nv = dot3(n, wo);
  if (!(nl > 0.0f && nv > 0.0f)) { *f = v3s(0.0f); *pdf = 0.0f; return; }
  float s = dot3(wi, wo) - nl * nv;
  float tterm = maxf(0.0f, s) / maxf(nl, nv);
  float k = kInvPi * (m.ax + m.ay * tterm);
  *f = base * k;
  *pdf = nl * kInvPi;
}
static inline bool bsdf_sample(const orc_gpu_material& m, V3 base, Trans tr, V3 wo, V3 n, float r1, float r2, float r3, V3* wi, V3* f, float* pdf) {
  V3 t, b;
  onb(n, &t, &b);
  if (m.type == 1u) {
    float nv = dot3(n, wo);
    if (!(nv > 0.0f)) { *f = v3s(0.0f); *pdf = 0.0f; return false; }
    DisneyLobes d = disney_lobes(m, base, tr, nv);
    if (r3 < d.pd) {
      *wi = to_world(cosine_hemisphere(r1, r2), t, b, n);
    } else if (r3 >= d.pd + d.ps + d.pc) {  // §7.1c: refract through a VNDF-sampled facet
      V3 lo = v3(dot3(wo, t), dot3(wo, b), nv);
      V3 h = ggx_sample_vndf(lo, m.ax, m.ay, r1, r2);
      float c = dot3(lo, h);
      float ie = 1.0f / tr.eta;
      float k = 1.0f - (1.0f - c * c) * (ie * ie);
      if (!(k > 0.0f)) { *f = v3s(0.0f); *pdf = 0.0f; return false; }
      V3 li = h * (c * ie - sqrtf(k)) - lo * ie;
      *wi = to_world(li, t, b, n);
    } else {
      bool spec = r3 < d.pd + d.ps;
      float ax = spec ? m.ax : d.cc_alpha, ay = spec ? m.ay : d.cc_alpha;
      V3 lo = v3(dot3(wo, t), dot3(wo, b), nv);
      V3 h = ggx_sample_vndf(lo, ax, ay, r1, r2);
      float k = 2.0f * dot3(lo, h);
      V3 li = h * k - lo;
      *wi = to_world(li, t, b, n);
    }
    disney_eval(m, base, tr, wo, *wi, n, f, pdf);
    return *pdf > 0.0f;
  }
  V3 l = cosine_hemisphere(r1, r2);
  *wi = to_world(l, t, b, n);
  bsdf_eval(m, base, tr, wo, *wi, n, f, pdf);
  return *pdf > 0.0f;
}

// ---- RENDER_SPEC §7.2 lights --------------------------------------------------------------------------------------
struct LightSample { V3 wi; float dist; V3 le; float pdf; bool delta; bool valid; };

static LightSample sample_light(const orc_gpu_light& l, V3 P, float r1, float r2) {
  LightSample s; s.valid = false; s.delta = true; s.pdf = 0.0f; s.dist = kTMax; s.le = v3s(0.0f); s.wi = v3(0, 1, 0);
  V3 inten = ld3(l.intensity), pos = ld3(l.position);
  switch (l.type) {
    case 0: case 2: {  // POINT, SPOT
      V3 to = pos - P;
      float d2 = dot3(to, to);
      if (!(d2 > 0.0f)) return s;
      float dist = sqrtf(d2);
      s.wi = to * (1.0f / dist);
      s.dist = dist;
      s.le = inten * (1.0f / d2);
      if (l.type == 2) {
        V3 axis = normalize3(ld3(l.u));
        float cosang = -dot3(s.wi, axis);
        float ci = l.v[0], co = l.v[1];
        float t;
        if (ci > co) t = clampf((cosang - co) / (ci - co), 0.0f, 1.0f); else t = cosang >= co ? 1.0f : 0.0f;
        float sm = t * t * (3.0f - 2.0f * t);
        s.le = s.le * sm;
      }
      s.valid = true;
      return s;
    }
    case 1: {  // DIRECTIONAL
      V3 axis = normalize3(-ld3(l.u));
      float cosmax = l.v[0];
      if (cosmax >= 1.0f) s.wi = axis;
      else {
        float ct = 1.0f - r1 * (1.0f - cosmax);
        float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
        float sp, cp;
        sincos_2pi(r2, &sp, &cp);
        V3 t, b;
        onb(axis, &t, &b);
        s.wi = to_world(v3(st * cp, st * sp, ct), t, b, axis);
      }
      s.dist = kTMax;
      s.le = inten;
      s.valid = true;
      return s;
    }
    case 3: {  // QUAD
      V3 u = ld3(l.u), v = ld3(l.v);
      V3 pt = madd3(v, r2, madd3(u, r1, pos));
      V3 n = normalize3(cross3(u, v));
      V3 to = pt - P;
      float d2 = dot3(to, to);
      if (!(d2 > 0.0f)) return s;
      float dist = sqrtf(d2);
      s.wi = to * (1.0f / dist);
      float cosl = -dot3(s.wi, n);
      if (!(cosl > 0.0f)) return s;
      s.dist = dist; s.le = inten; s.pdf = d2 / (l.area * cosl); s.delta = false; s.valid = true;
      return s;
    }
    case 4: {  // SPHERE
      float z = 1.0f - 2.0f * r1;
      float rr = sqrtf(maxf(0.0f, 1.0f - z * z));
      float sp, cp;
      sincos_2pi(r2, &sp, &cp);
      V3 nl = v3(rr * cp, rr * sp, z);
      V3 pt = madd3(nl, l.radius, pos);
      V3 to = pt - P;
      float d2 = dot3(to, to);
      if (!(d2 > 0.0f)) return s;
      float dist = sqrtf(d2);
      s.wi = to * (1.0f / dist);
      float cosl = -dot3(s.wi, nl);
      if (!(cosl > 0.0f)) return s;
      s.dist = dist; s.le = inten; s.pdf = d2 / (l.area * cosl); s.delta = false; s.valid = true;
      return s;
    }
    default: return s;
  }
}

// analytic intersection of the hittable lights (QUAD, SPHERE); returns t (<= 0: none) and the solid-angle pdf
static float intersect_light(const orc_gpu_light& l, V3 o, V3 d, float* pdf) {
  V3 pos = ld3(l.position);
  if (l.type == 3) {
    V3 u = ld3(l.u), v = ld3(l.v);
    V3 n = normalize3(cross3(u, v));
    float dn = dot3(d, n);
    if (!(dn < 0.0f)) return -1.0f;
    float t = dot3(pos - o, n) / dn;
    if (!(t > 0.0f)) return -1.0f;
    V3 hp = madd3(d, t, o) - pos;
    float a = dot3(hp, u) / dot3(u, u), b = dot3(hp, v) / dot3(v, v);
    if (!(a >= 0.0f && a <= 1.0f && b >= 0.0f && b <= 1.0f)) return -1.0f;
    *pdf = (t * t) / (l.area * (-dn));
    return t;
  }
  if (l.type == 4) {
    V3 oc = o - pos;
    float b = dot3(oc, d);
    float c = dot3(oc, oc) - l.radius * l.radius;
    float disc = b * b - c;
    if (!(disc > 0.0f)) return -1.0f;
    float t = -b - sqrtf(disc);
    if (!(t > 0.0f)) return -1.0f;
    V3 nl = (madd3(d, t, o) - pos) * (1.0f / l.radius);
    float cosl = -dot3(d, nl);
    if (!(cosl > 0.0f)) return -1.0f;
    *pdf = (t * t) / (l.area * cosl);
    return t;
  }
  return -1.0f;
}

// ---- RENDER_SPEC §6 surface reconstruction ----------------------------------------------------------------------
struct Surface { V3 P, ns, ng; orc_gpu_material m; V3 base; Trans tr; float opacity; V3 absorb, glow; float sigma, hg; V3 scol; };  // sigma / hg / scol: §7.1f  // absorb / glow: §7.1e  // opacity: §7.1d, material x base-colour-map alpha  // m: the packed material after texture modulation

static inline V3 transform_normal(const float* m, V3 n) {
  // inverse-transpose of the upper 3x3 = cofactor matrix / det; columns c0,c1,c2 of M
  V3 c0 = v3(m[0], m[1], m[2]), c1 = v3(m[4], m[5], m[6]), c2 = v3(m[8], m[9], m[10]);
  V3 k0 = cross3(c1, c2), k1 = cross3(c2, c0), k2 = cross3(c0, c1);
  float det = dot3(c0, k0);
  V3 r = v3(fmaf(k2.x, n.z, fmaf(k1.x, n.y, k0.x * n.x)), fmaf(k2.y, n.z, fmaf(k1.y, n.y, k0.y * n.x)),
            fmaf(k2.z, n.z, fmaf(k1.z, n.y, k0.z * n.x)));
  return det < 0.0f ? -r : r;
}

// ---- RENDER_SPEC §7.4 textures ----------------------------------------------------------------------------------
struct C4 { float x, y, z, w; };
static inline float log2_approx(float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  float e = (float)((int)((b >> 23) & 255u) - 127);
  uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;
  float m;
  memcpy(&m, &mb, 4);
  return e + (m - 1.0f);
}
static C4 tex_bilinear(const Image& img, uint32_t level, float u, float v) {
  int w = std::max((int)(img.width >> level), 1), h = std::max((int)(img.height >> level), 1);
  const float* base = img.levels[level].data();
  float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = wrapi((int)x0, w), iy0 = wrapi((int)y0, h);
  int ix1 = wrapi(ix0 + 1, w), iy1 = wrapi(iy0 + 1, h);
  const float* c00 = base + 4 * ((size_t)iy0 * w + ix0); const float* c10 = base + 4 * ((size_t)iy0 * w + ix1);
  const float* c01 = base + 4 * ((size_t)iy1 * w + ix0); const float* c11 = base + 4 * ((size_t)iy1 * w + ix1);
  float gx = 1.0f - fx, gy = 1.0f - fy;
  float r[4];
  for (int k = 0; k < 4; ++k) r[k] = (c00[k] * gx + c10[k] * fx) * gy + (c01[k] * gx + c11[k] * fx) * fy;
  return C4{r[0], r[1], r[2], r[3]};
}
static C4 tex_sample(const orc_scene* s, uint32_t tex, float u, float v, float lod) {
  const Image& img = s->images[s->texture_image[tex]];
  float top = (float)(img.mips - 1u);
  lod = lod < 0.0f ? 0.0f : (lod > top ? top : lod);
  float l0 = floorf(lod);
  float fl = lod - l0;
  uint32_t level = (uint32_t)l0;
  C4 a = tex_bilinear(img, level, u, v);
  if (fl > 0.0f && level + 1u < img.mips) {
    C4 b = tex_bilinear(img, level + 1u, u, v);
    float g = 1.0f - fl;
    a.x = a.x * g + b.x * fl; a.y = a.y * g + b.y * fl; a.z = a.z * g + b.z * fl; a.w = a.w * g + b.w * fl;
  }
  return a;
}
static float tex_lod(const orc_scene* s, uint32_t tex, float lod_base) {
  const Image& img = s->images[s->texture_image[tex]];
  return lod_base + 0.5f * log2_approx((float)img.width * (float)img.height);
}
static inline V3 transform_vector(const float* m, V3 p) {
  return v3(fmaf(m[8], p.z, fmaf(m[4], p.y, m[0] * p.x)), fmaf(m[9], p.z, fmaf(m[5], p.y, m[1] * p.x)), fmaf(m[10], p.z, fmaf(m[6], p.y, m[2] * p.x)));
}

// RENDER_SPEC 7.1d: what decides whether a translucent triangle blocks an any-hit ray: the material's opacity times the alpha of its
// base-colour map, read with the bilinear fetch of level 0 at the hit's interpolated texture coordinates
}  // namespace
namespace orc {
float hit_alpha(const orc_scene* s, uint32_t prim, float u, float v) {
  const Instance& inst = s->instances[s->tri_instance[prim]];
  const orc_gpu_material& m = s->materials[inst.material_index];
  float alpha = m.opacity;
  if (m.base_color_map_index < s->texture_image.size()) {
    const uint32_t lt = prim - inst.first_triangle;
    const orc_vertex& a = inst.vertices[inst.indices[3 * lt]];
    const orc_vertex& b = inst.vertices[inst.indices[3 * lt + 1]];
    const orc_vertex& c = inst.vertices[inst.indices[3 * lt + 2]];
    const float w0 = 1.0f - u - v;
    const float tu = fmaf(c.tex_coord[0], v, fmaf(b.tex_coord[0], u, a.tex_coord[0] * w0));
    const float tv = fmaf(c.tex_coord[1], v, fmaf(b.tex_coord[1], u, a.tex_coord[1] * w0));
    alpha = alpha * tex_bilinear(s->images[s->texture_image[m.base_color_map_index]], 0, tu, tv).w;
  }
  return alpha;
}
}  // namespace orc
namespace {

static Surface make_surface(const orc_scene* s, float pixel_spread, V3 o, V3 d, const Hit& h) {
  Surface sf;
  const Tri& tr = s->tris_by_id[h.prim];
  const Instance& inst = s->instances[s->tri_instance[h.prim]];
  uint32_t lt = h.prim - inst.first_triangle;
  const orc_vertex& a = inst.vertices[inst.indices[3 * lt]];
  const orc_vertex& b = inst.vertices[inst.indices[3 * lt + 1]];
  const orc_vertex& c = inst.vertices[inst.indices[3 * lt + 2]];
  float w0 = 1.0f - h.u - h.v;
  V3 nl = madd3(ld3(c.normal), h.v, madd3(ld3(b.normal), h.u, ld3(a.normal) * w0));
  sf.ns = normalize3(transform_normal(inst.transform, nl));
  // geometric normal: cross(e1, e2) of the stored edges — world space, or (RENDER_SPEC 4.5, instanced instances) object space, moved to
  // world space like a normal
  V3 gcross = cross3(ld3(tr.e1), ld3(tr.e2));
  if (inst.instanced) gcross = transform_normal(inst.transform, gcross);
  sf.ng = normalize3(gcross);
  sf.P = madd3(d, h.t, o);
  sf.m = s->materials[inst.material_index];
  sf.opacity = sf.m.opacity;
  orc_gpu_material& m = sf.m;
  sf.base = ld3(m.base_color);
  const orc_gpu_material& pm = s->materials[inst.material_index];  // as packed (the maps modulate a copy)
  const uint32_t nt = (uint32_t)s->texture_image.size();
  bool has_base = pm.base_color_map_index < nt, has_nrm = pm.normal_map_index < nt;
  bool has_mr = pm.metallic_roughness_map_index < nt, has_em = pm.emission_map_index < nt;
  if (has_base || has_nrm || has_mr || has_em) {
    float tu = fmaf(c.tex_coord[0], h.v, fmaf(b.tex_coord[0], h.u, a.tex_coord[0] * w0));
    float tv = fmaf(c.tex_coord[1], h.v, fmaf(b.tex_coord[1], h.u, a.tex_coord[1] * w0));
    float du1 = b.tex_coord[0] - a.tex_coord[0], dv1 = b.tex_coord[1] - a.tex_coord[1];
    float du2 = c.tex_coord[0] - a.tex_coord[0], dv2 = c.tex_coord[1] - a.tex_coord[1];
    float uv_area = fabsf(du1 * dv2 - dv1 * du2);
    float world_area = sqrtf(dot3(gcross, gcross));
    float cosi = maxf(fabsf(dot3(d, sf.ng)), 0.1f);
    float foot = h.t * pixel_spread / cosi;
    float ratio = foot * foot * uv_area / world_area;
    float lod_base = (ratio > 0.0f && ratio < 3.0e38f) ? 0.5f * log2_approx(ratio) : 0.0f;
    if (has_base) {
      C4 t = tex_sample(s, pm.base_color_map_index, tu, tv, tex_lod(s, pm.base_color_map_index, lod_base));
      sf.base = sf.base * v3(t.x, t.y, t.z);
      sf.opacity = sf.opacity * t.w;
    }
    if (has_em) {
      C4 t = tex_sample(s, pm.emission_map_index, tu, tv, tex_lod(s, pm.emission_map_index, lod_base));
      V3 e = ld3(m.emission) * v3(t.x, t.y, t.z);
      m.emission[0] = e.x; m.emission[1] = e.y; m.emission[2] = e.z;
    }
    if (has_mr) {
      C4 t = tex_sample(s, pm.metallic_roughness_map_index, tu, tv, tex_lod(s, pm.metallic_roughness_map_index, lod_base));
      m.metallic = m.metallic * t.z;
      if (pm.type == 1u) {
        float rl = sqrtf(pm.roughness) * t.y;
        float r2 = rl * rl;
        float aspect = sqrtf(1.0f - clampf(pm.anisotropic, 0.0f, 1.0f) * 0.9f);
        m.roughness = r2;
        m.ax = maxf(0.001f, r2 / aspect);
        m.ay = maxf(0.001f, r2 * aspect);
      }
    }
    if (has_nrm) {
      C4 t = tex_sample(s, pm.normal_map_index, tu, tv, tex_lod(s, pm.normal_map_index, lod_base));
      V3 tl = madd3(ld3(c.tangent), h.v, madd3(ld3(b.tangent), h.u, ld3(a.tangent) * w0));
      V3 tw = transform_vector(inst.transform, tl);
      tw = tw - sf.ns * dot3(sf.ns, tw);
      float tl2 = dot3(tw, tw);
      if (tl2 > 0.0f) {
        tw = tw * (1.0f / sqrtf(tl2));
        V3 bw = cross3(sf.ns, tw);
        V3 nts = v3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
        V3 nn = to_world(nts, tw, bw, sf.ns);
        float nn2 = dot3(nn, nn);
        if (nn2 > 0.0f) sf.ns = nn * (1.0f / sqrtf(nn2));
      }
    }
  }
  sf.tr.trans = 0.0f; sf.tr.eta = sf.m.ior;
  if (sf.m.type == 1u) sf.tr.trans = sf.m.specular_transmission * (1.0f - sf.m.metallic);  // after the metallic map
  if (dot3(sf.ns, sf.ng) < 0.0f) sf.ng = -sf.ng;
  sf.absorb = v3s(1.0f); sf.glow = v3s(0.0f);
  sf.sigma = 0.0f; sf.hg = 0.0f; sf.scol = v3s(1.0f);
  if (dot3(sf.ng, d) > 0.0f) {  // the path arrives from behind the surface: it is leaving the object
    sf.ns = -sf.ns; sf.ng = -sf.ng;
    sf.tr.eta = 1.0f / sf.m.ior;
    const orc_gpu_material& pm = s->materials[inst.material_index];  // §7.1e: the medium of the object just crossed
    float dt = pm.medium_density * h.t;
    if (pm.medium_type == 1u)
      sf.absorb = v3(exp_neg_poly(-(dt * (1.0f - pm.medium_color[0]))), exp_neg_poly(-(dt * (1.0f - pm.medium_color[1]))),
                     exp_neg_poly(-(dt * (1.0f - pm.medium_color[2]))));
    else if (pm.medium_type == 3u)
      sf.glow = v3(pm.medium_color[0] * dt, pm.medium_color[1] * dt, pm.medium_color[2] * dt);
    else if (pm.medium_type == 2u) {  // §7.1f SCATTER
      sf.sigma = pm.medium_density; sf.hg = pm.medium_anisotropy; sf.scol = ld3(pm.medium_color);
    }
  }
  return sf;
}

// ---- RENDER_SPEC §6 the path loop ----------------------------------------------------------------------------------
struct PixelOut { V3 L, albedo, normal; };

static const float kNoNeePdf = 1e18f;  // §7.1f: power_heuristic(kNoNeePdf, b) == 1 for every pdf b a light or the env map can report

static PixelOut trace_path(const Frame& f, uint32_t px, uint32_t py, uint32_t frame_index, orc_render_stats* st, Counters* ctr) {
  const orc_scene* s = f.s;
  uint32_t rng = rng_init(py * f.p.width + px, frame_index);
  V3 o, d;
  camera_ray(f, px, py, &rng, &o, &d);
  V3 L = v3s(0.0f), Le = v3s(0.0f), T = v3s(1.0f);
  float prev_pdf = kNoNeePdf;  // no vertex has sampled a direction yet: an emitter reached through skipped (7.1d) surfaces counts in full
  PixelOut out; out.albedo = v3s(0.0f); out.normal = v3s(0.0f);
  const uint32_t nl = (uint32_t)s->light_count;
  for (uint32_t depth = 0; depth < f.p.max_depth; ++depth) {
    Hit h = scene_trace_closest(s, o, d, 0.0f, kTMax, ctr);
    st->rays_closest++;
    float t_surf = h.prim != ORC_NONE ? h.t : kTMax;
    // hittable analytic lights
    int hit_light = -1; float t_light = t_surf, light_pdf = 0.0f;
    for (uint32_t i = 0; i < nl; ++i) {
      float lp = 0.0f;
      float tl = intersect_light(s->lights[i], o, d, &lp);
      if (tl > 0.0f && tl < t_light) { t_light = tl; hit_light = (int)i; light_pdf = lp; }
    }
    if (hit_light >= 0) {
      V3 le = ld3(s->lights[hit_light].intensity);
      float w = 1.0f;
      if (depth > 0) w = power_heuristic(prev_pdf, light_pdf * (1.0f / (float)nl));
      L = L + T * le * w;
      if (depth == 0) out.albedo = v3(minf(le.x, 1.0f), minf(le.y, 1.0f), minf(le.z, 1.0f));
      break;
    }
    if (h.prim == ORC_NONE) {
      V3 env; float w = 1.0f;
      if (f.env_type == 1) {
        env = env_map_eval(f, d);
        if (depth > 0) w = power_heuristic(prev_pdf, env_map_pdf(f, d));
      } else env = sky_eval(f, d);
      L = L + T * env * w;
      if (depth == 0) out.albedo = v3(minf(env.x, 1.0f), minf(env.y, 1.0f), minf(env.z, 1.0f));
      break;
    }
    Surface sf = make_surface(s, f.pixel_spread, o, d, h);
    if (depth == 0) { out.albedo = sf.base; out.normal = sf.ns; }
    if (sf.glow.x > 0.0f || sf.glow.y > 0.0f || sf.glow.z > 0.0f) L = L + T * sf.glow;  // §7.1e, before the absorption of the same segment... (only one of the two is ever set)
    T = T * sf.absorb;
    bool scattered = false;
    V3 pm = sf.P;       // the vertex the path is at: the surface point, or the scattering point inside the medium
    float hg_g = 0.0f;
    if (sf.sigma > 0.0f) {  // §7.1f: free flight through a scattering medium; the surface is only reached if the flight outlasts the segment
      float rs = rng_next(&rng);
      float dist = -log_poly(1.0f - rs) / sf.sigma;
      if (dist < h.t) {
        T = T * sf.scol;
        pm = madd3(d, dist, o);  // no offset: the point is inside the object
        hg_g = sf.hg;
        scattered = true;
      }
    }
    if (!scattered && sf.opacity < 1.0f) {  // §7.1d: the surface is skipped with probability 1 - opacity (one extra random number, drawn only here)
      float ro = rng_next(&rng);
      if (!(ro < sf.opacity)) { o = madd3(sf.ng, -s->ray_eps, sf.P); continue; }
    }
    // a vertex with next-event estimation: a surface (BSDF, cosine, offset origin) or — §7.1f — a scattering point (Henyey-Greenstein
    // phase function: value = pdf, no cosine, the connection starts at the point itself and is attenuated on its way out by §7.1g)
    if (!scattered) {
      V3 em = ld3(sf.m.emission);
      if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) L = L + T * em;
    }
    V3 wo = -d;
    // next-event estimation: one light
    if (nl > 0) {
      float rl = rng_next(&rng), r1 = rng_next(&rng), r2 = rng_next(&rng);
      uint32_t idx = std::min((uint32_t)(rl * (float)nl), nl - 1);
      LightSample ls = sample_light(s->lights[idx], pm, r1, r2);
      if (ls.valid) {
        V3 fb; float pdf_b;
        if (scattered) { pdf_b = hg_phase(hg_g, dot3(d, ls.wi)); fb = v3s(pdf_b); }
        else bsdf_eval(sf.m, sf.base, sf.tr, wo, ls.wi, sf.ns, &fb, &pdf_b);
        if (pdf_b > 0.0f) {
          float side = dot3(ls.wi, sf.ng) >= 0.0f ? s->ray_eps : -s->ray_eps;
          V3 so = scattered ? pm : madd3(sf.ng, side, sf.P);
          float tmax = ls.dist >= kTMax ? kTMax : maxf(ls.dist - 2.0f * s->ray_eps, 0.0f);
          st->rays_shadow++;
          V3 trans;
          bool occ = scene_trace_any(s, so, ls.wi, 0.0f, tmax, pcg_hash(rng ^ kAnyKeyLight), ctr, &trans);  // §7.1d: the connection's key
          if (!occ) {
            float cosl = scattered ? 1.0f : fabsf(dot3(sf.ns, ls.wi));
            V3 contrib;
            if (ls.delta) contrib = fb * ls.le * (cosl * (float)nl);
            else {
              float pl = ls.pdf * (1.0f / (float)nl);
              float w = power_heuristic(pl, pdf_b);
              contrib = fb * ls.le * (cosl * w / pl);
            }
            V3 tc = T * contrib;
            if (trans.x != 1.0f || trans.y != 1.0f || trans.z != 1.0f) tc = tc * trans;  // §7.1g: what the media the connection crossed leave of it
            L = L + tc;
          }
        }
      }
    }
    // next-event estimation: environment map
    if (f.env_type == 1) {
      float r1 = rng_next(&rng), r2 = rng_next(&rng);
      V3 wi; float pdf_e;
      if (env_map_sample(f, r1, r2, &wi, &pdf_e)) {
        V3 fb; float pdf_b;
        if (scattered) { pdf_b = hg_phase(hg_g, dot3(d, wi)); fb = v3s(pdf_b); }
        else bsdf_eval(sf.m, sf.base, sf.tr, wo, wi, sf.ns, &fb, &pdf_b);
        if (pdf_b > 0.0f) {
          float side = dot3(wi, sf.ng) >= 0.0f ? s->ray_eps : -s->ray_eps;
          V3 so = scattered ? pm : madd3(sf.ng, side, sf.P);
          st->rays_shadow++;
          V3 trans;
          bool occ = scene_trace_any(s, so, wi, 0.0f, kTMax, pcg_hash(rng ^ kAnyKeyEnv), ctr, &trans);
          if (!occ) {
            float cosl = scattered ? 1.0f : fabsf(dot3(sf.ns, wi));
            float w = power_heuristic(pdf_e, pdf_b);
            V3 col = env_map_eval(f, wi);
            V3 tc = T * (fb * col * (cosl * w / pdf_e));
            if (trans.x != 1.0f || trans.y != 1.0f || trans.z != 1.0f) tc = tc * trans;
            Le = Le + tc;  // RENDER_SPEC 6: the environment connections have their own running sum
          }
        }
      }
    }
    // continue the path
    V3 wi;
    if (scattered) {
      float r1 = rng_next(&rng), r2 = rng_next(&rng);
      wi = hg_sample(d, hg_g, r1, r2);
      prev_pdf = hg_phase(hg_g, dot3(d, wi));  // the phase function is sampled exactly: throughput unchanged, the pdf goes to the next vertex's MIS weight
    } else {
      float r1 = rng_next(&rng), r2 = rng_next(&rng), r3 = rng_next(&rng);
      V3 fb; float pdf_b;
      if (!bsdf_sample(sf.m, sf.base, sf.tr, wo, sf.ns, r1, r2, r3, &wi, &fb, &pdf_b)) break;
      T = T * fb * (fabsf(dot3(sf.ns, wi)) / pdf_b);
      prev_pdf = pdf_b;
    }
    if (depth >= f.p.rr_depth) {
      float q = minf(max3f(T), 0.95f);
      float rr = rng_next(&rng);
      if (!(rr < q)) break;
      T = T * (1.0f / q);
    }
    float side = dot3(wi, sf.ng) >= 0.0f ? s->ray_eps : -s->ray_eps;
    o = scattered ? pm : madd3(sf.ng, side, sf.P);
    d = wi;
  }
  if (f.env_type == 1u) L = L + Le;
  if (!(std::isfinite(L.x) && std::isfinite(L.y) && std::isfinite(L.z))) L = v3s(0.0f);
  out.L = L;
  return out;
}

static inline void fold(float* img, size_t i, V3 v, uint32_t frame_index) {
  // RENDER_SPEC §8: mean_new = (mean_old * n + x) / (n + 1)
  float n = (float)frame_index, n1 = (float)(frame_index + 1);
  if (frame_index == 0) { img[4 * i] = v.x; img[4 * i + 1] = v.y; img[4 * i + 2] = v.z; }
  else {
    img[4 * i] = (img[4 * i] * n + v.x) / n1;
    img[4 * i + 1] = (img[4 * i + 1] * n + v.y) / n1;
    img[4 * i + 2] = (img[4 * i + 2] * n + v.z) / n1;
  }
  img[4 * i + 3] = 1.0f;
}

static Frame make_frame(const orc_scene* s, const orc_render_params* p) {
  Frame f;
  f.s = s; f.p = *p;
  f.res_w = (float)p->width; f.res_h = (float)p->height;
  f.aspect = f.res_w / f.res_h;
  float sn, cs;
  sincos_rad(0.5f * s->cameras[0].yfov, &sn, &cs);
  f.tan_half = sn / cs;
  f.pixel_spread = 2.0f * f.tan_half / f.res_h;
  f.env_rotation = p->env_rotation_degrees / 360.0f;
  f.env_type = s->env.width > 0 ? 1u : 0u;
  return f;
}

}  // namespace

extern "C" void orc_render(const orc_scene* s, const orc_render_params* p, uint32_t first_frame, uint32_t frame_count,
                           uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float* accum, float* albedo, float* normal,
                           float* final_rgba, orc_render_stats* stats) {
  Frame f = make_frame(s, p);
  orc_render_stats total{};
#ifdef _OPENMP
  int nt = p->num_threads > 0 ? p->num_threads : omp_get_max_threads();
#else
  int nt = 1;
#endif
  for (uint32_t fi = first_frame; fi < first_frame + frame_count; ++fi) {
    uint64_t rc = 0, rs = 0, cn = 0, ct = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt) reduction(+ : rc, rs, cn, ct)
    for (int64_t y = y0; y < (int64_t)y1; ++y) {
      orc_render_stats st{};
      Counters ctr;
      for (uint32_t x = x0; x < x1; ++x) {
        PixelOut po = trace_path(f, x, (uint32_t)y, fi, &st, &ctr);
        size_t i = (size_t)y * p->width + x;
        fold(accum, i, po.L, fi);
        fold(albedo, i, po.albedo, fi);
        fold(normal, i, po.normal, fi);
        if (final_rgba) {
          V3 c = v3(accum[4 * i], accum[4 * i + 1], accum[4 * i + 2]) * p->exposure_value;
          c = tonemap_select(c, p->enable_tonemap, p->enable_aces, p->use_simple_aces);
          final_rgba[4 * i] = c.x; final_rgba[4 * i + 1] = c.y; final_rgba[4 * i + 2] = c.z; final_rgba[4 * i + 3] = 1.0f;
        }
      }
      rc += st.rays_closest; rs += st.rays_shadow; cn += ctr.nodes; ct += ctr.tris;
    }
    total.rays_closest += rc; total.rays_shadow += rs; total.nodes_visited += cn; total.triangles_tested += ct;
  }
  if (stats) *stats = total;
}

extern "C" void orc_generate_camera_rays(const orc_scene* s, uint32_t width, uint32_t height, uint32_t frame_index, orc_ray* rays) {
  orc_render_params p{};
  p.width = width; p.height = height;
  Frame f = make_frame(s, &p);
  for (uint32_t y = 0; y < height; ++y)
    for (uint32_t x = 0; x < width; ++x) {
      uint32_t rng = rng_init(y * width + x, frame_index);
      V3 o, d;
      camera_ray(f, x, y, &rng, &o, &d);
      orc_ray& r = rays[(size_t)y * width + x];
      r.origin[0] = o.x; r.origin[1] = o.y; r.origin[2] = o.z; r.tmin = 0.0f;
      r.direction[0] = d.x; r.direction[1] = d.y; r.direction[2] = d.z; r.tmax = kTMax;
    }
}

extern "C" int orc_scene_texture_info(const orc_scene* s, uint32_t tex, uint32_t* w, uint32_t* h, uint32_t* mips) {
  if (tex >= s->texture_image.size()) return 1;
  const Image& img = s->images[s->texture_image[tex]];
  *w = img.width; *h = img.height; *mips = img.mips;
  return 0;
}
extern "C" void orc_scene_texture_level(const orc_scene* s, uint32_t tex, uint32_t level, float* out) {
  const std::vector<float>& l = s->images[s->texture_image[tex]].levels[level];
  memcpy(out, l.data(), l.size() * sizeof(float));
}
extern "C" void orc_scene_sample_texture(const orc_scene* s, uint32_t tex, const float* uvl, uint32_t n, float* out) {
  for (uint32_t i = 0; i < n; ++i) {
    C4 c = tex_sample(s, tex, uvl[3 * i], uvl[3 * i + 1], uvl[3 * i + 2]);
    out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
  }
}

extern "C" void orc_tile_assignment(uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t* owner, uint32_t* slot) {
  // RENDER_SPEC §9: tiles are dealt round-robin in a scrambled order. perm(t) = (t * A + B) mod n with A coprime to n.
  uint32_t n = tiles_x * tiles_y;
  uint32_t A = 0x9E3779B1u % n;
  if (A == 0) A = 1;
  auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
  while (gcd(A, n) != 1) ++A;
  for (uint32_t t = 0; t < n; ++t) {
    uint32_t k = (uint32_t)(((uint64_t)t * A + 7u) % n);  // position of tile t in the dealing order
    owner[t] = k % world;
    slot[t] = k / world;
  }
}
