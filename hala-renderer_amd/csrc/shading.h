// shading.h — device code for what the reference's (absent) raygen / closest-hit / miss / callable shaders
// do between two traversals (docs/RENDER_SPEC.md §5-§8).  The records consumed are exactly the reference's
// bindings: HalaGlobalUniform (src/rt_renderer.rs:44-65), cameras/lights/materials/primitives
// (src/rt_renderer.rs:141-181), vertex/index buffers by device address (gpu_uploader.rs:869-870) and the
// env tables (src/envmap.rs:239-388).
#pragma once
#include <hip/hip_runtime.h>

#include "hala_types.h"
#include "rt_math.h"

namespace rt {

// ---- §9 pixel slots -------------------------------------------------------------------------------------
// world == 1: slots in 8 x 8 pixel blocks, row-major over the blocks (hala_types.h: kPixelBlock).  world > 1: tile-major slots of this rank's tiles.
RT_DI bool slot_to_pixel(const FrameConst& fc, uint32_t slot, uint32_t* px, uint32_t* py) {
  if (fc.world <= 1u) {
    if (kPixelBlock == 0u) {
      *py = slot / fc.width;
      *px = slot - *py * fc.width;
      return true;
    }
    constexpr uint32_t kB = kPixelBlock ? kPixelBlock : 1u, kB2 = kB * kB;
    const uint32_t blk = slot / kB2, within = slot - blk * kB2;
    const uint32_t ly = within / kB, lx = within - ly * kB;
    const uint32_t by = blk / fc.blocks_x, bx = blk - by * fc.blocks_x;
    *px = bx * kB + lx;
    *py = by * kB + ly;
    return *px < fc.width && *py < fc.height;  // false: padding slot of a border block
  }
  const uint32_t ts2 = fc.tile_size * fc.tile_size;
  const uint32_t lt = slot / ts2, within = slot - lt * ts2;
  uint32_t ly, lx;
  if (kPixelBlock && fc.tile_size % kPixelBlock == 0u) {  // RENDER_SPEC §9: 8 x 8 pixel blocks inside the tile, row-major over the blocks
    constexpr uint32_t kB = kPixelBlock ? kPixelBlock : 1u, kB2 = kB * kB;
    const uint32_t per_row = fc.tile_size / kB;
    const uint32_t blk = within / kB2, j = within - blk * kB2;
    const uint32_t by = blk / per_row, bx = blk - by * per_row;
    ly = by * kB + j / kB; lx = bx * kB + (j - (j / kB) * kB);
  } else { ly = within / fc.tile_size; lx = within - ly * fc.tile_size; }
  const uint32_t n = fc.tiles_x * fc.tiles_y;
  const uint32_t k = lt * fc.world + fc.rank;  // position in the dealing order
  if (k >= n) return false;                    // padding tile
  // tile t with (t * A + B) % n == k  ->  t = ((k + n - B % n) * A^-1) % n ; perm_a holds A^-1 here
  const uint32_t t = (uint32_t)(((unsigned long long)((k + n - fc.perm_b % n) % n) * fc.perm_a) % n);
  const uint32_t ty = t / fc.tiles_x, tx = t - ty * fc.tiles_x;
  *px = tx * fc.tile_size + lx;
  *py = ty * fc.tile_size + ly;
  return *px < fc.width && *py < fc.height;
}

// ---- §5 camera ---------------------------------------------------------------------------------------------
RT_DI void camera_ray(const FrameConst& fc, const hala_gpu_camera& cam, uint32_t px, uint32_t py, uint32_t& rng, f3* o, f3* d) {
  float r1 = rng_next(rng), r2 = rng_next(rng), r3 = rng_next(rng), r4 = rng_next(rng);
  float fx = ((float)px + r1) / fc.u.resolution[0];
  float fy = ((float)py + r2) / fc.u.resolution[1];
  float ndc_x = fx * 2.0f - 1.0f;
  float ndc_y = 1.0f - fy * 2.0f;
  f3 pos = ld3(cam.position), right = ld3(cam.right), up = ld3(cam.up), fwd = ld3(cam.forward);
  if (cam.type == 0u) {
    float dx = ndc_x * fc.aspect * fc.tan_half;
    float dy = ndc_y * fc.tan_half;
    f3 dir = normalize3(madd3(up, dy, madd3(right, dx, fwd)));
    float aperture = cam.aperture_or_ymag;
    if (aperture > 0.0f) {
      float ft = cam.focal_distance_or_xmag / dot3(dir, fwd);
      f3 focus = madd3(dir, ft, pos);
      float r = aperture * sqrtf(r3);
      float s, c;
      sincos_2pi(r4, &s, &c);
      f3 org = madd3(up, r * s, madd3(right, r * c, pos));
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

// ---- §7.3 environment ----------------------------------------------------------------------------------------
RT_DI int wrapi(int i, int n) {
  // REPEAT addressing: i mod n in [0, n).  Power-of-two sizes (every mip level of a power-of-two image, the usual case) take the mask —
  // the same integer for negative i too — and skip the ~35-instruction division sequence; a wave without odd sizes never runs it.
  if ((n & (n - 1)) == 0) return i & (n - 1);
  int m = i % n;
  return m < 0 ? m + n : m;
}
// i + 1 wrapped, for 0 <= i < n: no division at all
RT_DI int wrap_next(int i, int n) { return i + 1 == n ? 0 : i + 1; }
RT_DI f3 env_texel(const SceneView& sv, int w, int x, int y) {
  const float4 p = reinterpret_cast<const float4*>(sv.env_pixels)[(size_t)y * w + x];
  return mk3(p.x, p.y, p.z);
}
RT_DI void env_dir_to_uv(const FrameConst& fc, f3 d, float* uu, float* vv) {
  float theta = acos_poly(clampf(d.y, -1.0f, 1.0f));
  float phi = atan2_poly(d.z, d.x);
  *uu = (kPi + phi) * kInvTwoPi + fc.u.env_rotation;
  *vv = theta * kInvPi;
}
RT_DI f3 env_map_eval(const FrameConst& fc, const SceneView& sv, f3 d) {
  const int W = (int)fc.u.env_map_width, H = (int)fc.u.env_map_height;
  float uu, vv;
  env_dir_to_uv(fc, d, &uu, &vv);
  float x = uu * (float)W - 0.5f, y = vv * (float)H - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = wrapi((int)x0, W), iy0 = wrapi((int)y0, H);
  int ix1 = wrap_next(ix0, W), iy1 = wrap_next(iy0, H);
  f3 c00 = env_texel(sv, W, ix0, iy0), c10 = env_texel(sv, W, ix1, iy0), c01 = env_texel(sv, W, ix0, iy1), c11 = env_texel(sv, W, ix1, iy1);
  f3 top = c00 * (1.0f - fx) + c10 * fx;
  f3 bot = c01 * (1.0f - fx) + c11 * fx;
  return (top * (1.0f - fy) + bot * fy) * fc.u.env_intensity;
}
RT_DI float env_map_pdf(const FrameConst& fc, const SceneView& sv, f3 d) {
  const int W = (int)fc.u.env_map_width, H = (int)fc.u.env_map_height;
  float uu, vv;
  env_dir_to_uv(fc, d, &uu, &vv);
  int ix = wrapi((int)floorf(uu * (float)W), W);
  int iy = min((int)(vv * (float)H), H - 1);
  float lum = luminance(env_texel(sv, W, ix, iy));
  float st = sqrtf(maxf(0.0f, 1.0f - d.y * d.y));
  if (!(st > 0.0f) || !(lum > 0.0f)) return 0.0f;
  return (lum * (float)(fc.u.env_map_width * fc.u.env_map_height)) / (fc.u.env_total_sum * kTwoPiSq * st);
}
RT_DI bool env_map_sample(const FrameConst& fc, const SceneView& sv, float r1, float r2, f3* wi, float* pdf) {
  const int W = (int)fc.u.env_map_width, H = (int)fc.u.env_map_height;
  float fy = r1 * (float)H;
  int iy = min((int)fy, H - 1);
  float mv = sv.env_marginal[iy];
  if (!(mv >= 0.0f)) mv = 0.0f;
  int row = min((int)(mv * (float)H + 0.5f), H - 1);
  float fxx = r2 * (float)W;
  int ix = min((int)fxx, W - 1);
  float cu = sv.env_conditional[(size_t)row * W + ix];
  if (!(cu >= 0.0f)) cu = 0.0f;
  int col = min((int)(cu * (float)W + 0.5f), W - 1);
  float ju = fxx - (float)ix, jv = fy - (float)iy;
  float uu = ((float)col + ju) / (float)W;
  float vv = ((float)row + jv) / (float)H;
  float t = uu - fc.u.env_rotation;
  t = t - floorf(t);
  if (t >= 1.0f) t = 0.0f;
  float sp, cp, st, ct;
  sincos_2pi(t, &sp, &cp);
  sincos_2pi(vv * 0.5f, &st, &ct);
  *wi = mk3(-st * cp, ct, -st * sp);
  float lum = luminance(env_texel(sv, W, col, row));
  if (!(st > 0.0f) || !(lum > 0.0f)) { *pdf = 0.0f; return false; }
  *pdf = (lum * (float)(fc.u.env_map_width * fc.u.env_map_height)) / (fc.u.env_total_sum * kTwoPiSq * st);
  return true;
}
RT_DI f3 sky_eval(const FrameConst& fc, f3 d) {
  float t = 0.5f * (d.y + 1.0f);
  f3 g = ld3(fc.u.ground_color), s = ld3(fc.u.sky_color);
  return (g * (1.0f - t) + s * t) * fc.u.env_intensity;
}

// ---- §7.1 materials ----------------------------------------------------------------------------------------------
struct MatView {
  f3 base, emission;
  float ax, ay;
  uint32_t type;
  // DISNEY (type 1) only
  float metallic, roughness, specular_tint, sheen, sheen_tint, clearcoat, clearcoat_roughness, ior;
  float opacity;  // §7.1d: material opacity x base-colour-map alpha
  float trans;  // §7.1c: specular_transmission * (1 - metallic)
  float eta;    // §7.1c: index of the far side relative to the side the path arrives from (ior entering, 1/ior leaving)
};

// ---- §7.1b DISNEY ----------------------------------------------------------------------------------------------
RT_DI float schlick5(float x) {
  float m = clampf(1.0f - x, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
RT_DI f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
RT_DI float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
RT_DI float ggx_d(f3 h, float ax, float ay) {
  float hx = h.x / ax, hy = h.y / ay;
  float k = hx * hx + hy * hy + h.z * h.z;
  return 1.0f / (kPi * ax * ay * k * k);
}
RT_DI float ggx_g1(f3 w, float ax, float ay) {
  float a = ax * w.x, b = ay * w.y;
  float l = (a * a + b * b) / (w.z * w.z);
  return 2.0f / (1.0f + sqrtf(1.0f + l));
}
RT_DI f3 ggx_sample_vndf(f3 v, float ax, float ay, float r1, float r2) {
  f3 vh = normalize3(mk3(ax * v.x, ay * v.y, v.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  f3 t1 = mk3(1.0f, 0.0f, 0.0f);
  if (lensq > 0.0f) { float il = 1.0f / sqrtf(lensq); t1 = mk3(-vh.y * il, vh.x * il, 0.0f); }
  f3 t2 = cross3(vh, t1);
  float r = sqrtf(r1);
  float s, c;
  sincos_2pi(r2, &s, &c);
  float a = r * c, b = r * s;
  float sn = 0.5f * (1.0f + vh.z);
  b = (1.0f - sn) * sqrtf(maxf(0.0f, 1.0f - a * a)) + sn * b;
  float cz = sqrtf(maxf(0.0f, 1.0f - a * a - b * b));
  f3 nh = t1 * a + t2 * b + vh * cz;
  return normalize3(mk3(ax * nh.x, ay * nh.y, maxf(0.0f, nh.z)));
}
struct DisneyLobes {
  float pd, ps, pc, pt;
  f3 cspec0, csheen;
  float cc_alpha;
};
// §7.1c exact unpolarised Fresnel reflectance of a dielectric interface; c = |cos| on the arriving side
RT_DI float fresnel_dielectric(float c, float eta) {
  float g2 = eta * eta - 1.0f + c * c;
  if (!(g2 > 0.0f)) return 1.0f;  // total internal reflection
  float g = sqrtf(g2);
  float a = (g - c) / (g + c);
  float b = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
  return 0.5f * a * a * (1.0f + b * b);
}
RT_DI DisneyLobes disney_lobes(const MatView& m, float nv) {
  DisneyLobes d;
  float lb = luminance(m.base);
  f3 tint = lb > 0.0f ? m.base * (1.0f / lb) : splat3(1.0f);
  float f0 = (m.ior - 1.0f) / (m.ior + 1.0f);
  f0 = f0 * f0;
  d.cspec0 = mix3(mix3(splat3(1.0f), tint, m.specular_tint) * f0, m.base, m.metallic);
  d.csheen = mix3(splat3(1.0f), tint, m.sheen_tint);
  d.cc_alpha = maxf(0.001f, m.clearcoat_roughness * m.clearcoat_roughness);
  float fv = schlick5(nv);
  float wd = (1.0f - m.metallic) * lb * (1.0f - m.trans);
  float ws = luminance(mix3(d.cspec0, splat3(1.0f), fv));
  float wc = 0.25f * m.clearcoat * mixf(0.04f, 1.0f, fv);
  float wt = 0.0f;
  if (m.trans > 0.0f) {  // §7.1c: the transmissive share reflects by the exact dielectric Fresnel term (1 beyond the critical angle)
    const float fe = fresnel_dielectric(nv, m.eta);
    ws = mixf(ws, fe, m.trans);
    wt = m.trans * (1.0f - fe);
  }
  float sum = wd + ws + wc + wt;
  if (!(sum > 0.0f)) { d.pd = d.ps = d.pc = d.pt = 0.0f; return d; }
  float inv = 1.0f / sum;
  d.pd = wd * inv; d.ps = ws * inv; d.pc = wc * inv; d.pt = wt * inv;
  return d;
}
// What every evaluation of a vertex's BSDF shares — the light connection, the environment connection and the sampled continuation all
// look at the same (material, wo, n): the tangent frame, wo in it, the lobe weights and the wo-side masking terms are computed ONCE per
// vertex (the same expressions as before, so the same bits) instead of up to three times.
struct BsdfCtx {
  f3 t, b, lo;   // onb(n) and wo in that frame (lo.z = nv)
  float nv;
  DisneyLobes d; // DISNEY only (valid when nv > 0)
  float fv, g1o, c1o;
};
RT_DI BsdfCtx bsdf_prepare(const MatView& m, f3 wo, f3 n) {
  BsdfCtx c;
  c.nv = dot3(n, wo);
  onb(n, &c.t, &c.b);
  c.lo = mk3(dot3(wo, c.t), dot3(wo, c.b), c.nv);
  c.fv = 0.0f; c.g1o = 0.0f; c.c1o = 0.0f;
  c.d.pd = c.d.ps = c.d.pc = c.d.pt = 0.0f; c.d.cspec0 = splat3(0.0f); c.d.csheen = splat3(0.0f); c.d.cc_alpha = 0.0f;
  if (m.type == 1u && c.nv > 0.0f) {
    c.d = disney_lobes(m, c.nv);
    c.fv = schlick5(c.nv);
    c.g1o = ggx_g1(c.lo, m.ax, m.ay);
    c.c1o = ggx_g1(c.lo, c.d.cc_alpha, c.d.cc_alpha);
  }
  return c;
}
RT_DI void disney_eval(const MatView& m, const BsdfCtx& c, f3 wi, f3 n, f3* f, float* pdf) {
  const float nl = dot3(n, wi), nv = c.nv;
  *f = splat3(0.0f); *pdf = 0.0f;
  if (!(nv > 0.0f)) return;
  const f3 lo = c.lo, li = mk3(dot3(wi, c.t), dot3(wi, c.b), nl);
  const DisneyLobes& d = c.d;
  if (nl < 0.0f) {  // §7.1c: refraction through the microfacet with half vector h = -(lo + eta*li), flipped to the upper side
    if (!(m.trans > 0.0f)) return;
    f3 h = lo + li * m.eta;
    float h2 = dot3(h, h);
    if (!(h2 > 0.0f)) return;
    h = h * (1.0f / sqrtf(h2));
    if (h.z < 0.0f) h = -h;
    float odh = dot3(lo, h), idh = dot3(li, h);
    if (!(odh > 0.0f && idh < 0.0f)) return;  // both directions must see the front / the back of the same facet
    float fr = fresnel_dielectric(odh, m.eta);
    float ds = ggx_d(h, m.ax, m.ay);
    float g1o = c.g1o, g1i = ggx_g1(li, m.ax, m.ay);
    float den = odh + m.eta * idh;
    float jac = m.eta * m.eta * (-idh) / (den * den);  // |dh/dwi|
    float w = m.trans * (1.0f - fr) * ds * g1o * g1i * odh * jac / (-nl * nv);
    *f = mk3(sqrtf(m.base.x), sqrtf(m.base.y), sqrtf(m.base.z)) * w;
    *pdf = d.pt * (g1o * odh * ds / nv) * jac;
    return;
  }
  if (!(nl > 0.0f)) return;
  f3 h = normalize3(lo + li);
  float ldh = dot3(li, h);
  float fl = schlick5(nl), fv = c.fv, fh = schlick5(ldh);
  float fd90 = 0.5f + 2.0f * sqrtf(m.roughness) * ldh * ldh;
  float fd = mixf(1.0f, fd90, fl) * mixf(1.0f, fd90, fv);
  float dw = (1.0f - m.metallic) * (1.0f - m.trans);
  f3 fs = mix3(d.cspec0, splat3(1.0f), fh);
  if (m.trans > 0.0f) fs = mix3(fs, splat3(fresnel_dielectric(ldh, m.eta)), m.trans);  // §7.1c: pairs with the (1 - F) of the refraction lobe
  f3 fr = m.base * (splat3(1.0f) - fs) * (kInvPi * fd * dw) + d.csheen * (m.sheen * fh * dw);
  float ds = ggx_d(h, m.ax, m.ay);
  float g1o = c.g1o, g1i = ggx_g1(li, m.ax, m.ay);
  float denom = 4.0f * nl * nv;
  fr = fr + fs * (ds * g1o * g1i / denom);
  float dc = ggx_d(h, d.cc_alpha, d.cc_alpha);
  float c1o = c.c1o, c1i = ggx_g1(li, d.cc_alpha, d.cc_alpha);
  float fc = mixf(0.04f, 1.0f, fh);
  fr = fr + splat3(0.25f * m.clearcoat * fc * dc * c1o * c1i / denom);
  *f = fr;
  float inv4nv = 1.0f / (4.0f * nv);
  *pdf = d.pd * (nl * kInvPi) + d.ps * (g1o * ds * inv4nv) + d.pc * (c1o * dc * inv4nv);
}

RT_DI void bsdf_eval(const MatView& m, const BsdfCtx& c, f3 wo, f3 wi, f3 n, f3* f, float* pdf) {
  if (m.type == 1u) { disney_eval(m, c, wi, n, f, pdf); return; }
  float nl = dot3(n, wi), nv = c.nv;
  if (!(nl > 0.0f && nv > 0.0f)) { *f = splat3(0.0f); *pdf = 0.0f; return; }
  float s = dot3(wi, wo) - nl * nv;
  float tterm = maxf(0.0f, s) / maxf(nl, nv);
  float k = kInvPi * (m.ax + m.ay * tterm);
  *f = m.base * k;
  *pdf = nl * kInvPi;
}
RT_DI bool bsdf_sample(const MatView& m, const BsdfCtx& c, f3 wo, f3 n, float r1, float r2, float r3, f3* wi, f3* f, float* pdf) {
  const f3 t = c.t, b = c.b;
  if (m.type == 1u) {
    float nv = c.nv;
    if (!(nv > 0.0f)) { *f = splat3(0.0f); *pdf = 0.0f; return false; }
    const DisneyLobes& d = c.d;
    if (r3 < d.pd) {
      *wi = to_world(cosine_hemisphere(r1, r2), t, b, n);
    } else if (r3 >= d.pd + d.ps + d.pc) {  // §7.1c: refract through a VNDF-sampled facet (nothing when totally reflected)
      f3 lo = c.lo;
      f3 h = ggx_sample_vndf(lo, m.ax, m.ay, r1, r2);
      float cc = dot3(lo, h);
      float ie = 1.0f / m.eta;
      float k = 1.0f - (1.0f - cc * cc) * (ie * ie);
      if (!(k > 0.0f)) { *f = splat3(0.0f); *pdf = 0.0f; return false; }
      f3 li = h * (cc * ie - sqrtf(k)) - lo * ie;
      *wi = to_world(li, t, b, n);
    } else {
      bool spec = r3 < d.pd + d.ps;
      float ax = spec ? m.ax : d.cc_alpha, ay = spec ? m.ay : d.cc_alpha;
      f3 lo = c.lo;
      f3 h = ggx_sample_vndf(lo, ax, ay, r1, r2);
      float k = 2.0f * dot3(lo, h);
      f3 li = h * k - lo;
      *wi = to_world(li, t, b, n);
    }
    disney_eval(m, c, *wi, n, f, pdf);
    return *pdf > 0.0f;
  }
  f3 l = cosine_hemisphere(r1, r2);
  *wi = to_world(l, t, b, n);
  bsdf_eval(m, c, wo, *wi, n, f, pdf);
  return *pdf > 0.0f;
}

// ---- §7.2 lights --------------------------------------------------------------------------------------------------
struct LightSample {
  f3 wi, le;
  float dist, pdf;
  bool delta, valid;
};
RT_DI LightSample sample_light(const hala_gpu_light& l, f3 P, float r1, float r2) {
  LightSample s;
  s.valid = false; s.delta = true; s.pdf = 0.0f; s.dist = kTMax; s.le = splat3(0.0f); s.wi = mk3(0.0f, 1.0f, 0.0f);
  f3 inten = ld3(l.intensity), pos = ld3(l.position);
  if (l.type == 0u || l.type == 2u) {  // POINT, SPOT
    f3 to = pos - P;
    float d2 = dot3(to, to);
    if (!(d2 > 0.0f)) return s;
    float dist = sqrtf(d2);
    s.wi = to * (1.0f / dist);
    s.dist = dist;
    s.le = inten * (1.0f / d2);
    if (l.type == 2u) {
      f3 axis = normalize3(ld3(l.u));
      float cosang = -dot3(s.wi, axis);
      float ci = l.v[0], co = l.v[1];
      float t;
      if (ci > co) t = clampf((cosang - co) / (ci - co), 0.0f, 1.0f); else t = cosang >= co ? 1.0f : 0.0f;
      float sm = t * t * (3.0f - 2.0f * t);
      s.le = s.le * sm;
    }
    s.valid = true;
  } else if (l.type == 1u) {  // DIRECTIONAL
    f3 axis = normalize3(-ld3(l.u));
    float cosmax = l.v[0];
    if (cosmax >= 1.0f) s.wi = axis;
    else {
      float ct = 1.0f - r1 * (1.0f - cosmax);
      float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
      float sp, cp;
      sincos_2pi(r2, &sp, &cp);
      f3 t, b;
      onb(axis, &t, &b);
      s.wi = to_world(mk3(st * cp, st * sp, ct), t, b, axis);
    }
    s.dist = kTMax;
    s.le = inten;
    s.valid = true;
  } else if (l.type == 3u) {  // QUAD
    f3 u = ld3(l.u), v = ld3(l.v);
    f3 pt = madd3(v, r2, madd3(u, r1, pos));
    f3 n = normalize3(cross3(u, v));
    f3 to = pt - P;
    float d2 = dot3(to, to);
    if (!(d2 > 0.0f)) return s;
    float dist = sqrtf(d2);
    s.wi = to * (1.0f / dist);
    float cosl = -dot3(s.wi, n);
    if (!(cosl > 0.0f)) return s;
    s.dist = dist; s.le = inten; s.pdf = d2 / (l.area * cosl); s.delta = false; s.valid = true;
  } else if (l.type == 4u) {  // SPHERE
    float z = 1.0f - 2.0f * r1;
    float rr = sqrtf(maxf(0.0f, 1.0f - z * z));
    float sp, cp;
    sincos_2pi(r2, &sp, &cp);
    f3 nl = mk3(rr * cp, rr * sp, z);
    f3 pt = madd3(nl, l.radius, pos);
    f3 to = pt - P;
    float d2 = dot3(to, to);
    if (!(d2 > 0.0f)) return s;
    float dist = sqrtf(d2);
    s.wi = to * (1.0f / dist);
    float cosl = -dot3(s.wi, nl);
    if (!(cosl > 0.0f)) return s;
    s.dist = dist; s.le = inten; s.pdf = d2 / (l.area * cosl); s.delta = false; s.valid = true;
  }
  return s;
}
// analytic intersection of the hittable lights (the light BLAS of gpu_uploader.rs:818-840 + its intersection shader)
RT_DI float intersect_light(const hala_gpu_light& l, f3 o, f3 d, float* pdf) {
  f3 pos = ld3(l.position);
  if (l.type == 3u) {
    f3 u = ld3(l.u), v = ld3(l.v);
    f3 n = normalize3(cross3(u, v));
    float dn = dot3(d, n);
    if (!(dn < 0.0f)) return -1.0f;
    float t = dot3(pos - o, n) / dn;
    if (!(t > 0.0f)) return -1.0f;
    f3 hp = madd3(d, t, o) - pos;
    float a = dot3(hp, u) / dot3(u, u), b = dot3(hp, v) / dot3(v, v);
    if (!(a >= 0.0f && a <= 1.0f && b >= 0.0f && b <= 1.0f)) return -1.0f;
    *pdf = (t * t) / (l.area * (-dn));
    return t;
  }
  if (l.type == 4u) {
    f3 oc = o - pos;
    float b = dot3(oc, d);
    float c = dot3(oc, oc) - l.radius * l.radius;
    float disc = b * b - c;
    if (!(disc > 0.0f)) return -1.0f;
    float t = -b - sqrtf(disc);
    if (!(t > 0.0f)) return -1.0f;
    f3 nl = (madd3(d, t, o) - pos) * (1.0f / l.radius);
    float cosl = -dot3(d, nl);
    if (!(cosl > 0.0f)) return -1.0f;
    *pdf = (t * t) / (l.area * cosl);
    return t;
  }
  return -1.0f;
}

// ---- §6 surface reconstruction ------------------------------------------------------------------------------------------
RT_DI f3 transform_normal(const float* m, f3 n) {
  f3 c0 = mk3(m[0], m[1], m[2]), c1 = mk3(m[4], m[5], m[6]), c2 = mk3(m[8], m[9], m[10]);
  f3 k0 = cross3(c1, c2), k1 = cross3(c2, c0), k2 = cross3(c0, c1);
  float det = dot3(c0, k0);
  f3 r = mk3(__fmaf_rn(k2.x, n.z, __fmaf_rn(k1.x, n.y, k0.x * n.x)), __fmaf_rn(k2.y, n.z, __fmaf_rn(k1.y, n.y, k0.y * n.x)),
             __fmaf_rn(k2.z, n.z, __fmaf_rn(k1.z, n.y, k0.z * n.x)));
  return det < 0.0f ? -r : r;
}

// ---- §7.4 textures: software trilinear fetch (K8; CDNA exposes no image sampler to HIP that could be relied on) ------
// log2 of a positive finite float from its bit pattern: exponent + (mantissa - 1); piecewise linear, exact at powers
// of two, max error 0.086 — plenty for level-of-detail selection and identical on every implementation.
RT_DI float log2_approx(float x) {
  const uint32_t b = __float_as_uint(x);
  const float e = (float)((int)((b >> 23) & 255u) - 127);
  const float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
  return e + (m - 1.0f);
}
// Texel decode table (512 floats): [0, 256) the sRGB EOTF, [256, 512) b / 255.  The shade kernel keeps a copy in LDS (a table read is
// then a ds_read instead of one more request to the texture-address unit, which is what that kernel is short of); LUT is `const
// float*` (global) or `const RT_LDS float*`.
constexpr uint32_t kTexLutEntries = 512;
template <typename LUT>
RT_DI float4 tex8_fetch(LUT lut, const uint32_t* base, uint32_t format, int x, int y, int w) {
  const uint32_t t = base[tex_tiled_index((uint32_t)x, (uint32_t)y, (uint32_t)w)];
  const uint32_t o = format == kTexSrgb8 ? 0u : 256u;
  return make_float4(lut[o + (t & 0xffu)], lut[o + ((t >> 8) & 0xffu)], lut[o + ((t >> 16) & 0xffu)], lut[256u + (t >> 24)]);
}
template <typename LUT>
RT_DI float4 tex_bilinear(const SceneView& sv, LUT lut, const TexDesc& td, uint32_t level, float u, float v) {
  const int w = max((int)(td.width >> level), 1), h = max((int)(td.height >> level), 1);
  const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int ix0 = wrapi((int)x0, w), iy0 = wrapi((int)y0, h);  // REPEAT (gpu_uploader.rs:346)
  const int ix1 = wrap_next(ix0, w), iy1 = wrap_next(iy0, h);
  float4 c00, c10, c01, c11;
  if (td.format == kTexFloat) {
    const float4* base = sv.tex_arena + td.mip_offset[level];
    c00 = base[(size_t)iy0 * w + ix0]; c10 = base[(size_t)iy0 * w + ix1];
    c01 = base[(size_t)iy1 * w + ix0]; c11 = base[(size_t)iy1 * w + ix1];
  } else {  // 8-bit texels, tiled 4x4, decoded here (RENDER_SPEC 7.4)
    const uint32_t* base = sv.tex_arena8 + td.mip_offset[level];
    c00 = tex8_fetch(lut, base, td.format, ix0, iy0, w); c10 = tex8_fetch(lut, base, td.format, ix1, iy0, w);
    c01 = tex8_fetch(lut, base, td.format, ix0, iy1, w); c11 = tex8_fetch(lut, base, td.format, ix1, iy1, w);
  }
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  float4 r;
  r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
  r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
  r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
  r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
  return r;
}
// linear / linear-mip / repeat sampler of gpu_uploader.rs:341-353; lod is clamped to the chain
template <typename LUT>
RT_DI float4 tex_sample(const SceneView& sv, LUT lut, uint32_t tex, float u, float v, float lod) {
  const TexDesc& td = sv.textures[tex];
  const float top = (float)(td.mips - 1u);
  lod = lod < 0.0f ? 0.0f : (lod > top ? top : lod);
  const float l0 = floorf(lod);
  const float fl = lod - l0;
  const uint32_t level = (uint32_t)l0;
  float4 a = tex_bilinear(sv, lut, td, level, u, v);
  if (fl > 0.0f && level + 1u < td.mips) {
    const float4 b = tex_bilinear(sv, lut, td, level + 1u, u, v);
    const float g = 1.0f - fl;
    a.x = a.x * g + b.x * fl; a.y = a.y * g + b.y * fl; a.z = a.z * g + b.z * fl; a.w = a.w * g + b.w * fl;
  }
  return a;
}
// level of detail for texture `tex` given lod_base = 0.5*log2(footprint^2 * uv_area / world_area) (RENDER_SPEC §7.4)
RT_DI float tex_lod(const SceneView& sv, uint32_t tex, float lod_base) {
  const TexDesc& td = sv.textures[tex];
  return lod_base + 0.5f * log2_approx((float)td.width * (float)td.height);
}

// two-level trees (RENDER_SPEC 4.5): the instance a global triangle id belongs to (the last one whose first triangle is <= id; the
// table is tiny and hot) — shading records are then found by instance: shade_base + (id - first_tri)
RT_DI uint32_t instance_of(const SceneView& sv, uint32_t gid) {
  uint32_t lo = 0, hi = sv.instance_count;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sv.inst_info[mid].first_tri <= gid) lo = mid; else hi = mid;
  }
  return lo;
}
// the shading record of a triangle met by the traversal of a two-level tree: inside an instance the lane carries the primitive's first
// record (bit 31 set: "inside"), local = the triangle's number inside the primitive; in the world tree the instance is looked up
RT_DI uint32_t hit_record_of(const SceneView& sv, uint32_t gid, uint32_t shade_base_flagged, uint32_t local) {
  if (shade_base_flagged >> 31) return (shade_base_flagged & 0x7fffffffu) + local;
  const InstInfo ii = sv.inst_info[instance_of(sv, gid)];
  return ii.shade_base + (gid - ii.first_tri);
}
// RENDER_SPEC 7.1d: what decides whether a translucent triangle blocks an any-hit ray: the material's opacity times the alpha of its
// base-colour map (bilinear fetch of level 0 at the hit's interpolated texture coordinates); prim = its shading record
RT_DI float hit_alpha(const SceneView& sv, uint32_t prim, float u, float v) {
  const float4* sp = reinterpret_cast<const float4*>(sv.shade_tris + prim);
  const uint32_t material = __float_as_uint(sp[1].w);
  const hala_gpu_material& m = sv.materials[material];
  float alpha = m.opacity;
  const uint32_t tex = m.base_color_map_index;
  if (tex < sv.texture_count) {
    const float4 s4 = sp[4], s5 = sp[5];
    const float w0 = 1.0f - u - v;
    const float tu = __fmaf_rn(s5.x, v, __fmaf_rn(s4.z, u, s4.x * w0));
    const float tv = __fmaf_rn(s5.y, v, __fmaf_rn(s4.w, u, s4.y * w0));
    alpha = alpha * tex_bilinear(sv, sv.tex_lut, sv.textures[tex], 0u, tu, tv).w;
  }
  return alpha;
}
// RENDER_SPEC 7.1d / 7.1g: what a flagged triangle of the any-hit copy does to an any-hit ray that hits it inside (tmin, tmax) at distance t
// with Moeller-Trumbore determinant det.  flag bit 0: translucent — blocks iff hash(key, triangle) < opacity x alpha at the hit; bit 1:
// boundary of a medium — a ray that gets through adds +sigma t to its optical depth where it leaves the object (hit from behind:
// det < 0), -sigma t where it enters, per channel, in 2^-16 units with wrap-around integer arithmetic: the sums do not depend on the
// order of the crossings.  Returns true if the triangle blocks; q = what to add to the ray's optical depth.
// prim = the triangle's global id (what the hash is keyed by), rec = its shading record (== prim in one-level trees)
RT_DI bool any_hit_event(const SceneView& sv, uint32_t key, uint32_t prim, uint32_t rec, uint32_t flag, float t, float det, float u, float v, uint32_t q[3]) {
  q[0] = q[1] = q[2] = 0u;
  if (flag & 1u) {
    const float x = (float)(pcg_hash(key + prim * 0x9E3779B1u) >> 8) * (1.0f / 16777216.0f);
    if (x < hit_alpha(sv, rec, u, v)) return true;
  }
  if (flag & 2u) {
    const uint32_t material = __float_as_uint(reinterpret_cast<const float4*>(sv.shade_tris + rec)[1].w);
    const hala_gpu_material& m = sv.materials[material];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sigma = m.medium_type == 1u ? m.medium_density * (1.0f - m.medium_color[c]) : m.medium_density;
      const uint32_t v32 = (uint32_t)(int32_t)floorf(minf(t * sigma, 4096.0f) * 65536.0f + 0.5f);
      q[c] = det < 0.0f ? v32 : 0u - v32;
    }
  }
  return false;
}
// what an unblocked connection keeps after the media it crossed: exp_neg(-max(tau, 0) / 65536) per channel
RT_DI f3 any_transmittance(const uint32_t tau[3]) {
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int32_t qv = (int32_t)tau[c];
    r[c] = qv > 0 ? exp_neg_poly(-((float)qv * (1.0f / 65536.0f))) : 1.0f;
  }
  return mk3(r[0], r[1], r[2]);
}

struct Surface {
  f3 P, ns, ng;
  MatView mat;
  f3 absorb, glow;  // §7.1e: transmittance / emitted radiance of the medium the segment that ends here ran through
  float sigma, hg;  // §7.1f: scattering coefficient (0: none) and Henyey-Greenstein g of that medium
  f3 scol;          //        its single-scattering albedo
};
RT_DI f3 transform_vector(const float* m, f3 p) {
  return mk3(__fmaf_rn(m[8], p.z, __fmaf_rn(m[4], p.y, m[0] * p.x)), __fmaf_rn(m[9], p.z, __fmaf_rn(m[5], p.y, m[1] * p.x)),
             __fmaf_rn(m[10], p.z, __fmaf_rn(m[6], p.y, m[2] * p.x)));
}
// SIMPLE: the scene's materials are all untextured, opaque DIFFUSE ones without a medium (the host checks: SceneView::simple_materials);
// the compiler then drops the texture, Disney and medium code from the kernel variant — half the registers, twice the waves.
template <bool SIMPLE, typename LUT>
RT_DI Surface make_surface(const SceneView& sv, LUT lut, float pixel_spread, f3 o, f3 d, float t, float u, float v, uint32_t prim) {
  Surface sf;
  // two-level trees (RENDER_SPEC 4.5): shading records are per PRIMITIVE triangle, found through the instance of the hit's global id
  uint32_t rec = prim, inst_tl = 0;
  bool instanced = false;
  if (sv.two_level) {
    inst_tl = instance_of(sv, prim);
    const InstInfo ii = sv.inst_info[inst_tl];
    rec = ii.shade_base + (prim - ii.first_tri);
    instanced = ii.instanced != 0u;
  }
  // line 1 of the 128-B shading record: geometric normal, vertex normals, instance, material
  const float4* sp = reinterpret_cast<const float4*>(sv.shade_tris + rec);
  const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
  const uint32_t inst = sv.two_level ? inst_tl : __float_as_uint(s0.w), material = __float_as_uint(s1.w);
  const hala_gpu_mesh_data& md = sv.primitives[inst];
  float w0 = 1.0f - u - v;
  f3 nl = madd3(mk3(s3.x, s3.y, s3.z), v, madd3(mk3(s2.x, s2.y, s2.z), u, mk3(s1.x, s1.y, s1.z) * w0));
  sf.ns = normalize3(transform_normal(md.transform, nl));
  // geometric normal: cross(e1, e2) of the stored edges — world space, or (instanced primitives) object space moved to world space like a normal
  f3 gcross = mk3(s0.x, s0.y, s0.z);
  if (instanced) gcross = transform_normal(md.transform, gcross);
  sf.ng = normalize3(gcross);
  sf.P = madd3(d, t, o);
  const hala_gpu_material& m = sv.materials[material];
  sf.mat.base = ld3(m.base_color);
  sf.mat.emission = ld3(m.emission);
  sf.mat.ax = m.ax; sf.mat.ay = m.ay; sf.mat.type = SIMPLE ? 0u : m.type;
  sf.mat.metallic = m.metallic; sf.mat.roughness = m.roughness; sf.mat.specular_tint = m.specular_tint;
  sf.mat.sheen = m.sheen; sf.mat.sheen_tint = m.sheen_tint; sf.mat.clearcoat = m.clearcoat;
  sf.mat.clearcoat_roughness = m.clearcoat_roughness; sf.mat.ior = m.ior;
  sf.mat.trans = 0.0f; sf.mat.eta = m.ior; sf.mat.opacity = SIMPLE ? 1.0f : m.opacity;
  // texture maps (set 2; u32::MAX = none, gltf_loader.rs:346-353)
  const uint32_t nt = sv.texture_count;
  const bool has_base = !SIMPLE && m.base_color_map_index < nt, has_nrm = !SIMPLE && m.normal_map_index < nt;
  const bool has_mr = !SIMPLE && m.metallic_roughness_map_index < nt, has_em = !SIMPLE && m.emission_map_index < nt;
  if (has_base || has_nrm || has_mr || has_em) {
    // line 2: texture coordinates and tangents (in flight together with the texture descriptors)
    const float4 s4 = sp[4], s5 = sp[5], s6 = sp[6], s7 = sp[7];
    struct VA { float tex_coord[2]; };
    const VA a = {{s4.x, s4.y}}, b = {{s4.z, s4.w}}, c = {{s5.x, s5.y}};
    const float tu = __fmaf_rn(c.tex_coord[0], v, __fmaf_rn(b.tex_coord[0], u, a.tex_coord[0] * w0));
    const float tv = __fmaf_rn(c.tex_coord[1], v, __fmaf_rn(b.tex_coord[1], u, a.tex_coord[1] * w0));
    // footprint of one pixel's cone on the surface vs. the uv density of this triangle
    const float du1 = b.tex_coord[0] - a.tex_coord[0], dv1 = b.tex_coord[1] - a.tex_coord[1];
    const float du2 = c.tex_coord[0] - a.tex_coord[0], dv2 = c.tex_coord[1] - a.tex_coord[1];
    const float uv_area = fabsf(du1 * dv2 - dv1 * du2);
    const float world_area = sqrtf(dot3(gcross, gcross));
    const float cosi = maxf(fabsf(dot3(d, sf.ng)), 0.1f);
    const float foot = t * pixel_spread / cosi;
    const float ratio = foot * foot * uv_area / world_area;
    const float lod_base = (ratio > 0.0f && ratio < 3.0e38f) ? 0.5f * log2_approx(ratio) : 0.0f;
    if (has_base) {
      const float4 s = tex_sample(sv, lut, m.base_color_map_index, tu, tv, tex_lod(sv, m.base_color_map_index, lod_base));
      sf.mat.base = sf.mat.base * mk3(s.x, s.y, s.z);
      sf.mat.opacity = sf.mat.opacity * s.w;
    }
    if (has_em) {
      const float4 s = tex_sample(sv, lut, m.emission_map_index, tu, tv, tex_lod(sv, m.emission_map_index, lod_base));
      sf.mat.emission = sf.mat.emission * mk3(s.x, s.y, s.z);
    }
    if (has_mr) {  // glTF: G = roughness, B = metallic
      const float4 s = tex_sample(sv, lut, m.metallic_roughness_map_index, tu, tv, tex_lod(sv, m.metallic_roughness_map_index, lod_base));
      sf.mat.metallic = sf.mat.metallic * s.z;
      if (m.type == 1u) {  // re-derive the packed alphas exactly like src/scene/gpu/material.rs:61-69
        const float rl = sqrtf(m.roughness) * s.y;
        const float r2 = rl * rl;
        const float aspect = sqrtf(1.0f - clampf(m.anisotropic, 0.0f, 1.0f) * 0.9f);
        sf.mat.roughness = r2;
        sf.mat.ax = maxf(0.001f, r2 / aspect);
        sf.mat.ay = maxf(0.001f, r2 * aspect);
      }
    }
    if (has_nrm) {
      const float4 s = tex_sample(sv, lut, m.normal_map_index, tu, tv, tex_lod(sv, m.normal_map_index, lod_base));
      // tangents: s5.z s5.w s6.x | s6.y s6.z s6.w | s7.x s7.y s7.z
      const f3 tl = madd3(mk3(s7.x, s7.y, s7.z), v, madd3(mk3(s6.y, s6.z, s6.w), u, mk3(s5.z, s5.w, s6.x) * w0));
      f3 tw = transform_vector(md.transform, tl);
      tw = tw - sf.ns * dot3(sf.ns, tw);  // Gram-Schmidt against the shading normal
      const float tl2 = dot3(tw, tw);
      if (tl2 > 0.0f) {
        tw = tw * (1.0f / sqrtf(tl2));
        const f3 bw = cross3(sf.ns, tw);
        const f3 nts = mk3(s.x * 2.0f - 1.0f, s.y * 2.0f - 1.0f, s.z * 2.0f - 1.0f);
        const f3 nn = to_world(nts, tw, bw, sf.ns);
        const float nn2 = dot3(nn, nn);
        if (nn2 > 0.0f) sf.ns = nn * (1.0f / sqrtf(nn2));
      }
    }
  }
  if (!SIMPLE && m.type == 1u) sf.mat.trans = m.specular_transmission * (1.0f - sf.mat.metallic);  // after the metallic map
  if (dot3(sf.ns, sf.ng) < 0.0f) sf.ng = -sf.ng;
  sf.absorb = splat3(1.0f); sf.glow = splat3(0.0f);
  sf.sigma = 0.0f; sf.hg = 0.0f; sf.scol = splat3(1.0f);
  if (dot3(sf.ng, d) > 0.0f) {  // the path arrives from behind the surface: it is leaving the object
    sf.ns = -sf.ns; sf.ng = -sf.ng;
    sf.mat.eta = 1.0f / sf.mat.ior;
    // §7.1e: the segment ran through the object's medium (closed, non-nested objects: no per-path medium state needed)
    const float dt = m.medium_density * t;
    if (SIMPLE) {
    } else if (m.medium_type == 1u)       // ABSORB: Beer-Lambert with sigma = density * (1 - colour)
      sf.absorb = mk3(exp_neg_poly(-(dt * (1.0f - m.medium_color[0]))), exp_neg_poly(-(dt * (1.0f - m.medium_color[1]))),
                      exp_neg_poly(-(dt * (1.0f - m.medium_color[2]))));
    else if (m.medium_type == 3u)  // EMISSIVE: colour * density per unit length
      sf.glow = mk3(m.medium_color[0] * dt, m.medium_color[1] * dt, m.medium_color[2] * dt);
    else if (m.medium_type == 2u) {  // SCATTER (§7.1f): decided by the caller, which owns the random numbers
      sf.sigma = m.medium_density; sf.hg = m.medium_anisotropy; sf.scol = mk3(m.medium_color[0], m.medium_color[1], m.medium_color[2]);
    }
  }
  return sf;
}

// §8 running mean
RT_DI float fold_mean(float mean_old, float x, uint32_t frame_index) {
  if (frame_index == 0u) return x;
  return (mean_old * (float)frame_index + x) / (float)(frame_index + 1u);
}

// src/rt_renderer.rs:1256-1311 on the device (the raygen stage writes the tonemapped final image, :688)
RT_DI f3 clamp01(f3 c) { return mk3(clampf(c.x, 0.0f, 1.0f), clampf(c.y, 0.0f, 1.0f), clampf(c.z, 0.0f, 1.0f)); }
RT_DI f3 mat3_mul(f3 c0, f3 c1, f3 c2, f3 v) {
  return mk3(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z, c0.z * v.x + c1.z * v.y + c2.z * v.z);
}
RT_DI f3 tonemap_select(f3 color, uint32_t enable_tonemap, uint32_t enable_aces, uint32_t use_simple_aces) {
  if (!enable_tonemap) return color;
  if (enable_aces) {
    if (use_simple_aces) {
      const float A = 2.51f, B = 0.03f, Y = 2.43f, D = 0.59f, E = 0.14f;
      f3 num = color * (color * A + splat3(B));
      f3 den = color * (color * Y + splat3(D)) + splat3(E);
      return clamp01(mk3(num.x / den.x, num.y / den.y, num.z / den.z));
    }
    f3 c = mat3_mul(mk3(0.59719f, 0.07600f, 0.02840f), mk3(0.35458f, 0.90834f, 0.13383f), mk3(0.04823f, 0.01566f, 0.83777f), color);
    f3 a = c * (c + splat3(0.0245786f)) - splat3(0.000090537f);
    f3 b = c * (c * 0.983729f + splat3(0.432951f)) + splat3(0.238081f);
    c = mk3(a.x / b.x, a.y / b.y, a.z / b.z);
    c = mat3_mul(mk3(1.60475f, -0.10208f, -0.00327f), mk3(-0.53108f, 1.10813f, -0.07276f), mk3(-0.07367f, -0.00605f, 1.07602f), c);
    return clamp01(c);
  }
  float dd = 1.0f + luminance(color) / 1.5f;
  f3 n = color * 1.0f;
  return mk3(n.x / dd, n.y / dd, n.z / dd);
}

}  // namespace rt
