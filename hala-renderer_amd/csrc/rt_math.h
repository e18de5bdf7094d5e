// rt_math.h — device-side scalar float32 building blocks of docs/RENDER_SPEC.md §2 for the HIP kernels.
//
// Determinism contract: compiled with -ffp-contract=off, every expression below is a fixed sequence of IEEE
// binary32 operations (+ - * / sqrt and explicit __fmaf_rn), so a kernel's result is a pure function of its
// inputs and independent of traversal/scheduling order.  tests/ hold these kernels to the CPU oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rt {

struct f3 {
  float x, y, z;
};

#define RT_DI __device__ __forceinline__

RT_DI f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_DI f3 splat3(float s) { return f3{s, s, s}; }
RT_DI f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
RT_DI f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DI f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DI f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DI f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_DI f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

// §2.1
RT_DI float dot3(f3 a, f3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
RT_DI f3 cross3(f3 a, f3 b) {
  return f3{__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))};
}
RT_DI f3 madd3(f3 d, float t, f3 o) { return f3{__fmaf_rn(d.x, t, o.x), __fmaf_rn(d.y, t, o.y), __fmaf_rn(d.z, t, o.z)}; }
RT_DI f3 normalize3(f3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return a * inv;
}
RT_DI float maxf(float a, float b) { return a > b ? a : b; }
RT_DI float minf(float a, float b) { return a < b ? a : b; }
// IEEE-754 minNum / maxNum with -0 < +0: ONE v_min_f32 / v_max_f32 (the select forms above cost a compare and a
// select each); used where the spec says so (RENDER_SPEC §4.3b: the slab tests of the compressed BVH4 nodes)
RT_DI float hw_minf(float a, float b) { return __builtin_fminf(a, b); }
RT_DI float hw_maxf(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DI float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
RT_DI float max3f(f3 a) { return maxf(a.x, maxf(a.y, a.z)); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kTwoPiSq = 19.7392088021787172376f;

// §2.2 polynomial trigonometry
RT_DI float sin_poly(float a) {
  float a2 = a * a;
  float p = -2.50521083854417187751e-8f;
  p = __fmaf_rn(p, a2, 2.75573192239858906526e-6f);
  p = __fmaf_rn(p, a2, -1.98412698412698412698e-4f);
  p = __fmaf_rn(p, a2, 8.33333333333333333333e-3f);
  p = __fmaf_rn(p, a2, -1.66666666666666666667e-1f);
  p = __fmaf_rn(p, a2, 1.0f);
  return a * p;
}
RT_DI float cos_poly(float a) {
  float a2 = a * a;
  float p = 2.08767569878680989792e-9f;
  p = __fmaf_rn(p, a2, -2.75573192239858906526e-7f);
  p = __fmaf_rn(p, a2, 2.48015873015873015873e-5f);
  p = __fmaf_rn(p, a2, -1.38888888888888888889e-3f);
  p = __fmaf_rn(p, a2, 4.16666666666666666667e-2f);
  p = __fmaf_rn(p, a2, -0.5f);
  p = __fmaf_rn(p, a2, 1.0f);
  return p;
}
RT_DI void sincos_2pi(float u, float* s, float* c) {
  float x = u * 4.0f;
  int q = (int)x;
  float f = x - (float)q;
  float a = f * kHalfPi;
  float sa = sin_poly(a), ca = cos_poly(a);
  q &= 3;
  *s = q == 0 ? sa : (q == 1 ? ca : (q == 2 ? -sa : -ca));
  *c = q == 0 ? ca : (q == 1 ? -sa : (q == 2 ? -ca : sa));
}
// e^x for x <= 0 (RENDER_SPEC §2.2): 2^(x log2 e) = 2^n 2^f, n = floor, f in [0,1); 2^f = e^(f ln 2) by its Taylor polynomial
// to the 8th power (relative error < 3e-8 before rounding); the scale by 2^n is exact.  Arguments >= 0 (and NaN) give 1.
RT_DI float exp_neg_poly(float x) {
  if (!(x < 0.0f)) return 1.0f;
  float y = x * 1.44269504088896340736f;
  if (y < -126.0f) return 0.0f;
  float n = floorf(y);
  float g = (y - n) * 0.69314718055994530942f;
  float p = 2.48015873015873015873e-5f;
  p = __fmaf_rn(p, g, 1.98412698412698412698e-4f);
  p = __fmaf_rn(p, g, 1.38888888888888888889e-3f);
  p = __fmaf_rn(p, g, 8.33333333333333333333e-3f);
  p = __fmaf_rn(p, g, 4.16666666666666666667e-2f);
  p = __fmaf_rn(p, g, 1.66666666666666666667e-1f);
  p = __fmaf_rn(p, g, 0.5f);
  p = __fmaf_rn(p, g, 1.0f);
  p = __fmaf_rn(p, g, 1.0f);
  return p * __uint_as_float((uint32_t)((int)n + 127) << 23);
}
// ln x for a positive normal float (RENDER_SPEC §7.1f): x = m 2^e with m in [sqrt(1/2), sqrt 2); z = (m-1)/(m+1);
// ln m = 2z (1 + z^2/3 + z^4/5 + z^6/7 + z^8/9) (|z| <= 0.172: truncation < 2e-9); ln x = fma(e, ln 2, ln m).
RT_DI float log_poly(float x) {
  const uint32_t b = __float_as_uint(x);
  int e = (int)(b >> 23) - 127;
  float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
  const float z = (m - 1.0f) / (m + 1.0f);
  const float z2 = z * z;
  float p = 0.11111111111111111f;
  p = __fmaf_rn(p, z2, 0.14285714285714285f);
  p = __fmaf_rn(p, z2, 0.2f);
  p = __fmaf_rn(p, z2, 0.33333333333333333f);
  p = __fmaf_rn(p, z2, 1.0f);
  return __fmaf_rn((float)e, 0.69314718055994530942f, (2.0f * z) * p);
}
RT_DI float acos_poly(float x) {
  float ax = fabsf(x);
  if (ax > 1.0f) ax = 1.0f;
  float p = -0.0012624911f;
  p = __fmaf_rn(p, ax, 0.0066700901f);
  p = __fmaf_rn(p, ax, -0.0170881256f);
  p = __fmaf_rn(p, ax, 0.0308918810f);
  p = __fmaf_rn(p, ax, -0.0501743046f);
  p = __fmaf_rn(p, ax, 0.0889789874f);
  p = __fmaf_rn(p, ax, -0.2145988016f);
  p = __fmaf_rn(p, ax, 1.5707963050f);
  float r = sqrtf(1.0f - ax) * p;
  return x < 0.0f ? kPi - r : r;
}
RT_DI float atan_poly01(float z) {
  float z2 = z * z;
  float p = 0.0028662257f;
  p = __fmaf_rn(p, z2, -0.0161657367f);
  p = __fmaf_rn(p, z2, 0.0429096138f);
  p = __fmaf_rn(p, z2, -0.0752896400f);
  p = __fmaf_rn(p, z2, 0.1065626393f);
  p = __fmaf_rn(p, z2, -0.1420889944f);
  p = __fmaf_rn(p, z2, 0.1999355085f);
  p = __fmaf_rn(p, z2, -0.3333314528f);
  p = __fmaf_rn(p, z2, 1.0f);
  return z * p;
}
RT_DI float atan2_poly(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = maxf(ax, ay), mn = minf(ax, ay);
  if (mx == 0.0f) return 0.0f;
  float a = atan_poly01(mn / mx);
  if (ay > ax) a = kHalfPi - a;
  if (x < 0.0f) a = kPi - a;
  return y < 0.0f ? -a : a;
}

// §2.3 RNG
RT_DI uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
RT_DI uint32_t rng_init(uint32_t pixel_id, uint32_t frame_index) {
  return pcg_hash(pixel_id + pcg_hash(frame_index * 0x9E3779B9u + 0x85EBCA6Bu));
}
RT_DI float rng_next(uint32_t& s) {
  uint32_t x = pcg_hash(s);
  s += 1u;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// §2.4 frames and sampling
RT_DI void onb(f3 n, f3* t, f3* b) {
  float sign = copysignf(1.0f, n.z);
  float a = -1.0f / (sign + n.z);
  float bb = n.x * n.y * a;
  *t = f3{1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x};
  *b = f3{bb, sign + n.y * n.y * a, -n.y};
}
RT_DI f3 to_world(f3 l, f3 t, f3 b, f3 n) {
  return f3{__fmaf_rn(n.x, l.z, __fmaf_rn(b.x, l.y, t.x * l.x)), __fmaf_rn(n.y, l.z, __fmaf_rn(b.y, l.y, t.y * l.x)),
            __fmaf_rn(n.z, l.z, __fmaf_rn(b.z, l.y, t.z * l.x))};
}
RT_DI f3 cosine_hemisphere(float u1, float u2) {
  float r = sqrtf(u1);
  float s, c;
  sincos_2pi(u2, &s, &c);
  return f3{r * c, r * s, sqrtf(maxf(0.0f, 1.0f - u1))};
}
// Henyey-Greenstein direction around the unit vector d (RENDER_SPEC §7.1f)
RT_DI f3 hg_sample(f3 d, float g, float u1, float u2) {
  g = minf(maxf(g, -0.99f), 0.99f);
  float ct;
  if (fabsf(g) < 1e-3f) ct = 1.0f - 2.0f * u1;
  else {
    const float q = (1.0f - g * g) / ((1.0f - g) + (2.0f * g) * u1);
    ct = ((1.0f + g * g) - q * q) / (2.0f * g);
  }
  ct = minf(maxf(ct, -1.0f), 1.0f);
  const float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
  float s, c;
  sincos_2pi(u2, &s, &c);
  f3 t, b;
  onb(d, &t, &b);
  return to_world(f3{st * c, st * s, ct}, t, b, d);
}
// Henyey-Greenstein phase function (= the pdf of hg_sample) for the cosine c between the propagation direction and the new one
RT_DI float hg_phase(float g, float c) {
  g = minf(maxf(g, -0.99f), 0.99f);
  if (fabsf(g) < 1e-3f) return 0.07957747154594767f;  // 1 / (4 pi)
  c = minf(maxf(c, -1.0f), 1.0f);
  const float g2 = g * g;
  const float x = (1.0f + g2) - (2.0f * g) * c;
  return (1.0f - g2) / (12.566370614359172f * (x * sqrtf(x)));
}
RT_DI float luminance(f3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }
RT_DI float power_heuristic(float a, float b) {
  float a2 = a * a;
  return a2 / (a2 + b * b);
}

// §4.3 ray preparation shared by both traversal kernels
RT_DI float safe_inv(float d) {
  float dd = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
  return 1.0f / dd;
}

}  // namespace rt
