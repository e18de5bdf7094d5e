// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed from the
// product (hala-renderer_amd/, libhalart.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// spec_math.h — scalar float32 building blocks of docs/RENDER_SPEC.md, written for the CPU.
// Every operation is a single IEEE-754 binary32 operation (+ - * / sqrt fma); the file must be compiled with
// -ffp-contract=off so that the compiler never fuses a*b+c on its own.  fmaf() is used ONLY where the spec says
// "fma".  That makes the result of every function here a pure function of its inputs, reproducible bit for bit
// by the HIP kernels (which are written independently against the same spec).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 { float x, y, z; };

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3s(float s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// RENDER_SPEC §2.1: dot = fma(az,bz, fma(ay,by, ax*bx))
static inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// RENDER_SPEC §2.1: cross component = fma(p, q, -(r*s))
static inline V3 cross3(V3 a, V3 b) {
  return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// v + d*t per component as one fma
static inline V3 madd3(V3 d, float t, V3 o) { return V3{fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)}; }
static inline float length3(V3 a) { return sqrtf(dot3(a, a)); }
// RENDER_SPEC §2.1: normalize = v * (1 / sqrt(dot(v,v)))
static inline V3 normalize3(V3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return a * inv;
}
static inline float maxf(float a, float b) { return a > b ? a : b; }  // no NaN inputs by construction
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
static inline float max3f(V3 a) { return maxf(a.x, maxf(a.y, a.z)); }

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kTwoPi = 6.28318530717958647692f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInvTwoPi = 0.15915494309189533577f;
static constexpr float kHalfPi = 1.57079632679489661923f;

// ---- RENDER_SPEC §2.2 polynomial trigonometry (so that CPU and GPU agree bit for bit) ----------------------
// sin on [0, pi/2]: odd Taylor polynomial to a^11, Horner with fma.
static inline float sin_poly(float a) {
  float a2 = a * a;
  float p = -2.50521083854417187751e-8f;        // -1/11!
  p = fmaf(p, a2, 2.75573192239858906526e-6f);  //  1/9!
  p = fmaf(p, a2, -1.98412698412698412698e-4f); // -1/7!
  p = fmaf(p, a2, 8.33333333333333333333e-3f);  //  1/5!
  p = fmaf(p, a2, -1.66666666666666666667e-1f); // -1/3!
  p = fmaf(p, a2, 1.0f);
  return a * p;
}
// cos on [0, pi/2]: even Taylor polynomial to a^12.
static inline float cos_poly(float a) {
  float a2 = a * a;
  float p = 2.08767569878680989792e-9f;         //  1/12!
  p = fmaf(p, a2, -2.75573192239858906526e-7f); // -1/10!
  p = fmaf(p, a2, 2.48015873015873015873e-5f);  //  1/8!
  p = fmaf(p, a2, -1.38888888888888888889e-3f); // -1/6!
  p = fmaf(p, a2, 4.16666666666666666667e-2f);  //  1/4!
  p = fmaf(p, a2, -0.5f);
  p = fmaf(p, a2, 1.0f);
  return p;
}
// sin/cos of 2*pi*u for u in [0,1): quadrant split is exact in binary32.
static inline void sincos_2pi(float u, float* s, float* c) {
  float x = u * 4.0f;
  int q = (int)x;  // floor, x >= 0
  float f = x - (float)q;
  float a = f * kHalfPi;
  float sa = sin_poly(a), ca = cos_poly(a);
  switch (q & 3) {
    case 0: *s = sa; *c = ca; break;
    case 1: *s = ca; *c = -sa; break;
    case 2: *s = -sa; *c = -ca; break;
    default: *s = -ca; *c = sa; break;
  }
}
// sin/cos of an arbitrary finite angle (radians): reduce to turns, then sincos_2pi.
static inline void sincos_rad(float a, float* s, float* c) {
  float t = a * kInvTwoPi;
  t = t - floorf(t);                // [0,1]
  if (t >= 1.0f) t = 0.0f;
  sincos_2pi(t, s, c);
}
// e^x for x <= 0: 2^(x log2 e) = 2^n 2^f, n = floor, f in [0,1); 2^f = e^(f ln 2) by its Taylor polynomial to the 8th power;
// the scale by 2^n is exact.  Arguments >= 0 (and NaN) give 1.
static inline float exp_neg_poly(float x) {
  if (!(x < 0.0f)) return 1.0f;
  float y = x * 1.44269504088896340736f;
  if (y < -126.0f) return 0.0f;
  float n = floorf(y);
  float g = (y - n) * 0.69314718055994530942f;
  float p = 2.48015873015873015873e-5f;
  p = fmaf(p, g, 1.98412698412698412698e-4f);
  p = fmaf(p, g, 1.38888888888888888889e-3f);
  p = fmaf(p, g, 8.33333333333333333333e-3f);
  p = fmaf(p, g, 4.16666666666666666667e-2f);
  p = fmaf(p, g, 1.66666666666666666667e-1f);
  p = fmaf(p, g, 0.5f);
  p = fmaf(p, g, 1.0f);
  p = fmaf(p, g, 1.0f);
  uint32_t bits = (uint32_t)((int)n + 127) << 23;
  float scale;
  memcpy(&scale, &bits, 4);
  return p * scale;
}
// ln x for a positive normal float (RENDER_SPEC 7.1f): x = m 2^e with m in [sqrt(1/2), sqrt 2); z = (m-1)/(m+1);
// ln m = 2z (1 + z^2/3 + z^4/5 + z^6/7 + z^8/9); ln x = fma(e, ln 2, ln m).
static inline float log_poly(float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  int e = (int)(b >> 23) - 127;
  uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;
  float m;
  memcpy(&m, &mb, 4);
  if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
  const float z = (m - 1.0f) / (m + 1.0f);
  const float z2 = z * z;
  float p = 0.11111111111111111f;
  p = fmaf(p, z2, 0.14285714285714285f);
  p = fmaf(p, z2, 0.2f);
  p = fmaf(p, z2, 0.33333333333333333f);
  p = fmaf(p, z2, 1.0f);
  return fmaf((float)e, 0.69314718055994530942f, (2.0f * z) * p);
}
// acos on [-1,1], Abramowitz & Stegun 4.4.46 (|err| <= 2e-8 before rounding)
static inline float acos_poly(float x) {
  float ax = fabsf(x);
  if (ax > 1.0f) ax = 1.0f;
  float p = -0.0012624911f;
  p = fmaf(p, ax, 0.0066700901f);
  p = fmaf(p, ax, -0.0170881256f);
  p = fmaf(p, ax, 0.0308918810f);
  p = fmaf(p, ax, -0.0501743046f);
  p = fmaf(p, ax, 0.0889789874f);
  p = fmaf(p, ax, -0.2145988016f);
  p = fmaf(p, ax, 1.5707963050f);
  float r = sqrtf(1.0f - ax) * p;
  return x < 0.0f ? kPi - r : r;
}
// atan on [0,1], A&S 4.4.49
static inline float atan_poly01(float z) {
  float z2 = z * z;
  float p = 0.0028662257f;
  p = fmaf(p, z2, -0.0161657367f);
  p = fmaf(p, z2, 0.0429096138f);
  p = fmaf(p, z2, -0.0752896400f);
  p = fmaf(p, z2, 0.1065626393f);
  p = fmaf(p, z2, -0.1420889944f);
  p = fmaf(p, z2, 0.1999355085f);
  p = fmaf(p, z2, -0.3333314528f);
  p = fmaf(p, z2, 1.0f);
  return z * p;
}
// atan2(y, x) in (-pi, pi]; atan2(0,0) = 0
static inline float atan2_poly(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = maxf(ax, ay), mn = minf(ax, ay);
  if (mx == 0.0f) return 0.0f;
  float a = atan_poly01(mn / mx);
  if (ay > ax) a = kHalfPi - a;
  if (x < 0.0f) a = kPi - a;
  return y < 0.0f ? -a : a;
}

// ---- RENDER_SPEC §2.3 RNG: PCG-RXS-M-XS hash of a running 32-bit counter ------------------------------------
static inline uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
static inline uint32_t rng_init(uint32_t pixel_id, uint32_t frame_index) {
  return pcg_hash(pixel_id + pcg_hash(frame_index * 0x9E3779B9u + 0x85EBCA6Bu));
}
static inline float rng_next(uint32_t* s) {
  uint32_t x = pcg_hash(*s);
  *s += 1u;
  return (float)(x >> 8) * (1.0f / 16777216.0f);  // exact: 24-bit integer times 2^-24
}

// ---- RENDER_SPEC §2.4 frames and sampling ----------------------------------------------------------------
// Branchless orthonormal basis (Duff et al. 2017), all ops as written.
static inline void onb(V3 n, V3* t, V3* b) {
  float sign = copysignf(1.0f, n.z);
  float a = -1.0f / (sign + n.z);
  float bb = n.x * n.y * a;
  *t = V3{1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x};
  *b = V3{bb, sign + n.y * n.y * a, -n.y};
}
static inline V3 to_world(V3 l, V3 t, V3 b, V3 n) {
  // t*l.x + b*l.y + n*l.z, evaluated as fma(n, l.z, fma(b, l.y, t*l.x))
  return V3{fmaf(n.x, l.z, fmaf(b.x, l.y, t.x * l.x)), fmaf(n.y, l.z, fmaf(b.y, l.y, t.y * l.x)),
            fmaf(n.z, l.z, fmaf(b.z, l.y, t.z * l.x))};
}
static inline V3 cosine_hemisphere(float u1, float u2) {
  float r = sqrtf(u1);
  float s, c;
  sincos_2pi(u2, &s, &c);
  return V3{r * c, r * s, sqrtf(maxf(0.0f, 1.0f - u1))};
}
// Henyey-Greenstein direction around the unit vector d (RENDER_SPEC 7.1f)
static inline V3 hg_sample(V3 d, float g, float u1, float u2) {
  g = minf(maxf(g, -0.99f), 0.99f);
  float ct;
  if (fabsf(g) < 1e-3f) ct = 1.0f - 2.0f * u1;
  else {
    const float q = (1.0f - g * g) / ((1.0f - g) + (2.0f * g) * u1);
    ct = ((1.0f + g * g) - q * q) / (2.0f * g);
  }
  ct = minf(maxf(ct, -1.0f), 1.0f);
  const float st = sqrtf(maxf(0.0f, 1.0f - ct * ct));
  float sn, cs;
  sincos_2pi(u2, &sn, &cs);
  V3 t, b;
  onb(d, &t, &b);
  return to_world(V3{st * cs, st * sn, ct}, t, b, d);
}
// Henyey-Greenstein phase function (= the pdf of hg_sample) for the cosine c between the propagation direction and the new one
static inline float hg_phase(float g, float c) {
  g = minf(maxf(g, -0.99f), 0.99f);
  if (fabsf(g) < 1e-3f) return 0.07957747154594767f;  // 1 / (4 pi)
  c = minf(maxf(c, -1.0f), 1.0f);
  const float g2 = g * g;
  const float x = (1.0f + g2) - (2.0f * g) * c;
  return (1.0f - g2) / (12.566370614359172f * (x * sqrtf(x)));
}
static inline float luminance(V3 c) {
  // src/envmap.rs:249-251 and src/rt_renderer.rs:1257-1259: (0.212671*r + 0.715160*g) + 0.072169*b, no fma
  return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z;
}
static inline float power_heuristic(float a, float b) {
  float a2 = a * a;
  return a2 / (a2 + b * b);
}

}  // namespace orc
