// traverse.h — the BVH traversal device code of libhalart.so (docs/RENDER_SPEC.md §4): what
// vkCmdTraceRaysKHR + the RT cores do in the reference (src/rt_renderer.rs:458-464).
//
// Shape on CDNA4: one ray per lane, 64-lane waves inside persistent 256-thread workgroups; every wave pulls
// 64-ray batches from a global work counter until the queue is dry.  The top of the BVH (first `lds_nodes`
// nodes and first `lds_tris` triangles, all of them for small scenes) is staged in LDS once per workgroup;
// the per-lane traversal stack lives in LDS as [level][thread] (bank = thread, conflict-free), deeper levels
// spill to a global scratch area.  No MFMA: this is branchy scalar-per-ray work.
#pragma once
#include <hip/hip_runtime.h>

#include "hala_types.h"
#include "rt_math.h"

namespace rt {

constexpr int kTraverseThreads = 256;
// The traversal kernels are latency-bound (72 % of wave cycles wait on memory, profiles/r01_d_pmc_config4.txt); the
// second launch-bound argument is waves per SIMD and caps the register allocation accordingly.
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 6  // measured best (profiles/r01_e_variant_sweep.txt): 7-8 waves/SIMD force spills that cost more
#endif
#ifndef RT_TRI_PAIR
#define RT_TRI_PAIR 1
#endif
constexpr int kTraverseWavesPerSimd = RT_WAVES_PER_SIMD;
constexpr int kStackLds = 16;     // stack levels kept in LDS per lane (16 KB per workgroup -> 8 workgroups / CU)
constexpr int kStackSpill = 112;  // deeper levels, global scratch (max supported tree depth = 128)

struct TraverseLds {
  const float4* nodes;  // lds_nodes * 4
  const float4* tris;   // lds_tris * 3
  uint32_t* stack;      // kStackLds * kTraverseThreads
};

// cooperative staging of the BVH top into LDS (coalesced 16-B loads)
RT_DI TraverseLds stage_bvh(const SceneView& sv, unsigned char* smem) {
  float4* ln = reinterpret_cast<float4*>(smem);
  float4* lt = ln + (size_t)sv.lds_nodes * 4;
  uint32_t* st = reinterpret_cast<uint32_t*>(lt + (size_t)sv.lds_tris * 3);
  const float4* gn = reinterpret_cast<const float4*>(sv.nodes);
  const float4* gt = reinterpret_cast<const float4*>(sv.tris);
  for (uint32_t i = threadIdx.x; i < sv.lds_nodes * 4; i += blockDim.x) ln[i] = gn[i];
  for (uint32_t i = threadIdx.x; i < sv.lds_tris * 3; i += blockDim.x) lt[i] = gt[i];
  __syncthreads();
  return TraverseLds{ln, lt, st};
}

struct RayPre {
  f3 o, d, idir, ood;
  float tmin;
};
RT_DI RayPre make_ray(f3 o, f3 d, float tmin) {
  RayPre r;
  r.o = o; r.d = d; r.tmin = tmin;
  r.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  r.ood = r.o * r.idir;
  return r;
}
// §4.3 padded slab test
RT_DI bool box_test(const RayPre& r, f3 mn, f3 mx, float tlimit, float* tnear) {
  float x0 = __fmaf_rn(mn.x, r.idir.x, -r.ood.x), x1 = __fmaf_rn(mx.x, r.idir.x, -r.ood.x);
  float y0 = __fmaf_rn(mn.y, r.idir.y, -r.ood.y), y1 = __fmaf_rn(mx.y, r.idir.y, -r.ood.y);
  float z0 = __fmaf_rn(mn.z, r.idir.z, -r.ood.z), z1 = __fmaf_rn(mx.z, r.idir.z, -r.ood.z);
  float tn = maxf(maxf(minf(x0, x1), minf(y0, y1)), maxf(minf(z0, z1), r.tmin));
  float tf = minf(minf(maxf(x0, x1), maxf(y0, y1)), minf(maxf(z0, z1), tlimit));
  *tnear = tn;
  return tn <= tf * 1.0000004f;
}
// §4.2 Möller–Trumbore
RT_DI bool tri_test(const RayPre& r, float4 a, float4 b, float4 c, float* t, float* u, float* v) {
  f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
  f3 p = cross3(r.d, e2);
  float det = dot3(e1, p);
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  f3 tv = r.o - mk3(a.x, a.y, a.z);
  float uu = dot3(tv, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return false;
  f3 q = cross3(tv, e1);
  float vv = dot3(r.d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
  *t = dot3(e2, q) * inv; *u = uu; *v = vv;
  return true;
}

struct HitRec {
  float t, u, v;
  uint32_t prim;
};

// §4.4 traversal as a re-entrant state machine: one lane = one ray in flight; trav_step performs ONE node visit
// (both child box tests, leaf triangles, push/pop).  The persistent kernels interleave steps of all lanes and refill
// finished lanes with new rays, so a wave is not held hostage by its longest ray (wave64 divergence).
struct Trav {
  RayPre r;
  float tmax;
  HitRec best;
  uint32_t cur;
  int sp;
};
RT_DI void trav_begin(Trav& t, const RayPre& r, float tmax) {
  t.r = r; t.tmax = tmax;
  t.best.t = tmax; t.best.u = 0.0f; t.best.v = 0.0f; t.best.prim = kAbsent;
  t.cur = 0; t.sp = 0;
}
// returns true when the ray is finished (ANY: also on the first hit inside (tmin, tmax))
template <bool ANY, bool COUNT>
RT_DI bool trav_step(const SceneView& sv, const TraverseLds& lds, uint32_t* spill, Trav& t, uint32_t& n_nodes, uint32_t& n_tris) {
  const float4* gnodes = reinterpret_cast<const float4*>(sv.nodes);
  const float4* gtris = reinterpret_cast<const float4*>(sv.tris);
  uint32_t* stack = lds.stack + threadIdx.x;
  const RayPre& r = t.r;
  HitRec& best = t.best;
  const float tmax = t.tmax;
  const uint32_t cur = t.cur;
  int sp = t.sp;
  {
    float4 q0, q1, q2, q3;
    if (cur < sv.lds_nodes) {
      const float4* p = lds.nodes + (size_t)cur * 4;
      q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
    } else {
      const float4* p = gnodes + (size_t)cur * 4;
      q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
    }
    if (COUNT) n_nodes++;
    const uint32_t child0 = __float_as_uint(q3.x), child1 = __float_as_uint(q3.y);
    const uint32_t count0 = __float_as_uint(q3.z), count1 = __float_as_uint(q3.w);
    float tn0 = 0.0f, tn1 = 0.0f;
    const bool valid0 = !(count0 == 0u && child0 == kAbsent), valid1 = !(count1 == 0u && child1 == kAbsent);
    const bool h0 = valid0 && box_test(r, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), best.t, &tn0);
    const bool h1 = valid1 && box_test(r, mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), best.t, &tn1);
    const bool swp = h0 && h1 && tn1 < tn0;  // nearer entry first, ties -> child 0
    uint32_t next = kAbsent;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool second = (k == 1) != swp;  // which child this step looks at
      const bool h = second ? h1 : h0;
      if (!h) continue;
      const uint32_t child = second ? child1 : child0, count = second ? count1 : count0;
      const float tn = second ? tn1 : tn0;
      if (count > 0u) {
        if (!(tn <= best.t)) continue;
        if (COUNT) n_tris += count;
        // triangles are tested in storage order (spec), but fetched two at a time so that a leaf costs
        // ceil(count/2) memory round trips instead of count (the kernel is latency-bound: profiles/r01_d_pmc)
        for (uint32_t i = 0; i < count; i += (RT_TRI_PAIR ? 2u : 1u)) {
          const uint32_t ti = child + i;
          const bool two = RT_TRI_PAIR && (i + 1u < count);
          float4 a0, b0, c0, a1, b1, c1;
          if (ti < sv.lds_tris) {
            const float4* p = lds.tris + (size_t)ti * 3;
            a0 = p[0]; b0 = p[1]; c0 = p[2];
            if (two) { a1 = p[3]; b1 = p[4]; c1 = p[5]; }
          } else {
            const float4* p = gtris + (size_t)ti * 3;
            a0 = p[0]; b0 = p[1]; c0 = p[2];
            if (two) { a1 = p[3]; b1 = p[4]; c1 = p[5]; }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) break;
            const float4 a = j ? a1 : a0, b = j ? b1 : b0, c = j ? c1 : c0;
            float t, u, v;
            if (!tri_test(r, a, b, c, &t, &u, &v)) continue;
            const uint32_t id = __float_as_uint(a.w);
            if (ANY) {
              if (t > r.tmin && t < tmax) { best.t = t; best.u = u; best.v = v; best.prim = id; return true; }
            } else if (t > r.tmin && (t < best.t || (t == best.t && id < best.prim))) {
              best.t = t; best.u = u; best.v = v; best.prim = id;
            }
          }
        }
      } else if (next == kAbsent) {
        next = child;
      } else {
        if (sp < kStackLds) stack[sp * kTraverseThreads] = child; else spill[sp - kStackLds] = child;
        ++sp;
      }
    }
    if (next == kAbsent) {
      if (sp == 0) return true;
      --sp;
      next = sp < kStackLds ? stack[sp * kTraverseThreads] : spill[sp - kStackLds];
    }
    t.cur = next;
    t.sp = sp;
  }
  return false;
}

// ---- §4.4b: the same state machine over 64-B compressed BVH4 nodes ------------------------------------------------
RT_DI float ubyte_f32(uint32_t w, int c) { return (float)((w >> (8 * c)) & 0xffu); }  // v_cvt_f32_ubyte{c}
RT_DI void sort2(uint32_t& a, uint32_t& b) { const uint32_t lo = min(a, b), hi = max(a, b); a = lo; b = hi; }

// closest-hit / any-hit test of one leaf (count <= 8 triangles from `first`, storage order, fetched in pairs)
template <bool ANY>
RT_DI bool leaf_test(const SceneView& sv, const TraverseLds& lds, const RayPre& r, float tmax, HitRec& best, uint32_t first, uint32_t count) {
  const float4* gtris = reinterpret_cast<const float4*>(sv.tris);
  for (uint32_t i = 0; i < count; i += 2u) {
    const uint32_t ti = first + i;
    const bool two = i + 1u < count;
    float4 a0, b0, c0, a1, b1, c1;
    if (ti < sv.lds_tris) {
      const float4* p = lds.tris + (size_t)ti * 3;
      a0 = p[0]; b0 = p[1]; c0 = p[2];
      if (two) { a1 = p[3]; b1 = p[4]; c1 = p[5]; }
    } else {
      const float4* p = gtris + (size_t)ti * 3;
      a0 = p[0]; b0 = p[1]; c0 = p[2];
      if (two) { a1 = p[3]; b1 = p[4]; c1 = p[5]; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !two) break;
      const float4 a = j ? a1 : a0, b = j ? b1 : b0, c = j ? c1 : c0;
      float t, u, v;
      if (!tri_test(r, a, b, c, &t, &u, &v)) continue;
      const uint32_t id = __float_as_uint(a.w);
      if (ANY) {
        if (t > r.tmin && t < tmax) { best.t = t; best.u = u; best.v = v; best.prim = id; return true; }
      } else if (t > r.tmin && (t < best.t || (t == best.t && id < best.prim))) {
        best.t = t; best.u = u; best.v = v; best.prim = id;
      }
    }
  }
  return false;
}

template <bool ANY, bool COUNT>
RT_DI bool trav_step4(const SceneView& sv, const TraverseLds& lds, uint32_t* spill, Trav& t, uint32_t& n_nodes, uint32_t& n_tris) {
  const float4* gnodes = reinterpret_cast<const float4*>(sv.nodes);
  uint32_t* stack = lds.stack + threadIdx.x;
  const RayPre& r = t.r;
  HitRec& best = t.best;
  const uint32_t cur = t.cur;
  int sp = t.sp;
  float4 q0, q1, q2, q3;
  if (cur < sv.lds_nodes) {
    const float4* p = lds.nodes + (size_t)cur * 4;
    q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
  } else {
    const float4* p = gnodes + (size_t)cur * 4;
    q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
  }
  if (COUNT) n_nodes++;
  // plane distances without materialising the planes: t = q * (2^e * idir) + (pmin * idir - o * idir)
  const uint32_t ex = __float_as_uint(q0.w);
  const float kx = __uint_as_float((ex & 0xffu) << 23) * r.idir.x, ky = __uint_as_float(((ex >> 8) & 0xffu) << 23) * r.idir.y,
              kz = __uint_as_float(((ex >> 16) & 0xffu) << 23) * r.idir.z;
  const float ax = __fmaf_rn(q0.x, r.idir.x, -r.ood.x), ay = __fmaf_rn(q0.y, r.idir.y, -r.ood.y), az = __fmaf_rn(q0.z, r.idir.z, -r.ood.z);
  const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
  const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
  const uint32_t ref[4] = {__float_as_uint(q3.x), __float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w)};
  // ordering key: entry distance with the child slot in its two low mantissa bits (ties -> lower slot); misses sort last
  uint32_t key[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float x0 = __fmaf_rn(ubyte_f32(lox, c), kx, ax), x1 = __fmaf_rn(ubyte_f32(hix, c), kx, ax);
    const float y0 = __fmaf_rn(ubyte_f32(loy, c), ky, ay), y1 = __fmaf_rn(ubyte_f32(hiy, c), ky, ay);
    const float z0 = __fmaf_rn(ubyte_f32(loz, c), kz, az), z1 = __fmaf_rn(ubyte_f32(hiz, c), kz, az);
    const float tn = maxf(maxf(minf(x0, x1), minf(y0, y1)), maxf(minf(z0, z1), r.tmin));
    const float tf = minf(minf(maxf(x0, x1), maxf(y0, y1)), minf(maxf(z0, z1), best.t));
    const bool hit = ref[c] != kAbsent && tn <= tf * 1.0000004f;
    key[c] = hit ? ((__float_as_uint(maxf(tn, 0.0f)) & ~3u) | (uint32_t)c) : 0xffffffffu;
  }
  sort2(key[0], key[1]); sort2(key[2], key[3]); sort2(key[0], key[2]); sort2(key[1], key[3]); sort2(key[1], key[2]);
  // leaves first, nearest first (each one can shrink best.t for the ones after it) ...
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (key[k] == 0xffffffffu) break;
    const uint32_t slot = key[k] & 3u;
    const uint32_t rf = slot == 0u ? ref[0] : (slot == 1u ? ref[1] : (slot == 2u ? ref[2] : ref[3]));
    if (!(rf & kLeafRef)) continue;
    if (!(__uint_as_float(key[k] & ~3u) <= best.t)) continue;
    const uint32_t count = ((rf >> 28) & 7u) + 1u;
    if (COUNT) n_tris += count;
    if (leaf_test<ANY>(sv, lds, r, t.tmax, best, rf & 0x0fffffffu, count)) return true;
  }
  // ... then the inner children still in reach, farthest pushed first so that the nearest is visited next
  uint32_t next = kAbsent;
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    if (key[k] == 0xffffffffu) continue;
    const uint32_t slot = key[k] & 3u;
    const uint32_t rf = slot == 0u ? ref[0] : (slot == 1u ? ref[1] : (slot == 2u ? ref[2] : ref[3]));
    if (rf & kLeafRef) continue;
    if (!(__uint_as_float(key[k] & ~3u) <= best.t)) continue;
    if (next != kAbsent) {
      if (sp < kStackLds) stack[sp * kTraverseThreads] = next; else spill[sp - kStackLds] = next;
      ++sp;
    }
    next = rf;
  }
  if (next == kAbsent) {
    if (sp == 0) return true;
    --sp;
    next = sp < kStackLds ? stack[sp * kTraverseThreads] : spill[sp - kStackLds];
  }
  t.cur = next;
  t.sp = sp;
  return false;
}

// one node visit in whichever node format the scene was built with (uniform branch)
template <bool ANY, bool COUNT, bool WIDE>
RT_DI bool trav_visit(const SceneView& sv, const TraverseLds& lds, uint32_t* spill, Trav& t, uint32_t& n_nodes, uint32_t& n_tris) {
  if (WIDE) return trav_step4<ANY, COUNT>(sv, lds, spill, t, n_nodes, n_tris);
  return trav_step<ANY, COUNT>(sv, lds, spill, t, n_nodes, n_tris);
}

// whole-ray form (one lane runs its ray to completion)
template <bool ANY, bool COUNT>
RT_DI bool traverse(const SceneView& sv, const TraverseLds& lds, uint32_t* spill, const RayPre& r, float tmax, HitRec& best,
                    uint32_t& n_nodes, uint32_t& n_tris) {
  Trav t;
  trav_begin(t, r, tmax);
  while (!trav_step<ANY, COUNT>(sv, lds, spill, t, n_nodes, n_tris)) {}
  best = t.best;
  return best.prim != kAbsent;
}

}  // namespace rt
