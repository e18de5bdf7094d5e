// traverse.h — the BVH traversal device code of libhalart.so (docs/RENDER_SPEC.md §4): what
// vkCmdTraceRaysKHR + the RT cores do in the reference (src/rt_renderer.rs:458-464).
//
// Shape on CDNA4: one ray per lane, 64-lane waves inside persistent 256-thread workgroups; every wave pulls rays from
// sharded global work counters until the queue is dry.  The BVH is a tree of 64-B compressed 4-wide nodes (§4.1b).
// Scenes whose whole BVH fits the LDS budget are STAGED: every workgroup copies nodes and triangles into LDS once and
// the kernel variant compiled for that case reads them with ds_read_b128 only; all other scenes read nodes and
// triangles straight from L2 / Infinity Cache / HBM (a partially staged top-of-tree slice measured no gain:
// profiles/r01_h_experiments.txt, and mixing both sources in one variant turns every fetch into a flat_load).
// The per-lane traversal stack lives in LDS as [entry][thread] (8-B entries, conflict-free), deeper entries spill to a
// global scratch area.  No MFMA: this is branchy scalar-per-ray work, and it is VALU-issue bound (profiles/r01_h_pmc_*):
// what counts is the number of vector instructions per node visit and per triangle test.
#pragma once
#include <hip/hip_runtime.h>

#include "hala_types.h"
#include "rt_math.h"

namespace rt {

constexpr int kTraverseThreads = 256;
// second launch-bound argument = waves per SIMD; it caps the register allocation (512 / waves)
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 5  // 28 KB of LDS per workgroup (stacks + leaf work lists): five workgroups per CU; 96 VGPRs
#endif
constexpr int kTraverseWavesPerSimd = RT_WAVES_PER_SIMD;
#ifndef RT_WAVES_PER_SIMD_STAGED
#define RT_WAVES_PER_SIMD_STAGED 4  // 128 VGPRs: room for the packed triangle-pair test; LDS-resident scenes have no latency to hide
#endif
constexpr int kTraverseWavesPerSimdStaged = RT_WAVES_PER_SIMD_STAGED;
// Per-lane traversal stack: 8-B entries (ordering key = entry distance | slot, child reference).  96 B of LDS per lane
// = 24 KB per workgroup: six workgroups plus their staged BVH fill the 160 KB of a CU at 6 waves/SIMD.
constexpr int kStackLdsStaged = 12;  // entries kept in LDS per lane, LDS-staged scenes (shallow trees: never spills)
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 8  // large scenes: the LDS also holds the wave's leaf work list (below); pops drop culled entries, deep stacks are rare
#endif
constexpr int kStackLdsGlobal = RT_STACK_LDS;
// two-level trees park the world-space ray on the stack while a lane is inside an instance (4 entries).  More LDS entries per lane do
// not pay: 10 or 12 instead of 8 cost a resident workgroup per CU (configs[3] two-level: 12.8 instead of 11.8 ms per frame)
#ifndef RT_STACK_LDS_INST
#define RT_STACK_LDS_INST 8
#endif
constexpr int kStackLdsInst = RT_STACK_LDS_INST;
template <bool STAGED, bool INST = false> RT_DI constexpr int stack_lds() { return STAGED ? kStackLdsStaged : (INST ? kStackLdsInst : kStackLdsGlobal); }
constexpr int kStackSpill = 56;  // deeper entries, global scratch
// Wave-cooperative leaf pass of the large-scene kernels (trav_step, !STAGED): every wave owns a work list of up to 64 x 4 leaf items
// (8 B: leaf reference, owner lane) and one 16-B merge slot per lane (best hit so far as a 64-bit key | u, v).
#ifndef RT_LEAF_SLOTS
#define RT_LEAF_SLOTS 4
#endif
constexpr int kLeafSlots = RT_LEAF_SLOTS;  // consumer lanes per leaf item = the largest leaf the large-scene builds may emit (power of two)
constexpr int kCoopItems = 64 * 4;
constexpr size_t kCoopBytesPerWave = (size_t)kCoopItems * 8 + 64 * 16;

// LDS is addressed through explicit address-space-3 pointers to builtin vectors: a generic pointer kept in a struct makes hipcc emit
// flat_load / flat_store (both memory pipes, both wait counters) instead of ds_read / ds_write.
#define RT_LDS __attribute__((address_space(3)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct TraverseLds {
  const RT_LDS f32x4* nodes;  // staged variant only: node_count * 4
  const RT_LDS f32x4* tris;   // staged variant only: tri_count * 3
  RT_LDS u32x2* stack;        // stack_lds() * kTraverseThreads
  RT_LDS u32x2* items;        // !STAGED: this wave's leaf work list
  RT_LDS u32x4* slots;        // !STAGED: this wave's merge slots
};
RT_DI float4 ld4(const RT_LDS f32x4* p) { const f32x4 v = *p; return make_float4(v.x, v.y, v.z, v.w); }

// LDS layout: stack | STAGED: nodes | triangles; else: per-wave work lists | per-wave merge slots.  STAGED: cooperative copy of the
// whole BVH (coalesced 16-B loads).
template <bool STAGED, bool INST = false>
RT_DI TraverseLds stage_bvh(const SceneView& sv, unsigned char* smem) {
  RT_LDS unsigned char* base = (RT_LDS unsigned char*)smem;
  RT_LDS u32x2* st = (RT_LDS u32x2*)base;
  RT_LDS unsigned char* rest = base + (size_t)stack_lds<STAGED, INST>() * kTraverseThreads * 8;
  if (STAGED) {
    RT_LDS f32x4* ln = (RT_LDS f32x4*)rest;
    RT_LDS f32x4* lt = ln + (size_t)sv.lds_nodes * 4;
    const f32x4* gn = reinterpret_cast<const f32x4*>(sv.nodes);
    const f32x4* gt = reinterpret_cast<const f32x4*>(sv.tris);
    for (uint32_t i = threadIdx.x; i < sv.lds_nodes * 4; i += blockDim.x) ln[i] = gn[i];
    for (uint32_t i = threadIdx.x; i < sv.lds_tris * 3; i += blockDim.x) lt[i] = gt[i];
    __syncthreads();
    return TraverseLds{ln, lt, st, nullptr, nullptr};
  }
  const uint32_t w = threadIdx.x >> 6;
  RT_LDS u32x2* items = (RT_LDS u32x2*)rest + (size_t)w * kCoopItems;
  RT_LDS u32x4* slots = (RT_LDS u32x4*)(rest + (size_t)(kTraverseThreads / 64) * kCoopItems * 8) + (size_t)w * 64;
  return TraverseLds{nullptr, nullptr, st, items, slots};
}

struct RayPre {
  f3 o, d, idir, ood;
  float tmin;
};
RT_DI RayPre make_ray(f3 o, f3 d, float tmin) {
  RayPre r;
  r.o = o; r.d = d; r.tmin = maxf(tmin, 0.0f);  // rays start at or after their origin (RENDER_SPEC §4.2): entry distances are >= +0
  r.idir = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  r.ood = r.o * r.idir;
  return r;
}
// §4.2 Möller–Trumbore, straight-line: the same operations in the same order as the early-out form of the spec (an accepted hit
// has the same t, u, v bit for bit; a rejected one is rejected by the same comparisons, evaluated at the end).  No branch sits
// between the three 16-B loads of the triangle and their uses, so the loads leave together — with early-outs hipcc sank the load of
// v0 below the `det == 0` branch: two dependent memory round trips per triangle.
RT_DI bool tri_test_od(f3 o, f3 d, float4 a, float4 b, float4 c, float* t, float* u, float* v, float* det_out) {
  const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
  const f3 p = cross3(d, e2);
  const float det = dot3(e1, p);
  const float inv = 1.0f / det;
  const f3 tv = o - mk3(a.x, a.y, a.z);
  const float uu = dot3(tv, p) * inv;
  const f3 q = cross3(tv, e1);
  const float vv = dot3(d, q) * inv;
  *t = dot3(e2, q) * inv; *u = uu; *v = vv; *det_out = det;
  return det != 0.0f && uu >= 0.0f && uu <= 1.0f && vv >= 0.0f && uu + vv <= 1.0f;
}

struct HitRec {
  float t, u, v;
  uint32_t prim;
};

// §4.4 traversal as a re-entrant state machine: one lane = one ray in flight; trav_step performs ONE node visit
// (four child box tests, the leaves among them, push/pop).  The persistent kernels interleave steps of all lanes and
// refill finished lanes with new rays, so a wave is not held hostage by its longest ray (wave64 divergence).
struct Trav {
  RayPre r;
  float tmax;
  HitRec best;
  uint32_t cur;
  int sp;
  uint32_t key;  // any-hit rays: decides which translucent triangles block this ray (RENDER_SPEC 7.1d)
  uint32_t tau[3];  // any-hit rays: optical depth of the media crossed so far, 2^-16 units, wrap-around sums (RENDER_SPEC 7.1g)
  // two-level trees (INST variants): while the ray is inside an instance `r` is the OBJECT-space ray, the triangles' ids are local to the
  // primitive: + gid_base = global id, + (shade_base & 0x7fffffff) = shading record (bit 31 of shade_base: inside an instance)
  uint32_t gid_base, shade_base;
};
RT_DI void trav_begin(Trav& t, const RayPre& r, float tmax, uint32_t key) {
  t.r = r; t.tmax = tmax; t.key = key; t.tau[0] = t.tau[1] = t.tau[2] = 0u;
  t.best.t = tmax == tmax ? tmax : -1.0f;  // a NaN limit admits no hit (the leaf-count compare of trav_step works on the bits of best.t)
  t.best.u = 0.0f; t.best.v = 0.0f; t.best.prim = kAbsent;
  t.cur = 0; t.sp = 0; t.gid_base = 0u; t.shade_base = 0u;
}

typedef float v2f __attribute__((ext_vector_type(2)));  // operand pair of the packed FP32 instructions
RT_DI float ubyte_f32(uint32_t w, int c) { return (float)((w >> (8 * c)) & 0xffu); }  // v_cvt_f32_ubyte{c}
// (key, reference) compare-exchange of the 5-comparator sorting network
RT_DI void sort2kv(uint32_t& ka, uint32_t& kb, uint32_t& ra, uint32_t& rb) {
  const bool s = kb < ka;
  const uint32_t k0 = s ? kb : ka, k1 = s ? ka : kb, r0 = s ? rb : ra, r1 = s ? ra : rb;
  ka = k0; kb = k1; ra = r0; rb = r1;
}
constexpr uint32_t kMissKey = 0xffffffffu;
constexpr uint32_t kInnerKey = 0x80000000u;  // key bit of an inner child: inner children sort after all leaf children
RT_DI float key_tn(uint32_t key) { return __uint_as_float(key & 0x7ffffffcu); }

// §4.2 for TWO triangles at once: the same operation sequence as tri_test, component 0 = first triangle, component 1 =
// second, on the packed FP32 pipe (v_pk_mul / v_pk_fma / v_pk_add: two IEEE binary32 results per instruction, each
// rounded exactly like the scalar instruction).  A triangle pair costs about what one scalar test does; the kernels
// are VALU-issue bound and leaves hold up to two triangles.
RT_DI v2f pk2(float a, float b) { return v2f{a, b}; }
RT_DI v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
RT_DI v2f pk_dot3(v2f ax, v2f ay, v2f az, v2f bx, v2f by, v2f bz) { return pk_fma(az, bz, pk_fma(ay, by, ax * bx)); }
RT_DI void tri_test2(const RayPre& r, float4 a0, float4 b0, float4 c0, float4 a1, float4 b1, float4 c1, bool ok[2], float t[2], float u[2],
                     float v[2], float dt[2]) {
  const v2f dx = pk2(r.d.x, r.d.x), dy = pk2(r.d.y, r.d.y), dz = pk2(r.d.z, r.d.z);
  const v2f e1x = pk2(b0.x, b1.x), e1y = pk2(b0.y, b1.y), e1z = pk2(b0.z, b1.z);
  const v2f e2x = pk2(c0.x, c1.x), e2y = pk2(c0.y, c1.y), e2z = pk2(c0.z, c1.z);
  // p = cross3(d, e2)
  const v2f px = pk_fma(dy, e2z, -(dz * e2y)), py = pk_fma(dz, e2x, -(dx * e2z)), pz = pk_fma(dx, e2y, -(dy * e2x));
  const v2f det = pk_dot3(e1x, e1y, e1z, px, py, pz);
  const v2f inv = pk2(1.0f / det.x, 1.0f / det.y);
  const v2f tvx = pk2(r.o.x, r.o.x) - pk2(a0.x, a1.x), tvy = pk2(r.o.y, r.o.y) - pk2(a0.y, a1.y), tvz = pk2(r.o.z, r.o.z) - pk2(a0.z, a1.z);
  const v2f uu = pk_dot3(tvx, tvy, tvz, px, py, pz) * inv;
  // q = cross3(tv, e1)
  const v2f qx = pk_fma(tvy, e1z, -(tvz * e1y)), qy = pk_fma(tvz, e1x, -(tvx * e1z)), qz = pk_fma(tvx, e1y, -(tvy * e1x));
  const v2f vv = pk_dot3(dx, dy, dz, qx, qy, qz) * inv;
  const v2f tt = pk_dot3(e2x, e2y, e2z, qx, qy, qz) * inv;
  const v2f uv = uu + vv;
  ok[0] = det.x != 0.0f && uu.x >= 0.0f && uu.x <= 1.0f && vv.x >= 0.0f && uv.x <= 1.0f;
  ok[1] = det.y != 0.0f && uu.y >= 0.0f && uu.y <= 1.0f && vv.y >= 0.0f && uv.y <= 1.0f;
  t[0] = tt.x; t[1] = tt.y; u[0] = uu.x; u[1] = uu.y; v[0] = vv.x; v[1] = vv.y; dt[0] = det.x; dt[1] = det.y;
}

// LDS-staged scenes: closest-hit / any-hit test of one leaf (count <= 8 triangles from `first`, storage order; fetched and tested
// two at a time on the packed FP32 pipe: a leaf costs ceil(count/2) LDS round trips and ceil(count/2) packed tests; 4 waves/SIMD,
// 128 VGPRs).  ANY: true on the first accepted triangle.
template <bool ANY, bool ALPHA>
RT_DI bool leaf_test_staged(const SceneView& sv, const TraverseLds& lds, const RayPre& r, float tmax, uint32_t key, uint32_t tau[3], HitRec& best, uint32_t first,
                            uint32_t count) {
  for (uint32_t i = 0; i < count; i += 2u) {
    const bool two = i + 1u < count;
    const RT_LDS f32x4* p = lds.tris + (size_t)(first + i) * 3;
    const float4 a0 = ld4(p), b0 = ld4(p + 1), c0 = ld4(p + 2);
    float4 a1 = a0, b1 = b0, c1 = c0;  // a single triangle is tested against itself in the second component (result unused)
    if (two) { a1 = ld4(p + 3); b1 = ld4(p + 4); c1 = ld4(p + 5); }
    bool ok[2]; float tt[2], uu[2], vv[2], dd[2];
    tri_test2(r, a0, b0, c0, a1, b1, c1, ok, tt, uu, vv, dd);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!ok[j] || (j == 1 && !two)) continue;
      const uint32_t id = __float_as_uint(j ? a1.w : a0.w);
      if (ANY) {
        if (tt[j] > r.tmin && tt[j] < tmax) {
          if (ALPHA && __float_as_uint(j ? b1.w : b0.w) != 0u) {  // translucent and / or the boundary of a medium (7.1d, 7.1g)
            uint32_t q[3];
            if (!any_hit_event(sv, key, id, id, __float_as_uint(j ? b1.w : b0.w), tt[j], dd[j], uu[j], vv[j], q)) { tau[0] += q[0]; tau[1] += q[1]; tau[2] += q[2]; continue; }
          }
          best.t = tt[j]; best.u = uu[j]; best.v = vv[j]; best.prim = id; return true;
        }
      } else {
        // closest hit: the record carries id << 3 | shading kind (hala_types.h: hit_encode; ordered like the ids)
        const uint32_t enc = hit_encode(id, __float_as_uint(j ? c1.w : c0.w));
        if (tt[j] > r.tmin && (tt[j] < best.t || (tt[j] == best.t && enc < best.prim))) { best.t = tt[j]; best.u = uu[j]; best.v = vv[j]; best.prim = enc; }
      }
    }
  }
  return false;
}

// per-lane tallies of counting launches (summed over the wave at kernel end)
struct StepCounters {
  uint32_t nodes = 0, tris = 0, leaf_lanes = 0, leaf_passes = 0, wave_steps = 0;
};

RT_DI uint32_t mbcnt64(unsigned long long m) {  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
RT_DI float lane_read(uint32_t src_lane_x4, float v) {  // v of lane src_lane_x4 / 4 (ds_bpermute_b32: every lane of the wave must take part)
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)src_lane_x4, __float_as_int(v)));
}

// One node visit of every lane that holds a ray (`has`); returns true when the lane's ray is finished (ANY: also on the first hit
// inside (tmin, tmax)).  The whole wave must call it together: the large-scene variant (!STAGED) tests the leaves the wave's lanes
// reached in this step COOPERATIVELY — the (ray, triangle) pairs are dealt out over all 64 lanes, whoever owns the ray:
//   lane with n leaves in reach -> n items {leaf reference, owner lane} in the wave's LDS work list (position by a ballot prefix sum),
//                                   its best hit so far as a 64-bit key (t bits | triangle id) in its merge slot;
//   consumer lane s of a pass     -> item s / 4, triangle s % 4 of that leaf: fetches the OWNER's ray with ds_bpermute, the triangle
//                                   from memory, runs the one scalar triangle test, and merges an accepted hit into the owner's slot
//                                   with an LDS 64-bit unsigned min (t > 0, so the keys order like (t, id): exactly the closest-hit
//                                   rule of 4.2, whatever the order of the merges); the winner then publishes its (u, v);
//   owner                         -> reads its slot back.
// A per-lane leaf loop ran at 4-6 of 64 lanes on the 1 M-triangle scene (1.4-1.6 passes of <= 2 sequential triangle tests per wave
// step, each with its own dependent fetch); dealt out, a wave step has ONE pass of one triangle test at ~4x the lanes.
// INST: the tree has instance levels (RENDER_SPEC 4.5).  An instance leaf sorts and waits like an inner child; when its turn comes the
// lane parks the world-space ray on its traversal stack (three entries under an exit mark), moves the ray into the instance's object
// space (t is kept: the direction is not normalised) and goes on at the root of the primitive's tree; popping the exit mark brings the
// world-space ray back.  No registers and no LDS beyond the stack: the large-scene kernels have neither to spare at 5 waves per SIMD.
template <bool ANY, bool COUNT, bool STAGED, bool ALPHA, bool INST = false>
RT_DI bool trav_step(const SceneView& sv, const TraverseLds& lds, uint2* spill, Trav& t, bool has, StepCounters& sc) {
  static_assert(!(INST && STAGED), "LDS-staged trees have no instance levels");
  constexpr int kS = stack_lds<STAGED, INST>();
  constexpr bool kSlotOrder = ANY;  // RENDER_SPEC 4.4c
#ifndef RT_INST_BATCH
#define RT_INST_BATCH 8  // configs[3] as a two-level tree: 11.05 -> 10.88 ms per frame (4 / 8 / 12 / 16 / 24: 10.94 / 10.88 / 10.88 / 10.90 / 11.04); any-hit rays lose by waiting (their paths are short): they move at once
#endif
  constexpr int kInstBatch = (INST && !ANY) ? RT_INST_BATCH : 0;  // two-level trees: lanes per batch of instance transitions (0: each lane on its own, at once)
  RT_LDS u32x2* stack = lds.stack + threadIdx.x;
  const RayPre& r = t.r;
  HitRec& best = t.best;
  int sp = t.sp;
  uint32_t ref[4] = {kAbsent, kAbsent, kAbsent, kAbsent}, key[4] = {kMissKey, kMissKey, kMissKey, kMissKey};
  uint32_t next = kAbsent, next_key = 0;
  bool lf[4] = {false, false, false, false};  // lf[k]: the k-th sorted child is a leaf in reach (a prefix: lf[k] implies lf[k - 1])
  auto pop = [&]() -> uint2 {
    --sp;
    if (sp < kS) { const u32x2 v = stack[sp * kTraverseThreads]; return make_uint2(v.x, v.y); }
    return spill[sp - kS];
  };
  auto push = [&](uint32_t a, uint32_t b) {
    if (sp < kS) stack[sp * kTraverseThreads] = u32x2{a, b}; else spill[sp - kS] = make_uint2(a, b);
    ++sp;
  };
  // moving a ray into an instance's object space (RENDER_SPEC 4.5): the world-space ray is parked under an exit mark; returns the root of the primitive's tree
  auto enter_instance = [&](uint32_t leaf) -> uint32_t {
    const float4* ip = reinterpret_cast<const float4*>(sv.inst_refs + (leaf & 0x0fffffffu));
    const float4 i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];  // r0 | r1 | r2 | tr, then root, gid_base, shade_base, inst
    push(__float_as_uint(t.r.o.x), __float_as_uint(t.r.o.y));
    push(__float_as_uint(t.r.o.z), __float_as_uint(t.r.d.x));
    push(__float_as_uint(t.r.d.y), __float_as_uint(t.r.d.z));
    push(0u, kExitRef);  // key 0: never culled (parking 1 / d as well, to spare the exit its three divisions, measured no gain: 11.86 vs 11.83 ms)
    const f3 r0 = mk3(i0.x, i0.y, i0.z), r1 = mk3(i0.w, i1.x, i1.y), r2 = mk3(i1.z, i1.w, i2.x), tr = mk3(i2.y, i2.z, i2.w);
    const f3 tv = t.r.o - tr, d = t.r.d;
    t.r = make_ray(mk3(dot3(r0, tv), dot3(r1, tv), dot3(r2, tv)), mk3(dot3(r0, d), dot3(r1, d), dot3(r2, d)), t.r.tmin);
    t.gid_base = __float_as_uint(i3.y); t.shade_base = __float_as_uint(i3.z) | 0x80000000u;  // bit 31: inside an instance
    return __float_as_uint(i3.x);
  };
  auto leave_instance = [&]() {  // the exit mark has been popped: the world-space ray lies under it
    const uint2 c2 = pop(), c1 = pop(), c0 = pop();
    t.r = make_ray(mk3(__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c1.x)),
                   mk3(__uint_as_float(c1.y), __uint_as_float(c2.x), __uint_as_float(c2.y)), t.r.tmin);
    t.gid_base = 0u; t.shade_base = 0u;
  };
  bool work = has;         // the lane visits a node in this step
  bool done_early = false;  // INST, batched transitions: the ray ended while leaving an instance
  if (INST && kInstBatch > 0) {
    // Instance entries and exits are ~120 and ~45 vector instructions that a wave runs whenever ANY of its lanes needs them — 2-3 lanes
    // in 92 % of the steps on configs[3].  With kInstBatch a lane that reaches an instance leaf, or pops an exit mark, WAITS there
    // (t.cur = the leaf reference / kExitRef) until that many lanes of the wave wait, or none has a node to visit: the transitions then
    // run once for all of them.  Which nodes a ray visits, and in which order, does not change.
    const bool w_out = has && t.cur == kExitRef, w_in = has && is_inst_leaf(t.cur);
    const unsigned long long mo = __builtin_amdgcn_ballot_w64(w_out), mi = __builtin_amdgcn_ballot_w64(w_in);
    const unsigned long long mw = __builtin_amdgcn_ballot_w64(has && (int32_t)t.cur >= 0);
    if ((mo | mi) != 0ull && ((uint32_t)__popcll(mo) + (uint32_t)__popcll(mi) >= (uint32_t)kInstBatch || mw == 0ull)) {
      uint32_t c = t.cur;
      if (w_out) {
        leave_instance();
        c = kAbsent;
        while (c == kAbsent) {  // the first entry still in reach (world level: no exit mark can follow)
          if (sp == 0) { done_early = true; break; }
          const uint2 e = pop();
          if (kSlotOrder || key_tn(e.x) <= best.t) c = e.y;
        }
      }
      if (has && !done_early && is_inst_leaf(c)) c = enter_instance(c);
      if (has && !done_early) t.cur = c;
    }
    work = has && !done_early && (int32_t)t.cur >= 0;
  }
  if (work) {
    float4 q0, q1, q2, q3;
    if (STAGED) { const RT_LDS f32x4* p = lds.nodes + (size_t)t.cur * 4; q0 = ld4(p); q1 = ld4(p + 1); q2 = ld4(p + 2); q3 = ld4(p + 3); }
    else { const float4* p = reinterpret_cast<const float4*>(sv.nodes) + (size_t)t.cur * 4; q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3]; }
    if (COUNT) sc.nodes++;
    // plane distances without materialising the planes: t = q * (2^e * idir) + (pmin * idir - o * idir)
    const uint32_t ex = __float_as_uint(q0.w);
    const float kx = __uint_as_float((ex & 0xffu) << 23) * r.idir.x, ky = __uint_as_float(((ex >> 8) & 0xffu) << 23) * r.idir.y,
                kz = __uint_as_float(((ex >> 16) & 0xffu) << 23) * r.idir.z;
    const float ax = __fmaf_rn(q0.x, r.idir.x, -r.ood.x), ay = __fmaf_rn(q0.y, r.idir.y, -r.ood.y), az = __fmaf_rn(q0.z, r.idir.z, -r.ood.z);
    const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
    const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    ref[0] = __float_as_uint(q3.x); ref[1] = __float_as_uint(q3.y); ref[2] = __float_as_uint(q3.z); ref[3] = __float_as_uint(q3.w);
    // Slab test (§4.3b): tn = max(min(x0,x1), min(y0,y1), min(z0,z1), tmin), tf = min(max(x0,x1), ..., best.t) with
    // x0/x1 = fma(qlo/qhi, k, a).  qlo <= qhi and fma rounding is monotonic, so min(x0,x1) is simply the plane picked by the
    // sign of the direction: select the near / far byte words once per node (6 v_cndmask) instead of 6 min/max per child,
    // and compute (near, far) of one axis with one packed v_pk_fma_f32.  Bit-identical to the min/max form.
    const bool px = r.idir.x >= 0.0f, py = r.idir.y >= 0.0f, pz = r.idir.z >= 0.0f;
    const uint32_t nx = px ? lox : hix, fx = px ? hix : lox, ny = py ? loy : hiy, fy = py ? hiy : loy, nz = pz ? loz : hiz, fz = pz ? hiz : loz;
    const v2f kx2 = {kx, kx}, ky2 = {ky, ky}, kz2 = {kz, kz}, ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
    // ordering key: entry distance (>= tmin >= +0, so its sign bit is free) with the child slot in its two low mantissa
    // bits (ties -> lower slot) and the sign bit set for inner children: one sort puts the leaves first, nearest first,
    // then the inner children, nearest first, then the misses.  hw_minf / hw_maxf are the one-instruction IEEE minNum / maxNum
    // (v_min3 / v_max3 fuse them).
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const v2f tx = __builtin_elementwise_fma(v2f{ubyte_f32(nx, c), ubyte_f32(fx, c)}, kx2, ax2);
      const v2f ty = __builtin_elementwise_fma(v2f{ubyte_f32(ny, c), ubyte_f32(fy, c)}, ky2, ay2);
      const v2f tz = __builtin_elementwise_fma(v2f{ubyte_f32(nz, c), ubyte_f32(fz, c)}, kz2, az2);
      const float tn = hw_maxf(hw_maxf(tx.x, ty.x), hw_maxf(tz.x, r.tmin));
      const float tf = hw_minf(hw_minf(tx.y, ty.y), hw_minf(tz.y, best.t));
      const bool hit = ref[c] != kAbsent && tn <= tf * 1.0000004f;
      if (kSlotOrder) { ref[c] = hit ? ref[c] : kAbsent; continue; }  // no keys (4.4c): a child that was missed is an absent child from here on
      const uint32_t inner_bit = INST ? ((!(ref[c] >> 31) || (ref[c] >> 28) == 0xFu) ? kInnerKey : 0u) : (~ref[c] & kInnerKey);  // instance leaves wait like inner children
      uint32_t kc;  // (tn & ~3) | (slot | inner bit): one v_and_or_b32 (hipcc emits v_and + v_or for the C form)
      asm("v_and_or_b32 %0, %1, -4, %2" : "=v"(kc) : "v"(__float_as_uint(tn)), "v"((uint32_t)c | inner_bit));
      key[c] = hit ? kc : kMissKey;
    }
    // Any-hit rays (kSlotOrder, RENDER_SPEC 4.4c): nothing they find moves their limit, so no order of the children saves
    // them a visit that another order would not have cost, and a child that passed the slab test never needs a second look: they take
    // the children in slot order, without sort and without keys (45 of the ~290 vector instructions of a node step).  All other rays:
    // leaves first, nearest first, then the inner children, nearest first, then the misses.
    if (!kSlotOrder) {
      sort2kv(key[0], key[1], ref[0], ref[1]); sort2kv(key[2], key[3], ref[2], ref[3]); sort2kv(key[0], key[2], ref[0], ref[2]);
      sort2kv(key[1], key[3], ref[1], ref[3]); sort2kv(key[1], key[2], ref[1], ref[2]);
    }
    // inner children: the first of them (sorted: the nearest) is visited next, the others go on the stack last first, each with its key
    // so that a pop can drop entries that a hit found in the meantime has put out of reach.  "An inner child that was hit" is ONE signed
    // compare (0x80000000 <= key < 0xffffffff; slot order: a reference without the leaf bit, or an instance leaf); such a child is pushed
    // exactly when an inner child stands to its left (the leftmost one is `next`): three predicated stores, no loop-carried `next`
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      in[k] = !kSlotOrder ? (int32_t)key[k] < -1 : (INST ? (int32_t)ref[k] >= (int32_t)kInstLeafTag && ref[k] < kExitRef : (int32_t)ref[k] >= 0);
    const bool in0 = in[0], in1 = in[1], in2 = in[2], in3 = in[3];
#pragma unroll
    for (int k = 3; k >= 1; --k) {
      const bool left = k == 3 ? (in0 | in1 | in2) : (k == 2 ? (in0 | in1) : in0), self = k == 3 ? in3 : (k == 2 ? in2 : in1);
      if (self & left) {
        const uint32_t ek = kSlotOrder ? 0u : key[k];
        if (sp < kS) stack[sp * kTraverseThreads] = u32x2{ek, ref[k]}; else spill[sp - kS] = make_uint2(ek, ref[k]);
        ++sp;
      }
    }
    next = in0 ? ref[0] : (in1 ? ref[1] : (in2 ? ref[2] : (in3 ? ref[3] : kAbsent)));
    if (!kSlotOrder) next_key = in0 ? key[0] : (in1 ? key[1] : (in2 ? key[2] : key[3]));
  }
  // the leaves in reach as the node is entered (§4.4b): a sorted prefix.  Large trees: ALL of them are tested, none is culled by a
  // sibling's hit; small (LDS-staged) trees: the lane tests them one after the other, nearest first, while they stay in reach.
  // key_tn(key) <= best.t for a leaf key (sign bit clear, best.t > 0) is the unsigned compare key <= bits(best.t) | 3; inner children
  // and misses have the sign bit and fail it: four compares whose results stay lane masks — the ballots, the item positions and the
  // predicates of the leaf pass below are scalar work on them, the number of leaves is never formed as a vector value.  (Outside the
  // `has` block on purpose: a bool that crosses that join would be kept as a 0 / 1 byte in a register; a lane without a ray holds four
  // miss keys, 0xffffffff > 0 — slot order: four absent references.)
  const uint32_t reach = work ? (__float_as_uint(best.t) | 3u) : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k)  // a bare compare: its ballot is the compare's own lane mask
    lf[k] = !kSlotOrder ? key[k] <= reach : (INST ? (int32_t)ref[k] < (int32_t)kInstLeafTag : (int32_t)ref[k] < -1);
  if (COUNT && !STAGED) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (lf[k]) sc.tris += ((ref[k] >> 28) & 7u) + 1u;
  }
  bool found = false;  // ANY: an occluder was hit
  if (STAGED) {
    if (!has) return false;
    // LDS-resident scenes: the lane tests its own leaves (4.6 lanes of 64 would be the large-scene figure; here 28-36 are busy and a
    // triangle pair costs one packed test), nearest first
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (kSlotOrder) { if (!lf[k]) continue; }  // any-hit: the leaves that were hit, in slot order, until a triangle is accepted
      else {
        if (!lf[k]) break;
        if (!(key_tn(key[k]) <= best.t)) break;  // small trees (§4.4b): a hit in a nearer leaf culls the leaves behind it (sorted: all of them)
      }
      if (COUNT) {
        sc.tris += ((ref[k] >> 28) & 7u) + 1u;
        const unsigned long long m = __ballot(1);  // the lanes inside this copy of the leaf test
        sc.leaf_lanes++;
        if ((uint32_t)__ffsll((long long)m) - 1u == (threadIdx.x & 63u)) sc.leaf_passes++;
      }
      if (leaf_test_staged<ANY, ALPHA>(sv, lds, r, t.tmax, t.key, t.tau, best, ref[k] & 0x0fffffffu, ((ref[k] >> 28) & 7u) + 1u)) { found = true; break; }
    }
  } else {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(lf[0]), m1 = __builtin_amdgcn_ballot_w64(lf[1]), m2 = __builtin_amdgcn_ballot_w64(lf[2]),
                             m3 = __builtin_amdgcn_ballot_w64(lf[3]);  // (the builtin on a bool: HIP's __ballot compares an int with 0 in the vector unit)
    if ((kSlotOrder ? (m0 | m1 | m2 | m3) : m0) != 0ull) {  // wave-uniform: some lane reached a leaf in this step (sorted: m0 covers m1, m2, m3)
      // items of the lanes below this one: one mbcnt chain over the four masks (the accumulator operand of v_mbcnt is free)
      const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3,
                           __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2,
                           __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, mbcnt64(m0)))))));
      const uint32_t total = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2) + (uint32_t)__popcll(m3);  // leaf items of the wave
      RT_LDS u32x2* items = lds.items;
      RT_LDS u32x4* slots = lds.slots;
      if (!kSlotOrder) {  // the leaves in reach are a prefix of the sorted children: the lane's k-th item is child k
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lf[k]) items[pre + (uint32_t)k] = u32x2{ref[k], lane};
      } else {
        uint32_t pos = pre;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (lf[k]) items[pos] = u32x2{ref[k], lane};
          pos += lf[k] ? 1u : 0u;
        }
      }
      const bool any_leaf = kSlotOrder ? (lf[0] | lf[1] | lf[2] | lf[3]) : lf[0];
      if (any_leaf) {
        if (ANY && !ALPHA) *(RT_LDS uint32_t*)(slots + lane) = kAbsent;  // any-hit: only the "blocked by" word of the slot is used (a ray in flight is not blocked yet)
        else slots[lane] = ANY ? u32x4{kAbsent, 0u, 0u, 0u}  // + the optical depth gathered in this step
                               : u32x4{best.prim, __float_as_uint(best.t), __float_as_uint(best.u), __float_as_uint(best.v)};
      }
      // One wave, one instruction stream: its LDS operations execute in program order, so a lane sees what another lane of the wave
      // wrote by an earlier instruction.  The fences only keep the COMPILER from moving or forwarding LDS accesses across the phases.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float4* tris = reinterpret_cast<const float4*>(sv.tris);
      for (uint32_t base = 0; base < total * (uint32_t)kLeafSlots; base += 64u) {  // one pass unless > 16 leaves were reached at once
        const uint32_t s = base + lane;
        const u32x2 it = items[min(s / (uint32_t)kLeafSlots, total - 1u)];
        const uint32_t leaf = it.x, owner = it.y & 63u;
        const uint32_t j = s % (uint32_t)kLeafSlots;
        const bool valid = s / (uint32_t)kLeafSlots < total && j <= ((leaf >> 28) & 7u);
        // the owner's ray (all lanes take part in the permutes)
        const uint32_t src = owner << 2;
        const f3 o = mk3(lane_read(src, r.o.x), lane_read(src, r.o.y), lane_read(src, r.o.z));
        const f3 d = mk3(lane_read(src, r.d.x), lane_read(src, r.d.y), lane_read(src, r.d.z));
        const float tmin = lane_read(src, r.tmin);
        const float tlim = ANY ? lane_read(src, t.tmax) : 0.0f;
        const uint32_t okey_any = (ANY && ALPHA) ? __float_as_uint(lane_read(src, __uint_as_float(t.key))) : 0u;
        const uint32_t gbase = INST ? __float_as_uint(lane_read(src, __uint_as_float(t.gid_base))) : 0u;  // the owner may be inside an instance: local ids
        const uint32_t sbase = (INST && ANY && ALPHA) ? __float_as_uint(lane_read(src, __uint_as_float(t.shade_base))) : 0u;
        if (COUNT) {
          const unsigned long long m = __ballot(valid);
          if (valid) sc.leaf_lanes++;
          if (m && (uint32_t)__ffsll((long long)m) - 1u == lane) sc.leaf_passes++;
        }
        RT_LDS unsigned long long* okey = (RT_LDS unsigned long long*)(slots + owner);
        if (valid) {
          const float4* p = tris + (size_t)((leaf & 0x0fffffffu) + j) * 3;
          const float4 a = p[0], b = p[1], c = p[2];
          asm volatile("" ::"v"(a.w));  // the id travels with v0 (one dwordx4), not as a dependent dword load inside the hit branch
          float tt, tu, tv, det;
          if (tri_test_od(o, d, a, b, c, &tt, &tu, &tv, &det) && tt > tmin) {
            if (ANY) {  // any blocking triangle: the owner's prim field leaves kAbsent
              if (tt < tlim) {
                bool blocks = true;
                if (ALPHA && __float_as_uint(b.w) != 0u) {  // translucent and / or the boundary of a medium (7.1d, 7.1g)
                  uint32_t q[3];
                  const uint32_t gid = __float_as_uint(a.w) + gbase;
                  blocks = any_hit_event(sv, okey_any, gid, INST ? hit_record_of(sv, gid, sbase, __float_as_uint(a.w)) : gid, __float_as_uint(b.w), tt, det, tu, tv, q);
                  if (!blocks && (q[0] | q[1] | q[2])) {
                    RT_LDS uint32_t* ot = (RT_LDS uint32_t*)okey;
                    __hip_atomic_fetch_add(ot + 1, q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(ot + 2, q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(ot + 3, q[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  }
                }
                // the owner's prim word leaves kAbsent: it becomes the BVH-order index of a blocker (which one of several does not matter) —
                // the occluder cache of the connection launches tests that triangle first for the lane's next ray (integrator.hip)
                if (blocks) *(RT_LDS uint32_t*)okey = (leaf & 0x0fffffffu) + j;
              }
            }
            else {
              const unsigned long long mine = ((unsigned long long)__float_as_uint(tt) << 32) | (unsigned long long)hit_encode(__float_as_uint(a.w) + gbase, __float_as_uint(c.w));  // id << 3 | shading kind
              __hip_atomic_fetch_min(okey, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              // the merge that stands publishes its barycentrics (keys are unique: one triangle, one item).  Still inside the branch: every
              // lane that merged executes the ds_min_u64 in one instruction, the read-back in the next — the LDS works them off in that order
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              if (*okey == mine) ((RT_LDS u32x2*)okey)[1] = u32x2{__float_as_uint(tu), __float_as_uint(tv)};
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (any_leaf) {
        u32x4 w4;
        if (ANY && !ALPHA) w4.x = *(RT_LDS uint32_t*)(slots + lane); else w4 = slots[lane];
        if (ANY) {
          found = w4.x != kAbsent;
          if (found) best.prim = w4.x;  // the blocker's BVH-order index
          if (ALPHA) { t.tau[0] += w4.y; t.tau[1] += w4.z; t.tau[2] += w4.w; }
        } else { best.prim = w4.x; best.t = __uint_as_float(w4.y); best.u = __uint_as_float(w4.z); best.v = __uint_as_float(w4.w); }
      }
    }
    if (!work) { if (kInstBatch > 0) t.sp = sp; return done_early; }  // no ray, or a lane that waits at an instance transition (or ended in one)
  }
  if (ANY && found) return true;  // (best.prim != kAbsent: the blocker)
  // go on with the nearest inner child if it is still in reach, else with the first stack entry that is
  if (!kSlotOrder && next != kAbsent && !(key_tn(next_key) <= best.t)) next = kAbsent;
  if (!INST) {
    while (next == kAbsent) {
      if (sp == 0) return true;
      --sp;
      uint2 e;
      if (sp < kS) { const u32x2 v = stack[sp * kTraverseThreads]; e = make_uint2(v.x, v.y); } else e = spill[sp - kS];
      if (kSlotOrder || key_tn(e.x) <= best.t) next = e.y;
    }
  } else {
    for (;;) {
      if (next == kAbsent) {
        if (sp == 0) return true;
        const uint2 e = pop();
        if (e.y == kExitRef) {  // the instance's tree is done: back to the world-space ray parked under the mark
          if (kInstBatch > 0) { next = kExitRef; break; }  // ... with the wave's next batch of transitions
          leave_instance();
          continue;
        }
        if (kSlotOrder || key_tn(e.x) <= best.t) next = e.y;
        continue;
      }
      if (kInstBatch == 0 && is_inst_leaf(next)) next = enter_instance(next);  // (batched: the lane waits at the leaf)
      break;
    }
  }
  t.cur = next;
  t.sp = sp;
  return false;
}

}  // namespace rt
