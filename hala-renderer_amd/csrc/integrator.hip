// integrator.hip — the wavefront path tracer of libhalart.so: everything the reference gets from
// `trace_rays(width, height, 1)` (src/rt_renderer.rs:458-464), as HIP kernels for gfx950.
//
// One update() = one sample per pixel =
//   for each bounce { traverse_closest (depth 0: camera rays generated in place) -> shade (+ballot compaction) ->
//   traverse_shadow } -> resolve
// All queue sizes live in a device control block; nothing returns to the host inside a frame.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "shading.h"
#include "traverse.h"

namespace rt {
// Streaming stores (queue entries, hit records; RT_PLAIN_STORES restores ordinary ones): written once, read by the next launch at the
// earliest — nontemporal stores keep them from displacing what the traversal and shading gathers want in L2 (+1.4 ... 2.2 % on the
// headline).  The 12-B per-path records stay ordinary stores: three scalar nontemporal stores each measured slower.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
#ifndef RT_PLAIN_STORES
RT_DI void st4(float4* p, float4 v) { __builtin_nontemporal_store(v4f_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_nt*>(p)); }
#else
RT_DI void st4(float4* p, float4 v) { *p = v; }
#endif

// wave64 helpers ------------------------------------------------------------------------------------------
RT_DI uint32_t lane_id() { return threadIdx.x & 63u; }
RT_DI uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_down((int)v, off);
  return v;  // valid in lane 0
}

// Block-level compaction of up to two predicates at once: one atomic per workgroup and counter instead of one per
// wave (a single counter word takes ~88 atomics/us; a 2 M-path launch of 64-lane waves would need 32 K of them).
// All threads of the block must call it.  Returns this lane's output index for each predicate.
// Workgroup size of k_shade: 512 for the SIMPLE variant, 256 for the generic one (configs[3]: 512 / 256 threads = 3.33 / 3.21 ms of shade
// per frame — fewer waves behind every barrier of the kind sort and of the compaction — while the Cornell box loses 25 % at 256:
// profiles/r02_experiments.txt).  kShadeThreads sizes the shared arrays.
#ifndef RT_SHADE_THREADS
#define RT_SHADE_THREADS 512
#endif
#ifndef RT_SHADE_THREADS_GENERIC
#define RT_SHADE_THREADS_GENERIC 256
#endif
constexpr int kShadeThreads = RT_SHADE_THREADS;
constexpr int kShadeThreadsGeneric = RT_SHADE_THREADS_GENERIC < RT_SHADE_THREADS ? RT_SHADE_THREADS_GENERIC : RT_SHADE_THREADS;
struct BlockCompact {
  uint32_t cnt[3][kShadeThreads / 64];
  uint32_t base[3];
};
// Three predicates at once (surviving path, light connection, environment connection); counters[k] receives the
// block total of predicate k; shadow_total (64-bit) receives the number of connections (predicates 1 + 2).
RT_DI void block_compact3(BlockCompact& sm, const bool keep[3], uint32_t* const counters[3], unsigned long long* shadow_total,
                          uint32_t out[3]) {
  unsigned long long m[3];
  const uint32_t w = threadIdx.x >> 6, l = lane_id();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    m[k] = __ballot(keep[k]);
    if (l == 0u) sm.cnt[k][w] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  const uint32_t nw = blockDim.x >> 6;
  if (threadIdx.x < 3u) {
    uint32_t total = 0;
    for (uint32_t j = 0; j < nw; ++j) total += sm.cnt[threadIdx.x][j];
    sm.base[threadIdx.x] = total ? atomicAdd(counters[threadIdx.x], total) : 0u;
  } else if (threadIdx.x == 64u) {
    uint32_t total = 0;
    for (uint32_t j = 0; j < nw; ++j) total += sm.cnt[1][j] + sm.cnt[2][j];
    if (total) atomicAdd(shadow_total, (unsigned long long)total);
  }
  __syncthreads();
  const unsigned long long below = (1ull << l) - 1ull;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint32_t p = sm.base[k];
    for (uint32_t j = 0; j < w; ++j) p += sm.cnt[k][j];
    out[k] = p + (uint32_t)__popcll(m[k] & below);
  }
}

// Sharded dequeue of rays for the persistent kernels (see WorkCounters): the queue [0, n) is cut into kWorkShards
// contiguous shards, each with its own counter; a wave works on one shard until it is dry.  Wave-uniform.
struct WorkCursor {
  uint32_t shard, per;
  unsigned long long dead;  // shards that lie wholly beyond the queue's end: never probed
};
RT_DI WorkCursor work_begin(uint32_t n) {
  WorkCursor c;
  const uint32_t batches = (n + kWorkBatch - 1u) / kWorkBatch;
  c.per = ((batches + kWorkShards - 1u) / kWorkShards) * kWorkBatch;  // rays per shard, a multiple of the batch
  const uint32_t live = c.per ? min(kWorkShards, (n + c.per - 1u) / c.per) : 0u;
  c.dead = live >= 64u ? 0ull : ~((1ull << live) - 1ull);
  c.shard = live ? blockIdx.x % live : 0u;
  return c;
}
// Variable-size dequeue for the refill loop: asks for `want` rays, is granted 1..want of them from one shard.
RT_DI uint32_t work_take(WorkCounters* wc, WorkCursor& c, uint32_t n, uint32_t want, uint32_t* got) {
  constexpr unsigned long long kAll = kWorkShards >= 64u ? ~0ull : (1ull << (kWorkShards & 63u)) - 1ull;
  if ((c.dead & kAll) == kAll) { *got = 0; return kAbsent; }
  for (;;) {
    uint32_t v = 0;
    if (lane_id() == 0u) v = atomicAdd(&wc->c[c.shard * kWorkStride], want);
    v = (uint32_t)__shfl((int)v, 0);
    const unsigned long long lo = (unsigned long long)c.shard * c.per + v;
    const unsigned long long hi = min((unsigned long long)(c.shard + 1u) * c.per, (unsigned long long)n);
    if (v < c.per && lo < hi) {
      *got = (uint32_t)min((unsigned long long)want, hi - lo);
      return (uint32_t)lo;
    }
    // this shard is dry: publish that (once) and move to a shard nobody has reported dry yet.  The mask only ever
    // gains bits and a bit is set only after the shard's last ray was handed out, so a stale read costs at most an
    // extra probe and "all dry" is never reported early.
    unsigned long long m = 0;
    if (lane_id() == 0u) {
      m = __hip_atomic_load(&wc->dry[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(m & (1ull << c.shard))) { atomicOr(&wc->dry[0], 1ull << c.shard); m |= 1ull << c.shard; }
    }
    m = (unsigned long long)__shfl((long long)m, 0) | c.dead;
    if ((m & kAll) == kAll) { *got = 0; return kAbsent; }
    const unsigned long long avail = ~m & kAll;
    const uint32_t rot = (c.shard + 1u) & (kWorkShards - 1u);
    const unsigned long long r = rot ? (((avail >> rot) | (avail << (kWorkShards - rot))) & kAll) : avail;
    c.shard = (rot + (uint32_t)__ffsll((long long)r) - 1u) & (kWorkShards - 1u);
  }
}

// The persistent wavefront loop: every lane owns at most one ray in flight; whenever `refill` or more lanes are idle
// (and the queue is not dry) the wave dequeues exactly that many rays and hands them to its idle lanes by ballot rank
// — consecutive queue entries go to consecutive idle lanes, so refill loads stay as coalesced as the holes allow.
// Source: load(i, &o, &d, &tmin, &tmax) fetches queue entry i (false: no ray there); done(i, trav, payload) consumes the result.
template <bool ANY, bool COUNT, bool STAGED, bool ALPHA, bool INST, class Source>
RT_DI void persistent_trace(const SceneView& sv, const TraverseLds& lds, uint2* spill, WorkCounters* work, uint32_t n, uint32_t refill,
                            Source& src, StepCounters& sc) {
  WorkCursor cur = work_begin(n);
  bool more = n > 0u;  // wave-uniform: some shard may still hold rays
  bool has = false;
  uint32_t idx = 0;
  Trav t;
  typename Source::Payload pay;
  // Whole-wave batches (refill == 64: the LDS-staged scenes, whose rays are short) ask for their NEXT batch while the
  // current one is traced: the returning atomic's round trip (1-2 us, about as long as tracing 64 such rays) is off the
  // critical path.  A speculative request on a shard that has run dry just overshoots its counter; it is resolved —
  // taken if it lies inside the shard, else replaced by the searching dequeue — before anything else is asked for.
  bool pre_out = false;  // wave-uniform: a request is outstanding
  uint32_t pre_raw = 0, pre_shard = 0;
  // A finished ray keeps its result in the lane until the wave next refills (or ends): the results of all lanes that finished in
  // between leave in ONE store instruction — whole 1-KB runs for the whole-wave batches of the LDS-staged kernels (64 consecutive
  // queue entries: full lines for the nontemporal stores) — instead of one divergent store per lane and step.
  bool fin = false;
  // Occluder cache (-DRT_OCCLUDER_CACHE; connection launches of large one-level trees): the triangle that blocked the lane's previous
  // connection is tested first against its next one — consecutive rays of a lane are queue neighbours, i.e. connections of nearby surface
  // points towards the same light or the same part of the sky; 23 % of the light and 58 % of the environment connections of configs[3] are
  // blocked, often by the same large triangles (walls).  A hit inside (0, tmax) on a fully blocking triangle IS an any-hit result: the
  // connection is dropped without a traversal; otherwise the ray is traced as before.  Images cannot depend on it (the suite passes with
  // it).  It does not pay: the extra 48-B gather and triangle test of EVERY connection cost more than the traversals the hits save.
  constexpr bool kCache = ANY && !STAGED && !INST && Source::kOccluderCache;
  uint32_t last_occ = kAbsent;
  for (;;) {
    const unsigned long long idle = __ballot(!has);
    if (more && idle) {
      if (fin) { src.done(idx, t, pay); fin = false; }
      uint32_t got = 0, base = kAbsent;
      if (pre_out) {
        const uint32_t v = (uint32_t)__shfl((int)pre_raw, 0);
        const unsigned long long lo = (unsigned long long)pre_shard * cur.per + v;
        const unsigned long long hi = min((unsigned long long)(pre_shard + 1u) * cur.per, (unsigned long long)n);
        if (v < cur.per && lo < hi) { base = (uint32_t)lo; got = (uint32_t)min(64ull, hi - lo); }
        pre_out = false;
      }
      if (base == kAbsent) base = work_take(work, cur, n, (uint32_t)__popcll(idle), &got);
      if (base == kAbsent) more = false;
      else {
        if (STAGED && refill == 64u) {
          if (lane_id() == 0u) pre_raw = atomicAdd(&work->c[cur.shard * kWorkStride], 64u);
          pre_shard = cur.shard; pre_out = true;
        }
        if (!has) {
          const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane_id()) - 1ull));
          if (rank < got) {
            idx = base + rank;
            f3 o, d; float tmin, tmax; uint32_t key = 0u;
            if (src.load(idx, &o, &d, &tmin, &tmax, &key, &pay)) {  // false: the source had no ray for this entry and has dealt with it
              bool blocked = false;
              if (kCache && last_occ != kAbsent) {
                const float4* tp = reinterpret_cast<const float4*>(sv.tris) + (size_t)last_occ * 3;
                const float4 a = tp[0], b = tp[1], c = tp[2];
                float tt, tu, tv, det;
                if (COUNT) sc.tris++;
                // flag 0: blocks every ray (a translucent triangle or a medium boundary decides per ray: RENDER_SPEC 7.1d / 7.1g — not cached)
                blocked = tri_test_od(o, d, a, b, c, &tt, &tu, &tv, &det) && tt > maxf(tmin, 0.0f) && tt < tmax && __float_as_uint(b.w) == 0u;
              }
              if (!blocked) {
                trav_begin(t, make_ray(o, d, tmin), tmax, key);
                has = true;
              }  // else: occluded — a connection that is blocked leaves nothing behind (ShadowSource::done does nothing for it)
            }
          }
        }
      }
    }
    if (__ballot(has) == 0ull) { if (!more) break; continue; }
    for (;;) {
      if (COUNT && lane_id() == 0u) sc.wave_steps++;  // lane 0 runs every iteration of this wave-uniform loop
      // every lane calls it: the large-scene variant deals the wave's leaf work out over all 64 lanes (traverse.h)
#ifdef RT_NO_DEFER
      if (trav_step<ANY, COUNT, STAGED, ALPHA, INST>(sv, lds, spill, t, has, sc)) { src.done(idx, t, pay); has = false; }
#else
      if (trav_step<ANY, COUNT, STAGED, ALPHA, INST>(sv, lds, spill, t, has, sc)) {
        fin = true; has = false;
        if (kCache && t.best.prim != kAbsent) last_occ = t.best.prim;
      }
#endif
      const uint32_t nidle = (uint32_t)__popcll(__ballot(!has));
      if (nidle == 64u || (more && nidle >= refill)) break;
    }
  }
  if (fin) src.done(idx, t, pay);
}

struct BatchSource {
  static constexpr bool kOccluderCache = false;  // a caller's batch: traced ray by ray as the spec says (step counts are pinned to the oracle's)
  struct Payload {};
  const hala_ray* rays;
  hala_hit* hits;
  bool any;
  bool queue;  // the renderer's own bounce-ray queue: tmin = 0, tmax = FLT_MAX, their fields carry path slot and RNG counter
  // Hit records of whole-wave batches (LDS-staged kernels) leave as full 1-KB runs: nontemporal stores.  Per-lane refills finish
  // scattered entries at scattered times: ordinary stores, so that L2 puts the 16-B pieces of a line together before it goes to
  // memory (nontemporal 16-B pieces reached the fabric at 1.8-2.1x their bytes: profiles/r01_m_pmc_config2.txt, r02_a_pmc_config4).
  bool streaming;
  RT_DI bool load(uint32_t i, f3* o, f3* d, float* tmin, float* tmax, uint32_t* key, Payload*) const {
    const float4* rp = reinterpret_cast<const float4*>(rays + i);
    const float4 ro = rp[0], rd = rp[1];
    *o = mk3(ro.x, ro.y, ro.z); *d = mk3(rd.x, rd.y, rd.z); *tmin = queue ? 0.0f : ro.w; *tmax = queue ? kTMax : rd.w;
    if (any) *key = pcg_hash(i ^ kAnyKeyBatch);  // RENDER_SPEC 7.1d: the key of ray i of a batch
    return true;
  }
  RT_DI void done(uint32_t i, const Trav& t, const Payload&) const {
    const bool found = t.best.prim != kAbsent;
    float4 out;
    if (any) out = make_float4(found ? 1.0f : -1.0f, 0.0f, 0.0f, __uint_as_float(kAbsent));
    // the renderer's own queue keeps id << 3 | shading kind (hala_types.h: hit_encode); a caller's batch gets the plain triangle id
    else out = found ? make_float4(t.best.t, t.best.u, t.best.v, __uint_as_float(queue ? t.best.prim : t.best.prim >> kHitKindBits)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(kAbsent));
    if (streaming) st4(reinterpret_cast<float4*>(hits) + i, out);
    else reinterpret_cast<float4*>(hits)[i] = out;
  }
};
// The camera ray of path slot `slot` (RENDER_SPEC §5) and the RNG state after it; false for the padding slots of a
// sharded frame.  Primary rays are never stored: the depth-0 traversal and the depth-0 shading both evaluate this.
RT_DI bool primary_ray(const FrameConst& fc, const SceneView& sv, uint32_t slot, f3* o, f3* d, uint32_t* rng) {
  const uint32_t sample = slot / fc.pixel_slots, pslot = slot - sample * fc.pixel_slots;
  uint32_t px = 0, py = 0;
  *o = splat3(0.0f); *d = mk3(0.0f, 0.0f, 1.0f); *rng = 0u;
  if (!slot_to_pixel(fc, pslot, &px, &py)) return false;
  uint32_t state = rng_init(py * fc.width + px, fc.u.frame_index + sample);
  camera_ray(fc, sv.cameras[fc.u.camera_index], px, py, state, o, d);
  *rng = state;
  return true;
}
struct CameraSource {
  static constexpr bool kOccluderCache = false;
  struct Payload {};
  const FrameConst& fc;
  const SceneView& sv;
  hala_hit* hits;
  bool streaming;
  RT_DI bool load(uint32_t i, f3* o, f3* d, float* tmin, float* tmax, uint32_t*, Payload*) const {
    uint32_t rng;
    *tmin = 0.0f; *tmax = kTMax;
    if (primary_ray(fc, sv, i, o, d, &rng)) return true;
    reinterpret_cast<float4*>(hits)[i] = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(kAbsent));  // padding slot: no ray
    return false;
  }
  RT_DI void done(uint32_t i, const Trav& t, const Payload&) const {
    const bool found = t.best.prim != kAbsent;
    const float4 out = found ? make_float4(t.best.t, t.best.u, t.best.v, __uint_as_float(t.best.prim)) : make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(kAbsent));
    if (streaming) st4(reinterpret_cast<float4*>(hits) + i, out);
    else reinterpret_cast<float4*>(hits)[i] = out;
  }
};
#ifndef RT_SHADOW_PREFETCH
#define RT_SHADOW_PREFETCH 1  // large scenes too: 84 of the 102 VGPRs of 5 waves per SIMD are in use, the three words fit (shadow 4.06 -> 3.99 ms, configs[3])
#endif
template <bool PREFETCH, bool ALPHA>
struct ShadowSource {
#ifdef RT_OCCLUDER_CACHE  // measured: configs[3] shadow 3.95 -> 4.05 ms per frame with it (profiles/r03_experiments.txt): off
  static constexpr bool kOccluderCache = true;
#else
  static constexpr bool kOccluderCache = false;
#endif
  // contribution.xyz | pixel slot, fetched with the ray (one coalesced 48-B record), and the path's radiance as it stands
  struct Payload { float4 cs; float lx, ly, lz; };
  const ShadowEntry* entries;
  P3* radiance;
  // A path owns at most one connection per queue and the two queues add to different arrays, so nothing else touches this word of the
  // path's radiance during the launch: it is fetched here, behind the entry (the load is in flight while the ray is traced), and
  // an unoccluded ray stores radiance + contribution — one IEEE add per component, no ordering freedom.  (Three memory-side float
  // atomics per unoccluded ray did the same and cost 57 of the launch's 160 us on the headline config: profiles/r01_h_experiments.txt.)
  // PREFETCH = false (kept for register-starved variants): the add is done by fire-and-forget float atomics instead — at most one per
  // radiance word per launch, so exactly the same single IEEE add.  (The two connection kinds of a bounce share a launch but not an
  // array: light connections add to L, environment connections to Le.)
  RT_DI bool load(uint32_t i, f3* o, f3* d, float* tmin, float* tmax, uint32_t* key, Payload* p) const {
    const float4* e = reinterpret_cast<const float4*>(entries + i);
    const float4 ro = e[0], rd = e[1];
    p->cs = e[2];
    if (PREFETCH) {
      const float* l = reinterpret_cast<const float*>(radiance + __float_as_uint(p->cs.w));
      p->lx = l[0]; p->ly = l[1]; p->lz = l[2];
    }
    // a connection starts at its origin (tmin = 0): the field carries its any-hit key (RENDER_SPEC 7.1d)
    *o = mk3(ro.x, ro.y, ro.z); *d = mk3(rd.x, rd.y, rd.z); *tmin = 0.0f; *tmax = rd.w; *key = __float_as_uint(ro.w);
    return true;
  }
  RT_DI void done(uint32_t, const Trav& t, const Payload& p) const {
    if (t.best.prim != kAbsent) return;  // occluded
    float cx = p.cs.x, cy = p.cs.y, cz = p.cs.z;
    if (ALPHA && (t.tau[0] | t.tau[1] | t.tau[2])) {  // RENDER_SPEC 7.1g: what the media the connection crossed leave of it
      const f3 tr = any_transmittance(t.tau);
      if (tr.x != 1.0f || tr.y != 1.0f || tr.z != 1.0f) { cx = cx * tr.x; cy = cy * tr.y; cz = cz * tr.z; }
    }
    float* l = reinterpret_cast<float*>(radiance + __float_as_uint(p.cs.w));
    if (PREFETCH) { l[0] = p.lx + cx; l[1] = p.ly + cy; l[2] = p.lz + cz; }
    else { atomicAdd(l + 0, cx); atomicAdd(l + 1, cy); atomicAdd(l + 2, cz); }
  }
};

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

// counting launches: one set of atomics per wave
RT_DI void flush_counters(Control* ctl, int kind, const StepCounters& sc) {
  const uint32_t v[5] = {wave_sum(sc.nodes), wave_sum(sc.tris), wave_sum(sc.wave_steps), wave_sum(sc.leaf_passes), wave_sum(sc.leaf_lanes)};
  if (lane_id() != 0u) return;
  atomicAdd(&ctl->steps[kind][0], (unsigned long long)v[0]); atomicAdd(&ctl->steps[kind][1], (unsigned long long)v[1]);
  atomicAdd(&ctl->probe[kind][0], (unsigned long long)v[2]); atomicAdd(&ctl->probe[kind][1], (unsigned long long)v[3]);
  atomicAdd(&ctl->probe[kind][2], (unsigned long long)v[4]);
}

// ---------------------------------------------------------------------------------------------------------
// K5a: persistent closest-hit traversal over a compact ray queue (coalesced 32-B ray reads, 16-B hit writes)
// ---------------------------------------------------------------------------------------------------------
template <bool ANY, bool COUNT, bool STAGED, bool ALPHA, bool INST>
__global__ void __launch_bounds__(kTraverseThreads, STAGED ? kTraverseWavesPerSimdStaged : kTraverseWavesPerSimd)
k_trace_batch(SceneView sv, const hala_ray* __restrict__ rays, hala_hit* __restrict__ hits, const uint32_t* __restrict__ n_ptr,
              uint32_t n_imm, WorkCounters* __restrict__ work, uint2* __restrict__ spill_base, Control* __restrict__ ctl, int account,
              uint32_t refill) {
  const TraverseLds lds = stage_bvh<STAGED, INST>(sv, g_smem);
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  uint2* spill = spill_base ? spill_base + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kStackSpill : nullptr;
  StepCounters sc;
  if (account && blockIdx.x == 0 && threadIdx.x == 0) {
    if (ANY) ctl->rays_shadow += n; else ctl->rays_closest += n;
  }
  BatchSource src{rays, hits, ANY, account != 0, STAGED && refill == 64u};
  persistent_trace<ANY, COUNT, STAGED, ALPHA, INST>(sv, lds, spill, work, n, refill, src, sc);
  if (COUNT) flush_counters(ctl, ANY ? 1 : 0, sc);
}

// K5a at depth 0: the camera rays are generated in the lanes that trace them (RENDER_SPEC §5) — no ray-generation kernel,
// no primary-ray queue in HBM; entry i of the hit queue belongs to path slot i.  `n_account` = real (non-padding) paths.
template <bool COUNT, bool STAGED, bool INST>
__global__ void __launch_bounds__(kTraverseThreads, STAGED ? kTraverseWavesPerSimdStaged : kTraverseWavesPerSimd)
k_trace_primary(SceneView sv, FrameConst fc, hala_hit* __restrict__ hits, WorkCounters* __restrict__ work, uint2* __restrict__ spill_base,
                Control* __restrict__ ctl, uint32_t n_account, uint32_t refill) {
  const TraverseLds lds = stage_bvh<STAGED, INST>(sv, g_smem);
  uint2* spill = spill_base ? spill_base + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kStackSpill : nullptr;
  StepCounters sc;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->rays_closest += n_account;
  CameraSource src{fc, sv, hits, STAGED && refill == 64u};
  persistent_trace<false, COUNT, STAGED, false, INST>(sv, lds, spill, work, fc.slot_count, refill, src, sc);
  if (COUNT) flush_counters(ctl, 0, sc);
}

// ---------------------------------------------------------------------------------------------------------
// K5c: shadow traversal of the NEE connections of one bounce; unoccluded contributions are added to the
// path's radiance in the fixed order light, environment (RENDER_SPEC §6)
// ---------------------------------------------------------------------------------------------------------
template <bool COUNT, bool STAGED, bool ALPHA, bool INST>
__global__ void __launch_bounds__(kTraverseThreads, STAGED ? kTraverseWavesPerSimdStaged : kTraverseWavesPerSimd)
k_trace_shadow(SceneView sv, Queues q, PathState ps, Control* __restrict__ ctl, uint32_t depth, uint32_t kind, uint2* __restrict__ spill_base,
               uint32_t refill) {
  const TraverseLds lds = stage_bvh<STAGED, INST>(sv, g_smem);
  const uint32_t n = ctl->n_shadow[kind][depth];
  uint2* spill = spill_base ? spill_base + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kStackSpill : nullptr;
  StepCounters sc;
  ShadowSource<(STAGED || RT_SHADOW_PREFETCH), ALPHA> src{q.shadow[kind], kind ? ps.radiance_env : ps.radiance};
  persistent_trace<true, COUNT, STAGED, ALPHA, INST>(sv, lds, spill, &ctl->work_shadow[kind], n, refill, src, sc);
  if (COUNT) flush_counters(ctl, 1, sc);
}

// ---------------------------------------------------------------------------------------------------------
// K5c + K5a in one launch: the shadow passes of bounce d and the closest-hit traversal of bounce d + 1.  They are independent (a light
// connection only adds to its path's L, an environment connection to its Le, the closest-hit pass only reads the ray queue shade(d)
// wrote) and every persistent launch ends in a tail of a few long rays during which most of the chip idles — a fixed cost that weighs
// more the smaller the launch (late bounces, a rank's share of a strong-scaled frame; 10-20 % of the wave slots of a configs[3] launch).
// Here a wave that finds one queue dry moves straight on to the next: one tail instead of three.  (Launches side by side on two streams
// do not achieve this: each takes the chip from the other all the time, profiles/r02_experiments.txt.)  Not used by updates that carry
// per-launch timing events or counting kernels.
// ---------------------------------------------------------------------------------------------------------
template <bool STAGED, bool ALPHA, bool INST>
__global__ void __launch_bounds__(kTraverseThreads, STAGED ? kTraverseWavesPerSimdStaged : kTraverseWavesPerSimd)
k_trace_shadow_then_batch(SceneView sv, const Tri* __restrict__ tris_any, Queues q, PathState ps, Control* __restrict__ ctl, uint32_t depth,
                          uint32_t kinds, const hala_ray* __restrict__ rays, hala_hit* __restrict__ hits, uint2* __restrict__ spill_base, uint32_t refill) {
  const TraverseLds lds = stage_bvh<STAGED, INST>(sv, g_smem);
  uint2* spill = spill_base ? spill_base + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kStackSpill : nullptr;
  StepCounters sc;
  {
    SceneView sva = sv;
    sva.tris = tris_any;  // RENDER_SPEC 7.1d (STAGED: the launcher only fuses when both passes traverse the same triangles)
    for (uint32_t kind = 0; kind < 2u; ++kind) {  // bit 0: light connections, bit 1: environment connections
      if (!((kinds >> kind) & 1u)) continue;
      ShadowSource<(STAGED || RT_SHADOW_PREFETCH), ALPHA> src{q.shadow[kind], kind ? ps.radiance_env : ps.radiance};
      persistent_trace<true, false, STAGED, ALPHA, INST>(sva, lds, spill, &ctl->work_shadow[kind], ctl->n_shadow[kind][depth], refill, src, sc);
    }
  }
  if (rays == nullptr) return;  // the last bounce: its two shadow passes share the launch, no closest-hit pass follows
  const uint32_t n = ctl->n_active[depth + 1u];
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->rays_closest += n;
  BatchSource src{rays, hits, false, true, STAGED && refill == 64u};
  persistent_trace<false, false, STAGED, false, INST>(sv, lds, spill, &ctl->work_closest, n, refill, src, sc);
}

// ---------------------------------------------------------------------------------------------------------
// shade: closest-hit + miss + light/env NEE + BSDF sampling + Russian roulette for one bounce (RENDER_SPEC §6)
// ---------------------------------------------------------------------------------------------------------
// PRIMARY (depth 0): queue entry i IS path slot i, its ray is re-evaluated from the camera instead of being read, and
// the path state starts from its constants (throughput 1, radiance 0) — this kernel initialises every per-path record.
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
#ifndef RT_SHADE_WAVES_SIMPLE
#define RT_SHADE_WAVES_SIMPLE 6  // 80 VGPRs, no scratch (8 waves: 32-44 B of scratch per lane and slower; profiles/r02_experiments.txt)
#endif
// SCATTER: some material holds a scattering medium (§7.1f): only then does the kernel carry the free-flight / phase-function code
// Bounce launches of scenes whose materials span several shading kinds: the paths of a WINDOW of kSortWindow consecutive queue entries
// are shaded in the order of their shading kind, so that a wave runs one flavour of the BSDF code.  k_shade_sort writes the order (a
// permutation of each window: counting sort through LDS; the kind sits in the low bits of the hit record's prim word, so the sort reads
// nothing but the hit queue), k_shade reads it.  Sorting inside each 256-path workgroup of k_shade left 36 of 64 lanes per vector
// instruction — a partial wave at every kind boundary of every 4-wave block (profiles/r02_w_pmc_config4.txt); a window of 4096 has the
// same boundaries per 64 waves.  Which lane shades which path is not observable: every output is indexed by path slot or comes out
// of the block compaction.
#ifndef RT_SORT_ROUNDS
#define RT_SORT_ROUNDS 4
#endif
#ifndef RT_SORT_THREADS
#define RT_SORT_THREADS 1024  // a window is one workgroup of the sort kernel: 16 waves x 4 entries per lane (256 x 16: 27 us per launch instead of ~12)
#endif
constexpr uint32_t kSortThreads = RT_SORT_THREADS, kSortRounds = RT_SORT_ROUNDS, kSortWindow = kSortThreads * kSortRounds;
static_assert(kSortWindow <= 65536, "window-relative positions are 16-bit in LDS");
__global__ void __launch_bounds__(kSortThreads) k_shade_sort(Queues q, const Control* __restrict__ ctl, uint32_t depth) {
  const uint32_t n = ctl->n_active[depth];
  const uint32_t base = blockIdx.x * kSortWindow;
  if (base >= n) return;
  const uint32_t live = min(kSortWindow, n - base);
  constexpr uint32_t kWaves = kSortThreads / 64, kCells = kShadeKinds * kSortRounds * kWaves;
  __shared__ uint32_t s_bin[kCells];  // [kind][round][wave]: entries of that kind among the 64 the wave looked at in that round
  const uint32_t w = threadIdx.x >> 6, l = lane_id();
  uint32_t kr[kSortRounds];  // kind | rank << 4 of this thread's entry of round j
#pragma unroll
  for (uint32_t j = 0; j < kSortRounds; ++j) {
    const uint32_t e = j * kSortThreads + threadIdx.x;
    uint32_t kind = kShadeKinds - 1u;  // no path
    if (e < live) {
      const uint32_t pw = reinterpret_cast<const uint32_t*>(q.hits + base + e)[3];
      kind = pw == kAbsent ? kShadeKindMiss : (pw & 7u);
    }
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t b = 0; b < kShadeKinds; ++b) {
      const unsigned long long m = __ballot(kind == b);
      if (kind == b) rank = mbcnt64(m);
      if (l == 0u) s_bin[(b * kSortRounds + j) * kWaves + w] = (uint32_t)__popcll(m);
    }
    kr[j] = kind | (rank << 4);
  }
  __syncthreads();
  if (w == 0u) {  // exclusive prefix over the cells in (kind, round, wave) order, by one wave: each lane folds a run of cells, the runs are scanned with shuffles
    constexpr uint32_t kRun = (kCells + 63u) / 64u;
    uint32_t v[kRun], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRun; ++k) { const uint32_t c = l * kRun + k; v[k] = c < kCells ? s_bin[c] : 0u; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (l >= (uint32_t)off) incl += o; }
    uint32_t run = incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < kRun; ++k) { const uint32_t c = l * kRun + k; if (c < kCells) s_bin[c] = run; run += v[k]; }
  }
  __syncthreads();
  // "no path" entries (beyond the queue's end) sort last: positions >= live, never read
#pragma unroll
  for (uint32_t j = 0; j < kSortRounds; ++j) {
    const uint32_t e = j * kSortThreads + threadIdx.x;
    if (e >= live) continue;
    const uint32_t kind = kr[j] & 15u, rank = kr[j] >> 4;
    q.perm[base + s_bin[(kind * kSortRounds + j) * kWaves + w] + rank] = base + e;
  }
}

template <bool PRIMARY, bool SIMPLE, bool SCATTER>
__global__ void __launch_bounds__(SIMPLE ? kShadeThreads : kShadeThreadsGeneric, SIMPLE ? RT_SHADE_WAVES_SIMPLE : RT_SHADE_WAVES) k_shade(FrameConst fc, SceneView sv, Queues q, PathState ps, Control* __restrict__ ctl, uint32_t depth) {
  __shared__ BlockCompact s_compact;
  const uint32_t n = PRIMARY ? fc.slot_count : ctl->n_active[depth];
  // the traversal of this bounce is over and the next users (shadow pass of this bounce, closest-hit pass of the next)
  // have not started: re-arm their work counters here
  if (blockIdx.x == 0u && threadIdx.x < kWorkShards) {
    ctl->work_closest.c[threadIdx.x * kWorkStride] = 0u;
    ctl->work_shadow[0].c[threadIdx.x * kWorkStride] = 0u;
    ctl->work_shadow[1].c[threadIdx.x * kWorkStride] = 0u;
    if (threadIdx.x == 0u) { ctl->work_closest.dry[0] = 0ull; ctl->work_shadow[0].dry[0] = 0ull; ctl->work_shadow[1].dry[0] = 0ull; }
  }
  uint32_t blk = blockIdx.x;
#if !defined(RT_SHADE_NOSORT) && !defined(RT_SHADE_NO_XCD_MAP)
  if (!PRIMARY && !SIMPLE && sv.shade_sort) {
    // The workgroups that share a sort window read the same lines of the ray / state / hit queues (each takes a kind-slice of the
    // window's paths).  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2): give every
    // window to ONE XCD — hardware blocks x, x + 8, x + 16, ... of a group of 8 windows take the consecutive logical blocks of window x.
    constexpr uint32_t kBPW = kSortWindow / kShadeThreadsGeneric, kGroup = 8u * kBPW;  // (the launcher rounds the grid up to whole groups)
    const uint32_t wi = blk % kGroup;
    blk = blk - wi + (wi & 7u) * kBPW + (wi >> 3);
  }
#endif
  if (blk * blockDim.x >= n) return;  // whole workgroup beyond the queue (uniform exit: barriers below)
  uint32_t i = blk * blockDim.x + threadIdx.x;
  // the texel decode table in LDS (shading.h::tex8_fetch); the barriers of the block compaction come long after every read of it
  __shared__ float s_tex_lut[SIMPLE ? 1 : kTexLutEntries];
  const RT_LDS float* lut = (const RT_LDS float*)s_tex_lut;
  if (!SIMPLE && sv.texture_count) {
    for (uint32_t k = threadIdx.x; k < kTexLutEntries; k += blockDim.x) s_tex_lut[k] = sv.tex_lut[k];
    __syncthreads();
  }
#ifndef RT_SHADE_NOSORT
  if (!PRIMARY && !SIMPLE && sv.shade_sort && i < n) i = q.perm[i];  // kind order inside the window (k_shade_sort)
#endif
  const uint32_t in = depth & 1u, out = in ^ 1u;
  {
  const bool active = i < n;

  bool keep[3] = {false, false, false};  // path survives, light connection, environment connection
  f3 no = splat3(0.0f), nd = splat3(0.0f);
  float4 conn[2][3];  // the two NEE connections of this path, written to the compact queues after the block scan
  uint32_t slot = 0;
  f3 o, d;
  uint32_t rng = 0, rng_out = 0;                          // RNG counter on entry / for the next queue entry
  float4 state_out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // throughput | pdf for the next queue entry
  bool real = active;  // false: padding slot of a sharded frame (depth 0 only; later queues hold real paths only)
  if (PRIMARY && active) {
    slot = i;
    real = primary_ray(fc, sv, slot, &o, &d, &rng);
    if (fc.u.env_type == 1u) ps.radiance_env[slot] = P3{0.0f, 0.0f, 0.0f};
    if (!real) {  // resolve reads every slot of the rank's tile buffer
      ps.radiance[slot] = P3{0.0f, 0.0f, 0.0f};
      ps.albedo[slot] = P3{0.0f, 0.0f, 0.0f};
      ps.normal[slot] = P3{0.0f, 0.0f, 0.0f};
    }
  }
  if (real) {
    // L: what this bounce adds to the path's radiance (at most one term: light hit | environment | emission)
    f3 T = splat3(1.0f), L = splat3(0.0f);
    float prev_pdf = kNoNeePdf;  // no vertex has sampled a direction yet: an emitter reached through skipped (7.1d) surfaces counts in full
    if (!PRIMARY) {  // the whole state of a live path is its queue entry: coalesced reads, no gather by slot
      const float4* rp = reinterpret_cast<const float4*>(q.rays[in] + i);
      const float4 ro = rp[0], rd = rp[1], st = q.state[in][i];
      o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z);
      slot = __float_as_uint(ro.w); rng = __float_as_uint(rd.w);
      T = mk3(st.x, st.y, st.z); prev_pdf = st.w;
    }
    const float4 hv = reinterpret_cast<const float4*>(q.hits)[i];
    const uint32_t hit_word = __float_as_uint(hv.w);  // id << 3 | shading kind (hala_types.h: hit_encode) or kAbsent
    const uint32_t hit_prim = hit_word == kAbsent ? kAbsent : hit_word >> kHitKindBits;
    const uint32_t nl = fc.u.num_of_lights;
    const float t_surf = hit_prim != kAbsent ? hv.x : kTMax;

    int hit_light = -1;
    float t_light = t_surf, light_pdf = 0.0f;
    for (uint32_t k = 0; k < nl; ++k) {
      float lp;
      const float tl = intersect_light(sv.lights[k], o, d, &lp);
      if (tl > 0.0f && tl < t_light) { t_light = tl; hit_light = (int)k; light_pdf = lp; }
    }
    if (hit_light >= 0) {
      const f3 le = ld3(sv.lights[hit_light].intensity);
      float w = 1.0f;
      if (!PRIMARY) w = power_heuristic(prev_pdf, light_pdf * (1.0f / (float)nl));
      L = L + T * le * w;
      if (PRIMARY) {
        ps.albedo[slot] = P3{minf(le.x, 1.0f), minf(le.y, 1.0f), minf(le.z, 1.0f)};
        ps.normal[slot] = P3{0.0f, 0.0f, 0.0f};
      }
    } else if (hit_prim == kAbsent) {
      f3 env;
      float w = 1.0f;
      if (fc.u.env_type == 1u) {
        env = env_map_eval(fc, sv, d);
        if (!PRIMARY) w = power_heuristic(prev_pdf, env_map_pdf(fc, sv, d));
      } else env = sky_eval(fc, d);
      L = L + T * env * w;
      if (PRIMARY) {
        ps.albedo[slot] = P3{minf(env.x, 1.0f), minf(env.y, 1.0f), minf(env.z, 1.0f)};
        ps.normal[slot] = P3{0.0f, 0.0f, 0.0f};
      }
    } else {
      const Surface sf = make_surface<SIMPLE>(sv, lut, fc.pixel_spread, o, d, hv.x, hv.y, hv.z, hit_prim);
      if (PRIMARY) {
        ps.albedo[slot] = P3{sf.mat.base.x, sf.mat.base.y, sf.mat.base.z};
        ps.normal[slot] = P3{sf.ns.x, sf.ns.y, sf.ns.z};
      }
      // §7.1e: what the medium of an object just crossed did to the segment that ends here (identity for every other hit)
      if (sf.glow.x > 0.0f || sf.glow.y > 0.0f || sf.glow.z > 0.0f) {
        const f3 g = T * sf.glow;
        if (PRIMARY) L = L + g;
        else {  // a second term may follow in this bounce (emission): this one goes to the path's radiance right away, in order
          const P3 l = ps.radiance[slot];
          ps.radiance[slot] = P3{l.x + g.x, l.y + g.y, l.z + g.z};
        }
      }
      T = T * sf.absorb;
      bool through = false, scattered = false;
      f3 pm = sf.P;        // the vertex the path is at: the surface point, or the scattering point inside the medium
      float hg_g = 0.0f;
      if (SCATTER && sf.sigma > 0.0f) {  // §7.1f: free flight through a scattering medium; the surface is only reached if the flight outlasts the segment
        const float rs = rng_next(rng);
        const float dist = -log_poly(1.0f - rs) / sf.sigma;
        if (dist < hv.x) {
          T = T * sf.scol;
          pm = madd3(d, dist, o);  // no offset: the point is inside the object
          hg_g = sf.hg;
          scattered = true;
        }
      }
      if (!scattered && sf.mat.opacity < 1.0f) {  // §7.1d: the surface is skipped with probability 1 - opacity (one extra random number)
        const float ro = rng_next(rng);
        through = !(ro < sf.mat.opacity);
      }
      if (through) {
        const bool alive = depth + 1u < fc.u.max_depth;
        keep[0] = alive;
        if (alive) {
          no = madd3(sf.ng, -sv.ray_eps, sf.P);
          nd = d;
        }
      } else {
      // a vertex with next-event estimation: a surface (BSDF, cosine, offset origin) or — §7.1f — a scattering point (Henyey-Greenstein
      // phase function: value = pdf, no cosine, the connection starts at the point itself and is attenuated on its way out by §7.1g)
      if (!scattered) {
        const f3 em = sf.mat.emission;
        if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) L = L + T * em;
      }
      const f3 wo = -d;
      // the tangent frame, wo in it and the lobe weights of this vertex: shared by the two connections and the sampled continuation
      BsdfCtx bc{};
      if (!scattered) bc = bsdf_prepare(sf.mat, wo, sf.ns);
      if (nl > 0u) {  // NEE: one light
        const float rl = rng_next(rng), r1 = rng_next(rng), r2 = rng_next(rng);
        const uint32_t idx = min((uint32_t)(rl * (float)nl), nl - 1u);
        const LightSample ls = sample_light(sv.lights[idx], pm, r1, r2);
        if (ls.valid) {
          f3 fb; float pdf_b;
          if (scattered) { pdf_b = hg_phase(hg_g, dot3(d, ls.wi)); fb = splat3(pdf_b); }
          else bsdf_eval(sf.mat, bc, wo, ls.wi, sf.ns, &fb, &pdf_b);
          if (pdf_b > 0.0f) {
            const float side = dot3(ls.wi, sf.ng) >= 0.0f ? sv.ray_eps : -sv.ray_eps;
            const f3 so = scattered ? pm : madd3(sf.ng, side, sf.P);
            const float tmax = ls.dist >= kTMax ? kTMax : maxf(ls.dist - 2.0f * sv.ray_eps, 0.0f);
            const float cosl = scattered ? 1.0f : fabsf(dot3(sf.ns, ls.wi));  // |cos|: a connection may leave through the surface (§7.1c)
            f3 contrib;
            if (ls.delta) contrib = fb * ls.le * (cosl * (float)nl);
            else {
              const float pl = ls.pdf * (1.0f / (float)nl);
              const float w = power_heuristic(pl, pdf_b);
              contrib = fb * ls.le * (cosl * w / pl);
            }
            const f3 tc = T * contrib;
            conn[0][0] = make_float4(so.x, so.y, so.z, __uint_as_float(pcg_hash(rng ^ kAnyKeyLight)));  // tmin is 0: the field carries the any-hit key (7.1d)
            conn[0][1] = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, tmax);
            conn[0][2] = make_float4(tc.x, tc.y, tc.z, __uint_as_float(slot));
            keep[1] = true;
          }
        }
      }
      if (fc.u.env_type == 1u) {  // NEE: environment map
        const float r1 = rng_next(rng), r2 = rng_next(rng);
        f3 wi; float pdf_e;
        if (env_map_sample(fc, sv, r1, r2, &wi, &pdf_e)) {
          f3 fb; float pdf_b;
          if (scattered) { pdf_b = hg_phase(hg_g, dot3(d, wi)); fb = splat3(pdf_b); }
          else bsdf_eval(sf.mat, bc, wo, wi, sf.ns, &fb, &pdf_b);
          if (pdf_b > 0.0f) {
            const float side = dot3(wi, sf.ng) >= 0.0f ? sv.ray_eps : -sv.ray_eps;
            const f3 so = scattered ? pm : madd3(sf.ng, side, sf.P);
            const float cosl = scattered ? 1.0f : fabsf(dot3(sf.ns, wi));
            const float w = power_heuristic(pdf_e, pdf_b);
            const f3 col = env_map_eval(fc, sv, wi);
            const f3 tc = T * (fb * col * (cosl * w / pdf_e));
            conn[1][0] = make_float4(so.x, so.y, so.z, __uint_as_float(pcg_hash(rng ^ kAnyKeyEnv)));
            conn[1][1] = make_float4(wi.x, wi.y, wi.z, kTMax);
            conn[1][2] = make_float4(tc.x, tc.y, tc.z, __uint_as_float(slot));
            keep[2] = true;
          }
        }
      }
      // continue the path
      f3 wi = d; bool sampled;
      if (scattered) {
        const float r1 = rng_next(rng), r2 = rng_next(rng);
        wi = hg_sample(d, hg_g, r1, r2);
        prev_pdf = hg_phase(hg_g, dot3(d, wi));  // the phase function is sampled exactly: throughput unchanged, the pdf goes to the next vertex's MIS weight
        sampled = true;
      } else {
        const float r1 = rng_next(rng), r2 = rng_next(rng), r3 = rng_next(rng);
        f3 fb; float pdf_b;
        sampled = bsdf_sample(sf.mat, bc, wo, sf.ns, r1, r2, r3, &wi, &fb, &pdf_b);
        if (sampled) { T = T * fb * (fabsf(dot3(sf.ns, wi)) / pdf_b); prev_pdf = pdf_b; }
      }
      if (sampled) {
        bool alive = true;
        if (depth >= fc.u.rr_depth) {
          const float qq = minf(max3f(T), 0.95f);
          const float rr = rng_next(rng);
          if (!(rr < qq)) alive = false; else T = T * (1.0f / qq);
        }
        if (depth + 1u >= fc.u.max_depth) alive = false;
        keep[0] = alive;
        if (alive) {
          const float side = dot3(wi, sf.ng) >= 0.0f ? sv.ray_eps : -sv.ray_eps;
          no = scattered ? pm : madd3(sf.ng, side, sf.P);
          nd = wi;
        }
      }
      }  // !through
    }
    if (PRIMARY) ps.radiance[slot] = P3{L.x, L.y, L.z};
    else if (L.x != 0.0f || L.y != 0.0f || L.z != 0.0f) {  // radiance += term (this launch touches the word once: a plain update)
      const P3 l = ps.radiance[slot];
      ps.radiance[slot] = P3{l.x + L.x, l.y + L.y, l.z + L.z};
    }
    rng_out = rng;
    state_out = make_float4(T.x, T.y, T.z, prev_pdf);
  }
  // ballot compaction (block level) of the surviving paths and of the two kinds of NEE connections
  uint32_t pos[3];
  uint32_t* const counters[3] = {&ctl->n_active[depth + 1u], &ctl->n_shadow[0][depth], &ctl->n_shadow[1][depth]};
  block_compact3(s_compact, keep, counters, &ctl->rays_shadow, pos);
  if (keep[0]) {
    float4* rp = reinterpret_cast<float4*>(q.rays[out] + pos[0]);
    st4(rp, make_float4(no.x, no.y, no.z, __uint_as_float(slot)));
    st4(rp + 1, make_float4(nd.x, nd.y, nd.z, __uint_as_float(rng_out)));
    st4(&q.state[out][pos[0]], state_out);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (keep[1 + k]) {
      float4* ce = reinterpret_cast<float4*>(q.shadow[k] + pos[1 + k]);
      st4(ce, conn[k][0]); st4(ce + 1, conn[k][1]); st4(ce + 2, conn[k][2]);
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------------------
// resolve: fold this sample into the running means and write the tonemapped final image (RENDER_SPEC §8)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_resolve(FrameConst fc, PathState ps, float4* __restrict__ accum, float4* __restrict__ albedo,
                                                  float4* __restrict__ normal, float4* __restrict__ final_img) {
  const uint32_t pslot = blockIdx.x * blockDim.x + threadIdx.x;
  if (pslot >= fc.pixel_slots) return;
  // where the pixel of this slot lives in the images: sharded ranks keep their tile buffers in slot order (RENDER_SPEC §9), an unsharded
  // frame is row-major whatever the slot order
  uint32_t at = pslot;
  if (fc.world <= 1u) {
    uint32_t px, py;
    if (!slot_to_pixel(fc, pslot, &px, &py)) return;  // padding slot of a border block
    at = py * fc.width + px;
  }
  // the running means; a batch that starts an accumulation (frame_index 0) never looks at them (fold_mean)
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, n = a;
  if (fc.u.frame_index != 0u) { a = accum[at]; b = albedo[at]; n = normal[at]; }
  for (uint32_t k = 0; k < fc.samples; ++k) {  // the batch's samples, folded in frame order
    const uint32_t slot = k * fc.pixel_slots + pslot;
    const P3 lr = ps.radiance[slot];
    f3 L = mk3(lr.x, lr.y, lr.z);
    if (fc.u.env_type == 1u) { const P3 le = ps.radiance_env[slot]; L = L + mk3(le.x, le.y, le.z); }  // RENDER_SPEC §6: L + Le
    if (!(isfinite(L.x) && isfinite(L.y) && isfinite(L.z))) L = splat3(0.0f);
    const uint32_t fi = fc.u.frame_index + k;
    const P3 sa = ps.albedo[slot], sn = ps.normal[slot];
    a = make_float4(fold_mean(a.x, L.x, fi), fold_mean(a.y, L.y, fi), fold_mean(a.z, L.z, fi), 1.0f);
    b = make_float4(fold_mean(b.x, sa.x, fi), fold_mean(b.y, sa.y, fi), fold_mean(b.z, sa.z, fi), 1.0f);
    n = make_float4(fold_mean(n.x, sn.x, fi), fold_mean(n.y, sn.y, fi), fold_mean(n.z, sn.z, fi), 1.0f);
  }
  accum[at] = a; albedo[at] = b; normal[at] = n;
  const f3 c = tonemap_select(mk3(a.x, a.y, a.z) * fc.u.exposure_value, fc.u.enable_tonemap, fc.u.enable_aces, fc.u.use_simple_aces);
  final_img[at] = make_float4(c.x, c.y, c.z, 1.0f);
}

// stand-alone texture fetch (tests): uvl = (u, v, lod) per sample
__global__ void __launch_bounds__(256) k_sample_texture(SceneView sv, uint32_t tex, const float* __restrict__ uvl, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = tex_sample(sv, sv.tex_lut, tex, uvl[3 * i], uvl[3 * i + 1], uvl[3 * i + 2]);
}

// tile-major gathered buffer [world][tiles_per_rank][ts][ts] -> row-major full image (RENDER_SPEC §9)
__global__ void __launch_bounds__(256) k_scatter_tiles(FrameConst fc, const float4* __restrict__ gathered, float4* __restrict__ full) {
  const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_rank = fc.tiles_per_rank * fc.tile_size * fc.tile_size;
  if (gi >= per_rank * fc.world) return;
  FrameConst f2 = fc;
  f2.rank = gi / per_rank;
  uint32_t px, py;
  if (!slot_to_pixel(f2, gi - f2.rank * per_rank, &px, &py)) return;
  full[(size_t)py * fc.width + px] = gathered[gi];
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n, uint32_t per) { return (n + per - 1) / per; }

// tree: 0 large one-level, 1 LDS-staged, 2 two-level (kernels.h: TreeKind)
size_t traverse_fixed_lds_bytes(int tree) {  // per-lane stacks (+ the waves' leaf work lists and merge slots of the large-scene variants)
  return tree == 1 ? (size_t)kStackLdsStaged * kTraverseThreads * 8
                   : (size_t)(tree == 2 ? kStackLdsInst : kStackLdsGlobal) * kTraverseThreads * 8 + (kTraverseThreads / 64) * kCoopBytesPerWave;
}
uint32_t traverse_stack_lds_levels(int tree) { return tree == 1 ? kStackLdsStaged : (tree == 2 ? kStackLdsInst : kStackLdsGlobal); }
uint32_t traverse_stack_spill_levels() { return kStackSpill; }
uint32_t traverse_max_leaf(bool staged) { return staged ? 8u : (uint32_t)kLeafSlots; }
uint32_t traverse_blocks_per_cu(size_t dynamic_lds_bytes, int tree) {
  int a = 0, b = 0;
  const hipError_t ea = tree == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_batch<false, false, true, false, false>, kTraverseThreads, dynamic_lds_bytes)
                      : tree == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_batch<false, false, false, false, true>, kTraverseThreads, dynamic_lds_bytes)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_batch<false, false, false, false, false>, kTraverseThreads, dynamic_lds_bytes);
  const hipError_t eb = tree == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<false, true, false, false>, kTraverseThreads, dynamic_lds_bytes)
                      : tree == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<false, false, false, true>, kTraverseThreads, dynamic_lds_bytes)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<false, false, false, false>, kTraverseThreads, dynamic_lds_bytes);
  if (ea != hipSuccess || eb != hipSuccess) return 0;
  return (uint32_t)std::max(0, std::min(a, b));
}
static size_t traverse_smem(const SceneView& sv) {
  return traverse_fixed_lds_bytes(sv.staged ? 1 : (sv.two_level ? 2 : 0)) + (sv.staged ? (size_t)sv.lds_nodes * 64 + (size_t)sv.lds_tris * 48 : 0);
}

// the traversal kernels are compiled per (any-hit, counting, BVH staged in LDS); all three are launch-time constants
template <bool ANY, bool COUNT, bool STAGED, bool ALPHA, bool INST>
static void launch_trace_batch_t(const LaunchCfg& lc, const SceneView& sv, const hala_ray* rays, hala_hit* hits, const uint32_t* n_ptr,
                                 uint32_t n_imm, WorkCounters* work, Control* ctl, int acc, size_t smem, hipStream_t s) {
  hipLaunchKernelGGL((k_trace_batch<ANY, COUNT, STAGED, ALPHA, INST>), dim3(lc.persistent_blocks), dim3(kTraverseThreads), smem, s, sv, rays, hits, n_ptr, n_imm,
                     work, lc.spill, ctl, acc, lc.refill);
}
// the traversal kernels are compiled per (any-hit, counting, BVH staged in LDS | tree with instance levels) and, for the any-hit ones, per
// "the scene has translucent materials" (RENDER_SPEC 7.1d: the ALPHA variants carry the texture fetch that decides whether a flagged
// triangle blocks); all launch-time constants
void launch_trace_batch(const LaunchCfg& lc, const SceneView& sv, const hala_ray* rays, hala_hit* hits, const uint32_t* n_ptr,
                        uint32_t n_imm, WorkCounters* work, Control* ctl, bool any, bool count, bool account, hipStream_t s) {
  const int acc = account ? 1 : 0;
  using Fn = void (*)(const LaunchCfg&, const SceneView&, const hala_ray*, hala_hit*, const uint32_t*, uint32_t, WorkCounters*, Control*, int, size_t,
                      hipStream_t);
  // index: tree (0 large one-level, 1 LDS-staged, 2 two-level) + 3 * (count + 2 * ray kind (0 closest, 1 any, 2 any with ALPHA))
  static const Fn table[18] = {
      launch_trace_batch_t<false, false, false, false, false>, launch_trace_batch_t<false, false, true, false, false>, launch_trace_batch_t<false, false, false, false, true>,
      launch_trace_batch_t<false, true, false, false, false>,  launch_trace_batch_t<false, true, true, false, false>,  launch_trace_batch_t<false, true, false, false, true>,
      launch_trace_batch_t<true, false, false, false, false>,  launch_trace_batch_t<true, false, true, false, false>,  launch_trace_batch_t<true, false, false, false, true>,
      launch_trace_batch_t<true, true, false, false, false>,   launch_trace_batch_t<true, true, true, false, false>,   launch_trace_batch_t<true, true, false, false, true>,
      launch_trace_batch_t<true, false, false, true, false>,   launch_trace_batch_t<true, false, true, true, false>,   launch_trace_batch_t<true, false, false, true, true>,
      launch_trace_batch_t<true, true, false, true, false>,    launch_trace_batch_t<true, true, true, true, false>,    launch_trace_batch_t<true, true, false, true, true>};
  SceneView sva = sv;
  if (any) sva.tris = sv.tris_any;  // RENDER_SPEC 7.1d
  const int kind = any ? (sv.any_translucent ? 2 : 1) : 0;
  const int tree = sv.staged ? 1 : (sv.two_level ? 2 : 0);
  table[tree + 3 * ((count ? 1 : 0) + 2 * kind)](lc, sva, rays, hits, n_ptr, n_imm, work, ctl, acc, traverse_smem(sv), s);
}

template <bool COUNT, bool STAGED, bool ALPHA, bool INST>
static void launch_trace_shadow_t(const LaunchCfg& lc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth, uint32_t kind,
                                  size_t smem, hipStream_t s) {
  hipLaunchKernelGGL((k_trace_shadow<COUNT, STAGED, ALPHA, INST>), dim3(lc.persistent_blocks), dim3(kTraverseThreads), smem, s, sv, q, ps, ctl, depth, kind, lc.spill, lc.refill);
}
void launch_trace_shadow(const LaunchCfg& lc, const SceneView& sv0, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth,
                         uint32_t kind, bool count, hipStream_t s) {
  SceneView sv = sv0;
  sv.tris = sv0.tris_any;  // RENDER_SPEC 7.1d: shadow rays traverse the copy in which invisible surfaces are degenerate and translucent ones flagged
  using Fn = void (*)(const LaunchCfg&, const SceneView&, const Queues&, const PathState&, Control*, uint32_t, uint32_t, size_t, hipStream_t);
  // index: tree (0 large one-level, 1 LDS-staged, 2 two-level) + 3 * (count + 2 * alpha)
  static const Fn table[12] = {launch_trace_shadow_t<false, false, false, false>, launch_trace_shadow_t<false, true, false, false>, launch_trace_shadow_t<false, false, false, true>,
                               launch_trace_shadow_t<true, false, false, false>,  launch_trace_shadow_t<true, true, false, false>,  launch_trace_shadow_t<true, false, false, true>,
                               launch_trace_shadow_t<false, false, true, false>,  launch_trace_shadow_t<false, true, true, false>,  launch_trace_shadow_t<false, false, true, true>,
                               launch_trace_shadow_t<true, false, true, false>,   launch_trace_shadow_t<true, true, true, false>,   launch_trace_shadow_t<true, false, true, true>};
  const int tree = sv.staged ? 1 : (sv.two_level ? 2 : 0);
  table[tree + 3 * ((count ? 1 : 0) + 2 * (sv.any_translucent ? 1 : 0))](lc, sv, q, ps, ctl, depth, kind, traverse_smem(sv0), s);
}

// the fused launch; false: the caller must issue the two launches separately (an LDS-staged scene whose any-hit rays traverse a
// different triangle copy than its closest-hit rays: only one of them is staged)
bool launch_trace_shadow_then_batch(const LaunchCfg& lc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth,
                                    uint32_t kinds, bool with_closest, hipStream_t s) {
  if (sv.staged && sv.tris_any != sv.tris) return false;
  const size_t smem = traverse_smem(sv);
  const dim3 grid(lc.persistent_blocks), block(kTraverseThreads);
  const hala_ray* rays = with_closest ? q.rays[(depth + 1u) & 1u] : nullptr;
  if (sv.staged) {
    if (sv.any_translucent) hipLaunchKernelGGL((k_trace_shadow_then_batch<true, true, false>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
    else hipLaunchKernelGGL((k_trace_shadow_then_batch<true, false, false>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
  } else if (sv.two_level) {
    if (sv.any_translucent) hipLaunchKernelGGL((k_trace_shadow_then_batch<false, true, true>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
    else hipLaunchKernelGGL((k_trace_shadow_then_batch<false, false, true>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
  } else {
    if (sv.any_translucent) hipLaunchKernelGGL((k_trace_shadow_then_batch<false, true, false>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
    else hipLaunchKernelGGL((k_trace_shadow_then_batch<false, false, false>), grid, block, smem, s, sv, sv.tris_any, q, ps, ctl, depth, kinds, rays, q.hits, lc.spill, lc.refill);
  }
  return true;
}

void launch_trace_primary(const LaunchCfg& lc, const SceneView& sv, const FrameConst& fc, hala_hit* hits, WorkCounters* work, Control* ctl,
                          uint32_t n_account, bool count, hipStream_t s) {
  const size_t smem = traverse_smem(sv);
  static const uint32_t refill_env = tune_env("HALART_REFILL_PRIMARY") ? (uint32_t)atoi(tune_env("HALART_REFILL_PRIMARY")) : 0u;
  // camera rays of neighbouring pixels are about equally long: larger refills (40 idle lanes instead of 24) keep the 8 x 8 pixel blocks together
  const uint32_t refill = refill_env ? refill_env : (sv.staged ? lc.refill : std::max(lc.refill, 40u));
  dim3 grid(lc.persistent_blocks), block(kTraverseThreads);
  if (sv.staged) {
    if (count) hipLaunchKernelGGL((k_trace_primary<true, true, false>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
    else hipLaunchKernelGGL((k_trace_primary<false, true, false>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
  } else if (sv.two_level) {
    if (count) hipLaunchKernelGGL((k_trace_primary<true, false, true>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
    else hipLaunchKernelGGL((k_trace_primary<false, false, true>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
  } else {
    if (count) hipLaunchKernelGGL((k_trace_primary<true, false, false>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
    else hipLaunchKernelGGL((k_trace_primary<false, false, false>), grid, block, smem, s, sv, fc, hits, work, lc.spill, ctl, n_account, refill);
  }
}
void launch_shade(const FrameConst& fc, const SceneView& sv, const Queues& q, const PathState& ps, Control* ctl, uint32_t depth, hipStream_t s) {
  const uint32_t threads = sv.simple_materials ? (uint32_t)kShadeThreads : (uint32_t)kShadeThreadsGeneric;
  dim3 grid(blocks_for(fc.slot_count, threads)), block(threads);
#ifndef RT_SHADE_NOSORT
  // bounce launches of multi-kind scenes shade their paths in kind order inside windows of the queue
  if (depth != 0u && !sv.simple_materials && sv.shade_sort) {
    hipLaunchKernelGGL(k_shade_sort, dim3(blocks_for(fc.slot_count, kSortWindow)), dim3(kSortThreads), 0, s, q, ctl, depth);
    const uint32_t group = 8u * (kSortWindow / threads);  // whole groups of 8 windows: k_shade's window -> XCD mapping permutes the blocks of a group
    grid.x = blocks_for(grid.x, group) * group;
  }
#endif
  if (sv.simple_materials) {
    if (depth == 0u) hipLaunchKernelGGL((k_shade<true, true, false>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
    else hipLaunchKernelGGL((k_shade<false, true, false>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
  } else if (sv.scatter_media) {
    if (depth == 0u) hipLaunchKernelGGL((k_shade<true, false, true>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
    else hipLaunchKernelGGL((k_shade<false, false, true>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
  } else {
    if (depth == 0u) hipLaunchKernelGGL((k_shade<true, false, false>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
    else hipLaunchKernelGGL((k_shade<false, false, false>), grid, block, 0, s, fc, sv, q, ps, ctl, depth);
  }
}
void launch_resolve(const FrameConst& fc, const PathState& ps, float4* accum, float4* albedo, float4* normal, float4* final_img, hipStream_t s) {
  hipLaunchKernelGGL(k_resolve, dim3(blocks_for(fc.pixel_slots, 256)), dim3(256), 0, s, fc, ps, accum, albedo, normal, final_img);
}
void launch_sample_texture(const SceneView& sv, uint32_t tex, const float* uvl, uint32_t n, float4* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sample_texture, dim3(blocks_for(n, 256)), dim3(256), 0, s, sv, tex, uvl, n, out);
}
void launch_scatter_tiles(const FrameConst& fc, const float4* gathered, float4* full, hipStream_t s) {
  const uint32_t n = fc.tiles_per_rank * fc.tile_size * fc.tile_size * fc.world;
  hipLaunchKernelGGL(k_scatter_tiles, dim3(blocks_for(n, 256)), dim3(256), 0, s, fc, gathered, full);
}

}  // namespace rt
