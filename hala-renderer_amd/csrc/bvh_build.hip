// bvh_build.hip — K1/K3/K4 of SURVEY.md §2.1: what the reference hands to the Vulkan driver through
// HalaAccelerationStructure::new (src/scene/loader/gpu_uploader.rs:784-811 BLAS per primitive, :937-959 TLAS),
// done here on the GPU for gfx950:
//   flatten   every instance (node x primitive, in the reference's instance order gpu_uploader.rs:843-875) is
//             transformed to world space once -> one triangle soup, one BVH (288 GB of HBM make the copy free and
//             single-level traversal needs no per-instance ray transform)
//   build     binary hierarchy: >= 4096 triangles: top-down full-sweep SAH, one tree level per round (three centroid orders kept through
//             every partition, segmented box scans with rocPRIM; sah_hierarchy) | below: 60-bit Morton codes -> rocPRIM radix sort ->
//             Karras 2012 LBVH | HALART_BUILDER=ploc: PLOC over the Morton order, the fast large-scene build ->
//             fitted AABBs -> subtrees of <= leaf_max triangles become leaves ->
//             top-down collapse into 4-wide nodes, breadth-first, by surface area -> 64-B compressed nodes
//             (8-bit child boxes quantised conservatively against the node's own box, RENDER_SPEC §4.1b)
//   refit     (north_star; the reference only rebuilds) re-flatten + bottom-up fit + re-pack on the frozen topology
// Results are validated against the oracle through traversal results and a structural check, never topology.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <vector>

#include "kernels.h"
#include "rt_math.h"

namespace rt {

namespace {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(_e);            \
  } while (0)

struct Box6 {
  float mn[3], mx[3];
};

RT_DI uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// RENDER_SPEC §3: p' = fma(m8,z, fma(m4,y, m0*x)) + m12
RT_DI f3 transform_point(const float* m, f3 p) {
  return mk3(__fmaf_rn(m[8], p.z, __fmaf_rn(m[4], p.y, m[0] * p.x)) + m[12], __fmaf_rn(m[9], p.z, __fmaf_rn(m[5], p.y, m[1] * p.x)) + m[13],
             __fmaf_rn(m[10], p.z, __fmaf_rn(m[6], p.y, m[2] * p.x)) + m[14]);
}

// one thread per global triangle id.  The three records a triangle gets (48-B Tri, 128-B ShadeTri, 24-B Box6) are staged in LDS and
// leave the block as contiguous 16-B (8-B for the boxes) stores.  Scene bounds: NO global atomics here — a device-scope atomic
// costs 11.4 ns per 128-B line however many waves issue it (scripts/microbench/atomic_rate.hip), and six of them per wave on one
// line held this kernel at 1.07 ms per million triangles; each block leaves its bounds in block_ord and k_bounds_reduce folds them.
__global__ void __launch_bounds__(256) k_flatten(const hala_gpu_mesh_data* __restrict__ prims, const uint32_t* __restrict__ first_tri,
                                                  uint32_t inst_count, uint32_t n, Tri* __restrict__ tris_by_id, ShadeTri* __restrict__ shade_tris,
                                                  Box6* __restrict__ tri_box,
                                                  uint32_t* __restrict__ block_ord /* [gridDim.x][6] min xyz, max xyz */,
                                                  const uint32_t* __restrict__ gid_first, const uint32_t* __restrict__ inst_index, int object_space) {
  __shared__ float4 stage[256 * 8];
  __shared__ uint32_t wave_ord[4][6];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * 256u, g = base + tid;
  const uint32_t cnt = min(256u, n - base);
  uint32_t ord[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  float4 tri[3];
  Box6 b;
  if (g < n) {
    uint32_t lo = 0, hi = inst_count;  // last instance with first_tri <= g
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (first_tri[mid] <= g) lo = mid; else hi = mid;
    }
    const hala_gpu_mesh_data& md = prims[lo];
    const uint32_t lt = g - first_tri[lo];
    const uint32_t* idx = reinterpret_cast<const uint32_t*>(md.indices) + 3 * (size_t)lt;
    const hala_vertex* vb = reinterpret_cast<const hala_vertex*>(md.vertices);
    f3 v[3];
    ShadeTri st{};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const hala_vertex& vx = vb[idx[c]];
      v[c] = object_space ? ld3(vx.position) : transform_point(md.transform, ld3(vx.position));  // RENDER_SPEC 4.5: an instanced primitive keeps its local positions
      float* nc = c == 0 ? st.n0 : (c == 1 ? st.n1 : st.n2);
#pragma unroll
      for (int k = 0; k < 3; ++k) { nc[k] = vx.normal[k]; st.tg[c][k] = vx.tangent[k]; }
      st.uv[c][0] = vx.tex_coord[0]; st.uv[c][1] = vx.tex_coord[1];
    }
    const uint32_t inst = inst_index ? inst_index[lo] : lo;
    st.inst = inst; st.material = md.material_index;
    const f3 e1 = v[1] - v[0], e2 = v[2] - v[0];
    const f3 gc = cross3(e1, e2);  // RENDER_SPEC 6: the geometric normal is normalize(cross(e1, e2)) of the stored world-space edges
    st.gcross[0] = gc.x; st.gcross[1] = gc.y; st.gcross[2] = gc.z;
    tri[0] = make_float4(v[0].x, v[0].y, v[0].z, __uint_as_float(gid_first ? gid_first[lo] + lt : g));
    tri[1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
    tri[2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
    b.mn[0] = fminf(v[0].x, fminf(v[1].x, v[2].x)); b.mx[0] = fmaxf(v[0].x, fmaxf(v[1].x, v[2].x));
    b.mn[1] = fminf(v[0].y, fminf(v[1].y, v[2].y)); b.mx[1] = fmaxf(v[0].y, fmaxf(v[1].y, v[2].y));
    b.mn[2] = fminf(v[0].z, fminf(v[1].z, v[2].z)); b.mx[2] = fmaxf(v[0].z, fmaxf(v[1].z, v[2].z));
#pragma unroll
    for (int k = 0; k < 3; ++k) { ord[k] = f2ord(b.mn[k]); ord[3 + k] = f2ord(b.mx[k]); }
    const float4* sp = reinterpret_cast<const float4*>(&st);
#pragma unroll
    for (int k = 0; k < 8; ++k) stage[tid * 8u + k] = sp[k];
  }
  __syncthreads();
  {
    float4* out = reinterpret_cast<float4*>(shade_tris + base);
    for (uint32_t i = tid; i < cnt * 8u; i += 256u) out[i] = stage[i];
  }
  __syncthreads();
  if (g < n) { stage[tid * 3u] = tri[0]; stage[tid * 3u + 1u] = tri[1]; stage[tid * 3u + 2u] = tri[2]; }
  __syncthreads();
  {
    float4* out = reinterpret_cast<float4*>(tris_by_id + base);
    for (uint32_t i = tid; i < cnt * 3u; i += 256u) out[i] = stage[i];
  }
  __syncthreads();
  float2* stage2 = reinterpret_cast<float2*>(stage);
  if (g < n) {
    stage2[tid * 3u] = make_float2(b.mn[0], b.mn[1]); stage2[tid * 3u + 1u] = make_float2(b.mn[2], b.mx[0]);
    stage2[tid * 3u + 2u] = make_float2(b.mx[1], b.mx[2]);
  }
  __syncthreads();
  {
    float2* out = reinterpret_cast<float2*>(tri_box + base);
    for (uint32_t i = tid; i < cnt * 3u; i += 256u) out[i] = stage2[i];
  }
  // scene bounds of this block: wave shuffle reduction, then the four waves through LDS
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      ord[k] = min(ord[k], (uint32_t)__shfl_xor((int)ord[k], m, 64));
      ord[3 + k] = max(ord[3 + k], (uint32_t)__shfl_xor((int)ord[3 + k], m, 64));
    }
  }
  if ((tid & 63u) == 0u) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wave_ord[tid >> 6][k] = ord[k];
  }
  __syncthreads();
  if (tid < 6u) {
    uint32_t v = wave_ord[0][tid];
    for (int w = 1; w < 4; ++w) v = tid < 3u ? min(v, wave_ord[w][tid]) : max(v, wave_ord[w][tid]);
    block_ord[blockIdx.x * 6u + tid] = v;
  }
}
// one block: folds the per-block bounds of k_flatten into scene_ord[6]
__global__ void __launch_bounds__(256) k_bounds_reduce(const uint32_t* __restrict__ block_ord, uint32_t blocks, uint32_t* __restrict__ scene_ord) {
  __shared__ uint32_t wave_ord[4][6];
  uint32_t ord[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  for (uint32_t b = threadIdx.x; b < blocks; b += 256u) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { ord[k] = min(ord[k], block_ord[b * 6u + k]); ord[3 + k] = max(ord[3 + k], block_ord[b * 6u + 3 + k]); }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      ord[k] = min(ord[k], (uint32_t)__shfl_xor((int)ord[k], m, 64));
      ord[3 + k] = max(ord[3 + k], (uint32_t)__shfl_xor((int)ord[3 + k], m, 64));
    }
  }
  if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wave_ord[threadIdx.x >> 6][k] = ord[k];
  }
  __syncthreads();
  if (threadIdx.x < 6u) {
    uint32_t v = wave_ord[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v = threadIdx.x < 3u ? min(v, wave_ord[w][threadIdx.x]) : max(v, wave_ord[w][threadIdx.x]);
    scene_ord[threadIdx.x] = v;
  }
}

RT_DI unsigned long long spread21(uint32_t v) {  // 21 bits -> every third bit
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ void __launch_bounds__(256) k_morton(const Box6* __restrict__ tri_box, uint32_t n, const uint32_t* __restrict__ scene_ord,
                                                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids, uint32_t size_classes) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const Box6 b = tri_box[g];
  uint32_t q[3];
  float scene_d2 = 0.0f, tri_d2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo = ord2f(scene_ord[k]), hi = ord2f(scene_ord[3 + k]);
    const float c = 0.5f * (b.mn[k] + b.mx[k]);
    const float ext = hi - lo;
    float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    q[k] = min((uint32_t)(t * 1048576.0f), 1048575u);  // 20 bits per axis
    scene_d2 += ext * ext;
    tri_d2 += (b.mx[k] - b.mn[k]) * (b.mx[k] - b.mn[k]);
  }
  // Size class in the two top key bits: LBVH sorts by centroid only, so one huge triangle (a ground quad, a wall) would
  // inflate the boxes of every ancestor it shares with small neighbours.  With the class on top of the key the Karras
  // hierarchy splits by class first: big triangles form their own shallow subtrees next to the well-formed rest.
  // class 0: box diagonal > 1/8 of the scene's, 1: > 1/32, 2: > 1/128, 3: the rest.
  const float ratio2 = scene_d2 > 0.0f ? tri_d2 / scene_d2 : 0.0f;
  uint32_t cls = ratio2 > (1.0f / 64.0f) ? 0u : (ratio2 > (1.0f / 1024.0f) ? 1u : (ratio2 > (1.0f / 16384.0f) ? 2u : 3u));
  if (!size_classes) cls = 0u;
  keys[g] = ((unsigned long long)cls << 60) | (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  ids[g] = g;
}

// ---- Karras 2012 ------------------------------------------------------------------------------------------
RT_DI int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll((long long)(a ^ b));
}
// child reference in the build tree: bit 31 set = leaf (sorted position), else internal node index
constexpr uint32_t kLeafBit = 0x80000000u;

__global__ void __launch_bounds__(256) k_hierarchy(const unsigned long long* __restrict__ keys, int n, uint32_t* __restrict__ left,
                                                    uint32_t* __restrict__ right, uint32_t* __restrict__ first, uint32_t* __restrict__ last,
                                                    uint32_t* __restrict__ node_parent, uint32_t* __restrict__ leaf_parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const uint32_t lc = (lo == gamma) ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
  const uint32_t rc = (hi == gamma + 1) ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  left[i] = lc; right[i] = rc; first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
  if (lc & kLeafBit) leaf_parent[gamma] = (uint32_t)i; else node_parent[gamma] = (uint32_t)i;
  if (rc & kLeafBit) leaf_parent[gamma + 1] = (uint32_t)i; else node_parent[gamma + 1] = (uint32_t)i;
  if (i == 0) node_parent[0] = kAbsent;
}

__global__ void __launch_bounds__(256) k_leaf_boxes(const Box6* __restrict__ tri_box, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                     Box6* __restrict__ leaf_box, const Tri* __restrict__ tris_by_id, Tri* __restrict__ tris, float pad,
                                                     Tri* __restrict__ tris_any, const ShadeTri* __restrict__ shade_tris,
                                                     const uint8_t* __restrict__ any_class, uint32_t material_count,
                                                     const uint8_t* __restrict__ material_kind) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t id = sorted_ids[k];
  const uint32_t mat = shade_tris[id].material;
  // word 11: the shading kind of the triangle's material; the closest-hit traversal puts it into the hit record (hala_types.h: hit_encode)
  const float kind_word = __uint_as_float(material_kind && mat < material_count ? (uint32_t)material_kind[mat] : kShadeKindSpecial);
  // The slab test's plane distances carry a rounding error of a few ulp of the largest coordinate involved, like the triangle
  // test's own; quantisation usually adds far more slack, but a plane that falls exactly on the grid gets none.  `pad` (2^-19 of
  // the scene's largest coordinate) keeps a hit that lies exactly on a box face inside its box.
  Box6 b = tri_box[id];
#pragma unroll
  for (int c = 0; c < 3; ++c) { b.mn[c] -= pad; b.mx[c] += pad; }
  leaf_box[k] = b;
  const float4* src = reinterpret_cast<const float4*>(tris_by_id + id);
  float4* dst = reinterpret_cast<float4*>(tris + k);
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = make_float4(src[2].x, src[2].y, src[2].z, kind_word);
  if (tris_any) {  // RENDER_SPEC 7.1d: any-hit rays do not see surfaces of opacity exactly 0 — their triangles are degenerate in this copy —
                   // and decide per (ray key, triangle) whether a translucent one blocks them: those carry a flag in word 7
    const uint32_t cls = mat < material_count ? any_class[mat] : 0u;
    const bool gone = cls == 1u;
    float4* da = reinterpret_cast<float4*>(tris_any + k);
    da[0] = src[0];
    const uint32_t flag = cls == 2u ? 1u : (cls == 3u ? 2u : (cls == 4u ? 3u : 0u));  // bit 0: translucent, bit 1: boundary of a medium (7.1g)
    da[1] = gone ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(src[1].x, src[1].y, src[1].z, __uint_as_float(flag));
    da[2] = gone ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(src[2].x, src[2].y, src[2].z, kind_word);
  }
}

RT_DI Box6 box_union(const Box6& a, const Box6& b) {
  Box6 r;
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
  return r;
}
// bottom-up fit: the second thread to arrive at a node owns it.  Producer: stores -> __threadfence (agent release)
// -> atomic arrival; consumer: atomic arrival -> __threadfence (agent acquire) -> loads.
__global__ void __launch_bounds__(256) k_fit(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                              const uint32_t* __restrict__ node_parent, const uint32_t* __restrict__ leaf_parent,
                                              const Box6* __restrict__ leaf_box, Box6* node_box, uint32_t* arrivals, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t p = leaf_parent[k];
  while (p != kAbsent) {
    __threadfence();
    const uint32_t old = atomicAdd(&arrivals[p], 1u);
    if (old == 0u) return;
    __threadfence();
    const uint32_t lc = left[p], rc = right[p];
    const volatile Box6* lb = (lc & kLeafBit) ? &leaf_box[lc & ~kLeafBit] : &node_box[lc];
    const volatile Box6* rb = (rc & kLeafBit) ? &leaf_box[rc & ~kLeafBit] : &node_box[rc];
    Box6 a, b;
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.mn[c] = lb->mn[c]; a.mx[c] = lb->mx[c]; b.mn[c] = rb->mn[c]; b.mx[c] = rb->mx[c]; }
    const Box6 u = box_union(a, b);
    volatile Box6* o = &node_box[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o->mn[c] = u.mn[c]; o->mx[c] = u.mx[c]; }
    p = node_parent[p];
  }
}

__global__ void __launch_bounds__(256) k_keep_flags(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t n_internal,
                                                     uint32_t leaf_max, uint32_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_internal) return;
  keep[i] = (last[i] - first[i] + 1u) > leaf_max ? 1u : 0u;
}

// ---- 4-wide compressed emission (RENDER_SPEC §4.1b) ---------------------------------------------------------------
struct Child4 { Box6 box; uint32_t ref; };

RT_DI uint32_t leaf_ref(uint32_t first, uint32_t count) { return kLeafRef | ((count - 1u) << 28) | first; }

// quantum exponent of one axis: the smallest power of two s = 2^e with extent / s <= 255 (never below 2^-100)
RT_DI int quantum_exponent(float lo, float hi) {
  const double q = ((double)hi - (double)lo) / 255.0;
  if (!(q > 0.0)) return -100;
  const int e = (int)((__double_as_longlong(q) >> 52) & 0x7ff) - 1022;  // q = m * 2^e, m in [0.5, 1): 2^e > q or m == 0.5
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

RT_DI BvhNode4 pack_node4(const Child4* ch, int n) {
  BvhNode4 nd{};
  Box6 all = ch[0].box;
  for (int c = 1; c < n; ++c) all = box_union(all, ch[c].box);
  int e[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    nd.pmin[a] = all.mn[a];
    e[a] = quantum_exponent(all.mn[a], all.mx[a]);
    nd.exps |= (uint32_t)(e[a] + 127) << (8 * a);
  }
  for (int c = 0; c < 4; ++c) {
    if (c >= n) { nd.ref[c] = kAbsent; continue; }
    nd.ref[c] = ch[c].ref;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double s = __longlong_as_double((long long)(e[a] + 1023) << 52), base = (double)all.mn[a];
      double lo = floor(((double)ch[c].box.mn[a] - base) / s), hi = ceil(((double)ch[c].box.mx[a] - base) / s);
      if (base + lo * s > (double)ch[c].box.mn[a]) lo -= 1.0;  // rounding of the subtraction must not shrink the box
      if (base + hi * s < (double)ch[c].box.mx[a]) hi += 1.0;
      lo = fmin(fmax(lo, 0.0), 255.0); hi = fmin(fmax(hi, 0.0), 255.0);
      nd.qlo[a] |= (uint32_t)lo << (8 * c);
      nd.qhi[a] |= (uint32_t)hi << (8 * c);
    }
  }
  return nd;
}

// Top-down collapse of the binary tree into 4-wide nodes, one BFS level per launch, so that node indices come out in
// breadth-first order (the first `lds_nodes` nodes — the slice the traversal kernels stage in LDS — are the top of the
// tree, which every ray visits).  The 4-node rooted at binary node i starts with i's two children and opens, while a
// slot is free, (1) the kept inner child of largest surface area, then (2) the collapsed leaf of largest surface area
// (splitting a leaf costs no bytes: the slot exists anyway, and two tighter boxes cull more triangle tests).
// refs4[node] = the binary-tree references its slots were filled from — the frozen topology refit re-packs from.
RT_DI float half_area(const Box6& b) {
  const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
  return dx * dy + dy * dz + dz * dx;
}
// The level kernels take the level's first node and size from device memory (level[0], level[1]; k_level_advance closes a level):
// the host only knows an upper bound of the size — the grid — and looks at the counters every few levels.
__global__ void __launch_bounds__(256) k_collapse_level(const uint32_t* __restrict__ root_of, const uint32_t* __restrict__ level, uint32_t bound,
                                                         const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                         const uint32_t* __restrict__ keep, const Box6* __restrict__ node_box,
                                                         uint4* __restrict__ refs4, uint32_t* __restrict__ cnt,
                                                         const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, int order_mode) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= bound) return;
  const uint32_t base = level[0], size = level[1];
  if (j >= size) { cnt[j] = 0u; return; }  // the scan runs over the host's bound
  const uint32_t i = root_of[base + j];
  uint32_t c[4] = {left[i], right[i], kAbsent, kAbsent};
  int n = 2;
  for (int phase = 0; phase < 2; ++phase) {
    while (n < 4) {
      int pick = -1;
      float best = -1.0f;
      for (int k = 0; k < n; ++k) {
        const uint32_t r = c[k];
        if (r & kLeafBit) continue;
        if ((keep[r] != 0u) != (phase == 0)) continue;
        const float a = half_area(node_box[r]);
        if (a > best) { best = a; pick = k; }
      }
      if (pick < 0) break;
      const uint32_t r = c[pick];
      for (int k = n; k > pick + 1; --k) c[k] = c[k - 1];
      c[pick] = left[r]; c[pick + 1] = right[r];
      ++n;
    }
  }
  // Slot order = the order in which an any-hit ray of a large tree takes the inner children (RENDER_SPEC 4.4c); every other ray sorts
  // the children by distance (the slot only breaks ties)
  if (order_mode != 0) {
    float w[4];
    for (int k = 0; k < n; ++k) {
      const uint32_t r = c[k];
      if ((r & kLeafBit) || !keep[r]) { w[k] = -1.0f; continue; }
      const float a = half_area(node_box[r]), cntf = (float)(last[r] - first[r] + 1u);
      w[k] = order_mode == 1 ? a : (order_mode == 2 ? 1.0f / (a + 1e-30f) : (order_mode == 3 ? cntf : (order_mode == 4 ? cntf / (a + 1e-30f) : (a + 1e-30f) / cntf)));
    }
    for (int a = 1; a < n; ++a)  // insertion sort, descending weight, stable
      for (int b = a; b > 0 && w[b] > w[b - 1]; --b) {
        const float tw = w[b]; w[b] = w[b - 1]; w[b - 1] = tw;
        const uint32_t tc = c[b]; c[b] = c[b - 1]; c[b - 1] = tc;
      }
  }
  refs4[base + j] = make_uint4(c[0], c[1], c[2], c[3]);
  uint32_t inner = 0;
  for (int k = 0; k < n; ++k) inner += (!(c[k] & kLeafBit) && keep[c[k]]) ? 1u : 0u;
  cnt[j] = inner;
}
__global__ void __launch_bounds__(256) k_scatter_level(uint32_t* __restrict__ root_of, const uint32_t* __restrict__ level, const uint4* __restrict__ refs4,
                                                        const uint32_t* __restrict__ keep, const uint32_t* __restrict__ off,
                                                        uint32_t* __restrict__ index4) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t base = level[0], size = level[1];
  if (j >= size) return;
  const uint4 q = refs4[base + j];
  const uint32_t c[4] = {q.x, q.y, q.z, q.w};
  uint32_t pos = base + size + off[j];
  for (int k = 0; k < 4; ++k) {
    const uint32_t r = c[k];
    if ((r & kLeafBit) || !keep[r]) continue;
    root_of[pos] = r;
    index4[r] = pos;
    ++pos;
  }
}
// closes a level: level = {base + size, inner children found, levels so far + 1}; bases[l] = first node of level l
__global__ void k_level_advance(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, uint32_t* __restrict__ level,
                                uint32_t* __restrict__ bases, uint32_t max_levels) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t base = level[0], size = level[1], l = level[2];
  if (size == 0u) return;
  if (l < max_levels) bases[l] = base;
  level[0] = base + size;
  level[1] = off[size - 1u] + cnt[size - 1u];
  level[2] = l + 1u;
}
__global__ void __launch_bounds__(256) k_pack4(const uint4* __restrict__ refs4, uint32_t node_count, const uint32_t* __restrict__ first,
                                                const uint32_t* __restrict__ last, const uint32_t* __restrict__ keep,
                                                const uint32_t* __restrict__ index4, const Box6* __restrict__ leaf_box,
                                                const Box6* __restrict__ node_box, BvhNode4* __restrict__ nodes) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= node_count) return;
  const uint4 q = refs4[idx];
  const uint32_t c[4] = {q.x, q.y, q.z, q.w};
  Child4 ch[4];
  int n = 0;
  for (int k = 0; k < 4; ++k) {
    const uint32_t r = c[k];
    if (r == kAbsent) continue;
    if (r & kLeafBit) { ch[n].box = leaf_box[r & ~kLeafBit]; ch[n].ref = leaf_ref(r & ~kLeafBit, 1u); }
    else if (!keep[r]) { ch[n].box = node_box[r]; ch[n].ref = leaf_ref(first[r], last[r] - first[r] + 1u); }
    else { ch[n].box = node_box[r]; ch[n].ref = index4[r]; }
    ++n;
  }
  nodes[idx] = pack_node4(ch, n);
}

// Refit of one BFS level of the kept 4-wide topology.  node_box is re-used as scratch indexed by 4-NODE index here (the
// binary boxes it held are not needed once the tree is collapsed): a node leaves the union of its children there for its
// parent, one level up, which runs in a later launch.  Same boxes as a fresh fit (min / max are exact), no fences.
__global__ void __launch_bounds__(256) k_refit_level(const uint4* __restrict__ refs4, uint32_t base, uint32_t size, const uint32_t* __restrict__ first,
                                                      const uint32_t* __restrict__ last, const uint32_t* __restrict__ keep,
                                                      const uint32_t* __restrict__ index4, const Box6* __restrict__ leaf_box, Box6* box4,
                                                      BvhNode4* __restrict__ nodes) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= size) return;
  const uint32_t idx = base + k;
  const uint4 q = refs4[idx];
  const uint32_t c[4] = {q.x, q.y, q.z, q.w};
  Child4 ch[4];
  int n = 0;
  for (int s = 0; s < 4; ++s) {
    const uint32_t r = c[s];
    if (r == kAbsent) continue;
    if (r & kLeafBit) { ch[n].box = leaf_box[r & ~kLeafBit]; ch[n].ref = leaf_ref(r & ~kLeafBit, 1u); }
    else if (!keep[r]) {
      const uint32_t f = first[r], l = last[r];
      Box6 u = leaf_box[f];
      for (uint32_t t = f + 1; t <= l; ++t) u = box_union(u, leaf_box[t]);
      ch[n].box = u; ch[n].ref = leaf_ref(f, l - f + 1u);
    } else { ch[n].box = box4[index4[r]]; ch[n].ref = index4[r]; }
    ++n;
  }
  nodes[idx] = pack_node4(ch, n);
  Box6 all = ch[0].box;
  for (int s = 1; s < n; ++s) all = box_union(all, ch[s].box);
  box4[idx] = all;
}

// scene of <= leaf_max triangles: one 4-node with a single leaf child
__global__ void k_emit_single4(const Box6* __restrict__ leaf_box, uint32_t n, BvhNode4* __restrict__ nodes) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Child4 ch[1];
  ch[0].box = Box6{};
  if (n > 0) {
    ch[0].box = leaf_box[0];
    for (uint32_t k = 1; k < n; ++k) ch[0].box = box_union(ch[0].box, leaf_box[k]);
  }
  ch[0].ref = n > 0 ? leaf_ref(0u, n) : kAbsent;
  nodes[0] = pack_node4(ch, 1);
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) ----------------------------------------------
// An alternative to the Karras hierarchy over the same Morton-sorted triangles: clusters (initially the triangles, in
// Morton order) repeatedly merge with their nearest neighbour — nearest = smallest surface area of the union, searched
// kPlocRadius positions to either side — whenever the choice is mutual.  Same arrays out as k_hierarchy (left, right,
// parents, per-node triangle ranges), so fit / collapse / pack / refit are shared; because merged clusters need not be
// adjacent, the triangles are re-ordered once at the end so that every subtree is a contiguous range again.
constexpr int kPlocRadius = 16;
RT_DI uint32_t ploc_size(uint32_t ref, const uint32_t* __restrict__ subtree) { return (ref & kLeafBit) ? 1u : subtree[ref]; }

__global__ void __launch_bounds__(256) k_ploc_init(const Box6* __restrict__ tri_box, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                    uint32_t* __restrict__ cl_ref, Box6* __restrict__ cl_box) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  cl_ref[k] = kLeafBit | k;
  cl_box[k] = tri_box[sorted_ids[k]];
}
// The round kernels take the live cluster count m (and the nodes created so far) from device memory: the host only knows an upper
// bound of m (the grid), refreshed every few rounds, so that a round does not cost a host round trip (k_ploc_advance).
__global__ void __launch_bounds__(256) k_ploc_nearest(const Box6* __restrict__ cl_box, const uint32_t* __restrict__ state, uint32_t radius,
                                                       uint32_t* __restrict__ nn) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = state[0];
  if (i >= m) return;
  const Box6 bi = cl_box[i];
  const uint32_t lo = i > radius ? i - radius : 0u, hi = min(m - 1u, i + radius);
  float best = 3.402823466e+38f;
  uint32_t bj = i;
  for (uint32_t j = lo; j <= hi; ++j) {  // ascending j + strict '<': ties go to the lower index (guarantees a mutual pair)
    if (j == i) continue;
    const float a = half_area(box_union(bi, cl_box[j]));
    if (a < best || bj == i) { best = a; bj = j; }
  }
  nn[i] = bj;
}
__global__ void __launch_bounds__(256) k_ploc_flags(const uint32_t* __restrict__ nn, const uint32_t* __restrict__ state, uint32_t bound,
                                                     uint32_t* __restrict__ merge, uint32_t* __restrict__ keepc) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const uint32_t m = state[0];
  if (i >= m) { merge[i] = 0u; keepc[i] = 0u; return; }  // the scans run over the host's bound
  const uint32_t j = nn[i];
  const bool mutual = j != i && nn[j] == i;
  merge[i] = (mutual && i < j) ? 1u : 0u;
  keepc[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ void __launch_bounds__(256) k_ploc_apply(const uint32_t* __restrict__ nn, const uint32_t* __restrict__ merge,
                                                     const uint32_t* __restrict__ keepc, const uint32_t* __restrict__ merge_scan,
                                                     const uint32_t* __restrict__ keep_scan, const uint32_t* __restrict__ state, uint32_t last_id,
                                                     const uint32_t* __restrict__ cl_ref, const Box6* __restrict__ cl_box,
                                                     uint32_t* __restrict__ cl_ref_out, Box6* __restrict__ cl_box_out,
                                                     uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ node_parent,
                                                     uint32_t* __restrict__ leaf_parent, uint32_t* __restrict__ subtree,
                                                     Box6* __restrict__ node_box, float pad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = state[0], next_id = last_id - state[1];  // last_id = n - 2: ids count down from it as nodes are created
  if (i >= m || !keepc[i]) return;
  const uint32_t pos = keep_scan[i];
  if (!merge[i]) { cl_ref_out[pos] = cl_ref[i]; cl_box_out[pos] = cl_box[i]; return; }
  const uint32_t j = nn[i], id = next_id - merge_scan[i];  // ids count down: the last merge of all creates node 0, the root
  const uint32_t ra = cl_ref[i], rb = cl_ref[j];
  left[id] = ra; right[id] = rb;
  if (ra & kLeafBit) leaf_parent[ra & ~kLeafBit] = id; else node_parent[ra] = id;
  if (rb & kLeafBit) leaf_parent[rb & ~kLeafBit] = id; else node_parent[rb] = id;
  subtree[id] = ploc_size(ra, subtree) + ploc_size(rb, subtree);
  cl_ref_out[pos] = id;
  const Box6 u = box_union(cl_box[i], cl_box[j]);
  cl_box_out[pos] = u;
  // the fitted box of node id, without a bottom-up pass: the leaf boxes are the triangle boxes widened by `pad` (k_leaf_boxes),
  // and x -> fl(x -/+ pad) is monotone, so widening the union gives bit for bit the union of the widened boxes
  Box6 w;
#pragma unroll
  for (int c = 0; c < 3; ++c) { w.mn[c] = u.mn[c] - pad; w.mx[c] = u.mx[c] + pad; }
  node_box[id] = w;
}
// closes a round: state = {m - merged, created + merged}
__global__ void k_ploc_advance(const uint32_t* __restrict__ merge, const uint32_t* __restrict__ merge_scan, uint32_t* __restrict__ state) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t m = state[0];
  if (m == 0u) return;
  const uint32_t merged = merge_scan[m - 1u] + merge[m - 1u];
  state[0] = m - merged;
  state[1] += merged;
}
// The last rounds in ONE workgroup: once kPlocTail or fewer clusters are left (27 of the ~60 rounds of a million-triangle build)
// a round is far shorter than its six launches and two host round trips.  Same nearest-neighbour rule, same flags, same ids
// (next_id - exclusive scan of the merge flags) as k_ploc_nearest / k_ploc_flags / k_ploc_apply: the tree is bit for bit the same.
constexpr uint32_t kPlocTail = 512;
__global__ void __launch_bounds__(kPlocTail) k_ploc_tail(uint32_t m, uint32_t radius, uint32_t next_id, const uint32_t* __restrict__ cl_ref_in,
                                                          const Box6* __restrict__ cl_box_in, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                          uint32_t* __restrict__ node_parent, uint32_t* __restrict__ leaf_parent, uint32_t* subtree,
                                                          Box6* __restrict__ node_box, float pad) {
  __shared__ uint32_t s_ref[2][kPlocTail], s_size[2][kPlocTail];  // s_size: triangles below the cluster (subtree[] of this kernel's own
  __shared__ Box6 s_box[2][kPlocTail];                            // nodes is only written, never read back through the caches)
  __shared__ uint32_t s_nn[kPlocTail];
  __shared__ uint32_t s_wave[kPlocTail / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  int cur = 0;
  if (tid < m) { const uint32_t r = cl_ref_in[tid]; s_ref[0][tid] = r; s_size[0][tid] = ploc_size(r, subtree); s_box[0][tid] = cl_box_in[tid]; }
  __syncthreads();
  while (m > 1u) {
    if (tid < m) {
      const Box6 bi = s_box[cur][tid];
      const uint32_t lo = tid > radius ? tid - radius : 0u, hi = min(m - 1u, tid + radius);
      float best = 3.402823466e+38f;
      uint32_t bj = tid;
      for (uint32_t j = lo; j <= hi; ++j) {
        if (j == tid) continue;
        const float a = half_area(box_union(bi, s_box[cur][j]));
        if (a < best || bj == tid) { best = a; bj = j; }
      }
      s_nn[tid] = bj;
    }
    __syncthreads();
    uint32_t merge = 0, keepc = 0, j = tid;
    if (tid < m) {
      j = s_nn[tid];
      const bool mutual = j != tid && s_nn[j] == tid;
      merge = (mutual && tid < j) ? 1u : 0u;
      keepc = (mutual && tid > j) ? 0u : 1u;
    }
    // one block-wide scan for both flags: merge count in the high half, keep count in the low half (<= 512 each)
    const uint32_t packed = (merge << 16) | keepc;
    uint32_t incl = packed;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kPlocTail / 64; ++w) { const uint32_t v = s_wave[w]; if (w < wave) before += v; total += v; }
    const uint32_t excl = before + incl - packed;
    const uint32_t merge_scan = excl >> 16, keep_scan = excl & 0xffffu;
    if (tid < m && keepc) {
      if (!merge) { s_ref[cur ^ 1][keep_scan] = s_ref[cur][tid]; s_size[cur ^ 1][keep_scan] = s_size[cur][tid]; s_box[cur ^ 1][keep_scan] = s_box[cur][tid]; }
      else {
        const uint32_t id = next_id - merge_scan;
        const uint32_t ra = s_ref[cur][tid], rb = s_ref[cur][j];
        left[id] = ra; right[id] = rb;
        if (ra & kLeafBit) leaf_parent[ra & ~kLeafBit] = id; else node_parent[ra] = id;
        if (rb & kLeafBit) leaf_parent[rb & ~kLeafBit] = id; else node_parent[rb] = id;
        const uint32_t size = s_size[cur][tid] + s_size[cur][j];
        subtree[id] = size;
        s_ref[cur ^ 1][keep_scan] = id; s_size[cur ^ 1][keep_scan] = size;
        const Box6 u = box_union(s_box[cur][tid], s_box[cur][j]);
        s_box[cur ^ 1][keep_scan] = u;
        Box6 w;
#pragma unroll
        for (int c = 0; c < 3; ++c) { w.mn[c] = u.mn[c] - pad; w.mx[c] = u.mx[c] + pad; }
        node_box[id] = w;
      }
    }
    __syncthreads();
    next_id -= total >> 16;
    m = total & 0xffffu;
    cur ^= 1;
  }
}
// position of leaf k / first position of node i in the depth-first order of the finished tree: the sizes of all left
// siblings passed on the way up
RT_DI uint32_t ploc_offset(uint32_t ref, uint32_t p, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                           const uint32_t* __restrict__ node_parent, const uint32_t* __restrict__ subtree) {
  uint32_t pos = 0;
  while (p != kAbsent) {
    if (right[p] == ref) pos += ploc_size(left[p], subtree);
    ref = p;
    p = node_parent[p];
  }
  return pos;
}
__global__ void __launch_bounds__(256) k_ploc_leaf_order(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                          const uint32_t* __restrict__ node_parent, const uint32_t* __restrict__ leaf_parent,
                                                          const uint32_t* __restrict__ subtree, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                          uint32_t* __restrict__ new_pos, uint32_t* __restrict__ sorted_ids_out,
                                                          uint32_t* __restrict__ leaf_parent_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t pos = ploc_offset(kLeafBit | k, leaf_parent[k], left, right, node_parent, subtree);
  new_pos[k] = pos;
  sorted_ids_out[pos] = sorted_ids[k];
  leaf_parent_out[pos] = leaf_parent[k];
}
__global__ void __launch_bounds__(256) k_ploc_node_ranges(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                           const uint32_t* __restrict__ node_parent, const uint32_t* __restrict__ subtree,
                                                           uint32_t n_internal, uint32_t* __restrict__ first, uint32_t* __restrict__ last) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_internal) return;
  const uint32_t off = ploc_offset(i, node_parent[i], left, right, node_parent, subtree);
  first[i] = off;
  last[i] = off + subtree[i] - 1u;
}
__global__ void __launch_bounds__(256) k_ploc_renumber(uint32_t* __restrict__ left, uint32_t* __restrict__ right, const uint32_t* __restrict__ new_pos,
                                                        uint32_t n_internal) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_internal) return;
  const uint32_t l = left[i], r = right[i];
  if (l & kLeafBit) left[i] = kLeafBit | new_pos[l & ~kLeafBit];
  if (r & kLeafBit) right[i] = kLeafBit | new_pos[r & ~kLeafBit];
}


// ---- full-sweep SAH hierarchy (top-down, one tree level per round) --------------------------------------------------------------
// The triangles are sorted ONCE along each axis by the centroid of their box; every open node owns the same range [first, last] of
// positions in all three orders.  A round, for all open nodes at once: per axis a segmented prefix and suffix union of the boxes
// (rocPRIM scan-by-key over the node ids; min / max are exactly associative, so the result does not depend on how the scan is cut up),
// the surface-area cost area(L)*|L| + area(R)*|R| of every split position, the cheapest (cost, axis, position) per node through a
// 64-bit atomic min, then a stable partition of the three orders by the side each triangle went to.  The contract is PLOC's:
// left / right / first / last / parents / fitted node boxes / sorted_ids, node 0 = root, a full binary tree down to single triangles.
struct BoxUnionOp {
  __host__ __device__ Box6 operator()(const Box6& a, const Box6& b) const {
    Box6 r;
    for (int k = 0; k < 3; ++k) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return r;
  }
};
struct Flag3 { uint32_t v[3]; };
struct Flag3Plus {
  __host__ __device__ Flag3 operator()(const Flag3& a, const Flag3& b) const { return Flag3{{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; }
};
constexpr unsigned long long kSahNoSplit = ~0ull;
constexpr int kSahQuant = 4;
constexpr uint32_t kSahPosMask = 0x0fffffffu;  // split position relative to the node's first position (n < 2^28); axis in bits 28-29

__global__ void __launch_bounds__(256) k_sah_keys(const Box6* __restrict__ tri_box, uint32_t n, int axis, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = f2ord(tri_box[i].mn[axis] + tri_box[i].mx[axis]);  // twice the centroid; ties keep id order (stable sort)
  ids[i] = i;
}
__global__ void __launch_bounds__(256) k_sah_init(uint32_t n, uint32_t* __restrict__ node_of, uint32_t* __restrict__ first, uint32_t* __restrict__ last,
                                                   uint32_t* __restrict__ node_parent, uint32_t* __restrict__ act, unsigned long long* __restrict__ best) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) node_of[i] = 0u;
  if (i == 0u) { first[0] = 0u; last[0] = n - 1u; node_parent[0] = kAbsent; act[0] = 0u; best[0] = kSahNoSplit; }
}
__global__ void __launch_bounds__(256) k_sah_gather(const uint32_t* __restrict__ ord, const Box6* __restrict__ tri_box, uint32_t n, Box6* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[p] = tri_box[ord[p]];
}
// cost of splitting the node of position p between p - 1 and p along `axis`; axis 0 also leaves the node's fitted box
__global__ void __launch_bounds__(256) k_sah_cost(uint32_t n, uint32_t axis, const uint32_t* __restrict__ node_of, const uint32_t* __restrict__ first,
                                                   const uint32_t* __restrict__ last, const Box6* __restrict__ box_l, const Box6* __restrict__ box_r,
                                                   unsigned long long* __restrict__ best, Box6* __restrict__ node_box, float pad, uint32_t quant) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t node = kAbsent;
  unsigned long long key = kSahNoSplit;
  if (p < n) {
    node = node_of[p];
    if (node != kAbsent) {
      const uint32_t lo = first[node], hi = last[node] + 1u;
      if (p > lo) {
        // triangles are tested `quant` at a time (a leaf item of the cooperative pass occupies kLeafSlots lanes however full it is)
        const float cost = half_area(box_l[p - 1u]) * (float)((p - lo + quant - 1u) / quant) + half_area(box_r[p]) * (float)((hi - p + quant - 1u) / quant);
        // areas and counts are >= 0: the bits of the cost order like the cost (a NaN cannot arise from finite boxes; +inf sorts last)
        key = ((unsigned long long)__float_as_uint(cost) << 32) | ((unsigned long long)axis << 28) | (unsigned long long)(p - lo);
      }
      if (axis == 0u && p + 1u == hi) {
        const Box6 u = box_l[p];
        Box6 w;
#pragma unroll
        for (int c = 0; c < 3; ++c) { w.mn[c] = u.mn[c] - pad; w.mx[c] = u.mx[c] + pad; }  // = the union of the widened leaf boxes (k_ploc_apply)
        node_box[node] = w;
      }
    }
  }
  // the positions of a node are contiguous: fold the wave's keys per run of equal nodes first, one atomic per run
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const unsigned long long ok = __shfl_down(key, off);
    const uint32_t on = __shfl_down(node, off);
    if (lane + off < 64u && on == node && ok < key) key = ok;
  }
  const uint32_t prev = __shfl_up(node, 1u);
  if (node != kAbsent && (lane == 0u || prev != node) && key != kSahNoSplit) atomicMin(&best[node], key);
}
// per open node: the split it takes, and how many of its two children are inner nodes (>= 2 triangles)
__global__ void __launch_bounds__(256) k_sah_decide(const uint32_t* __restrict__ act, uint32_t count, const unsigned long long* __restrict__ best,
                                                     const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, uint32_t force_median,
                                                     uint2* __restrict__ split, uint32_t* __restrict__ cnt) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const uint32_t node = act[j], lo = first[node], hi = last[node] + 1u;
  const unsigned long long key = best[node];
  uint32_t axis = (uint32_t)(key >> 28) & 3u, k = lo + ((uint32_t)key & kSahPosMask);
  if (force_median || key == kSahNoSplit || axis > 2u || k <= lo || k >= hi) { axis = 0u; k = lo + (hi - lo) / 2u; }  // very deep trees: halve
  split[node] = make_uint2(axis, k);
  cnt[j] = (k - lo >= 2u ? 1u : 0u) + (hi - k >= 2u ? 1u : 0u);
}
__global__ void __launch_bounds__(256) k_sah_children(const uint32_t* __restrict__ act, uint32_t count, const uint32_t* __restrict__ cnt,
                                                       const uint32_t* __restrict__ off, uint32_t next_id, const uint2* __restrict__ split,
                                                       uint32_t* __restrict__ first, uint32_t* __restrict__ last, uint32_t* __restrict__ left,
                                                       uint32_t* __restrict__ right, uint32_t* __restrict__ node_parent, uint32_t* __restrict__ leaf_parent,
                                                       uint32_t* __restrict__ act_next, unsigned long long* __restrict__ best, uint32_t* __restrict__ made) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const uint32_t node = act[j], lo = first[node], hi = last[node] + 1u, k = split[node].y;
  uint32_t slot = off[j];
  if (k - lo >= 2u) {
    const uint32_t id = next_id + slot;
    left[node] = id; first[id] = lo; last[id] = k - 1u; node_parent[id] = node; best[id] = kSahNoSplit; act_next[slot] = id;
    ++slot;
  } else { left[node] = kLeafBit | lo; leaf_parent[lo] = node; }
  if (hi - k >= 2u) {
    const uint32_t id = next_id + slot;
    right[node] = id; first[id] = k; last[id] = hi - 1u; node_parent[id] = node; best[id] = kSahNoSplit; act_next[slot] = id;
  } else { right[node] = kLeafBit | k; leaf_parent[k] = node; }
  if (j + 1u == count) made[0] = off[j] + cnt[j];
}
// side[triangle] = 1 when it goes to the left child of its node
__global__ void __launch_bounds__(256) k_sah_side(uint32_t n, const uint32_t* __restrict__ node_of, const uint2* __restrict__ split,
                                                   const uint32_t* __restrict__ ord0, const uint32_t* __restrict__ ord1, const uint32_t* __restrict__ ord2,
                                                   uint8_t* __restrict__ side) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t node = node_of[p];
  if (node == kAbsent) return;
  const uint2 sp = split[node];
  const uint32_t id = sp.x == 0u ? ord0[p] : (sp.x == 1u ? ord1[p] : ord2[p]);
  side[id] = p < sp.y ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_sah_flags(uint32_t n, const uint32_t* __restrict__ node_of, const uint32_t* __restrict__ ord0,
                                                    const uint32_t* __restrict__ ord1, const uint32_t* __restrict__ ord2, const uint8_t* __restrict__ side,
                                                    Flag3* __restrict__ flags) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  Flag3 f{{0u, 0u, 0u}};
  if (node_of[p] != kAbsent) { f.v[0] = side[ord0[p]]; f.v[1] = side[ord1[p]]; f.v[2] = side[ord2[p]]; }
  flags[p] = f;
}
// stable partition of every open node's range in the three orders, and the node every position belongs to in the next round
__global__ void __launch_bounds__(256) k_sah_scatter(uint32_t n, uint32_t* __restrict__ node_of, const uint2* __restrict__ split,
                                                      const uint32_t* __restrict__ first, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                      const Flag3* __restrict__ flags, const Flag3* __restrict__ scan, const uint32_t* __restrict__ in0,
                                                      const uint32_t* __restrict__ in1, const uint32_t* __restrict__ in2, uint32_t* __restrict__ out0,
                                                      uint32_t* __restrict__ out1, uint32_t* __restrict__ out2) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t node = node_of[p];
  const uint32_t id[3] = {in0[p], in1[p], in2[p]};
  uint32_t* const out[3] = {out0, out1, out2};
  if (node == kAbsent) {
#pragma unroll
    for (int b = 0; b < 3; ++b) out[b][p] = id[b];
    return;
  }
  const uint32_t lo = first[node], k = split[node].y;
  const Flag3 f = flags[p], sp = scan[p], s0 = scan[lo];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const uint32_t lefts = sp.v[b] - s0.v[b];  // left-going triangles before p in this node's range of order b
    out[b][f.v[b] ? lo + lefts : k + ((p - lo) - lefts)] = id[b];
  }
  const uint32_t child = p < k ? left[node] : right[node];
  node_of[p] = (child & kLeafBit) ? kAbsent : child;
}

// Device buffers of the builder.  A build makes ~60 of them; hipMalloc / hipFree cost 50-200 us each (the free also waits for the
// device), which was a third of a 1 M-triangle commit.  While an Arena is active on this thread, alloc() carves from it instead
// (256-B aligned bump allocation, released all at once with the arena); anything that does not fit falls back to hipMalloc.
struct Arena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  Arena* prev = nullptr;
  static Arena*& active() { static thread_local Arena* a = nullptr; return a; }
  std::string open(size_t bytes) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), bytes ? bytes : 256));
    cap = bytes;
    prev = active(); active() = this;
    return "";
  }
  void close() { if (active() == this) active() = prev; }  // no more carving; the memory lives until the destructor
  void* take(size_t bytes) {
    const size_t at = (used + 255u) & ~size_t(255);
    if (!base || at + bytes > cap) return nullptr;
    used = at + bytes;
    return base + at;
  }
  ~Arena() { close(); if (base) (void)hipFree(base); }
};
struct DevBuf {
  void* p = nullptr;
  bool owned = true;
  ~DevBuf() { if (p && owned) (void)hipFree(p); }
  std::string alloc(size_t bytes) {
    if (Arena* a = Arena::active()) {
      if ((p = a->take(bytes ? bytes : 16)) != nullptr) { owned = false; return ""; }
    }
    owned = true;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    return "";
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

inline uint32_t nblk(uint32_t n) { return (n + 255u) / 256u; }

}  // namespace

// Persistent topology kept for refit.
struct BvhTopology {
  Arena arena;  // first member: destroyed last, after the buffers carved from it
  uint32_t n = 0, leaf_max = 0;
  DevBuf left, right, first, last, node_parent, leaf_parent, keep, sorted_ids, tri_box, leaf_box, node_box, arrivals, scene_ord, block_ord;
  // 4-wide collapse: binary root of every 4-node (BFS order), the binary refs its slots were filled from, the 4-node
  // index of every binary root, per-level scratch
  DevBuf root_of, refs4, index4, cnt, off;
  bool collapsed = false;
  bool fitted = false;               // node_box already holds the fitted boxes (PLOC writes them as it merges)
  std::vector<uint32_t> level_base;  // first 4-node of every BFS level, then node_count: refit walks the levels bottom-up
};

// RENDER_SPEC 4.1b: leaf boxes are widened by 2^-19 of the scene's largest coordinate
static float box_pad(const BvhBuffers& b) {
  float amax = 0.0f;
  for (int k = 0; k < 3; ++k) amax = std::max(amax, std::max(std::fabs(b.scene_min[k]), std::fabs(b.scene_max[k])));
  return amax * 1.9073486328125e-06f;
}

static std::string flatten_and_bounds(BvhBuffers& b, BvhTopology& t, hipStream_t s) {
  const uint32_t n = b.tri_count;
  if (n) {
    hipLaunchKernelGGL(k_flatten, dim3(nblk(n)), dim3(256), 0, s, b.primitives, b.inst_first_tri, b.instance_count, n, b.tris_by_id, b.shade_tris,
                       t.tri_box.as<Box6>(), t.block_ord.as<uint32_t>(), b.gid_first, b.inst_index, b.object_space ? 1 : 0);
    hipLaunchKernelGGL(k_bounds_reduce, dim3(1), dim3(256), 0, s, t.block_ord.as<uint32_t>(), nblk(n), t.scene_ord.as<uint32_t>());
  }
  uint32_t ord[6];
  HIP_TRY(hipMemcpyAsync(ord, t.scene_ord.p, sizeof(ord), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int k = 0; k < 3; ++k) {
    b.scene_min[k] = n ? ord2f(ord[k]) : 0.0f;
    b.scene_max[k] = n ? ord2f(ord[3 + k]) : 0.0f;
  }
  return "";
}

static std::string fit_and_emit(BvhBuffers& b, BvhTopology& t, hipStream_t s) {
  const uint32_t n = b.tri_count;
  if (n) hipLaunchKernelGGL(k_leaf_boxes, dim3(nblk(n)), dim3(256), 0, s, t.tri_box.as<Box6>(), t.sorted_ids.as<uint32_t>(), n,
                            t.leaf_box.as<Box6>(), b.tris_by_id, b.tris, box_pad(b), b.tris_any, b.shade_tris, b.material_any_class, b.material_count, b.material_kind);
  if (n <= t.leaf_max || n < 2) {
    hipLaunchKernelGGL(k_emit_single4, dim3(1), dim3(64), 0, s, t.leaf_box.as<Box6>(), n, b.nodes);
    b.node_count = 1;
    b.max_depth = 1;
    b.stack_need = 1;
    HIP_TRY(hipStreamSynchronize(s));
    return "";
  }
  const uint32_t ni = n - 1;
  if (t.collapsed) {
    // refit: the 4-wide topology is kept; levels are contiguous in BFS order, so one launch per level from the deepest up
    // re-derives every node from its children's boxes (leaf boxes, or the union a child node left in node_box[its index])
    for (size_t l = t.level_base.size() - 1; l-- > 0;) {
      const uint32_t base = t.level_base[l], size = t.level_base[l + 1] - base;
      hipLaunchKernelGGL(k_refit_level, dim3(nblk(size)), dim3(256), 0, s, t.refs4.as<uint4>(), base, size, t.first.as<uint32_t>(),
                         t.last.as<uint32_t>(), t.keep.as<uint32_t>(), t.index4.as<uint32_t>(), t.leaf_box.as<Box6>(), t.node_box.as<Box6>(),
                         b.nodes);
    }
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return "";
  }
  if (!t.fitted) {
    HIP_TRY(hipMemsetAsync(t.arrivals.p, 0, (size_t)ni * 4, s));
    hipLaunchKernelGGL(k_fit, dim3(nblk(n)), dim3(256), 0, s, t.left.as<uint32_t>(), t.right.as<uint32_t>(), t.node_parent.as<uint32_t>(),
                       t.leaf_parent.as<uint32_t>(), t.leaf_box.as<Box6>(), t.node_box.as<Box6>(), t.arrivals.as<uint32_t>(), n);
  }
  if (!t.collapsed) {  // first build: choose the 4-wide topology from the fitted boxes; refit keeps it
    static const uint32_t zero = 0;
    HIP_TRY(hipMemcpyAsync(t.root_of.p, &zero, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(t.index4.p, &zero, 4, hipMemcpyHostToDevice, s));
    DevBuf tmp;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, t.cnt.as<uint32_t>(), t.off.as<uint32_t>(), 0u, ni, rocprim::plus<uint32_t>(), s));
    std::string e = tmp.alloc(tmp_bytes);
    if (!e.empty()) return e;
    // levels are driven like PLOC's rounds: {first node, size, level} on the device, the host looks every few levels; between two
    // looks the grid covers the largest size the level can have reached (a level is at most 4 times the one above, and < n)
    constexpr uint32_t kMaxLevels = 96;
    uint32_t look_every = 8;
    if (b.opt.collapse_look_every) look_every = b.opt.collapse_look_every;
    // slots by descending surface area of the inner children: configs[3] 9.09 instead of 9.40 node visits per connection ray (in-order
    // slots), 3.40 instead of 3.45 ms per frame in the shadow passes; ascending area / triangle count / density: 9.60 / 9.48 / 9.36
    int order_mode = 1;
    if (const char* ev = tune_env("HALART_CHILD_ORDER")) order_mode = atoi(ev);
    DevBuf level, bases;
    if (!(e = level.alloc(16)).empty()) return e;
    if (!(e = bases.alloc(kMaxLevels * 4)).empty()) return e;
    const uint32_t init_level[3] = {0u, 1u, 0u};
    HIP_TRY(hipMemcpyAsync(level.p, init_level, 12, hipMemcpyHostToDevice, s));
    uint32_t now[3] = {0u, 1u, 0u};
    while (now[1] > 0u) {
      unsigned long long bound = now[1];
      for (uint32_t k = 0; k < look_every; ++k) {
        const uint32_t bd = (uint32_t)std::min<unsigned long long>(bound, ni);
        hipLaunchKernelGGL(k_collapse_level, dim3(nblk(bd)), dim3(256), 0, s, t.root_of.as<uint32_t>(), level.as<uint32_t>(), bd, t.left.as<uint32_t>(),
                           t.right.as<uint32_t>(), t.keep.as<uint32_t>(), t.node_box.as<Box6>(), t.refs4.as<uint4>(), t.cnt.as<uint32_t>(),
                           t.first.as<uint32_t>(), t.last.as<uint32_t>(), order_mode);
        size_t tb = tmp_bytes;
        HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, t.cnt.as<uint32_t>(), t.off.as<uint32_t>(), 0u, bd, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(k_scatter_level, dim3(nblk(bd)), dim3(256), 0, s, t.root_of.as<uint32_t>(), level.as<uint32_t>(), t.refs4.as<uint4>(),
                           t.keep.as<uint32_t>(), t.off.as<uint32_t>(), t.index4.as<uint32_t>());
        hipLaunchKernelGGL(k_level_advance, dim3(1), dim3(64), 0, s, t.cnt.as<uint32_t>(), t.off.as<uint32_t>(), level.as<uint32_t>(), bases.as<uint32_t>(),
                           kMaxLevels);
        bound *= 4ull;
      }
      HIP_TRY(hipMemcpyAsync(now, level.p, 12, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (now[2] > kMaxLevels) return "bvh_build: the tree is deeper than " + std::to_string(kMaxLevels) + " 4-wide levels";
    }
    const uint32_t base = now[0], levels = now[2];
    t.level_base.assign(levels, 0u);
    HIP_TRY(hipMemcpy(t.level_base.data(), bases.p, (size_t)levels * 4, hipMemcpyDeviceToHost));
    t.level_base.push_back(base);
    b.node_count = base;
    b.max_depth = levels;
    b.stack_need = 3u * levels;  // a 4-node visit defers at most three siblings
    t.collapsed = true;
  }
  hipLaunchKernelGGL(k_pack4, dim3(nblk(b.node_count)), dim3(256), 0, s, t.refs4.as<uint4>(), b.node_count, t.first.as<uint32_t>(),
                     t.last.as<uint32_t>(), t.keep.as<uint32_t>(), t.index4.as<uint32_t>(), t.leaf_box.as<Box6>(), t.node_box.as<Box6>(),
                     b.nodes);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return "";
}

// Builds left/right/parents/first/last with PLOC instead of k_hierarchy (same contract; t.sorted_ids is re-ordered).
static std::string ploc_hierarchy(BvhBuffers& b, BvhTopology& t, hipStream_t s) {
  const uint32_t n = b.tri_count, ni = n - 1;
  DevBuf ref[2], box[2], nn, merge, keepc, merge_scan, keep_scan, subtree, new_pos, ids2, lp2, tmp;
  std::string e;
  for (int k = 0; k < 2; ++k) {
    if (!(e = ref[k].alloc((size_t)n * 4)).empty()) return e;
    if (!(e = box[k].alloc((size_t)n * sizeof(Box6))).empty()) return e;
  }
  for (DevBuf* d : {&nn, &merge, &keepc, &merge_scan, &keep_scan, &new_pos, &ids2, &lp2})
    if (!(e = d->alloc((size_t)n * 4)).empty()) return e;
  if (!(e = subtree.alloc((size_t)ni * 4)).empty()) return e;
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, merge.as<uint32_t>(), merge_scan.as<uint32_t>(), 0u, n, rocprim::plus<uint32_t>(), s));
  if (!(e = tmp.alloc(tmp_bytes)).empty()) return e;
  hipLaunchKernelGGL(k_ploc_init, dim3(nblk(n)), dim3(256), 0, s, t.tri_box.as<Box6>(), t.sorted_ids.as<uint32_t>(), n, ref[0].as<uint32_t>(),
                     box[0].as<Box6>());
  uint32_t m = n, created = 0;
  int cur = 0;
  uint32_t radius = kPlocRadius;
  if (const char* ev = tune_env("HALART_PLOC_RADIUS")) radius = (uint32_t)std::max(1, atoi(ev));
  bool tail = true;
  if (b.opt.ploc_tail == 2u) tail = false;  // every round as separate launches
  // rounds between two looks at the device's counters: a look is a host round trip (~50 us, as long as a late round itself); the
  // grid of the rounds in between is sized for the count of the last look (clusters only get fewer).  Same tree for any value.
  uint32_t look_every = 6;
  if (b.opt.ploc_look_every) look_every = b.opt.ploc_look_every;
  DevBuf state;
  if (!(e = state.alloc(8)).empty()) return e;
  const uint32_t init_state[2] = {n, 0u};
  HIP_TRY(hipMemcpyAsync(state.p, init_state, 8, hipMemcpyHostToDevice, s));
  while (m > 1) {
    if (tail && m <= kPlocTail) {
      hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kPlocTail), 0, s, m, radius, (ni - 1u) - created, ref[cur].as<uint32_t>(), box[cur].as<Box6>(),
                         t.left.as<uint32_t>(), t.right.as<uint32_t>(), t.node_parent.as<uint32_t>(), t.leaf_parent.as<uint32_t>(),
                         subtree.as<uint32_t>(), t.node_box.as<Box6>(), box_pad(b));
      break;
    }
    const uint32_t bound = m;
    for (uint32_t round = 0; round < look_every; ++round) {
      hipLaunchKernelGGL(k_ploc_nearest, dim3(nblk(bound)), dim3(256), 0, s, box[cur].as<Box6>(), state.as<uint32_t>(), radius, nn.as<uint32_t>());
      hipLaunchKernelGGL(k_ploc_flags, dim3(nblk(bound)), dim3(256), 0, s, nn.as<uint32_t>(), state.as<uint32_t>(), bound, merge.as<uint32_t>(),
                         keepc.as<uint32_t>());
      size_t tb = tmp_bytes;
      HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, merge.as<uint32_t>(), merge_scan.as<uint32_t>(), 0u, bound, rocprim::plus<uint32_t>(), s));
      tb = tmp_bytes;
      HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, keepc.as<uint32_t>(), keep_scan.as<uint32_t>(), 0u, bound, rocprim::plus<uint32_t>(), s));
      hipLaunchKernelGGL(k_ploc_apply, dim3(nblk(bound)), dim3(256), 0, s, nn.as<uint32_t>(), merge.as<uint32_t>(), keepc.as<uint32_t>(),
                         merge_scan.as<uint32_t>(), keep_scan.as<uint32_t>(), state.as<uint32_t>(), ni - 1u, ref[cur].as<uint32_t>(), box[cur].as<Box6>(),
                         ref[cur ^ 1].as<uint32_t>(), box[cur ^ 1].as<Box6>(), t.left.as<uint32_t>(), t.right.as<uint32_t>(),
                         t.node_parent.as<uint32_t>(), t.leaf_parent.as<uint32_t>(), subtree.as<uint32_t>(), t.node_box.as<Box6>(), box_pad(b));
      hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(64), 0, s, merge.as<uint32_t>(), merge_scan.as<uint32_t>(), state.as<uint32_t>());
      cur ^= 1;
    }
    uint32_t now[2] = {0u, 0u};
    HIP_TRY(hipMemcpyAsync(now, state.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (now[0] >= m) return "bvh_build: PLOC made no progress";  // cannot happen: the closest pair is always mutual
    m = now[0];
    created = now[1];
  }
  static const uint32_t absent = kAbsent;
  HIP_TRY(hipMemcpyAsync(t.node_parent.p, &absent, 4, hipMemcpyHostToDevice, s));  // node 0 is the root
  hipLaunchKernelGGL(k_ploc_leaf_order, dim3(nblk(n)), dim3(256), 0, s, t.left.as<uint32_t>(), t.right.as<uint32_t>(), t.node_parent.as<uint32_t>(),
                     t.leaf_parent.as<uint32_t>(), subtree.as<uint32_t>(), t.sorted_ids.as<uint32_t>(), n, new_pos.as<uint32_t>(),
                     ids2.as<uint32_t>(), lp2.as<uint32_t>());
  hipLaunchKernelGGL(k_ploc_node_ranges, dim3(nblk(ni)), dim3(256), 0, s, t.left.as<uint32_t>(), t.right.as<uint32_t>(), t.node_parent.as<uint32_t>(),
                     subtree.as<uint32_t>(), ni, t.first.as<uint32_t>(), t.last.as<uint32_t>());
  hipLaunchKernelGGL(k_ploc_renumber, dim3(nblk(ni)), dim3(256), 0, s, t.left.as<uint32_t>(), t.right.as<uint32_t>(), new_pos.as<uint32_t>(), ni);
  HIP_TRY(hipMemcpyAsync(t.sorted_ids.p, ids2.p, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(t.leaf_parent.p, lp2.p, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return "";
}


// Builds the same tables as ploc_hierarchy with the full-sweep SAH rounds above (t.sorted_ids is re-ordered; node boxes come out fitted).
static std::string sah_hierarchy(BvhBuffers& b, BvhTopology& t, hipStream_t s) {
  const uint32_t n = b.tri_count, ni = n - 1;
  DevBuf keys_in, keys_out, ids_in, ord[3][2], node_of, box_in, box_l, box_r, best, side, flags, scan, act[2], cnt, off, split, made, tmp;
  std::string e;
  for (DevBuf* d : {&keys_in, &keys_out, &ids_in, &ord[0][0], &ord[0][1], &ord[1][0], &ord[1][1], &ord[2][0], &ord[2][1], &node_of, &act[0], &act[1], &cnt, &off})
    if (!(e = d->alloc((size_t)n * 4)).empty()) return e;
  for (DevBuf* d : {&box_in, &box_l, &box_r})
    if (!(e = d->alloc((size_t)n * sizeof(Box6))).empty()) return e;
  if (!(e = best.alloc((size_t)ni * 8)).empty()) return e;
  if (!(e = split.alloc((size_t)ni * 8)).empty()) return e;
  if (!(e = side.alloc(n)).empty()) return e;
  if (!(e = flags.alloc((size_t)n * sizeof(Flag3))).empty()) return e;
  if (!(e = scan.alloc((size_t)n * sizeof(Flag3))).empty()) return e;
  if (!(e = made.alloc(16)).empty()) return e;
  const uint32_t* nk = node_of.as<uint32_t>();
  auto rk = rocprim::make_reverse_iterator(nk + n);
  auto rin = rocprim::make_reverse_iterator(box_in.as<Box6>() + n);
  auto rout = rocprim::make_reverse_iterator(box_r.as<Box6>() + n);
  size_t need[5] = {0, 0, 0, 0, 0};
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, need[0], keys_in.as<uint32_t>(), keys_out.as<uint32_t>(), ids_in.as<uint32_t>(), ord[0][0].as<uint32_t>(), n, 0, 32, s));
  HIP_TRY(rocprim::inclusive_scan_by_key(nullptr, need[1], nk, box_in.as<Box6>(), box_l.as<Box6>(), n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
  HIP_TRY(rocprim::inclusive_scan_by_key(nullptr, need[2], rk, rin, rout, n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
  HIP_TRY(rocprim::exclusive_scan(nullptr, need[3], cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, n, rocprim::plus<uint32_t>(), s));
  HIP_TRY(rocprim::exclusive_scan(nullptr, need[4], flags.as<Flag3>(), scan.as<Flag3>(), Flag3{{0u, 0u, 0u}}, n, Flag3Plus(), s));
  size_t tmp_bytes = 0;
  for (size_t v : need) tmp_bytes = std::max(tmp_bytes, v);
  if (!(e = tmp.alloc(tmp_bytes)).empty()) return e;
  for (int a = 0; a < 3; ++a) {
    hipLaunchKernelGGL(k_sah_keys, dim3(nblk(n)), dim3(256), 0, s, t.tri_box.as<Box6>(), n, a, keys_in.as<uint32_t>(), ids_in.as<uint32_t>());
    size_t tb = tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys_in.as<uint32_t>(), keys_out.as<uint32_t>(), ids_in.as<uint32_t>(), ord[a][0].as<uint32_t>(), n, 0, 32, s));
  }
  hipLaunchKernelGGL(k_sah_init, dim3(nblk(n)), dim3(256), 0, s, n, node_of.as<uint32_t>(), t.first.as<uint32_t>(), t.last.as<uint32_t>(),
                     t.node_parent.as<uint32_t>(), act[0].as<uint32_t>(), best.as<unsigned long long>());
  const float pad = box_pad(b);
  uint32_t quant = (uint32_t)kSahQuant;  // = the leaf slots of the cooperative pass (traverse.h: RT_LEAF_SLOTS)
  if (const char* ev = tune_env("HALART_SAH_QUANT")) quant = (uint32_t)std::max(1, atoi(ev));
  uint32_t open = 1, next_id = 1, round = 0;
  int cur = 0, ac = 0;
  constexpr uint32_t kSweepRounds = 64;  // below that depth every node is halved instead: at most 28 more rounds
  while (open > 0) {
    if (round > kSweepRounds + 32u) return "bvh_build: the SAH rounds do not terminate";
    const uint32_t* o[3] = {ord[0][cur].as<uint32_t>(), ord[1][cur].as<uint32_t>(), ord[2][cur].as<uint32_t>()};
    if (round < kSweepRounds) {
      for (uint32_t a = 0; a < 3; ++a) {
        hipLaunchKernelGGL(k_sah_gather, dim3(nblk(n)), dim3(256), 0, s, o[a], t.tri_box.as<Box6>(), n, box_in.as<Box6>());
        size_t tb = tmp_bytes;
        HIP_TRY(rocprim::inclusive_scan_by_key(tmp.p, tb, nk, box_in.as<Box6>(), box_l.as<Box6>(), n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
        tb = tmp_bytes;
        HIP_TRY(rocprim::inclusive_scan_by_key(tmp.p, tb, rk, rin, rout, n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
        hipLaunchKernelGGL(k_sah_cost, dim3(nblk(n)), dim3(256), 0, s, n, a, nk, t.first.as<uint32_t>(), t.last.as<uint32_t>(), box_l.as<Box6>(),
                           box_r.as<Box6>(), best.as<unsigned long long>(), t.node_box.as<Box6>(), pad, quant);
      }
    } else {  // the boxes are still needed: axis 0 only
      hipLaunchKernelGGL(k_sah_gather, dim3(nblk(n)), dim3(256), 0, s, o[0], t.tri_box.as<Box6>(), n, box_in.as<Box6>());
      size_t tb = tmp_bytes;
      HIP_TRY(rocprim::inclusive_scan_by_key(tmp.p, tb, nk, box_in.as<Box6>(), box_l.as<Box6>(), n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
      tb = tmp_bytes;
      HIP_TRY(rocprim::inclusive_scan_by_key(tmp.p, tb, rk, rin, rout, n, BoxUnionOp(), rocprim::equal_to<uint32_t>(), s));
      hipLaunchKernelGGL(k_sah_cost, dim3(nblk(n)), dim3(256), 0, s, n, 0u, nk, t.first.as<uint32_t>(), t.last.as<uint32_t>(), box_l.as<Box6>(),
                         box_r.as<Box6>(), best.as<unsigned long long>(), t.node_box.as<Box6>(), pad, quant);
    }
    hipLaunchKernelGGL(k_sah_decide, dim3(nblk(open)), dim3(256), 0, s, act[ac].as<uint32_t>(), open, best.as<unsigned long long>(), t.first.as<uint32_t>(),
                       t.last.as<uint32_t>(), round >= kSweepRounds ? 1u : 0u, split.as<uint2>(), cnt.as<uint32_t>());
    size_t tb = tmp_bytes;
    HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, open, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_sah_children, dim3(nblk(open)), dim3(256), 0, s, act[ac].as<uint32_t>(), open, cnt.as<uint32_t>(), off.as<uint32_t>(), next_id,
                       split.as<uint2>(), t.first.as<uint32_t>(), t.last.as<uint32_t>(), t.left.as<uint32_t>(), t.right.as<uint32_t>(),
                       t.node_parent.as<uint32_t>(), t.leaf_parent.as<uint32_t>(), act[ac ^ 1].as<uint32_t>(), best.as<unsigned long long>(),
                       made.as<uint32_t>());
    hipLaunchKernelGGL(k_sah_side, dim3(nblk(n)), dim3(256), 0, s, n, nk, split.as<uint2>(), o[0], o[1], o[2], side.as<uint8_t>());
    hipLaunchKernelGGL(k_sah_flags, dim3(nblk(n)), dim3(256), 0, s, n, nk, o[0], o[1], o[2], side.as<uint8_t>(), flags.as<Flag3>());
    tb = tmp_bytes;
    HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, flags.as<Flag3>(), scan.as<Flag3>(), Flag3{{0u, 0u, 0u}}, n, Flag3Plus(), s));
    hipLaunchKernelGGL(k_sah_scatter, dim3(nblk(n)), dim3(256), 0, s, n, node_of.as<uint32_t>(), split.as<uint2>(), t.first.as<uint32_t>(),
                       t.left.as<uint32_t>(), t.right.as<uint32_t>(), flags.as<Flag3>(), scan.as<Flag3>(), o[0], o[1], o[2], ord[0][cur ^ 1].as<uint32_t>(),
                       ord[1][cur ^ 1].as<uint32_t>(), ord[2][cur ^ 1].as<uint32_t>());
    uint32_t now = 0;
    HIP_TRY(hipMemcpyAsync(&now, made.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    next_id += now;
    if (next_id > ni) return "bvh_build: the SAH rounds made more nodes than a binary tree has";
    open = now;
    cur ^= 1; ac ^= 1; ++round;
  }
  if (next_id != ni) return "bvh_build: the SAH rounds made " + std::to_string(next_id) + " of " + std::to_string(ni) + " nodes";
  HIP_TRY(hipMemcpyAsync(t.sorted_ids.p, ord[0][cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return "";
}

void bvh_free_topology(void* topo) { delete static_cast<BvhTopology*>(topo); }

std::string bvh_build(BvhBuffers& b, uint32_t leaf_max, hipStream_t s) {
  const uint32_t n = b.tri_count;
  if (b.topology) { bvh_free_topology(b.topology); b.topology = nullptr; }
  BvhTopology* tp = new BvhTopology();
  b.topology = tp;
  BvhTopology& t = *tp;
  t.n = n; t.leaf_max = leaf_max;
  std::string e;
  const size_t ni = n > 1 ? n - 1 : 1;
  // one allocation for everything the topology keeps (~150 B per triangle) and one for the build's temporaries (~170 B per
  // triangle: sort keys, PLOC's double-buffered clusters and scans), instead of ~60 hipMalloc / hipFree pairs
  if (!(e = t.arena.open((size_t)n * 160 + (1u << 20))).empty()) return e;
#define ALLOC(buf, bytes) if (!(e = t.buf.alloc(bytes)).empty()) return e
  ALLOC(left, ni * 4); ALLOC(right, ni * 4); ALLOC(first, ni * 4); ALLOC(last, ni * 4); ALLOC(node_parent, ni * 4);
  ALLOC(leaf_parent, (size_t)n * 4); ALLOC(keep, ni * 4); ALLOC(sorted_ids, (size_t)n * 4);
  ALLOC(tri_box, (size_t)n * sizeof(Box6)); ALLOC(leaf_box, (size_t)n * sizeof(Box6)); ALLOC(node_box, ni * sizeof(Box6));
  ALLOC(arrivals, ni * 4); ALLOC(scene_ord, 6 * 4); ALLOC(block_ord, (size_t)(nblk(n) + 1u) * 24);
  ALLOC(root_of, ni * 4); ALLOC(refs4, ni * 16); ALLOC(index4, ni * 4); ALLOC(cnt, ni * 4); ALLOC(off, ni * 4);
#undef ALLOC
  t.arena.close();
  Arena scratch;  // declared before the temporaries below: released after them
  if (!(e = scratch.open((size_t)n * 200 + (8u << 20))).empty()) return e;
  if (leaf_max > 8u || n >= (1u << 28)) return "bvh_build: the 4-wide node format holds leaves of <= 8 triangles and < 2^28 triangles";
  if (!(e = flatten_and_bounds(b, t, s)).empty()) return e;
  if (n >= 2) {
    // hierarchy over the triangles: full-sweep SAH rounds for scenes large enough to repay them (1 M triangles: 34 ms of build instead
    // of PLOC's 11 ms for 10-15 % fewer node visits per closest-hit ray, profiles/r02_experiments.txt), Karras' LBVH over the Morton
    // order otherwise; PLOC (nearest-neighbour clustering along the Morton order) stays selectable as the fast large-scene build
    const bool sah = b.opt.builder ? b.opt.builder == 1u : n >= 4096u;
    const bool ploc = b.opt.builder == 2u;
    if (sah) {  // (needs no Morton order)
      if (!(e = sah_hierarchy(b, t, s)).empty()) return e;
      t.fitted = true;
    } else {
      DevBuf keys_in, keys_out, ids_in, tmp;
      if (!(e = keys_in.alloc((size_t)n * 8)).empty()) return e;
      if (!(e = keys_out.alloc((size_t)n * 8)).empty()) return e;
      if (!(e = ids_in.alloc((size_t)n * 4)).empty()) return e;
      uint32_t size_classes = 1;
      if (const char* ev = tune_env("HALART_SIZE_CLASSES")) size_classes = (uint32_t)atoi(ev);
      hipLaunchKernelGGL(k_morton, dim3(nblk(n)), dim3(256), 0, s, t.tri_box.as<Box6>(), n, t.scene_ord.as<uint32_t>(),
                         keys_in.as<unsigned long long>(), ids_in.as<uint32_t>(), size_classes);
      size_t tmp_bytes = 0;
      HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                        ids_in.as<uint32_t>(), t.sorted_ids.as<uint32_t>(), n, 0, 64, s));
      if (!(e = tmp.alloc(tmp_bytes)).empty()) return e;
      HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                        ids_in.as<uint32_t>(), t.sorted_ids.as<uint32_t>(), n, 0, 64, s));
      if (ploc) {
        if (!(e = ploc_hierarchy(b, t, s)).empty()) return e;
        t.fitted = true;
      } else
        hipLaunchKernelGGL(k_hierarchy, dim3(nblk(n - 1)), dim3(256), 0, s, keys_out.as<unsigned long long>(), (int)n, t.left.as<uint32_t>(),
                           t.right.as<uint32_t>(), t.first.as<uint32_t>(), t.last.as<uint32_t>(), t.node_parent.as<uint32_t>(),
                           t.leaf_parent.as<uint32_t>());
    }
    hipLaunchKernelGGL(k_keep_flags, dim3(nblk(n - 1)), dim3(256), 0, s, t.first.as<uint32_t>(), t.last.as<uint32_t>(), n - 1, leaf_max,
                       t.keep.as<uint32_t>());
  } else if (n == 1) {
    static const uint32_t zero = 0;
    HIP_TRY(hipMemcpyAsync(t.sorted_ids.p, &zero, 4, hipMemcpyHostToDevice, s));
  }
  if (!(e = fit_and_emit(b, t, s)).empty()) return e;
  return "";
}

__global__ void __launch_bounds__(256) k_relocate(BvhNode4* __restrict__ nodes, uint32_t count, uint32_t node_offset, uint32_t tri_offset) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * 4u) return;
  uint32_t& ref = nodes[i >> 2].ref[i & 3u];
  if (ref == kAbsent) return;
  ref += (ref & kLeafRef) ? tri_offset : node_offset;  // a leaf's first triangle sits in the low 28 bits: the sum stays below 2^28 (checked by the caller)
}
std::string bvh_relocate(BvhBuffers& b, uint32_t node_offset, uint32_t tri_offset, hipStream_t s) {
  if (node_offset == 0u && tri_offset == 0u) return "";
  if ((unsigned long long)tri_offset + b.tri_count >= (1ull << 28)) return "bvh_relocate: the scene holds 2^28 triangles or more";
  hipLaunchKernelGGL(k_relocate, dim3(nblk(b.node_count * 4u)), dim3(256), 0, s, b.nodes, b.node_count, node_offset, tri_offset);
  HIP_TRY(hipGetLastError());
  return "";
}

std::string bvh_refit(BvhBuffers& b, hipStream_t s) {
  if (!b.topology) return "bvh_refit: no BVH has been built";
  BvhTopology& t = *static_cast<BvhTopology*>(b.topology);
  if (t.n != b.tri_count) return "bvh_refit: triangle count changed, rebuild required";
  std::string e;
  if (!(e = flatten_and_bounds(b, t, s)).empty()) return e;
  const uint32_t nc = b.node_count, md = b.max_depth, sn = b.stack_need;
  if (!(e = fit_and_emit(b, t, s)).empty()) return e;
  if (b.tri_count > t.leaf_max && b.tri_count >= 2) { b.node_count = nc; b.max_depth = md; b.stack_need = sn; }
  return "";
}

}  // namespace rt
