// bvh_tlas.cpp — the top levels of a two-level tree (RENDER_SPEC 4.5): what the reference asks the driver for with its instance list
// (src/scene/loader/gpu_uploader.rs:843-885, :937-959).  Host code: a scene has tens to thousands of instances, their tree is rebuilt
// from scratch in microseconds whenever a node moves (hala_rt_refit), and the big per-primitive trees underneath stay untouched.
//
// Items are boxes with a child reference each: an instance leaf (kInstLeafTag | index into the InstRef table) or the root of a subtree
// that needs no transform (the tree over all triangles that are NOT instanced: flattened to world space as before).  Output: 64-B
// compressed 4-wide nodes (RENDER_SPEC 4.1b), breadth-first, root = 0, every child box conservatively quantised.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.h"

namespace rt {
namespace {

struct Box { float mn[3], mx[3]; };
inline Box unite(const Box& a, const Box& b) {
  Box r;
  for (int k = 0; k < 3; ++k) { r.mn[k] = std::min(a.mn[k], b.mn[k]); r.mx[k] = std::max(a.mx[k], b.mx[k]); }
  return r;
}
inline double half_area(const Box& b) {
  const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
  return dx * dy + dy * dz + dz * dx;
}
struct BinNode { Box box; int left = -1, right = -1; int item = -1; };  // item >= 0: leaf of the binary tree = one TLAS item

// full-sweep SAH over the items' boxes (cost = area x items on each side), median split when the sweep finds nothing
int build_binary(std::vector<BinNode>& nodes, const std::vector<TlasItem>& items, std::vector<uint32_t>& ids, size_t lo, size_t hi) {
  BinNode n;
  n.box = Box{};
  memcpy(n.box.mn, items[ids[lo]].mn, 12); memcpy(n.box.mx, items[ids[lo]].mx, 12);
  for (size_t k = lo + 1; k < hi; ++k) { Box b; memcpy(b.mn, items[ids[k]].mn, 12); memcpy(b.mx, items[ids[k]].mx, 12); n.box = unite(n.box, b); }
  const int index = (int)nodes.size();
  nodes.push_back(n);
  if (hi - lo == 1) { nodes[index].item = (int)ids[lo]; return index; }
  double best = -1.0;
  int best_axis = 0;
  size_t best_split = (lo + hi) / 2;
  std::vector<uint32_t> order(ids.begin() + lo, ids.begin() + hi), best_order;
  std::vector<double> right_area(hi - lo);
  for (int axis = 0; axis < 3; ++axis) {
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      return items[a].mn[axis] + items[a].mx[axis] < items[b].mn[axis] + items[b].mx[axis];
    });
    Box acc{};
    for (size_t k = order.size(); k-- > 0;) {
      Box b; memcpy(b.mn, items[order[k]].mn, 12); memcpy(b.mx, items[order[k]].mx, 12);
      acc = k + 1 == order.size() ? b : unite(acc, b);
      right_area[k] = half_area(acc);
    }
    for (size_t k = 0; k + 1 < order.size(); ++k) {
      Box b; memcpy(b.mn, items[order[k]].mn, 12); memcpy(b.mx, items[order[k]].mx, 12);
      acc = k == 0 ? b : unite(acc, b);
      const double cost = half_area(acc) * (double)(k + 1) + right_area[k + 1] * (double)(order.size() - k - 1);
      if (best < 0.0 || cost < best) { best = cost; best_axis = axis; best_split = lo + k + 1; best_order = order; }
    }
  }
  (void)best_axis;
  if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), ids.begin() + lo);
  const int l = build_binary(nodes, items, ids, lo, best_split);
  const int r = build_binary(nodes, items, ids, best_split, hi);
  nodes[index].left = l; nodes[index].right = r;
  return index;
}

// quantum exponent of one axis: the smallest power of two s = 2^e with extent / s <= 255 (never below 2^-100) — as bvh_build.hip
int quantum_exponent(float lo, float hi) {
  const double q = ((double)hi - (double)lo) / 255.0;
  if (!(q > 0.0)) return -100;
  int e = 0;
  (void)std::frexp(q, &e);  // q = m * 2^e, m in [0.5, 1)
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

}  // namespace

uint32_t tlas_build(const std::vector<TlasItem>& items, std::vector<BvhNode4>& out, uint32_t* levels, uint32_t* stack_need) {
  out.clear();
  if (levels) *levels = 0;
  if (stack_need) *stack_need = 0;
  if (items.empty()) return 0;
  std::vector<BinNode> bin;
  std::vector<uint32_t> ids(items.size());
  for (size_t k = 0; k < ids.size(); ++k) ids[k] = (uint32_t)k;
  bin.reserve(items.size() * 2);
  build_binary(bin, items, ids, 0, items.size());
  // collapse to 4-wide, breadth-first: a 4-node starts with the two children of its binary root and opens the inner child of largest
  // surface area while a slot is free
  std::vector<int> roots{0};   // binary root of every 4-node, in BFS order
  std::vector<uint32_t> depth{1};
  if (bin[0].item >= 0) {  // a single item: one node with one child
    roots.clear();
  }
  std::vector<std::array<int, 4>> slots;
  for (size_t q = 0; q < roots.size(); ++q) {
    const BinNode& r = bin[(size_t)roots[q]];
    std::array<int, 4> c{r.left, r.right, -1, -1};
    int n = 2;
    while (n < 4) {
      int pick = -1;
      double best = -1.0;
      for (int k = 0; k < n; ++k) {
        const BinNode& ch = bin[(size_t)c[k]];
        if (ch.item >= 0) continue;
        const double a = half_area(ch.box);
        if (a > best) { best = a; pick = k; }
      }
      if (pick < 0) break;
      const BinNode& open = bin[(size_t)c[pick]];
      c[pick] = open.left;
      c[n++] = open.right;
    }
    slots.push_back(c);
    for (int k = 0; k < n; ++k)
      if (bin[(size_t)c[k]].item < 0) { roots.push_back(c[k]); depth.push_back(depth[q] + 1u); }
  }
  if (roots.empty()) {  // single item
    slots.push_back({0, -1, -1, -1});
    depth.assign(1, 1u);
  }
  // node index of every binary root
  std::vector<int> node_of(bin.size(), -1);
  for (size_t q = 0; q < roots.size(); ++q) node_of[(size_t)roots[q]] = (int)q;
  out.resize(slots.size());
  for (size_t q = 0; q < slots.size(); ++q) {
    BvhNode4 nd{};
    Box all{};
    int n = 0;
    for (int k = 0; k < 4; ++k) {
      if (slots[q][k] < 0) continue;
      all = n == 0 ? bin[(size_t)slots[q][k]].box : unite(all, bin[(size_t)slots[q][k]].box);
      ++n;
    }
    int e[3];
    for (int a = 0; a < 3; ++a) {
      nd.pmin[a] = all.mn[a];
      e[a] = quantum_exponent(all.mn[a], all.mx[a]);
      nd.exps |= (uint32_t)(e[a] + 127) << (8 * a);
    }
    int c = 0;
    for (int k = 0; k < 4; ++k) nd.ref[k] = kAbsent;
    for (int k = 0; k < 4; ++k) {
      if (slots[q][k] < 0) continue;
      const BinNode& ch = bin[(size_t)slots[q][k]];
      nd.ref[c] = ch.item >= 0 ? items[(size_t)ch.item].ref : (uint32_t)node_of[(size_t)slots[q][k]];
      for (int a = 0; a < 3; ++a) {
        const double s = std::ldexp(1.0, e[a]), base = (double)all.mn[a];
        double lo = std::floor(((double)ch.box.mn[a] - base) / s), hi = std::ceil(((double)ch.box.mx[a] - base) / s);
        if (base + lo * s > (double)ch.box.mn[a]) lo -= 1.0;  // rounding of the subtraction must not shrink the box
        if (base + hi * s < (double)ch.box.mx[a]) hi += 1.0;
        lo = std::min(std::max(lo, 0.0), 255.0); hi = std::min(std::max(hi, 0.0), 255.0);
        nd.qlo[a] |= (uint32_t)lo << (8 * c);
        nd.qhi[a] |= (uint32_t)hi << (8 * c);
      }
      ++c;
    }
    out[q] = nd;
  }
  uint32_t lv = 0;
  for (uint32_t d : depth) lv = std::max(lv, d);
  if (levels) *levels = lv;
  // traversal stack entries a ray can need below each node: (children that wait - 1) + the deepest need among them; an item brings its own
  if (stack_need) {
    std::vector<uint32_t> need(out.size(), 0u);
    for (size_t q = out.size(); q-- > 0;) {
      uint32_t waiting = 0, deepest = 0;
      for (int k = 0; k < 4; ++k) {
        if (slots[q][k] < 0) continue;
        const BinNode& ch = bin[(size_t)slots[q][k]];
        ++waiting;
        deepest = std::max(deepest, ch.item >= 0 ? items[(size_t)ch.item].need : need[(size_t)node_of[(size_t)slots[q][k]]]);
      }
      need[q] = waiting ? waiting - 1u + deepest : 0u;
    }
    *stack_need = std::max(1u, need[0]);
  }
  return (uint32_t)out.size();
}

}  // namespace rt
