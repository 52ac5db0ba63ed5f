// bvh8.cpp -- binned-SAH BVH2 -> greedy collapse to 8-wide -> octant slot assignment -> 8-bit quantisation.
//
// Replaces the opaque driver build behind cgpuCreateBlas/cgpuCreateTlas
// (/root/reference/src/cgpu/impl/CgpuVk.cpp:2561-2854, PREFER_FAST_TRACE at :2575).  Instances are flattened into
// world space before the build (DESIGN.md "Why one flat BVH"), so there is a single level.
//
// Conservativeness contract: a child's dequantised box always contains the (padded) boxes of everything below
// it, so the traversal kernel can never cull a triangle the exact Moeller-Trumbore test would accept.

#include "bvh8.h"
#include "gi_options.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <future>
#include <memory>
#include <queue>
#include <thread>

namespace gi {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() { for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
  void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.0f) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Node2 {
  Box box;
  uint32_t left = 0, right = 0; // children (internal)
  uint32_t first = 0, count = 0; // leaf range in refs[]
  uint32_t total = 0;            // references below this node: refs[first, first + total)
};

constexpr int kBins = 16;
constexpr uint32_t kMaxLeaf = 3; // 3-bit unary triangle count in Node8::meta
constexpr float kMaxCoordinate = 1.0e18f; // bvh8.h "Inactive items"

// Cost-optimal collapse (Ylitie, Karras, Laine 2017, section 3.1, implemented from the paper): c[i-1] = C(n, i), the cheapest way to
// represent the subtree of BVH2 node n as at most i slots of an ancestor's 8-wide node; eff[i-1]: slots actually used (1 = the subtree is
// ONE slot -- a leaf slot if `leaf`, else an 8-wide node of its own); split[j-2]: how many of j >= 2 slots go to the left child when the
// subtree is spread over j slots (j = 8: the child list of the 8-wide node rooted at n)
struct Dp { float c[7]; uint8_t eff[7]; uint8_t split[7]; uint8_t leaf; };

struct Builder {
  const std::vector<TriRec>& tris;
  std::vector<Box> triBox;
  std::vector<float> centroid; // 3 per tri
  std::vector<uint32_t> refs;
  // Node storage is indexed deterministically -- the subtree over `count` references rooted at index b owns [b, b + 2*count - 1):
  // left child at b + 1, right child at b + 2*leftCount -- so that subtrees can be built by different threads without any
  // shared counter and the tree is identical whatever the thread count.
  std::vector<Node2> nodes;
  std::atomic<int> spareThreads{0};

  // (cutting the bottom 24 triangles of a subtree into full leaves of three by object-median splits -- nodes per ray 16.6 / 6.9 / 20.1 -> 16.0 / 6.8 / 20.0 on
  // C3 / C4 / C5 but triangles per ray 13.1 / 3.4 / 10.8 -> 18.5 / 5.1 / 22.4, traversal 8-20 % slower: SAH leaves win, r02)
  // (not zero-initialised: 44 B per BVH2 node) filled bottom-up by build() when leafSize == 1 (each thread completes its own subtrees)
  std::unique_ptr<Dp[]> dp; float cPrim = 0.5f;
  uint32_t maxLeaf = kMaxLeaf;  // references a leaf slot may hold (1 for the top tree over subtrees: every leaf slot is then exactly one item)
  uint32_t leafSize = kMaxLeaf; // the BVH2 stops splitting at this many references (1 for the cost-optimal collapse, which forms the leaves itself)
  const float* extBoxes = nullptr; size_t extCount = 0; // box mode (TLAS over instances, BLAS over pre-padded triangle boxes): 6 floats per item
  size_t itemCount() const { return extBoxes ? extCount : tris.size(); }
  // Items the tree does not hold (bvh8.h "Inactive items"): refs[0, activeCount) are the active ones after prepare(), `inactive` the rest in input order.
  std::vector<uint8_t> dead; std::vector<uint32_t> inactive; size_t activeCount = 0;
  explicit Builder(const std::vector<TriRec>& t, const float* boxes = nullptr, size_t boxCount = 0) : tris(t), extBoxes(boxes), extCount(boxCount)
  {
    int threads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("GATLING_BUILD_THREADS")) threads = atoi(e);
    threads = std::min(std::max(threads, 1), 32); // several ranks build on one host: stay modest
    spareThreads = threads - 1;
  }

  void prepare()
  {
    size_t n = itemCount();
    triBox.resize(n); centroid.resize(3 * n); refs.resize(n); dead.assign(n, 0);
    const int workers = (n > (1u << 16)) ? spareThreads.load() + 1 : 1;
    std::vector<std::future<void>> jobs;
    for (int w = 1; w < workers;
        w++) jobs.push_back(std::async(std::launch::async, [this, n, w, workers] { prepareRange(n * w / workers, n * (w + 1) / workers); }));
    prepareRange(0, n / workers);
    for (auto& j : jobs) j.get();
    size_t k = 0;
    for (size_t i = 0; i < n; i++) { if (dead[i]) inactive.push_back((uint32_t)i); else refs[k++] = (uint32_t)i; }
    activeCount = k; refs.resize(k);
    nodes.assign(k ? 2 * k - 1 : 1, Node2{});
    if (leafSize == 1u) dp.reset(new Dp[nodes.size()]);
  }

  // a coordinate the build can work with: finite and small enough that extents, areas and the padded planes stay far from overflow
  static bool usable(float x) { return std::fabs(x) <= kMaxCoordinate; } // (false for NaN)

  void prepareRange(size_t begin, size_t end)
  {
    for (size_t i = begin; i < end; i++) {
      if (extBoxes) { // the caller's boxes are taken as they are (already padded)
        Box b;
            for (int a = 0; a < 3; a++) { b.lo[a] = extBoxes[6 * i + a]; b.hi[a] = extBoxes[6 * i + 3 + a]; centroid[3 * i + a] = 0.5f * (b.lo[a] + b.hi[a]); }
        triBox[i] = b;
        for (int a = 0; a < 3; a++) if (!usable(b.lo[a]) || !usable(b.hi[a]) || !(b.lo[a] <= b.hi[a])) dead[i] = 1;
        continue;
      }
      const TriRec& t = tris[i];
      Box b; b.reset();
      float p1[3], p2[3];
      for (int a = 0; a < 3; a++) { p1[a] = t.v0[a] + t.e1[a]; p2[a] = t.v0[a] + t.e2[a]; }
      for (int a = 0; a < 3; a++) if (!usable(t.v0[a]) || !usable(p1[a]) || !usable(p2[a])) dead[i] = 1;
      if (dead[i]) continue;
      b.grow(t.v0); b.grow(p1); b.grow(p2);
      for (int a = 0; a < 3; a++) {
        // pad: the MT test works on {v0, v0+e1, v0+e2} in exact arithmetic; cover float rounding generously
        float mag = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])) + (b.hi[a] - b.lo[a]);
        float pad = mag * 9.5367431640625e-7f /* 2^-20 */ + 1.0e-30f;
        b.lo[a] -= pad; b.hi[a] += pad;
        centroid[3 * i + a] = 0.5f * (b.lo[a] + b.hi[a]);
      }
      triBox[i] = b;
    }
  }

  uint32_t build(uint32_t first, uint32_t count, uint32_t idx = 0)
  {
    Box box; box.reset(); Box cb; cb.reset();
    for (uint32_t i = first; i < first + count; i++) { box.grow(triBox[refs[i]]); cb.grow(&centroid[3 * refs[i]]); }
    nodes[idx].box = box; nodes[idx].first = first; nodes[idx].total = count;
    if (count <= leafSize) { nodes[idx].count = count; if (dp) dpNode(idx); return idx; }
    // binned SAH over the longest centroid axes
    int bestAxis = -1; int bestSplit = -1; float bestCost = 3.0e38f;
    for (int a = 0; a < 3; a++) {
      float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.0f)) continue;
      Box bb[kBins]; uint32_t bc[kBins];
      for (int k = 0; k < kBins; k++) { bb[k].reset(); bc[k] = 0; }
      float scale = (float)kBins / ext;
      for (uint32_t i = first; i < first + count; i++) {
        uint32_t r = refs[i];
        int k = (int)((centroid[3 * r + a] - lo) * scale); k = std::min(std::max(k, 0), kBins - 1);
        bb[k].grow(triBox[r]); bc[k]++;
      }
      float rightArea[kBins]; uint32_t rightCount[kBins];
      Box acc; acc.reset(); uint32_t cnt = 0;
      for (int k = kBins - 1; k > 0; k--) { acc.grow(bb[k]); cnt += bc[k]; rightArea[k] = acc.area(); rightCount[k] = cnt; }
      acc.reset(); cnt = 0;
      for (int k = 0; k < kBins - 1; k++) {
        acc.grow(bb[k]); cnt += bc[k];
        if (cnt == 0 || rightCount[k + 1] == 0) continue;
        float cost = acc.area() * (float)cnt + rightArea[k + 1] * (float)rightCount[k + 1];
        if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
      }
    }
    uint32_t mid;
    if (bestAxis >= 0) {
      float lo = cb.lo[bestAxis], scale = (float)kBins / (cb.hi[bestAxis] - cb.lo[bestAxis]);
      auto it = std::partition(refs.begin() + first, refs.begin() + first + count, [&](uint32_t r) {
        int k = (int)((centroid[3 * r + bestAxis] - lo) * scale); k = std::min(std::max(k, 0), kBins - 1);
        return k <= bestSplit;
      });
      mid = (uint32_t)(it - refs.begin());
    } else {
      mid = first + count / 2; // all centroids coincide: split by index
    }
    if (mid == first || mid == first + count) mid = first + count / 2;
    const uint32_t leftCount = mid - first, leftIdx = idx + 1u, rightIdx = idx + 2u * leftCount;
    uint32_t l, r;
    if (count > (1u << 15) && spareThreads.fetch_sub(1) > 0) { // big subtree and a thread to spare: left half on another thread
      auto job = std::async(std::launch::async, [this, first, leftCount, leftIdx] { return build(first, leftCount, leftIdx); });
      r = build(mid, count - leftCount, rightIdx);
      l = job.get();
      spareThreads.fetch_add(1);
    } else {
      if (count > (1u << 15)) spareThreads.fetch_add(1); // undo the failed claim
      l = build(first, leftCount, leftIdx);
      r = build(mid, count - leftCount, rightIdx);
    }
    nodes[idx].left = l; nodes[idx].right = r;
    if (dp) dpNode(idx);
    return idx;
  }

  void dpNode(uint32_t ii)
  {
    const Node2& n = nodes[ii]; Dp& d = dp[ii];
    const float area = n.box.area();
    if (n.count > 0) { // a BVH2 leaf (one reference)
      for (int i = 0; i < 7; i++) { d.c[i] = area * cPrim * (float)n.count; d.eff[i] = 1; d.split[i] = 0; }
      d.leaf = 1;
      return;
    }
    const Dp& L = dp[n.left]; const Dp& R = dp[n.right];
    float dist[9];
    for (int j = 2; j <= 8; j++) {
      float best = 3.0e38f; int bk = 1;
      for (int k = std::max(1, j - 7); k <= std::min(7, j - 1); k++) { const float c = L.c[k - 1] + R.c[j - k - 1]; if (c < best) { best = c; bk = k; } }
      dist[j] = best; d.split[j - 2] = (uint8_t)bk;
    }
    const float cLeaf = n.total <= maxLeaf ? area * cPrim * (float)n.total : 3.0e38f;
    const float cInt = area + dist[8];
    // (the count is tested on its own: with overflowing areas cInt is +inf and the 3e38 stand-in would win)
    d.leaf = (n.total <= maxLeaf && cLeaf <= cInt) ? 1 : 0;
    d.c[0] = d.leaf ? cLeaf : cInt; d.eff[0] = 1;
    for (int i = 2; i <= 7; i++) {
      if (dist[i] < d.c[i - 2]) { d.c[i - 1] = dist[i]; d.eff[i - 1] = (uint8_t)i; }
      else { d.c[i - 1] = d.c[i - 2]; d.eff[i - 1] = d.eff[i - 2]; }
    }
  }
};

inline int exponentFor(float extent)
{
  // smallest e with 255 * 2^e >= extent
  if (!(extent > 0.0f)) return -126;
  int e; float m = std::frexp(extent / 255.0f, &e); // extent/255 = m * 2^e, m in [0.5,1)
  if (m == 0.5f) e -= 1;
  e = std::min(std::max(e, -126), 127);
  while (255.0f * std::ldexp(1.0f, e) < extent && e < 127) e++;
  return e;
}

} // namespace

// `itemRoots` (top mode, with `boxes`): every item is a subtree that already exists; its leaf slot becomes an INTERNAL child whose node is a copy of the item's
// root node (absolute child / triangle indices inside), so the result is one ordinary tree -- the traversal kernels never learn it was assembled from pieces.
static void buildCore(const std::vector<TriRec>& trisIn, const float* boxes, size_t boxCount, Bvh8& out, std::vector<uint32_t>* order,
    const Node8* itemRoots = nullptr)
{
  out.nodes.clear(); out.tris.clear(); out.maxDepth = 0;
  if (order) order->clear();
  const size_t itemCount = boxes ? boxCount : trisIn.size();
  out.activeTris = 0;
  auto emptyRoot = [&] {
    Node8 root; std::memset(&root, 0, sizeof(root));
    for (int a = 0; a < 3; a++) { root.e[a] = 127; for (int s = 0; s < 8; s++) { root.qlo[a][s] = 255; root.qhi[a][s] = 0; } }
    out.nodes.push_back(root); out.maxDepth = 1;
  };
  if (itemCount == 0) { emptyRoot(); return; }
  const bool timing = getenv("GATLING_BUILD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tA = now();
  Builder B(trisIn, boxes, boxCount);
  // Collapse rule.  1 (default): cost-optimal -- the BVH2 is built down to single references and a dynamic programme over it (Ylitie,
  // Karras, Laine 2017, section 3.1, implemented from the paper) chooses per BVH2 node whether its subtree becomes a leaf slot (<= 3
  // references), an 8-wide node, or part of its parent's child list, minimising  sum(area * (c_node | c_prim * references)).
  // 0: the round-1 rule (SAH leaves of <= 3, then greedily open the child with the largest area until 8 slots are used).
  int collapse = 1;
  collapse = (int)optionValue("bvh_collapse", collapse);
  // a triangle test costs about half a node test (~110 vs ~214 VALU instructions); measured flat between 0.2 and 0.5 (profiles/r02j_bvh_collapse.txt)
  float cPrim = 0.5f;
  if (itemRoots) { collapse = 1; cPrim = 1.0f; B.maxLeaf = 1u; } // an item costs (at least) a node visit; one item per leaf slot
  if (collapse == 1) { B.leafSize = 1; B.cPrim = cPrim; }
  B.prepare();
  // inactive items (bvh8.h) take no part in the tree; they keep their place in the numbering and sit, unreferenced, behind the leaf-ordered items
  auto appendInactive = [&] {
    if (itemRoots) return;
    for (uint32_t ref : B.inactive) {
      if (order) { order->push_back(ref); continue; }
      TriRec t = trisIn[ref]; t.origId = ref; out.tris.push_back(t);
    }
  };
  out.activeTris = (uint32_t)B.activeCount;
  if (B.activeCount == 0) { emptyRoot(); appendInactive(); return; }
  const double tB = now();
  uint32_t root2 = B.build(0, (uint32_t)B.activeCount, 0u);
  const double tC = now();

  const Dp* dp = B.dp.get();
  struct Child { uint32_t n2; bool leaf; };
  // the child list of the 8-wide node rooted at BVH2 node `root` under the optimal collapse
  auto gatherOptimal = [&](uint32_t root, Child* ch) {
    int n = 0;
    struct Todo { uint32_t n2; int slots; bool expand; };
    Todo todo[32]; int tp = 0;
    todo[tp++] = {root, 8, true};
    while (tp > 0) {
      Todo t = todo[--tp];
      const Node2& nd = B.nodes[t.n2];
      int j = t.slots;
      if (!t.expand) { j = dp[t.n2].eff[t.slots - 1]; if (j == 1) { ch[n++] = {t.n2, dp[t.n2].leaf != 0}; continue; } }
      const int k = dp[t.n2].split[j - 2];
      todo[tp++] = {nd.right, j - k, false};
      todo[tp++] = {nd.left, k, false};
    }
    return n;
  };

  // ---- emit, one breadth-first level at a time (node and triangle order = the order a queue would give; the levels' nodes are planned
  // and written in parallel, a prefix sum in between hands out child and triangle indices)
  struct Pend { uint32_t n2; uint32_t n8; };
  struct Plan { uint32_t ch[8]; int8_t childInSlot[8]; uint8_t leafMask; uint8_t n; uint32_t internal, tris; uint32_t childBase, triBase; Box nb; };
  int workers = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("GATLING_BUILD_THREADS")) workers = atoi(e);
  workers = std::min(std::max(workers, 1), 32);
  auto parallelFor = [&](size_t m, auto&& fn) {
    const int w = m >= 2048 ? workers : 1;
    if (w == 1) { for (size_t i = 0; i < m; i++) fn(i); return; }
    std::vector<std::future<void>> jobs;
    for (int t = 1; t < w; t++) jobs.push_back(std::async(std::launch::async, [&, t] { for (size_t i = m * t / w; i < m * (t + 1) / w; i++) fn(i); }));
    for (size_t i = 0; i < m / w; i++) fn(i);
    for (auto& j : jobs) j.get();
  };
  std::vector<Pend> level{{root2, 0u}}, next;
  std::vector<Plan> plans;
  out.nodes.resize(1);
  size_t triCount = 0;
  while (!level.empty()) {
    out.maxDepth++;
    const size_t m = level.size();
    plans.resize(m);
    parallelFor(m, [&](size_t li) {
      Plan& P = plans[li]; const uint32_t n2 = level[li].n2;
      // --- gather up to 8 children
      uint32_t* ch = P.ch; bool chLeaf[8]; int n = 0;
      const Node2& r = B.nodes[n2];
      if (itemRoots && r.count > 0) { P.n = 0; P.internal = 0; P.tris = 0; P.leafMask = 0; return; } // a copied item root: written below, nothing to plan
      // root that is itself a leaf (level 0 only: below the root the DP's own choice stands -- a subtree of <= 3 references it made an 8-wide
      // node of is cheaper that way than as a node holding one 3-reference leaf slot, which is what this shortcut would emit)
      if (r.count > 0 || (collapse == 1 && r.total <= B.maxLeaf && out.maxDepth == 1u)) { ch[0] = n2; chLeaf[0] = true; n = 1; }
      else if (collapse == 1) {
        Child cs[8]; n = gatherOptimal(n2, cs);
        for (int i = 0; i < n; i++) { ch[i] = cs[i].n2; chLeaf[i] = cs[i].leaf; }
      } else { // by opening the largest internal child
        ch[n++] = r.left; ch[n++] = r.right;
        while (n < 8) {
          int best = -1; float bestArea = -1.0f;
          for (int i = 0; i < n; i++) {
            const Node2& c = B.nodes[ch[i]];
            if (c.count == 0 && c.box.area() > bestArea) { bestArea = c.box.area(); best = i; }
          }
          if (best < 0) break;
          uint32_t opened = ch[best];
          ch[best] = B.nodes[opened].left; ch[n++] = B.nodes[opened].right;
        }
        for (int i = 0; i < n; i++) chLeaf[i] = B.nodes[ch[i]].count > 0;
      }
      // --- node box + slot assignment (greedy max of centroid projection on the slot's octant direction)
      Box nb; nb.reset();
      for (int i = 0; i < n; i++) nb.grow(B.nodes[ch[i]].box);
      float center[3]; for (int a = 0; a < 3; a++) center[a] = 0.5f * (nb.lo[a] + nb.hi[a]);
      float cost[8][8];
      for (int i = 0; i < n; i++) {
        const Box& b = B.nodes[ch[i]].box;
        float d[3]; for (int a = 0; a < 3; a++) d[a] = 0.5f * (b.lo[a] + b.hi[a]) - center[a];
        for (int s = 0; s < 8; s++) cost[i][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
      }
      int slotOf[8]; bool slotUsed[8] = {false}; bool childDone[8] = {false};
      for (int k = 0; k < n; k++) {
        int bi = -1, bs = -1; float bc = -3.0e38f;
        for (int i = 0; i < n; i++) if (!childDone[i]) for (int s = 0; s < 8; s++) if (!slotUsed[s] && cost[i][s] > bc) { bc = cost[i][s]; bi = i; bs = s; }
        slotOf[bi] = bs; slotUsed[bs] = true; childDone[bi] = true;
      }
      for (int s = 0; s < 8; s++) P.childInSlot[s] = -1;
      P.leafMask = 0; P.internal = 0; P.tris = 0; P.n = (uint8_t)n; P.nb = nb;
      for (int i = 0; i < n; i++) {
        P.childInSlot[slotOf[i]] = (int8_t)i;
        if (chLeaf[i] && !itemRoots) { P.leafMask |= (uint8_t)(1u << i); P.tris += B.nodes[ch[i]].total; } else P.internal++;
      }
    });
    // --- indices: internal children and triangles in level order, slot order within a node
    size_t nodeBase = out.nodes.size(), nodeEnd = nodeBase, triEnd = triCount;
    for (size_t li = 0; li < m; li++) { plans[li].childBase = (uint32_t)nodeEnd; plans[li].triBase = (uint32_t)triEnd; nodeEnd += plans[li].internal;
        triEnd += plans[li].tris; }
    out.nodes.resize(nodeEnd);
    if (order) order->resize(triEnd); else out.tris.resize(triEnd);
    next.resize(nodeEnd - nodeBase);
    parallelFor(m, [&](size_t li) {
      const Plan& P = plans[li];
      if (itemRoots && B.nodes[level[li].n2].count > 0) { out.nodes[level[li].n8] = itemRoots[B.refs[B.nodes[level[li].n2].first]]; return; }
      Node8 node; std::memset(&node, 0, sizeof(node));
      int ex[3]; float scale[3];
      for (int a = 0; a < 3; a++) { node.p[a] = P.nb.lo[a]; ex[a] = exponentFor(P.nb.hi[a] - P.nb.lo[a]); node.e[a] = (uint8_t)(ex[a] + 127);
          scale[a] = std::ldexp(1.0f, ex[a]); }
      node.childBase = P.childBase;
      node.triBase = P.triBase;
      uint32_t triOffset = 0, childIdx = P.childBase;
      for (int s = 0; s < 8; s++) {
        const int i = P.childInSlot[s];
        if (i < 0) { for (int a = 0; a < 3; a++) { node.qlo[a][s] = 255; node.qhi[a][s] = 0; } continue; }
        const Node2& c = B.nodes[P.ch[i]];
        for (int a = 0; a < 3; a++) {
          int lo = (int)std::floor(((double)c.box.lo[a] - (double)node.p[a]) / (double)scale[a]);
          int hi = (int)std::ceil(((double)c.box.hi[a] - (double)node.p[a]) / (double)scale[a]);
          lo = std::min(std::max(lo, 0), 255); hi = std::min(std::max(hi, 0), 255);
          while (lo > 0 && node.p[a] + (float)lo * scale[a] > c.box.lo[a]) lo--;
          while (hi < 255 && node.p[a] + (float)hi * scale[a] < c.box.hi[a]) hi++;
          node.qlo[a][s] = (uint8_t)lo; node.qhi[a][s] = (uint8_t)hi;
        }
        if (P.leafMask & (1u << i)) { // leaf slot: unary count in the high 3 bits, triangle offset in the low 5
          const uint32_t cnt = c.total;
          uint32_t unary = (1u << cnt) - 1u;
          node.meta[s] = (uint8_t)((unary << 5) | triOffset);
          for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t ref = B.refs[c.first + k], dst = P.triBase + triOffset + k;
            if (order) { (*order)[dst] = ref; continue; }
            TriRec t = B.tris[ref];
            t.origId = ref;
            out.tris[dst] = t;
          }
          triOffset += cnt;
        } else {
          node.imask |= (uint8_t)(1u << s);
          node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
          next[childIdx - nodeBase] = Pend{P.ch[i], childIdx};
          childIdx++;
        }
      }
      out.nodes[level[li].n8] = node;
    });
    triCount = triEnd;
    level.swap(next);
  }
  appendInactive();
  if (timing) fprintf(stderr, "[gatling_gi] bvh8: prepare %.0f ms, bvh2 %.0f ms, collapse+quantise %.0f ms (%zu items, %zu nodes)\n", tB - tA, tC - tB,
      now() - tC, itemCount, out.nodes.size());
}

void buildBvh8(const std::vector<TriRec>& trisIn, Bvh8& out) { buildCore(trisIn, nullptr, 0, out, nullptr); }

void buildTopBvh8(const float* boxes, size_t count, const Node8* itemRoots, Bvh8& out)
{
  static const std::vector<TriRec> none;
  std::vector<uint32_t> order; // (unused: no leaf slot survives)
  buildCore(none, boxes, count, out, &order, itemRoots);
}

void buildBvh8Boxes(const float* boxes, size_t count, Bvh8& out, std::vector<uint32_t>& order)
{
  static const std::vector<TriRec> none;
  buildCore(none, boxes, count, out, &order);
}

} // namespace gi
