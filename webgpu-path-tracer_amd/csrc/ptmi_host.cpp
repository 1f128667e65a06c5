// ptmi_host.cpp — host-side natives of libptmi.so (no GPU needed).
//
// ptmi_build_bvh: the reference's BVH construction (lib/BVH/bvhNode.js:21-101: top-down, split the
// longest axis of the node's box at the median after a STABLE sort on bbox.min[axis]; one primitive
// per leaf) and its pre-order flattening (lib/BVH/bvhBuilder.js:37-54), in doubles like the JS, rounded
// to f32 on the final store like `new Float32Array(...)` (lib/scene.js:304).  Output is byte-identical
// to the reference's for the same boxes (tests/test_host_buffers.py checks it against goldens).
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ptmi.h"

namespace {
// Math.min / Math.max (lib/BVH/AABB.js:8-28) on finite values: -0 < +0 whatever the argument order
inline double js_min(double a, double b) { return (a < b || (a == b && (a != 0.0 || std::signbit(a)))) ? a : b; }
inline double js_max(double a, double b) { return (a > b || (a == b && (a != 0.0 || !std::signbit(a)))) ? a : b; }


// std::stable_sort with the upper levels of the merge tree forked: a stable sort's result is unique for a given order, so
// halves sorted on their own threads + std::inplace_merge (stable) give exactly what one stable_sort call gives.
template <class Cmp>
void par_stable_sort(int64_t* first, int64_t* last, Cmp cmp, int forks) {
  const int64_t n = last - first;
  if (forks <= 0 || n < 32768) {
    std::stable_sort(first, last, cmp);
    return;
  }
  int64_t* mid = first + n / 2;
  std::future<void> other;
  bool forked = false;
  try {
    other = std::async(std::launch::async, [=] { par_stable_sort(first, mid, cmp, forks - 1); });
    forked = true;
  } catch (...) {
  }
  if (!forked) par_stable_sort(first, mid, cmp, forks - 1);
  par_stable_sort(mid, last, cmp, forks - 1);
  if (forked) other.get();
  std::inplace_merge(first, mid, last, cmp);
}

struct Builder {
  const double* bmin;
  const double* bmax;
  int prim_type;
  float* nodes;
  int64_t* order;
  std::vector<int64_t> left, right;
  int par_depth = 0;  // subtrees above this depth are built by their own threads

  // Pre-order ids need no shared counter: every leaf holds one primitive, so a subtree over k primitives has 2k-1 nodes —
  // the left child of node `id` is id+1, the right child id + 2*(primitives on the left).  Subtrees touch disjoint rows
  // of `nodes` and disjoint ranges of `order`, which is what lets the upper levels fork.
  void gen(int64_t start, int64_t end, int64_t id, int depth) {
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};  // new AABB() (AABB.js:2-5)
    for (int64_t i = start; i <= end; i++) {
      const double* a = bmin + 3 * order[i];
      const double* b = bmax + 3 * order[i];
      for (int k = 0; k < 3; k++) {
        lo[k] = js_min(a[k], lo[k]);
        hi[k] = js_max(b[k], hi[k]);
      }
    }
    double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    int axis = 0;
    if (ext[1] > ext[0]) axis = 1;
    if (ext[2] > ext[axis]) axis = 2;
    float* row = nodes + 12 * id;
    row[0] = (float)lo[0], row[1] = (float)lo[1], row[2] = (float)lo[2];
    row[4] = (float)hi[0], row[5] = (float)hi[1], row[6] = (float)hi[2];
    row[10] = -1.0f;
    const int64_t span = end - start;
    if (span <= 0) {  // leaf (bvhNode.js:47-53)
      row[3] = -1.0f;
      row[7] = (float)prim_type;
      row[8] = (float)start;
      row[9] = (float)(end - start + 1);
      row[11] = 0.0f;
      left[id] = right[id] = -1;
      return;
    }
    int64_t* first = order + start;
    const double* keys = bmin;
    // the levels that cannot fork into enough subtrees yet spend their spare threads inside the sort
    par_stable_sort(first, first + span + 1, [keys, axis](int64_t a, int64_t b) { return keys[3 * a + axis] < keys[3 * b + axis]; }, par_depth - depth);
    const int64_t mid = start + span / 2;
    const int64_t l = id + 1, r = id + 2 * (mid - start + 1);
    left[id] = l;
    right[id] = r;
    row[3] = (float)r;
    row[7] = row[8] = row[9] = -1.0f;
    row[11] = (float)axis;
    if (depth < par_depth && span >= 4096) {
      std::future<void> other;
      bool forked = false;
      try {
        other = std::async(std::launch::async, [this, start, mid, l, depth] { gen(start, mid, l, depth + 1); });
        forked = true;
      } catch (...) {  // no thread to be had: build it here
      }
      if (!forked) gen(start, mid, l, depth + 1);
      gen(mid + 1, end, r, depth + 1);
      if (forked) other.get();
    } else {
      gen(start, mid, l, depth + 1);
      gen(mid + 1, end, r, depth + 1);
    }
  }
};

}  // namespace

extern "C" int ptmi_build_bvh(size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out, int64_t* order_out) {
  if (n_prims == 0) return PTMI_OK;
  if (!bmin || !bmax || !nodes_out || !order_out) return PTMI_ERR_INVALID_ARG;
  if (n_prims > (size_t)1 << 27) return PTMI_ERR_UNSUPPORTED;
  Builder b;
  b.bmin = bmin;
  b.bmax = bmax;
  b.prim_type = prim_type;
  b.nodes = nodes_out;
  b.order = order_out;
  const int64_t nn = 2 * (int64_t)n_prims - 1;
  try {
    b.left.assign((size_t)nn, -1);
    b.right.assign((size_t)nn, -1);
  } catch (...) {
    return PTMI_ERR_NO_MEMORY;
  }
  for (size_t i = 0; i < n_prims; i++) order_out[i] = (int64_t)i;
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = getenv("PTMI_BUILD_THREADS")) hw = (unsigned)std::max(1, atoi(e));
  hw = std::min(hw ? hw : 1u, 32u);
  while ((1u << b.par_depth) < hw) b.par_depth++;  // 2^par_depth subtrees in flight
  try {
    b.gen(0, (int64_t)n_prims - 1, 0, 0);
  } catch (...) {
    return PTMI_ERR_NO_MEMORY;
  }
  // populate_links (bvhNode.js:76-93): skip link = node to visit when this subtree is done or missed
  std::vector<std::pair<int64_t, int64_t>> st;
  st.emplace_back(0, -1);
  while (!st.empty()) {
    auto [n, nxt] = st.back();
    st.pop_back();
    nodes_out[12 * n + 10] = (float)nxt;
    if (b.left[n] >= 0) {
      st.emplace_back(b.right[n], nxt);
      st.emplace_back(b.left[n], b.right[n]);
    }
  }
  return PTMI_OK;
}


// ---- binned-SAH builder, opt-in --------------------------------------------------------------------------------------
// Restates the reference's SECOND builder, BVH.generate_bvh_heirarchy_SAH (lib/BVH/bvhNode.js:108-283: 8 bins per axis,
// 7 candidate planes, leaf when the best split costs no less than the node; leaves hold any number of primitives).  The
// reference never calls it (create_bvh uses the median split), so the renderer's default stays the median builder; this
// one is for users who want the ~2x cheaper traversal the reference's own benchmarks.txt shows for SAH trees.  Output is
// byte-identical to running that method through the reference's flattening (tests/golden/*sah*).
namespace {

struct Box {
  double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};  // new AABB()
  void merge(const double* a, const double* b) {                      // AABB.merge: min(a.min, this.min) ...
    for (int k = 0; k < 3; k++) {
      lo[k] = js_min(a[k], lo[k]);
      hi[k] = js_max(b[k], hi[k]);
    }
  }
  void merge(const Box& o) { merge(o.lo, o.hi); }
  double area() const {  // AABB.surface_area
    const double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
    return e0 * e1 + e1 * e2 + e2 * e0;
  }
};

struct SahNode {
  Box box;
  int64_t start = 0, end = 0, left = -1, right = -1;
  int axis = 0;
  bool leaf = false;
};

struct SahBuilder {
  const double* bmin;
  const double* bmax;
  int64_t* order;
  std::vector<SahNode> nodes;

  double centroid(int64_t prim, int a) const { return (bmin[3 * prim + a] + bmax[3 * prim + a]) / 2; }

  // BVH.FindBestSplitPlane (bvhNode.js:221-283)
  void best_split(int64_t start, int64_t end, int& axis, double& split_pos, double& best_cost) const {
    best_cost = 1e30;
    axis = 0;
    split_pos = 0;
    constexpr int BINS = 8;
    for (int a = 0; a < 3; a++) {
      double bounds_min = 1e30, bounds_max = -1e30;
      for (int64_t i = start; i <= end; i++) {
        const double c = centroid(order[i], a);
        bounds_min = std::min(bounds_min, c);
        bounds_max = std::max(bounds_max, c);
      }
      if (bounds_min == bounds_max) continue;
      Box bin_box[BINS];
      double bin_count[BINS] = {0, 0, 0, 0, 0, 0, 0, 0};
      double scale = BINS / (bounds_max - bounds_min);
      for (int64_t i = start; i <= end; i++) {
        const int64_t p = order[i];
        const double f = std::floor((centroid(p, a) - bounds_min) * scale);
        const int b = (int)std::min((double)(BINS - 1), f);
        bin_count[b] += 1;
        bin_box[b].merge(bmin + 3 * p, bmax + 3 * p);
      }
      double left_area[BINS - 1], right_area[BINS - 1], left_count[BINS - 1], right_count[BINS - 1];
      Box lbox, rbox;
      double lsum = 0, rsum = 0;
      for (int i = 0; i < BINS - 1; i++) {
        lsum += bin_count[i];
        left_count[i] = lsum;
        lbox.merge(bin_box[i]);
        left_area[i] = lbox.area();
        rsum += bin_count[BINS - 1 - i];
        right_count[BINS - 2 - i] = rsum;
        rbox.merge(bin_box[BINS - 1 - i]);
        right_area[BINS - 2 - i] = rbox.area();
      }
      scale = (bounds_max - bounds_min) / BINS;
      for (int i = 0; i < BINS - 1; i++) {
        const double cost = left_count[i] * left_area[i] + right_count[i] * right_area[i];
        if (cost < best_cost) {
          axis = a;
          split_pos = bounds_min + scale * (i + 1);
          best_cost = cost;
        }
      }
    }
  }

  // generate_bvh_heirarchy_SAH (bvhNode.js:108-202); an explicit stack as there: SAH trees can be very deep.  Subtrees over disjoint ranges of `order` are
  // independent, so the upper levels hand their right halves to threads of their own (round 5: 4.4 s -> a few hundred ms at 871 k boxes): node records come out of
  // a pre-sized array through an atomic counter — their numbers depend on the timing, the tree and its pre-order flattening do not.
  std::atomic<int64_t> n_alloc{1};
  std::atomic<int> running{1};  // threads at work on subtrees (SAH splits are lopsided: a fixed fork depth leaves most threads idle, so a subtree forks whenever one is free)
  int max_threads = 1;
  void build_subtree(int64_t root) {
    std::vector<int64_t> todo{root};
    std::vector<std::future<void>> kids;
    while (!todo.empty()) {
      const int64_t id = todo.back();
      todo.pop_back();
      const int64_t start = nodes[id].start, end = nodes[id].end;
      Box box;
      for (int64_t i = start; i <= end; i++) box.merge(bmin + 3 * order[i], bmax + 3 * order[i]);
      nodes[id].box = box;  // (the reference re-merges it from the children on the way back: the same union)
      const double parent_cost = (double)(end - start + 1) * box.area();
      int axis;
      double split_pos, best_cost;
      best_split(start, end, axis, split_pos, best_cost);
      if (best_cost >= parent_cost) {
        nodes[id].leaf = true;
        continue;
      }
      const double* keys = bmin;
      // (the big ranges near the root also sort on the threads that have nothing else to do yet)
      int sort_forks = 0;
      for (int spare = max_threads - running.load(); end - start >= 65536 && (1 << sort_forks) <= spare; sort_forks++) {
      }
      par_stable_sort(order + start, order + end + 1, [keys, axis](int64_t a, int64_t b) { return keys[3 * a + axis] < keys[3 * b + axis]; }, sort_forks);
      int64_t split = start;
      while (split < end - 1) {
        if (centroid(order[split], axis) <= split_pos) split++;
        else break;
      }
      const int64_t l = n_alloc.fetch_add(2), r = l + 1;
      nodes[l].start = start, nodes[l].end = split;
      nodes[r].start = split + 1, nodes[r].end = end;
      nodes[id].left = l, nodes[id].right = r, nodes[id].axis = axis;
      bool forked = false;
      if (end - split >= 8192 && split - start >= 8192 && running.load() < max_threads) {
        running.fetch_add(1);
        try {
          kids.push_back(std::async(std::launch::async, [this, r] {
            build_subtree(r);
            running.fetch_sub(1);
          }));
          forked = true;
        } catch (...) {  // no thread to be had: the right half is built here too
          running.fetch_sub(1);
        }
      }
      if (!forked) todo.push_back(r);
      todo.push_back(l);
    }
    for (auto& k : kids) k.get();
  }
  void build(int64_t n, int threads) {
    max_threads = threads;
    running = 1;
    nodes.assign((size_t)(2 * n - 1), SahNode());  // every leaf holds at least one primitive: at most 2n - 1 nodes
    n_alloc = 1;
    nodes[0].start = 0;
    nodes[0].end = n - 1;
    build_subtree(0);
    nodes.resize((size_t)n_alloc.load());
  }
};

}  // namespace

extern "C" int ptmi_build_bvh_sah(size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out, int64_t* order_out,
                                  size_t* n_nodes_out) {
  if (!n_nodes_out) return PTMI_ERR_INVALID_ARG;
  *n_nodes_out = 0;
  if (n_prims == 0) return PTMI_OK;
  if (!bmin || !bmax || !nodes_out || !order_out) return PTMI_ERR_INVALID_ARG;
  if (n_prims > (size_t)1 << 27) return PTMI_ERR_UNSUPPORTED;
  try {
    SahBuilder b;
    b.bmin = bmin, b.bmax = bmax, b.order = order_out;
    for (size_t i = 0; i < n_prims; i++) order_out[i] = (int64_t)i;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("PTMI_BUILD_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    hw = std::min(hw ? hw : 1u, 32u);
    b.build((int64_t)n_prims, (int)hw);
    // flattenBVH (bvhBuilder.js:37-54): pre-order ids; populate_links (bvhNode.js:76-93): the skip link
    const size_t nn = b.nodes.size();
    std::vector<int64_t> flat_id(nn, -1);
    struct Item {
      int64_t node, next;
    };
    std::vector<Item> st{{0, -1}};
    std::vector<int64_t> pre;  // tree node of each flat row, in order
    pre.reserve(nn);
    std::vector<int64_t> next_of(nn, -1);
    while (!st.empty()) {
      const Item it = st.back();
      st.pop_back();
      flat_id[it.node] = (int64_t)pre.size();
      pre.push_back(it.node);
      next_of[it.node] = it.next;
      const SahNode& nd = b.nodes[it.node];
      if (!nd.leaf) {
        st.push_back({nd.right, it.next});
        st.push_back({nd.left, nd.right});
      }
    }
    for (size_t k = 0; k < pre.size(); k++) {
      const SahNode& nd = b.nodes[pre[k]];
      float* row = nodes_out + 12 * k;
      row[0] = (float)nd.box.lo[0], row[1] = (float)nd.box.lo[1], row[2] = (float)nd.box.lo[2];
      row[4] = (float)nd.box.hi[0], row[5] = (float)nd.box.hi[1], row[6] = (float)nd.box.hi[2];
      row[10] = next_of[pre[k]] < 0 ? -1.0f : (float)flat_id[next_of[pre[k]]];
      if (nd.leaf) {
        row[3] = -1.0f;
        row[7] = (float)prim_type;
        row[8] = (float)nd.start;
        row[9] = (float)(nd.end - nd.start + 1);
        row[11] = 0.0f;
      } else {
        row[3] = (float)flat_id[nd.right];
        row[7] = row[8] = row[9] = -1.0f;
        row[11] = (float)nd.axis;
      }
    }
    *n_nodes_out = pre.size();
  } catch (...) {
    return PTMI_ERR_NO_MEMORY;
  }
  return PTMI_OK;
}


// ---- OBJ parsing with the reference's grammar (lib/primitives/objReader.js:10-68) ---------------------------------
namespace {

bool js_space(unsigned char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\v' || ch == '\f'; }

// JavaScript Number(token) for the tokens an OBJ line can produce: "" -> 0, decimal literals with optional sign /
// fraction / exponent, [+-]Infinity, 0x / 0o / 0b integers; anything else -> NaN.  Decimal literals — every token of a
// well-formed file — are checked against the strict grammar in place and converted by std::from_chars (correctly rounded like
// V8's, no locale, no copy); the rare other forms take the slow path.
double js_number_slow(const char* b, const char* e) {
  std::string t(b, e);
  const char* p = t.c_str();
  bool neg = false;
  if (t.size() > 2 && p[0] == '0' && (p[1] == 'x' || p[1] == 'X' || p[1] == 'o' || p[1] == 'O' || p[1] == 'b' || p[1] == 'B')) {
    int base = (p[1] == 'x' || p[1] == 'X') ? 16 : (p[1] == 'o' || p[1] == 'O') ? 8 : 2;
    char* end = nullptr;
    unsigned long long v = strtoull(p + 2, &end, base);
    return (*end == 0) ? (double)v : NAN;
  }
  const char* q = p;
  if (*q == '+' || *q == '-') neg = (*q == '-'), q++;
  if (strcmp(q, "Infinity") == 0) return neg ? -INFINITY : INFINITY;
  // strict decimal: digits [. digits] [e[+-]digits], at least one digit in the mantissa
  const char* r = q;
  int digits = 0;
  while (*r >= '0' && *r <= '9') r++, digits++;
  if (*r == '.') {
    r++;
    while (*r >= '0' && *r <= '9') r++, digits++;
  }
  if (digits == 0) return NAN;
  if (*r == 'e' || *r == 'E') {
    const char* x = r + 1;
    if (*x == '+' || *x == '-') x++;
    if (!(*x >= '0' && *x <= '9')) return NAN;
    while (*x >= '0' && *x <= '9') x++;
    r = x;
  }
  if (*r != 0) return NAN;
  return strtod(p, nullptr);
}
double js_number(const char* b, const char* e) {
  while (b < e && js_space((unsigned char)*b)) b++;
  while (e > b && js_space((unsigned char)e[-1])) e--;
  if (b == e) return 0.0;
  const char* q = b;
  if (*q == '+' || *q == '-') q++;
  const char* r = q;
  int digits = 0;
  while (r < e && *r >= '0' && *r <= '9') r++, digits++;
  if (r < e && *r == '.') {
    r++;
    while (r < e && *r >= '0' && *r <= '9') r++, digits++;
  }
  if (digits == 0) return js_number_slow(b, e);  // Infinity, or not a number
  if (r < e && (*r == 'e' || *r == 'E')) {
    const char* x = r + 1;
    if (x < e && (*x == '+' || *x == '-')) x++;
    if (!(x < e && *x >= '0' && *x <= '9')) return js_number_slow(b, e);
    while (x < e && *x >= '0' && *x <= '9') x++;
    r = x;
  }
  if (r != e) return js_number_slow(b, e);  // 0x.., trailing garbage
  double v = 0.0;
  const auto res = std::from_chars(*b == '+' ? b + 1 : b, e, v, std::chars_format::general);
  if (res.ec != std::errc() || res.ptr != e) return js_number_slow(b, e);  // (out of range: strtod's +-inf / 0, JavaScript's answer too)
  return v;
}

struct ObjRows {
  std::vector<double> data;      // all numbers of all rows, concatenated
  std::vector<uint32_t> offset;  // row i = data[offset[i] .. offset[i+1])
  ObjRows() { offset.push_back(0); }
  void add_row_split_on_space(const char* b, const char* e) {  // line.split(" ").slice(1).map(Number)
    const char* p = b;
    bool first = true;
    while (true) {
      const char* q = p;
      while (q < e && *q != ' ') q++;
      if (!first) data.push_back(js_number(p, q));
      first = false;
      if (q >= e) break;
      p = q + 1;
    }
    offset.push_back((uint32_t)data.size());
  }
  size_t rows() const { return offset.size() - 1; }
};

}  // namespace

extern "C" void ptmi_free(void* p) { free(p); }

// One run of whole lines [p, end) of the file: the rows and index tokens it contains, in order.
struct ObjChunk {
  ObjRows V, N;
  std::vector<double> vidx, nidx;  // doubles: an index token may be NaN
};
static void obj_parse_lines(const char* p, const char* end, ObjChunk& c) {
  // One allocation per array instead of a dozen doublings (each a fresh mapping, a copy and page faults under the process's one mm lock — with several
  // threads at it the parser spent more time in the kernel than parsing): a number takes >= 2 characters of the text, rows >= 8; untouched pages cost nothing.
  const size_t len = (size_t)(end - p);
  c.V.data.reserve(len / 6), c.N.data.reserve(len / 6), c.V.offset.reserve(len / 16 + 2), c.N.offset.reserve(len / 16 + 2);
  c.vidx.reserve(len / 6), c.nidx.reserve(len / 6);
  while (p <= end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    const char *b = p, *e = le;  // line.trim()
    while (b < e && js_space((unsigned char)*b)) b++;
    while (e > b && js_space((unsigned char)e[-1])) e--;
    if (b < e && *b != '#') {
      if (e - b >= 2 && b[0] == 'v' && b[1] == ' ') {
        c.V.add_row_split_on_space(b, e);
      } else if (e - b >= 2 && b[0] == 'f' && b[1] == ' ') {  // split(/[\s/]+/).slice(1); i % 3 == 0 -> vertex, == 2 -> normal
        const char* q = b;
        int tok = -1;  // token 0 is "f"
        while (q < e) {
          const char* t0 = q;
          while (q < e && !js_space((unsigned char)*q) && *q != '/') q++;
          if (tok >= 0) {
            if (tok % 3 == 0) c.vidx.push_back(js_number(t0, q) - 1);
            else if (tok % 3 == 2) c.nidx.push_back(js_number(t0, q) - 1);
          }
          tok++;
          while (q < e && (js_space((unsigned char)*q) || *q == '/')) q++;
        }
      } else if (e - b >= 3 && b[0] == 'v' && b[1] == 'n' && b[2] == ' ') {
        c.N.add_row_split_on_space(b, e);
      }
    }
    if (!nl) break;
    p = nl + 1;
  }
}

// Runs fn(0) .. fn(n - 1), fn(0) on the calling thread and the others on threads of their own.  Nothing a worker throws leaves its thread (an exception escaping a
// std::thread is std::terminate: the host process, not a status code), every started thread is joined on every way out — also when starting one fails —, and the
// first failure is what the caller gets: false = some worker threw (out of memory, in practice).
template <class F>
static bool run_workers(unsigned n, F fn) {
  std::vector<char> failed(n, 0);
  auto guarded = [&](unsigned k) {
    try {
      fn(k);
    } catch (...) {
      failed[k] = 1;
    }
  };
  struct Joiner {
    std::vector<std::thread> th;
    ~Joiner() {
      for (auto& t : th)
        if (t.joinable()) t.join();
    }
  } j;
  bool ok = true;
  try {
    j.th.reserve(n);
    for (unsigned k = 1; k < n; k++) j.th.emplace_back(guarded, k);
  } catch (...) {  // no thread to be had: the chunks without one are done here
    for (unsigned k = (unsigned)j.th.size() + 1; k < n; k++) guarded(k);
  }
  guarded(0);
  for (auto& t : j.th) t.join();
  for (char f : failed) ok = ok && !f;
  return ok;
}

static unsigned host_threads(size_t work_items, size_t per_thread) {
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  if (const char* e = getenv("PTMI_BUILD_THREADS")) hw = (unsigned)std::max(1, atoi(e));
  return (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(hw, 32u), work_items / std::max<size_t>(per_thread, 1) + 1));
}

// The file is cut at line ends into one run of lines per host thread (lines are independent: indices are absolute row numbers), the runs'
// rows and index tokens are concatenated in file order, and the de-indexing is parallel over the index tokens.  PTMI_BUILD_THREADS limits
// the threads (1 = the sequential parser); the output is the same bytes whatever the number.
extern "C" int ptmi_obj_parse(const char* text, size_t len, float** vertices_out, size_t* n_vertices, float** normals_out, size_t* n_normals) {
  if (!text || !vertices_out || !n_vertices || !normals_out || !n_normals) return PTMI_ERR_INVALID_ARG;
  *vertices_out = *normals_out = nullptr;
  *n_vertices = *n_normals = 0;
  try {
    const char* end = text + len;
    const unsigned nt = host_threads(len, (size_t)1 << 20);
    std::vector<const char*> cut(nt + 1, end);
    cut[0] = text;
    for (unsigned k = 1; k < nt; k++) {  // the first line start at or after k/nt of the file
      const char* want = text + (len / nt) * k;
      if (want < cut[k - 1]) want = cut[k - 1];
      const char* nl = want < end ? (const char*)memchr(want, '\n', (size_t)(end - want)) : nullptr;
      cut[k] = nl ? nl + 1 : end;
    }
    std::vector<ObjChunk> chunks(nt);
    // (a run that ends just behind a '\n' sees one empty extra line at its end: ignored like every empty line)
    if (!run_workers(nt, [&](unsigned k) { obj_parse_lines(cut[k], cut[k + 1], chunks[k]); })) return PTMI_ERR_NO_MEMORY;
    ObjRows V, N;
    std::vector<double> vidx, nidx;
    if (nt == 1) {
      V = std::move(chunks[0].V), N = std::move(chunks[0].N), vidx = std::move(chunks[0].vidx), nidx = std::move(chunks[0].nidx);
    } else {
      auto merge_rows = [&](ObjRows& dst, ObjRows ObjChunk::*m) {
        size_t nd = 0, nr = 0;
        for (auto& c : chunks) nd += (c.*m).data.size(), nr += (c.*m).rows();
        dst.data.reserve(nd);
        dst.offset.reserve(nr + 1);
        for (auto& c : chunks) {
          const ObjRows& r = c.*m;
          const uint32_t base = (uint32_t)dst.data.size();
          dst.data.insert(dst.data.end(), r.data.begin(), r.data.end());
          for (size_t i = 1; i < r.offset.size(); i++) dst.offset.push_back(base + r.offset[i]);
        }
      };
      merge_rows(V, &ObjChunk::V);
      merge_rows(N, &ObjChunk::N);
      for (auto& c : chunks) vidx.insert(vidx.end(), c.vidx.begin(), c.vidx.end()), nidx.insert(nidx.end(), c.nidx.begin(), c.nidx.end());
    }
    auto flatten = [](const ObjRows& R, const std::vector<double>& idx, float** out, size_t* n) -> int {
      // array[v] with v not a valid index is `undefined`; .flat(1) keeps it as one element -> NaN in the Float32Array
      auto valid = [&](double d) { return d >= 0 && d == std::floor(d) && d < (double)R.rows(); };
      const unsigned nt = host_threads(idx.size(), (size_t)1 << 18);
      std::vector<size_t> start(nt + 1, 0);
      const size_t per = (idx.size() + nt - 1) / std::max(1u, nt);
      auto count = [&](unsigned k) {
        size_t c = 0;
        for (size_t i = std::min(idx.size(), k * per), e = std::min(idx.size(), (k + 1) * per); i < e; i++)
          c += valid(idx[i]) ? (size_t)(R.offset[(size_t)idx[i] + 1] - R.offset[(size_t)idx[i]]) : 1;
        start[k + 1] = c;
      };
      if (!run_workers(nt, count)) return PTMI_ERR_NO_MEMORY;
      for (unsigned k = 0; k < nt; k++) start[k + 1] += start[k];
      *n = start[nt];
      if (*n == 0) return PTMI_OK;
      float* flat = (float*)malloc(*n * sizeof(float));
      if (!flat) return PTMI_ERR_NO_MEMORY;
      auto fill = [&](unsigned k) {
        float* o = flat + start[k];
        for (size_t i = std::min(idx.size(), k * per), e = std::min(idx.size(), (k + 1) * per); i < e; i++) {
          if (!valid(idx[i])) {
            *o++ = NAN;
            continue;
          }
          const size_t r = (size_t)idx[i];
          for (uint32_t j = R.offset[r]; j < R.offset[r + 1]; j++) *o++ = (float)R.data[j];
        }
      };
      if (!run_workers(nt, fill)) {
        free(flat);
        return PTMI_ERR_NO_MEMORY;
      }
      *out = flat;
      return PTMI_OK;
    };
    int rc = flatten(V, vidx, vertices_out, n_vertices);
    if (rc) return rc;
    rc = flatten(N, nidx, normals_out, n_normals);
    if (rc) {
      free(*vertices_out);
      *vertices_out = nullptr;
      return rc;
    }
  } catch (...) {
    return PTMI_ERR_NO_MEMORY;
  }
  return PTMI_OK;
}
