// ptmi_host.cpp — host-side natives of libptmi.so (no GPU needed).
//
// ptmi_build_bvh: the reference's BVH construction (lib/BVH/bvhNode.js:21-101: top-down, split the
// longest axis of the node's box at the median after a STABLE sort on bbox.min[axis]; one primitive
// per leaf) and its pre-order flattening (lib/BVH/bvhBuilder.js:37-54), in doubles like the JS, rounded
// to f32 on the final store like `new Float32Array(...)` (lib/scene.js:304).  Output is byte-identical
// to the reference's for the same boxes (tests/test_host_buffers.py checks it against goldens).
#include <algorithm>
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
        lo[k] = std::min(a[k], lo[k]);
        hi[k] = std::max(b[k], hi[k]);
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


// ---- OBJ parsing with the reference's grammar (lib/primitives/objReader.js:10-68) ---------------------------------
namespace {

bool js_space(unsigned char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\v' || ch == '\f'; }

// JavaScript Number(token) for the tokens an OBJ line can produce: "" -> 0, decimal literals with optional sign /
// fraction / exponent, [+-]Infinity, 0x / 0o / 0b integers; anything else -> NaN.
double js_number(const char* b, const char* e) {
  while (b < e && js_space((unsigned char)*b)) b++;
  while (e > b && js_space((unsigned char)e[-1])) e--;
  if (b == e) return 0.0;
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

extern "C" int ptmi_obj_parse(const char* text, size_t len, float** vertices_out, size_t* n_vertices, float** normals_out, size_t* n_normals) {
  if (!text || !vertices_out || !n_vertices || !normals_out || !n_normals) return PTMI_ERR_INVALID_ARG;
  *vertices_out = *normals_out = nullptr;
  *n_vertices = *n_normals = 0;
  ObjRows V, N;
  std::vector<double> vidx, nidx;  // doubles: an index token may be NaN
  try {
    const char* p = text;
    const char* end = text + len;
    while (p <= end) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* le = nl ? nl : end;
      const char *b = p, *e = le;  // line.trim()
      while (b < e && js_space((unsigned char)*b)) b++;
      while (e > b && js_space((unsigned char)e[-1])) e--;
      if (b < e && *b != '#') {
        if (e - b >= 2 && b[0] == 'v' && b[1] == ' ') {
          V.add_row_split_on_space(b, e);
        } else if (e - b >= 2 && b[0] == 'f' && b[1] == ' ') {  // split(/[\s/]+/).slice(1); i % 3 == 0 -> vertex, == 2 -> normal
          const char* q = b;
          int tok = -1;  // token 0 is "f"
          while (q < e) {
            const char* t0 = q;
            while (q < e && !js_space((unsigned char)*q) && *q != '/') q++;
            if (tok >= 0) {
              if (tok % 3 == 0) vidx.push_back(js_number(t0, q) - 1);
              else if (tok % 3 == 2) nidx.push_back(js_number(t0, q) - 1);
            }
            tok++;
            while (q < e && (js_space((unsigned char)*q) || *q == '/')) q++;
          }
        } else if (e - b >= 3 && b[0] == 'v' && b[1] == 'n' && b[2] == ' ') {
          N.add_row_split_on_space(b, e);
        }
      }
      if (!nl) break;
      p = nl + 1;
    }
    auto flatten = [](const ObjRows& R, const std::vector<double>& idx, float** out, size_t* n) -> int {
      std::vector<float> flat;
      flat.reserve(idx.size() * 3);
      for (double d : idx) {
        // array[v] with v not a valid index is `undefined`; .flat(1) keeps it as one element -> NaN in the Float32Array
        if (!(d >= 0) || d != std::floor(d) || d >= (double)R.rows()) {
          flat.push_back(NAN);
          continue;
        }
        size_t r = (size_t)d;
        for (uint32_t k = R.offset[r]; k < R.offset[r + 1]; k++) flat.push_back((float)R.data[k]);
      }
      *n = flat.size();
      if (flat.empty()) return PTMI_OK;
      *out = (float*)malloc(flat.size() * sizeof(float));
      if (!*out) return PTMI_ERR_NO_MEMORY;
      memcpy(*out, flat.data(), flat.size() * sizeof(float));
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
