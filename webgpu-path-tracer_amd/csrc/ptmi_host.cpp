// ptmi_host.cpp — host-side natives of libptmi.so (no GPU needed).
//
// ptmi_build_bvh: the reference's BVH construction (lib/BVH/bvhNode.js:21-101: top-down, split the
// longest axis of the node's box at the median after a STABLE sort on bbox.min[axis]; one primitive
// per leaf) and its pre-order flattening (lib/BVH/bvhBuilder.js:37-54), in doubles like the JS, rounded
// to f32 on the final store like `new Float32Array(...)` (lib/scene.js:304).  Output is byte-identical
// to the reference's for the same boxes (tests/test_host_buffers.py checks it against goldens).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ptmi.h"

namespace {

struct Builder {
  const double* bmin;
  const double* bmax;
  int prim_type;
  float* nodes;
  int64_t* order;
  std::vector<int64_t> tmp;
  std::vector<int64_t> left, right;
  int64_t counter = 0;

  int64_t gen(int64_t start, int64_t end) {
    const int64_t id = counter++;
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
    } else {
      int64_t* first = order + start;
      const double* keys = bmin;
      std::stable_sort(first, first + span + 1, [keys, axis](int64_t a, int64_t b) { return keys[3 * a + axis] < keys[3 * b + axis]; });
      const int64_t mid = start + span / 2;
      const int64_t l = gen(start, mid);
      const int64_t r = gen(mid + 1, end);
      left[id] = l;
      right[id] = r;
      row[3] = (float)r;
      row[7] = row[8] = row[9] = -1.0f;
      row[11] = (float)axis;
    }
    return id;
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
  b.gen(0, (int64_t)n_prims - 1);
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
