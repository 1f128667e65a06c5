// ptmi_bvh_device.hip — the reference's median-split BVH build (lib/BVH/bvhNode.js:21-101) and its pre-order flattening
// with skip links (lib/BVH/bvhBuilder.js:37-54, bvhNode.js:76-93) ON THE GPU, level by level.
//
// Output is byte-identical to ptmi_build_bvh (the host builder) and hence to the reference's JavaScript:
//   * a node's box is the min / max over its primitives' boxes starting from AABB() = (+1e30, -1e30): min and max are exact,
//     so the order of the reduction does not matter (rocprim::segmented_reduce);
//   * the split axis comes from the same three f64 subtractions and comparisons;
//   * "stable sort of the node's primitives by bbox.min[axis]" is a stable segmented radix sort on the f64 keys (rocPRIM's
//     radix sort is stable; -0 is normalised to +0 first, so the radix order is the order of JavaScript's `a - b`);
//   * the pre-order id needs no counter: every leaf holds one primitive, a subtree over k primitives has 2k-1 nodes, so the
//     children of node `id` over [start, end] split at mid are id+1 and id + 2*(mid-start+1); the skip link is handed down
//     (left child: the right sibling; right child: the parent's link).
// One level = one device-wide reduce-by-key for the node boxes, one node kernel, one scan, TWO device-wide stable radix sorts
// (by the f64 key, then by the start of the primitive's node: together a stable sort inside every node's range, whatever the
// sizes of the ranges) and a few element kernels; ~21 levels for 871 k triangles.  (Round 2 used rocPRIM's SEGMENTED reduce and
// sort: they give one workgroup to each large segment, so the first ~17 levels — 1, 2, 4 ... huge segments — took 4.5 ms each
// and the GPU build lost to the host's threads: 118 ms against 98 for 871 k boxes.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/ptmi.h"

namespace {

struct Box6 {
  double lo[3], hi[3];
};
// Math.min / Math.max (AABB.js:8-28): exact, and -0 < +0 whatever the argument order — so the reduction is associative and
// commutative bit for bit and its order does not matter
__host__ __device__ inline double js_min(double a, double b) { return (a < b || (a == b && (a != 0.0 || __builtin_signbit(a)))) ? a : b; }
__host__ __device__ inline double js_max(double a, double b) { return (a > b || (a == b && (a != 0.0 || !__builtin_signbit(a)))) ? a : b; }
struct BoxMerge {
  __host__ __device__ Box6 operator()(const Box6& a, const Box6& b) const {
    Box6 r;
    for (int k = 0; k < 3; k++) {
      r.lo[k] = js_min(a.lo[k], b.lo[k]);
      r.hi[k] = js_max(a.hi[k], b.hi[k]);
    }
    return r;
  }
};
struct BoxOfPrim {  // order[i] -> that primitive's box
  const double* bmin;
  const double* bmax;
  __host__ __device__ Box6 operator()(uint32_t p) const {
    Box6 r;
    for (int k = 0; k < 3; k++) r.lo[k] = bmin[3 * (size_t)p + k], r.hi[k] = bmax[3 * (size_t)p + k];
    return r;
  }
};

struct LevelNode {
  uint32_t start, end;  // inclusive range in the primitive order
  uint32_t id;          // pre-order id = row in the flattened array
  int32_t next;         // skip link (row id), -1 = none
};
// per node of the level: write its row, decide leaf / inner, choose the axis; inner[j] = 1 if it has children
__global__ void k_level_rows(const LevelNode* __restrict__ nodes, const Box6* __restrict__ boxes, uint32_t m, int prim_type, float* __restrict__ rows,
                             int32_t* __restrict__ axis, uint32_t* __restrict__ inner) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const LevelNode nd = nodes[j];
  const Box6 b = boxes[j];
  const double e0 = b.hi[0] - b.lo[0], e1 = b.hi[1] - b.lo[1], e2 = b.hi[2] - b.lo[2];
  int a = 0;
  if (e1 > e0) a = 1;
  if (e2 > (a == 0 ? e0 : e1)) a = 2;
  float* row = rows + 12 * (size_t)nd.id;
  row[0] = (float)b.lo[0], row[1] = (float)b.lo[1], row[2] = (float)b.lo[2];
  row[4] = (float)b.hi[0], row[5] = (float)b.hi[1], row[6] = (float)b.hi[2];
  row[10] = nd.next < 0 ? -1.0f : (float)nd.next;
  const bool leaf = nd.end <= nd.start;
  if (leaf) {  // bvhNode.js:47-53
    row[3] = -1.0f;
    row[7] = (float)prim_type;
    row[8] = (float)nd.start;
    row[9] = (float)(nd.end - nd.start + 1);
    row[11] = 0.0f;
  } else {
    const uint32_t mid = nd.start + (nd.end - nd.start) / 2;
    row[3] = (float)(nd.id + 2u * (mid - nd.start + 1u));
    row[7] = row[8] = row[9] = -1.0f;
    row[11] = (float)a;
  }
  axis[j] = a;
  inner[j] = leaf ? 0u : 1u;
}

// children of the inner nodes, in node order (left before right)
__global__ void k_level_children(const LevelNode* __restrict__ nodes, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, uint32_t m,
                                 LevelNode* __restrict__ next_nodes) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m || !inner[j]) return;
  const LevelNode nd = nodes[j];
  const uint32_t r = rank[j];  // index among this level's inner nodes
  const uint32_t mid = nd.start + (nd.end - nd.start) / 2;
  const uint32_t lid = nd.id + 1u, rid = nd.id + 2u * (mid - nd.start + 1u);
  next_nodes[2 * r] = LevelNode{nd.start, mid, lid, (int32_t)rid};
  next_nodes[2 * r + 1] = LevelNode{mid + 1u, nd.end, rid, nd.next};
}

// the boxes of the level's nodes out of the runs reduce_by_key found (a node's primitives are contiguous, so one run per node;
// runs of finished primitives carry the key -1)
__global__ void k_level_scatter_boxes(const int32_t* __restrict__ run_key, const Box6* __restrict__ run_box, const uint32_t* __restrict__ n_runs, uint32_t n,
                                      Box6* __restrict__ boxes) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || r >= *n_runs) return;
  const int32_t j = run_key[r];
  if (j >= 0) boxes[j] = run_box[r];
}

// Sort keys of this level.  A primitive in an inner node: minor key = bbox.min[axis] (with -0 -> +0: `a - b` treats them as equal),
// major key = the start of its node's range; every other primitive (leaf reached, or finished earlier): major key = its own
// position, so it stays where it is.  Ranges are disjoint and contiguous, hence "stable sort by minor, then stable sort by major" =
// a stable sort by the minor key inside every range.  seg_of[i] = index of the primitive's node in the level, -1 = done.
__global__ void k_level_keys(const uint32_t* __restrict__ order, const int32_t* __restrict__ seg_of, const int32_t* __restrict__ axis, const uint32_t* __restrict__ inner,
                             const LevelNode* __restrict__ nodes, const double* __restrict__ bmin, uint32_t n, double* __restrict__ keys, uint64_t* __restrict__ packed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t j = seg_of[i];
  const uint32_t p = order[i];
  double k = 0.0;
  uint32_t major = i;
  if (j >= 0 && inner[j]) {
    k = bmin[3 * (size_t)p + axis[j]] + 0.0;
    major = nodes[j].start;
  }
  keys[i] = k;
  packed[i] = ((uint64_t)major << 32) | p;
}
__global__ void k_level_unpack(const uint64_t* __restrict__ packed, uint32_t n, uint32_t* __restrict__ major, uint32_t* __restrict__ prim) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = packed[i];
  major[i] = (uint32_t)(v >> 32);
  prim[i] = (uint32_t)v;
}

// after the sort: which node of the NEXT level each primitive belongs to
__global__ void k_level_descend(const LevelNode* __restrict__ nodes, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, uint32_t n,
                                int32_t* __restrict__ seg_of) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t j = seg_of[i];
  if (j < 0) return;
  if (!inner[j]) {
    seg_of[i] = -1;
    return;
  }
  const LevelNode nd = nodes[j];
  const uint32_t mid = nd.start + (nd.end - nd.start) / 2;
  seg_of[i] = (int32_t)(2u * rank[j] + (i > mid ? 1u : 0u));
}

__global__ void k_iota(uint32_t* __restrict__ order, int32_t* __restrict__ seg_of, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  order[i] = i;
  seg_of[i] = 0;
}

struct Dev {  // frees everything on scope exit
  std::vector<void*> ptrs;
  template <class T>
  hipError_t alloc(T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(q);
    *p = (T*)q;
    return e;
  }
  ~Dev() {
    for (void* p : ptrs) (void)hipFree(p);
  }
};

#define TRY(expr)                      \
  do {                                 \
    hipError_t _e = (expr);            \
    if (_e != hipSuccess) return _e;   \
  } while (0)

// The level loop on boxes that are already on the device; leaves the rows (12 floats per node, 2n-1 nodes) and the primitive order on
// the device too (the caller's buffers).  `d` owns the scratch.
hipError_t build_levels(hipStream_t stream, Dev& d, uint32_t n, const double* bmin, const double* bmax, int prim_type, float* rows, uint32_t* order_out,
                        int* depth_out = nullptr) {
  double* keys[2];
  uint64_t* packed[2];
  uint32_t *order[2], *major[2], *inner, *rank, *n_runs;
  int32_t *seg_of, *axis, *run_key;
  LevelNode* level[2];
  Box6 *boxes, *run_box;
  for (int k = 0; k < 2; k++) {
    TRY(d.alloc(&keys[k], n));
    TRY(d.alloc(&packed[k], n));
    TRY(d.alloc(&order[k], n));
    TRY(d.alloc(&major[k], n));
    TRY(d.alloc(&level[k], n));
  }
  TRY(d.alloc(&seg_of, n));
  TRY(d.alloc(&inner, n));
  TRY(d.alloc(&rank, n));
  TRY(d.alloc(&axis, n));
  TRY(d.alloc(&boxes, n));
  TRY(d.alloc(&run_key, n));
  TRY(d.alloc(&run_box, n));
  TRY(d.alloc(&n_runs, 1));

  const BoxOfPrim box_of{bmin, bmax};
  unsigned major_bits = 1;
  while ((1ull << major_bits) < (unsigned long long)n) major_bits++;

  // temporary storage for the rocPRIM calls at their largest size
  size_t t_reduce = 0, t_scan = 0, t_sort1 = 0, t_sort2 = 0;
  {
    auto in = rocprim::make_transform_iterator(order[0], box_of);
    TRY(rocprim::reduce_by_key(nullptr, t_reduce, seg_of, in, n, run_key, run_box, n_runs, BoxMerge(), rocprim::equal_to<int32_t>(), stream));
    TRY(rocprim::exclusive_scan(nullptr, t_scan, inner, rank, 0u, n, rocprim::plus<uint32_t>(), stream));
    TRY(rocprim::radix_sort_pairs(nullptr, t_sort1, keys[0], keys[1], packed[0], packed[1], n, 0, 64, stream));
    TRY(rocprim::radix_sort_pairs(nullptr, t_sort2, major[0], major[1], order[0], order[1], n, 0, major_bits, stream));
  }
  char* temp;
  const size_t t_bytes = std::max(std::max(t_reduce, t_scan), std::max(t_sort1, t_sort2));
  TRY(d.alloc(&temp, t_bytes));

  const unsigned B = 256;
  hipLaunchKernelGGL(k_iota, dim3((n + B - 1) / B), dim3(B), 0, stream, order[0], seg_of, n);
  const LevelNode root{0u, n - 1u, 0u, -1};
  TRY(hipMemcpyAsync(level[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
  TRY(hipStreamSynchronize(stream));  // `root` is a stack variable

  uint32_t m = 1;
  int cur = 0, ocur = 0;  // ping-pong indices of the level arrays and of the order arrays
  int levels = 0;
  while (m > 0) {
    levels++;
    // node boxes of ALL nodes of the level (leaves included): one run of equal seg_of per node
    {
      auto in = rocprim::make_transform_iterator(order[ocur], box_of);
      size_t tb = t_bytes;
      TRY(rocprim::reduce_by_key(temp, tb, seg_of, in, n, run_key, run_box, n_runs, BoxMerge(), rocprim::equal_to<int32_t>(), stream));
    }
    hipLaunchKernelGGL(k_level_scatter_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, run_key, run_box, n_runs, n, boxes);
    hipLaunchKernelGGL(k_level_rows, dim3((m + B - 1) / B), dim3(B), 0, stream, level[cur], boxes, m, prim_type, rows, axis, inner);
    {
      size_t tb = t_bytes;
      TRY(rocprim::exclusive_scan(temp, tb, inner, rank, 0u, m, rocprim::plus<uint32_t>(), stream));
    }
    // number of inner nodes = rank[m-1] + inner[m-1]
    uint32_t tail[2];
    TRY(hipMemcpyAsync(&tail[0], rank + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    TRY(hipMemcpyAsync(&tail[1], inner + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    TRY(hipStreamSynchronize(stream));
    const uint32_t n_inner = tail[0] + tail[1];
    if (n_inner == 0) break;
    hipLaunchKernelGGL(k_level_children, dim3((m + B - 1) / B), dim3(B), 0, stream, level[cur], inner, rank, m, level[cur ^ 1]);
    hipLaunchKernelGGL(k_level_keys, dim3((n + B - 1) / B), dim3(B), 0, stream, order[ocur], seg_of, axis, inner, level[cur], bmin, n, keys[0], packed[0]);
    {
      size_t tb = t_bytes;
      TRY(rocprim::radix_sort_pairs(temp, tb, keys[0], keys[1], packed[0], packed[1], n, 0, 64, stream));  // stable, by bbox.min[axis]
    }
    hipLaunchKernelGGL(k_level_unpack, dim3((n + B - 1) / B), dim3(B), 0, stream, packed[1], n, major[0], order[ocur]);
    {
      size_t tb = t_bytes;
      TRY(rocprim::radix_sort_pairs(temp, tb, major[0], major[1], order[ocur], order[ocur ^ 1], n, 0, major_bits, stream));  // stable, by node
    }
    ocur ^= 1;
    hipLaunchKernelGGL(k_level_descend, dim3((n + B - 1) / B), dim3(B), 0, stream, level[cur], inner, rank, n, seg_of);
    cur ^= 1;
    m = 2 * n_inner;
  }
  TRY(hipGetLastError());
  TRY(hipMemcpyAsync(order_out, order[ocur], (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  if (depth_out) *depth_out = levels - 1;  // inner nodes on the longest root-to-leaf path
  return hipSuccess;
}

hipError_t build_on_device(hipStream_t stream, uint32_t n, const double* h_bmin, const double* h_bmax, int prim_type, float* h_rows, int64_t* h_order) {
  Dev d;
  const uint32_t nn = 2 * n - 1;
  double *bmin, *bmax;
  float* rows;
  uint32_t* order;
  TRY(d.alloc(&bmin, 3 * (size_t)n));
  TRY(d.alloc(&bmax, 3 * (size_t)n));
  TRY(d.alloc(&rows, 12 * (size_t)nn));
  TRY(d.alloc(&order, n));
  TRY(hipMemcpyAsync(bmin, h_bmin, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(bmax, h_bmax, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
  TRY(build_levels(stream, d, n, bmin, bmax, prim_type, rows, order));
  std::vector<uint32_t> ord(n);
  TRY(hipMemcpyAsync(h_rows, rows, 12 * (size_t)nn * sizeof(float), hipMemcpyDeviceToHost, stream));
  TRY(hipMemcpyAsync(ord.data(), order, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  TRY(hipStreamSynchronize(stream));
  for (uint32_t i = 0; i < n; i++) h_order[i] = (int64_t)ord[i];
  return hipSuccess;
}


// ---- the scene's BVH built where the triangles already are (ptmi_build_scene_bvh) -------------------------------------------------
// World-space box of every uploaded triangle exactly as the reference computes it on the host: vertices through the mesh's model
// matrix in double (gl-matrix vec3.transformMat4 on f32 inputs, lib/primitives/triangle.js:27-39), stored as f32, min / max over
// the three, then AABB.pad() (lib/BVH/AABB.js:35-51: an axis thinner than 0.00005 grows by that much on both sides).
__global__ void k_scene_boxes(const float* __restrict__ tris, uint32_t n, const int32_t* __restrict__ meshes, int n_meshes, const float* __restrict__ xforms, int n_xforms,
                              double* __restrict__ bmin, double* __restrict__ bmax, uint32_t* __restrict__ first_bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* t = tris + 24 * (size_t)i;
  const float mf = t[23];  // mesh_id travels as a float (triangle.js:42-52)
  int gid = -1;
  if (mf >= 0.0f && mf < 2147483000.0f && (int)mf < n_meshes) gid = meshes[4 * (int)mf + 2];
  if (gid < 0 || gid >= n_xforms) {
    atomicMin(first_bad, i);
    gid = -1;
  }
  double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (gid >= 0)
    for (int k = 0; k < 16; k++) m[k] = (double)xforms[32 * (size_t)gid + k];
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int v = 0; v < 3; v++) {
    const double x = (double)t[4 * v], y = (double)t[4 * v + 1], z = (double)t[4 * v + 2];
    double w = m[3] * x + m[7] * y + m[11] * z + m[15];
    if (w == 0.0 || w != w) w = 1.0;  // `w = w || 1.0`
    const double p[3] = {(double)(float)((m[0] * x + m[4] * y + m[8] * z + m[12]) / w), (double)(float)((m[1] * x + m[5] * y + m[9] * z + m[13]) / w),
                         (double)(float)((m[2] * x + m[6] * y + m[10] * z + m[14]) / w)};
    for (int k = 0; k < 3; k++) {
      lo[k] = v == 0 ? p[k] : js_min(lo[k], p[k]);
      hi[k] = v == 0 ? p[k] : js_max(hi[k], p[k]);
    }
  }
  const double delta = 0.0001 / 2;
  for (int k = 0; k < 3; k++) {
    const bool thin = (hi[k] - lo[k]) < delta;
    bmin[3 * (size_t)i + k] = thin ? lo[k] - delta : lo[k];
    bmax[3 * (size_t)i + k] = thin ? hi[k] + delta : hi[k];
  }
}

// triangles into BVH leaf order (lib/scene.js:257)
__global__ void k_permute_triangles(const float4* __restrict__ src, const uint32_t* __restrict__ order, uint32_t n, float4* __restrict__ dst) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 (of the 6 per triangle) per thread
  if (g >= 6u * n) return;
  const uint32_t k = g / 6u, part = g - 6u * k;
  dst[g] = src[6 * (size_t)order[k] + part];
}

// pair64 (csrc/ptmi_device.h) from rows that stay on the device: is_inner -> exclusive scan = the pair index, then one record per inner node
__global__ void k_rows_inner_flag(const float* __restrict__ rows, uint32_t nn, uint32_t* __restrict__ inner) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nn) inner[i] = rows[12 * (size_t)i + 7] == -1.0f ? 1u : 0u;  // prim_type -1 = inner (bvhBuilder.js:45,49)
}
__global__ void k_rows_to_pairs(const float* __restrict__ rows, uint32_t nn, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, float* __restrict__ pairs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn || !inner[i]) return;
  const float* nd = rows + 12 * (size_t)i;
  const uint32_t L = i + 1u, R = (uint32_t)nd[3];
  const float *nl = rows + 12 * (size_t)L, *nr = rows + 12 * (size_t)R;
  auto ref_of = [&](uint32_t j, const float* row) -> uint32_t { return inner[j] ? rank[j] : (0x80000000u | (uint32_t)row[8]); };  // REF_LEAF | prim_id: one triangle per leaf
  float* o = pairs + 16 * (size_t)rank[i];
  o[0] = nl[0], o[1] = nl[1], o[2] = nl[2], o[3] = __uint_as_float(ref_of(L, nl));
  o[4] = nl[4], o[5] = nl[5], o[6] = nl[6], o[7] = __uint_as_float(ref_of(R, nr));
  o[8] = nr[0], o[9] = nr[1], o[10] = nr[2], o[11] = __int_as_float((int)nd[11]);
  o[12] = nr[4], o[13] = nr[5], o[14] = nr[6], o[15] = 0.0f;
}

}  // namespace

// ptmi.hip supplies the context's device and error slot through these two accessors (the stream through ptmi_stream).
int ptmi_ctx_set_device(ptmi_ctx* ctx);                        // ptmi.hip
int ptmi_ctx_fail(ptmi_ctx* ctx, int code, const char* what);  // ptmi.hip

extern "C" int ptmi_build_bvh_device(ptmi_ctx* ctx, size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out, int64_t* order_out) {
  if (!ctx) return PTMI_ERR_INVALID_ARG;
  if (n_prims == 0) return PTMI_OK;
  if (!bmin || !bmax || !nodes_out || !order_out) return ptmi_ctx_fail(ctx, PTMI_ERR_INVALID_ARG, "ptmi_build_bvh_device: null argument");
  if (n_prims > (size_t)1 << 27) return ptmi_ctx_fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_build_bvh_device: more than 2^27 primitives");
  int r = ptmi_ctx_set_device(ctx);
  if (r) return r;
  void* s = nullptr;
  ptmi_stream(ctx, &s);
  hipError_t e;
  try {
    e = build_on_device((hipStream_t)s, (uint32_t)n_prims, bmin, bmax, prim_type, nodes_out, order_out);
  } catch (...) {
    return ptmi_ctx_fail(ctx, PTMI_ERR_NO_MEMORY, "ptmi_build_bvh_device: host allocation failed");
  }
  if (e != hipSuccess) return ptmi_ctx_fail(ctx, e == hipErrorOutOfMemory ? PTMI_ERR_NO_MEMORY : PTMI_ERR_DEVICE, hipGetErrorString(e));
  return PTMI_OK;
}

// ---- for ptmi.hip (ptmi_build_scene_bvh / prepare_scene): everything stays on the device -------------------------------------------
// Builds the reference's median-split BVH over the n triangles at d_tris (24 f32 each, upload order): boxes, level loop, rows into
// d_rows (12 f32 x (2n-1)), the triangles in leaf order into d_tris_out.  meshes / transforms are host arrays (small).  *bad_tri = first
// triangle whose mesh / transform index is out of range (0xffffffff = none).  Returns a hipError_t.
int ptmi_bvhdev_build_scene(void* stream_, const float* d_tris, uint32_t n, const int32_t* h_meshes, int n_meshes, const float* h_xforms, int n_xforms, float* d_rows,
                            float* d_tris_out, int* depth_out, uint32_t* bad_tri) {
  hipStream_t stream = (hipStream_t)stream_;
  try {
    Dev d;
    double *bmin, *bmax;
    int32_t* meshes;
    float* xforms;
    uint32_t *order, *bad;
    TRY(d.alloc(&bmin, 3 * (size_t)n));
    TRY(d.alloc(&bmax, 3 * (size_t)n));
    TRY(d.alloc(&meshes, 4 * (size_t)n_meshes));
    TRY(d.alloc(&xforms, 32 * (size_t)n_xforms));
    TRY(d.alloc(&order, n));
    TRY(d.alloc(&bad, 1));
    const uint32_t none = 0xffffffffu;
    TRY(hipMemcpyAsync(bad, &none, 4, hipMemcpyHostToDevice, stream));
    if (n_meshes) TRY(hipMemcpyAsync(meshes, h_meshes, 16 * (size_t)n_meshes, hipMemcpyHostToDevice, stream));
    if (n_xforms) TRY(hipMemcpyAsync(xforms, h_xforms, 128 * (size_t)n_xforms, hipMemcpyHostToDevice, stream));
    const unsigned B = 256;
    hipLaunchKernelGGL(k_scene_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, d_tris, n, meshes, n_meshes, xforms, n_xforms, bmin, bmax, bad);
    TRY(hipMemcpyAsync(bad_tri, bad, 4, hipMemcpyDeviceToHost, stream));
    TRY(hipStreamSynchronize(stream));
    if (*bad_tri != none) return (int)hipSuccess;  // the caller reports it
    TRY(build_levels(stream, d, n, bmin, bmax, 2, d_rows, order, depth_out));
    hipLaunchKernelGGL(k_permute_triangles, dim3((6 * n + B - 1) / B), dim3(B), 0, stream, reinterpret_cast<const float4*>(d_tris), order, n,
                       reinterpret_cast<float4*>(d_tris_out));
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(stream));  // the scratch dies with `d`
    return (int)hipSuccess;
  } catch (...) {
    return (int)hipErrorOutOfMemory;
  }
}

// pair64 records (16 f32 per inner node, (nn-1)/2 of them) from device-resident rows
int ptmi_bvhdev_make_pairs(void* stream_, const float* d_rows, uint32_t nn, float* d_pairs) {
  hipStream_t stream = (hipStream_t)stream_;
  try {
    Dev d;
    uint32_t *inner, *rank;
    char* temp;
    TRY(d.alloc(&inner, nn));
    TRY(d.alloc(&rank, nn));
    size_t tb = 0;
    TRY(rocprim::exclusive_scan(nullptr, tb, inner, rank, 0u, nn, rocprim::plus<uint32_t>(), stream));
    TRY(d.alloc(&temp, tb));
    const unsigned B = 256;
    hipLaunchKernelGGL(k_rows_inner_flag, dim3((nn + B - 1) / B), dim3(B), 0, stream, d_rows, nn, inner);
    TRY(rocprim::exclusive_scan(temp, tb, inner, rank, 0u, nn, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_rows_to_pairs, dim3((nn + B - 1) / B), dim3(B), 0, stream, d_rows, nn, inner, rank, d_pairs);
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(stream));
    return (int)hipSuccess;
  } catch (...) {
    return (int)hipErrorOutOfMemory;
  }
}

#ifdef PTMI_EXPERIMENTS
// PTMI_DIAG_SORT (experiment): device-wide radix sort of (key, slot) pairs; scratch allocated per call
int ptmi_diag_sort_pairs(void* stream_, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, uint32_t n) {
  hipStream_t stream = (hipStream_t)stream_;
  try {
    Dev d;
    size_t tb = 0;
    TRY(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, 0, 32, stream));
    char* temp;
    TRY(d.alloc(&temp, tb));
    TRY(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, vals_out, n, 0, 32, stream));
    TRY(hipStreamSynchronize(stream));
    return (int)hipSuccess;
  } catch (...) {
    return (int)hipErrorOutOfMemory;
  }
}
#endif  // PTMI_EXPERIMENTS
