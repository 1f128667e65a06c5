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


// ---- the reference's binned-SAH builder (lib/BVH/bvhNode.js:108-283) on the GPU, level by level ------------------------------------------
// Byte-identical to ptmi_build_bvh_sah (csrc/ptmi_host.cpp), the host restatement that is pinned to the reference's JavaScript
// (tests/golden/c2sah_bvh.bin).  Why a level-synchronous build gives the recursive one's bytes:
//   * a node's box, its centroid bounds and its 3 x 8 bins are min / max / counts over the node's primitives: exact and order-free.  Boxes go
//     through Math.min / Math.max (-0 < +0: AABB.merge) — on the device as integer atomics on an order-preserving 64-bit image of the
//     double, where -0 sorts below +0 by construction; the centroid bounds' zero sign never reaches a result (it only enters x - bound,
//     x == bound and bound + positive);
//   * the 7 x 3 candidate costs are the same f64 expressions evaluated per node by one thread (-ffp-contract=off: nothing fuses);
//   * "stable sort of the node's range by bbox.min[axis]" = the two device-wide stable radix sorts of the median builder above (ranges of
//     sibling subtrees are disjoint, so depth-first and level order sort the same ranges in the same state);
//   * the split position — the first primitive, in sorted order, whose centroid lies beyond the plane, at most end - 1 — is an atomicMin
//     over positions;
//   * leaves hold any number of primitives here, so pre-order ids cannot be counted top-down: the tree is built under temporary ids (level
//     order), subtree sizes are summed level by level from the bottom, pre-order ids and skip links (bvhBuilder.js:37-54, bvhNode.js:76-93)
//     handed down from the root, and the rows written last.
struct SahAgg {
  Box6 box;
  double cmin[3], cmax[3];  // bounds of the centroids (FindBestSplitPlane's boundsMin / boundsMax per axis)
};
struct SahAggMerge {
  __host__ __device__ SahAgg operator()(const SahAgg& a, const SahAgg& b) const {
    SahAgg r;
    for (int k = 0; k < 3; k++) {
      r.box.lo[k] = js_min(a.box.lo[k], b.box.lo[k]);
      r.box.hi[k] = js_max(a.box.hi[k], b.box.hi[k]);
      r.cmin[k] = b.cmin[k] < a.cmin[k] ? b.cmin[k] : a.cmin[k];
      r.cmax[k] = b.cmax[k] > a.cmax[k] ? b.cmax[k] : a.cmax[k];
    }
    return r;
  }
};
struct SahAggOfPrim {
  const double* bmin;
  const double* bmax;
  __host__ __device__ SahAgg operator()(uint32_t p) const {
    SahAgg r;
    for (int k = 0; k < 3; k++) {
      const double a = bmin[3 * (size_t)p + k], b = bmax[3 * (size_t)p + k];
      r.box.lo[k] = a, r.box.hi[k] = b;
      r.cmin[k] = r.cmax[k] = (a + b) / 2;  // BVH.get_centroid
    }
    return r;
  }
};
// order-preserving image of a finite double in u64: a < b  <=>  enc(a) < enc(b), with -0 below +0 (Math.min / Math.max's rule)
__host__ __device__ inline unsigned long long sah_enc(double x) {
  unsigned long long u;
  memcpy(&u, &x, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double sah_dec(unsigned long long u) {
  u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  double x;
  memcpy(&x, &u, 8);
  return x;
}
constexpr int kSahBins = 8, kSahMaxLevels = 4096;
struct alignas(64) SahBin {
  unsigned long long lo[3], hi[3];  // sah_enc'ed box of the primitives whose centroid falls into the bin
  uint32_t count;
};
// one per node under construction (temporary id = level order)
struct SahTreeNode {
  uint32_t start, end;  // inclusive range in the primitive order
  int32_t left, right;  // temporary ids of the children, -1 = leaf
  int32_t axis;
  float box[6];
};

__global__ void k_sah_scatter(const int32_t* __restrict__ run_key, const SahAgg* __restrict__ run_val, const uint32_t* __restrict__ n_runs, uint32_t n, SahAgg* __restrict__ agg) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || r >= *n_runs) return;
  const int32_t j = run_key[r];
  if (j >= 0) agg[j] = run_val[r];
}
__global__ void k_sah_bins_init(SahBin* __restrict__ bins, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  SahBin b;
  for (int k = 0; k < 3; k++) b.lo[k] = sah_enc(1e30), b.hi[k] = sah_enc(-1e30);  // new AABB()
  b.count = 0u;
  bins[i] = b;
}
// FindBestSplitPlane's binning pass (bvhNode.js:239-247) for every primitive of the level at once.  A block takes a run of positions; when the
// run lies within at most kSahLdsNodes nodes (the upper levels: a few huge nodes, where every thread of the machine would otherwise hammer the
// same 24 bins) it bins into LDS first and merges what it touched into the global bins afterwards.
constexpr int kSahLdsNodes = 8, kSahChunk = 2048;
struct SahAcc {
  __device__ static void add(SahBin* b, const double* lo, const double* hi) {
    atomicAdd(&b->count, 1u);
    for (int k = 0; k < 3; k++) {
      atomicMin(&b->lo[k], sah_enc(lo[k]));
      atomicMax(&b->hi[k], sah_enc(hi[k]));
    }
  }
};
__global__ __launch_bounds__(256) void k_sah_bins(const uint32_t* __restrict__ order, const int32_t* __restrict__ seg_of, const SahAgg* __restrict__ agg, const double* __restrict__ bmin,
                                                  const double* __restrict__ bmax, uint32_t n, SahBin* __restrict__ bins) {
  __shared__ SahBin s_bins[kSahLdsNodes * 3 * kSahBins];
  __shared__ int s_first, s_last;
  const uint32_t base = blockIdx.x * (uint32_t)kSahChunk;
  if (threadIdx.x == 0) s_first = 0x7fffffff, s_last = -1;
  __syncthreads();
  int jf = 0x7fffffff, jl = -1;
  for (uint32_t i = base + threadIdx.x; i < min(base + (uint32_t)kSahChunk, n); i += 256) {
    const int j = seg_of[i];
    if (j >= 0) jf = min(jf, j), jl = max(jl, j);
  }
  if (jl >= 0) atomicMin(&s_first, jf), atomicMax(&s_last, jl);
  __syncthreads();
  jf = s_first, jl = s_last;
  if (jl < 0) return;
  const bool lds = jl - jf < kSahLdsNodes;
  if (lds) {
    for (int k = threadIdx.x; k < (jl - jf + 1) * 3 * kSahBins; k += 256) {
      SahBin b;
      for (int c = 0; c < 3; c++) b.lo[c] = sah_enc(1e30), b.hi[c] = sah_enc(-1e30);
      b.count = 0u;
      s_bins[k] = b;
    }
    __syncthreads();
  }
  for (uint32_t i = base + threadIdx.x; i < min(base + (uint32_t)kSahChunk, n); i += 256) {
    const int j = seg_of[i];
    if (j < 0) continue;
    const uint32_t p = order[i];
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) lo[k] = bmin[3 * (size_t)p + k], hi[k] = bmax[3 * (size_t)p + k];
    const SahAgg& a = agg[j];
    for (int ax = 0; ax < 3; ax++) {
      const double bounds_min = a.cmin[ax], bounds_max = a.cmax[ax];
      if (bounds_min == bounds_max) continue;
      const double scale = kSahBins / (bounds_max - bounds_min);
      const double f = floor(((lo[ax] + hi[ax]) / 2 - bounds_min) * scale);
      const int b = (int)fmin((double)(kSahBins - 1), f);
      SahBin* dst = lds ? &s_bins[((j - jf) * 3 + ax) * kSahBins + b] : &bins[((size_t)j * 3 + ax) * kSahBins + b];
      SahAcc::add(dst, lo, hi);
    }
  }
  if (!lds) return;
  __syncthreads();
  for (int k = threadIdx.x; k < (jl - jf + 1) * 3 * kSahBins; k += 256) {
    const SahBin b = s_bins[k];
    if (b.count == 0u) continue;
    SahBin* dst = &bins[(size_t)jf * 3 * kSahBins + k];
    atomicAdd(&dst->count, b.count);
    for (int c = 0; c < 3; c++) {
      atomicMin(&dst->lo[c], b.lo[c]);
      atomicMax(&dst->hi[c], b.hi[c]);
    }
  }
}
__device__ inline double sah_area(const double* lo, const double* hi) {  // AABB.surface_area
  const double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
  return e0 * e1 + e1 * e2 + e2 * e0;
}
// per node of the level: the best of the 7 x 3 planes against the cost of not splitting (bvhNode.js:150-160, 249-281); records the node
__global__ void k_sah_decide(const LevelNode* __restrict__ nodes, const SahAgg* __restrict__ agg, const SahBin* __restrict__ bins, uint32_t m, SahTreeNode* __restrict__ tree,
                             int32_t* __restrict__ axis_out, double* __restrict__ split_pos_out, uint32_t* __restrict__ inner, uint32_t* __restrict__ split_first) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const LevelNode nd = nodes[j];
  const SahAgg a = agg[j];
  const double parent_cost = (double)(nd.end - nd.start + 1u) * sah_area(a.box.lo, a.box.hi);
  double best_cost = 1e30, split_pos = 0;
  int axis = 0;
  for (int ax = 0; ax < 3; ax++) {
    const double bounds_min = a.cmin[ax], bounds_max = a.cmax[ax];
    if (bounds_min == bounds_max) continue;
    const SahBin* bb = bins + ((size_t)j * 3 + ax) * kSahBins;
    double left_area[kSahBins - 1], right_area[kSahBins - 1], left_count[kSahBins - 1], right_count[kSahBins - 1];
    double llo[3] = {1e30, 1e30, 1e30}, lhi[3] = {-1e30, -1e30, -1e30}, rlo[3] = {1e30, 1e30, 1e30}, rhi[3] = {-1e30, -1e30, -1e30};
    double lsum = 0, rsum = 0;
    for (int i = 0; i < kSahBins - 1; i++) {
      const SahBin bl = bb[i], br = bb[kSahBins - 1 - i];
      lsum += (double)bl.count;
      left_count[i] = lsum;
      for (int k = 0; k < 3; k++) llo[k] = js_min(sah_dec(bl.lo[k]), llo[k]), lhi[k] = js_max(sah_dec(bl.hi[k]), lhi[k]);
      left_area[i] = sah_area(llo, lhi);
      rsum += (double)br.count;
      right_count[kSahBins - 2 - i] = rsum;
      for (int k = 0; k < 3; k++) rlo[k] = js_min(sah_dec(br.lo[k]), rlo[k]), rhi[k] = js_max(sah_dec(br.hi[k]), rhi[k]);
      right_area[kSahBins - 2 - i] = sah_area(rlo, rhi);
    }
    const double scale = (bounds_max - bounds_min) / kSahBins;
    for (int i = 0; i < kSahBins - 1; i++) {
      const double cost = left_count[i] * left_area[i] + right_count[i] * right_area[i];
      if (cost < best_cost) {
        axis = ax;
        split_pos = bounds_min + scale * (i + 1);
        best_cost = cost;
      }
    }
  }
  const bool leaf = best_cost >= parent_cost;
  SahTreeNode t;
  t.start = nd.start, t.end = nd.end;
  t.left = t.right = -1;
  t.axis = leaf ? 0 : axis;
  for (int k = 0; k < 3; k++) t.box[k] = (float)a.box.lo[k], t.box[3 + k] = (float)a.box.hi[k];
  tree[nd.id] = t;
  axis_out[j] = axis;
  split_pos_out[j] = split_pos;
  inner[j] = leaf ? 0u : 1u;
  split_first[j] = nd.end - 1u;  // `while (split < end - 1)`: the left side never takes the last primitive
}
// after the sort: the first position of each inner node whose centroid lies beyond the plane (bvhNode.js:170-178)
__global__ void k_sah_find_split(const uint32_t* __restrict__ order, const int32_t* __restrict__ seg_of, const uint32_t* __restrict__ inner, const int32_t* __restrict__ axis,
                                 const double* __restrict__ split_pos, const double* __restrict__ bmin, const double* __restrict__ bmax, uint32_t n, uint32_t* __restrict__ split_first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t j = seg_of[i];
  if (j < 0 || !inner[j]) return;
  const uint32_t p = order[i];
  const int ax = axis[j];
  const double c = (bmin[3 * (size_t)p + ax] + bmax[3 * (size_t)p + ax]) / 2;
  if (c <= split_pos[j]) return;
  // (a plain read first: the value only ever decreases, so a position that is not below it cannot lower it — at the top levels nearly every
  // thread behind the plane is turned away here instead of queueing on one address)
  if (i < *(volatile uint32_t*)&split_first[j]) atomicMin(&split_first[j], i);
}
__global__ void k_sah_children(const LevelNode* __restrict__ nodes, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ split_first, uint32_t m,
                               uint32_t next_base, LevelNode* __restrict__ next_nodes, SahTreeNode* __restrict__ tree) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m || !inner[j]) return;
  const LevelNode nd = nodes[j];
  const uint32_t r = rank[j], split = split_first[j];
  const uint32_t lid = next_base + 2u * r, rid = lid + 1u;
  next_nodes[2 * r] = LevelNode{nd.start, split, lid, -1};
  next_nodes[2 * r + 1] = LevelNode{split + 1u, nd.end, rid, -1};
  tree[nd.id].left = (int32_t)lid;
  tree[nd.id].right = (int32_t)rid;
}
__global__ void k_sah_descend(const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ split_first, uint32_t n, int32_t* __restrict__ seg_of) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t j = seg_of[i];
  if (j < 0) return;
  seg_of[i] = inner[j] ? (int32_t)(2u * rank[j] + (i > split_first[j] ? 1u : 0u)) : -1;
}
// subtree sizes, one level at a time from the bottom; then pre-order ids and skip links from the top
__global__ void k_sah_sizes(const SahTreeNode* __restrict__ tree, uint32_t base, uint32_t m, uint32_t* __restrict__ size) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const SahTreeNode& t = tree[base + j];
  size[base + j] = t.left < 0 ? 1u : 1u + size[t.left] + size[t.right];
}
__global__ void k_sah_flat(const SahTreeNode* __restrict__ tree, uint32_t base, uint32_t m, const uint32_t* __restrict__ size, uint32_t* __restrict__ flat, int32_t* __restrict__ next) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const SahTreeNode& t = tree[base + j];
  if (t.left < 0) return;
  const uint32_t id = flat[base + j];
  flat[t.left] = id + 1u;
  flat[t.right] = id + 1u + size[t.left];
  next[t.left] = (int32_t)flat[t.right];  // populate_links: the left child's link is its sibling, the right child inherits the parent's
  next[t.right] = next[base + j];
}
__global__ void k_sah_rows(const SahTreeNode* __restrict__ tree, uint32_t total, const uint32_t* __restrict__ flat, const int32_t* __restrict__ next, int prim_type, float* __restrict__ rows) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const SahTreeNode nd = tree[t];
  float* row = rows + 12 * (size_t)flat[t];
  row[0] = nd.box[0], row[1] = nd.box[1], row[2] = nd.box[2];
  row[4] = nd.box[3], row[5] = nd.box[4], row[6] = nd.box[5];
  row[10] = next[t] < 0 ? -1.0f : (float)next[t];
  if (nd.left < 0) {
    row[3] = -1.0f;
    row[7] = (float)prim_type;
    row[8] = (float)nd.start;
    row[9] = (float)(nd.end - nd.start + 1u);
    row[11] = 0.0f;
  } else {
    row[3] = (float)flat[nd.right];
    row[7] = row[8] = row[9] = -1.0f;
    row[11] = (float)nd.axis;
  }
}

// The SAH level loop on boxes that are on the device: rows (12 floats per node; room for 2n-1) and the primitive order into the caller's
// buffers, the number of rows into *n_nodes_out.  `d` owns the scratch.
hipError_t build_levels_sah(hipStream_t stream, Dev& d, uint32_t n, const double* bmin, const double* bmax, int prim_type, float* rows, uint32_t* order_out, uint32_t* n_nodes_out,
                            int* depth_out = nullptr) {
  double *keys[2], *split_pos;
  uint64_t* packed[2];
  uint32_t *order[2], *major[2], *inner, *rank, *n_runs, *split_first, *size, *flat;
  int32_t *seg_of, *axis, *run_key, *next;
  LevelNode* level[2];
  SahAgg *agg, *run_val;
  SahTreeNode* tree;
  const uint32_t max_nodes = 2 * n - 1;
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
  TRY(d.alloc(&split_pos, n));
  TRY(d.alloc(&split_first, n));
  TRY(d.alloc(&agg, n));
  TRY(d.alloc(&run_key, n));
  TRY(d.alloc(&run_val, n));
  TRY(d.alloc(&n_runs, 1));
  TRY(d.alloc(&tree, max_nodes));
  TRY(d.alloc(&size, max_nodes));
  TRY(d.alloc(&flat, max_nodes));
  TRY(d.alloc(&next, max_nodes));

  const SahAggOfPrim agg_of{bmin, bmax};
  unsigned major_bits = 1;
  while ((1ull << major_bits) < (unsigned long long)n) major_bits++;
  size_t t_reduce = 0, t_scan = 0, t_sort1 = 0, t_sort2 = 0;
  {
    auto in = rocprim::make_transform_iterator(order[0], agg_of);
    TRY(rocprim::reduce_by_key(nullptr, t_reduce, seg_of, in, n, run_key, run_val, n_runs, SahAggMerge(), rocprim::equal_to<int32_t>(), stream));
    TRY(rocprim::exclusive_scan(nullptr, t_scan, inner, rank, 0u, n, rocprim::plus<uint32_t>(), stream));
    TRY(rocprim::radix_sort_pairs(nullptr, t_sort1, keys[0], keys[1], packed[0], packed[1], n, 0, 64, stream));
    TRY(rocprim::radix_sort_pairs(nullptr, t_sort2, major[0], major[1], order[0], order[1], n, 0, major_bits, stream));
  }
  char* temp;
  const size_t t_bytes = std::max(std::max(t_reduce, t_scan), std::max(t_sort1, t_sort2));
  TRY(d.alloc(&temp, t_bytes));
  // the bins of the level's nodes: grown to the widest level seen (a level has at most n nodes, the widest one of a real mesh about n / 2)
  SahBin* bins = nullptr;
  size_t bins_cap = 0;
  struct BinsGuard {
    SahBin*& p;
    ~BinsGuard() {
      if (p) (void)hipFree(p);
    }
  } guard{bins};

  const unsigned B = 256;
  hipLaunchKernelGGL(k_iota, dim3((n + B - 1) / B), dim3(B), 0, stream, order[0], seg_of, n);
  const LevelNode root{0u, n - 1u, 0u, -1};
  TRY(hipMemcpyAsync(level[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
  TRY(hipStreamSynchronize(stream));  // `root` is a stack variable

  std::vector<std::pair<uint32_t, uint32_t>> levels;  // (first temporary id, nodes) per level
  uint32_t m = 1, total = 0;
  int cur = 0, ocur = 0;
  while (m > 0) {
    // A level costs a pass over all n primitives whatever it splits.  Binned SAH shrinks a node's centroid range by at least 1/8 per split along the chosen axis, so doubles
    // bound the depth by a few thousand levels (clusters of nearly coincident centroids — tests/test_parity_gpu.py has one — reach several hundred; real meshes 27-31).
    // The cap only keeps a build from turning into a hang should that reasoning miss a case; no traversal could use such a tree anyway (STACK_SIZE <= 64).
    if (levels.size() >= (size_t)kSahMaxLevels) return hipErrorNotSupported;
    levels.emplace_back(total, m);
    {
      auto in = rocprim::make_transform_iterator(order[ocur], agg_of);
      size_t tb = t_bytes;
      TRY(rocprim::reduce_by_key(temp, tb, seg_of, in, n, run_key, run_val, n_runs, SahAggMerge(), rocprim::equal_to<int32_t>(), stream));
    }
    hipLaunchKernelGGL(k_sah_scatter, dim3((n + B - 1) / B), dim3(B), 0, stream, run_key, run_val, n_runs, n, agg);
    const size_t need = (size_t)m * 3 * kSahBins;
    if (need > bins_cap) {
      TRY(hipStreamSynchronize(stream));
      if (bins) (void)hipFree(bins);
      bins = nullptr;
      bins_cap = 0;
      const size_t ask = std::max(need, std::min<size_t>((size_t)n * 3 * kSahBins, need * 2));
      TRY(hipMalloc((void**)&bins, ask * sizeof(SahBin)));
      bins_cap = ask;
    }
    hipLaunchKernelGGL(k_sah_bins_init, dim3((unsigned)((need + B - 1) / B)), dim3(B), 0, stream, bins, (uint32_t)need);
    hipLaunchKernelGGL(k_sah_bins, dim3((n + kSahChunk - 1) / kSahChunk), dim3(256), 0, stream, order[ocur], seg_of, agg, bmin, bmax, n, bins);
    hipLaunchKernelGGL(k_sah_decide, dim3((m + B - 1) / B), dim3(B), 0, stream, level[cur], agg, bins, m, tree, axis, split_pos, inner, split_first);
    {
      size_t tb = t_bytes;
      TRY(rocprim::exclusive_scan(temp, tb, inner, rank, 0u, m, rocprim::plus<uint32_t>(), stream));
    }
    uint32_t tail[2];
    TRY(hipMemcpyAsync(&tail[0], rank + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    TRY(hipMemcpyAsync(&tail[1], inner + (m - 1), 4, hipMemcpyDeviceToHost, stream));
    TRY(hipStreamSynchronize(stream));
    const uint32_t n_inner = tail[0] + tail[1];
    total += m;
    if (n_inner == 0) break;
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
    hipLaunchKernelGGL(k_sah_find_split, dim3((n + B - 1) / B), dim3(B), 0, stream, order[ocur], seg_of, inner, axis, split_pos, bmin, bmax, n, split_first);
    hipLaunchKernelGGL(k_sah_children, dim3((m + B - 1) / B), dim3(B), 0, stream, level[cur], inner, rank, split_first, m, total, level[cur ^ 1], tree);
    hipLaunchKernelGGL(k_sah_descend, dim3((n + B - 1) / B), dim3(B), 0, stream, inner, rank, split_first, n, seg_of);
    cur ^= 1;
    m = 2 * n_inner;
  }
  for (size_t l = levels.size(); l-- > 0;)
    hipLaunchKernelGGL(k_sah_sizes, dim3((levels[l].second + B - 1) / B), dim3(B), 0, stream, tree, levels[l].first, levels[l].second, size);
  const uint32_t zero = 0u;
  const int32_t none = -1;
  TRY(hipMemcpyAsync(flat, &zero, 4, hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(next, &none, 4, hipMemcpyHostToDevice, stream));
  for (size_t l = 0; l < levels.size(); l++)
    hipLaunchKernelGGL(k_sah_flat, dim3((levels[l].second + B - 1) / B), dim3(B), 0, stream, tree, levels[l].first, levels[l].second, size, flat, next);
  hipLaunchKernelGGL(k_sah_rows, dim3((total + B - 1) / B), dim3(B), 0, stream, tree, total, flat, next, prim_type, rows);
  TRY(hipGetLastError());
  TRY(hipMemcpyAsync(order_out, order[ocur], (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  TRY(hipStreamSynchronize(stream));  // (`zero` / `none` are stack variables; the bins die with the guard)
  *n_nodes_out = total;
  if (depth_out) *depth_out = (int)levels.size() - 1;  // inner nodes on the longest root-to-leaf path
  return hipSuccess;
}

hipError_t build_sah_on_device(hipStream_t stream, uint32_t n, const double* h_bmin, const double* h_bmax, int prim_type, float* h_rows, int64_t* h_order, size_t* n_nodes_out) {
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
  uint32_t total = 0;
  TRY(build_levels_sah(stream, d, n, bmin, bmax, prim_type, rows, order, &total));
  std::vector<uint32_t> ord(n);
  TRY(hipMemcpyAsync(h_rows, rows, 12 * (size_t)total * sizeof(float), hipMemcpyDeviceToHost, stream));
  TRY(hipMemcpyAsync(ord.data(), order, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  TRY(hipStreamSynchronize(stream));
  for (uint32_t i = 0; i < n; i++) h_order[i] = (int64_t)ord[i];
  *n_nodes_out = total;
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
__global__ void k_rows_inner_flag(const float* __restrict__ rows, uint32_t nn, uint32_t* __restrict__ inner, uint32_t* __restrict__ multi) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn) return;
  const bool in = rows[12 * (size_t)i + 7] == -1.0f;  // prim_type -1 = inner (bvhBuilder.js:45,49)
  inner[i] = in ? 1u : 0u;
  if (multi) multi[i] = (!in && rows[12 * (size_t)i + 9] != 1.0f) ? 1u : 0u;  // a leaf of the SAH builder that holds several triangles
}
// {first triangle, count} of the leaves that hold anything but one triangle (csrc/ptmi_device.h: REF_MULTI), numbered in node order
__global__ void k_rows_leaf_table(const float* __restrict__ rows, uint32_t nn, const uint32_t* __restrict__ multi, const uint32_t* __restrict__ mrank, int2* __restrict__ table) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn || !multi[i]) return;
  table[mrank[i]] = make_int2((int)rows[12 * (size_t)i + 8], (int)rows[12 * (size_t)i + 9]);
}
__global__ void k_rows_to_pairs(const float* __restrict__ rows, uint32_t nn, const uint32_t* __restrict__ inner, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ multi,
                                const uint32_t* __restrict__ mrank, float* __restrict__ pairs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nn || !inner[i]) return;
  const float* nd = rows + 12 * (size_t)i;
  const uint32_t L = i + 1u, R = (uint32_t)nd[3];
  const float *nl = rows + 12 * (size_t)L, *nr = rows + 12 * (size_t)R;
  auto ref_of = [&](uint32_t j, const float* row) -> uint32_t {  // pair index | REF_LEAF | prim_id (one triangle) | REF_LEAF | REF_MULTI | leaf-table index
    if (inner[j]) return rank[j];
    if (multi && multi[j]) return 0xc0000000u | mrank[j];
    return 0x80000000u | (uint32_t)row[8];
  };
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

extern "C" int ptmi_build_bvh_sah_device(ptmi_ctx* ctx, size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out, int64_t* order_out, size_t* n_nodes_out) {
  if (!ctx) return PTMI_ERR_INVALID_ARG;
  if (!n_nodes_out) return ptmi_ctx_fail(ctx, PTMI_ERR_INVALID_ARG, "ptmi_build_bvh_sah_device: n_nodes_out is null");
  *n_nodes_out = 0;
  if (n_prims == 0) return PTMI_OK;
  if (!bmin || !bmax || !nodes_out || !order_out) return ptmi_ctx_fail(ctx, PTMI_ERR_INVALID_ARG, "ptmi_build_bvh_sah_device: null argument");
  if (n_prims > (size_t)1 << 27) return ptmi_ctx_fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_build_bvh_sah_device: more than 2^27 primitives");
  int r = ptmi_ctx_set_device(ctx);
  if (r) return r;
  void* s = nullptr;
  ptmi_stream(ctx, &s);
  hipError_t e;
  try {
    e = build_sah_on_device((hipStream_t)s, (uint32_t)n_prims, bmin, bmax, prim_type, nodes_out, order_out, n_nodes_out);
  } catch (...) {
    return ptmi_ctx_fail(ctx, PTMI_ERR_NO_MEMORY, "ptmi_build_bvh_sah_device: host allocation failed");
  }
  if (e == hipErrorNotSupported) return ptmi_ctx_fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_build_bvh_sah_device: the SAH tree of these boxes is deeper than 4096 levels (no STACK_SIZE <= 64 can traverse it)");
  if (e != hipSuccess) return ptmi_ctx_fail(ctx, e == hipErrorOutOfMemory ? PTMI_ERR_NO_MEMORY : PTMI_ERR_DEVICE, hipGetErrorString(e));
  return PTMI_OK;
}

// ---- for ptmi.hip (ptmi_build_scene_bvh / prepare_scene): everything stays on the device -------------------------------------------
// Builds the reference's median-split BVH (sah = 0) or its binned-SAH one (sah = 1) over the n triangles at d_tris (24 f32 each, upload order):
// boxes, level loop, rows into d_rows (12 f32 x *n_nodes_out, room for 2n-1), the triangles in leaf order into d_tris_out.  meshes / transforms are host arrays (small).  *bad_tri = first
// triangle whose mesh / transform index is out of range (0xffffffff = none).  Returns a hipError_t.
int ptmi_bvhdev_build_scene(void* stream_, const float* d_tris, uint32_t n, const int32_t* h_meshes, int n_meshes, const float* h_xforms, int n_xforms, float* d_rows,
                            float* d_tris_out, int* depth_out, uint32_t* bad_tri, int sah, uint32_t* n_nodes_out) {
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
    *n_nodes_out = 2 * n - 1;
    if (sah) TRY(build_levels_sah(stream, d, n, bmin, bmax, 2, d_rows, order, n_nodes_out, depth_out));  // BVH.generate_bvh_heirarchy_SAH (bvhNode.js:108-283), the opt-in
    else TRY(build_levels(stream, d, n, bmin, bmax, 2, d_rows, order, depth_out));
    hipLaunchKernelGGL(k_permute_triangles, dim3((6 * n + B - 1) / B), dim3(B), 0, stream, reinterpret_cast<const float4*>(d_tris), order, n,
                       reinterpret_cast<float4*>(d_tris_out));
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(stream));  // the scratch dies with `d`
    return (int)hipSuccess;
  } catch (...) {
    return (int)hipErrorOutOfMemory;
  }
}

// pair64 records (16 f32 per inner node, (nn-1)/2 of them) from device-resident rows; with d_leaf_table != null (room for (nn+1)/2 entries) leaves
// of several triangles are allowed (SAH trees) and *n_multi receives the number of table entries written
int ptmi_bvhdev_make_pairs(void* stream_, const float* d_rows, uint32_t nn, float* d_pairs, int* d_leaf_table, uint32_t* n_multi) {
  hipStream_t stream = (hipStream_t)stream_;
  try {
    Dev d;
    uint32_t *inner, *rank, *multi = nullptr, *mrank = nullptr;
    char* temp;
    TRY(d.alloc(&inner, nn));
    TRY(d.alloc(&rank, nn));
    if (d_leaf_table) {
      TRY(d.alloc(&multi, nn));
      TRY(d.alloc(&mrank, nn));
    }
    size_t tb = 0;
    TRY(rocprim::exclusive_scan(nullptr, tb, inner, rank, 0u, nn, rocprim::plus<uint32_t>(), stream));
    TRY(d.alloc(&temp, tb));
    const unsigned B = 256;
    hipLaunchKernelGGL(k_rows_inner_flag, dim3((nn + B - 1) / B), dim3(B), 0, stream, d_rows, nn, inner, multi);
    size_t tb1 = tb;
    TRY(rocprim::exclusive_scan(temp, tb1, inner, rank, 0u, nn, rocprim::plus<uint32_t>(), stream));
    if (d_leaf_table) {
      size_t tb2 = tb;
      TRY(rocprim::exclusive_scan(temp, tb2, multi, mrank, 0u, nn, rocprim::plus<uint32_t>(), stream));
      hipLaunchKernelGGL(k_rows_leaf_table, dim3((nn + B - 1) / B), dim3(B), 0, stream, d_rows, nn, multi, mrank, reinterpret_cast<int2*>(d_leaf_table));
      uint32_t tail[2];
      TRY(hipMemcpyAsync(&tail[0], mrank + (nn - 1), 4, hipMemcpyDeviceToHost, stream));
      TRY(hipMemcpyAsync(&tail[1], multi + (nn - 1), 4, hipMemcpyDeviceToHost, stream));
      TRY(hipStreamSynchronize(stream));
      if (n_multi) *n_multi = tail[0] + tail[1];
    } else if (n_multi) {
      *n_multi = 0;
    }
    hipLaunchKernelGGL(k_rows_to_pairs, dim3((nn + B - 1) / B), dim3(B), 0, stream, d_rows, nn, inner, rank, multi, mrank, d_pairs);
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(stream));
    return (int)hipSuccess;
  } catch (...) {
    return (int)hipErrorOutOfMemory;
  }
}

