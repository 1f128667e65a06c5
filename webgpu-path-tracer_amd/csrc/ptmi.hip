// ptmi.hip — libptmi.so: context management, scene upload/validation/digests, wavefront scheduling and
// the C ABI of include/ptmi.h.  gfx950 (MI355X) only; build: see webgpu-path-tracer_amd/_build.py.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is loaded on first multi-device use (see Rccl below)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptmi.h"
#include "ptmi_kernels.h"

using namespace ptmi;

// csrc/ptmi_bvh_device.hip
int ptmi_bvhdev_build_scene(void* stream, const float* d_tris, uint32_t n, const int32_t* h_meshes, int n_meshes, const float* h_xforms, int n_xforms, float* d_rows,
                            float* d_tris_out, int* depth_out, uint32_t* bad_tri, int sah, uint32_t* n_nodes_out);
int ptmi_bvhdev_make_pairs(void* stream, const float* d_rows, uint32_t nn, float* d_pairs, int* d_leaf_table, uint32_t* n_multi);

namespace {

thread_local std::string g_create_error;

struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (bytes == 0) return hipSuccess;
    size_t ask = bytes;
#ifdef PTMI_TEST_HOOKS  // the tests' own build of the library (_build.build_testhooks): pretend the board is smaller — through a hipMalloc that really fails
    if (const char* lim = getenv("PTMI_TEST_ALLOC_LIMIT"))
      if (bytes > strtoull(lim, nullptr, 10)) ask = (size_t)1 << 60;
#endif
    hipError_t e = hipMalloc(&p, ask);
    if (e == hipSuccess) cap = bytes;
    else {
      p = nullptr;
      (void)hipGetLastError();  // the failure is reported through the return value; do not leave it behind as the runtime's "last error"
    }
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

enum TimerTag { T_PRIMS = 0, T_SHADE = 1, T_GENERATE = 2, T_RENDER = 3, T_BVH = 4, T_ACCUM = 5, T_TAIL = 6 };

}  // namespace

// One persistent host thread per peer of a multi-device context: the calls that may block (allocation, the occasional queue-length
// readback of a long render) run concurrently on all devices without starting a thread per call (ptmi_render_frame is per displayed frame).
struct PeerWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  int result = 0;
  bool has_job = false, done = false, quit = false;
  void loop() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [this] { return has_job || quit; });
      if (quit) return;
      std::function<int()> j = std::move(job);
      has_job = false;
      lk.unlock();
      const int r = j();
      lk.lock();
      result = r;
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    std::lock_guard<std::mutex> lk(m);
    if (!th.joinable()) th = std::thread([this] { loop(); });
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
    return result;
  }
  ~PeerWorker() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

// Tuning knobs (PTMI_* environment variables): read ONCE, when the context is created (ptmi_reload_tuning reads them again — tests and A/B
// scripts that change a variable under a live context call it); the render path itself never touches the environment.
struct Tuning {
  int lds_stack = 10;          // PTMI_LDS_STACK: traversal stack entries per lane kept in LDS (deeper ones: per-wave global spill area)
  bool noabort = true;         // PTMI_NOABORT=0: keep the literal stack discipline even where Q7's abort cannot trigger
  int waves_per_cu = 0;        // PTMI_WAVES_PER_CU: k_bvh's grid (0 = auto)
  int bvh_teams = 16;          // PTMI_BVH_TEAMS: claim counters of k_bvh
  int refill = kRefillThreshold, leaf_batch = kLeafBatch, bvh_range = (int)kBvhRange;  // PTMI_REFILL, PTMI_LEAF_BATCH, PTMI_BVH_RANGE
  int tail_waves_per_cu = 0;   // PTMI_TAIL_WAVES_PER_CU (0 = 16, or 24 for the 6-wave build)
  bool tail6 = true;           // PTMI_TAIL6=0: never the 80-VGPR build of k_tail
  int tail_park = 16;          // PTMI_TAIL_PARK: k_tail's tree walk parks its last lanes once fewer than this many are left in it (trees of >= 12 levels only; 0 = never)
  int bvh_carry = 32;          // PTMI_BVH_CARRY: iterations a k_bvh wave goes on after the queue is exhausted before it carries its unfinished rays into the next
                               // step's queue (Carry, ptmi_device.h); 0 = never (every launch traces its longest ray to the end)
  int bvh_carry_slots = 1 << 18;  // PTMI_BVH_CARRY_SLOTS: the queues' carry prefix
  int bvh_carry_last = 0;      // PTMI_BVH_CARRY_LAST: the last this-many steps carry nothing over (measured: 0 is best — the drain launch costs 1.5-2.5 ms either way)
  int bvh_carry_min_paths = 4 << 20, bvh_carry_min_depth = 12;  // PTMI_BVH_CARRY_MIN_PATHS / _MIN_DEPTH: batches and trees below these are traced without carrying (tests: 0)
  int sort = -1;               // PTMI_SORT: k_shade sorts its chunks by material class (-1 = when the scene has more than one)
  int shade_blocks_per_cu = 0; // PTMI_SHADE_BLOCKS_PER_CU (0 = from the variant's occupancy)
  int tail_limit = -1;         // PTMI_TAIL_LIMIT: k_tail takes queues of at most this many slots (-1 = kTailLimitFirst / kTailLimitLater, 0 = never)
  bool render_ahead = true;    // PTMI_RENDER_AHEAD=0
  int path_budget_log2 = 30;   // PTMI_PATH_BUDGET_LOG2: paths per wavefront pass with frames_in_flight = auto (round 5: 29 -> 30)
  int placement_tries = 6;     // PTMI_PLACEMENT_TRIES (round 5: 4 -> 6 — the sets now differ, the losers staying allocated during the search: best of 8 ran 0.7 % ahead of best of 4)
  bool debug_placement = false;
};

struct ptmi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  ptmi_params prm;
  Tuning tun;
  int num_cus = 256;

  // host copies of the uploaded arrays (reference layouts)
  std::vector<float> h_spheres, h_quads, h_xforms, h_mats, h_bvh;
  size_t n_tris_uploaded = 0;  // the triangles go straight to the device (d_tris): the host never needs their values again
  std::vector<int32_t> h_meshes;
  bool scene_dirty = true;

  DBuf d_quad_unit_n;
  // ptmi_build_scene_bvh: the BVH rows (binding 9's layout) live on the device only — no host copy, pair64 made there too
  DBuf d_bvh_rows;
  bool bvh_on_device = false;
  size_t bvh_dev_prims = 0;  // triangles the device-resident tree was built over
  size_t bvh_dev_nodes = 0;  // its rows: 2 x triangles - 1 from the median builder, fewer from the SAH one (leaves of several triangles)
  bool bvh_dev_sah = false;  // built by ptmi_build_scene_bvh_sah
  bool bvh_dev_stale = false;  // triangles / meshes / transforms were uploaded after the build: its boxes and leaf order describe another scene
  int bvh_dev_depth = 0;
  DBuf d_spheres, d_sphere_info, d_quads, d_quad_mat, d_tris, d_pretri, d_trinorm, d_meshes, d_xforms, d_mats, d_pairs, d_leaf_table;
  DevScene S{};
  int bvh_depth = 0;             // max number of inner nodes on a root-to-leaf path
  bool has_unknown_material = false;
  int material_classes = 0;      // distinct shade bins among the materials: k_shade sorts only when > 1
  int shade_blocks_per_cu[16] = {0};  // per k_shade variant: resident 256-thread blocks per CU (0 = not asked yet)

  int W = 0, H = 0;
  DBuf d_fb_own;
  float4* fb = nullptr;
  size_t fb_bytes = 0;
  int rank = 0, world = 1, tile = 64;

  size_t path_cap = 0;
  bool pixsum_alloc = false;
  size_t slot_cap = 0;  // slots per queue buffer (paths + room for the holes k_shade's regions leave)
  DBuf d_q0[2], d_q1[2], d_q2[2], d_tp[2], d_hm[2];  // slot-indexed live state and hit records, ping-pong
  DBuf d_uv, d_acc, d_pixsum, d_touched, d_ctl, d_totals, d_scratch, d_spill, d_heads;
  DBuf d_carry[2];  // Carry: the pools of saved traversal state, ping-pong like the queues
  int ctl_cap = 0;

  // Render-ahead of ptmi_render_frame (see there): per-frame colours of frames [frame0, frame0 + count) sit in d_acc,
  // the first `next` of them are already in the framebuffer.
  struct Ahead {
    bool valid = false;
    float view[16];
    uint32_t frame0 = 0;
    int count = 0, next = 0;
    RenderConst rc;
    uint32_t grid = 1;
  } ahead;
  bool have_last_frame = false;
  uint32_t last_frame = 0;
  float last_view[16];
  int static_streak = 0;  // consecutive ptmi_render_frame calls with the same view, frame numbers +1, no reset
  bool placement_pending = false;  // ensure_paths has just (re)allocated the queue arrays of a batch worth a placement search: render_batch runs it
  bool interactive = false;  // inside ptmi_render_frame: its batches grow 8 -> 64 frames while the user watches — no placement search (40 ms each)
  int ahead_batch = 8;

  // Multi-device context (ptmi_create_multi): this object is local device 0, `peers` are the contexts of local devices
  // 1..n-1 (plain single-device contexts).  The caller's shard (ptmi_set_shard) is subdivided among the n of them.
  bool multi = false;
  bool shares_device = false;  // another shard of the same multi-device context lives on this GPU
  std::vector<ptmi_ctx*> peers;
  PeerWorker* worker = nullptr;  // a peer's host thread (created on first use)
  int proc_rank = 0, proc_world = 1, proc_tile = 4032;  // (63 waves of pixels; not 4096: see dist.py — a rank's pixel count must not be a large power of two)
  bool use_rccl = false;
  bool use_gather = false;        // the default collective of a multi-device context: every device's OWN tiles are copied into place on the root (1/N of the bytes, no arithmetic)
  bool root_reads = false;        // (a peer's field) the root device can read this device's memory directly: peer access is on, or it is the same GPU
  uint64_t gather_bytes = 0;      // bytes that crossed between GPUs in the last tile gather
  int test_rccl_fail = 0;  // -DPTMI_TEST_HOOKS builds: PTMI_TEST_RCCL_FAIL=init|reduce|mid read at ptmi_create_multi — pretend ncclCommInitAll (1) / ncclGroupStart (2) failed, or
                           // a call in the middle of the reduce's group after the first ncclReduce was enqueued (3); 0 in the product build
  std::vector<ncclComm_t> comms;  // one per local device, same order as {this, peers...}
  int reduce_mode = 0;            // ptmi_stats.reduce_mode: how the multi-device sum runs (0 single device, 1 RCCL, 2 peer copies + add, 3 the same as a FALLBACK)
  int peer_links = 0;             // directed device pairs (root <-> peer) with peer access enabled
  std::string reduce_info;        // ptmi_reduce_info
  DBuf d_fb_gather, d_fb_stage;   // on this device: the sum of all local accumulation buffers / a peer's buffer in transit

  bool batch_enqueued = false;  // render_batch: has anything been put on the stream yet? (a failure before that point may be retried)
  bool counters = false;
  int timing = 0;  // 0 off, 1 every kernel, 2 only k_bvh (the dominant kernel) — see ptmi_set_timing
  ptmi_stats stats{};
  std::vector<hipEvent_t> ev_pool;
  struct Span {
    hipEvent_t a, b;
    int tag;
  };
  std::vector<Span> spans;
};

namespace {

int fail(ptmi_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}
#define HIP_TRY(c, expr)                                                                              \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return fail((c), _e == hipErrorOutOfMemory ? PTMI_ERR_NO_MEMORY : PTMI_ERR_DEVICE,              \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
  } while (0)

hipEvent_t get_event(ptmi_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
struct ScopedSpan {  // records a begin/end event pair around launches when timing is on
  ptmi_ctx* c;
  ptmi_ctx::Span s{};
  bool on;
  static bool wanted(int timing, int tag) {  // ptmi_set_timing: 1 = everything, 2..5 = one kernel only
    return timing == 1 || (timing == 2 && tag == T_BVH) || (timing == 3 && tag == T_SHADE) || (timing == 4 && tag == T_GENERATE) || (timing == 5 && tag == T_ACCUM) || (timing == 6 && tag == T_TAIL);
  }
  ScopedSpan(ptmi_ctx* c_, int tag) : c(c_), on(wanted(c_->timing, tag)) {
    if (!on) return;
    s.tag = tag;
    s.a = get_event(c);
    s.b = get_event(c);
    (void)hipEventRecord(s.a, c->stream);
  }
  ~ScopedSpan() {
    if (!on) return;
    (void)hipEventRecord(s.b, c->stream);
    c->spans.push_back(s);
  }
};
void drain_spans(ptmi_ctx* c) {  // call after the stream is idle
  for (auto& s : c->spans) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      if (s.tag == T_PRIMS) c->stats.prims_ms += ms, c->stats.intersect_ms += ms;
      else if (s.tag == T_BVH) c->stats.bvh_ms += ms, c->stats.intersect_ms += ms;
      else if (s.tag == T_SHADE) c->stats.shade_ms += ms;
      else if (s.tag == T_GENERATE) c->stats.generate_ms += ms, c->stats.other_ms += ms;
      else if (s.tag == T_ACCUM) c->stats.accumulate_ms += ms, c->stats.other_ms += ms;
      else if (s.tag == T_TAIL) c->stats.tail_ms += ms;
      else c->stats.render_ms += ms;
    }
    c->ev_pool.push_back(s.a);
    c->ev_pool.push_back(s.b);
  }
  c->spans.clear();
}

size_t stride_of(int which) {
  switch (which) {
    case PTMI_BUF_SPHERES: return 32;
    case PTMI_BUF_QUADS: return 80;
    case PTMI_BUF_TRIANGLES: return 96;
    case PTMI_BUF_MESHES: return 16;
    case PTMI_BUF_TRANSFORMS: return 128;
    case PTMI_BUF_MATERIALS: return 64;
    case PTMI_BUF_BVH: return 48;
  }
  return 0;
}

bool id_from_float(float f, int n, int* out) {  // i32(f32) of an index field, range checked
  if (!(f >= 0.0f) || !(f < 2147483000.0f)) return false;
  int v = (int)f;
  if (v >= n) return false;
  *out = v;
  return true;
}

// Validate every index the kernels will dereference and build the digests (see DevScene).
int prepare_scene(ptmi_ctx* c) {
  if (!c->scene_dirty) return PTMI_OK;
  const int n_sph = (int)(c->h_spheres.size() / 8), n_quad = (int)(c->h_quads.size() / 20), n_tri = (int)c->n_tris_uploaded;
  const int n_mesh = (int)(c->h_meshes.size() / 4), n_xf = (int)(c->h_xforms.size() / 32), n_mat = (int)(c->h_mats.size() / 16);
  const int n_node = c->bvh_on_device ? (int)c->bvh_dev_nodes : (int)(c->h_bvh.size() / 12);
  char msg[256];

  c->has_unknown_material = false;
  for (int i = 0; i < n_mat; i++) {
    float ty = c->h_mats[16 * (size_t)i + 14];
    if (!(ty == 0.0f || ty == 1.0f || ty == 2.0f || ty == 3.0f)) c->has_unknown_material = true;
  }
  // material word = id | shade bin << 28 (ptmi_device.h)
  std::vector<int32_t> mat_word((size_t)n_mat);
  uint32_t classes = 0;  // shade bins of the materials some primitive actually refers to (the table may hold unused ones)
  for (int i = 0; i < n_mat; i++) {
    float ty = c->h_mats[16 * (size_t)i + 14];
    int bin = (ty == 0.0f) ? BIN_LAMBERTIAN : (ty == 1.0f) ? BIN_MIRROR : (ty == 2.0f) ? BIN_GLASS : (ty == 3.0f) ? BIN_ISOTROPIC : BIN_OTHER;
    mat_word[i] = i | (bin << HITMAT_BIN_SHIFT);
  }
  auto use_class = [&](int m) { classes |= 1u << ((uint32_t)mat_word[m] >> HITMAT_BIN_SHIFT); };
  if (n_mat >= (1 << HITMAT_BIN_SHIFT)) return fail(c, PTMI_ERR_UNSUPPORTED, "more than 2^28 materials");
  std::vector<int32_t> sphere_info(2 * (size_t)n_sph), quad_mat((size_t)n_quad);
  for (int i = 0; i < n_sph; i++) {
    int m;
    if (!id_from_float(c->h_spheres[8 * (size_t)i + 6], n_mat, &m)) {
      snprintf(msg, sizeof msg, "sphere %d: material_id out of range [0,%d)", i, n_mat);
      return fail(c, PTMI_ERR_BAD_SCENE, msg);
    }
    float medium = c->h_mats[16 * (size_t)m + 14];
    sphere_info[2 * (size_t)i] = mat_word[m];
    use_class(m);
    sphere_info[2 * (size_t)i + 1] = (medium < 3.0f) ? 0 : 1;  // hitRay.wgsl:8-9
  }
  int light = -1;
  for (int i = 0; i < n_quad; i++) {
    int m;
    if (!id_from_float(c->h_quads[20 * (size_t)i + 19], n_mat, &m)) {
      snprintf(msg, sizeof msg, "quad %d: material_id out of range [0,%d)", i, n_mat);
      return fail(c, PTMI_ERR_BAD_SCENE, msg);
    }
    quad_mat[i] = mat_word[m];
    use_class(m);
    if (light < 0 && c->h_mats[16 * (size_t)m + 8] > 0.0f) light = i;  // common.wgsl:258-269
  }
  for (int i = 0; i < n_mesh; i++) {
    const int32_t* me = &c->h_meshes[4 * (size_t)i];
    if (me[2] < 0 || me[2] >= n_xf || me[3] < 0 || me[3] >= n_mat) {
      snprintf(msg, sizeof msg, "mesh %d: global_id %d / material_id %d out of range (transforms %d, materials %d)", i, me[2], me[3], n_xf, n_mat);
      return fail(c, PTMI_ERR_BAD_SCENE, msg);
    }
  }
  // (the pretri digest is computed on the device from the uploaded triangles: k_pretri_digest, below)
  std::vector<int32_t> mesh_matword((size_t)n_mesh);
  for (int i = 0; i < n_mesh; i++) {
    mesh_matword[i] = mat_word[c->h_meshes[4 * (size_t)i + 3]];
    use_class(c->h_meshes[4 * (size_t)i + 3]);
  }
  c->material_classes = __builtin_popcount(classes);
  // Tree check + pair64 digest.  Children always have larger indices than their parent (left = i+1,
  // right > i+1), so a walk from the root terminates; "every node reached at most once" excludes
  // shared subtrees (which could make a traversal exponentially long).
  std::vector<float> pairs;
  std::vector<int32_t> leaf_table;
  float root_lo[4] = {0, 0, 0, 0}, root_hi[4] = {0, 0, 0, 0};
  c->bvh_depth = 0;
  bool pairs_on_device = false;
  if (n_node > 0 && c->bvh_on_device) {
    // The tree came out of this library's own builder (ptmi_build_scene_bvh) over the triangles that are on the device: a tree by
    // construction, its leaves a partition of the triangles [0, n) — all that has to hold is that those triangles are still the uploaded ones.
    if ((size_t)n_tri != c->bvh_dev_prims) {
      snprintf(msg, sizeof msg, "the device-resident BVH was built over %zu triangles, %d are uploaded now (build again, or upload a BVH)", c->bvh_dev_prims, n_tri);
      return fail(c, PTMI_ERR_BAD_SCENE, msg);
    }
    if (c->bvh_dev_stale)  // same count, other content: the tree's world-space boxes and leaf order belong to what was there before
      return fail(c, PTMI_ERR_BAD_SCENE, "triangles, meshes or transforms were uploaded after ptmi_build_scene_bvh: the device-resident BVH describes the earlier scene (build again, or upload a BVH)");
    c->bvh_depth = c->bvh_dev_depth;
    const size_t n_inner = (size_t)(n_node - 1) / 2;
    HIP_TRY(c, c->d_pairs.ensure(std::max<size_t>(n_inner * 64, 16)));
    // (a SAH tree's leaves may hold several triangles: those go through the leaf table, made on the device too — at most one entry per leaf)
    if (c->bvh_dev_sah) HIP_TRY(c, c->d_leaf_table.ensure(std::max<size_t>(((size_t)n_node + 1) / 2 * 8, 16)));
    {
      uint32_t n_multi = 0;
      const int e = ptmi_bvhdev_make_pairs((void*)c->stream, c->d_bvh_rows.as<float>(), (uint32_t)n_node, c->d_pairs.as<float>(), c->bvh_dev_sah ? c->d_leaf_table.as<int>() : nullptr, &n_multi);
      if (e) return fail(c, e == (int)hipErrorOutOfMemory ? PTMI_ERR_NO_MEMORY : PTMI_ERR_DEVICE, std::string("pair records on the device: ") + hipGetErrorString((hipError_t)e));
    }
    float r0[12];
    HIP_TRY(c, hipMemcpy(r0, c->d_bvh_rows.p, sizeof r0, hipMemcpyDeviceToHost));
    // the root is pair 0, or the only leaf (of one triangle, or — a SAH tree that never split — of several: leaf-table entry 0)
    const uint32_t rref = n_inner ? 0u : (r0[9] == 1.0f ? (REF_LEAF | (uint32_t)(int)r0[8]) : (REF_LEAF | REF_MULTI | 0u));
    root_lo[0] = r0[0], root_lo[1] = r0[1], root_lo[2] = r0[2];
    memcpy(&root_lo[3], &rref, 4);
    root_hi[0] = r0[4], root_hi[1] = r0[5], root_hi[2] = r0[6];
    pairs_on_device = true;
  } else if (n_node > 0) {
    std::vector<uint8_t> seen((size_t)n_node, 0);  // 1 = inner, 2 = leaf
    std::vector<std::pair<int, int>> st;           // node, inner depth so far
    st.emplace_back(0, 0);
    size_t n_inner = 0;
    while (!st.empty()) {
      auto [i, depth] = st.back();
      st.pop_back();
      if (seen[i]) {
        snprintf(msg, sizeof msg, "bvh: node %d reachable twice (not a tree)", i);
        return fail(c, PTMI_ERR_BAD_SCENE, msg);
      }
      const float* nd = &c->h_bvh[12 * (size_t)i];
      if (nd[7] >= 2.0f && nd[7] < 3.0f) {  // leaf: i32(prim_type) == 2 (hitRay.wgsl:45,56)
        int first = 0, cnt = 0;
        // count in [0, n_tri] and first + count <= n_tri, compared without an addition that could wrap
        const bool cnt_ok = nd[9] >= 0.0f && nd[9] <= (float)n_tri && (cnt = (int)nd[9]) <= n_tri;
        if (!cnt_ok || (cnt > 0 && (!id_from_float(nd[8], n_tri, &first) || cnt > n_tri - first))) {
          snprintf(msg, sizeof msg, "bvh: leaf %d references triangles outside [0,%d)", i, n_tri);
          return fail(c, PTMI_ERR_BAD_SCENE, msg);
        }
        seen[i] = 2;
        c->bvh_depth = std::max(c->bvh_depth, depth);
      } else {
        int right = 0;
        const bool type_ok = nd[7] > -2147483000.0f && nd[7] < 2147483000.0f;  // i32(prim_type) is defined (and != 2)
        const bool axis_ok = nd[11] >= 0.0f && nd[11] < 3.0f;  // i32(axis) in {0, 1, 2}
        if (!type_ok || !axis_ok || !id_from_float(nd[3], n_node, &right) || right <= i + 1 || i + 1 >= n_node) {
          snprintf(msg, sizeof msg, "bvh: inner node %d has right_offset/axis out of range", i);
          return fail(c, PTMI_ERR_BAD_SCENE, msg);
        }
        seen[i] = 1;
        n_inner++;
        st.emplace_back(right, depth + 1);
        st.emplace_back(i + 1, depth + 1);
      }
    }
    std::vector<int32_t> rank((size_t)n_node, -1);
    int32_t r = 0;
    for (int i = 0; i < n_node; i++)
      if (seen[i] == 1) rank[i] = r++;
    // Leaves that hold anything but one triangle (external / SAH trees) go through the leaf table; their entries are numbered
    // first, in node order, so that the pair records below can be written by several host threads.
    std::vector<int32_t> multi_ref;  // per node: index into the leaf table, -1 = not a multi-triangle leaf
    for (int j = 0; j < n_node; j++) {
      if (seen[j] != 2) continue;
      const float* nd = &c->h_bvh[12 * (size_t)j];
      const int cnt = (int)nd[9];
      if (cnt == 1) continue;
      if (multi_ref.empty()) multi_ref.assign((size_t)n_node, -1);
      multi_ref[j] = (int32_t)(leaf_table.size() / 2);
      leaf_table.push_back(cnt > 0 ? (int)nd[8] : 0);
      leaf_table.push_back(cnt);
    }
    auto ref_of = [&](int j) -> uint32_t {  // (every node passed the range checks of the walk above)
      const float* nd = &c->h_bvh[12 * (size_t)j];
      if (seen[j] == 1) return (uint32_t)rank[j];
      if ((int)nd[9] == 1) return REF_LEAF | (uint32_t)(int)nd[8];
      return REF_LEAF | REF_MULTI | (uint32_t)multi_ref[j];
    };
    pairs.assign(16 * n_inner, 0.0f);
    auto fill_pairs = [&](int begin, int end) {
      for (int i = begin; i < end; i++) {
        if (seen[i] != 1) continue;
        const float* nd = &c->h_bvh[12 * (size_t)i];
        const int L = i + 1, R = (int)nd[3];
        const float *nl = &c->h_bvh[12 * (size_t)L], *nr = &c->h_bvh[12 * (size_t)R];
        float* o = &pairs[16 * (size_t)rank[i]];
        uint32_t rl = ref_of(L), rr = ref_of(R);
        int32_t axis = (int)nd[11];
        o[0] = nl[0], o[1] = nl[1], o[2] = nl[2];
        memcpy(&o[3], &rl, 4);
        o[4] = nl[4], o[5] = nl[5], o[6] = nl[6];
        memcpy(&o[7], &rr, 4);
        o[8] = nr[0], o[9] = nr[1], o[10] = nr[2];
        memcpy(&o[11], &axis, 4);
        o[12] = nr[4], o[13] = nr[5], o[14] = nr[6];
        o[15] = 0.0f;
      }
    };
    {
      const int nt = n_node < (1 << 16) ? 1 : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
      std::vector<std::thread> th;
      const int per = (n_node + nt - 1) / nt;
      for (int k = 1; k < nt; k++) th.emplace_back(fill_pairs, std::min(n_node, k * per), std::min(n_node, (k + 1) * per));
      fill_pairs(0, std::min(n_node, per));
      for (auto& t : th) t.join();
    }
    const float* rn = &c->h_bvh[0];
    uint32_t rref = ref_of(0);
    root_lo[0] = rn[0], root_lo[1] = rn[1], root_lo[2] = rn[2];
    memcpy(&root_lo[3], &rref, 4);
    root_hi[0] = rn[4], root_hi[1] = rn[5], root_hi[2] = rn[6];
  }

  auto up = [&](DBuf& d, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = d.ensure(std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, c->stream);
    return e;
  };
  HIP_TRY(c, up(c->d_spheres, c->h_spheres.data(), c->h_spheres.size() * 4));
  HIP_TRY(c, up(c->d_sphere_info, sphere_info.data(), sphere_info.size() * 4));
  HIP_TRY(c, up(c->d_quads, c->h_quads.data(), c->h_quads.size() * 4));
  HIP_TRY(c, up(c->d_quad_mat, quad_mat.data(), quad_mat.size() * 4));
  HIP_TRY(c, c->d_quad_unit_n.ensure(std::max<size_t>((size_t)n_quad * 16, 16)));
  if (n_quad > 0) {
    hipLaunchKernelGGL(k_quad_digest, dim3((unsigned)((n_quad + 63) / 64)), dim3(64), 0, c->stream, c->d_quads.as<float4>(), n_quad, c->d_quad_unit_n.as<float4>());
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, up(c->d_meshes, c->h_meshes.data(), c->h_meshes.size() * 4));
  HIP_TRY(c, c->d_pretri.ensure(std::max<size_t>((size_t)n_tri * 64, 16)));
  HIP_TRY(c, c->d_trinorm.ensure(std::max<size_t>((size_t)n_tri * 48, 16)));
  if (n_tri > 0) {
    // scratch: [first bad triangle index][mesh material words...]
    HIP_TRY(c, c->d_scratch.ensure(16 + (size_t)n_mesh * 4));
    const uint32_t none = 0xffffffffu;
    HIP_TRY(c, hipMemcpyAsync(c->d_scratch.p, &none, 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync((char*)c->d_scratch.p + 16, mesh_matword.data(), (size_t)n_mesh * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pretri_digest, dim3((unsigned)((n_tri + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->d_tris.as<float4>(), n_tri, c->d_meshes.as<int4>(), n_mesh,
                       reinterpret_cast<const int*>((char*)c->d_scratch.p + 16), c->d_pretri.as<float4>(), c->d_trinorm.as<float4>(), c->d_scratch.as<uint32_t>());
    HIP_TRY(c, hipGetLastError());
    uint32_t bad = none;
    HIP_TRY(c, hipMemcpyAsync(&bad, c->d_scratch.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad != none) {
      snprintf(msg, sizeof msg, "triangle %u: mesh_id out of range [0,%d)", bad, n_mesh);
      return fail(c, PTMI_ERR_BAD_SCENE, msg);
    }
  }
  HIP_TRY(c, up(c->d_xforms, c->h_xforms.data(), c->h_xforms.size() * 4));
  HIP_TRY(c, up(c->d_mats, c->h_mats.data(), c->h_mats.size() * 4));
  if (!pairs_on_device) HIP_TRY(c, up(c->d_pairs, pairs.data(), pairs.size() * 4));
  if (!(pairs_on_device && c->bvh_dev_sah)) HIP_TRY(c, up(c->d_leaf_table, leaf_table.data(), leaf_table.size() * 4));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the staging vectors die at scope exit

  DevScene& S = c->S;
  S.spheres = c->d_spheres.as<float4>();
  S.sphere_info = c->d_sphere_info.as<int2>();
  S.quads = c->d_quads.as<float4>();
  S.quad_mat = c->d_quad_mat.as<int>();
  S.quad_unit_n = c->d_quad_unit_n.as<float4>();
  S.tris = c->d_tris.as<float4>();
  S.pretri = c->d_pretri.as<float4>();
  S.trinorm = c->d_trinorm.as<float4>();
  S.meshes = c->d_meshes.as<int4>();
  S.xforms = c->d_xforms.as<float4>();
  S.mats = c->d_mats.as<float4>();
  S.pairs = c->d_pairs.as<float4>();
  S.leaf_table = c->d_leaf_table.as<int2>();
  S.root_lo = make_float4(root_lo[0], root_lo[1], root_lo[2], root_lo[3]);
  S.root_hi = make_float4(root_hi[0], root_hi[1], root_hi[2], root_hi[3]);
  S.n_spheres = n_sph, S.n_quads = n_quad, S.n_tris = n_tri, S.n_meshes = n_mesh, S.n_xforms = n_xf, S.n_mats = n_mat, S.n_nodes = n_node;
  S.light_quad = light;
  S.uniform_gid = n_mesh > 0 ? c->h_meshes[2] : -1;
  for (int i = 1; i < n_mesh; i++)
    if (c->h_meshes[4 * (size_t)i + 2] != S.uniform_gid) S.uniform_gid = -1;
  S.tmin = c->prm.tmin;
  c->scene_dirty = false;
  return PTMI_OK;
}

uint32_t count_local(uint32_t npix, int rank, int world, int tile) {
  uint64_t n = 0;
  uint64_t ntiles = ((uint64_t)npix + tile - 1) / tile;
  for (uint64_t t = (uint64_t)rank; t < ntiles; t += (uint64_t)world) {
    uint64_t b = t * tile, e = std::min<uint64_t>(b + tile, npix);
    n += e - b;
  }
  return (uint32_t)n;
}

Paths paths_of(ptmi_ctx* c, int step, bool with_pixsum);

int ensure_paths(ptmi_ctx* c, size_t npaths, int n_ctl, bool need_pixsum) {
  if (npaths > c->path_cap) {
    // A queue holds at most `npaths` paths plus the holes of k_shade's output regions: at most one region per block
    // (region <= slots/grid/16 rounded up to 512) — 1/8 of the paths plus 1024 slots per possible block is ample.
    const size_t slots = npaths + npaths / 8 + (size_t)c->num_cus * 8 * 1024 + (size_t)c->tun.bvh_carry_slots;  // (+ the carry prefix, Carry)
    c->path_cap = 0;  // a failed allocation below leaves some buffers released: nothing may be assumed allocated then
    c->slot_cap = 0;
    c->pixsum_alloc = false;
    for (int k = 0; k < 2; k++) {
      HIP_TRY(c, c->d_q0[k].ensure(slots * 16));
      HIP_TRY(c, c->d_q1[k].ensure(slots * 16));
      HIP_TRY(c, c->d_q2[k].ensure(slots * 16));
      HIP_TRY(c, c->d_tp[k].ensure(slots * 8));
      HIP_TRY(c, c->d_hm[k].ensure(slots * 4));
    }
    HIP_TRY(c, c->d_uv.ensure(slots * 8));
    HIP_TRY(c, c->d_acc.ensure(npaths * 16));
    HIP_TRY(c, c->d_touched.ensure(npaths));
    c->path_cap = npaths;
    c->slot_cap = slots;
    c->pixsum_alloc = false;
    // Batches worth a placement search (placement_search below): >= 16 Mi slots — smaller ones belong to k_tail or last microseconds —, sets of at most 32 GB (configs[2]'s
    // 256-frame batches — 72 GB per set — would gain too, k_shade 32.2 ms per step instead of 33.5-34.8, but allocating five more such sets takes the runtime 10 s:
    // profiles/r05_placement_dry_run.txt), not from ptmi_render_frame, not where shards share the GPU or the board could not hold two sets.
    const int tries = c->tun.placement_tries;
    c->stats.placement_sets = 0;
    c->stats.placement_ms = 0.0;
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0, (void)hipGetLastError();
    // (the search itself runs in render_batch, on the batch that asked for these buffers: placement_search)
    c->placement_pending = tries > 1 && !c->interactive && !c->shares_device && slots >= ((size_t)1 << 24) && slots * 120 <= ((size_t)32 << 30) && mem_free >= slots * 120 * 2;
  }
  if (need_pixsum && !c->pixsum_alloc) {
    HIP_TRY(c, c->d_pixsum.ensure(c->path_cap * 16));
    c->pixsum_alloc = true;
  }
  if (n_ctl > c->ctl_cap) {
    HIP_TRY(c, c->d_ctl.ensure((size_t)n_ctl * sizeof(StepCtl)));
    c->ctl_cap = n_ctl;
  }
  HIP_TRY(c, c->d_heads.ensure(kMaxTeams * kHeadStride * sizeof(uint32_t)));
  if (!c->d_totals.p) {
    HIP_TRY(c, c->d_totals.ensure(kTotalsBytes));  // 16 persistent counters, then the tally lines of the batch in flight
    HIP_TRY(c, hipMemsetAsync(c->d_totals.p, 0, kTotalsBytes, c->stream));
  }
  return PTMI_OK;
}

// Path state as step `step` sees it: `in` = buffers (step & 1), `out` = the other pair.
Paths paths_of(ptmi_ctx* c, int step, bool with_pixsum) {
  Paths P;
  const int a = step & 1, b = a ^ 1;
  P.in = Slots{c->d_q0[a].as<float4>(), c->d_q1[a].as<float4>(), c->d_q2[a].as<float4>()};
  P.out = Slots{c->d_q0[b].as<float4>(), c->d_q1[b].as<float4>(), c->d_q2[b].as<float4>()};
  P.hin = HitBuf{c->d_tp[a].as<float2>(), c->d_hm[a].as<uint32_t>()};
  P.hout = HitBuf{c->d_tp[b].as<float2>(), c->d_hm[b].as<uint32_t>()};
  P.uv = c->d_uv.as<float2>();
  P.acc = c->d_acc.as<float4>();
  P.pixsum = with_pixsum ? c->d_pixsum.as<float4>() : nullptr;
  P.touched = with_pixsum ? nullptr : c->d_touched.as<uint8_t>();  // NUM_SAMPLES == 1: acc[pid] is written lazily
  P.cap = (uint32_t)c->slot_cap;
  return P;
}

// the k_shade instance for a parameter combination (occupancy queries)
const void* shade_kernel(bool is, bool so, bool cn, bool mu) {
  if (!is && !mu) {  // progressive mode without importance sampling: the 80-VGPR build (k_shade6)
    if (so) return cn ? reinterpret_cast<const void*>(&k_shade6<true, true>) : reinterpret_cast<const void*>(&k_shade6<true, false>);
    return cn ? reinterpret_cast<const void*>(&k_shade6<false, true>) : reinterpret_cast<const void*>(&k_shade6<false, false>);
  }
#define PTMI_SK(I, S, C, M) \
  if (is == I && so == S && cn == C && mu == M) return reinterpret_cast<const void*>(&k_shade<I, S, C, M>)
  PTMI_SK(false, false, false, true); PTMI_SK(false, false, true, true);  // (IS = MULTI = false is k_shade6 above: those four are never instantiated)
  PTMI_SK(false, true, false, true); PTMI_SK(false, true, true, true);
  PTMI_SK(true, false, false, false); PTMI_SK(true, false, false, true); PTMI_SK(true, false, true, false); PTMI_SK(true, false, true, true);
  PTMI_SK(true, true, false, false); PTMI_SK(true, true, false, true); PTMI_SK(true, true, true, false); PTMI_SK(true, true, true, true);
#undef PTMI_SK
  return nullptr;
}

int stack_alloc_for(const ptmi_ctx* c) { return std::max(1, std::min(c->prm.stack_size, std::max(c->bvh_depth, 1))); }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
void load_tuning(ptmi_ctx* c) {
  Tuning t;
  t.lds_stack = std::max(1, env_int("PTMI_LDS_STACK", t.lds_stack));
  t.noabort = env_int("PTMI_NOABORT", 1) != 0;
  t.waves_per_cu = env_int("PTMI_WAVES_PER_CU", 0);
  t.bvh_teams = (int)std::max<uint32_t>(1, std::min<uint32_t>(kMaxTeams, (uint32_t)env_int("PTMI_BVH_TEAMS", t.bvh_teams)));
  t.refill = env_int("PTMI_REFILL", t.refill);
  t.leaf_batch = env_int("PTMI_LEAF_BATCH", t.leaf_batch);
  t.bvh_range = std::max(64, std::min(1 << 16, env_int("PTMI_BVH_RANGE", t.bvh_range))) & ~63;
  t.tail_waves_per_cu = std::max(0, std::min(32, env_int("PTMI_TAIL_WAVES_PER_CU", t.tail_waves_per_cu)));
  t.tail6 = env_int("PTMI_TAIL6", 1) != 0;
  t.tail_park = std::max(0, std::min(63, env_int("PTMI_TAIL_PARK", t.tail_park)));
  t.bvh_carry = std::max(0, env_int("PTMI_BVH_CARRY", t.bvh_carry));
  t.bvh_carry_slots = std::max(64, std::min(1 << 22, env_int("PTMI_BVH_CARRY_SLOTS", t.bvh_carry_slots)));
  t.bvh_carry_last = std::max(0, env_int("PTMI_BVH_CARRY_LAST", t.bvh_carry_last));
  t.bvh_carry_min_paths = std::max(0, env_int("PTMI_BVH_CARRY_MIN_PATHS", t.bvh_carry_min_paths));
  t.bvh_carry_min_depth = std::max(0, env_int("PTMI_BVH_CARRY_MIN_DEPTH", t.bvh_carry_min_depth));
  t.sort = env_int("PTMI_SORT", -1);
  t.shade_blocks_per_cu = env_int("PTMI_SHADE_BLOCKS_PER_CU", 0);
  t.tail_limit = env_int("PTMI_TAIL_LIMIT", -1);
  t.render_ahead = env_int("PTMI_RENDER_AHEAD", 1) != 0;
  t.path_budget_log2 = std::max(16, std::min(31, env_int("PTMI_PATH_BUDGET_LOG2", t.path_budget_log2)));
  t.placement_tries = env_int("PTMI_PLACEMENT_TRIES", t.placement_tries);
  t.debug_placement = getenv("PTMI_DEBUG_PLACEMENT") != nullptr;
  c->tun = t;
}

// hitScene, part 2 for the step's queue (k_bvh).  Part 1 has already been run by whoever created the rays (k_generate,
// k_shade); ptmi_trace's rays come from the host, so it asks for k_prims first.
int launch_intersect(ptmi_ctx* c, const Paths& P, StepCtl* ctl, uint32_t max_items, bool with_prims, const RenderConst* first_rc = nullptr, const Carry& cy = Carry{}) {
  unsigned long long* tot = c->d_totals.as<unsigned long long>();
  const uint32_t pgrid = std::max<uint32_t>(1, std::min<uint32_t>((max_items + kBlock - 1) / kBlock, (uint32_t)c->num_cus * 32));
  if (with_prims) {
    ScopedSpan sp(c, T_PRIMS);
    if (c->counters) hipLaunchKernelGGL(k_prims<true>, dim3(pgrid), dim3(kBlock), 0, c->stream, c->S, P, ctl, c->d_heads.as<uint32_t>(), tot);
    else hipLaunchKernelGGL(k_prims<false>, dim3(pgrid), dim3(kBlock), 0, c->stream, c->S, P, ctl, c->d_heads.as<uint32_t>(), tot);
  }
  if (c->S.n_nodes <= 0) return PTMI_OK;
  ScopedSpan sp(c, T_BVH);
  // Stack entries per lane: the first tun.lds_stack (10) in LDS, the rest (rarely reached) in a per-wave spill area.
  // 10 entries x 512 B + the candidate buffer (1 KB since round 4's three-group scan) = 6 KB per wave: 26 waves fit a CU's 160 KB, and the kernel's 66 VGPRs
  // admit 7 waves per SIMD.  (Round 3: the kernel runs at the rate of the CU's L1 gather path — tools/gather_probe*.hip, 2.8 clocks per
  // 64-byte record — so occupancy beyond ~18 waves buys 0-3 %: configs[1] 4.53 -> 4.18 ms, configs[3] 482 -> 481.)
  const Tuning& tun = c->tun;
  const int sa = stack_alloc_for(c);
  const int le = std::min(sa, tun.lds_stack);
  const int se = sa - le;
  const size_t lds = (size_t)le * 2 * 64 * sizeof(int) + kCandSlots * sizeof(uint32_t);  // stacks (2 words/entry) + candidate buffer
  // The abort of Q7 (hitRay.wgsl:106-109) needs sp to reach STACK_SIZE; sp never exceeds the number of inner nodes on a
  // root-to-leaf path.  PTMI_NOABORT=0 keeps the literal stack discipline for A/B runs.
  const bool noabort = c->bvh_depth < c->prm.stack_size && tun.noabort;
  int waves_per_cu = (int)std::min<size_t>(28, (size_t)(160 * 1024) / (lds + 64));
  if (tun.waves_per_cu > 0) waves_per_cu = tun.waves_per_cu;  // tuning aid; 0/unset = auto
  const uint32_t want = (max_items + 63) / 64;
  const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(want, (uint32_t)c->num_cus * (uint32_t)waves_per_cu));
  // Range claims go through 16 team counters (128 B apart) instead of one: a launch makes tens of thousands of claims and
  // same-address global atomics serialise at ~11 ns each (configs[1]: +2 %).
  const uint32_t n_teams = (uint32_t)tun.bvh_teams;
  // (the counters are zeroed by the kernel that filled this queue)
  HIP_TRY(c, c->d_spill.ensure(std::max<size_t>(16, (size_t)std::max<uint32_t>(grid, (uint32_t)c->num_cus * 32) * (size_t)se * 64 * sizeof(int2))));
  const int thr = tun.refill, leaf_batch = tun.leaf_batch;
  const uint32_t range_cap = (uint32_t)tun.bvh_range;
  // step 0's queue does not store the rays' common origin (k_generate): the kernel is handed cam_origin
  float4 cam = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (first_rc) cam = make_float4(first_rc->cam_o[0], first_rc->cam_o[1], first_rc->cam_o[2], 1.0f);
#define PTMI_LAUNCH_BVH(CNT, NA)                                                                                                                                       \
  hipLaunchKernelGGL((k_bvh2<CNT, NA>), dim3(grid), dim3(64), lds, c->stream, c->S, P, ctl, c->d_heads.as<uint32_t>(), n_teams, c->prm.stack_size, le, se, \
                     c->d_spill.as<int2>(), thr, leaf_batch, tot, range_cap, cam, cy)
  if (c->counters) {
    if (noabort) PTMI_LAUNCH_BVH(true, true);
    else PTMI_LAUNCH_BVH(true, false);
  } else {
    if (noabort) PTMI_LAUNCH_BVH(false, true);
    else PTMI_LAUNCH_BVH(false, false);
  }
#undef PTMI_LAUNCH_BVH
  HIP_TRY(c, hipGetLastError());
  return PTMI_OK;
}

// k_tail in front of a step: traces the step's queue to the end if it is short (PTMI_TAIL_LIMIT slots, 0 = never launched), else returns at once.
int launch_tail(ptmi_ctx* c, const RenderConst& rc, const Paths& P, StepCtl* ctl, int first, uint32_t limit, const Carry& cy_in) {
  // On trees of 12 levels and more a walk stops once fewer than tun.tail_park lanes are left in it while other lanes have work; the stragglers' state waits in three entries
  // on top of their stacks (tail_body).  Shallow trees never park: their walks are short, and a parked ray's path waits for the next walk (round 4 measured both).
  Carry cy = cy_in;
  cy.park_below = (c->S.n_nodes > 0 && c->bvh_depth >= 12) ? c->tun.tail_park : 0;  // (shallower: configs[1] -4 % at 8 lanes, +4 % at 16, the default scene +37 %: profiles/r05_tail_park_shallow.txt)
  const int sa = stack_alloc_for(c) + (cy.park_below > 0 ? 3 : 0);
  const int le = std::min(sa, c->tun.lds_stack);
  const int se = sa - le;
  const size_t lds = (size_t)le * 2 * 64 * sizeof(int);
  const bool noabort = c->bvh_depth < c->prm.stack_size && c->tun.noabort;
  // progressive mode without importance sampling on a scene without spheres: the 80-VGPR build, 6 waves per SIMD (k_tail6)
  const bool six = !c->prm.importance_sampling && rc.num_samples == 1 && c->S.n_spheres == 0 && c->tun.tail6;
  const int waves_per_cu = c->tun.tail_waves_per_cu > 0 ? c->tun.tail_waves_per_cu : (six ? 24 : 16);
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)limit + 63) / 64, (uint64_t)c->num_cus * (uint64_t)waves_per_cu));
  HIP_TRY(c, c->d_spill.ensure(std::max<size_t>(16, (size_t)c->num_cus * 32 * (size_t)se * 64 * sizeof(int2))));  // (k_bvh's grids are no larger: one size for both)
  unsigned long long* tot = c->d_totals.as<unsigned long long>();
  ScopedSpan sp(c, T_TAIL);
#define PTMI_LAUNCH_TAIL(IS, CN, MU, NA) \
  hipLaunchKernelGGL((k_tail<IS, CN, MU, NA>), dim3(grid), dim3(64), lds, c->stream, c->S, rc, P, ctl, tot, first, limit, c->prm.stack_size, le, se, c->d_spill.as<int2>(), cy)
#define PTMI_LAUNCH_TAIL3(IS, CN, MU)          \
  do {                                         \
    if (noabort) PTMI_LAUNCH_TAIL(IS, CN, MU, true); \
    else PTMI_LAUNCH_TAIL(IS, CN, MU, false);  \
  } while (0)
#define PTMI_LAUNCH_TAIL2(IS, CN)                             \
  do {                                                        \
    if (rc.num_samples > 1) PTMI_LAUNCH_TAIL3(IS, CN, true);  \
    else PTMI_LAUNCH_TAIL3(IS, CN, false);                    \
  } while (0)
  if (six) {
#define PTMI_LAUNCH_TAIL6(CN, NA) \
  hipLaunchKernelGGL((k_tail6<CN, NA>), dim3(grid), dim3(64), lds, c->stream, c->S, rc, P, ctl, tot, first, limit, c->prm.stack_size, le, se, c->d_spill.as<int2>(), cy)
    if (c->counters) {
      if (noabort) PTMI_LAUNCH_TAIL6(true, true);
      else PTMI_LAUNCH_TAIL6(true, false);
    } else {
      if (noabort) PTMI_LAUNCH_TAIL6(false, true);
      else PTMI_LAUNCH_TAIL6(false, false);
    }
#undef PTMI_LAUNCH_TAIL6
  } else if (c->prm.importance_sampling) {
    if (c->counters) PTMI_LAUNCH_TAIL2(true, true);
    else PTMI_LAUNCH_TAIL2(true, false);
  } else {
    if (c->counters) PTMI_LAUNCH_TAIL2(false, true);
    else PTMI_LAUNCH_TAIL2(false, false);
  }
#undef PTMI_LAUNCH_TAIL2
#undef PTMI_LAUNCH_TAIL3
#undef PTMI_LAUNCH_TAIL
  HIP_TRY(c, hipGetLastError());
  c->stats.tail_launches++;
  return PTMI_OK;
}

// progressive mode without importance sampling is k_shade6 (80 VGPRs); the k_shade instances for it are never made
template <bool IS, bool SO, bool CN, bool MU>
void launch_shade(ptmi_ctx* c, uint32_t sgrid, const RenderConst& rc, const Paths& P, StepCtl* ctl, unsigned long long* tot, int first, uint32_t resv) {
  if constexpr (!IS && !MU) hipLaunchKernelGGL((k_shade6<SO, CN>), dim3(sgrid), dim3(kBlock), 0, c->stream, c->S, rc, P, ctl, c->d_heads.as<uint32_t>(), tot, first, resv);
  else hipLaunchKernelGGL((k_shade<IS, SO, CN, MU>), dim3(sgrid), dim3(kBlock), 0, c->stream, c->S, rc, P, ctl, c->d_heads.as<uint32_t>(), tot, first, resv);
}

// Placement search (DESIGN.md §3 "placement"): where the driver puts the queue arrays decides how often their streams meet in the same HBM channels — k_shade runs up to 12 %
// slower in some contexts than in others, for their whole life.  For batches worth the trouble the context therefore tries up to PTMI_PLACEMENT_TRIES (6) sets of queue
// arrays, two alive at a time, and keeps the fastest.  Round 5: what is timed on each set is THE BATCH ITSELF — its k_generate and its first two steps, run dry (`dry` =
// render_batch(..., dry_steps = 2): no accumulation, no counters, no statistics; the real run that follows starts from scratch anyway) — instead of a synthetic kernel that
// imitated a step's access pattern and predicted the kernels' times poorly (profiles/r05_placement_tries.txt).  The first dry run of a set pages it in and is not counted.
int placement_search(ptmi_ctx* c, const std::function<int()>& dry) {
  const auto t_search = std::chrono::steady_clock::now();
  const size_t slots = c->slot_cap;
  struct Events {  // destroyed on every way out of the search
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  HIP_TRY(c, hipEventCreate(&ev.e0));
  HIP_TRY(c, hipEventCreate(&ev.e1));
  struct Restore {  // the dry runs leave no trace in what the caller can observe
    ptmi_ctx* c;
    ptmi_stats stats;
    int timing;
    bool counters;
    ~Restore() {
      c->stats = stats;
      c->timing = timing;
      c->counters = counters;
      c->batch_enqueued = false;
    }
  } restore{c, c->stats, c->timing, c->counters};
  c->timing = 0;
  c->counters = false;
  auto score = [&](float* ms) -> int {
    if (int r = dry()) return r;
    HIP_TRY(c, hipEventRecord(ev.e0, c->stream));
    if (int r = dry()) return r;
    HIP_TRY(c, hipEventRecord(ev.e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(ev.e1));
    HIP_TRY(c, hipEventElapsedTime(ms, ev.e0, ev.e1));
    return PTMI_OK;
  };
  float best = 0.0f;
  if (int r = score(&best)) return r;
  uint64_t sets = 1;
  if (c->tun.debug_placement) fprintf(stderr, "ptmi placement: set 0 %.3f ms\n", best);
  DBuf* mine[10] = {&c->d_q0[0], &c->d_q0[1], &c->d_q1[0], &c->d_q1[1], &c->d_q2[0], &c->d_q2[1], &c->d_tp[0], &c->d_tp[1], &c->d_hm[0], &c->d_hm[1]};
  const size_t width[10] = {16, 16, 16, 16, 16, 16, 8, 8, 4, 4};
  int rc_out = PTMI_OK;
  // The sets that lost stay allocated until the search is over (while the board has room for them): a set that is freed at once hands its pages to the next
  // candidate, which then scores the same to the microsecond.
  struct Losers {
    std::vector<DBuf> bufs;
    void drop() {
      for (DBuf& b : bufs) b.release();
      bufs.clear();
    }
    ~Losers() { drop(); }
  } losers;
  size_t set_bytes = 0;
  for (int k = 0; k < 10; k++) set_bytes += slots * width[k];
  for (int t = 1; t < c->tun.placement_tries; t++) {
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0, (void)hipGetLastError();
    if (mem_free < set_bytes + ((size_t)8 << 30)) {
      if (losers.bufs.empty()) break;
      losers.drop();
    }
    DBuf cand[10];
    bool ok = true;
    for (int k = 0; k < 10 && ok; k++) ok = cand[k].ensure(slots * width[k]) == hipSuccess;  // (an extra allocation that fails just ends the search)
    float ms = 0.0f;
    if (ok) {
      for (int k = 0; k < 10; k++) std::swap(*mine[k], cand[k]);  // (cand now holds the best set so far)
      rc_out = score(&ms);
      ok = rc_out == PTMI_OK;
      if (ok) sets++;
      if (c->tun.debug_placement) fprintf(stderr, "ptmi placement: set %d %.3f ms\n", t, ms);
      if (!ok || ms >= best) {
        for (int k = 0; k < 10; k++) std::swap(*mine[k], cand[k]);  // keep the old one
      } else {
        best = ms;
      }
    }
    if (!ok) (void)hipStreamSynchronize(c->stream);  // nothing may still run on a set that is about to go
    for (int k = 0; k < 10; k++) {
      if (ok && cand[k].p) {
        losers.bufs.emplace_back();
        std::swap(losers.bufs.back(), cand[k]);
      } else {
        cand[k].release();
      }
    }
    if (!ok) break;
  }
  restore.stats.placement_sets = sets;
  restore.stats.placement_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_search).count();
  return rc_out;
}

// `fold` = how many of the batch's leading frames are added to the framebuffer now (-1 = all of them); dry_steps > 0: placement_search's timing run
int render_batch(ptmi_ctx* c, const float* view16, uint32_t frame0, int n_frames, int reset_first, int fold = -1, int dry_steps = 0) {
  const ptmi_params& p = c->prm;
  c->ahead.valid = false;  // the path buffers are about to be overwritten
  RenderConst rc{};
  rc.W = (float)c->W;
  rc.H = (float)c->H;
  memcpy(rc.view, view16, 64);
  {  // cam_origin = (view * vec4(0,0,0,1)).xyz, the column-weighted sum of SURVEY.md §8a-W in f32 (ptmi_device.h: cam_origin)
    const float* m = rc.view;
    for (int k = 0; k < 3; k++) rc.cam_o[k] = ((m[k] * 0.0f + m[4 + k] * 0.0f) + m[8 + k] * 0.0f) + m[12 + k] * 1.0f;
  }
  rc.fov_factor = (float)(1.0 / std::tan((double)p.fov_degrees * (3.14159265358979323846 / 180.0) / 2.0));  // main.wgsl:7, folded in f64
  rc.bg[0] = p.background[0], rc.bg[1] = p.background[1], rc.bg[2] = p.background[2];
  rc.max_bounces = p.max_bounces;
  rc.stratify = p.stratify ? 1 : 0;
  if (p.stratify) {  // shootRay.wgsl:9-31
    float sqrt_spp = (float)std::sqrt((double)p.num_samples);
    int side = 0;
    for (float i = 0.0f; i < sqrt_spp; i += 1.0f) side++;
    rc.strat_side = side;
    rc.recip_sqrt_spp = 1.0f / (float)(int)sqrt_spp;
    rc.num_samples = side * side;
    rc.sample_div = (float)(side * side);
  } else {
    rc.strat_side = 1;
    rc.recip_sqrt_spp = 1.0f;
    rc.num_samples = p.num_samples;
    rc.sample_div = (float)p.num_samples;
  }
  rc.stack_size = p.stack_size;
  rc.light_mix = p.light_mix;
  rc.surface_mix = 1.0f - p.light_mix;  // 1.0f - 0.2f == 0.8f, the shader's literal
  rc.npix = (uint32_t)c->W * (uint32_t)c->H;
  rc.frame0 = frame0;
  rc.n_frames = n_frames;
  rc.reset_first = reset_first;
  rc.rank = c->rank, rc.world = c->world, rc.tile = c->tile;
  rc.n_local = count_local(rc.npix, c->rank, c->world, c->tile);
  if (rc.n_local == 0) return PTMI_OK;

  const int n_steps = rc.num_samples * p.max_bounces;
  const size_t npaths = (size_t)rc.n_local * (size_t)n_frames;
  c->batch_enqueued = false;
  int rcode = ensure_paths(c, npaths, n_steps + 2, rc.num_samples > 1);
  if (rcode) return rcode;  // nothing is on the stream yet: ptmi_render may retry with a smaller batch
  StepCtl* ctl = c->d_ctl.as<StepCtl>();
  const uint32_t total = rc.n_local * (uint32_t)n_frames;
  // Carry (ptmi_device.h): rays that outlive their k_bvh launch move into the next step's queue.  Worth it where a launch's longest ray is long
  // against the launch — deep trees, batches of millions of paths; a shallow tree's tail is microseconds, and small batches belong to k_tail anyway.
  const int sa_carry = stack_alloc_for(c);
  const bool carry = c->tun.bvh_carry > 0 && c->S.n_nodes > 0 && c->bvh_depth >= c->tun.bvh_carry_min_depth && total >= (uint32_t)c->tun.bvh_carry_min_paths && p.max_bounces > 1;
  const uint32_t resv = carry ? (uint32_t)c->tun.bvh_carry_slots : 0u;
  const int rec_words = 8 + 2 * sa_carry;
  if (carry)
    for (int k = 0; k < 2; k++) HIP_TRY(c, c->d_carry[k].ensure((size_t)resv * (size_t)rec_words * 4));
  auto carry_of = [&](int s) {  // what step s's kernels need to know: its queue's prefix, the next one's, the two pools
    Carry cy{};
    cy.resv = s == 0 ? 0u : resv;
    cy.resv_next = resv;
    cy.after = c->tun.bvh_carry;
    cy.rec_words = rec_words;
    cy.pool_in = c->d_carry[s & 1].as<uint32_t>();
    cy.pool_out = c->d_carry[(s + 1) & 1].as<uint32_t>();
    return cy;
  };
  const uint32_t bound = total + total / 8 + (uint32_t)c->num_cus * 8 * 1024 + resv;  // slots a step's queue can span
  const uint32_t ew_grid = std::max<uint32_t>(1, std::min<uint32_t>((total + kBlock - 1) / kBlock, (uint32_t)c->num_cus * 16));
  // k_shade sorts its chunks by material class only when the scene has more than one (PTMI_SORT=0/1 overrides, for A/B runs)
  const int sort_env = c->tun.sort;
  const bool sort = sort_env >= 0 ? sort_env != 0 : c->material_classes > 1;

  // k_shade's grid: as many blocks per CU as the variant's registers and LDS admit (the progressive-mode variants need 79 VGPRs since
  // the build dropped the SLP vectoriser: 6 blocks = 6 waves per SIMD; the importance-sampling ones 93: 5) — asked of the runtime once per variant
  const bool shade_multi = rc.num_samples > 1;
  int shade_bpc = c->tun.shade_blocks_per_cu;
  if (shade_bpc <= 0) {
    int& cached = c->shade_blocks_per_cu[(p.importance_sampling ? 8 : 0) | (sort ? 4 : 0) | (c->counters ? 2 : 0) | (shade_multi ? 1 : 0)];
    if (cached == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, shade_kernel(p.importance_sampling != 0, sort, c->counters, shade_multi), kBlock, 0) != hipSuccess || nb < 1) nb = 5;
      cached = nb;
    }
    shade_bpc = cached;
  }
  const uint32_t sgrid = std::max<uint32_t>(1, std::min<uint32_t>((bound + kSChunk - 1) / kSChunk, (uint32_t)c->num_cus * (uint32_t)std::min(8, std::max(1, shade_bpc))));  // <= 8: the queue buffers' slack is sized for that (ensure_paths)
  unsigned long long* tot = c->d_totals.as<unsigned long long>();
  // queues of at most this many slots are traced to the end by one k_tail launch instead of a k_bvh + k_shade pair per bounce
  // (a whole small batch — a lone 1080p frame — at step 0; later steps hand over only their thin ends: on deep trees a lane-per-path
  // wave waits for its longest traversal, and the wavefront kernels with their lane refill stay ahead down to ~0.5 Mi slots)
  const int tail_env = c->tun.tail_limit;
  // Where k_tail's weak spot — the tree walk, as long as the wave's longest ray — is short, it stays ahead of the per-bounce kernels far longer (round 5,
  // profiles/r05_tail_first_sweep*.txt): on configs[1]'s 11-level tree it wins up to ~24 Mi paths per batch (4 Mi paths: 0.94 ms against 1.54; 16.6 Mi — one rank of eight at
  // fixed total spp, or an 8-frame render-ahead batch — 2.87 against 3.26).  On deep trees, since its walks park their stragglers (launch_tail), up to ~16 Mi paths on the
  // 871 k-triangle scene (4 Mi: 3.5 ms against 6.5; 8 Mi: 4.9 against 6.4) but only ~4 Mi inside the 262 k-triangle room (8 Mi: 22.1 against 20.8): 6 Mi.  The builds with
  // importance sampling, several samples or spheres (100-130 VGPRs, 4 waves per SIMD) keep 2 Mi.  Later steps' queues: from 2 Mi down on shallow trees, 1 Mi otherwise.
  const bool six = !p.importance_sampling && rc.num_samples == 1 && c->S.n_spheres == 0 && c->tun.tail6;
  const bool shallow = six && c->S.n_nodes > 0 && c->bvh_depth < 12, deep = six && c->S.n_nodes > 0 && c->bvh_depth >= 12 && c->tun.tail_park > 0;
  const uint32_t tail_limit_first = (uint32_t)(tail_env >= 0 ? tail_env : shallow ? kTailLimitFirstShallow : deep ? kTailLimitFirstDeep : kTailLimitFirst),
                 tail_limit_later = (uint32_t)(tail_env >= 0 ? tail_env : shallow ? kTailLimitLaterShallow : kTailLimitLater);

  if (dry_steps == 0 && c->placement_pending) {
    c->placement_pending = false;
    if (total > tail_limit_first) {  // (a batch that k_tail takes whole reads its queue once: nothing to search for)
      // (two steps: the whole batch as the probe — eight steps — chose no better: profiles/r05_placement_dry_run.txt)
      int r = placement_search(c, [&]() { return render_batch(c, view16, frame0, n_frames, reset_first, fold, 2); });
      if (r) return r;
    }
  }
  ScopedSpan whole(c, T_RENDER);
  c->batch_enqueued = true;  // from here on a failure leaves a partly traced batch behind: never retried
  hipLaunchKernelGGL(k_init_ctl, dim3((unsigned)((n_steps + 2 + 63) / 64)), dim3(64), 0, c->stream, ctl, n_steps + 2, resv);  // n_rays = the carry prefix (0 for step 0)
  HIP_TRY(c, hipMemsetAsync(tot + 16, 0, kTotalsBytes - 128, c->stream));  // the batch's hitScene tally
  {
    ScopedSpan s(c, T_GENERATE);
    if (rc.num_samples == 1) HIP_TRY(c, hipMemsetAsync(c->d_touched.p, 0, npaths, c->stream));
    if (c->counters) hipLaunchKernelGGL(k_generate<true>, dim3(ew_grid), dim3(kBlock), 0, c->stream, c->S, rc, paths_of(c, 0, rc.num_samples > 1), ctl, c->d_heads.as<uint32_t>(), tot);
    else hipLaunchKernelGGL(k_generate<false>, dim3(ew_grid), dim3(kBlock), 0, c->stream, c->S, rc, paths_of(c, 0, rc.num_samples > 1), ctl, c->d_heads.as<uint32_t>(), tot);
    HIP_TRY(c, hipGetLastError());
    c->stats.generate_launches++;
  }
  bool drained = false;
  const int run_steps = dry_steps > 0 ? std::min(n_steps, dry_steps) : n_steps;
  for (int s = 0; s < run_steps; s++) {
    // With the reference's MAX_BOUNCES = 100 nearly all steps run on an empty queue (Russian roulette ends paths after
    // a dozen bounces): from step 12 on, look at the queue length every 8 steps and stop enqueuing once it is empty.
    if (s >= 12 && (s & 7) == 4) {
      uint32_t left = 1;
      uint32_t rec3[3] = {1, 0, 0};  // n_rays, tail_done, n_carried
      HIP_TRY(c, hipMemcpyAsync(rec3, &ctl[s].n_rays, sizeof rec3, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      left = (rec3[0] <= resv && rec3[2] == 0) ? 0u : 1u;
      if (left == 0) {
        drained = true;
        break;
      }
    }
    Paths P = paths_of(c, s, rc.num_samples > 1);
    // (dry runs: k_generate and k_shade only — the kernels whose time depends on where the queue arrays lie; k_bvh reads them sparsely, and on a deep tree it would
    // be nine tenths of the search's time.  Rays that entered the root box are then shaded with what part 1 of hitScene found: other paths, the same access pattern.)
    if (const uint32_t tail_limit = dry_steps > 0 ? 0u : (s == 0 ? tail_limit_first : tail_limit_later)) {
      int lr = launch_tail(c, rc, P, ctl + s, s == 0 ? 1 : 0, tail_limit, carry_of(s));
      if (lr) return lr;
      // step 0's queue is the whole batch (k_generate fills one slot per path): if that fits the limit k_tail has just been handed all of it —
      // nothing is left for the per-bounce kernels, and a lone frame is three launches instead of 3 x MAX_BOUNCES + 2
      if (s == 0 && total <= tail_limit) {
        drained = true;
        break;
      }
    }
    if (dry_steps == 0) {
      Carry cy = carry_of(s);
      if (!carry || s >= n_steps - c->tun.bvh_carry_last) cy.resv_next = 0u;
      int lr = launch_intersect(c, P, ctl + s, bound, false, s == 0 ? &rc : nullptr, cy);
      if (lr) return lr;
    }
    {
      ScopedSpan sp(c, T_SHADE);
#define PTMI_LAUNCH_SHADE(IS, SO, CN, MU) launch_shade<IS, SO, CN, MU>(c, sgrid, rc, P, ctl + s, tot, s == 0 ? 1 : 0, s == 0 ? 0u : resv)
#define PTMI_LAUNCH_SHADE2(IS, SO)                          \
  do {                                                      \
    if (rc.num_samples > 1) {                               \
      if (c->counters) PTMI_LAUNCH_SHADE(IS, SO, true, true); \
      else PTMI_LAUNCH_SHADE(IS, SO, false, true);          \
    } else {                                                \
      if (c->counters) PTMI_LAUNCH_SHADE(IS, SO, true, false); \
      else PTMI_LAUNCH_SHADE(IS, SO, false, false);         \
    }                                                       \
  } while (0)
      if (p.importance_sampling) {
        if (sort) PTMI_LAUNCH_SHADE2(true, true);
        else PTMI_LAUNCH_SHADE2(true, false);
      } else {
        if (sort) PTMI_LAUNCH_SHADE2(false, true);
        else PTMI_LAUNCH_SHADE2(false, false);
      }
#undef PTMI_LAUNCH_SHADE2
#undef PTMI_LAUNCH_SHADE
      HIP_TRY(c, hipGetLastError());  // launch errors surface per step, before k_accumulate touches the framebuffer
    }
    c->stats.intersect_launches++;
    c->stats.shade_launches++;
  }
  if (dry_steps > 0) return PTMI_OK;
  if (carry && !drained && n_steps > 0) {
    // paths that were carried over lag behind the step count: whatever the last k_shade left in the queue (and what the last k_bvh carried) is
    // traced to its end by one k_tail launch — a few thousand paths at most
    int lr = launch_tail(c, rc, paths_of(c, n_steps, rc.num_samples > 1), ctl + n_steps, 0, 0xffffffffu, carry_of(n_steps));
    if (lr) return lr;
  }
  {
    ScopedSpan s(c, T_ACCUM);
    hipLaunchKernelGGL(k_accumulate, dim3(ew_grid), dim3(kBlock), 0, c->stream, rc, paths_of(c, 0, rc.num_samples > 1), c->fb, n_steps, tot, 0,
                       fold < 0 ? n_frames : std::min(fold, n_frames));
    c->stats.accumulate_launches++;
  }
  HIP_TRY(c, hipGetLastError());
  c->stats.frames += (uint64_t)n_frames;
  c->ahead.rc = rc;
  c->ahead.grid = ew_grid;
  return PTMI_OK;
}

// totals[15]: a k_shade block found the next queue full and dropped paths (cannot happen with ensure_paths' sizing).  Every
// synchronising call reports it, so that a damaged image is never handed out as a good one.  Call with the stream idle.
int check_queue_overflow(ptmi_ctx* c) {
  if (!c->d_totals.p) return PTMI_OK;
  unsigned long long flag = 0;
  HIP_TRY(c, hipMemcpy(&flag, c->d_totals.as<unsigned long long>() + 15, sizeof flag, hipMemcpyDeviceToHost));
  if (flag) return fail(c, PTMI_ERR_STATE, "internal: a step's queue outgrew its buffer (paths were dropped); the framebuffer is incomplete");
  return PTMI_OK;
}

int check_renderable(ptmi_ctx* c) {
  if (!c->fb || c->W <= 0) return fail(c, PTMI_ERR_STATE, "no framebuffer: call ptmi_resize first");
  const ptmi_params& p = c->prm;
  if (p.importance_sampling && c->has_unknown_material)
    return fail(c, PTMI_ERR_UNSUPPORTED,
                "importance_sampling with a material_type outside {0,1,2,3}: the shader would read stale private state (scatterRec)");
  return PTMI_OK;
}

// ---- multi-device contexts ---------------------------------------------------------------------------------------
// librccl (RCCL = the NCCL API on ROCm; collectives run over xGMI between the GPUs of a node) is loaded when the first
// multi-device context is created, so that single-GPU users of libptmi.so do not need it.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
  bool load() {
    if (lib) return true;
    const char* names[] = {getenv("PTMI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
      why = dlerror();
    }
    if (!lib) return false;
#define PTMI_RCCL_SYM(field, sym)                                  \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, #sym));     \
  if (!field) {                                                    \
    why = std::string("librccl lacks ") + #sym;                    \
    dlclose(lib);                                                  \
    lib = nullptr;                                                 \
    return false;                                                  \
  }
    PTMI_RCCL_SYM(CommInitAll, ncclCommInitAll)
    PTMI_RCCL_SYM(CommDestroy, ncclCommDestroy)
    PTMI_RCCL_SYM(CommAbort, ncclCommAbort)
    PTMI_RCCL_SYM(Reduce, ncclReduce)
    PTMI_RCCL_SYM(GroupStart, ncclGroupStart)
    PTMI_RCCL_SYM(GroupEnd, ncclGroupEnd)
    PTMI_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef PTMI_RCCL_SYM
    return true;
  }
};
Rccl g_rccl;
// the fault a -DPTMI_TEST_HOOKS build was asked to simulate for this context; a constant 0 in the product build, so the branches and their messages fold away
inline int rccl_fail_hook(const ptmi_ctx* c) {
#ifdef PTMI_TEST_HOOKS
  return c->test_rccl_fail;
#else
  (void)c;
  return 0;
#endif
}
std::once_flag g_rccl_once;  // two host threads may create multi-device contexts at the same time
bool rccl_loaded() {
  std::call_once(g_rccl_once, [] { (void)g_rccl.load(); });
  return g_rccl.lib != nullptr;
}

#define RCCL_TRY(c, expr)                                                                                            \
  do {                                                                                                               \
    ncclResult_t _r = (expr);                                                                                        \
    if (_r != ncclSuccess) return fail((c), PTMI_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
  } while (0)

// fn(ctx) on the context itself and on every peer; `parallel` = one host thread per device, so that calls which block
// (allocation, the occasional queue-length readback of a long render) do not serialise the GPUs.
template <class F>
int on_all_devices(ptmi_ctx* c, F fn, bool parallel = false) {
  const size_t n = c->peers.size();
  if (n == 0) return fn(c);
  std::vector<int> rcs(n + 1, PTMI_OK);
  if (parallel) {
    for (size_t i = 0; i < n; i++) {
      ptmi_ctx* q = c->peers[i];
      if (!q->worker) q->worker = new PeerWorker();
      q->worker->post([&fn, q] { return fn(q); });
    }
    rcs[0] = fn(c);
    for (size_t i = 0; i < n; i++) rcs[i + 1] = c->peers[i]->worker->wait();
  } else {
    rcs[0] = fn(c);
    for (size_t i = 0; i < n && !rcs[i]; i++) rcs[i + 1] = fn(c->peers[i]);  // stop at the first device that fails
  }
  for (size_t i = 1; i <= n; i++)
    if (rcs[i] && !rcs[0]) {
      c->err = "local device #" + std::to_string(i) + " (GPU " + std::to_string(c->peers[i - 1]->device) + "): " + c->peers[i - 1]->err;
      return rcs[i];
    }
  return rcs[0];
}

// One device's own tiles (ptmi_set_shard: pixels p with (p / tile) % world == rank) copied from its accumulation buffer into place in the root's gather buffer.  `src` may be
// the peer's memory itself (peer access over xGMI: the kernel reads exactly the bytes it needs, 1/N of the buffer) or a staged copy of it.
__global__ __launch_bounds__(kBlock) void k_gather_tiles(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n_local, int rank, int world, int tile) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < n_local; j += gridDim.x * kBlock) {
    const uint32_t tl = j / (uint32_t)tile, within = j - tl * (uint32_t)tile;
    const uint32_t pix = (tl * (uint32_t)world + (uint32_t)rank) * (uint32_t)tile + within;
    dst[pix] = src[pix];
  }
}

__global__ __launch_bounds__(kBlock) void k_add_into(float4* __restrict__ dst, const float4* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// The one collective of a multi-device render: sum the per-device accumulation buffers into d_fb_gather on local device 0.
// Every pixel is non-zero in exactly one of them (x + 0 = x), so the sum is the single-GPU image bit for bit whatever
// the order.  The per-device buffers are left as they are, so rendering can go on afterwards.
int gather_framebuffer(ptmi_ctx* c, float4** out) {
  if (!c->multi) {
    *out = c->fb;
    return PTMI_OK;
  }
  const size_t bytes = c->fb_bytes, n4 = bytes / 16;
  int r = on_all_devices(c, [](ptmi_ctx* q) -> int {
    HIP_TRY(q, hipSetDevice(q->device));
    HIP_TRY(q, hipStreamSynchronize(q->stream));
    return PTMI_OK;
  });
  if (r) return r;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->d_fb_gather.ensure(bytes));
  float4* g = c->d_fb_gather.as<float4>();
  if (c->use_rccl) {
    // ncclReduce(sendbuff = this device's buffer, recvbuff = the gather buffer on the root, W*H*4 floats, sum, root 0).
    // A group that has been started is always ended — an open group would swallow every later RCCL call of the process —;
    // the first error inside it is noted after ncclGroupEnd.  An RCCL failure does not fail the read-back: the per-device buffers are
    // untouched partial sums, so the context switches to the peer-copy reduce below for good (reduce_mode 3) and sums them that way.
    std::string first_error;
    ncclResult_t gs = rccl_fail_hook(c) == 2 ? ncclSystemError : g_rccl.GroupStart();
    if (gs != ncclSuccess) {
      first_error = std::string("ncclGroupStart: ") + (rccl_fail_hook(c) == 2 ? "simulated failure (PTMI_TEST_RCCL_FAIL=reduce)" : g_rccl.GetErrorString(gs));
    } else {
      for (size_t i = 0; i <= c->peers.size() && first_error.empty(); i++) {
        ptmi_ctx* q = i ? c->peers[i - 1] : c;
        const hipError_t he = hipSetDevice(q->device);
        if (he != hipSuccess) {
          first_error = std::string("hipSetDevice: ") + hipGetErrorString(he);
          break;
        }
        const ncclResult_t nr = g_rccl.Reduce(q->fb, i ? (void*)q->fb : (void*)g, n4 * 4, ncclFloat, ncclSum, 0, c->comms[i], q->stream);
        if (nr != ncclSuccess) first_error = std::string("ncclReduce (local device #") + std::to_string(i) + "): " + g_rccl.GetErrorString(nr);
        else if (rccl_fail_hook(c) == 3) first_error = "simulated failure after the first ncclReduce was enqueued (PTMI_TEST_RCCL_FAIL=mid)";
      }
      const ncclResult_t ge = g_rccl.GroupEnd();
      if (first_error.empty() && ge != ncclSuccess) first_error = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge);
    }
    if (!first_error.empty()) {
      // A reduce that went out on some ranks only never completes: waiting for the streams first would hang instead of falling back.  So the
      // communicators are aborted — that ends whatever they have in flight — and dropped before anybody synchronises (ncclCommDestroy on a
      // communicator with an unfinished collective can block too: ptmi_destroy never sees these).
      for (ncclComm_t cm : c->comms)
        if (cm) (void)g_rccl.CommAbort(cm);
      c->comms.clear();
    }
    hipError_t se = hipSuccess;
    for (size_t i = 0; i <= c->peers.size() && se == hipSuccess; i++) {  // whatever was enqueued has to drain before anybody reads or re-sums the buffers
      ptmi_ctx* q = i ? c->peers[i - 1] : c;
      se = hipSetDevice(q->device);
      if (se == hipSuccess) se = hipStreamSynchronize(q->stream);
    }
    (void)hipSetDevice(c->device);
    if (first_error.empty() && se != hipSuccess) {
      first_error = std::string("after ncclReduce: ") + hipGetErrorString(se);
      for (ncclComm_t cm : c->comms)
        if (cm) (void)g_rccl.CommAbort(cm);
      c->comms.clear();
    }
    if (!first_error.empty()) {
      (void)hipGetLastError();
      c->use_rccl = false;
      c->reduce_mode = 3;
      c->reduce_info = "FALLBACK: hipMemcpyPeer + add (" + first_error + ")";
    }
  }
  if (!c->use_rccl && c->use_gather) {
    // Tile gather (the default): every pixel belongs to exactly one device, so the image is the devices' own tiles put side by side — each peer's tiles are read
    // straight out of its buffer by a kernel on the root (peer access over xGMI: 1/N of the buffer per peer, N - 1 reads in total instead of N - 1 full buffers,
    // no arithmetic, no staging); a peer the root cannot read is staged through one hipMemcpyPeerAsync first.  Pixels no local device owns (another process's shard) stay zero.
    HIP_TRY(c, hipMemsetAsync(g, 0, bytes, c->stream));
    const uint32_t npix = (uint32_t)c->W * (uint32_t)c->H;
    c->gather_bytes = 0;
    for (size_t i = 0; i <= c->peers.size(); i++) {
      ptmi_ctx* q = i ? c->peers[i - 1] : c;
      const uint32_t n_local = count_local(npix, q->rank, q->world, q->tile);
      if (n_local == 0) continue;
      const float4* src = q->fb;
      if (q->device != c->device && !q->root_reads) {
        HIP_TRY(c, c->d_fb_stage.ensure(bytes));
        HIP_TRY(c, hipMemcpyPeerAsync(c->d_fb_stage.p, c->device, q->fb, q->device, bytes, c->stream));
        src = c->d_fb_stage.as<float4>();
        c->gather_bytes += bytes;
      } else if (q->device != c->device) {
        c->gather_bytes += (uint64_t)n_local * 16;
      }
      const unsigned grid = (unsigned)std::min<size_t>(((size_t)n_local + kBlock - 1) / kBlock, (size_t)c->num_cus * 8);
      hipLaunchKernelGGL(k_gather_tiles, dim3(grid), dim3(kBlock), 0, c->stream, g, src, n_local, q->rank, q->world, q->tile);
      HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    char msg[200];
    snprintf(msg, sizeof msg, "tile gather: every device's own tiles read into place by the root (%.1f MB between GPUs per read-back, of %.1f MB a full-buffer reduce would move)",
             (double)c->gather_bytes / 1e6, (double)(c->peers.size() * bytes) / 1e6);
    c->reduce_info = msg;
  } else if (!c->use_rccl) {
    // shards that share this GPU (tests on a one-GPU box), or PTMI_MULTI_REDUCE=copy: peer copy + add kernel
    HIP_TRY(c, hipMemcpyAsync(g, c->fb, bytes, hipMemcpyDeviceToDevice, c->stream));
    const unsigned grid = (unsigned)std::min<size_t>((n4 + kBlock - 1) / kBlock, (size_t)c->num_cus * 8);
    for (ptmi_ctx* q : c->peers) {
      const float4* src = q->fb;
      if (q->device != c->device) {
        HIP_TRY(c, c->d_fb_stage.ensure(bytes));
        HIP_TRY(c, hipMemcpyPeerAsync(c->d_fb_stage.p, c->device, q->fb, q->device, bytes, c->stream));
        src = c->d_fb_stage.as<float4>();
      }
      hipLaunchKernelGGL(k_add_into, dim3(grid), dim3(kBlock), 0, c->stream, g, src, n4);
      HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  *out = g;
  return PTMI_OK;
}

int apply_shard(ptmi_ctx* c) {  // deal the caller's shard to the local devices
  const int n = (int)c->peers.size() + 1;
  for (int i = 0; i < n; i++) {
    ptmi_ctx* q = i ? c->peers[i - 1] : c;
    q->rank = c->proc_rank * n + i;
    q->world = c->proc_world * n;
    q->tile = c->proc_tile;
    q->ahead.valid = false;
  }
  return PTMI_OK;
}

uint32_t u32_of_f32(float f) {  // WGSL u32(f32): truncation, clamped to the u32 range
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)f;
}

}  // namespace

// for the other translation units of the library (ptmi_bvh_device.hip)
int ptmi_ctx_set_device(ptmi_ctx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  return PTMI_OK;
}
int ptmi_ctx_fail(ptmi_ctx* c, int code, const char* what) { return fail(c, code, what); }

extern "C" {

int ptmi_version(void) { return PTMI_API_VERSION; }

const char* ptmi_status_string(int s) {
  switch (s) {
    case PTMI_OK: return "ok";
    case PTMI_ERR_INVALID_ARG: return "invalid argument";
    case PTMI_ERR_DEVICE: return "device (HIP) error";
    case PTMI_ERR_STATE: return "invalid state / call order";
    case PTMI_ERR_NO_MEMORY: return "out of memory";
    case PTMI_ERR_BAD_SCENE: return "scene buffers failed validation";
    case PTMI_ERR_UNSUPPORTED: return "unsupported parameter combination";
  }
  return "unknown status";
}

const char* ptmi_last_error(const ptmi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void ptmi_default_params(ptmi_params* p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->num_samples = 1;
  p->max_bounces = 100;
  p->stratify = 0;
  p->importance_sampling = 0;
  p->stack_size = 20;
  p->background[0] = 0.0f, p->background[1] = 1.0f, p->background[2] = 1.0f;
  p->fov_degrees = 60.0f;
  p->frames_in_flight = 0;
  p->tmin = 0.000001f;   // header.wgsl:37
  p->light_mix = 0.2f;   // traceRay.wgsl:43,49
}

int ptmi_device_count(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* ptmi_reduce_info(const ptmi_ctx* c) { return c ? c->reduce_info.c_str() : ""; }

int ptmi_create(ptmi_ctx** out, int device_id) {
  if (!out) return fail(nullptr, PTMI_ERR_INVALID_ARG, "ptmi_create: out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, PTMI_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(nullptr, PTMI_ERR_INVALID_ARG, "ptmi_create: device_id out of range");
  HIP_TRY(nullptr, hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, PTMI_ERR_DEVICE, std::string("libptmi is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
  ptmi_ctx* c = new (std::nothrow) ptmi_ctx();
  if (!c) return fail(nullptr, PTMI_ERR_NO_MEMORY, "ptmi_create: host allocation failed");
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  ptmi_default_params(&c->prm);
  load_tuning(c);
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(nullptr, PTMI_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  *out = c;
  return PTMI_OK;
}

int ptmi_create_multi(ptmi_ctx** out, const int* device_ids, int n_devices) {
  if (!out) return fail(nullptr, PTMI_ERR_INVALID_ARG, "ptmi_create_multi: out is null");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(nullptr, PTMI_ERR_INVALID_ARG, "ptmi_create_multi: need 1 <= n_devices <= 64 device ids");
  ptmi_ctx* c = nullptr;
  int r = ptmi_create(&c, device_ids[0]);
  if (r) return r;
  c->multi = true;
  bool distinct = true;
  for (int i = 1; i < n_devices; i++) {
    ptmi_ctx* q = nullptr;
    r = ptmi_create(&q, device_ids[i]);
    if (r) {
      ptmi_destroy(c);
      return r;  // (the message is in the thread's create-error slot)
    }
    c->peers.push_back(q);
    for (int j = 0; j < i; j++) distinct = distinct && device_ids[j] != device_ids[i];
  }
  if (!distinct) {
    c->shares_device = true;
    for (ptmi_ctx* q : c->peers) q->shares_device = true;
  }
  // The reduce: RCCL whenever every shard has a GPU of its own (a communicator cannot hold one GPU twice); shards that
  // share a GPU are summed by a kernel.  PTMI_MULTI_REDUCE=copy forces the peer-copy path, =rccl forces RCCL even for a
  // single device (a one-rank communicator: exercises the library on a one-GPU box).
  // PTMI_MULTI_REDUCE: gather (default: the devices' own tiles copied into place, 1/N of the bytes), rccl (ncclReduce of the full buffers — north_star's wording —; needs
  // distinct GPUs; forces RCCL even for a single device: a one-rank communicator, which exercises the library on a one-GPU box), copy (peer copies + add kernel)
  const char* mode = getenv("PTMI_MULTI_REDUCE");
  const bool force_rccl = mode && !strcmp(mode, "rccl"), force_copy = mode && !strcmp(mode, "copy");
  c->use_rccl = force_rccl && distinct;
  c->use_gather = !force_rccl && !force_copy;
  if (force_rccl && !distinct) {
    ptmi_destroy(c);
    return fail(nullptr, PTMI_ERR_UNSUPPORTED, "PTMI_MULTI_REDUCE=rccl needs distinct device ids (an RCCL communicator cannot hold a GPU twice)");
  }
  // Peer access between the root and every other device, both ways (the peer-copy reduce reads the peers' buffers from the root; RCCL sets up
  // its own).  Not fatal when refused: hipMemcpyPeerAsync then stages through the host — slower, same bytes.
  if (distinct && n_devices > 1) {
    for (ptmi_ctx* q : c->peers) {
      for (int dir = 0; dir < 2; dir++) {
        const int from = dir ? q->device : c->device, to = dir ? c->device : q->device;
        int can = 0;
        if (hipSetDevice(from) != hipSuccess || hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) continue;
        const hipError_t pe = hipDeviceEnablePeerAccess(to, 0);
        if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) {
          c->peer_links++;
          if (dir == 0) q->root_reads = true;  // (enabled on the root for the peer's memory)
        }
        (void)hipGetLastError();
      }
    }
    (void)hipSetDevice(c->device);
  }
#ifdef PTMI_TEST_HOOKS
  if (const char* tf = getenv("PTMI_TEST_RCCL_FAIL")) c->test_rccl_fail = !strcmp(tf, "init") ? 1 : !strcmp(tf, "reduce") ? 2 : !strcmp(tf, "mid") ? 3 : 0;
#endif
  for (ptmi_ctx* q : c->peers)
    if (q->device == c->device) q->root_reads = true;
  c->reduce_mode = n_devices > 1 || force_rccl ? (c->use_gather ? 4 : 2) : 0;
  c->reduce_info = n_devices > 1 ? (c->use_gather ? "tile gather: every device's own tiles read into place by the root" : distinct ? "hipMemcpyPeer + add (PTMI_MULTI_REDUCE=copy)" : "add kernel (shards share a GPU)")
                                 : "single device";
  if (c->use_rccl) {
    // The RCCL path has to be able to fail without taking the context with it: if the library cannot be loaded or the communicators cannot be
    // made, the reduce falls back to peer copies + an add kernel (bit-identical: every pixel is non-zero in one buffer) and says so.
    std::string why;
    if (!rccl_loaded()) {
      why = "cannot load librccl (" + g_rccl.why + "); set PTMI_RCCL_LIB";
    } else {
      c->comms.assign((size_t)n_devices, nullptr);
      const ncclResult_t nr = rccl_fail_hook(c) == 1 ? ncclSystemError : g_rccl.CommInitAll(c->comms.data(), n_devices, device_ids);
      if (nr != ncclSuccess) {
        why = std::string("ncclCommInitAll: ") + (rccl_fail_hook(c) == 1 ? "simulated failure (PTMI_TEST_RCCL_FAIL=init)" : g_rccl.GetErrorString(nr));
        c->comms.clear();
        (void)hipGetLastError();
        (void)hipSetDevice(c->device);
      }
    }
    if (why.empty()) {
      c->reduce_mode = 1;
      c->reduce_info = "ncclReduce over " + std::to_string(n_devices) + " device" + (n_devices > 1 ? "s" : "") + " (RCCL, xGMI)";
    } else {
      c->use_rccl = false;
      c->reduce_mode = 3;
      c->reduce_info = "FALLBACK: hipMemcpyPeer + add (" + why + ")";
    }
  }
  apply_shard(c);
  *out = c;
  return PTMI_OK;
}

void ptmi_destroy(ptmi_ctx* c) {
  if (!c) return;
  for (ptmi_ctx* q : c->peers) {
    if (q->stream) {
      (void)hipSetDevice(q->device);
      (void)hipStreamSynchronize(q->stream);
    }
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (ncclComm_t cm : c->comms)
    if (cm) (void)g_rccl.CommDestroy(cm);
  c->comms.clear();
  for (ptmi_ctx* q : c->peers) ptmi_destroy(q);
  c->peers.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drain_spans(c);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (DBuf* b : {&c->d_quad_unit_n, &c->d_spheres, &c->d_sphere_info, &c->d_quads, &c->d_quad_mat, &c->d_tris, &c->d_pretri, &c->d_trinorm, &c->d_meshes, &c->d_xforms,
                  &c->d_mats, &c->d_pairs, &c->d_leaf_table, &c->d_fb_own, &c->d_q0[0], &c->d_q0[1], &c->d_q1[0], &c->d_q1[1], &c->d_q2[0],
                  &c->d_q2[1], &c->d_tp[0], &c->d_tp[1], &c->d_hm[0], &c->d_hm[1], &c->d_uv, &c->d_acc, &c->d_pixsum, &c->d_touched, &c->d_ctl, &c->d_totals,
                  &c->d_scratch, &c->d_spill, &c->d_heads, &c->d_fb_gather, &c->d_fb_stage, &c->d_bvh_rows, &c->d_carry[0], &c->d_carry[1]})
    b->release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c->worker;
  delete c;
}

int ptmi_set_params(ptmi_ctx* c, const ptmi_params* p) {
  if (!c || !p) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_params: null argument");
  if (p->num_samples < 1 || p->max_bounces < 0 || p->stack_size < 1 || p->stack_size > 64 || p->frames_in_flight < 0)
    return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_params: need num_samples >= 1, max_bounces >= 0, 1 <= stack_size <= 64, frames_in_flight >= 0");
  if ((int64_t)p->num_samples * p->max_bounces > 65536)
    return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_set_params: num_samples * max_bounces > 65536");
  if (!(p->fov_degrees > 0.0f && p->fov_degrees < 180.0f)) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_params: fov_degrees must be in (0,180)");
  if (!(p->tmin >= 0.0f && p->tmin < 3.0e38f)) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_params: tmin must be finite and >= 0 (the reference: 0.000001)");
  if (!(p->light_mix >= 0.0f && p->light_mix <= 1.0f)) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_params: light_mix must be in [0,1] (the reference: 0.2)");
  c->prm = *p;
  c->S.tmin = p->tmin;  // (the kernels take it with the scene; prepare_scene sets it too)
  c->ahead.valid = false;
  for (ptmi_ctx* q : c->peers) {
    q->prm = *p;
    q->S.tmin = p->tmin;
    q->ahead.valid = false;
  }
  return PTMI_OK;
}

int ptmi_reload_tuning(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  for (size_t i = 0; i <= c->peers.size(); i++) {
    ptmi_ctx* q = i ? c->peers[i - 1] : c;
    const int slots_before = q->tun.bvh_carry_slots;
    load_tuning(q);
    q->ahead.valid = false;
    if (q->tun.bvh_carry_slots > slots_before) q->path_cap = 0;  // the queues were sized with the old carry prefix (ensure_paths): allocate them again
  }
  return PTMI_OK;
}

int ptmi_get_params(const ptmi_ctx* c, ptmi_params* p) {
  if (!c || !p) return PTMI_ERR_INVALID_ARG;
  *p = c->prm;
  return PTMI_OK;
}

// One device's share of ptmi_upload.  The triangles (the largest buffer: 84 MB at 871 k triangles) go straight from the caller's memory to
// the device — no host copy — in two steps, so that a failure leaves every device of a multi-device context with the scene it had:
// upload_stage allocates what the new array needs WITHOUT touching the old one; upload_commit copies and swaps.
static int upload_stage(ptmi_ctx* c, int which, size_t bytes, DBuf* fresh) {
  if (which != PTMI_BUF_TRIANGLES || bytes <= c->d_tris.cap) return PTMI_OK;  // fits what is there: nothing to allocate
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, fresh->ensure(std::max<size_t>(bytes, 16)));
  return PTMI_OK;
}
static int upload_commit(ptmi_ctx* c, int which, const void* data, size_t bytes, DBuf* fresh) {
  const float* f = (const float*)data;
  switch (which) {
    case PTMI_BUF_SPHERES: c->h_spheres.assign(f, f + bytes / 4); break;
    case PTMI_BUF_QUADS: c->h_quads.assign(f, f + bytes / 4); break;
    case PTMI_BUF_TRIANGLES: {
      hipError_t e = hipSetDevice(c->device);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // nothing in flight may still read the old triangles
      if (e == hipSuccess && fresh->p) {
        c->d_tris.release();
        c->d_tris = *fresh;
        *fresh = DBuf();
      }
      if (e == hipSuccess && bytes) e = hipMemcpy(c->d_tris.p, data, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        // the device no longer holds a valid triangle array: make the context say so instead of tracing stale or freed memory
        c->n_tris_uploaded = 0;
        c->S.tris = nullptr;
        c->S.n_tris = 0;
        c->scene_dirty = true;
        c->ahead.valid = false;
        return fail(c, PTMI_ERR_DEVICE, std::string("ptmi_upload(triangles): ") + hipGetErrorString(e) + " — the triangle buffer is now empty; upload it again");
      }
      c->n_tris_uploaded = bytes / 96;
      c->bvh_dev_stale = true;  // (meaningful only while bvh_on_device: mesh-order triangles under a tree built over leaf-order ones)
      break;
    }
    case PTMI_BUF_MESHES:
      c->h_meshes.assign((const int32_t*)data, (const int32_t*)data + bytes / 4);
      c->bvh_dev_stale = true;  // the boxes were made with the old meshes' transform ids
      break;
    case PTMI_BUF_TRANSFORMS:
      c->h_xforms.assign(f, f + bytes / 4);
      c->bvh_dev_stale = true;  // world-space boxes of the old transforms
      break;
    case PTMI_BUF_MATERIALS: c->h_mats.assign(f, f + bytes / 4); break;
    case PTMI_BUF_BVH:
      c->h_bvh.assign(f, f + bytes / 4);
      c->bvh_on_device = false;  // an uploaded BVH replaces a device-resident one
      break;
  }
  c->scene_dirty = true;
  c->ahead.valid = false;
  return PTMI_OK;
}

int ptmi_upload(ptmi_ctx* c, int which, const void* data, size_t bytes) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  size_t stride = stride_of(which);
  if (!stride) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_upload: unknown buffer id");
  if (bytes % stride) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_upload: byte size is not a multiple of the buffer's stride");
  if (bytes && !data) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_upload: data is null");
  if (bytes / stride > 0x0fffffffull) return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_upload: more than 2^28-1 elements");
  // the scene is replicated on every device of a multi-device context: stage everywhere first, commit only when every device can take it
  const size_t n = c->peers.size() + 1;
  std::vector<DBuf> fresh(n);
  for (size_t i = 0; i < n; i++) {
    ptmi_ctx* q = i ? c->peers[i - 1] : c;
    const int r = upload_stage(q, which, bytes, &fresh[i]);
    if (r) {
      for (size_t k = 0; k <= i; k++) {
        (void)hipSetDevice((k ? c->peers[k - 1] : c)->device);
        fresh[k].release();
      }
      (void)hipSetDevice(c->device);
      return i ? fail(c, r, "local device #" + std::to_string(i) + ": " + q->err + " (no device was changed)") : r;
    }
  }
  int rc = PTMI_OK;
  for (size_t i = 0; i < n; i++) {
    ptmi_ctx* q = i ? c->peers[i - 1] : c;
    const int r = upload_commit(q, which, data, bytes, &fresh[i]);
    if (r && !rc) rc = i ? fail(c, r, "local device #" + std::to_string(i) + ": " + q->err) : r;
  }
  for (size_t k = 0; k < n; k++) fresh[k].release();  // (only after a device error during the commit)
  (void)hipSetDevice(c->device);
  return rc;
}

// lib/scene.js:253-259 (create_bvh + the reordering of the triangles) on the GPU, over what is already there.
static int build_scene_bvh_one(ptmi_ctx* c, bool sah) {
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t n = c->n_tris_uploaded;
  if (n == 0) {
    c->bvh_on_device = false;
    c->h_bvh.clear();
    c->scene_dirty = true;
    return PTMI_OK;
  }
  // node and primitive ids travel as f32 in the reference's rows (lib/BVH/bvhBuilder.js:45,49) and come back out of them when the pair records are
  // made: exact only below 2^24 = 2n - 1 nodes
  if (n > ((size_t)1 << 23)) return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_build_scene_bvh: more than 2^23 triangles (node ids are f32 in the BVH rows, exact below 2^24)");
  const int n_mesh = (int)(c->h_meshes.size() / 4), n_xf = (int)(c->h_xforms.size() / 32);
  DBuf rows, tris2;
  hipError_t e = rows.ensure((2 * n - 1) * 48);
  if (e == hipSuccess) e = tris2.ensure(n * 96);
  if (e != hipSuccess) {
    rows.release();
    tris2.release();
    return fail(c, PTMI_ERR_NO_MEMORY, std::string("ptmi_build_scene_bvh: ") + hipGetErrorString(e));
  }
  int depth = 0;
  uint32_t bad = 0xffffffffu, n_nodes = 0;
  const int r = ptmi_bvhdev_build_scene((void*)c->stream, c->d_tris.as<float>(), (uint32_t)n, c->h_meshes.data(), n_mesh, c->h_xforms.data(), n_xf, rows.as<float>(), tris2.as<float>(),
                                         &depth, &bad, sah ? 1 : 0, &n_nodes);
  if (r || bad != 0xffffffffu) {
    rows.release();
    tris2.release();
    if (r == (int)hipErrorNotSupported) return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_build_scene_bvh_sah: the SAH tree of these triangles is deeper than 4096 levels (no STACK_SIZE <= 64 can traverse it)");
    if (r) return fail(c, r == (int)hipErrorOutOfMemory ? PTMI_ERR_NO_MEMORY : PTMI_ERR_DEVICE, std::string("ptmi_build_scene_bvh: ") + hipGetErrorString((hipError_t)r));
    char msg[160];
    snprintf(msg, sizeof msg, "ptmi_build_scene_bvh: triangle %u: mesh_id / the mesh's global_id out of range (meshes %d, transforms %d)", bad, n_mesh, n_xf);
    return fail(c, PTMI_ERR_BAD_SCENE, msg);
  }
  c->d_tris.release();  // the triangles now sit in leaf order (the reference reorders them on the host, lib/scene.js:257)
  c->d_tris = tris2;
  c->d_bvh_rows.release();
  c->d_bvh_rows = rows;
  c->bvh_on_device = true;
  c->bvh_dev_stale = false;
  c->bvh_dev_prims = n;
  c->bvh_dev_nodes = n_nodes;
  c->bvh_dev_sah = sah;
  c->bvh_dev_depth = depth;
  c->h_bvh.clear();
  c->scene_dirty = true;
  c->ahead.valid = false;
  return PTMI_OK;
}

int ptmi_build_scene_bvh(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  return on_all_devices(c, [](ptmi_ctx* q) { return build_scene_bvh_one(q, false); }, true);
}

int ptmi_build_scene_bvh_sah(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  return on_all_devices(c, [](ptmi_ctx* q) { return build_scene_bvh_one(q, true); }, true);
}

int ptmi_scene_bvh_info(ptmi_ctx* c, uint64_t* n_nodes, int32_t* depth, int32_t* on_device) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  if (n_nodes) *n_nodes = c->bvh_on_device ? (uint64_t)c->bvh_dev_nodes : (uint64_t)(c->h_bvh.size() / 12);
  if (depth) *depth = c->bvh_on_device ? c->bvh_dev_depth : c->bvh_depth;
  if (on_device) *on_device = c->bvh_on_device ? 1 : 0;
  return PTMI_OK;
}

int ptmi_read_scene_buffer(ptmi_ctx* c, int which, void* dst, size_t bytes) {
  if (!c || !dst) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_scene_buffer: null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const void* src = nullptr;
  size_t have = 0;
  if (which == PTMI_BUF_TRIANGLES) src = c->d_tris.p, have = c->n_tris_uploaded * 96;
  else if (which == PTMI_BUF_BVH && c->bvh_on_device) src = c->d_bvh_rows.p, have = c->bvh_dev_nodes * 48;
  else if (which == PTMI_BUF_BVH) {
    have = c->h_bvh.size() * 4;
    if (bytes != have) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_scene_buffer: byte size differs from the buffer's");
    memcpy(dst, c->h_bvh.data(), have);
    return PTMI_OK;
  } else return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_scene_buffer: triangles (5) or bvh (9) only");
  if (bytes != have) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_scene_buffer: byte size differs from the buffer's");
  if (have) HIP_TRY(c, hipMemcpy(dst, src, have, hipMemcpyDeviceToHost));
  return PTMI_OK;
}

int ptmi_resize(ptmi_ctx* c, int width, int height) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  if (width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 28))
    return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_resize: need 0 < W*H <= 2^28");
  HIP_TRY(c, hipSetDevice(c->device));
  size_t bytes = (size_t)width * height * 16;
  c->ahead.valid = false;
  HIP_TRY(c, c->d_fb_own.ensure(bytes));
  c->fb = c->d_fb_own.as<float4>();
  c->fb_bytes = bytes;
  c->W = width;
  c->H = height;
  HIP_TRY(c, hipMemsetAsync(c->fb, 0, bytes, c->stream));
  for (ptmi_ctx* q : c->peers) {
    int r = ptmi_resize(q, width, height);
    if (r) return fail(c, r, q->err);
  }
  return PTMI_OK;
}

int ptmi_clear_framebuffer(ptmi_ctx* c) {
  if (!c || !c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_clear_framebuffer: no framebuffer");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(c->fb, 0, c->fb_bytes, c->stream));
  for (ptmi_ctx* q : c->peers) {
    int r = ptmi_clear_framebuffer(q);
    if (r) return fail(c, r, q->err);
  }
  return PTMI_OK;
}

int ptmi_set_shard(ptmi_ctx* c, int rank, int world, int tile_pixels) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  if (world < 1 || rank < 0 || rank >= world || tile_pixels < 1) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_shard: need 0 <= rank < world, tile_pixels >= 1");
  if (c->multi) {
    if ((int64_t)world * (int64_t)(c->peers.size() + 1) > (1 << 30)) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_set_shard: world too large");
    c->proc_rank = rank, c->proc_world = world, c->proc_tile = tile_pixels;
    return apply_shard(c);
  }
  c->rank = rank, c->world = world, c->tile = tile_pixels;
  c->ahead.valid = false;
  return PTMI_OK;
}

static int render_frame_one(ptmi_ctx* c, const float* u) {
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // a stale error of an earlier, already reported failure must not be blamed on this call
  int r = prepare_scene(c);
  if (r) return r;
  r = check_renderable(c);
  if (r) return r;
  if (u[0] != (float)c->W || u[1] != (float)c->H) return fail(c, PTMI_ERR_STATE, "ptmi_render_frame: uniforms screenDims differ from ptmi_resize");
  const uint32_t k = u32_of_f32(u[2]);
  const int reset = u[3] == 0.0f ? 0 : 1;
  const float* view = u + 4;
  // Render-ahead.  The reference's loop asks for one frame at a time (renderer.js:173-188) and a lone frame cannot fill the
  // machine: every k_bvh launch lasts as long as its longest ray (7.6 ms per 1080p frame on an 871 k-triangle scene against
  // 0.8 ms per frame in a batch).  Once the camera has been at rest for a few frames, the frames that will be asked for
  // next are traced in the same pass; their colours wait in d_acc and each later call only adds its frame to the
  // framebuffer — bit for bit what tracing it then would have produced.  A different view, a reset, a gap in the frame
  // numbers or any call that touches the path buffers drops what was rendered ahead.  PTMI_RENDER_AHEAD=0 turns it off.
  ptmi_ctx::Ahead& A = c->ahead;
  const bool follows = c->have_last_frame && !reset && k == c->last_frame + 1u && memcmp(view, c->last_view, 64) == 0;
  c->static_streak = follows ? c->static_streak + 1 : 0;
  c->have_last_frame = true;
  c->last_frame = k;
  memcpy(c->last_view, view, 64);
  if (A.valid && follows && A.next < A.count && k == A.frame0 + (uint32_t)A.next && memcmp(view, A.view, 64) == 0) {
    hipLaunchKernelGGL(k_accumulate, dim3(A.grid), dim3(kBlock), 0, c->stream, A.rc, paths_of(c, 0, A.rc.num_samples > 1), c->fb, 0,
                       c->d_totals.as<unsigned long long>(), A.next, A.next + 1);
    HIP_TRY(c, hipGetLastError());
    A.next++;
    return PTMI_OK;
  }
  int batch = 1;
  if (c->static_streak >= 2 && !c->counters && c->timing == 0 && c->tun.render_ahead) {
    const size_t npix = std::max<size_t>(1, count_local((uint32_t)c->W * (uint32_t)c->H, c->rank, c->world, c->tile));
    batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)c->ahead_batch, ((size_t)1 << 29) / npix));
    c->ahead_batch = std::min(64, c->ahead_batch * 2);  // 8, 16, 32, 64 frames while the camera stays put
  } else {
    c->ahead_batch = 8;
  }
  c->interactive = true;
  r = render_batch(c, view, k, batch, reset, 1);
  c->interactive = false;
  if (r) return r;
  if (batch > 1) {
    A.valid = true;
    memcpy(A.view, view, 64);
    A.frame0 = k;
    A.count = batch;
    A.next = 1;
  }
  return PTMI_OK;
}

int ptmi_render_frame(ptmi_ctx* c, const float* u) {
  if (!c || !u) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_render_frame: null argument");
  return on_all_devices(c, [u](ptmi_ctx* q) { return render_frame_one(q, u); }, true);
}

static int render_one(ptmi_ctx* c, const float* view16, uint32_t first_frame, uint32_t n_frames) {
  HIP_TRY(c, hipSetDevice(c->device));
  (void)hipGetLastError();  // a stale error of an earlier, already reported failure must not be blamed on this call
  int r = prepare_scene(c);
  if (r) return r;
  r = check_renderable(c);
  if (r) return r;
  // pixels this context owns; a rank that renders 1/N of the image keeps N times more frames in flight
  const size_t npix = std::max<size_t>(1, count_local((uint32_t)c->W * (uint32_t)c->H, c->rank, c->world, c->tile));
  // auto: as many frames per wavefront pass as a 2^30-path budget allows (512 at 1080p, 128 at 4K; ~160 B of state per path
  // = 160 of the 288 GB; a board — or what is left of one — that cannot hold that gets half, and half again: render_one below).  Every k_bvh launch ends with a
  // tail as long as its longest ray (~2 ms per step on an 871 k-triangle tree, whatever the batch size: configs[2] gains 13 % from 64 -> 128 frames), and the
  // sparse Russian-roulette steps and the launches are amortised over more rays too.  Round 5: 2^29 -> 2^30 — 4K at 128 spp +2.5 %, 512 spp of the 871 k-triangle
  // scene +4 %, of the 262 k interior +0.4 % (profiles/r05_path_budget_ab.txt; 2^27 loses 2-8 %).  PTMI_PATH_BUDGET_LOG2 overrides (tests, smaller boards).
  const int budget_log2 = c->tun.path_budget_log2;
  uint32_t F = c->prm.frames_in_flight > 0 ? (uint32_t)c->prm.frames_in_flight : (uint32_t)std::max<size_t>(1, std::min<size_t>(1024, ((size_t)1 << budget_log2) / npix));
  size_t max_f = std::max<size_t>(1, ((size_t)1 << 31) / npix);  // slot indices (paths + 1/8 + holes) stay below 2^32
  F = (uint32_t)std::min<size_t>(F, max_f);
  for (uint32_t done = 0; done < n_frames;) {
    uint32_t nb = std::min(F, n_frames - done);
    r = render_batch(c, view16, first_frame + done, (int)nb, 0);
    if (r == PTMI_ERR_NO_MEMORY && !c->batch_enqueued && nb > 1 && c->prm.frames_in_flight <= 0) {
      // the automatic budget did not fit this board (or what is left of it) and the path buffers could not be allocated —
      // nothing of this batch has been enqueued, so halving and rendering the same frames again cannot count them twice
      F = std::max<uint32_t>(1, nb / 2);
      continue;
    }
    if (r) return r;
    done += nb;
  }
  return PTMI_OK;
}

int ptmi_render(ptmi_ctx* c, const float* view16, uint32_t first_frame, uint32_t n_frames) {
  if (!c || !view16) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_render: null argument");
  return on_all_devices(c, [=](ptmi_ctx* q) { return render_one(q, view16, first_frame, n_frames); }, true);
}

static int synchronize_one(ptmi_ctx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  return check_queue_overflow(c);
}

int ptmi_synchronize(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  return on_all_devices(c, synchronize_one);
}

int ptmi_prepare(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  return on_all_devices(c, [](ptmi_ctx* q) -> int {
    HIP_TRY(q, hipSetDevice(q->device));
    int r = prepare_scene(q);
    if (r) return r;
    HIP_TRY(q, hipStreamSynchronize(q->stream));
    return PTMI_OK;
  }, true);
}

int ptmi_read_framebuffer(ptmi_ctx* c, float* dst, size_t bytes) {
  if (!c || !dst) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_framebuffer: null argument");
  if (!c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_read_framebuffer: no framebuffer");
  if (bytes != (size_t)c->W * c->H * 16) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_read_framebuffer: bytes != W*H*16");
  HIP_TRY(c, hipSetDevice(c->device));
  float4* src = nullptr;
  int r = gather_framebuffer(c, &src);  // multi-device: the one reduce of the render (RCCL over xGMI); else c->fb itself
  if (r) return r;
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  return on_all_devices(c, [](ptmi_ctx* q) -> int {
    HIP_TRY(q, hipSetDevice(q->device));
    drain_spans(q);
    return check_queue_overflow(q);
  });
}

int ptmi_reduce_framebuffer(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  if (!c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_reduce_framebuffer: no framebuffer");
  HIP_TRY(c, hipSetDevice(c->device));
  float4* src = nullptr;
  int r = gather_framebuffer(c, &src);
  if (r) return r;
  if (!c->multi) HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PTMI_OK;
}

int ptmi_write_framebuffer(ptmi_ctx* c, const float* src, size_t bytes) {
  if (!c || !src) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_write_framebuffer: null argument");
  if (!c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_write_framebuffer: no framebuffer");
  if (bytes != (size_t)c->W * c->H * 16) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_write_framebuffer: bytes != W*H*16");
  if (c->multi && !c->peers.empty()) {
    // every device gets its own tiles of the image and zeros elsewhere, as if it had rendered them itself
    const size_t npix = (size_t)c->W * c->H;
    std::vector<float> part(npix * 4);
    for (size_t i = 0; i <= c->peers.size(); i++) {
      ptmi_ctx* q = i ? c->peers[i - 1] : c;
      for (size_t px = 0; px < npix; px++) {
        const bool mine = (px / (size_t)q->tile) % (size_t)q->world == (size_t)q->rank;
        for (int k = 0; k < 4; k++) part[4 * px + k] = mine ? src[4 * px + k] : 0.0f;
      }
      HIP_TRY(c, hipSetDevice(q->device));
      q->ahead.valid = false;
      HIP_TRY(c, hipMemcpyAsync(q->fb, part.data(), bytes, hipMemcpyHostToDevice, q->stream));
      HIP_TRY(c, hipStreamSynchronize(q->stream));
    }
    HIP_TRY(c, hipSetDevice(c->device));
    return PTMI_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(c->fb, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PTMI_OK;
}

int ptmi_framebuffer_device_ptr(ptmi_ctx* c, void** p, size_t* bytes) {
  if (!c || !p) return PTMI_ERR_INVALID_ARG;
  if (!c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_framebuffer_device_ptr: no framebuffer");
  if (!c->peers.empty()) return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_framebuffer_device_ptr: a multi-device context has one buffer per GPU; use ptmi_read_framebuffer");
  *p = c->fb;
  if (bytes) *bytes = (size_t)c->W * c->H * 16;
  return PTMI_OK;
}

int ptmi_bind_framebuffer(ptmi_ctx* c, void* dev_ptr, size_t bytes) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  if (c->W <= 0) return fail(c, PTMI_ERR_STATE, "ptmi_bind_framebuffer: call ptmi_resize first");
  if (!c->peers.empty()) return fail(c, PTMI_ERR_UNSUPPORTED, "ptmi_bind_framebuffer: not available on a multi-device context");
  if (!dev_ptr || ((uintptr_t)dev_ptr & 15) || bytes < (size_t)c->W * c->H * 16)
    return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_bind_framebuffer: need a 16-byte aligned device pointer of >= W*H*16 bytes");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->ahead.valid = false;
  c->fb = (float4*)dev_ptr;
  c->fb_bytes = (size_t)c->W * c->H * 16;
  return PTMI_OK;
}

int ptmi_stream(ptmi_ctx* c, void** stream) {
  if (!c || !stream) return PTMI_ERR_INVALID_ARG;
  *stream = (void*)c->stream;
  return PTMI_OK;
}

int ptmi_resolve_rgba8(ptmi_ctx* c, float frame_num, uint8_t* dst, size_t bytes) {
  if (!c || !dst) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_resolve_rgba8: null argument");
  if (!c->fb) return fail(c, PTMI_ERR_STATE, "ptmi_resolve_rgba8: no framebuffer");
  size_t npix = (size_t)c->W * c->H;
  if (bytes != npix * 4) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_resolve_rgba8: bytes != W*H*4");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->d_scratch.ensure(bytes));
  float4* src = nullptr;
  int gr = gather_framebuffer(c, &src);
  if (gr) return gr;
  hipLaunchKernelGGL(k_resolve_rgba8, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, src, (uint32_t)npix, frame_num,
                     c->d_scratch.as<uchar4>());
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(dst, c->d_scratch.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PTMI_OK;
}

int ptmi_set_counters(ptmi_ctx* c, int on) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  c->counters = on != 0;
  for (ptmi_ctx* q : c->peers) q->counters = c->counters;
  return PTMI_OK;
}
int ptmi_set_timing(ptmi_ctx* c, int on) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  c->timing = (on < 0 || on > 6) ? 0 : on;
  for (ptmi_ctx* q : c->peers) q->timing = c->timing;
  return PTMI_OK;
}

static int get_stats_one(ptmi_ctx* c, ptmi_stats* out) {
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  if (c->d_totals.p) {
    unsigned long long t[16];
    HIP_TRY(c, hipMemcpy(t, c->d_totals.p, sizeof t, hipMemcpyDeviceToHost));
    c->stats.rays = t[0], c->stats.paths = t[1], c->stats.node_visits = t[2], c->stats.tri_tests = t[3];
    c->stats.sphere_tests = t[4], c->stats.quad_tests = t[5], c->stats.mat_fetches = t[6];
    c->stats.bvh_node_visits = t[7], c->stats.bvh_mat_fetches = t[8];
    if (t[15]) return fail(c, PTMI_ERR_STATE, "internal: a step's queue outgrew its buffer (paths were dropped)");
  }
  c->stats.devices = 1;
  c->stats.reduce_mode = (uint64_t)c->reduce_mode;
  c->stats.peer_links = (uint64_t)c->peer_links;
  *out = c->stats;
  return PTMI_OK;
}

int ptmi_get_stats(ptmi_ctx* c, ptmi_stats* out) {
  if (!c || !out) return PTMI_ERR_INVALID_ARG;
  int r = get_stats_one(c, out);
  if (r) return r;
  for (ptmi_ctx* q : c->peers) {  // counters add up over the local devices, times are the slowest device's
    ptmi_stats s;
    r = get_stats_one(q, &s);
    if (r) return fail(c, r, q->err);
    out->rays += s.rays, out->paths += s.paths, out->node_visits += s.node_visits, out->tri_tests += s.tri_tests;
    out->sphere_tests += s.sphere_tests, out->quad_tests += s.quad_tests, out->mat_fetches += s.mat_fetches;
    out->bvh_node_visits += s.bvh_node_visits, out->bvh_mat_fetches += s.bvh_mat_fetches;
    out->intersect_launches += s.intersect_launches, out->shade_launches += s.shade_launches;
    out->generate_launches += s.generate_launches, out->accumulate_launches += s.accumulate_launches, out->tail_launches += s.tail_launches;
    out->render_ms = std::max(out->render_ms, s.render_ms), out->intersect_ms = std::max(out->intersect_ms, s.intersect_ms);
    out->shade_ms = std::max(out->shade_ms, s.shade_ms), out->other_ms = std::max(out->other_ms, s.other_ms);
    out->prims_ms = std::max(out->prims_ms, s.prims_ms), out->bvh_ms = std::max(out->bvh_ms, s.bvh_ms);
    out->generate_ms = std::max(out->generate_ms, s.generate_ms), out->accumulate_ms = std::max(out->accumulate_ms, s.accumulate_ms);
    out->tail_ms = std::max(out->tail_ms, s.tail_ms);
    out->devices += 1;
  }
  return PTMI_OK;
}

static int reset_stats_one(ptmi_ctx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  const uint64_t ps = c->stats.placement_sets;  // facts about the context's allocation, not counters of the interval
  const double pm = c->stats.placement_ms;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.placement_sets = ps, c->stats.placement_ms = pm;
  if (c->d_totals.p) HIP_TRY(c, hipMemset(c->d_totals.p, 0, 16 * sizeof(unsigned long long)));
  return PTMI_OK;
}

int ptmi_reset_stats(ptmi_ctx* c) {
  if (!c) return PTMI_ERR_INVALID_ARG;
  return on_all_devices(c, reset_stats_one);
}

int ptmi_trace(ptmi_ctx* c, size_t n, const float* rays6, uint32_t* rng_inout, ptmi_hit* out) {
  if (!c || !rays6 || !out) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_trace: null argument");
  if (n == 0) return PTMI_OK;
  if (n > 0x0fffffffull) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_trace: too many rays");
  static_assert(sizeof(HitOut) == sizeof(ptmi_hit), "hit record layouts must match");
  HIP_TRY(c, hipSetDevice(c->device));
  int r = prepare_scene(c);
  if (r) return r;
  c->ahead.valid = false;
  r = ensure_paths(c, n, 4, false);
  if (r) return r;
  Paths P = paths_of(c, 0, false);
  // stage: one slot per ray (slot i = path i)
  std::vector<float> so(4 * n, 0.0f), sd(4 * n, 0.0f);
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) so[4 * i + k] = rays6[6 * i + k], sd[4 * i + k] = rays6[6 * i + 3 + k];
    const uint32_t ident = (uint32_t)i, rng = rng_inout ? rng_inout[i] : 0u;
    memcpy(&so[4 * i + 3], &rng, 4);
    memcpy(&sd[4 * i + 3], &ident, 4);
  }
  StepCtl ctl0{};
  ctl0.n_rays = (uint32_t)n;
  StepCtl* ctl = c->d_ctl.as<StepCtl>();
  HIP_TRY(c, hipMemcpyAsync(P.in.q0, so.data(), so.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(P.in.q1, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(ctl, &ctl0, sizeof ctl0, hipMemcpyHostToDevice, c->stream));
  r = launch_intersect(c, P, ctl, (uint32_t)n, true);
  if (r) return r;
  HIP_TRY(c, c->d_scratch.ensure(n * sizeof(HitOut)));
  hipLaunchKernelGGL(k_resolve_hits, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->S, P, (uint32_t)n, c->d_scratch.as<HitOut>());
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch.p, n * sizeof(HitOut), hipMemcpyDeviceToHost, c->stream));
  if (rng_inout) HIP_TRY(c, hipMemcpyAsync(so.data(), P.in.q0, so.size() * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (rng_inout)
    for (size_t i = 0; i < n; i++) memcpy(&rng_inout[i], &so[4 * i + 3], 4);
  return PTMI_OK;
}

#ifdef PTMI_LANE_TALLY
// measurement builds only (tools/shade_lanes.py): {visits, lanes} per tally point of k_shade since the last reset
int ptmi_lane_tally(ptmi_ctx* c, uint64_t* out, int n_points, int reset) {
  if (!c || !out || n_points < 0 || n_points > kLaneTallies) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_lane_tally: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_tally), (size_t)n_points * 16));
  if (reset) {
    static const unsigned long long zeros[kLaneTallies * 2] = {0};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_lane_tally), zeros, sizeof zeros));
  }
  return PTMI_OK;
}
// wave-cycles per region of k_shade since the last reset (TT() marks, ptmi_device.h)
int ptmi_time_tally(ptmi_ctx* c, uint64_t* out, int n_regions, int reset) {
  if (!c || !out || n_regions < 0 || n_regions > kTimeTallies) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_time_tally: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_time_tally), (size_t)n_regions * 8));
  if (reset) {
    static const unsigned long long zeros[2 * kTimeTallies] = {0};
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_time_tally), zeros, sizeof zeros));
  }
  return PTMI_OK;
}
// {wave-cycles, marks, lanes} per region of k_bvh (which = 0) or k_tail (which = 1) since the last reset (BT() marks, ptmi_kernels.h; tools/bvh_regions.py)
int ptmi_bvh_tally(ptmi_ctx* c, uint64_t* out, int n_regions, int reset_and_which) {
  if (!c || !out || n_regions < 0 || n_regions > kBvhTallies) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_bvh_tally: bad argument");
  const bool tail = (reset_and_which & 2) != 0, reset = (reset_and_which & 1) != 0;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  static const unsigned long long zeros[3 * kBvhTallies] = {0};
  if (tail) {
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_tally), (size_t)n_regions * 24));
    if (reset) HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_tail_tally), zeros, sizeof zeros));
  } else {
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bvh_tally), (size_t)n_regions * 24));
    if (reset) HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_tally), zeros, sizeof zeros));
  }
  return PTMI_OK;
}
#endif

int ptmi_selftest(ptmi_ctx* c, int which, uint64_t* mismatches, uint32_t* first_bad_bits) {
  if (!c || !mismatches) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_selftest: null argument");
  if (which < 0 || which > 7) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_selftest: unknown test id");
  HIP_TRY(c, hipSetDevice(c->device));
  DBuf d;
  HIP_TRY(c, d.ensure(16));
  int rc = PTMI_OK;
  do {
    unsigned long long init[2] = {0ull, 0xffffffffull};
    hipError_t e = hipMemcpyAsync(d.p, init, 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_selftest, dim3((unsigned)c->num_cus * 16), dim3(256), 0, c->stream, which, d.as<unsigned long long>(), reinterpret_cast<uint32_t*>(d.as<unsigned long long>() + 1));
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(init, d.p, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, PTMI_ERR_DEVICE, std::string("ptmi_selftest: ") + hipGetErrorString(e));
    *mismatches = init[0];
    if (first_bad_bits) *first_bad_bits = (uint32_t)init[1];
  } while (0);
  d.release();
  return rc;
}

int ptmi_math_eval(ptmi_ctx* c, int fn, size_t n, const float* x, const float* y, float* out) {
  if (!c || !x || !out) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_math_eval: null argument");
  if (fn < 0 || fn > 13) return fail(c, PTMI_ERR_INVALID_ARG, "ptmi_math_eval: unknown function id");
  if (n == 0) return PTMI_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  DBuf dx, dy, dout;
  HIP_TRY(c, dx.ensure(n * 4));
  HIP_TRY(c, dout.ensure(n * 4));
  if (y) HIP_TRY(c, dy.ensure(n * 4));
  int rc = PTMI_OK;
  do {
    hipError_t e = hipMemcpyAsync(dx.p, x, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && y) e = hipMemcpyAsync(dy.p, y, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, fn, n, dx.as<float>(), y ? dy.as<float>() : nullptr,
                         dout.as<float>());
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, PTMI_ERR_DEVICE, std::string("ptmi_math_eval: ") + hipGetErrorString(e));
  } while (0);
  dx.release();
  dy.release();
  dout.release();
  return rc;
}

}  // extern "C"
