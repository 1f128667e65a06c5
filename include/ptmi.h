/*
 * ptmi.h — C ABI of libptmi.so, the MI355X-native path-tracing integrator.
 *
 * The reference (Shridhar2602/WebGPU-Path-Tracer) has no FFI: its device boundary is WebGPU itself —
 * `class WebGPU` (webgpu-utils.js:1-212) as driven by `class Renderer` (renderer.js:68-124,163-215):
 * create a storage buffer from a typed array (x8), write 80 bytes of uniforms, dispatch ONE compute
 * entry point `computeFrameBuffer` (shaders/main.wgsl:1) per frame.  Each entry point below names the
 * reference call it replaces.  All buffers use the reference's own byte layouts;
 * the library copies on upload and never keeps caller pointers.
 *
 * A context is single-caller (not thread-safe).  Multi-GPU, two ways, both = disjoint pixel tiles per GPU and ONE sum-reduce
 * of the accumulation buffers: (a) ptmi_create_multi — one context drives several GPUs of the node, shards the pixel tiles
 * across them itself and reduces with RCCL (ncclReduce over xGMI) inside ptmi_read_framebuffer: what a single-process host
 * such as the Node program needs; (b) one process per GPU (ptmi_create + ptmi_set_shard) with the reduce done by the host
 * program's own collective (bench.py: torch.distributed).
 * Every call returns PTMI_OK (0) or a negative ptmi_status; ptmi_last_error() gives the message.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_API_VERSION 5

typedef struct ptmi_ctx ptmi_ctx;

typedef enum ptmi_status {
  PTMI_OK = 0,
  PTMI_ERR_INVALID_ARG = -1, /* null pointer, bad size/stride, bad enum                      */
  PTMI_ERR_DEVICE = -2,      /* a HIP call failed (message carries hipGetErrorString)        */
  PTMI_ERR_STATE = -3,       /* call order: render before resize, size mismatch with uniforms */
  PTMI_ERR_NO_MEMORY = -4,   /* host or device allocation failed                             */
  PTMI_ERR_BAD_SCENE = -5,   /* an index in the uploaded buffers is out of range / BVH not a tree */
  PTMI_ERR_UNSUPPORTED = -6  /* parameter combination outside the supported set               */
} ptmi_status;

/* Buffer ids = the WGSL @binding numbers of shaders/header.wgsl:15-23 (binding 0 = uniforms and
 * binding 3 = framebuffer have their own calls; binding 4 is unused in the reference too). */
typedef enum ptmi_buffer {
  PTMI_BUF_SPHERES = 1,    /*  8 f32 / sphere   — renderer.js:93,  lib/primitives/sphere.js:25-29  */
  PTMI_BUF_QUADS = 2,      /* 20 f32 / quad     — renderer.js:95,  lib/primitives/quad.js:21-36    */
  PTMI_BUF_TRIANGLES = 5,  /* 24 f32 / triangle — renderer.js:99,  lib/primitives/triangle.js:42-52 */
  PTMI_BUF_MESHES = 6,     /*  4 i32 / mesh     — renderer.js:94,  lib/primitives/mesh.js:58-63    */
  PTMI_BUF_TRANSFORMS = 7, /* 32 f32 / object   — renderer.js:97,  lib/transform.js:38-40          */
  PTMI_BUF_MATERIALS = 8,  /* 16 f32 / material — renderer.js:96,  lib/scene.js:261-273            */
  PTMI_BUF_BVH = 9         /* 12 f32 / node     — renderer.js:98,  lib/BVH/bvhBuilder.js:37-54     */
} ptmi_buffer;

/* The reference's compile-time knobs (shaders/header.wgsl:9-13, traceRay.wgsl:8, main.wgsl:7) as
 * run-time parameters.  ptmi_default_params() fills the reference's values. */
typedef struct ptmi_params {
  int32_t num_samples;         /* NUM_SAMPLES = 1                                             */
  int32_t max_bounces;         /* MAX_BOUNCES = 100                                           */
  int32_t stratify;            /* STRATIFY = false                                            */
  int32_t importance_sampling; /* IMPORTANCE_SAMPLING = false                                 */
  int32_t stack_size;          /* STACK_SIZE = 20 (traversal aborts when the stack fills, Q7) */
  float background[3];         /* (0,1,1)                                                     */
  float fov_degrees;           /* 60                                                          */
  int32_t frames_in_flight;    /* ptmi_render batches this many frames per wavefront pass; 0 = auto: a 2^30-path budget
                                * (512 frames at 1080p, ~160 GB of path state, halved while it does not fit; PTMI_PATH_BUDGET_LOG2 overrides) */
  float tmin;                  /* ray_tmin = 0.000001 (header.wgsl:37): lower end of every t interval and the triangle test's epsilon;
                                * >= 0 and finite                                             */
  float light_mix;             /* 0.2: probability of following the light sample and its weight in the mixture pdf, the surface
                                * sample gets 1 - light_mix = the shader's 0.8 (traceRay.wgsl:43,49); in [0,1]; importance sampling only */
  int32_t reserved[3];
} ptmi_params;

/* Exact work counters (device-side when ptmi_set_counters(ctx,1); `rays` and `paths` always) and
 * GPU timings (HIP events on the context's stream when ptmi_set_timing(ctx,1)). Cumulative since
 * ptmi_reset_stats. */
typedef struct ptmi_stats {
  uint64_t rays;         /* hitScene invocations (shaders/hitRay.wgsl:1)                      */
  uint64_t paths;        /* ray_color invocations (shaders/traceRay.wgsl:3)                   */
  uint64_t node_visits;  /* hit_aabb calls the REFERENCE traversal makes for these rays       */
  uint64_t tri_tests;    /* hit_triangle calls                                                */
  uint64_t sphere_tests; /* hit_sphere + hit_volume calls                                     */
  uint64_t quad_tests;   /* hit_quad calls                                                    */
  uint64_t mat_fetches;  /* `hitRec.material = materials[..]` executions                      */
  uint64_t frames;       /* frames rendered                                                   */
  uint64_t intersect_launches; /* steps: each launches k_prims and (if there are triangles) k_bvh */
  uint64_t shade_launches;
  uint64_t bvh_node_visits;    /* the part of node_visits made by k_bvh (everything below the root) */
  uint64_t bvh_mat_fetches;    /* the part of mat_fetches made by k_bvh (accepted triangle hits)    */
  double render_ms;      /* generate..accumulate, all batches                                 */
  double intersect_ms;   /* prims_ms + bvh_ms                                                 */
  double shade_ms;       /* sum over k_shade launches                                         */
  double other_ms;       /* k_generate + k_accumulate                                         */
  double prims_ms;       /* sum over k_prims launches (spheres, quads, root box)              */
  double bvh_ms;         /* sum over k_bvh launches (traversal)                               */
  double generate_ms;    /* sum over k_generate launches (part of other_ms)                   */
  double accumulate_ms;  /* sum over k_accumulate launches (part of other_ms)                 */
  uint64_t generate_launches;
  uint64_t accumulate_launches;
  uint64_t devices;      /* GPUs behind this context (ptmi_create_multi); counters are summed over them, times are
                          * the maximum over them                                            */
  double tail_ms;        /* sum over k_tail launches: short queues traced to the end in one launch (most decline
                          * at once); counts towards neither intersect_ms nor shade_ms       */
  uint64_t tail_launches;
  /* API v4 */
  uint64_t reduce_mode;  /* how this context sums its devices' accumulation buffers (ptmi_create_multi): 0 = single device, nothing to sum;
                          * 1 = ncclReduce (RCCL over xGMI); 2 = peer copies + add kernel (shards share a GPU, or PTMI_MULTI_REDUCE=copy);
                          * 3 = peer copies + add kernel as a FALLBACK after librccl failed to load / initialise / reduce (ptmi_reduce_info says why);
                          * 4 (API v5, the default of a multi-device context) = tile gather: every device's own tiles copied into place on the first device, 1/N of the
                          * bytes of a full-buffer reduce and no arithmetic (PTMI_MULTI_REDUCE=rccl / copy select 1 / 2) */
  uint64_t peer_links;   /* directed root<->peer device pairs with hipDeviceEnablePeerAccess in force */
  uint64_t placement_sets; /* queue-array sets the last placement search timed (0 = none ran; ensure_paths) */
  double placement_ms;   /* host time that search added to the render that allocated the path buffers */
} ptmi_stats;

/* One hitScene result, the fields of the reference's HitRecord (shaders/header.wgsl:119-125). */
typedef struct ptmi_hit {
  int32_t hit;
  float t;
  float p[3];
  float normal[3];
  int32_t front_face;
  float material[16];
} ptmi_hit;

int ptmi_version(void);
const char* ptmi_status_string(int status);
/* Message of the last failing call on this context ("" if none). ctx may be NULL: returns the
 * message of the last failing ptmi_create in this thread. */
const char* ptmi_last_error(const ptmi_ctx* ctx);

/* replaces WebGPU.init() (webgpu-utils.js:178-211): binds device `device_id`, creates one stream */
int ptmi_create(ptmi_ctx** out, int device_id);
/* The same for n_devices GPUs of this node behind ONE context: every call below is applied to all of them
 * (uploads are replicated, renders run concurrently, one stream per GPU); the pixel tiles of this context's shard are
 * dealt round-robin to the devices, and ptmi_read_framebuffer / ptmi_resolve_rgba8 first assemble the per-device accumulation
 * buffers in a gather buffer on device_ids[0] — by default every device's OWN tiles are read into place over xGMI (peer access; 1/N of
 * the bytes, no arithmetic); with PTMI_MULTI_REDUCE=rccl one ncclReduce of the full buffers (f32 sum, W*H*4 values) through librccl,
 * loaded on first use — so the caller sees one image, bit-identical to the single-GPU one.  The reference's caller
 * (renderer.js:91-124,184-191) needs no change.  A device id may be listed more than once (its shards then share that
 * GPU and are summed by a kernel instead: how the multi-device path is tested on a one-GPU box). */
int ptmi_create_multi(ptmi_ctx** out, const int* device_ids, int n_devices);
/* Number of HIP devices this process sees (0 if the runtime reports none or fails). */
int ptmi_device_count(void);
/* One line about the multi-device reduce of this context: "ncclReduce over 8 devices (RCCL, xGMI)", "add kernel (shards share a GPU)", or
 * "FALLBACK: hipMemcpyPeer + add (<what failed>)" when librccl could not be loaded, ncclCommInitAll failed or a reduce failed — the context
 * then keeps working through peer copies (same bits) instead of failing.  Owned by the context. */
const char* ptmi_reduce_info(const ptmi_ctx* ctx);
void ptmi_destroy(ptmi_ctx* ctx);

void ptmi_default_params(ptmi_params* p);
int ptmi_set_params(ptmi_ctx* ctx, const ptmi_params* p);
int ptmi_get_params(const ptmi_ctx* ctx, ptmi_params* p);
/* The PTMI_* tuning variables (kernel grid sizes, k_tail's hand-over limit, render-ahead, ...) are read from the environment ONCE, by
 * ptmi_create; this reads them again for a live context (tests and A/B scripts).  The render path never calls getenv. */
int ptmi_reload_tuning(ptmi_ctx* ctx);

/* replaces createStorageBuffer_WriteOnly(label, typedArray) (webgpu-utils.js:29-41, renderer.js:93-99):
 * copies `bytes` bytes (a multiple of the buffer's stride; 0 allowed = empty array). */
int ptmi_upload(ptmi_ctx* ctx, int which, const void* data, size_t bytes);

/* replaces createStorageBuffer_ReadWrite('frameNum buffer', Float32Array(W*H*4).fill(0))
 * (renderer.js:88,100): allocates and zeroes the W*H RGBA f32 accumulation buffer. */
int ptmi_resize(ptmi_ctx* ctx, int width, int height);
int ptmi_clear_framebuffer(ptmi_ctx* ctx);

/* Pixel-tile sharding for multi-GPU: this context renders only pixels p with
 * (p / tile_pixels) % world == rank; other pixels of its framebuffer stay untouched (zero).
 * On a multi-device context the n local devices subdivide this shard: device i renders the tiles of
 * rank * n + i out of world * n. */
int ptmi_set_shard(ptmi_ctx* ctx, int rank, int world, int tile_pixels);

/* replaces queue.writeBuffer(uniforms) + computePass(...) of one animation frame
 * (renderer.js:173-188, webgpu-utils.js:125-134): uniforms20 = [W, H, frameNum, resetBuffer,
 * viewMatrix[16] column-major].  Asynchronous; ordering = call order. */
int ptmi_render_frame(ptmi_ctx* ctx, const float* uniforms20);

/* n_frames consecutive ptmi_render_frame calls with frameNum = first_frame .. first_frame+n_frames-1,
 * resetBuffer = 0 and a fixed view matrix — the progressive-rendering steady state
 * (renderer.js:163-184), batched so that several frames are in flight per wavefront pass.
 * Result is bit-identical to the frame-by-frame calls. */
int ptmi_render(ptmi_ctx* ctx, const float* view16, uint32_t first_frame, uint32_t n_frames);

int ptmi_synchronize(ptmi_ctx* ctx);

/* Validates the uploaded buffers and builds the device-side digests now instead of inside the first render call
 * (the analogue of createBindGroup, webgpu-utils.js:100-123).  Synchronous, so that a caller can time scene set-up. */
int ptmi_prepare(ptmi_ctx* ctx);

/* Framebuffer access (the reference never reads back; COPY_SRC exists, webgpu-utils.js:47).
 * read/write synchronise the stream; bytes must be W*H*16. */
int ptmi_read_framebuffer(ptmi_ctx* ctx, float* rgba_sum, size_t bytes);
int ptmi_write_framebuffer(ptmi_ctx* ctx, const float* rgba_sum, size_t bytes);
/* The one collective of a multi-device render on its own (renderer.js has no counterpart: one GPU): waits for every device, then sums
 * the per-device accumulation buffers into the gather buffer on device_ids[0] — ncclReduce over xGMI, see ptmi_create_multi — without
 * copying anything to the host.  ptmi_read_framebuffer / ptmi_resolve_rgba8 do this themselves; bench.py times it as part of a step.
 * On a single-device context it only synchronises. */
int ptmi_reduce_framebuffer(ptmi_ctx* ctx);
/* Device pointer of the accumulation buffer (for an in-place RCCL reduce by the host program). */
int ptmi_framebuffer_device_ptr(ptmi_ctx* ctx, void** dev_ptr, size_t* bytes);
/* Use caller-owned device memory (>= W*H*16 bytes, 16-byte aligned) as the accumulation buffer. */
int ptmi_bind_framebuffer(ptmi_ctx* ctx, void* dev_ptr, size_t bytes);
/* hipStream_t the context launches on (as void*), so callers can record their own HIP events. */
int ptmi_stream(ptmi_ctx* ctx, void** stream);

/* Display pass (shaders/fragment.js:22-36, shaders/common.wgsl:273-282): color = fb/frameNum ->
 * ACES approximation -> pow(1/2.2) -> RGBA8.  dst = W*H*4 bytes on the host. */
int ptmi_resolve_rgba8(ptmi_ctx* ctx, float frame_num, uint8_t* dst, size_t bytes);

int ptmi_set_counters(ptmi_ctx* ctx, int enabled);
/* 0 = off; 1 = HIP events around every kernel launch; 2 / 3 / 4 / 5 / 6 = only around k_bvh / k_shade / k_generate /
 * k_accumulate / k_tail: fewer stream markers, for timing ONE kernel inside a region whose wall clock also matters. */
int ptmi_set_timing(ptmi_ctx* ctx, int enabled);
int ptmi_get_stats(ptmi_ctx* ctx, ptmi_stats* out); /* synchronises */
int ptmi_reset_stats(ptmi_ctx* ctx);

/* Test hook: hitScene (shaders/hitRay.wgsl:1-113) for n caller-supplied rays (6 f32 each: origin,
 * dir), each with its own RNG state (consumed by hit_volume only; may be NULL). */
int ptmi_trace(ptmi_ctx* ctx, size_t n, const float* rays6, uint32_t* rng_inout, ptmi_hit* out);
/* Test hook: evaluates include/ptmi_math.h functions ON THE DEVICE.
 * fn: 0 sin 1 cos 2 acos 3 log 4 log2 5 exp2 6 pow(x,y) 7 sqrt 8 min(x,y) 9 max(x,y) 10 x/y
 *     11..13 = components of (x, x*2^-20, x*2^20) / y through the device's vector division */
int ptmi_math_eval(ptmi_ctx* ctx, int fn, size_t n, const float* x, const float* y, float* out);

/* Test hook: exhaustive check ON THE DEVICE of the unary shortcuts the kernels use instead of the compiler's IEEE expansions
 * (csrc/ptmi_device.h): which = 0: 1/x, 1: sqrt(x), 2: the three-component reciprocal — each against the IEEE operation over all
 * 2^32 arguments; 3 / 4: the bare v_rcp_f32 / v_sqrt_f32 instructions (controls that must report mismatches).  *mismatches = number of arguments whose bits differ (NaNs compare equal), *first_bad_bits = the smallest. */
int ptmi_selftest(ptmi_ctx* ctx, int which, uint64_t* mismatches, uint32_t* first_bad_bits);

/* Scene.create_bvh() (lib/scene.js:253-259 -> lib/BVH/bvhBuilder.js:6, bvhNode.js:28-73) for the triangles that are ALREADY uploaded (binding 5,
 * in any order) with their meshes (6) and transforms (7): world-space boxes as lib/primitives/triangle.js:27-39 + AABB.js:35-51 compute
 * them, the median-split build, and the reordering of the triangles into leaf order (lib/scene.js:257) — all on the GPU, nothing comes back:
 * the BVH rows (byte-identical to ptmi_build_bvh's and the reference's) stay in device memory as binding 9 and the traversal digests are
 * made from them there.  871 k triangles: ~15 ms (the reference's JavaScript: seconds; benchmarks.txt:19).  A later ptmi_upload(PTMI_BUF_BVH)
 * replaces the tree; after an upload of triangles, meshes or transforms the tree is stale and every render fails with PTMI_ERR_BAD_SCENE
 * until it is built again (or a BVH is uploaded).  At most 2^23 triangles: node ids are f32 in the rows (exact below 2^24), as in the
 * reference's own format — more returns PTMI_ERR_UNSUPPORTED. */
int ptmi_build_scene_bvh(ptmi_ctx* ctx);
/* API v5.  The same with the reference's OTHER builder, BVH.generate_bvh_heirarchy_SAH (lib/BVH/bvhNode.js:108-283: 8 bins per axis, leaf when the best plane
 * costs no less than the node), which the reference ships but never calls (lib/BVH/bvhBuilder.js:10-13 takes the median split): an opt-in for callers
 * who want the cheaper traversal (benchmarks.txt:2-12) and accept a tree the reference's renderer does not use.  Level-synchronous on the GPU
 * (csrc/ptmi_bvh_device.hip), rows byte-identical to ptmi_build_bvh_sah's; leaves may hold several triangles, so the row count is not 2n - 1:
 * ptmi_scene_bvh_info reports it.  The image is the one the reference's shader renders from that tree (same traversal code), not the median tree's
 * bit for bit: triangle test order and the stack-depth abort (Q7) depend on the tree — use stack_size > depth. */
int ptmi_build_scene_bvh_sah(ptmi_ctx* ctx);
/* Rows (nodes) and depth (inner nodes on the longest root-to-leaf path; valid after ptmi_prepare for an uploaded tree) of the scene's BVH, and whether it
 * lives on the device only (ptmi_build_scene_bvh*).  Any pointer may be NULL. */
int ptmi_scene_bvh_info(ptmi_ctx* ctx, uint64_t* n_nodes, int32_t* depth, int32_t* on_device);
/* Test / tool hook: copies the context's triangles (which = 5: in their current order) or BVH rows (which = 9) to the host; bytes must be
 * exactly the buffer's size (triangles x 96, ptmi_scene_bvh_info's n_nodes x 48 for a device-resident tree). */
int ptmi_read_scene_buffer(ptmi_ctx* ctx, int which, void* dst, size_t bytes);

/* ---- host-side natives (no GPU needed) -------------------------------------------------------- */

/* Median-split BVH build + pre-order flatten with the reference's exact semantics
 * (lib/BVH/bvhNode.js:21-101, lib/BVH/bvhBuilder.js:6-54): prim boxes as n x 3 doubles each;
 * writes (2n-1) x 12 f32 nodes and the primitive permutation (order[k] = input index of the
 * primitive stored at position k). */
int ptmi_build_bvh(size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out,
                   int64_t* order_out);

/* Opt-in binned-SAH build: the reference's second builder, BVH.generate_bvh_heirarchy_SAH
 * (lib/BVH/bvhNode.js:108-283; 8 bins, leaves of any size), which the reference itself never calls —
 * its renderer uses the median split above.  Same inputs; nodes_out must hold (2n-1) x 12 floats,
 * *n_nodes_out receives the number of rows written.  Host threads: PTMI_BUILD_THREADS (default: all, at most 32);
 * the same bytes for any number of them. */
int ptmi_build_bvh_sah(size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out,
                       int64_t* order_out, size_t* n_nodes_out);

/* ptmi_build_bvh on the GPU of `ctx` (level-synchronous: segmented reduce for the boxes, stable segmented radix sort per
 * level; csrc/ptmi_bvh_device.hip).  Same arguments, byte-identical output.  871k boxes: 0.11 s including the transfers
 * (the host builder: 0.13-0.17 s on a 32-thread share of the GPU box, 1.75 s on 8 cores).  Synchronous. */
int ptmi_build_bvh_device(ptmi_ctx* ctx, size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out,
                          int64_t* order_out);
/* API v5.  ptmi_build_bvh_sah on the GPU of `ctx` (csrc/ptmi_bvh_device.hip: per level one reduce-by-key for boxes and centroid bounds, LDS-privatised
 * binning, one thread per node for the 21 candidate planes, the median builder's two stable radix sorts, an atomicMin for the split position; pre-order
 * ids from subtree sizes).  Same arguments, byte-identical output.  Synchronous. */
int ptmi_build_bvh_sah_device(ptmi_ctx* ctx, size_t n_prims, const double* bmin, const double* bmax, int prim_type, float* nodes_out,
                              int64_t* order_out, size_t* n_nodes_out);

/* OBJ text -> de-indexed vertex / normal arrays with the reference's accepted grammar and quirks
 * (lib/primitives/objReader.js:10-68: `v`, `vn`, `f a/b/c` triangles; tokens go through JS Number()).  The arrays are
 * malloc'ed; release them with ptmi_free.  Counts are in floats. */
int ptmi_obj_parse(const char* text, size_t len, float** vertices_out, size_t* n_vertices, float** normals_out, size_t* n_normals);
void ptmi_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
