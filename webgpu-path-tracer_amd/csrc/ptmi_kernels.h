// ptmi_kernels.h — the wavefront integrator's kernels (gfx950).
//
//   k_generate   one thread per (frame slot, owned pixel): seeds the RNG (main.wgsl:16), builds the camera ray
//                (shootRay.wgsl), runs hitScene part 1 on it, fills step 0's queue (slot = path id).
//   k_bvh        hitScene part 2 (hitRay.wgsl:42-110): persistent single-wave blocks, one ray per lane, LDS traversal
//                stacks, ballot-based lane refill by scanning the flags; no barriers, tails only at kernel end.
//   k_shade      per 512-slot chunk: LDS counting sort by material class (bin-uniform waves), ray_color's loop body
//                (traceRay.wgsl:10-80) + material_scatter + Russian roulette; the survivors' next state is staged in
//                LDS, gets hitScene part 1 (hitRay.wgsl:6-54: spheres, quads, root box) for the NEW ray and is written
//                densely into the next step's queue (path state is compacted every step).
//   k_prims      hitScene part 1 as a kernel of its own: only ptmi_trace needs it.
//   k_accumulate framebuffer read-modify-write of main.wgsl:22-27 for every frame slot, in frame order.
#pragma once
#include "ptmi_device.h"

namespace ptmi {

constexpr int kBlock = 256;

// k_bvh's claim counters: one per team of waves, 128 bytes apart.  Whoever fills a queue (k_generate, k_shade, k_prims)
// zeroes them for the k_bvh launch that follows.
constexpr uint32_t kHeadStride = 32, kMaxTeams = 64;
DEV void reset_heads(uint32_t* __restrict__ heads) {
  if (blockIdx.x == 0 && threadIdx.x < kMaxTeams) heads[threadIdx.x * kHeadStride] = 0u;
}

DEV void reduce_counters(const Counters& cn, unsigned long long* __restrict__ totals, bool bvh) {
  uint32_t v[5] = {cn.node_visits, cn.tri_tests, cn.sphere_tests, cn.quad_tests, cn.mat_fetches};
#pragma unroll
  for (int k = 0; k < 5; k++) {
    unsigned long long x = v[k];
    for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
    if (lane_id() == 0 && x) {
      atomicAdd(&totals[2 + k], x);
      if (bvh && k == 0) atomicAdd(&totals[7], x);  // node visits below the root
      if (bvh && k == 4) atomicAdd(&totals[8], x);  // accepted triangle hits
    }
  }
}

// step 0's queue: the randState of slot g (see Slots::q0)
DEV uint32_t* rng0_of(const Paths& P) { return reinterpret_cast<uint32_t*>(P.in.q0); }

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_generate(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads,
                                                     unsigned long long* __restrict__ totals) {
  reset_heads(heads);
  const bool trace = rc.max_bounces > 0;  // MAX_BOUNCES = 0: ray_color's loop body never runs, no hitScene at all
  uint32_t total = rc.n_local * (uint32_t)rc.n_frames;
  Counters cn = {0, 0, 0, 0, 0};
  // A thread takes a pixel for kGenFrames consecutive frames of the batch: the two integer divisions that turn a path number into (frame, pixel) and the
  // f32 division + remainder that turn the pixel into (x, y) — a seventh of the kernel's instructions — are made once per 16 paths (round 4).  Slot g of
  // step 0's queue is still path g = frame_slot * n_local + local pixel: a wave's stores are 64 neighbours of one frame, then of the next.
  constexpr uint32_t kGenFrames = 16;
  const uint32_t chunks = ((uint32_t)rc.n_frames + kGenFrames - 1u) / kGenFrames, work = rc.n_local * chunks;
  for (uint32_t w = blockIdx.x * kBlock + threadIdx.x; w < work; w += gridDim.x * kBlock) {
   const uint32_t chunk = w / rc.n_local, j = w - chunk * rc.n_local;
   const uint32_t pix = local_to_pixel(rc, j);
   float px, py;
   camera_pixel(rc, pix, px, py);
   const uint32_t f_end = min((chunk + 1u) * kGenFrames, (uint32_t)rc.n_frames);
#pragma unroll 1
   for (uint32_t f = chunk * kGenFrames; f < f_end; f++) {
    const uint32_t g = f * rc.n_local + j;
    uint32_t pid = g;  // path id = frame_slot * n_local + local pixel index (dense per rank)
    // u32(uniforms.frameNum): the frame number travels through an f32 uniform (renderer.js:173)
    uint32_t rng = pix + (uint32_t)(float)(rc.frame0 + f) * 719393u;
    f3 o, d;
    camera_ray_at(rc, px, py, 0, rng, o, d);
    float2 tp = make_float2(0.0f, 0.0f);
    uint32_t hm = HITMAT_MISS;
    if (trace) prims_for_ray<COUNT>(S, o, d, rng, tp, hm, cn);
    // 32 bytes per path.  What every new path starts with — origin = cam_origin, throughput (1,1,1), bounce 0, acc_radiance 0 — is not stored: the
    // readers of step 0's queue are told so (`first`), and acc[pid] counts as zero until P.touched[pid] is set (NUM_SAMPLES == 1).
    rng0_of(P)[g] = rng;
    P.in.q1[g] = make_float4(d.x, d.y, d.z, __uint_as_float(pid));
    P.hin.tp[g] = tp;
    P.hin.mat[g] = hm;
    if (!P.touched) {
      P.acc[pid] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
      P.pixsum[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
   }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl[0].n_rays = total;
  if (COUNT) reduce_counters(cn, totals, false);
}

// 96 VGPRs (5 waves/SIMD, no spills) measured 27 % faster than the compiler's default 106 VGPRs / 4 waves: the kernel
// is latency bound; 6 and 8 waves/SIMD spill and lose again.
#define PTMI_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))
#define PTMI_BVH_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))

// Rank of this lane's entry within its material bin for the current chunk (counting sort, pass 1).
// Must be called from wave-uniform control flow; lanes without an entry pass bin = -1.
DEV uint32_t bin_rank(int bin, uint32_t* s_cnt) {
  uint32_t rank = 0;
#pragma unroll
  for (int b = 0; b < NUM_BINS; b++) {
    const bool mine = (bin == b);
    const uint64_t mk = __ballot(mine);
    if (mk) {
      const int leader = __ffsll((unsigned long long)mk) - 1;
      uint32_t bb = 0;
      if (lane_id() == leader) bb = atomicAdd(&s_cnt[b], (uint32_t)__popcll(mk));
      bb = (uint32_t)__shfl((int)bb, leader, 64);
      if (mine) rank = bb + lanes_below(mk);
    }
  }
  return rank;
}

// hitScene, part 1 as a stand-alone, element-wise kernel (one slot per thread): ptmi_trace's entry into the pipeline.
// Renders never launch it — k_generate and k_shade run prims_for_ray on the rays they create.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_prims(DevScene S, Paths P, const StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads,
                                                  unsigned long long* __restrict__ totals) {
  reset_heads(heads);
  const uint32_t n = ctl->n_rays;
  Counters cn = {0, 0, 0, 0, 0};
  for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < n; slot += gridDim.x * kBlock) {
    const float4 a0 = P.in.q0[slot], a1 = P.in.q1[slot];
    if (__float_as_uint(a1.w) == PID_HOLE) {
      P.hin.mat[slot] = HITMAT_HOLE;
      continue;
    }
    uint32_t rng = __float_as_uint(a0.w);
    const uint32_t rng0 = rng;
    float2 tp;
    uint32_t hm;
    prims_for_ray<COUNT>(S, mk3(a0), mk3(a1), rng, tp, hm, cn);
    if (rng != rng0) P.in.q0[slot] = make_float4(a0.x, a0.y, a0.z, __uint_as_float(rng));
    P.hin.tp[slot] = tp;
    P.hin.mat[slot] = hm;
  }
  if (COUNT) reduce_counters(cn, totals, false);
}

// A wave refills its idle lanes once this many lanes are idle (or all are).
constexpr int kRefillThreshold = 32;  // (round 3: 16 -> 32, configs[3] -3 %: idle lanes cost nothing on the gather path, and rays picked up together share their first fetches)
// The triangle phase of the flat traversal runs once this many lanes hold a pending leaf (or nothing else can run).
constexpr int kLeafBatch = 16;
constexpr uint32_t kBvhRange = 512;  // slots a wave claims per global atomic (less when the queue is short); round 3: 256..1024 equal within noise, 2048 +1 %, 8192 +6 % (the last ranges are a tail)



// k_bvh, second edition (round 3).  Same scheduling (persistent single-wave blocks, team counters, flag scan, ballot refill),
// same per-ray visit order, outcomes and counters; three changes to where the instructions and the round trips go:
//   * ONE fetch per lane per iteration whatever the lane is about to do — the pair record of its inner node or the pretri record
//     of its pending triangle (both 4 x float4) — so a triangle phase is no longer a memory round trip of its own: a lane with a
//     pending leaf fetches its record with everybody else's next fetch and keeps it in the same registers until the triangle
//     batch is due (exec-masked loads leave the other lanes' registers alone);
//   * one shared pop section behind both kinds of step, on 8-byte LDS stack entries (ptmi_device.h).
// (Storing an accepted hit at once instead of at the end of the ray saved four registers and lost 12 %: on gfx9 stores count on
// vmcnt like loads, so every later fetch waited for them.)
#ifdef PTMI_LANE_TALLY
// k_bvh's stopwatch (measurement builds, tools/bvh_regions.py): a wave keeps cycles, marks and lanes per region in registers and adds them to
// g_bvh_tally when it ends.  BT(k, dep, lanes) closes the interval since the previous mark and charges it to region k.
enum : int { BT_SCAN = 0, BT_PICKUP = 1, BT_INNER_FETCH = 2, BT_INNER_STEP = 3, BT_LEAF_FETCH = 4, BT_LEAF_TEST = 5, BT_RETIRE = 6, BT_VOTE = 7, BT_CARRY = 8, BT_START = 9, kBvhTallies = 10 };
__device__ unsigned long long g_bvh_tally[3 * kBvhTallies];  // {cycles, marks, lanes} per region
enum : int { TB_TAKE = 0, TB_WALK_FETCH = 1, TB_WALK_STEP = 2, TB_WALK_LEAF = 3, TB_SHADE = 4, TB_PRIMS = 5, TB_VOTE = 6 };  // k_tail's regions (same stopwatch)
__device__ unsigned long long g_tail_tally[3 * kBvhTallies];
#define BT(k, dep, lanes)                                                                                  \
  do {                                                                                                     \
    unsigned long long now_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) : "v"(dep) : "memory");               \
    bt_cyc[k] += now_ - bt_last, bt_marks[k] += 1u, bt_lanes[k] += (uint32_t)(lanes);                      \
    bt_last = now_;                                                                                        \
  } while (0)
#else
#define BT(k, dep, lanes) ((void)0)
#endif
constexpr int kScanGroups = 3, kCandSlots = 64 * (kScanGroups + 1);  // a pass adds at most 64 x kScanGroups candidates to fewer than 64
template <bool COUNT, bool NOABORT>
DEV void bvh2_body(const DevScene& S, const Paths& P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads, uint32_t n_teams, int stack_size, int lds_entries, int spill_entries,
                   int2* __restrict__ spill, int refill_threshold, int leaf_batch, unsigned long long* __restrict__ totals, uint32_t range_cap, float4 cam, const Carry& cy,
                   int* lds_stack, uint32_t wave_id, uint32_t n_waves  // (stand where blockIdx.x / gridDim.x would: a wave works on its own)
) {
  const int lane = lane_id();
  LaneStack2 stk;
  stk.lds = (lds_v2i_t*)lds_stack + lane;
  stk.spill = spill + (size_t)wave_id * (size_t)spill_entries * 64 + lane;
  stk.lds_entries = lds_entries;
  uint32_t* cand = reinterpret_cast<uint32_t*>(lds_stack + lds_entries * 2 * 64);  // [kCandSlots] candidate slots
  const uint32_t n_carried = cy.resv ? min(ctl->n_carried, cy.resv) : 0u;  // slots [0, n_carried) hold rays the previous launch carried over, [n_carried, resv) nothing
  uint32_t n = ctl->n_rays;
  if (n <= cy.resv && n_carried == 0u) n = 0u;  // nothing but the empty carry prefix
  const uint32_t range = min(range_cap, max(64u, ((n / (2u * n_waves)) + 63u) & ~63u));
  const uint32_t team = wave_id % n_teams;
  Counters cn = {0, 0, 0, 0, 0};
  uint32_t rb = 0, re = 0;   // this wave's claimed range of slots still to be scanned (wave-uniform)
  uint32_t ncand = 0;        // candidates waiting in `cand` (wave-uniform)
  bool exhausted = (n == 0);
  int tail_iters = 0;        // iterations since this wave found the queue exhausted (wave-uniform)
  bool may_carry = cy.resv_next != 0u;
  // the lane's ray
  uint32_t node = N_DONE, myslot = 0, negmask = 0;
  int sp = 0;
  f3 o = mk3(0, 0, 0), d = o, inv = o;
  float ct = 0.0f;
  ObjRay orr;
  orr.mesh = -1;
  orr.o = orr.d = o;
  TriHit hit = {0.0f, 0.0f, 0u, 0u};
  bool has = false;  // this lane holds a ray (traversing, or finished and not yet retired)
  const uint32_t root = __float_as_uint(S.root_lo.w);
  const uint32_t root_node = (root & REF_LEAF) ? root : (root & REF_IDX);
#ifdef PTMI_LANE_TALLY
  unsigned long long bt_cyc[kBvhTallies] = {0}, bt_last;
  uint32_t bt_marks[kBvhTallies] = {0}, bt_lanes[kBvhTallies] = {0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt_last)::"memory");
#endif
  for (;;) {
    // retire finished rays (stores only): a triangle beat what part 1 had found; otherwise the record stands as it is
    BT(BT_VOTE, 0.0f, 0);
    if (has && node == N_DONE) {
      if (hit.prim != 0u) {
        P.hin.tp[myslot] = make_float2(ct, __uint_as_float(hit.prim));
        P.uv[myslot] = make_float2(hit.u, hit.v);
        P.hin.mat[myslot] = hit.mat;
      }
      has = false;
    }
    BT(BT_RETIRE, 0.0f, 0);
    uint64_t hm = __ballot(node != N_DONE);
    int nact = __popcll(hm);
    if ((64 - nact) >= (nact == 0 ? 1 : refill_threshold)) {
      const uint32_t want = (uint32_t)(64 - nact);
      while (ncand < want && !(exhausted && rb == re)) {
        if (rb == re) {  // claim the next range of slots (see k_bvh)
          // (Taking most ranges in a fixed order instead — wave w the ranges w, w + n_waves, ... — and claiming only the last quarter: configs[2] / [3]
          // +40 % k_bvh time.  Waves that claim one after the other work on neighbouring stretches of the queue; that is worth more than the claims cost.)
          uint32_t i = 0;
          if (lane == 0) i = atomicAdd(&heads[team * kHeadStride], 1u);
          i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
          const uint64_t nb64 = ((uint64_t)team + (uint64_t)n_teams * i) * range;
          if (nb64 >= (uint64_t)n) {
            exhausted = true;
            continue;
          }
          const uint32_t nb = (uint32_t)nb64;
          rb = nb;
          re = min(nb + range, n);
        }
        // kScanGroups x 64 flag words per pass, the loads in flight together: where few slots are flagged (configs[1]: one in twelve — the mesh is
        // small in the room) a refill is a chain of such round trips, and with the tree in LDS they were a quarter of the wave's time
        uint32_t word[kScanGroups];
#pragma unroll
        for (int g = 0; g < kScanGroups; g++) {
          const uint32_t slot = rb + 64u * (uint32_t)g + (uint32_t)lane;
          word[g] = P.hin.mat[min(slot, re - 1u)];  // (no branch around the load: under one, each of the three got an s_waitcnt of its own and they went one after the other; rb < re here)
        }
#pragma unroll
        for (int g = 0; g < kScanGroups; g++) {
          const uint32_t slot = rb + 64u * (uint32_t)g + (uint32_t)lane;
          const bool flagged = slot < re && (word[g] & HITMAT_BVH) != 0u && !dead_slot(slot, n_carried, cy.resv);
          const uint64_t fm = __ballot(flagged);
          if (flagged) cand[ncand + lanes_below(fm)] = slot;
          ncand += (uint32_t)__popcll(fm);
        }
        rb = min(rb + 64u * (uint32_t)kScanGroups, re);
        BT(BT_SCAN, 0.0f, ncand);
      }
      if (ncand) {
        const uint64_t idle = ~hm;
        const uint32_t k = lanes_below(idle);
        const uint32_t take = min(ncand, want);
        if (node == N_DONE && k < take) {
          myslot = cand[ncand - 1u - k];
          const float4 r1 = P.in.q1[myslot];
          ct = P.hin.tp[myslot].x;  // closest_so_far after part 1 of hitScene; the rest of that record stands unless a triangle wins
          o = mk3(cam);
          if (cam.w == 0.0f) o = mk3(P.in.q0[myslot]);  // (wave-uniform: step 0 picks a ray up with two gathers instead of three)
          d = mk3(r1);
          inv = rcp3_exact_il(d);
          negmask = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
          orr.mesh = -1;
          if (S.uniform_gid >= 0) obj_ray_uniform(S, o, d, orr);
          sp = 0;
          hit.prim = 0u;
          has = true;
          node = root_node;
          if (myslot < cy.resv) {  // a ray the previous launch carried over: its traversal goes on where it stopped
            const uint32_t* rec = cy.pool_in + (size_t)myslot * (size_t)cy.rec_words;
            node = rec[0];
            sp = (int)rec[1];
            ct = __uint_as_float(rec[2]);
            hit.u = __uint_as_float(rec[3]), hit.v = __uint_as_float(rec[4]), hit.prim = rec[5], hit.mat = rec[6];
            for (int e = 0; e < sp; e++) stack2_write(stk, e, rec[8 + 2 * e], __uint_as_float(rec[9 + 2 * e]));
          }
        }
        ncand -= take;
        hm = __ballot(node != N_DONE);
        nact = __popcll(hm);
        BT(BT_PICKUP, inv.x + ct, take);
      }
    }
    if (nact == 0) break;  // nothing in flight, nothing buffered, queue exhausted
    const bool more = ncand > 0 || !(exhausted && rb == re);
    const int min_working = more ? (64 - refill_threshold + 1) : 1;
    int working;
    {
    // Two phases per iteration, each with its own fetch: the triangle records are the coldest data of the
    // scene, and a wave waits for the slowest lane of a fetch — mixing them into every pair fetch (UNIFIED) made every wait a slow one
    // (configs[3]: 581 ms against 514 per 128 spp although it issues 11 % fewer vector and 46 % fewer memory instructions).
    do {
      const uint64_t pm = __ballot((int)node < 0), im = __ballot(node < N_INNER_LIMIT);
      BT(BT_VOTE, 0.0f, 0);
      if (pm != 0ull && (__popcll(pm) >= leaf_batch || im == 0ull)) {
        if ((int)node < 0) {
          const int2 lc = (node & REF_MULTI) ? S.leaf_table[node & REF_IDX] : make_int2((int)(node & REF_IDX), 1);
          for (int j = 0; j < lc.y; j++) {  // one triangle per leaf with the reference's builder
            const float4* rec = S.pretri + 4 * (size_t)(lc.x + j);
            const float4 g0 = rec[0], g1 = rec[1], g2 = rec[2], g3 = rec[3];
#ifdef PTMI_LANE_TALLY
            if (j == 0) BT(BT_LEAF_FETCH, g0.x + g1.x + g2.x + g3.x, __popcll(__ballot(1)));
#endif
            tri_test2<COUNT>(S, lc.x + j, g0, g1, g2, g3, o, d, orr, ct, hit, cn);
          }
          node = pop_until_pass2(stk, sp, ct, cn, COUNT);
        }
        BT(BT_LEAF_TEST, ct, __popcll(pm));
      }
      if (node < N_INNER_LIMIT) {
        const float4* rec = S.pairs + 4 * (size_t)node;
        const float4 f0 = rec[0], f1 = rec[1], f2 = rec[2], f3v = rec[3];
        BT(BT_INNER_FETCH, f0.x + f1.x + f2.x + f3v.x, __popcll(__ballot(1)));
        node = inner_step2<COUNT, NOABORT>(f0, f1, f2, f3v, o, inv, S.tmin, negmask, ct, stack_size, stk, sp, cn);
        if (node == N_POP) node = pop_until_pass2(stk, sp, ct, cn, COUNT);
      }
      BT(BT_INNER_STEP, __uint_as_float(node), __popcll(im));
      working = __popcll(__ballot(node != N_DONE));
      if (!more && may_carry && ++tail_iters >= cy.after) break;  // (wave-uniform)
    } while (working >= min_working);
    }
    if (!more && may_carry && tail_iters >= cy.after) {
      // The queue is exhausted and this wave has gone on for `after` iterations: what it still holds are the launch's long rays.  Each one moves
      // into a slot of the next step's queue with its traversal state in the pool; its slot here becomes a hole, so this step's k_shade passes it by.
      bool failed = false;
      if (node != N_DONE) {
        const uint32_t ns = atomicAdd(&ctl[1].n_carried, 1u);
        if (ns < cy.resv_next) {
          const float4 r1 = P.in.q1[myslot];
          float4 r0 = cam, r2 = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0));  // (step 0's queue stores neither the origin nor the initial throughput / bounce)
          if (cam.w == 0.0f) r0 = P.in.q0[myslot], r2 = P.in.q2[myslot];
          else r0.w = __uint_as_float(rng0_of(P)[myslot]);
          P.out.q0[ns] = r0, P.out.q1[ns] = r1, P.out.q2[ns] = r2;
          P.hout.tp[ns] = P.hin.tp[myslot];    // hitScene part 1's record: stands unless a triangle has won or wins later
          P.hout.mat[ns] = P.hin.mat[myslot];  // (HITMAT_BVH still set)
          uint32_t* rec = cy.pool_out + (size_t)ns * (size_t)cy.rec_words;
          rec[0] = node, rec[1] = (uint32_t)sp, rec[2] = __float_as_uint(ct), rec[3] = __float_as_uint(hit.u), rec[4] = __float_as_uint(hit.v), rec[5] = hit.prim, rec[6] = hit.mat;
          for (int e = 0; e < sp; e++) {
            uint32_t w0;
            float w1;
            stack2_read(stk, e, w0, w1);
            rec[8 + 2 * e] = w0, rec[9 + 2 * e] = __float_as_uint(w1);
          }
          reinterpret_cast<uint32_t*>(P.in.q1 + myslot)[3] = PID_HOLE;
          P.hin.mat[myslot] = HITMAT_HOLE;
          node = N_DONE;
          has = false;
        } else {  // the pool is full: this ray is traversed to its end here after all
          atomicSub(&ctl[1].n_carried, 1u);
          failed = true;
        }
      }
      if (__ballot(failed) != 0ull) may_carry = false;
      BT(BT_CARRY, 0.0f, 0);
    }
  }
#ifdef PTMI_LANE_TALLY
  if (lane == 0)
    for (int k = 0; k < kBvhTallies; k++) {
      if (bt_marks[k] == 0u) continue;
      atomicAdd(&g_bvh_tally[3 * k], bt_cyc[k]);
      atomicAdd(&g_bvh_tally[3 * k + 1], (unsigned long long)bt_marks[k]);
      atomicAdd(&g_bvh_tally[3 * k + 2], (unsigned long long)bt_lanes[k]);
    }
#endif
  if (COUNT) reduce_counters(cn, totals, true);
}

template <bool COUNT, bool NOABORT>
__global__ __launch_bounds__(64) PTMI_BVH_ATTR void k_bvh2(DevScene S, Paths P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads, uint32_t n_teams, int stack_size,
                                                           int lds_entries, int spill_entries, int2* __restrict__ spill, int refill_threshold, int leaf_batch,
                                                           unsigned long long* __restrict__ totals, uint32_t range_cap, float4 cam,  // cam.w != 0: step 0's queue — every ray starts at cam.xyz
                                                           Carry cy
) {
  extern __shared__ int lds_stack[];
  bvh2_body<COUNT, NOABORT>(S, P, ctl, heads, n_teams, stack_size, lds_entries, spill_entries, spill, refill_threshold, leaf_batch, totals, range_cap, cam, cy, lds_stack,
                                     blockIdx.x, gridDim.x
  );
}


// What a surviving path carries into the next step's queue.
struct NewState {
  f3 o, d, T;
  int bounce;
  uint32_t rng, pid;
};

// One path's iteration of the `for i < MAX_BOUNCES` loop body after hitScene (traceRay.wgsl:10-80).
// Returns true when the path continues (its state for the next step is in `ns`), false when the slot is done.
// `acc` (by path id) is touched only when it changes or is needed: `acc + emission*T` with emission*T == +-0
// is `acc` bit for bit (acc is never -0: it starts at +0 and x + y = -0 only for x = y = -0), and with
// NUM_SAMPLES == 1 the pixel colour (0 + acc) / 1 is `acc` itself.
// A slot's state and hit record, fetched in one go (seven independent loads in flight).
struct SlotState {
  float4 q0, q1, q2;
  float2 tp;
  uint32_t hitmat, slot;
};
DEV SlotState load_slot(const Paths& P, uint32_t slot, bool first, const RenderConst& rc) {
  SlotState st;
  st.slot = slot;
  st.hitmat = P.hin.mat[slot];
  st.q1 = P.in.q1[slot];
  if (first) {  // step 0: k_generate does not store what every path starts with
    const f3 co = cam_origin(rc);
    st.q0 = make_float4(co.x, co.y, co.z, __uint_as_float(rng0_of(P)[slot]));
    st.q2 = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0));
  } else {
    st.q0 = P.in.q0[slot];
    st.q2 = P.in.q2[slot];
  }
  st.tp = P.hin.tp[slot];
  return st;
}
// acc_radiance of a path as ray_color sees it; never-written entries stand for zero (P.touched, NUM_SAMPLES == 1).  Whether a path has
// written its entry before travels WITH the path — bit kAccWritten of the bounce word in q2.w — so that k_shade never reads the flag
// array: round 3 looked `touched[pid]` up and then `acc[pid]`, two dependent round trips to HBM in the middle of a group for every path
// that ends or meets a light (one lane in four per step); now a path's first write is a blind store and only a path that has added
// emission before (rare) loads its entry.  The flag array is still written (first store) — k_accumulate reads it.
constexpr int kAccWritten = 0x40000000, kBounceMask = 0x3fffffff;
DEV float4 acc_load(const Paths& P, uint32_t pid, bool written) {
  if (P.touched && !written) return make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
  return P.acc[pid];
}
DEV void acc_store(const Paths& P, uint32_t pid, float4 v, bool written) {
  P.acc[pid] = v;
  if (P.touched && !written) P.touched[pid] = 1;
}

// The end of a path's only sample in progressive mode (NUM_SAMPLES == 1): acc_radiance += add; pixColor = (0 + acc_radiance) / 1.
// `acc` is touched only when it changes: acc + (+-0) is acc bit for bit, and (0 + acc) / 1 is acc.
DEV void end_sample_progressive(const Paths& P, uint32_t pid, f3 add, bool written) {
  const bool changes = !(add.x == 0.0f && add.y == 0.0f && add.z == 0.0f);  // NaN counts as a change
  if (changes) {
    LT(LT_END_CHANGES);
    float4 A4 = acc_load(P, pid, written);
    f3 fin = mk3(0, 0, 0) + (mk3(A4) + add);  // "/ NUM_SAMPLES" with NUM_SAMPLES == 1: x / 1.0f is x
    acc_store(P, pid, make_float4(fin.x, fin.y, fin.z, __int_as_float(1)), written);
  }  // else: (0 + acc) / 1 == acc, already in place — or never written, which k_accumulate reads as zero
}

// MULTI = NUM_SAMPLES > 1 (the per-pixel sample loop of shootRay.wgsl:5-49 lives in the slot: pixsum, in-slot camera ray); the
// reference's progressive mode (NUM_SAMPLES = 1) compiles without it, which also frees the scalar registers the view matrix and
// the image constants would occupy through the whole kernel.
template <bool IS, bool MULTI>
DEV bool shade_one(const DevScene& S, const RenderConst& rc, const Paths& P, const SlotState& st, const TriFetch& tf, const QuadL& L, NewState& ns) {
  const uint32_t pid = __float_as_uint(st.q1.w);
  const f3 o = mk3(st.q0), d = mk3(st.q1);
  const float4 T4 = st.q2;
  f3 T = mk3(T4);
  int bounce = __float_as_int(T4.w) & kBounceMask;
  bool acc_written = (__float_as_int(T4.w) & kAccWritten) != 0;  // (progressive mode) this path has stored its acc_radiance before
  uint32_t rng = __float_as_uint(st.q0.w);
  const uint32_t prim = __float_as_uint(st.tp.y);

  bool sample_done = false;
  bool drop_acc = false;     // the sample's radiance is `add` alone (importance-sampling early return, Q8)
  f3 add = mk3(0, 0, 0);     // what this iteration adds to acc_radiance
  f3 no = o, nd = d;

  if ((prim >> 28) == K_NONE) {  // traceRay.wgsl:12-16
    LT(LT_MISS);
    add = mk3(rc.bg[0], rc.bg[1], rc.bg[2]) * T;
    sample_done = true;
  } else {
    LT(LT_HIT);
    int mat = (int)(st.hitmat & HITMAT_ID);
    Material m = load_material(S, mat);
    // The material's class: from the material word of the hit record (prepare_scene derives it from material_type exactly as scatterRay.wgsl's branches
    // read it), not from the material's record — so that the scatter can start when the hit's normal has arrived and the record's albedo / emission are
    // awaited where they are used, after it (round 4).  The chunk is sorted by this class, so `bin` is wave-uniform almost everywhere.
    const int bin = (int)((st.hitmat >> HITMAT_BIN_SHIFT) & 7u);
    HitGeom g = resolve_hit(S, o, d, st.tp.x, tf, prim);
    TT(TT_LOAD2, g.n.x + m.type + m.color.x);  // the material's record and the hit's normal data have arrived
    f3 emission = m.emission;
    if (!g.front) emission = mk3(0, 0, 0);  // traceRay.wgsl:19-22
    float doSpecular;
    bool skip_pdf;
    f3 unit_w = mk3(0, 0, 0);
    f3 sdir = material_scatter(bin, m, g, d, rng, doSpecular, skip_pdf, unit_w);
    f3 sorg = (bin == BIN_OTHER) ? mk3(0, 0, 0) : g.p;
    bool roulette = true;
    add = emission * T;  // acc_radiance += emissionColor * throughput (traceRay.wgsl:26,55,64)
    if (IS) {  // traceRay.wgsl:24-58
      if (skip_pdf) {
        T = T * mix3(m.color, m.spec, doSpecular);
        no = sorg;
        nd = sdir;
        roulette = false;  // `continue` skips the Russian roulette
      } else {
        // get_random_on_quad(lights, hitRec.p) (importanceSampling.wgsl:78-81): u draw, then v draw
        LT(LT_IS_LIGHT);
        float ru = rand2D(rng);
        f3 pu = ru * L.u;
        float rv = rand2D(rng);
        f3 pv = rv * L.v;
        f3 lp = L.Q + pu + pv;
        f3 ldir = norm3(lp - g.p);
        f3 so = g.p, sd = ldir;
        float rnd = rand2D(rng);
        if (rnd > rc.light_mix) {
          so = sorg;
          sd = sdir;
        }
        float cosine_theta = dot3(norm3(sd), unit_w);  // onb_lambertian_scattering_pdf :73-76
        float lambertian_pdf = ptm_max(0.0f, cosine_theta / kPi);
        float lpdf = light_pdf(L, so, sd);
        float pdf = rc.light_mix * lpdf + rc.surface_mix * lambertian_pdf;  // 0.2 and 0.8 (traceRay.wgsl:43,49) by default
        if (pdf <= 0.00001f) {  // returns emission*throughput, dropping acc (Q8)
          drop_acc = true;
          sample_done = true;
        } else {
          T = T * ((lambertian_pdf * mix3(m.color, m.spec, doSpecular)) / pdf);
          no = so;
          nd = sd;
        }
      }
    } else {  // traceRay.wgsl:61-68
      T = T * mix3(m.color, m.spec, doSpecular);
      no = sorg;
      nd = sdir;
    }
    if (!sample_done) {
      if (roulette && bounce > 2) {  // traceRay.wgsl:71-79
        LT(LT_RR);
        float p = ptm_max(T.x, ptm_max(T.y, T.z));
        if (rand2D(rng) > p) {
          sample_done = true;
        } else {
          T = T * rcp_exact(p);
        }
      }
      if (!sample_done) {
        bounce++;
        if (bounce >= rc.max_bounces) sample_done = true;  // loop exhausted: returns acc (Q8)
      }
    }
  }

  const bool changes = !(add.x == 0.0f && add.y == 0.0f && add.z == 0.0f);  // NaN counts as a change
  ns.pid = pid;
  if (!sample_done) {
    if (changes) {
      LT(LT_ACC_CONT);
      float4 A4 = acc_load(P, pid, acc_written);
      f3 acc = mk3(A4) + add;
      acc_store(P, pid, make_float4(acc.x, acc.y, acc.z, A4.w), acc_written);
      acc_written = true;
    }
    ns.o = no, ns.d = nd, ns.T = T, ns.bounce = bounce | (acc_written ? kAccWritten : 0), ns.rng = rng;
    return true;
  }

  // pathTrace (shootRay.wgsl:5-49): pixColor += ray_color(ray); pixColor /= NUM_SAMPLES
  LT(LT_END_SAMPLE);
  if (!MULTI) {
    if (drop_acc) {
      f3 fin = mk3(0, 0, 0) + add;  // (NUM_SAMPLES == 1: no division)
      acc_store(P, pid, make_float4(fin.x, fin.y, fin.z, __int_as_float(1)), acc_written);
    } else {
      end_sample_progressive(P, pid, add, acc_written);
    }
    return false;
  } else {
  float4 A4 = P.acc[pid];
  int sample = __float_as_int(A4.w);
  f3 radiance = drop_acc ? add : (mk3(A4) + add);
  f3 sum = mk3(P.pixsum[pid]) + radiance;
  sample++;
  if (sample < rc.num_samples) {  // the next sample continues the same RNG stream in the same slot
    P.pixsum[pid] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    camera_ray(rc, local_to_pixel(rc, pid % rc.n_local), sample, rng, no, nd);
    P.acc[pid] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(sample));
    ns.o = no, ns.d = nd, ns.T = mk3(1.0f, 1.0f, 1.0f), ns.bounce = 0, ns.rng = rng;
    return true;
  }
  f3 fin = sum / rc.sample_div;
  P.acc[pid] = make_float4(fin.x, fin.y, fin.z, __int_as_float(sample));
  return false;
  }
}

// (Measured and settled, so no longer build options: the sorted variants wave by wave with per-class bins instead of the block version's LDS counting sort
// — bit-exact, 2.5 % slower on configs[4], profiles/r04_sort_wave_ab.txt —; every variant through the block version; other region / chunk sizes.)
constexpr uint32_t kRegionDiv = 16;  // a block's share of the queue is claimed in this many regions (x4: every wave claims its own)
constexpr int kTailLimitFirst = 2 << 20, kTailLimitLater = 1 << 20;  // k_tail takes a queue over when it is at most this long (slots): at step 0 (a lone 1080p frame fits) / later (round 4: 512 Ki -> 1 Mi since its tree walk waits for 24 lanes — 8-frame batches of configs[1] +5 %, 64-frame ones and configs[2] unchanged: profiles/r04_tail_limit_ab.txt)
constexpr int kTailLimitFirstShallow = 24 << 20, kTailLimitLaterShallow = 2 << 20;  // ... on trees under 12 levels (render_batch)
constexpr int kTailLimitFirstDeep = 6 << 20;  // ... on deeper ones, whose walks park their stragglers (render_batch)
constexpr int kTailRefill = 16;  // k_tail: idle lanes before a wave takes new paths
constexpr bool kMissShortcut = true;  // definite misses are settled in k_shade's flush phase
constexpr int kSChunk = 512;  // slots a k_shade block sorts, shades and compacts at a time

// ray_color's loop body for one step, 512 slots at a time per block:
//   1  (SORT) LDS counting sort of the chunk by the shade bin in each slot's material word (ballot ranks), so that
//      the waves are (almost) uniform in the 4-way material switch of scatterRay.wgsl; holes drop out here.  Scenes
//      whose materials all fall into one bin skip the sort (SORT = false): only misses would diverge, and cheaply;
//   2  shade_one per slot (state streamed in by slot); survivors' next state is staged in LDS, densely;
//   3  every staged survivor — dense again, all lanes busy — gets hitScene part 1 for its new ray, and state plus hit
//      record go to the block's current OUTPUT REGION of the next queue, coalesced.  A block claims a region with one
//      global atomic (16 or so per launch), fills it across chunks — an entry that does not fit any more continues in
//      the next region — and marks what is left at the end as holes.
template <bool IS, bool COUNT, bool MULTI>
DEV void shade_body(const DevScene& S, const RenderConst& rc, const Paths& P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads, unsigned long long* __restrict__ totals,
                    int first, uint32_t resv) {
  reset_heads(heads);
  __shared__ float4 s_q0[kSChunk], s_q1[kSChunk], s_q2[kSChunk];
  __shared__ uint16_t s_sorted[kSChunk];
  __shared__ uint32_t s_cnt[NUM_BINS + 2];
  __shared__ uint32_t s_nout, s_next;
  const QuadL L = load_light(S);
  const int lane = lane_id();
  const uint32_t n_carried = resv ? min(ctl->n_carried, resv) : 0u;  // (Carry: slots [n_carried, resv) of this queue hold nothing)
  uint32_t n = ctl->n_rays;
  if (n <= resv && n_carried == 0u) n = 0u;
  // region size: a block handles about n / gridDim slots per launch; 1/16 of that per claim keeps both the
  // number of atomics and the holes left at the end (at most one region per block) small
  const uint32_t region = max((uint32_t)kSChunk, ((n / gridDim.x / kRegionDiv) + 511u) & ~511u);
  const uint32_t wregion = region / (kBlock / 64);  // every wave fills output regions of its own (>= 128 slots): no barrier, no serial section in the flush phase
  uint32_t w_cur = 0, w_rend = 0;                   // this wave's current output region [w_cur, w_rend) of the next queue (wave-uniform)
  // The waves' FIRST regions come from one claim per block, made by whichever wave needs a region first; the others pick their quarter
  // up from LDS (a launch made at least one claim per wave before: 6144 atomics on one address ~ 70 us — of a tail step with 100 us of
  // work).  A block without survivors claims nothing, so a queue of holes still dies out.
  constexpr uint32_t kR0Empty = 0xffffffffu, kR0Busy = 0xfffffffeu, kR0Full = 0xfffffffdu;
  __shared__ uint32_t s_region0;
  if (threadIdx.x == 0) s_region0 = kR0Empty;  // (visible after the first barrier of the chunk loop)
  uint32_t my_valid = 0;  // lane 0 of a wave: slots holding a path seen so far (= hitScene invocations)
  uint32_t my_missed = 0;  // lane 0 of a wave: new rays that turned out to be misses in the flush phase (progressive mode)
  Counters cn = {0, 0, 0, 0, 0};
  // stage one survivor per lane in LDS, densely (one LDS atomic per wave)
  auto stage = [&](bool survive, const NewState& ns) {
    const uint64_t mk = __ballot(survive);
    if (mk) {
      const int leader = __ffsll((unsigned long long)mk) - 1;
      uint32_t bb = 0;
      if (lane == leader) bb = atomicAdd(&s_nout, (uint32_t)__popcll(mk));
      bb = (uint32_t)__shfl((int)bb, leader, 64);
      if (survive) {
        const uint32_t q = bb + lanes_below(mk);
        s_q0[q] = make_float4(ns.o.x, ns.o.y, ns.o.z, __uint_as_float(ns.rng));
        s_q1[q] = make_float4(ns.d.x, ns.d.y, ns.d.z, __uint_as_float(ns.pid));
        s_q2[q] = make_float4(ns.T.x, ns.T.y, ns.T.z, __int_as_float(ns.bounce));
      }
    }
  };
  for (uint32_t base = blockIdx.x * (uint32_t)kSChunk; base < n; base += gridDim.x * (uint32_t)kSChunk) {
    const uint32_t m = min((uint32_t)kSChunk, n - base);
    if (threadIdx.x < NUM_BINS) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
      s_nout = 0;
      s_next = 0;
    }
    __syncthreads();
    {
      // ---- 1: sort ----
      uint32_t keys[kSChunk / kBlock];
      uint32_t words[kSChunk / kBlock];  // the thread's material words, asked for together (with a branch around each load they went one after the other: round 4's reading of the ISA)
  #pragma unroll
      for (int r = 0; r < kSChunk / kBlock; r++) words[r] = P.hin.mat[base + min((uint32_t)r * kBlock + threadIdx.x, m - 1u)];
  #pragma unroll
      for (int r = 0; r < kSChunk / kBlock; r++) {
        const uint32_t j = (uint32_t)r * kBlock + threadIdx.x;
        int bin = -1;
        if (j < m && !dead_slot(base + j, n_carried, resv)) {
          const uint32_t b = (words[r] >> HITMAT_BIN_SHIFT) & 7u;
          if (b < (uint32_t)NUM_BINS) bin = (int)b;  // 7 = hole
        }
        uint32_t rank = 0;
        if ((uint32_t)r * kBlock < m) rank = bin_rank(bin, s_cnt);  // block-uniform condition
        keys[r] = (bin < 0) ? 0xffffffffu : ((uint32_t)bin | (rank << 3));
      }
      __syncthreads();
      uint32_t nvalid = 0;
      {
        uint32_t off[NUM_BINS];
  #pragma unroll
        for (int b = 0; b < NUM_BINS; b++) {
          off[b] = nvalid;
          nvalid += s_cnt[b];
        }
  #pragma unroll
        for (int r = 0; r < kSChunk / kBlock; r++) {
          if (keys[r] != 0xffffffffu) {
            const uint32_t b = keys[r] & 7u;
            uint32_t o = off[0];
  #pragma unroll
            for (int k = 1; k < NUM_BINS; k++) o = (b == (uint32_t)k) ? off[k] : o;
            s_sorted[o + (keys[r] >> 3)] = (uint16_t)((uint32_t)r * kBlock + threadIdx.x);
          }
        }
      }
      if (threadIdx.x == 0) my_valid += nvalid;
      __syncthreads();
      // ---- 2: shade, stage survivors ----
      // waves take 64-entry groups dynamically: the sort puts the cheap MISS entries last, static rounds would leave
      // the waves that got them idle at the barrier
#pragma unroll 1
      for (;;) {
        uint32_t k0 = 0;
        if (lane == 0) k0 = atomicAdd(&s_next, 64u);
        k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
        if (k0 >= nvalid) break;
        const uint32_t k = k0 + lane;
        bool survive = false;
        NewState ns;
        ns.o = ns.d = ns.T = mk3(0, 0, 0);
        ns.bounce = 0, ns.rng = 0, ns.pid = 0;
        if (k < nvalid) {
          const SlotState st = load_slot(P, base + s_sorted[k], first != 0, rc);
          const TriFetch tf = tri_fetch(S, P.uv, st.slot, __float_as_uint(st.tp.y));  // (issued ahead of the material's loads, consumed after them)
          survive = shade_one<IS, MULTI>(S, rc, P, st, tf, L, ns);
        }
        stage(survive, ns);
      }
    }
    __syncthreads();
    // ---- 3: hitScene part 1 for the survivors' new rays, then place them in the next queue ----
    // A new ray that hits no sphere and no quad and misses the root box is a MISS already (hitScene returns false): in progressive
    // mode its path ends here — background * throughput goes to acc_radiance now, exactly what the next step would have added as this
    // path's last addition — instead of travelling through the queue to occupy a lane of a shading wave that has nothing to do for it
    // (configs[1]: 4 of 10 rays leave the open box that way).  Its hitScene invocation is tallied for the next step, where it belongs.
    const uint32_t cnt = s_nout;
    uint32_t missed = 0;
#pragma unroll 1
    for (uint32_t q0 = (threadIdx.x & ~63u); q0 < cnt; q0 += kBlock) {  // one survivor per lane and pass; no barrier in here: every wave fills regions of its own
      const uint32_t q = q0 + (uint32_t)lane;
      bool keep = false;
      float2 tp = make_float2(0.0f, 0.0f);
      uint32_t hm = 0u, rng = 0u;
      if (q < cnt) {
        const float4 a0 = s_q0[q], a1 = s_q1[q];
        rng = __float_as_uint(a0.w);
        prims_for_ray<COUNT>(S, mk3(a0), mk3(a1), rng, tp, hm, cn);  // (hit_volume draws from the path's stream: rng goes back into the state)
        if (kMissShortcut && !MULTI && hm == HITMAT_MISS) {  // traceRay.wgsl:12-16
          const float4 a2 = s_q2[q];
          end_sample_progressive(P, __float_as_uint(a1.w), mk3(rc.bg[0], rc.bg[1], rc.bg[2]) * mk3(a2), (__float_as_int(a2.w) & kAccWritten) != 0);
          missed++;
        } else {
          keep = true;
        }
      }
      const uint64_t km = __ballot(keep);
      const uint32_t kept = (uint32_t)__popcll(km);
      if (kept == 0) continue;
      const uint32_t rank = lanes_below(km);
      const uint32_t b0 = w_cur, n0 = min(kept, w_rend - w_cur);  // what still fits this wave's current region (wave-uniform)
      uint32_t b1 = 0xffffffffu;
      w_cur += n0;
      if (kept > n0) {  // claim the wave's next region for the rest
        uint32_t nb = 0;
        bool full;
        if (w_rend == 0u) {  // the wave's first region: its quarter of the block's claim
          if (lane == 0) nb = atomicCAS(&s_region0, kR0Empty, kR0Busy);
          nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          if (nb == kR0Empty) {  // ours to make
            if (lane == 0) {
              nb = atomicAdd(&ctl[1].n_rays, region);
              if (nb + region > P.cap) {  // cannot happen with the host's sizing, see below
                atomicAdd(&totals[15], 1ull);
                atomicSub(&ctl[1].n_rays, region);
                nb = kR0Full;
              }
              atomicExch(&s_region0, nb);
            }
            nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          } else {
            while (nb == kR0Busy) {  // another wave of the block is at it: a global atomic's round trip
              __builtin_amdgcn_s_sleep(8);
              if (lane == 0) nb = atomicAdd(&s_region0, 0u);
              nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
            }
          }
          full = nb == kR0Full;
          nb += (threadIdx.x >> 6) * wregion;
        } else {
          if (lane == 0) nb = atomicAdd(&ctl[1].n_rays, wregion);
          nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          full = nb + wregion > P.cap;
          if (full && lane == 0) {
            atomicAdd(&totals[15], 1ull);
            atomicSub(&ctl[1].n_rays, wregion);
          }
        }
        if (full) {  // cannot happen with the host's sizing; never write out of bounds
          // The survivors that still fit the current region are written as usual, the rest is dropped and the claim is
          // handed back (every later claim overflows too and does the same, so the queue length ends up within the buffer);
          // the host reports the flag as an error on the next synchronising call.
        } else {
          b1 = nb;
          w_cur = nb + (kept - n0);
          w_rend = nb + wregion;
        }
      }
      if (keep && (rank < n0 || b1 != 0xffffffffu)) {
        const uint32_t dst = (rank < n0) ? (b0 + rank) : (b1 + (rank - n0));
        float4 a0 = s_q0[q];
        a0.w = __uint_as_float(rng);
        P.out.q0[dst] = a0;
        P.out.q1[dst] = s_q1[q];
        P.out.q2[dst] = s_q2[q];
        P.hout.tp[dst] = tp;
        P.hout.mat[dst] = hm;
      }
    }
    if (!MULTI) {
      for (int off2 = 32; off2 > 0; off2 >>= 1) missed += __shfl_down(missed, off2, 64);
      if (lane == 0) my_missed += missed;
    }
    __syncthreads();
  }
  // hitScene invocations: this step's slots that held a path + the next step's that were settled here; one of kTallyLines counters per block
  if (lane == 0 && my_missed + my_valid) atomicAdd(tally_line(totals, blockIdx.x), my_missed + my_valid);
  // what is left of each wave's last region becomes holes — all of its quarter of the block's claim if it never needed one
  if (w_rend == 0u && blockIdx.x * (uint32_t)kSChunk < n && s_region0 < kR0Full) {  // (a block without a chunk passed no barrier: s_region0 is not its to read)
    w_cur = s_region0 + (threadIdx.x >> 6) * wregion;
    w_rend = w_cur + wregion;
  }
  for (uint32_t i = w_cur + (uint32_t)lane; i < w_rend; i += 64u) {
    reinterpret_cast<uint32_t*>(P.out.q1 + i)[3] = PID_HOLE;
    P.hout.mat[i] = HITMAT_HOLE;
  }
  if (COUNT) reduce_counters(cn, totals, false);
}

// ray_color's loop body for scenes with ONE material class (no sort): every WAVE on its own, no barrier inside the loop.  A wave shades
// the same 64-slot groups of the block's chunks as in shade_body, stages its survivors in an LDS ring of its own (128 entries) and runs a
// flush pass — hitScene part 1 for 64 new rays, all lanes busy — whenever 64 are waiting; what is left goes out in one last, partial pass.
// (shade_body's three barriers per chunk had every wave wait for the block's slowest three times per 128 slots of its own work.)
template <bool IS, bool COUNT, bool MULTI>
DEV void shade_body_wave(const DevScene& S, const RenderConst& rc, const Paths& P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads,
                         unsigned long long* __restrict__ totals, int first, uint32_t resv) {
  reset_heads(heads);
  constexpr uint32_t kRing = 128, kWaves = kBlock / 64;
  static_assert((size_t)kWaves * kRing == (size_t)kSChunk, "the rings take the LDS the block version's staging arrays take");
  __shared__ float4 s_q0[kWaves * kRing], s_q1[kWaves * kRing], s_q2[kWaves * kRing];
  constexpr uint32_t kR0Empty = 0xffffffffu, kR0Busy = 0xfffffffeu, kR0Full = 0xfffffffdu;
  __shared__ uint32_t s_region0;  // the waves' first regions: one claim per block (see shade_body)
  if (threadIdx.x == 0) s_region0 = kR0Empty;
#ifdef PTMI_LANE_TALLY
  if (threadIdx.x < kLaneTallies * 2) s_lane_tally[threadIdx.x] = 0u;
  if (threadIdx.x < 4 * kTimeTallies) s_time_tally[threadIdx.x] = 0ull;
  if (threadIdx.x < 4) s_time_last[threadIdx.x] = 0ull;
#endif
  __syncthreads();
  const QuadL L = load_light(S);
  const int lane = lane_id();
  const uint32_t wv = threadIdx.x >> 6;
  float4 *const r0 = s_q0 + wv * kRing, *const r1 = s_q1 + wv * kRing, *const r2 = s_q2 + wv * kRing;
  const uint32_t n_carried = resv ? min(ctl->n_carried, resv) : 0u;  // (Carry: slots [n_carried, resv) of this queue hold nothing)
  uint32_t n = ctl->n_rays;
  if (n <= resv && n_carried == 0u) n = 0u;
  const uint32_t region = max((uint32_t)kSChunk, ((n / gridDim.x / kRegionDiv) + 511u) & ~511u);
  const uint32_t wregion = region / kWaves;
  uint32_t w_cur = 0, w_rend = 0;  // this wave's current output region of the next queue (wave-uniform)
  uint32_t head = 0, cnt = 0;      // the ring: `cnt` survivors wait from entry `head` on (wave-uniform)
  uint32_t my_valid = 0, my_missed = 0;
  Counters cn = {0, 0, 0, 0, 0};
  // hitScene part 1 for the first `take` (<= 64) waiting survivors, one per lane; settles definite misses, places the others in the next queue
  auto flush_pass = [&](uint32_t take) {
    const uint32_t q = (head + (uint32_t)lane) & (kRing - 1u);
    bool keep = false;
    float2 tp = make_float2(0.0f, 0.0f);
    uint32_t hm = 0u, rng = 0u;
    if ((uint32_t)lane < take) {
      LT(LT_FLUSH);
      const float4 a0 = r0[q], a1 = r1[q];
      TT(TT_RING_READ, a0.x + a1.x);
      rng = __float_as_uint(a0.w);
      prims_for_ray<COUNT>(S, mk3(a0), mk3(a1), rng, tp, hm, cn);
      TT(TT_FLUSH_PRIMS, tp.x + __uint_as_float(hm));
      if (kMissShortcut && !MULTI && hm == HITMAT_MISS) {  // traceRay.wgsl:12-16 (see shade_body)
        LT(LT_MISS_SHORTCUT);
        const float4 a2 = r2[q];
        end_sample_progressive(P, __float_as_uint(a1.w), mk3(rc.bg[0], rc.bg[1], rc.bg[2]) * mk3(a2), (__float_as_int(a2.w) & kAccWritten) != 0);
      } else {
        keep = true;
      }
    }
    const uint64_t km = __ballot(keep);
    const uint32_t kept = (uint32_t)__popcll(km);
    my_missed += take - kept;
    if (kept) {
      const uint32_t rank = lanes_below(km);
      const uint32_t b0 = w_cur, n0 = min(kept, w_rend - w_cur);
      uint32_t b1 = 0xffffffffu;
      w_cur += n0;
      if (kept > n0) {  // claim the wave's next region for the rest
        uint32_t nb = 0;
        bool full;
        if (w_rend == 0u) {  // the wave's first region: its quarter of the block's claim
          if (lane == 0) nb = atomicCAS(&s_region0, kR0Empty, kR0Busy);
          nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          if (nb == kR0Empty) {
            if (lane == 0) {
              nb = atomicAdd(&ctl[1].n_rays, region);
              if (nb + region > P.cap) {  // cannot happen with the host's sizing
                atomicAdd(&totals[15], 1ull);
                atomicSub(&ctl[1].n_rays, region);
                nb = kR0Full;
              }
              atomicExch(&s_region0, nb);
            }
            nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          } else {
            while (nb == kR0Busy) {
              __builtin_amdgcn_s_sleep(8);
              if (lane == 0) nb = atomicAdd(&s_region0, 0u);
              nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
            }
          }
          full = nb == kR0Full;
          nb += wv * wregion;
        } else {
          if (lane == 0) nb = atomicAdd(&ctl[1].n_rays, wregion);
          nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
          full = nb + wregion > P.cap;
          if (full && lane == 0) {
            atomicAdd(&totals[15], 1ull);
            atomicSub(&ctl[1].n_rays, wregion);
          }
        }
        if (!full) {  // (full: what still fitted is written, the rest is dropped and flagged — see shade_body)
          b1 = nb;
          w_cur = nb + (kept - n0);
          w_rend = nb + wregion;
        }
      }
      if (keep && (rank < n0 || b1 != 0xffffffffu)) {
        LT(LT_KEEP);
        const uint32_t dst = (rank < n0) ? (b0 + rank) : (b1 + (rank - n0));
        float4 a0 = r0[q];
        a0.w = __uint_as_float(rng);
        P.out.q0[dst] = a0;
        P.out.q1[dst] = r1[q];
        P.out.q2[dst] = r2[q];
        P.hout.tp[dst] = tp;
        P.hout.mat[dst] = hm;
      }
    }
    head = (head + take) & (kRing - 1u);
    cnt -= take;
    TT(TT_FLUSH_STORE, 0.0f);
  };
  // ray_color's loop body for the slots of the lanes with `active`, their state in `st` (load_slot); survivors go into the ring
  auto shade_group = [&](bool active, const SlotState& st) {
    bool survive = false, valid = false;
    NewState ns;  // (zeroed although only the survivors' values are read: left undefined, the allocator needs 8 registers more — scratch at 80 VGPRs)
    ns.o = ns.d = ns.T = mk3(0, 0, 0);
    ns.bounce = 0, ns.rng = 0, ns.pid = 0;
    TT(TT_OTHER, 0.0f);  // (what came before this group: loop bookkeeping, the previous flush's tail)
    if (active) {
      LT(LT_GROUP);
      TT(TT_LOAD1, st.q0.x + st.q1.x + st.q2.x + st.tp.x + __uint_as_float(st.hitmat));  // the slot's state has arrived
      valid = __float_as_uint(st.q1.w) != PID_HOLE;
      if (valid) {
        LT(LT_VALID);
        const TriFetch tf = tri_fetch(S, P.uv, st.slot, __float_as_uint(st.tp.y));  // (issued ahead of the material's loads, consumed after them)
        survive = shade_one<IS, MULTI>(S, rc, P, st, tf, L, ns);
      }
    }
    TT(TT_SHADE, ns.o.x + ns.T.x);  // material + hit geometry fetched (TT_LOAD2, marked inside shade_one) and the bounce computed
    my_valid += (uint32_t)__popcll(__ballot(valid));
    const uint64_t mk = __ballot(survive);
    if (survive) {
      LT(LT_STAGE);
      const uint32_t q = (head + cnt + lanes_below(mk)) & (kRing - 1u);
      r0[q] = make_float4(ns.o.x, ns.o.y, ns.o.z, __uint_as_float(ns.rng));
      r1[q] = make_float4(ns.d.x, ns.d.y, ns.d.z, __uint_as_float(ns.pid));
      r2[q] = make_float4(ns.T.x, ns.T.y, ns.T.z, __int_as_float(ns.bounce));
    }
    cnt += (uint32_t)__popcll(mk);
    TT(TT_STAGE, 0.0f);
  };
  {
    // The wave's groups — slots base + j0 .. + 63 of the block's chunks — one after the other, the NEXT group's state requested before the flush pass of
    // the current one: the loads travel while the wave tests quads (round 4: a wave spent 11 % of its cycles waiting for exactly these loads)
    uint32_t base = blockIdx.x * (uint32_t)kSChunk, j0 = wv * 64u;
    auto settle = [&]() {  // -> is there a group at (base, j0), moving on to the block's next chunk when this wave's groups of the current one are used up
      while (base < n && j0 >= min((uint32_t)kSChunk, n - base)) {
        base += gridDim.x * (uint32_t)kSChunk;
        j0 = wv * 64u;
      }
      return base < n;
    };
    auto fetch = [&](SlotState& st) {  // -> this lane's slot of group (base, j0) holds something; its state is on its way then
      const uint32_t j = j0 + (uint32_t)lane;
      const bool act = j < min((uint32_t)kSChunk, n - base) && !dead_slot(base + j, n_carried, resv);
      if (act) st = load_slot(P, base + j, first != 0, rc);
      return act;
    };
    SlotState cur;
    bool have = settle(), cur_act = false;
    if (have) cur_act = fetch(cur);
#pragma unroll 1
    while (have) {
      shade_group(cur_act, cur);
      j0 += kBlock;
      have = settle();
      SlotState nxt;
      bool nxt_act = false;
      if (have) nxt_act = fetch(nxt);
      if (cnt >= 64u) flush_pass(64u);
      cur = nxt;
      cur_act = nxt_act;
    }
  }
  if (cnt) flush_pass(cnt);
  if (MULTI) my_missed = 0;  // (no path ends in the flush phase then)
  if (lane == 0 && my_missed + my_valid) atomicAdd(tally_line(totals, blockIdx.x), my_missed + my_valid);
  __syncthreads();  // the block's claim, if any wave made one, is in s_region0 now
#ifdef PTMI_LANE_TALLY
  if (threadIdx.x < kLaneTallies * 2 && s_lane_tally[threadIdx.x]) atomicAdd(&g_lane_tally[threadIdx.x], (unsigned long long)s_lane_tally[threadIdx.x]);
  if (threadIdx.x < kTimeTallies) {
    unsigned long long t = 0;
    for (int w = 0; w < 4; w++) t += s_time_tally[w * kTimeTallies + threadIdx.x];
    if (t) atomicAdd(&g_time_tally[threadIdx.x], t);
  }
#endif
  if (w_rend == 0u && s_region0 < kR0Full) {  // never needed a region: all of this wave's quarter of the block's claim becomes holes
    w_cur = s_region0 + wv * wregion;
    w_rend = w_cur + wregion;
  }
  for (uint32_t i = w_cur + (uint32_t)lane; i < w_rend; i += 64u) {
    reinterpret_cast<uint32_t*>(P.out.q1 + i)[3] = PID_HOLE;
    P.hout.mat[i] = HITMAT_HOLE;
  }
  if (COUNT) reduce_counters(cn, totals, false);
}

// (Round 5 built two more forms of this body, both bit-exact through the parity suite, neither faster, neither kept — the code is in the history under "Experiment: ...":
//  * the new rays BINNED BY THE SIGNS OF THEIR DIRECTION before the flush pass (eight waves of a 512-thread block sharing eight lock-free LDS rings): rays of one octant agree
//    on every axis-aligned quad's facing test, so the pass runs 2.0 quads instead of 3.9 per group at 63 lanes instead of 32 — a quarter of the kernel's vector instructions
//    gone, and its time unchanged (7.9 against 7.75 ms on configs[1]);
//  * the body AS A LOOP: a lane keeps its path from bounce to bounce while the new ray needs no tree walk, only rays that entered the root box go through the next queue
//    (one in twelve on configs[1]: most of the kernel's traffic gone, k_bvh -16 % on dense queues) — and k_shade +7 %, its passes at 45-56 lanes instead of 63.
//  Together with a probe that ADDS traffic (16 bytes more per kept ray, +12 %: +6-8 % time) they say what the kernel is bound by: neither issue slots nor bytes but the
//  round trips of a wave's dependent chain at six waves per SIMD.  profiles/r05_shade_bins_ab.txt, r05_shade_loop_ab.txt, NOTES_r05 §3.)
// The kernel proper, twice: the progressive-mode variants without importance sampling fit 80 VGPRs — 6 waves per SIMD, which this
// latency-bound kernel turns into throughput (round 3: 5 -> 6 blocks per CU, -8 %) —, the others need up to 96 (5 waves; at 80 they spill).
template <bool IS, bool SORT, bool COUNT, bool MULTI>
__global__ __launch_bounds__(kBlock) PTMI_SHADE_ATTR void k_shade(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl, uint32_t* __restrict__ heads,
                                                                  unsigned long long* __restrict__ totals, int first, uint32_t resv) {
  if constexpr (SORT) shade_body<IS, COUNT, MULTI>(S, rc, P, ctl, heads, totals, first, resv);  // several material classes: block by block, sorted
  else shade_body_wave<IS, COUNT, MULTI>(S, rc, P, ctl, heads, totals, first, resv);            // one class: wave by wave
}
template <bool SORT, bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_shade6(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl,
                                                                                                uint32_t* __restrict__ heads, unsigned long long* __restrict__ totals, int first, uint32_t resv) {
  if constexpr (SORT) shade_body<false, COUNT, false>(S, rc, P, ctl, heads, totals, first, resv);
  else shade_body_wave<false, COUNT, false>(S, rc, P, ctl, heads, totals, first, resv);
}

// k_tail — a SHORT queue traced to the end in one launch: every lane takes a path and runs ray_color's loop for it (hitScene part 2 on
// LDS stacks with the state machine of k_bvh, ray_color's body by shade_one, hitScene part 1 for the new ray) until the path ends —
// the reference's own megakernel shape, used where the wavefront pipeline has run out of parallelism: the last bounces of a batch (3 %
// of configs[1]'s rays took 11 % of its launches and 0.8 ms of 15), a lone interactive frame, and the long thin tail of the reference's
// default MAX_BOUNCES = 100.  Launched in front of every step's k_bvh; declines (all blocks return at once) unless 0 < n_rays <= limit;
// when it has run, the block that finishes last zeroes the queue length, so the step's k_bvh / k_shade and every later step find nothing.
// Same per-ray arithmetic and visit order as the wavefront kernels (the same device functions), same counters and tallies.
constexpr int kTailTravBatch = 24;  // (12 / 32 / 40 lanes measured: profiles/r04_tail_trav_batch.txt)
template <bool IS, bool COUNT, bool MULTI, bool NOABORT>
DEV void tail_body(const DevScene& S, const RenderConst& rc, const Paths& P, StepCtl* __restrict__ ctl, unsigned long long* __restrict__ totals, int first, uint32_t limit,
                   int stack_size, int lds_entries, int spill_entries, int2* __restrict__ spill, const Carry& cy) {
  const uint32_t n_carried = cy.resv ? min(ctl->n_carried, cy.resv) : 0u;  // (Carry: slots [n_carried, resv) hold nothing, [0, n_carried) rays whose traversal goes on)
  const uint32_t n = ctl->n_rays;
  if (n <= cy.resv && n_carried == 0u) return;            // empty
  if ((n > cy.resv ? n - cy.resv : 0u) + n_carried > limit) return;  // too long for this kernel: the per-bounce kernels take the step
  extern __shared__ int lds_stack[];
  const int lane = lane_id();
  LaneStack2 stk;
  stk.lds = (lds_v2i_t*)lds_stack + lane;
  stk.spill = spill + (size_t)blockIdx.x * (size_t)spill_entries * 64 + lane;
  stk.lds_entries = lds_entries;
  const QuadL L = load_light(S);
  Counters cn = {0, 0, 0, 0, 0}, cb = {0, 0, 0, 0, 0};  // hitScene part 1 / part 2 (k_bvh's share is reported separately)
  uint32_t tally = 0;  // hitScene invocations
  const uint32_t root = __float_as_uint(S.root_lo.w);
  const uint32_t root_node = (root & REF_LEAF) ? root : (root & REF_IDX);
  // Lanes whose path has ended take the next slots of the wave's current 64-slot group (groups are dealt round-robin to the waves) once
  // kTailRefill of them are idle: a wave keeps its lanes busy across paths of different lengths instead of waiting for its longest one.
  SlotState st;
  st.q0 = st.q1 = st.q2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  st.tp = make_float2(0.0f, 0.0f);
  st.hitmat = HITMAT_HOLE, st.slot = 0;
  bool alive = false;
  bool resume = false;  // the lane's path was taken from the carry prefix: its hitScene part 2 goes on from the pool's record (first iteration only)
  int park_sp = -1;     // >= 0: the lane's walk was interrupted (Carry::park_below); its state waits in three entries on top of its own stack, from this one on
  float2 uv = make_float2(0.0f, 0.0f);  // barycentrics of the lane's triangle hit (the queue comes from k_generate / k_shade: none in it yet)
  uint32_t gnext = blockIdx.x, gbase = 0, pos = 64;  // next group to open; the open group's first slot and how many of its slots are taken
#ifdef PTMI_LANE_TALLY
  unsigned long long bt_cyc[kBvhTallies] = {0}, bt_last;
  uint32_t bt_marks[kBvhTallies] = {0}, bt_lanes[kBvhTallies] = {0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt_last)::"memory");
#endif
#pragma unroll 1
  for (;;) {
    BT(TB_VOTE, 0.0f, 0);
    uint64_t am = __ballot(alive);
    const uint32_t nidle = 64u - (uint32_t)__popcll(am);
    if (nidle >= (uint32_t)kTailRefill || am == 0ull) {
      if (pos == 64u && gnext * 64u < n) {
        gbase = gnext * 64u;
        gnext += gridDim.x;
        pos = 0;
      }
      if (pos < 64u) {
        const uint32_t take = min(nidle, 64u - pos), rank = lanes_below(~am);
        if (!alive && rank < take) {
          const uint32_t slot = gbase + pos + rank;
          if (slot < n && !dead_slot(slot, n_carried, cy.resv)) {
            st = load_slot(P, slot, first != 0, rc);
            alive = __float_as_uint(st.q1.w) != PID_HOLE;
            uv = make_float2(0.0f, 0.0f);
            resume = alive && slot < cy.resv;
          }
        }
        pos += take;
        am = __ballot(alive);
        BT(TB_TAKE, st.q1.x + st.tp.x + __uint_as_float(st.hitmat), take);
      }
    }
    if (am == 0ull) {
      if (pos == 64u && gnext * 64u >= n) break;  // nothing in flight, nothing left to take
      continue;
    }
    {
      // ---- hitScene part 2 (hitRay.wgsl:42-110) for the lanes whose ray entered the root box ----
      // ... once kTailTravBatch of them wait, or no other lane has anything to do: a lane whose ray needs the tree keeps it, untouched, while the others
      // go on shading and bouncing.  (Round 4.  The wave walks the tree for as long as its longest ray; where one new ray in twelve enters the root box —
      // configs[1] — that walk, five lanes wide, was most of what the kernel did.)
      const bool flagged = alive && (st.hitmat & HITMAT_BVH) != 0u;
      const uint64_t fmask = __ballot(flagged);
      const bool more_to_take = !(pos == 64u && gnext * 64u >= n);  // (once the queue is used up there is nothing to gain by waiting: every pass the walk is put off lengthens the wave's end)
      if (fmask != 0ull && ((int)__popcll(fmask) >= (more_to_take ? kTailTravBatch : 1) || __ballot(alive && !flagged) == 0ull)) {
        uint32_t node = flagged ? root_node : N_DONE;
        int sp = 0;
        const f3 o = mk3(st.q0), d = mk3(st.q1);
        const f3 inv = rcp3_exact_il(d);
        const uint32_t negmask = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        float ct = st.tp.x;
        ObjRay orr;
        orr.mesh = -1;
        orr.o = orr.d = o;
        if (S.uniform_gid >= 0) obj_ray_uniform(S, o, d, orr);
        TriHit hit = {0.0f, 0.0f, 0u, 0u};
        if (flagged && resume) {  // carried over by the last k_bvh launch: state word, stack and the closest hit so far come from the pool
          const uint32_t* rec = cy.pool_in + (size_t)st.slot * (size_t)cy.rec_words;
          node = rec[0];
          sp = (int)rec[1];
          ct = __uint_as_float(rec[2]);
          hit.u = __uint_as_float(rec[3]), hit.v = __uint_as_float(rec[4]), hit.prim = rec[5], hit.mat = rec[6];
          for (int e = 0; e < sp; e++) stack2_write(stk, e, rec[8 + 2 * e], __uint_as_float(rec[9 + 2 * e]));
        }
        resume = false;
        if (flagged && park_sp >= 0) {  // interrupted in an earlier walk: on from where it stopped
          uint32_t w0;
          float w1;
          sp = park_sp;
          stack2_read(stk, sp, w0, w1);
          node = w0, ct = w1;
          stack2_read(stk, sp + 1, w0, w1);
          hit.u = __uint_as_float(w0), hit.v = w1;
          stack2_read(stk, sp + 2, w0, w1);
          hit.prim = w0, hit.mat = __float_as_uint(w1);
          park_sp = -1;
        }
#pragma unroll 1
        for (;;) {
          const uint64_t wm = __ballot(node != N_DONE);
          if (wm == 0ull) break;
          // A walk lasts as long as its longest ray, and on a deep tree most of it runs a handful of lanes wide (871 k triangles: 7 of 64 on average, two thirds of a lone
          // frame's wave-cycles).  Once fewer than park_below lanes are left in it and another lane has something to do — a path to shade, or the queue a path to
          // take — the stragglers are parked and join the next walk.
          if (cy.park_below > 0 && (int)__popcll(wm) < cy.park_below && (more_to_take || __ballot(alive && !(flagged && node != N_DONE)) != 0ull)) break;
          if ((int)node < 0) {
            const int2 lc = (node & REF_MULTI) ? S.leaf_table[node & REF_IDX] : make_int2((int)(node & REF_IDX), 1);
            for (int j = 0; j < lc.y; j++) {
              const float4* rec = S.pretri + 4 * (size_t)(lc.x + j);
              const float4 g0 = rec[0], g1 = rec[1], g2 = rec[2], g3 = rec[3];
              tri_test2<COUNT>(S, lc.x + j, g0, g1, g2, g3, o, d, orr, ct, hit, cb);
            }
            node = pop_until_pass2(stk, sp, ct, cb, COUNT);
            BT(TB_WALK_LEAF, ct, __popcll(__ballot(1)));
          }
          if (node < N_INNER_LIMIT) {
            const float4* rec = S.pairs + 4 * (size_t)node;
            const float4 f0 = rec[0], f1 = rec[1], f2 = rec[2], f3v = rec[3];
            BT(TB_WALK_FETCH, f0.x + f1.x + f2.x + f3v.x, __popcll(__ballot(1)));
            node = inner_step2<COUNT, NOABORT>(f0, f1, f2, f3v, o, inv, S.tmin, negmask, ct, stack_size, stk, sp, cb);
            if (node == N_POP) node = pop_until_pass2(stk, sp, ct, cb, COUNT);
          }
          BT(TB_WALK_STEP, __uint_as_float(node), __popcll(__ballot(node != N_DONE)));
        }
        if (flagged && node != N_DONE) {  // parked: node / closest hit so far on top of the lane's own stack (its entries below stay where they are); HITMAT_BVH stays set
          stack2_write(stk, sp, node, ct);
          stack2_write(stk, sp + 1, __float_as_uint(hit.u), hit.v);
          stack2_write(stk, sp + 2, hit.prim, __uint_as_float(hit.mat));
          park_sp = sp;
        } else {
          if (hit.prim != 0u) {  // a triangle beat what part 1 had found
            st.tp = make_float2(ct, __uint_as_float(hit.prim));
            uv = make_float2(hit.u, hit.v);
            st.hitmat = hit.mat;
          }
          st.hitmat &= ~HITMAT_BVH;  // (the walk is done: the lane shades now)
        }
      }
      // ---- ray_color's loop body (traceRay.wgsl:10-80), for the lanes that do not wait for the tree ----
      const bool go = alive && (st.hitmat & HITMAT_BVH) == 0u;
      NewState ns;  // (zeroed although only the survivors' values are read: left undefined, the allocator needs 8 registers more — scratch at 80 VGPRs)
      ns.o = ns.d = ns.T = mk3(0, 0, 0);
      ns.bounce = 0, ns.rng = 0, ns.pid = 0;
      bool survive = false;
      if (go) {
        tally++;
        const TriFetch tf = tri_fetch_uv(S, uv, __float_as_uint(st.tp.y));
        survive = shade_one<IS, MULTI>(S, rc, P, st, tf, L, ns);
        alive = survive;
      }
      BT(TB_SHADE, ns.o.x + ns.T.x, __popcll(__ballot(go)));
      // ---- hitScene part 1 for the new ray (hitRay.wgsl:6-54) ----
      if (go && alive) {
        uint32_t rng = ns.rng, hm;
        float2 tp;
        prims_for_ray<COUNT>(S, ns.o, ns.d, rng, tp, hm, cn);
        if (kMissShortcut && !MULTI && hm == HITMAT_MISS) {  // traceRay.wgsl:12-16, as in k_shade's flush phase
          end_sample_progressive(P, ns.pid, mk3(rc.bg[0], rc.bg[1], rc.bg[2]) * ns.T, (ns.bounce & kAccWritten) != 0);
          tally++;
          alive = false;
        } else {
          st.q0 = make_float4(ns.o.x, ns.o.y, ns.o.z, __uint_as_float(rng));
          st.q1 = make_float4(ns.d.x, ns.d.y, ns.d.z, __uint_as_float(ns.pid));
          st.q2 = make_float4(ns.T.x, ns.T.y, ns.T.z, __int_as_float(ns.bounce));
          st.tp = tp;
          st.hitmat = hm;
        }
      }
      BT(TB_PRIMS, st.tp.x, __popcll(__ballot(go && survive)));
    }
  }
#ifdef PTMI_LANE_TALLY
  if (lane == 0)
    for (int k = 0; k < kBvhTallies; k++) {
      if (bt_marks[k] == 0u) continue;
      atomicAdd(&g_tail_tally[3 * k], bt_cyc[k]);
      atomicAdd(&g_tail_tally[3 * k + 1], (unsigned long long)bt_marks[k]);
      atomicAdd(&g_tail_tally[3 * k + 2], (unsigned long long)bt_lanes[k]);
    }
#endif
  for (int off2 = 32; off2 > 0; off2 >>= 1) tally += __shfl_down(tally, off2, 64);
  if (lane == 0 && tally) atomicAdd(tally_line(totals, blockIdx.x), tally);
  if (COUNT) {
    reduce_counters(cn, totals, false);
    reduce_counters(cb, totals, true);
  }
  if (lane == 0) {  // whoever finishes last closes the queue (every block has read its length by then)
    __threadfence();
    if (atomicAdd(&ctl->tail_done, 1u) == gridDim.x - 1u) {
      ctl->n_rays = 0u;
      ctl->n_carried = 0u;
    }
  }
}

// The kernel proper, twice (like k_shade / k_shade6): in progressive mode without importance sampling, on scenes WITHOUT spheres, the body fits 80 VGPRs with 16
// bytes of scratch — 6 waves per SIMD instead of 4: a lone 1080p frame of configs[1] 0.84 -> 0.74 ms, of the 871 k-triangle scene 3.4 -> 3.1 (round 4,
// profiles/r04_tail_occupancy.txt); with spheres and fog volumes in the scene (the reference's default) the same build spills in their code and loses 8 %, so
// those scenes, importance sampling and NUM_SAMPLES > 1 keep the compiler's own allocation (100-130 VGPRs, 4 waves).
template <bool IS, bool COUNT, bool MULTI, bool NOABORT>
__global__ __launch_bounds__(64) void k_tail(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl, unsigned long long* __restrict__ totals, int first, uint32_t limit,
                                             int stack_size, int lds_entries, int spill_entries, int2* __restrict__ spill, Carry cy) {
  tail_body<IS, COUNT, MULTI, NOABORT>(S, rc, P, ctl, totals, first, limit, stack_size, lds_entries, spill_entries, spill, cy);
}
template <bool COUNT, bool NOABORT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_tail6(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl,
                                                                                         unsigned long long* __restrict__ totals, int first, uint32_t limit, int stack_size,
                                                                                         int lds_entries, int spill_entries, int2* __restrict__ spill, Carry cy) {
  tail_body<false, COUNT, false, NOABORT>(S, rc, P, ctl, totals, first, limit, stack_size, lds_entries, spill_entries, spill, cy);
}

// A batch's step records: every queue but step 0's starts behind its carry prefix (Carry)
__global__ void k_init_ctl(StepCtl* __restrict__ ctl, int n, uint32_t resv) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  ctl[i].n_rays = i == 0 ? 0u : resv;
  ctl[i].tail_done = 0u;
  ctl[i].n_carried = 0u;
}

// main.wgsl:22-27 for the frame slots [f_begin, f_end) of the batch, in frame order; the call that folds slot 0 also
// tallies the batch's rays/paths.  (ptmi_render_frame's render-ahead folds one slot per call.)
__global__ __launch_bounds__(kBlock) void k_accumulate(RenderConst rc, Paths P, float4* __restrict__ fb, int n_steps,
                                                       unsigned long long* __restrict__ totals, int f_begin, int f_end) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < rc.n_local; j += gridDim.x * kBlock) {
    uint32_t pix = local_to_pixel(rc, j);
    float4 cur = fb[pix];
    f3 c = mk3(cur);
    // The frames are added in frame order (the f32 sum is the reference's), but their colours are FETCHED eight at a time: a pixel's thread used to walk
    // flag -> colour -> add one frame after the other, two dependent round trips per frame — with pixel tiles sharded over 8 GPUs that is 512 frames per
    // thread and an eighth of the threads (tools/shard_sim.py).
    constexpr int kAhead = 8;
    for (int f0 = f_begin; f0 < f_end; f0 += kAhead) {
      bool have[kAhead];
      float4 colv[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; k++) {
        const int f = f0 + k;
        have[k] = f < f_end && (!P.touched || P.touched[(size_t)f * rc.n_local + j]);
      }
#pragma unroll
      for (int k = 0; k < kAhead; k++) {
        colv[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // a path that never wrote its acc_radiance returned (0,0,0)
        if (have[k]) colv[k] = P.acc[(size_t)(f0 + k) * rc.n_local + j];
      }
#pragma unroll
      for (int k = 0; k < kAhead; k++) {
        const int f = f0 + k;
        if (f >= f_end) break;
        const f3 col = mk3(colv[k]);
        if (f == 0 && rc.reset_first) {
          c = col;
        } else {
          c = c + col;
        }
      }
    }
    fb[pix] = make_float4(c.x, c.y, c.z, 1.0f);
  }
  if (f_begin == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long rays = 0;
    if (n_steps > 0)
      for (int k = 0; k < kTallyLines; k++) rays += *tally_line(totals, (uint32_t)k);
    totals[0] += rays;
    totals[1] += (unsigned long long)rc.n_local * (unsigned long long)rc.n_frames * (unsigned long long)rc.num_samples;
  }
}

// Upload-time digest: the unit normal resolve_hit needs for a quad hit (common.wgsl:176), computed once per quad by the
// same device function the shading path would call per hit — same instructions, same bits.
__global__ void k_quad_digest(const float4* __restrict__ quads, int n, float4* __restrict__ unit_n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const f3 u = norm3(mk3(quads[5 * i + 3]));
  unit_n[i] = make_float4(u.x, u.y, u.z, 0.0f);
}

// Upload-time digest: pretri (DevScene) from the raw triangles, one thread per triangle — the subtractions and the cross product
// the shader performs per test (common.wgsl:199-201), the same f32 operations in the same order.  A mesh id outside [0, n_meshes)
// is reported through `first_bad` (smallest offending triangle index) instead of being dereferenced.
__global__ __launch_bounds__(kBlock) void k_pretri_digest(const float4* __restrict__ tris, int n_tris, const int4* __restrict__ meshes, int n_meshes,
                                                          const int* __restrict__ mesh_matword, float4* __restrict__ pretri, float4* __restrict__ trinorm,
                                                          uint32_t* __restrict__ first_bad) {
  const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
  if (i >= n_tris) return;
  const float4* t = tris + 6 * (size_t)i;
  const float4 A = t[0], B = t[1], C = t[2];
  const float mf = t[5].w;  // mesh_id travels as a float (lib/primitives/triangle.js:42-52)
  if (!(mf >= 0.0f) || !(mf < 2147483000.0f) || (int)mf >= n_meshes) {
    atomicMin(first_bad, (uint32_t)i);
    return;
  }
  const int mesh = (int)mf;
  const int4 me = meshes[mesh];
  const float ABx = B.x - A.x, ABy = B.y - A.y, ABz = B.z - A.z;
  const float ACx = C.x - A.x, ACy = C.y - A.y, ACz = C.z - A.z;
  float4* o = pretri + 4 * (size_t)i;
  o[0] = make_float4(A.x, A.y, A.z, __int_as_float(mesh));
  o[1] = make_float4(ABx, ABy, ABz, __int_as_float(mesh_matword[mesh]));
  o[2] = make_float4(ACx, ACy, ACz, __int_as_float(me.z));  // the mesh's global_id = its transform index
  o[3] = make_float4(ABy * ACz - ABz * ACy, ABz * ACx - ABx * ACz, ABx * ACy - ABy * ACx, 0.0f);
  const float4 nA = t[3], nB = t[4], nC = t[5];  // the shading normals of the winning hit (common.wgsl:230), copied bit for bit
  float4* tn = trinorm + 3 * (size_t)i;
  tn[0] = make_float4(nA.x, nA.y, nA.z, __int_as_float(me.z));
  tn[1] = make_float4(nB.x, nB.y, nB.z, 0.0f);
  tn[2] = make_float4(nC.x, nC.y, nC.z, 0.0f);
}

// ---- test hooks ------------------------------------------------------------------------------------
struct HitOut {
  int32_t hit;
  float t;
  float p[3];
  float normal[3];
  int32_t front_face;
  float material[16];
};

__global__ __launch_bounds__(kBlock) void k_resolve_hits(DevScene S, Paths P, uint32_t n, HitOut* __restrict__ out) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float2 h = P.hin.tp[i];
  uint32_t prim = __float_as_uint(h.y);
  HitOut o;
  for (int k = 0; k < 16; k++) o.material[k] = 0.0f;
  o.t = 0.0f;
  o.p[0] = o.p[1] = o.p[2] = o.normal[0] = o.normal[1] = o.normal[2] = 0.0f;
  o.front_face = 0;
  o.hit = (prim >> 28) != K_NONE;
  if (o.hit) {
    float4 r0 = P.in.q0[i], r1 = P.in.q1[i];
    HitGeom g = resolve_hit(S, mk3(r0), mk3(r1), h.x, tri_fetch(S, P.uv, i, prim), prim);
    o.t = h.x;
    o.p[0] = g.p.x, o.p[1] = g.p.y, o.p[2] = g.p.z;
    o.normal[0] = g.n.x, o.normal[1] = g.n.y, o.normal[2] = g.n.z;
    o.front_face = g.front ? 1 : 0;
    const float4* m = S.mats + 4 * (P.hin.mat[i] & HITMAT_ID);
    for (int k = 0; k < 4; k++) {
      float4 v = m[k];
      o.material[4 * k] = v.x, o.material[4 * k + 1] = v.y, o.material[4 * k + 2] = v.z, o.material[4 * k + 3] = v.w;
    }
  }
  out[i] = o;
}

__global__ void k_math_eval(int fn, size_t n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y ? y[i] : 0.0f, r = 0.0f;
  switch (fn) {
    case 0: r = ptm_sin(a); break;
    case 1: r = ptm_cos(a); break;
    case 2: r = ptm_acos(a); break;
    case 3: r = ptm_log(a); break;
    case 4: r = ptm_log2(a); break;
    case 5: r = ptm_exp2(a); break;
    case 6: r = ptm_pow(a, b); break;
    case 7: r = ptm_sqrt(a); break;
    case 8: r = ptm_min(a, b); break;
    case 9: r = ptm_max(a, b); break;
    case 10: r = a / b; break;
    // 11..13: the components of (a, a * 2^-20, a * 2^20) / b through operator/(f3, float) — the f64-reciprocal vector division
    case 11: r = (mk3(a, a * 9.5367431640625e-07f, a * 1048576.0f) / b).x; break;
    case 12: r = (mk3(a, a * 9.5367431640625e-07f, a * 1048576.0f) / b).y; break;
    case 13: r = (mk3(a, a * 9.5367431640625e-07f, a * 1048576.0f) / b).z; break;
  }
  out[i] = r;
}

// Exhaustive self-tests of the unary shortcuts (ptmi_selftest): every one of the 2^32 f32 arguments, candidate against the
// compiler's IEEE expansion; NaN results count as equal among themselves.  which: 0 rcp_exact, 1 sqrt_exact, 3 / 4 the bare v_rcp_f32 / v_sqrt_f32 (controls: these must fail), 2 rcp3_exact
// (the argument in turn in each of the three slots, next to two in-range companions, and all three slots equal).
__global__ __launch_bounds__(256) void k_selftest(int which, unsigned long long* __restrict__ bad, uint32_t* __restrict__ first_bad) {
  unsigned long long mine = 0;
  uint32_t first = 0xffffffffu;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * 256) {
    const uint32_t u = (uint32_t)i;
    const float x = __uint_as_float(u);
    bool ok = true;
    auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); };
    if (which == 0) ok = same(rcp_exact(x), rcp_ieee_slow(x)) && same(rcp_exact_il(x), rcp_ieee_slow(x));
    else if (which == 1) ok = same(sqrt_exact(x), sqrt_ieee_slow(x));
    else if (which == 3) ok = same(__builtin_amdgcn_rcpf(x), rcp_ieee_slow(x));   // the harness itself: the raw 1-ulp instructions must NOT pass
    else if (which == 4) ok = same(__builtin_amdgcn_sqrtf(x), sqrt_ieee_slow(x));
    else if (which >= 5 && which <= 7) ok = same(sqrt_cand(x, which - 5), sqrt_ieee_slow(x));  // candidates (ptmi_device.h), not in use unless one passes
    else {
      const float want = rcp_ieee_slow(x);
      const f3 a = rcp3_exact(mk3(x, 3.0f, -0.7f)), b = rcp3_exact_il(mk3(1.5f, x, 1e-3f)), c = rcp3_exact(mk3(-2.0f, 1e6f, x)), d = rcp3_exact_il(mk3(x, x, x));
      ok = same(a.x, want) && same(b.y, want) && same(c.z, want) && same(d.x, want) && same(d.y, want) && same(d.z, want) && a.y == 1.0f / 3.0f && a.z == 1.0f / -0.7f &&
           b.x == 1.0f / 1.5f && b.z == 1.0f / 1e-3f && c.x == -0.5f && c.y == 1.0f / 1e6f;
    }
    if (!ok) {
      mine++;
      first = min(first, u);
    }
  }
  if (mine) {
    atomicAdd(bad, mine);
    atomicMin(first_bad, first);
  }
}

// shaders/fragment.js:22-36 + shaders/common.wgsl:273-282: colour = fb/frameNum -> ACES approx -> gamma
__global__ __launch_bounds__(kBlock) void k_resolve_rgba8(const float4* __restrict__ fb, uint32_t npix, float frame_num, uchar4* __restrict__ out) {
  uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= npix) return;
  float4 c = fb[i];
  float v[3] = {c.x / frame_num, c.y / frame_num, c.z / frame_num};
  unsigned char q[3];
  const float inv_gamma = 1 / 2.2;
  for (int k = 0; k < 3; k++) {
    float v1 = v[k] * 0.6f;
    float a = (v1 * (2.51f * v1 + 0.03f)) / (v1 * (2.43f * v1 + 0.59f) + 0.14f);
    a = ptm_min(ptm_max(a, 0.0f), 1.0f);
    float g = ptm_pow(a, inv_gamma);
    // canvas store: unorm8 round-to-nearest
    float s = g * 255.0f + 0.5f;
    s = ptm_min(ptm_max(s, 0.0f), 255.0f);
    q[k] = (unsigned char)s;
  }
  out[i] = make_uchar4(q[0], q[1], q[2], 255);
}

}  // namespace ptmi
