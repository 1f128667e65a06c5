// ptmi_kernels.h — the wavefront integrator's kernels (gfx950).
//
//   k_generate   one thread per (frame slot, owned pixel): seeds the RNG (main.wgsl:16), builds the
//                camera ray (shootRay.wgsl), fills step 0's ray queue.
//   k_prims      hitScene part 1 (hitRay.wgsl:6-54): spheres, quads, root-box test for every queued ray — coherent;
//                rays that enter the root box go to the step's BVH list.
//   k_bvh        hitScene part 2 (hitRay.wgsl:42-110): persistent single-wave blocks, one ray per lane, LDS
//                traversal stacks, ballot-based lane refill from the BVH list; no barriers, tails only at the end.
//   k_shade      per 2048-path chunk: LDS counting sort by material class (bin-uniform waves), then ray_color's
//                loop body (traceRay.wgsl:10-80) + material_scatter + Russian roulette; survivors are compacted
//                in LDS into the next step's ray queue, finished samples fold into the pixel colour.
//   k_accumulate framebuffer read-modify-write of main.wgsl:22-27 for every frame slot, in frame order.
#pragma once
#include "ptmi_device.h"

namespace ptmi {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_generate(RenderConst rc, Paths P, uint32_t* __restrict__ q0, StepCtl* __restrict__ ctl) {
  uint32_t total = rc.n_local * (uint32_t)rc.n_frames;
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < total; g += gridDim.x * kBlock) {
    uint32_t f = g / rc.n_local, j = g - f * rc.n_local;
    uint32_t pix = local_to_pixel(rc, j);
    uint32_t pid = f * rc.npix + pix;
    // u32(uniforms.frameNum): the frame number travels through an f32 uniform (renderer.js:173)
    uint32_t rng = pix + (uint32_t)(float)(rc.frame0 + f) * 719393u;
    f3 o, d;
    camera_ray(rc, pix, 0, rng, o, d);
    P.ray[2 * (size_t)pid] = make_float4(o.x, o.y, o.z, 0.0f);
    P.ray[2 * (size_t)pid + 1] = make_float4(d.x, d.y, d.z, 0.0f);
    P.thr[pid] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0));
    P.acc[pid] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
    if (P.pixsum) P.pixsum[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    P.rng[pid] = rng;
    q0[g] = pid;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl[0].n_rays = total;
}

// Work distribution (both k_intersect and k_shade): the step's ray queue is cut into chunks of kChunk
// entries.  A block claims a chunk with ONE global atomic, its waves pull 64-entry sub-chunks from an
// LDS counter, and all queue bookkeeping (sorting by material class, compaction of survivors) happens
// in LDS.  Same-address global atomics cost ~11 ns each on MI355X, so per-wave global atomics would
// cap a step at ~90 M rays/s per counter; per-chunk atomics are 32x rarer and off the critical path.
constexpr int kChunk = 2048;
// Small queues (the Russian-roulette tail of a batch) use smaller chunks so that the work still
// spreads over ~target_blocks blocks instead of a few blocks walking 2048 rays each.
DEV uint32_t chunk_size_for(uint32_t n, uint32_t target_blocks) {
  uint32_t c = (n / target_blocks + 255u) & ~255u;
  return min((uint32_t)kChunk, max(256u, c));
}

DEV int bin_of(const DevScene& S, uint32_t prim, int mat) {
  if ((prim >> 28) == K_NONE) return BIN_MISS;
  float ty = S.mats[4 * mat + 3].z;
  return (ty == 0.0f) ? BIN_LAMBERTIAN : (ty == 1.0f) ? BIN_MIRROR : (ty == 2.0f) ? BIN_GLASS : (ty == 3.0f) ? BIN_ISOTROPIC : BIN_OTHER;
}

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

// hitScene, part 1 (hitRay.wgsl:6-54): spheres, quads and the ROOT box test for every ray of the step's queue.
// Coherent work (the primitive tables sit in SGPRs); rays that do not enter the root box are final, the others
// are appended to the step's BVH list (staged in LDS, one global atomic per 2048-ray chunk).
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_prims(DevScene S, Paths P, StepCtl* __restrict__ ctl, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ bvh_list, unsigned long long* __restrict__ totals) {
  __shared__ uint32_t s_list[kChunk];
  __shared__ uint32_t s_n, s_base;
  const int lane = lane_id();
  const uint32_t n = ctl->n_rays;
  const bool have_bvh = S.n_nodes > 0;
  Counters cn = {0, 0, 0, 0, 0};
  for (uint32_t base = blockIdx.x * (uint32_t)kChunk; base < n; base += gridDim.x * (uint32_t)kChunk) {
    const uint32_t m = min((uint32_t)kChunk, n - base);
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t j0 = (threadIdx.x & ~63u); j0 < m; j0 += kBlock) {
      const uint32_t j = j0 + lane;
      bool to_bvh = false;
      uint32_t pid = 0;
      if (j < m) {
        pid = queue[base + j];
        float4 r0 = P.ray[2 * (size_t)pid], r1 = P.ray[2 * (size_t)pid + 1];
        f3 o = mk3(r0), d = mk3(r1);
        Closest c;
        c.t = kMaxFloat;
        c.u = c.v = 0.0f;
        c.prim = K_NONE;
        c.mat = 0;
        if (S.n_spheres > 0) {
          uint32_t rng = P.rng[pid];
          uint32_t rng0 = rng;
          hit_spheres<COUNT>(S, o, d, rng, c, cn);
          if (rng != rng0) P.rng[pid] = rng;
        }
        hit_quads<COUNT>(S, o, d, c, cn);
        if (have_bvh) {
          if (COUNT) cn.node_visits++;
          const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          to_bvh = hit_aabb(S.root_lo, S.root_hi, c.t, o, inv);
        }
        P.hit[pid] = make_float4(c.t, c.u, c.v, __uint_as_float(c.prim));
        P.hitmat[pid] = ((c.prim >> 28) != K_NONE) ? (uint32_t)c.mat : 0xffffffffu;
      }
      const uint64_t lm = __ballot(to_bvh);
      if (lm) {
        const int leader = __ffsll((unsigned long long)lm) - 1;
        uint32_t lb = 0;
        if (lane == leader) lb = atomicAdd(&s_n, (uint32_t)__popcll(lm));
        lb = (uint32_t)__shfl((int)lb, leader, 64);
        if (to_bvh) s_list[lb + lanes_below(lm)] = pid;
      }
    }
    __syncthreads();
    const uint32_t cnt = s_n;
    if (threadIdx.x == 0 && cnt) s_base = atomicAdd(&ctl->n_bvh, cnt);
    __syncthreads();
    const uint32_t ob = s_base;
    for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) bvh_list[ob + j] = s_list[j];
    __syncthreads();
  }
  if (COUNT) {
    uint32_t v[5] = {cn.node_visits, cn.tri_tests, cn.sphere_tests, cn.quad_tests, cn.mat_fetches};
#pragma unroll
    for (int k = 0; k < 5; k++) {
      unsigned long long x = v[k];
      for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
      if (lane == 0 && x) atomicAdd(&totals[2 + k], x);
    }
  }
}

// A wave refills its idle lanes from its range of the BVH list once this many lanes are idle (or all are).
constexpr int kRefillThreshold = 16;
constexpr uint32_t kBvhRange = 512;  // list entries a wave claims per global atomic (less when the list is short)

// hitScene, part 2 (hitRay.wgsl:42-110): BVH traversal of the listed rays by persistent, barrier-free waves.
// One block = one wave, so LDS (the traversal stacks, stack_alloc x 512 B per wave) is the only thing that limits
// how many waves a CU holds.  Each lane runs the traversal state machine on one ray; when kRefillThreshold lanes
// have finished, the wave refills them (ballot) from the range of the list it has claimed (one global atomic per
// kBvhRange rays), so short rays never leave lanes idle behind a long one and tails exist only at kernel end.
//   FLAT = false (BVH fits the caches, VALU bound): while-while — inner steps and triangle tests run in
//   separate loops so that each runs with as many lanes as possible.
//   FLAT = true (large BVH, latency bound): one 64-byte record fetch per lane per iteration.
template <bool COUNT, bool FLAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_bvh(DevScene S, Paths P, StepCtl* __restrict__ ctl,
                                                                                       const uint32_t* __restrict__ bvh_list, int stack_size,
                                                                                       int refill_threshold, unsigned long long* __restrict__ totals) {
  extern __shared__ int lds_stack[];
  const int lane = lane_id();
  int* stk = lds_stack + lane;
  const uint32_t n = ctl->n_bvh;
  // short lists (the Russian-roulette tail) are cut into smaller ranges so that they still spread over all waves
  const uint32_t range = min(kBvhRange, max(64u, ((n / (2u * gridDim.x)) + 63u) & ~63u));
  Counters cn = {0, 0, 0, 0, 0};
  uint32_t rb = 0, re = 0;  // this wave's claimed range of the list (wave-uniform)
  bool exhausted = (n == 0);
  bool has = false;
  uint32_t mypid = 0;
  Trav t;
  t.cur = T_DONE;
  t.pending = 0;
  t.sp = 0;
  t.negmask = 0;
  t.o = t.d = t.inv = mk3(0, 0, 0);
  t.orr.mesh = -1;
  t.orr.o = t.orr.d = mk3(0, 0, 0);
  t.c.t = 0.0f, t.c.u = t.c.v = 0.0f, t.c.prim = K_NONE, t.c.mat = 0;
  const uint32_t root = __float_as_uint(S.root_lo.w);
  for (;;) {
    // retire finished rays (stores only: nothing here waits on memory)
    if (has && t.cur == T_DONE && t.pending == 0u) {
      P.hit[mypid] = make_float4(t.c.t, t.c.u, t.c.v, __uint_as_float(t.c.prim));
      P.hitmat[mypid] = ((t.c.prim >> 28) != K_NONE) ? (uint32_t)t.c.mat : 0xffffffffu;
      has = false;
    }
    uint64_t hm = __ballot(has);
    int nact = __popcll(hm);
    if ((64 - nact) >= (nact == 0 ? 1 : refill_threshold)) {
      if (rb == re && !exhausted) {  // claim the next range
        uint32_t nb = 0;
        if (lane == 0) nb = atomicAdd(&ctl->head_b, range);
        nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        if (nb >= n) {
          exhausted = true;
        } else {
          rb = nb;
          re = min(nb + range, n);
        }
      }
      const uint32_t avail = re - rb;
      if (avail) {
        const uint64_t idle = ~hm;
        const uint32_t k = lanes_below(idle);
        if (!has && k < avail) {
          mypid = bvh_list[rb + k];
          float4 r0 = P.ray[2 * (size_t)mypid], r1 = P.ray[2 * (size_t)mypid + 1];
          float4 h = P.hit[mypid];
          const uint32_t hmat = P.hitmat[mypid];
          t.o = mk3(r0);
          t.d = mk3(r1);
          t.inv = mk3(1.0f / t.d.x, 1.0f / t.d.y, 1.0f / t.d.z);
          t.negmask = (t.d.x < 0 ? 1u : 0u) | (t.d.y < 0 ? 2u : 0u) | (t.d.z < 0 ? 4u : 0u);
          t.c.t = h.x, t.c.u = h.y, t.c.v = h.z, t.c.prim = __float_as_uint(h.w);
          t.c.mat = (hmat != 0xffffffffu) ? (int)hmat : 0;
          t.orr.mesh = -1;
          t.sp = 0;
          if (root & REF_LEAF) {
            t.pending = root;
            t.cur = T_POP;
          } else {
            t.pending = 0;
            t.cur = root & REF_IDX;
          }
          has = true;
        }
        rb += min(avail, (uint32_t)(64 - nact));
        hm = __ballot(has);
        nact = __popcll(hm);
      }
    }
    if (nact == 0) {
      if (exhausted) break;
      continue;  // range was empty: claim the next one
    }
    const bool more = !(exhausted && rb == re);
    const int min_working = more ? (64 - refill_threshold + 1) : 1;
    int working;
    if (FLAT) {
      do {
        if (has && !(t.cur == T_DONE && t.pending == 0u)) trav_flat_iter<COUNT>(S, stack_size, stk, t, cn);
        working = __popcll(__ballot(has && !(t.cur == T_DONE && t.pending == 0u)));
      } while (working >= min_working);
    } else {
      do {
        while (has && t.cur != T_DONE && t.pending == 0u) trav_step<COUNT>(S, stack_size, stk, t, cn);
        if (has && t.pending != 0u) {
          visit_leaf<COUNT>(S, t.pending, t.o, t.d, t.orr, t.c, cn);
          t.pending = 0u;
        }
        working = __popcll(__ballot(has && t.cur != T_DONE));
      } while (working >= min_working);
    }
  }
  if (COUNT) {
    uint32_t v[5] = {cn.node_visits, cn.tri_tests, cn.sphere_tests, cn.quad_tests, cn.mat_fetches};
#pragma unroll
    for (int k = 0; k < 5; k++) {
      unsigned long long x = v[k];
      for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
      if (lane == 0 && x) {
        atomicAdd(&totals[2 + k], x);
        if (k == 0) atomicAdd(&totals[7], x);  // node visits below the root
        if (k == 4) atomicAdd(&totals[8], x);  // accepted triangle hits
      }
    }
  }
}

// One path's iteration of the `for i < MAX_BOUNCES` loop body after hitScene (traceRay.wgsl:10-80).
// Returns true when the path continues with a new ray (already stored), false when this slot is done.
template <bool IS>
DEV bool shade_one(const DevScene& S, const RenderConst& rc, const Paths& P, uint32_t pid, const QuadL& L) {
  float4 r0 = P.ray[2 * (size_t)pid], r1 = P.ray[2 * (size_t)pid + 1];
  f3 o = mk3(r0), d = mk3(r1);
  float4 T4 = P.thr[pid], A4 = P.acc[pid];
  f3 T = mk3(T4), acc = mk3(A4);
  int bounce = __float_as_int(T4.w), sample = __float_as_int(A4.w);
  uint32_t rng = P.rng[pid];

  bool sample_done = false;
  f3 radiance = acc;
  f3 no = o, nd = d;
  const float4 h = P.hit[pid];

  if ((__float_as_uint(h.w) >> 28) == K_NONE) {  // traceRay.wgsl:12-16
    radiance = acc + (mk3(rc.bg[0], rc.bg[1], rc.bg[2]) * T);
    sample_done = true;
  } else {
    int mat = (int)P.hitmat[pid];
    Material m = load_material(S, mat);
    // the queue is sorted by this class within each chunk, so `bin` is wave-uniform almost everywhere
    const int bin = (m.type == 0.0f) ? BIN_LAMBERTIAN : (m.type == 1.0f) ? BIN_MIRROR : (m.type == 2.0f) ? BIN_GLASS : (m.type == 3.0f) ? BIN_ISOTROPIC : BIN_OTHER;
    HitGeom g = resolve_hit(S, o, d, h.x, h.y, h.z, __float_as_uint(h.w));
    f3 emission = m.emission;
    if (!g.front) emission = mk3(0, 0, 0);  // traceRay.wgsl:19-22
    float doSpecular;
    bool skip_pdf;
    f3 unit_w = mk3(0, 0, 0);
    f3 sdir = material_scatter(bin, m, g, d, rng, doSpecular, skip_pdf, unit_w);
    f3 sorg = (bin == BIN_OTHER) ? mk3(0, 0, 0) : g.p;
    bool roulette = true;
    if (IS) {  // traceRay.wgsl:24-58
      if (skip_pdf) {
        acc = acc + emission * T;
        T = T * mix3(m.color, m.spec, doSpecular);
        no = sorg;
        nd = sdir;
        roulette = false;  // `continue` skips the Russian roulette
      } else {
        // get_random_on_quad(lights, hitRec.p) (importanceSampling.wgsl:78-81): u draw, then v draw
        float ru = rand2D(rng);
        f3 pu = ru * L.u;
        float rv = rand2D(rng);
        f3 pv = rv * L.v;
        f3 lp = L.Q + pu + pv;
        f3 ldir = norm3(lp - g.p);
        f3 so = g.p, sd = ldir;
        float rnd = rand2D(rng);
        if (rnd > 0.2f) {
          so = sorg;
          sd = sdir;
        }
        float cosine_theta = dot3(norm3(sd), unit_w);  // onb_lambertian_scattering_pdf :73-76
        float lambertian_pdf = ptm_max(0.0f, cosine_theta / kPi);
        float lpdf = light_pdf(L, so, sd);
        float pdf = 0.2f * lpdf + 0.8f * lambertian_pdf;
        if (pdf <= 0.00001f) {  // returns emission*throughput, dropping acc (Q8)
          radiance = emission * T;
          sample_done = true;
        } else {
          acc = acc + emission * T;
          T = T * ((lambertian_pdf * mix3(m.color, m.spec, doSpecular)) / pdf);
          no = so;
          nd = sd;
        }
      }
    } else {  // traceRay.wgsl:61-68
      acc = acc + emission * T;
      T = T * mix3(m.color, m.spec, doSpecular);
      no = sorg;
      nd = sdir;
    }
    if (!sample_done) {
      if (roulette && bounce > 2) {  // traceRay.wgsl:71-79
        float p = ptm_max(T.x, ptm_max(T.y, T.z));
        if (rand2D(rng) > p) {
          sample_done = true;
        } else {
          T = T * (1.0f / p);
        }
      }
      if (!sample_done) {
        bounce++;
        if (bounce >= rc.max_bounces) sample_done = true;  // loop exhausted: returns acc (Q8)
      }
      radiance = acc;
    }
  }

  if (!sample_done) {
    P.ray[2 * (size_t)pid] = make_float4(no.x, no.y, no.z, 0.0f);
    P.ray[2 * (size_t)pid + 1] = make_float4(nd.x, nd.y, nd.z, 0.0f);
    P.thr[pid] = make_float4(T.x, T.y, T.z, __int_as_float(bounce));
    P.acc[pid] = make_float4(acc.x, acc.y, acc.z, __int_as_float(sample));
    P.rng[pid] = rng;
    return true;
  }

  // pathTrace (shootRay.wgsl:5-49): pixColor += ray_color(ray); next sample continues the same RNG stream
  f3 sum = radiance;
  if (rc.num_samples > 1) {
    sum = mk3(P.pixsum[pid]) + radiance;
  } else {
    sum = mk3(0, 0, 0) + radiance;
  }
  sample++;
  if (sample < rc.num_samples) {
    P.pixsum[pid] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    uint32_t pix = pid % rc.npix;
    camera_ray(rc, pix, sample, rng, no, nd);
    P.ray[2 * (size_t)pid] = make_float4(no.x, no.y, no.z, 0.0f);
    P.ray[2 * (size_t)pid + 1] = make_float4(nd.x, nd.y, nd.z, 0.0f);
    P.thr[pid] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(0));
    P.acc[pid] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(sample));
    P.rng[pid] = rng;
    return true;
  }
  f3 fin = sum / rc.sample_div;
  P.acc[pid] = make_float4(fin.x, fin.y, fin.z, __int_as_float(sample));
  return false;
}

template <bool IS>
__global__ __launch_bounds__(kBlock) void k_shade(DevScene S, RenderConst rc, Paths P, StepCtl* __restrict__ ctl, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ q_next, uint32_t target_blocks) {
  __shared__ uint32_t s_out[kChunk];     // survivors of the chunk
  __shared__ uint32_t s_sorted[kChunk];  // the chunk's path ids grouped by material class
  __shared__ uint32_t s_cnt[NUM_BINS + 2];
  __shared__ uint32_t s_chunk, s_next, s_nout, s_base;
  const QuadL L = load_light(S);
  const int lane = lane_id();
  const uint32_t n = ctl->n_rays;
  const uint32_t csz = chunk_size_for(n, target_blocks);
  while (true) {
    if (threadIdx.x == 0) {
      s_chunk = atomicAdd(&ctl->head_s, 1u);
      s_next = 0;
      s_nout = 0;
    }
    if (threadIdx.x < NUM_BINS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = s_chunk * csz;
    if (base >= n) break;
    const uint32_t m = min(csz, n - base);
    // prologue: counting sort of the chunk by the material class of each path's hit (ballot ranks + LDS
    // counters), so that the waves below are (almost) uniform in the 4-way material switch of scatterRay.wgsl
    uint32_t keys[kChunk / kBlock];
    uint32_t pids[kChunk / kBlock];
#pragma unroll
    for (int r = 0; r < kChunk / kBlock; r++) {
      const uint32_t j = (uint32_t)r * kBlock + threadIdx.x;
      int bin = -1;
      pids[r] = 0;
      if (j < m) {
        const uint32_t pid = queue[base + j];
        pids[r] = pid;
        const uint32_t hmat = P.hitmat[pid];
        if (hmat == 0xffffffffu) {
          bin = BIN_MISS;
        } else {
          const float ty = S.mats[4 * (int)hmat + 3].z;
          bin = (ty == 0.0f) ? BIN_LAMBERTIAN : (ty == 1.0f) ? BIN_MIRROR : (ty == 2.0f) ? BIN_GLASS : (ty == 3.0f) ? BIN_ISOTROPIC : BIN_OTHER;
        }
      }
      const bool any = (uint32_t)r * kBlock < m;  // block-uniform: skip empty rounds of small chunks
      uint32_t rank = 0;
      if (any) rank = bin_rank(bin, s_cnt);
      keys[r] = (uint32_t)(bin & 7) | (rank << 3);
    }
    __syncthreads();
    {
      uint32_t off[NUM_BINS];
      uint32_t run = 0;
#pragma unroll
      for (int b = 0; b < NUM_BINS; b++) {
        off[b] = run;
        run += s_cnt[b];
      }
#pragma unroll
      for (int r = 0; r < kChunk / kBlock; r++) {
        const uint32_t j = (uint32_t)r * kBlock + threadIdx.x;
        if (j < m) {
          const uint32_t b = keys[r] & 7u;
          uint32_t o = off[0];
#pragma unroll
          for (int k = 1; k < NUM_BINS; k++) o = (b == (uint32_t)k) ? off[k] : o;
          s_sorted[o + (keys[r] >> 3)] = pids[r];
        }
      }
    }
    __syncthreads();
    while (true) {
      uint32_t sub = 0;
      if (lane == 0) sub = atomicAdd(&s_next, 64u);
      sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
      if (sub >= m) break;
      const uint32_t j = sub + lane;
      bool survive = false;
      uint32_t pid = 0;
      if (j < m) {
        pid = s_sorted[j];
        survive = shade_one<IS>(S, rc, P, pid, L);
      }
      const uint64_t mk = __ballot(survive);
      if (mk) {
        const int leader = __ffsll((unsigned long long)mk) - 1;
        uint32_t bb = 0;
        if (lane == leader) bb = atomicAdd(&s_nout, (uint32_t)__popcll(mk));
        bb = (uint32_t)__shfl((int)bb, leader, 64);
        if (survive) s_out[bb + lanes_below(mk)] = pid;
      }
    }
    __syncthreads();
    const uint32_t nout = s_nout;
    if (threadIdx.x == 0 && nout) s_base = atomicAdd(&ctl[1].n_rays, nout);
    __syncthreads();
    const uint32_t ob = s_base;
    for (uint32_t j = threadIdx.x; j < nout; j += kBlock) q_next[ob + j] = s_out[j];
    __syncthreads();
  }
}

// main.wgsl:22-27 for all frame slots of the batch, in frame order; also tallies rays/paths.
__global__ __launch_bounds__(kBlock) void k_accumulate(RenderConst rc, Paths P, float4* __restrict__ fb, const StepCtl* __restrict__ ctl, int n_steps,
                                                       unsigned long long* __restrict__ totals) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < rc.n_local; j += gridDim.x * kBlock) {
    uint32_t pix = local_to_pixel(rc, j);
    float4 cur = fb[pix];
    f3 c = mk3(cur);
    for (int f = 0; f < rc.n_frames; f++) {
      float4 a = P.acc[(size_t)f * rc.npix + pix];
      f3 col = mk3(a);
      if (f == 0 && rc.reset_first) {
        c = col;
      } else {
        c = c + col;
      }
    }
    fb[pix] = make_float4(c.x, c.y, c.z, 1.0f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long rays = 0;
    for (int s = 0; s < n_steps; s++) rays += ctl[s].n_rays;
    totals[0] += rays;
    totals[1] += (unsigned long long)rc.n_local * (unsigned long long)rc.n_frames * (unsigned long long)rc.num_samples;
  }
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
  float4 h = P.hit[i];
  uint32_t prim = __float_as_uint(h.w);
  HitOut o;
  for (int k = 0; k < 16; k++) o.material[k] = 0.0f;
  o.t = 0.0f;
  o.p[0] = o.p[1] = o.p[2] = o.normal[0] = o.normal[1] = o.normal[2] = 0.0f;
  o.front_face = 0;
  o.hit = (prim >> 28) != K_NONE;
  if (o.hit) {
    float4 r0 = P.ray[2 * (size_t)i], r1 = P.ray[2 * (size_t)i + 1];
    HitGeom g = resolve_hit(S, mk3(r0), mk3(r1), h.x, h.y, h.z, prim);
    o.t = h.x;
    o.p[0] = g.p.x, o.p[1] = g.p.y, o.p[2] = g.p.z;
    o.normal[0] = g.n.x, o.normal[1] = g.n.y, o.normal[2] = g.n.z;
    o.front_face = g.front ? 1 : 0;
    const float4* m = S.mats + 4 * P.hitmat[i];
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
  }
  out[i] = r;
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
