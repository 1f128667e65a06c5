// ptmi_device.h — device-side building blocks of the wavefront integrator (gfx950 only).
//
// Arithmetic rules: every float expression keeps the association of the WGSL source it implements
// (cited per function, paths relative to the reference repo); this file is compiled with
// -ffp-contract=off, so the only fused multiply-adds are the explicit ones inside ptmi_math.h and
// the compiler's correctly-rounded division / sqrt expansions.  Results are bit-identical to a
// scalar IEEE-754 evaluation of the shader.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptmi_math.h"

#define DEV __device__ __forceinline__

namespace ptmi {

// ---- lane tallies (measurement builds only: -DPTMI_LANE_TALLY, tools/shade_lanes.py) ------------------------------------------------------
// LT(k) at the head of a region of k_shade's code counts how often a wave enters it and with how many lanes of its exec mask set:
// where do the idle lanes of the kernel's instructions go?  Tallied in LDS per block, flushed by k_shade alone (the other kernels that share
// these device functions tally into LDS that nobody reads).  The `; LT_MARK k` comment lands in the ISA, so that tools/shade_lanes.py can
// attribute static instruction counts to the same regions.  Without the macro LT() is nothing.
#ifdef PTMI_LANE_TALLY
constexpr int kLaneTallies = 48;
__shared__ uint32_t s_lane_tally[kLaneTallies * 2];  // {visits, lanes} per point
__device__ unsigned long long g_lane_tally[kLaneTallies * 2];
template <int K>
DEV void lane_tally() {
  static_assert(K >= 0 && K < kLaneTallies, "tally point out of range");
  const unsigned long long m = __ballot(1);
  asm volatile("; LT_MARK %0" ::"n"(K));
  if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u) {
    atomicAdd(&s_lane_tally[2 * K], 1u);
    atomicAdd(&s_lane_tally[2 * K + 1], (uint32_t)__popcll(m));
  }
}
#define LT(k) lane_tally<k>()
// ... and a stopwatch per wave: TT(k, dep) closes the interval that began at the previous mark and charges it to region k; `dep` is a value the mark must wait
// for (the asm takes it as an operand, so the compiler puts the s_waitcnt for its load in front of the clock read).  Where do a wave's cycles go?
constexpr int kTimeTallies = 12;
__shared__ unsigned long long s_time_tally[4 * kTimeTallies];  // per wave of the block: cycles per region
__shared__ unsigned long long s_time_last[4];
__device__ unsigned long long g_time_tally[2 * kTimeTallies];  // {cycles, marks} per region
template <int K>
DEV void time_tally(float dep) {
  unsigned long long now;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) : "v"(dep) : "memory");
  const unsigned long long m = __ballot(1);
  if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u) {
    const uint32_t wv = threadIdx.x >> 6;
    const unsigned long long last = s_time_last[wv];
    if (last) atomicAdd(&s_time_tally[wv * kTimeTallies + K], now - last);
    s_time_last[wv] = now;
  }
}
#define TT(k, dep) time_tally<k>(dep)
#else
#define LT(k) ((void)0)
#define TT(k, dep) ((void)0)
#endif
enum : int { TT_OTHER = 0, TT_LOAD1 = 1, TT_LOAD2 = 2, TT_SHADE = 3, TT_STAGE = 4, TT_FLUSH_PRIMS = 5, TT_FLUSH_STORE = 6, TT_RING_READ = 7 };
// tally points
enum : int {
  LT_GROUP = 0, LT_VALID = 1, LT_MISS = 2, LT_HIT = 3, LT_RH_SPHERE = 4, LT_RH_VOLUME = 5, LT_RH_QUAD = 6, LT_RH_TRI = 7, LT_MS_LAMBERT = 8, LT_MS_MIRROR = 9,
  LT_MS_GLASS = 10, LT_MS_ISO = 11, LT_RR = 12, LT_ACC_CONT = 13, LT_END_SAMPLE = 14, LT_END_CHANGES = 15, LT_FLUSH = 16, LT_QUAD_LOOP = 17, LT_QUAD_FRONT = 18,
  LT_QUAD_DENOM = 19, LT_QUAD_T = 20, LT_QUAD_ACCEPT = 21, LT_ROOT_BOX = 22, LT_MISS_SHORTCUT = 23, LT_KEEP = 24, LT_DIV3_SLOW = 25, LT_RCP_SLOW = 26, LT_SQRT_SLOW = 27,
  LT_SPHERE_LOOP = 28, LT_IS_LIGHT = 29, LT_STAGE = 30
};

// ---- constants of shaders/header.wgsl:1-13,37 (abstract-float consts folded in f64, rounded once) ----
constexpr float kPi = 3.14159265358979323846;
constexpr float kTwoPi = 2.0 * 3.14159265358979323846;
constexpr float kMinFloat = 0.0001;
constexpr float kMaxFloat = 999999999.999;        // 1.0e9f
constexpr float kMaxFloatP1 = 999999999.999 + 1;  // also 1.0e9f: f32 ulp there is 64

// hit kinds (upper 4 bits of the packed primitive word)
enum : uint32_t { K_NONE = 0, K_SPHERE = 1, K_VOLUME = 2, K_QUAD = 3, K_TRI = 4 };
// shade bins = material_type of the effective material; MISS paths get their own bin
enum : int { BIN_LAMBERTIAN = 0, BIN_MIRROR = 1, BIN_GLASS = 2, BIN_ISOTROPIC = 3, BIN_OTHER = 4, BIN_MISS = 5, NUM_BINS = 6 };

struct f3 {
  float x, y, z;
};
DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 mk3(float4 v) { return f3{v.x, v.y, v.z}; }
DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
// (a.x / s, a.y / s, a.z / s): three IEEE-754 f32 divisions by one divisor, through ONE f64 reciprocal.
// The expansion of a correctly rounded f32 division is 11 instructions; this is 6 for the reciprocal + 4 per component.
// Why the bits are the same: r = 1/s in f64, v_rcp_f64 refined by one third-order step (measured on MI355X: 2^-24.4 relative
// error as it comes, 2^-48.7 after one Newton step — not enough — so the step is r(1 + e + e^2)), is within 1 ulp64; x * r, rounded
// to f64, is then within 2^-51 of x/s (relative).  A quotient of two f32 numbers that is not representable lies at least
// 2^-49 (relative) away from every f32 rounding boundary — write a = A*2^i, b = B*2^j, boundary (2M+1)*2^(k-1) with 24-bit
// A, B, M: the difference is a non-zero integer multiple of a power of two over B — so the f64 value rounds to f32 exactly
// as the infinitely precise quotient does (and overflow to inf, signed zeros, inf and NaN operands come out of the
// multiplication as they do out of the division).  The two exceptions are redone with the real division: a divisor that is
// 0, inf or NaN (the Newton steps turn those into NaN), and a SUBNORMAL quotient, where exact ties exist (the grid is
// coarser than the quotient's precision) and a tie must not be broken by the 2^-51.
__device__ __attribute__((noinline)) f3 div3_ieee(f3 a, float s) {
  LT(LT_DIV3_SLOW);
  return mk3(a.x / s, a.y / s, a.z / s);
}  // the rare path, kept out of line
DEV f3 operator/(f3 a, float s) {
  const double ds = (double)s;
  double r = __builtin_amdgcn_rcp(ds);  // 2^-24.4
  const double e = __builtin_fma(-ds, r, 1.0);
  r = __builtin_fma(r, __builtin_fma(e, e, e), r);  // r (1 + e + e^2): the residual drops to e^3 = 2^-73, what is left is the rounding
  f3 q = mk3((float)((double)a.x * r), (float)((double)a.y * r), (float)((double)a.z * r));
  constexpr int kDenorm = 0x090, kZeroInfNan = 0x267;  // v_cmp_class masks: +-denormal; +-0, +-inf, sNaN, qNaN
  if (__builtin_amdgcn_class(s, kZeroInfNan) | __builtin_amdgcn_class(q.x, kDenorm) | __builtin_amdgcn_class(q.y, kDenorm) | __builtin_amdgcn_class(q.z, kDenorm))
    q = div3_ieee(a, s);
  return q;
}
DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// ---- 1.0f / x and sqrt(x) with the bits of the IEEE operations, in fewer issue cycles -------------------------------------
// The compiler expands a correctly rounded f32 division into v_div_scale x2, v_rcp, five fma/mul, v_div_fmas, v_div_fixup
// (11 instructions, ~39 issue cycles, tools/valu_peak.hip), and a square root into 18 instructions of which 7 only serve
// subnormal, zero and infinite arguments.  For a UNARY function "the same bits" needs no argument: ptmi_selftest runs the
// candidate against the compiler's expansion over all 2^32 arguments on the device (tests/test_math.py::test_exhaustive_*).
//   rcp:  inside |x| in [2^-100, 2^100] v_div_scale and v_div_fixup are identities and v_div_fmas is an fma, so the expansion
//         with numerator 1 is y1 = fma(fma(-x, y0, 1), y0, y0); q1 = fma(fma(-x, y1, 1), y1, y1); q = fma(fma(-x, q1, 1), y1, q1).
//         Exhaustively (ptmi_selftest 0): the first step, y1, already has the bits of 1.0f / x for every x in the range; outside it, the division.
//   sqrt: inside x in [2^-90, 2^120] the scaling, un-scaling and the zero/inf pass-through drop out of the expansion.
// The out-of-range path: a call where registers are plentiful and code size matters (k_shade: inlining it cost configs[1] 4 %),
// inline inside k_bvh, whose triangle test pays for the registers saved around a call (configs[3]: 4 %).
__device__ __attribute__((noinline)) float rcp_ieee_slow(float x) {
  LT(LT_RCP_SLOW);
  return 1.0f / x;
}
__device__ __attribute__((noinline)) float sqrt_ieee_slow(float x) {
  LT(LT_SQRT_SLOW);
  return __builtin_sqrtf(x);
}
DEV float rcp_core(float x) {  // valid for |x| in [2^-100, 2^100]
  const float y0 = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, y0, 1.0f), y0, y0);
}
DEV bool rcp_in_range(uint32_t bits) { return ((bits & 0x7fffffffu) - (27u << 23)) < (200u << 23); }  // biased exponent in [27, 227)
DEV float rcp_exact(float x) {
  float r = rcp_core(x);
  if (!rcp_in_range(__float_as_uint(x))) r = rcp_ieee_slow(x);
  return r;
}
DEV float rcp_exact_il(float x) {  // the same with the fallback inline (k_bvh)
  float r = rcp_core(x);
  if (!rcp_in_range(__float_as_uint(x))) r = 1.0f / x;
  return r;
}
// (1/x, 1/y, 1/z) with one range test for the three
DEV f3 rcp3_exact(f3 a) {
  f3 r = mk3(rcp_core(a.x), rcp_core(a.y), rcp_core(a.z));
  const uint32_t lo = 27u << 23;
  const uint32_t ex = (__float_as_uint(a.x) & 0x7fffffffu) - lo, ey = (__float_as_uint(a.y) & 0x7fffffffu) - lo, ez = (__float_as_uint(a.z) & 0x7fffffffu) - lo;
  if (max(ex, max(ey, ez)) >= (200u << 23)) r = mk3(rcp_ieee_slow(a.x), rcp_ieee_slow(a.y), rcp_ieee_slow(a.z));
  return r;
}
DEV f3 rcp3_exact_il(f3 a) {
  f3 r = mk3(rcp_core(a.x), rcp_core(a.y), rcp_core(a.z));
  const uint32_t lo = 27u << 23;
  const uint32_t ex = (__float_as_uint(a.x) & 0x7fffffffu) - lo, ey = (__float_as_uint(a.y) & 0x7fffffffu) - lo, ez = (__float_as_uint(a.z) & 0x7fffffffu) - lo;
  if (max(ex, max(ey, ez)) >= (200u << 23)) r = mk3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
  return r;
}
DEV float sqrt_cand(float x, int which);
DEV float sqrt_exact(float x) { return sqrt_cand(x, 2); }  // candidate C below (round 3's v_sqrt_f32 + residual test of both neighbours was 12 instructions)
// Shorter sequences with the IEEE bits (round 4; ptmi_selftest 5..7 run each over all 2^32 arguments on the device): one Markstein correction
// s + (x - s*s) * h of v_sqrt_f32's result with h ~ 1/(2s) from v_rsq_f32 (A: 0 mismatches) or from v_rcp_f32 (B: 105 mismatches — not used), and the two-step
// scheme from v_rsq_f32 alone (C: 0 mismatches; one transcendental, 7 instructions, ~20 issue cycles — the one in use).  Same guarded range as round 3's.
DEV float sqrt_cand(float x, int which) {
  float r;
  if (which == 0) {
    const float s = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
    r = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  } else if (which == 1) {
    const float s = __builtin_amdgcn_sqrtf(x), h = __builtin_amdgcn_rcpf(s + s);
    r = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  } else {
    const float g = __builtin_amdgcn_rsqf(x), h = 0.5f * g;
    const float s0 = x * g;
    const float s1 = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
    r = __builtin_fmaf(__builtin_fmaf(-s1, s1, x), h, s1);
  }
  if ((__float_as_uint(x) - (37u << 23)) >= (210u << 23)) r = sqrt_ieee_slow(x);
  return r;
}
DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float len3(f3 a) { return sqrt_exact(dot3(a, a)); }
DEV f3 norm3(f3 a) { return a / len3(a); }
DEV f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }  // stays arithmetic (Q12)
DEV f3 reflect3(f3 e1, f3 e2) { return e1 - (2.0f * dot3(e2, e1)) * e2; }
DEV f3 refract3(f3 e1, f3 e2, float e3) {
  float d = dot3(e2, e1);
  float k = 1.0f - e3 * e3 * (1.0f - d * d);
  if (k < 0.0f) return mk3(0, 0, 0);
  return e3 * e1 - (e3 * d + sqrt_exact(k)) * e2;
}

// column-major mat4 (4 float4 columns) times (v, w): ((c0*x + c1*y) + c2*z) + c3*w
DEV float4 mat_mul_cols(float4 c0, float4 c1, float4 c2, float4 c3, f3 v, float w) {
  float4 r;
  r.x = ((c0.x * v.x + c1.x * v.y) + c2.x * v.z) + c3.x * w;
  r.y = ((c0.y * v.x + c1.y * v.y) + c2.y * v.z) + c3.y * w;
  r.z = ((c0.z * v.x + c1.z * v.y) + c2.z * v.z) + c3.z * w;
  r.w = ((c0.w * v.x + c1.w * v.y) + c2.w * v.z) + c3.w * w;
  return r;
}
DEV float4 mat_mul(const float4* __restrict__ m, f3 v, float w) {
  float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
  float4 r;
  r.x = ((c0.x * v.x + c1.x * v.y) + c2.x * v.z) + c3.x * w;
  r.y = ((c0.y * v.x + c1.y * v.y) + c2.y * v.z) + c3.y * w;
  r.z = ((c0.z * v.x + c1.z * v.y) + c2.z * v.z) + c3.z * w;
  r.w = ((c0.w * v.x + c1.w * v.y) + c2.w * v.z) + c3.w * w;
  return r;
}
// transpose(m) * (v, 0): component i is the dot of column i with (v, 0)
DEV f3 mat_mul_transposed_dir(float4 c0, float4 c1, float4 c2, f3 v) {
  return mk3(((c0.x * v.x + c0.y * v.y) + c0.z * v.z) + c0.w * 0.0f, ((c1.x * v.x + c1.y * v.y) + c1.z * v.z) + c1.w * 0.0f,
             ((c2.x * v.x + c2.y * v.y) + c2.z * v.z) + c2.w * 0.0f);
}

// shaders/common.wgsl:7-12
DEV float rand2D(uint32_t& s) {
  s = s * 747796405u + 2891336453u;
  uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (float)((word >> 22u) ^ word) / 4294967296.0f;
}

// ---- scene as the kernels see it -------------------------------------------------------------------
// Raw arrays keep the reference's byte layout.  Two digests are derived at upload:
//   pair64 : one 64-byte record per INNER node holding BOTH children's boxes, so that one fetch decides
//            two box tests:  {L.min.xyz, L.ref} {L.max.xyz, R.ref} {R.min.xyz, axis} {R.max.xyz, 0}
//            with L = node i+1, R = node right_offset(i).  A child ref is
//              inner : index of the child's own pair record
//              leaf  : REF_LEAF | prim_id                      (prim_count == 1, the reference's builder)
//                      REF_LEAF | REF_MULTI | leaf_table index (any other prim_count: {prim_id, count})
//   pretri : {A.xyz, mesh_id} {AB.xyz, material word} {AC.xyz, global_id} {cross(AB,AC).xyz,0}   (64 B, aligned)
//            the mesh's material word and transform index ride along, so that accepting a hit or building the
//            object-space ray needs no dependent `meshes[mesh]` fetch first.
// Both hold values the shader would load or compute itself (common.wgsl:199-201) — same f32 operations.
// "Material word" = material id | shade bin of that material << 28: sphere_info.x, quad_mat and pretri carry it, it
// flows through Closest.mat into the hitmat word, and k_shade sorts on it without touching the material table.
constexpr uint32_t REF_LEAF = 0x80000000u, REF_MULTI = 0x40000000u, REF_B = 0x20000000u, REF_A = 0x10000000u, REF_IDX = 0x0fffffffu;
struct DevScene {
  const float4* spheres;  // 2 float4 / sphere
  const int2* sphere_info;  // {material word, is_volume}
  const float4* quads;  // 5 float4 / quad
  const int* quad_mat;      // material word per quad
  const float4* trinorm;      // 3 x float4 per triangle: the vertex normals nA, nB, nC (common.wgsl:230) with the mesh's transform index in nA.w — what
                              // resolve_hit needs of a triangle hit in one 48-byte run (the raw 96-byte record + the pretri record: three cache lines)
  const float4* quad_unit_n;  // normalize(quad.normal), evaluated once per quad by k_quad_digest with the very same norm3()
  const float4* tris;    // 6 float4 / triangle (raw)
  const float4* pretri;  // 4 float4 / triangle
  const int4* meshes;
  const float4* xforms;  // 8 float4 / object: model[4], invModel[4]
  const float4* mats;    // 4 float4 / material
  const float4* pairs;   // 4 float4 / inner node
  const int2* leaf_table;
  float4 root_lo, root_hi;  // root box; root_lo.w = root's child ref
  int n_spheres, n_quads, n_tris, n_meshes, n_xforms, n_mats, n_nodes;
  int light_quad;  // first quad with emission.x > 0 (common.wgsl:258-269), -1 if none
  int uniform_gid;  // the transform index all meshes share (one mesh: every configuration of BASELINE.json), -1 if they differ: a triangle hit's
                    // normal matrix then comes through the scalar cache with the hit's other data instead of one dependent VMEM round trip later
  float tmin;      // ray_tmin (header.wgsl:37; ptmi_params.tmin, 0.000001 by default)
};

// ---- path state ----------------------------------------------------------------------------------------
// Live state is indexed by QUEUE SLOT and compacted every step: step s reads the `in` buffers, k_shade writes the
// survivors densely into the `out` buffers (= `in` of step s+1) TOGETHER with the first part of their next hitScene
// (spheres, quads, root box), so every kernel streams its state coalesced instead of gathering by path id, and a
// ray's state crosses HBM once per bounce in each direction.  Slots can be holes (pid == PID_HOLE): a block of k_shade claims output space in
// regions and marks what it did not use.  The hit record belongs to the slot of `in`.  Radiance (`acc`) stays
// indexed by path id = frame_slot * n_local + local pixel index: it changes rarely (emissive hits, misses) and k_accumulate needs it
// by pixel.
constexpr uint32_t PID_HOLE = 0xffffffffu;
constexpr uint32_t HITMAT_ID = 0x0fffffffu;    // bits 0..27: material id
constexpr int HITMAT_BIN_SHIFT = 28;           // bits 28..30: shade bin (BIN_*) of that material; 7 = no path
constexpr uint32_t HITMAT_WORD = 0x7fffffffu;  // id + bin = the "material word"
constexpr uint32_t HITMAT_MISS = ((uint32_t)BIN_MISS << HITMAT_BIN_SHIFT) | 0x0fffffffu;  // hitScene returned false (so far)
constexpr uint32_t HITMAT_HOLE = (7u << HITMAT_BIN_SHIFT) | 0x0ffffffeu;                   // slot holds no path
constexpr uint32_t HITMAT_BVH = 0x80000000u;   // flag: the ray entered the root box, k_bvh still has to traverse it

struct Slots {     // 48 bytes of live state per slot
  float4* q0;      // {origin.xyz, randState bits}; STEP 0's queue (k_generate's output) holds the randState alone, 4 bytes per slot at the start of the
                   // same buffer: every camera ray starts at cam_origin (rng0_of / load_slot)
  float4* q1;      // {dir.xyz, path id bits — PID_HOLE = no path}
  float4* q2;      // {throughput.xyz, bounce index | kAccWritten (ptmi_kernels.h: the path has stored its acc_radiance before)}
};
struct HitBuf {    // hitScene's result for the ray in the same slot: 12 bytes, plus 8 more for a triangle hit
  float2* tp;      // {t, kind<<28 | index}
  uint32_t* mat;   // effective material word (after hit_volume's clobber, Q3) / MISS / HOLE, | HITMAT_BVH
};
struct Paths {
  Slots in, out;     // this step's queue, next step's queue
  HitBuf hin, hout;  // hit records of `in` (k_generate / the previous k_shade + k_bvh), of `out` (this k_shade)
  float2* uv;        // by slot of `in`: barycentrics, written by k_bvh and read by k_shade for triangle hits only
  float4* acc;       // by path id: {acc_radiance.xyz, sample index as int bits}; the final pixel colour at the end
  float4* pixsum;    // by path id: {pixColor.xyz, -}  (num_samples > 1 only)
  uint8_t* touched;  // by path id, NUM_SAMPLES == 1 only (else null): 0 = acc[pid] has never been written and stands for (0,0,0) — the
                     // batch starts with one small memset of these flags instead of k_generate streaming 16 bytes of zeros per path
  uint32_t cap;      // slots per queue buffer
};

// One 128-byte line per step: k_shade's blocks and waves claim their output regions of the NEXT step's queue with atomics on n_rays, and
// same-line atomics serialise in one L2 channel at ~11 ns each — nothing else may share the line.
struct alignas(128) StepCtl {
  uint32_t n_rays;     // slots of this step's queue (holes and the carry prefix included)
  uint32_t tail_done;  // k_tail: blocks that have finished (the last one closes the queue)
  uint32_t n_carried;  // rays the previous step's k_bvh carried over into slots [0, n_carried) of this queue (Carry)
  uint32_t pad[29];
};
// Rays that outlive their k_bvh launch.  Every launch used to end with a tail as long as its longest ray (~1 ms per launch on an 871 k-triangle tree whatever
// the batch size: 8-10 % of the kernel).  Paths are independent, so a path may lag a step behind: a wave that has found the queue exhausted goes on for
// `after` iterations and then CARRIES its unfinished rays over — path state into a slot of the NEXT step's queue, traversal state (state word, stack,
// closest hit so far) into a record of the pool — and the next launch picks them up first, among all its other rays.  Slots [0, resv) of every queue but
// step 0's are reserved for them; [n_carried, resv) hold nothing and every reader of a queue skips them.  Same visits, same outcomes, same counters — only later.
struct Carry {
  uint32_t resv;       // reserved prefix of THIS queue
  uint32_t resv_next;  // of the next one; 0 = this launch carries nothing over
  int after;           // iterations a wave goes on after the queue is exhausted
  int rec_words;       // words per pool record: 8 (state word, sp, closest t, u, v, prim, material word, -) + 2 per stack entry
  uint32_t* pool_in;   // records of this queue's carried rays, by slot
  uint32_t* pool_out;  // records of the next queue's
  int park_below;      // k_tail only (round 5): its tree walk stops once fewer than this many lanes are still in it while other lanes have work, and the stragglers resume
                       // in the next walk (0 = a walk always runs to its longest ray's end) — tail_body
};
DEV bool dead_slot(uint32_t slot, uint32_t n_carried, uint32_t resv) { return slot >= n_carried && slot < resv; }
// The hitScene tally of the batch being traced (slots that held a path, summed over its steps) is spread over kTallyLines counters on
// lines of their own behind the 16 persistent totals (zeroed when a batch starts, added up by the k_accumulate call that folds its slot 0).
constexpr int kTallyLines = 32;
constexpr size_t kTotalsBytes = 128 + (size_t)kTallyLines * 128;
DEV uint32_t* tally_line(unsigned long long* totals, uint32_t k) {
  return reinterpret_cast<uint32_t*>(totals + 16) + (k % (uint32_t)kTallyLines) * 32u;
}

struct RenderConst {
  float W, H;
  float view[16];
  float cam_o[3];  // cam_origin (main.wgsl:8), see cam_origin()
  float fov_factor;
  float bg[3];
  int max_bounces;
  int num_samples;    // samples per pixel per frame actually traced (side*side when stratified)
  int stratify;
  int strat_side;
  float recip_sqrt_spp;
  float sample_div;   // divisor of pixColor (NUM_SAMPLES or numSamples)
  int stack_size;
  float light_mix, surface_mix;  // traceRay.wgsl:43,49: 0.2 and 1 - 0.2 = 0.8 (ptmi_params.light_mix)
  uint32_t npix;
  uint32_t frame0;
  int n_frames;
  int reset_first;
  // shard
  uint32_t n_local;
  int rank, world, tile;
};

DEV uint32_t local_to_pixel(const RenderConst& rc, uint32_t j) {
  uint32_t tl = j / (uint32_t)rc.tile, within = j - tl * (uint32_t)rc.tile;
  return (tl * (uint32_t)rc.world + (uint32_t)rc.rank) * (uint32_t)rc.tile + within;
}

// cam_origin = (view * (0,0,0,1)).xyz  (main.wgsl:8): the origin of EVERY camera ray of a frame, so step 0's queue does not store it — whoever
// reads that queue (k_shade / k_tail with `first`, k_bvh with its `cam` argument) takes it from here.  Evaluated once per batch by the host
// (render_batch: ((m0*0 + m4*0) + m8*0) + m12*1 per component, the shader's mat4 * vec4 in f32 — multiplications by 0 and 1 and additions,
// the same IEEE results on the host as on the device) instead of occupying twelve scalar registers with the view matrix in every kernel.
DEV f3 cam_origin(const RenderConst& rc) { return mk3(rc.cam_o[0], rc.cam_o[1], rc.cam_o[2]); }

// shaders/main.wgsl:3-8 + shaders/shootRay.wgsl:5-60: jittered camera ray for sample k of a pixel
// ... in two parts: what depends on the pixel alone (k_generate makes it once per pixel for all the frames of a batch), and the rest
DEV void camera_pixel(const RenderConst& rc, uint32_t pix, float& px, float& py) {
  const float W = rc.W;
  float fidx = (float)pix;
  float q = fidx / W;
  px = fidx - W * truncf(q);  // f32 %: x - y*trunc(x/y)
  py = q;                     // not floored (Q1)
}
DEV void camera_ray_at(const RenderConst& rc, float px, float py, int k, uint32_t& rng, f3& o, f3& d) {
  const float W = rc.W, H = rc.H;
  float a, b;
  if (rc.stratify) {
    float i = (float)(k / rc.strat_side), j = (float)(k % rc.strat_side);
    a = (W / H) * (2.0f * ((px - 0.5f + (rc.recip_sqrt_spp * (i + rand2D(rng)))) / W) - 1.0f);
    b = -1.0f * (2.0f * ((py - 0.5f + (rc.recip_sqrt_spp * (j + rand2D(rng)))) / H) - 1.0f);
  } else {
    a = (W / H) * (2.0f * ((px - 0.5f + rand2D(rng)) / W) - 1.0f);
    b = -1.0f * (2.0f * ((py - 0.5f + rand2D(rng)) / H) - 1.0f);
  }
  const float* m = rc.view;
  float nf = -rc.fov_factor;
  float dx = ((m[0] * a + m[4] * b) + m[8] * nf) + m[12] * 0.0f;
  float dy = ((m[1] * a + m[5] * b) + m[9] * nf) + m[13] * 0.0f;
  float dz = ((m[2] * a + m[6] * b) + m[10] * nf) + m[14] * 0.0f;
  float dw = ((m[3] * a + m[7] * b) + m[11] * nf) + m[15] * 0.0f;
  float len = sqrt_exact(((dx * dx + dy * dy) + dz * dz) + dw * dw);  // normalize() of the vec4, then .xyz
  d = mk3(dx, dy, dz) / len;
  o = cam_origin(rc);
}
DEV void camera_ray(const RenderConst& rc, uint32_t pix, int k, uint32_t& rng, f3& o, f3& d) {
  float px, py;
  camera_pixel(rc, pix, px, py);
  camera_ray_at(rc, px, py, k, rng, o, d);
}

// Loads whose address is the same for every lane of a wave (primitive tables walked by a wave-uniform loop index):
// read through the constant address space, so that the compiler emits scalar loads (s_load, SGPR results, several
// in flight, no VMEM round trip per access).  Legal because no kernel ever writes the scene tables.
#define PTMI_CONST_AS __attribute__((address_space(4)))
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
DEV float4 ldu(const float4* p) {  // (HIP's float4 is a class: its copy would fall back to a generic-pointer load)
  const v4f_t v = *(const PTMI_CONST_AS v4f_t*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
DEV int2 ldu(const int2* p) {
  const v2i_t v = *(const PTMI_CONST_AS v2i_t*)(p);
  return make_int2(v.x, v.y);
}
DEV int ldu(const int* p) { return *(const PTMI_CONST_AS int*)(p); }

// ---- closest-hit record carried in registers during hitScene ----------------------------------------
struct Closest {
  float t;        // closest_so_far
  float u, v;     // barycentrics of the winning triangle
  uint32_t prim;  // kind<<28 | index, K_NONE if nothing accepted yet
  int mat;        // effective material (material word: id | bin << 28)
};

struct Counters {
  uint32_t node_visits, tri_tests, sphere_tests, quad_tests, mat_fetches;
};

// roots of shaders/common.wgsl:33-52 (hit_sphere) / :78-99 (hit_sphere_local); returns false on miss
DEV bool sphere_root(f3 center, float r, float tmin, float tmax, f3 o, f3 d, float& root_out) {
  f3 oc = o - center;
  float a = dot3(d, d);
  float half_b = dot3(d, oc);
  float c = dot3(oc, oc) - r * r;
  float disc = half_b * half_b - a * c;
  if (disc < 0) return false;
  float sqrtd = sqrt_exact(disc);
  float root = (-half_b - sqrtd) / a;
  if (root <= tmin || root >= tmax) {
    root = (-half_b + sqrtd) / a;
    if (root <= tmin || root >= tmax) return false;
  }
  root_out = root;
  return true;
}

// shaders/hitRay.wgsl:6-31 with hit_sphere (common.wgsl:29-73) and hit_volume (:102-146)
template <bool COUNT>
DEV void hit_spheres(const DevScene& S, f3 o, f3 d, uint32_t& rng, Closest& c, Counters& cn) {
  for (int i = 0; i < S.n_spheres; i++) {
    float4 s0 = ldu(S.spheres + 2 * i);
    int2 info = ldu(S.sphere_info + i);
    f3 center = mk3(s0);
    float r = s0.w;
    LT(LT_SPHERE_LOOP);
    if (COUNT) cn.sphere_tests++;
    if (!info.y) {
      float root;
      if (sphere_root(center, r, S.tmin, c.t, o, d, root)) {
        c.t = root;
        c.prim = (K_SPHERE << 28) | (uint32_t)i;
        c.mat = info.x;
        if (COUNT) cn.mat_fetches++;
      }
    } else {
      float rec1, rec2;
      if (!sphere_root(center, r, -kMaxFloat, kMaxFloat, o, d, rec1)) continue;  // == MAX_FLOAT + 1 sentinel
      if (!sphere_root(center, r, rec1 + 0.0001f, kMaxFloat, o, d, rec2)) continue;
      if (rec1 < S.tmin) rec1 = S.tmin;
      if (rec2 > c.t) rec2 = c.t;
      if (rec1 >= rec2) continue;
      if (rec1 < 0) rec1 = 0;
      c.mat = info.x;  // hitRec.material written before the final accept/reject (Q3)
      if (COUNT) cn.mat_fetches++;
      float roughness = ldu(S.mats + 4 * (info.x & (int)HITMAT_ID) + 3).x;
      float ray_length = len3(d);
      float dist_inside = (rec2 - rec1) * ray_length;
      float hit_dist = roughness * ptm_log(rand2D(rng));
      if (hit_dist > dist_inside) continue;
      c.t = rec1 + (hit_dist / ray_length);
      c.prim = (K_VOLUME << 28) | (uint32_t)i;
    }
  }
}

// shaders/hitRay.wgsl:33-40 with hit_quad (common.wgsl:148-187)
// (Round 5 tried the quads in fewer, fuller passes — the loop's result is the accepted quad with the smallest t, the lowest index among equal t, so the order of the tests is
// free for finite rays; quads in parallel planes with exactly opposite normals (the walls of every scene the reference ships) then share ONE pass in which every lane works on
// the one it faces.  Bit-exact through the whole parity suite, and no faster: k_shade 7.6 -> 7.7 ms on configs[1] (the selects between two scalar-resident quad records — one
// SGPR operand per VALU instruction on gfx9 — and the finite-ray guard cost what the merged pass saves), k_generate 1.18 -> 1.48 ms (its coherent waves skip a back-facing wall
// as a whole in this loop; a shared pass never can).  profiles/r05_quad_pairs_ab.txt; the code is in the history (commit "Experiment: hit_quads ...").)
template <bool COUNT>
DEV void hit_quads(const DevScene& S, f3 o, f3 d, Closest& c, Counters& cn) {
  for (int i = 0; i < S.n_quads; i++) {
    const float4* q = S.quads + 5 * i;
    // the whole 80-byte record and the material id at once: scalar loads, one wait
    const float4 q0 = ldu(q), q1 = ldu(q + 1), q2 = ldu(q + 2), q3 = ldu(q + 3), q4 = ldu(q + 4);
    const int qmat = ldu(S.quad_mat + i);
    f3 n = mk3(q3);
    LT(LT_QUAD_LOOP);
    if (COUNT) cn.quad_tests++;
    if (dot3(d, n) > 0) continue;
    LT(LT_QUAD_FRONT);
    float denom = dot3(n, d);
    if (ptm_abs(denom) < 1e-8f) continue;
    LT(LT_QUAD_DENOM);
    float t = (q3.w - dot3(n, o)) / denom;
    if (t <= S.tmin || t >= c.t) continue;
    LT(LT_QUAD_T);
    f3 isect = o + t * d;
    f3 ph = isect - mk3(q0);
    f3 w = mk3(q4);
    float alpha = dot3(w, cross3(ph, mk3(q2)));
    float beta = dot3(w, cross3(mk3(q1), ph));
    if (alpha < 0 || 1 < alpha || beta < 0 || 1 < beta) continue;
    LT(LT_QUAD_ACCEPT);
    c.t = t;
    c.prim = (K_QUAD << 28) | (uint32_t)i;
    c.mat = qmat;
    if (COUNT) cn.mat_fetches++;
  }
}

// shaders/common.wgsl:245-256
DEV bool hit_aabb(float4 lo, float4 hi, float tmin, float tmax, f3 o, f3 inv) {
  float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
  float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
  float sx = ptm_min(t0x, t1x), sy = ptm_min(t0y, t1y), sz = ptm_min(t0z, t1z);
  float bx = ptm_max(t0x, t1x), by = ptm_max(t0y, t1y), bz = ptm_max(t0z, t1z);
  float t_min = ptm_max(tmin, ptm_max(sx, ptm_max(sy, sz)));
  float t_max = ptm_min(tmax, ptm_min(bx, ptm_min(by, bz)));
  return t_max > t_min;
}

// hitScene, part 1 (hitRay.wgsl:6-54): spheres, quads and the ROOT box test of the BVH loop's first iteration.  A ray
// that enters the root box gets HITMAT_BVH set in its material word: k_bvh finds its work by scanning those flags.
template <bool COUNT>
DEV void prims_for_ray(const DevScene& S, f3 o, f3 d, uint32_t& rng, float2& tp, uint32_t& hitmat, Counters& cn) {
  Closest c;
  c.t = kMaxFloat;
  c.u = c.v = 0.0f;
  c.prim = K_NONE;
  c.mat = 0;
  if (S.n_spheres > 0) hit_spheres<COUNT>(S, o, d, rng, c, cn);
  hit_quads<COUNT>(S, o, d, c, cn);
  bool to_bvh = false;
  if (S.n_nodes > 0) {
    if (COUNT) cn.node_visits++;
    LT(LT_ROOT_BOX);
    const f3 inv = rcp3_exact(d);  // 1 / ray.dir (hitRay.wgsl:46)
    to_bvh = hit_aabb(S.root_lo, S.root_hi, S.tmin, c.t, o, inv);
  }
  tp = make_float2(c.t, __uint_as_float(c.prim));
  hitmat = (((c.prim >> 28) != K_NONE) ? (uint32_t)c.mat : HITMAT_MISS) | (to_bvh ? HITMAT_BVH : 0u);
}

// Object-space ray of shaders/common.wgsl:193-197, cached per mesh (it is a pure function of the
// world ray and the mesh's invModelMatrix, so hoisting it out of the per-triangle test is exact).
struct ObjRay {
  f3 o, d;
  int mesh;
};
DEV void obj_ray_for(const DevScene& S, int mesh, int gid, f3 o, f3 d, ObjRay& r) {
  const float4* inv = S.xforms + 8 * gid + 4;
  float4 o4 = mat_mul(inv, o, 1.0f), d4 = mat_mul(inv, d, 0.0f);
  r.o = mk3(o4);
  r.d = mk3(d4);
  r.mesh = mesh;
}

// The same when every mesh of the scene shares one transform (S.uniform_gid >= 0: one mesh, or several placed together — every configuration of
// BASELINE.json): the matrix comes through the scalar cache and the object-space ray is made ONCE, where the ray is picked up, so that a ray's first
// triangle test no longer starts with a vector-memory round trip for the matrix (round 4: a fifth of k_bvh's wave-cycles went to the leaf phase).
DEV void obj_ray_uniform(const DevScene& S, f3 o, f3 d, ObjRay& r) {
  const float4* inv = S.xforms + 8 * S.uniform_gid + 4;
  const float4 c0 = ldu(inv), c1 = ldu(inv + 1), c2 = ldu(inv + 2), c3 = ldu(inv + 3);
  r.o = mk3(mat_mul_cols(c0, c1, c2, c3, o, 1.0f));
  r.d = mk3(mat_mul_cols(c0, c1, c2, c3, d, 0.0f));
}


// t-interval of a ray against one box, the closest-independent part of hit_aabb (common.wgsl:246-253):
// ts = max(tmin, max3(tsmaller)), tb = min3(tbigger); the box passes iff min(closest, tb) > ts.
DEV void slab(float4 lo, float4 hi, f3 o, f3 inv, float tmin, float& ts, float& tb) {
  float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
  float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
  float sx = ptm_min(t0x, t1x), sy = ptm_min(t0y, t1y), sz = ptm_min(t0z, t1z);
  float bx = ptm_max(t0x, t1x), by = ptm_max(t0y, t1y), bz = ptm_max(t0z, t1z);
  ts = ptm_max(tmin, ptm_max(sx, ptm_max(sy, sz)));
  tb = ptm_min(bx, ptm_min(by, bz));
}


// ---- second edition of the traversal state machine (k_bvh2) -----------------------------------------------------------
// Same visits, same outcomes, same counters as trav_inner_phase / trav_leaf_phase / trav_pop_until_pass; what changed is how
// the wave spends its instructions (round-2 profile: the pop loop — two FLAT loads per attempt because the LDS / spill choice
// had been folded into one generic pointer, ~45 instructions per attempt at 7-25 % active lanes — cost as many issue slots as
// the slab tests, and every triangle phase was a memory round trip of its own):
//   * one state word per lane: pair index of an inner node whose box has passed | leaf ref (REF_LEAF set) whose triangles are
//     still to be tested | N_POP | N_DONE;
//   * a stack entry is 8 contiguous bytes per lane (entry e of lane l at int2 index e*64 + l): ds_read_b64 / ds_write_b64, always
//     an LDS instruction; the rare deeper entries (per-wave spill area in global memory) are a separate, skipped branch;
constexpr uint32_t N_DONE = 0x7fffffffu, N_POP = 0x7ffffffeu, N_INNER_LIMIT = 0x10000000u;
// The LDS part is addressed through an explicit address-space-3 pointer: with a generic one the compiler folded the LDS / spill choice
// into FLAT loads and stores (both pipes, both counters) — the round-2 kernel's pop did exactly that.
typedef v2i_t __attribute__((address_space(3))) lds_v2i_t;
struct LaneStack2 {
  lds_v2i_t* lds;
  int2* spill;
  int lds_entries;
};
DEV void stack2_write(const LaneStack2& k, int e, uint32_t w0, float w1) {
  if (e < k.lds_entries) {
    v2i_t v;
    v.x = (int)w0, v.y = __float_as_int(w1);
    k.lds[e * 64] = v;
  } else {
    k.spill[(e - k.lds_entries) * 64] = make_int2((int)w0, __float_as_int(w1));
  }
}
DEV void stack2_read(const LaneStack2& k, int e, uint32_t& w0, float& w1) {
  const v2i_t v = k.lds[min(e, k.lds_entries - 1) * 64];
  w0 = (uint32_t)v.x;
  w1 = __int_as_float(v.y);
  if (e >= k.lds_entries) {
    const int2 g = k.spill[(e - k.lds_entries) * 64];
    w0 = (uint32_t)g.x;
    w1 = __int_as_float(g.y);
  }
}

// One reference visit of an inner node's near child (boxes in the fetched pair record), far child pushed: trav_inner_phase
// without its fetch and without the pops.  Returns the lane's next state word.
template <bool COUNT, bool NOABORT>
DEV uint32_t inner_step2(float4 f0, float4 f1, float4 f2, float4 f3v, f3 o, f3 inv, float tmin, uint32_t negmask, float ct, int stack_size, const LaneStack2& stk,
                         int& sp, Counters& cn) {
  float tsL, tbL, tsR, tbR;
  slab(f0, f1, o, inv, tmin, tsL, tbL);
  slab(f2, f3v, o, inv, tmin, tsR, tbR);
  const int axis = __float_as_int(f2.w);
  const bool neg = ((negmask >> axis) & 1u) != 0u;
  const uint32_t refL = __float_as_uint(f0.w), refR = __float_as_uint(f1.w);
  const uint32_t nearRef = neg ? refR : refL;
  uint32_t farRef = neg ? refL : refR;
  const float tsN = neg ? tsR : tsL, tbN = neg ? tbR : tbL;
  const float tsF = neg ? tsL : tsR, tbF = neg ? tbL : tbR;
  const bool fB = tbF > tsF;
  const bool fA = fB || (tbF != tbF);
  farRef |= (fA ? REF_A : 0u) | (fB ? REF_B : 0u);
  if (NOABORT) {
    if (fA) {
      stack2_write(stk, sp, farRef, tsF);
      sp++;
    } else if (COUNT) {
      cn.node_visits++;  // the pop + failed re-test the reference performs later
    }
  } else {
    stack2_write(stk, sp, farRef, tsF);
    sp++;
    if (sp >= stack_size) return N_DONE;  // Q7
  }
  if (COUNT) cn.node_visits++;
  return (ptm_min(ct, tbN) > tsN) ? nearRef : N_POP;  // a leaf ref has REF_LEAF set, an inner ref is the pair index
}

// (Round 5 built the walk that takes TWO levels of the reference's tree per fetch — a 128-byte record with a node's four grandchildren in the binary walk's nested near / far
// order, the first tested at once, the others pushed with their own interval starts; the children's own boxes can be skipped because boxes nest exactly and slab() is monotone.
// Bit-exact through the parity suite (same triangles in the same order), half the fetches and steps at twice the bytes and ~1.1x the VALU work: k_bvh -2.3 % on configs[2],
// -4 % on configs[1] / [4], +6 % on configs[3], whatever the LDS stack depth — profiles/r05_bvh_wide_ab.txt, r05_bvh_wide_regions.txt, r05_bvh_wide_lds_stack.txt.  Under the
// round's bar (-10 % on configs[2]), so it is in the history (commit "Experiment: k_bvh walking two levels ..."), not in the tree.)
// Pops until an entry passes its re-test against the current closest hit (or the stack is empty): trav_pop_until_pass.
DEV uint32_t pop_until_pass2(const LaneStack2& stk, int& sp, float ct, Counters& cn, bool count) {
  uint32_t node = N_POP;
  while (node == N_POP) {
    if (sp == 0) {
      node = N_DONE;
      break;
    }
    sp--;
    uint32_t e;
    float ts;
    stack2_read(stk, sp, e, ts);
    if (count) cn.node_visits++;
    const bool pass = (((e & REF_A) != 0u) & (ct > ts)) | ((ct != ct) & ((e & REF_B) != 0u));
    if (pass) node = (e & REF_LEAF) ? e : (e & REF_IDX);
  }
  return node;
}

// hit_triangle on a fetched pretri record (same arithmetic as tri_record_test), state in plain registers.
struct TriHit {
  float u, v;
  uint32_t prim, mat;  // prim == 0: no triangle accepted yet
};
template <bool COUNT>
DEV void tri_test2(const DevScene& S, int k, float4 t0, float4 t1, float4 t2, float4 t3, f3 o, f3 d, ObjRay& orr, float& ct, TriHit& h, Counters& cn) {
  int mesh = __float_as_int(t0.w);
  if (S.uniform_gid < 0 && mesh != orr.mesh) obj_ray_for(S, mesh, __float_as_int(t2.w), o, d, orr);  // (uniform_gid >= 0: made at pick-up, obj_ray_uniform)
  if (COUNT) cn.tri_tests++;
  f3 A = mk3(t0), AB = mk3(t1), AC = mk3(t2), N = mk3(t3);
  float det = -dot3(orr.d, N);
  if (ptm_abs(det) < S.tmin) return;
  f3 ao = orr.o - A;
  f3 dao = cross3(ao, orr.d);
  float invDet = rcp_exact_il(det);
  float dst = dot3(ao, N) * invDet;
  float u = dot3(AC, dao) * invDet;
  float v = -dot3(AB, dao) * invDet;
  float w = 1.0f - u - v;
  if (dst < S.tmin || dst > ct || u < S.tmin || v < S.tmin || w < S.tmin) return;
  ct = dst;
  h.u = u, h.v = v;
  h.prim = (K_TRI << 28) | (uint32_t)k;
  h.mat = __float_as_uint(t1.w);
  if (COUNT) cn.mat_fetches++;
}

// ---- HitRecord reconstruction (the accepting branch of the winning primitive test) ------------------
struct HitGeom {
  f3 p, n;
  bool front;
};
// What resolve_hit needs of a TRIANGLE hit, fetched in one go as soon as the hit record is known (k_shade issues these loads together
// with the material's instead of one dependent round trip after the other: barycentrics, the vertex normals, the mesh's transform id).
struct TriFetch {
  float2 uv;
  float4 nA, nB, nC;
  // (the mesh's transform index is nA.w — = meshes[i32(nC.w)].global_id, put there by k_pretri_digest — read where it is used: a copy made here would make
  // the fetch wait for the record before the material's loads are even issued)
};
DEV TriFetch tri_fetch(const DevScene& S, const float2* __restrict__ uvbuf, uint32_t slot, uint32_t prim) {
  TriFetch f;  // (read by resolve_hit's K_TRI and K_QUAD branches only: nothing to initialise for the other lanes — 14 moves per group)
  const uint32_t idx = prim & 0x0fffffffu;
  if ((prim >> 28) == K_TRI) {
    f.uv = uvbuf[slot];
    const float4* tn = S.trinorm + 3 * (size_t)idx;
    f.nA = tn[0], f.nB = tn[1], f.nC = tn[2];
  } else if ((prim >> 28) == K_QUAD) {
    f.nA = S.quad_unit_n[idx];  // a quad hit's unit normal rides in the same registers: asked for here, with everything else, instead of in the middle of resolve_hit (round 4)
  }
  return f;
}
// (k_tail holds the barycentrics in registers)
DEV TriFetch tri_fetch_uv(const DevScene& S, float2 uv, uint32_t prim) {
  TriFetch f;
  f.uv = uv;
  const uint32_t idx = prim & 0x0fffffffu;
  if ((prim >> 28) == K_TRI) {
    const float4* tn = S.trinorm + 3 * (size_t)idx;
    f.nA = tn[0], f.nB = tn[1], f.nC = tn[2];
  } else if ((prim >> 28) == K_QUAD) {
    f.nA = S.quad_unit_n[idx];
  }
  return f;
}
DEV HitGeom resolve_hit(const DevScene& S, f3 o, f3 d, float t, const TriFetch& tf, uint32_t prim) {
  HitGeom g;
  uint32_t kind = prim >> 28, idx = prim & 0x0fffffffu;
  g.p = o + t * d;  // at(ray, t)
  if (kind == K_SPHERE) {  // common.wgsl:54-68
    LT(LT_RH_SPHERE);
    float4 s0 = S.spheres[2 * idx];
    g.n = norm3((g.p - mk3(s0)) / s0.w);
    g.front = dot3(d, g.n) < 0;
    if (!g.front) g.n = -g.n;
  } else if (kind == K_VOLUME) {  // common.wgsl:140-143
    LT(LT_RH_VOLUME);
    float4 s0 = S.spheres[2 * idx];
    g.n = norm3(g.p - mk3(s0));
    g.front = true;
  } else if (kind == K_QUAD) {  // common.wgsl:176-183: normalize(quad.normal) is a per-quad constant, read from the digest
    LT(LT_RH_QUAD);
    g.n = mk3(tf.nA);  // = S.quad_unit_n[idx] (tri_fetch)
    g.front = dot3(d, g.n) < 0;
    if (!g.front) g.n = -g.n;
  } else {  // K_TRI, common.wgsl:224-237
    LT(LT_RH_TRI);
    float w = 1.0f - tf.uv.x - tf.uv.y;
    f3 nn = mk3(tf.nA) * w + mk3(tf.nB) * tf.uv.x + mk3(tf.nC) * tf.uv.y;
    float4 c0, c1, c2;  // columns 0..2 of the mesh's invModelMatrix
    if (S.uniform_gid >= 0) {  // one transform for every mesh: a wave-uniform address, known before the triangle's record arrives
      const float4* m = S.xforms + 8 * S.uniform_gid + 4;
      c0 = ldu(m), c1 = ldu(m + 1), c2 = ldu(m + 2);
    } else {
      const float4* m = S.xforms + 8 * __float_as_int(tf.nA.w) + 4;
      c0 = m[0], c1 = m[1], c2 = m[2];
    }
    g.n = norm3(mat_mul_transposed_dir(c0, c1, c2, nn));
    g.front = dot3(d, g.n) < 0;
    if (!g.front) g.n = -g.n;
  }
  return g;
}

// ---- sampling helpers (shaders/importanceSampling.wgsl) ---------------------------------------------
struct Onb {
  f3 u, v, w;
};
DEV Onb onb_build_from_w(f3 wdir) {  // :60-67
  Onb b;
  b.w = norm3(wdir);
  f3 a = (ptm_abs(b.w.x) > 0.9f) ? mk3(0, 1, 0) : mk3(1, 0, 0);
  b.v = norm3(cross3(b.w, a));
  b.u = cross3(b.w, b.v);
  return b;
}
DEV f3 onb_get_local(const Onb& b, f3 a) { return b.u * a.x + b.v * a.y + b.w * a.z; }  // :69-71
DEV f3 uniform_random_in_unit_sphere(uint32_t& rng) {  // :7-16
  float phi = rand2D(rng) * 2.0f * kPi;
  float theta = ptm_acos(2.0f * rand2D(rng) - 1.0f);
  float st = ptm_sin(theta);
  float x = st * ptm_cos(phi);
  float y = st * ptm_sin(phi);
  float z = ptm_cos(theta);
  return norm3(mk3(x, y, z));
}
DEV f3 cosine_sampling_wrt_Z(uint32_t& rng) {  // :35-45
  float r1 = rand2D(rng);
  float r2 = rand2D(rng);
  float phi = kTwoPi * r1;
  float sr = sqrt_exact(r2);
  return mk3(ptm_cos(phi) * sr, ptm_sin(phi) * sr, sqrt_exact(1.0f - r2));
}
DEV float reflectance(float cosine, float ref_idx) {  // :1-5
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  return r0 + (1.0f - r0) * ptm_pow((1.0f - cosine), 5.0f);
}

struct QuadL {
  f3 Q, u, v, normal, w;
  float D;
};
DEV QuadL load_light(const DevScene& S) {
  QuadL q;
  if (S.light_quad < 0) {  // `lights` stays zero-initialised
    q.Q = q.u = q.v = q.normal = q.w = mk3(0, 0, 0);
    q.D = 0;
    return q;
  }
  const float4* p = S.quads + 5 * S.light_quad;
  q.Q = mk3(p[0]);
  q.u = mk3(p[1]);
  q.v = mk3(p[2]);
  float4 n = p[3];
  q.normal = mk3(n);
  q.D = n.w;
  q.w = mk3(p[4]);
  return q;
}
// shaders/importanceSampling.wgsl:88-125 (quad argument == the global `lights` at the only call site)
DEV float light_pdf(const QuadL& L, f3 ro, f3 rd) {
  if (dot3(rd, L.normal) > 0) return kMinFloat;
  float denom = dot3(L.normal, rd);
  if (ptm_abs(denom) < 1e-8f) return kMinFloat;
  float t = (L.D - dot3(L.normal, ro)) / denom;
  if (t <= 0.001f || t >= kMaxFloat) return kMinFloat;
  f3 isect = ro + t * rd;
  f3 ph = isect - L.Q;
  float alpha = dot3(L.w, cross3(ph, L.v));
  float beta = dot3(L.w, cross3(L.u, ph));
  if (alpha < 0 || 1 < alpha || beta < 0 || 1 < beta) return kMinFloat;
  f3 hn = L.normal;
  bool front = dot3(rd, L.normal) < 0;
  if (!front) hn = -hn;
  float dl = len3(rd);
  float distance_squared = t * t * dl * dl;
  float cosine = ptm_abs(dot3(rd, hn) / dl);
  return distance_squared / (cosine * len3(cross3(L.u, L.v)));
}

struct Material {
  f3 color, spec, emission;
  float specularStrength, roughness, eta, type;
};
DEV Material load_material(const DevScene& S, int id) {
  const float4* m = S.mats + 4 * id;
  float4 a = m[0], b = m[1], c = m[2], e = m[3];
  Material r;
  r.color = mk3(a);
  r.spec = mk3(b);
  r.emission = mk3(c);
  r.specularStrength = c.w;
  r.roughness = e.x;
  r.eta = e.y;
  r.type = e.z;
  return r;
}

// shaders/scatterRay.wgsl:2-95.  `bin` is the wave-uniform material class the path was sorted into.
// Returns the scattered direction (origin is hitRec.p); sets doSpecular, skip_pdf and, for
// LAMBERTIAN, the ONB w axis that onb_lambertian_scattering_pdf reads afterwards.
DEV f3 material_scatter(int bin, const Material& m, const HitGeom& g, f3 din, uint32_t& rng, float& doSpecular, bool& skip_pdf, f3& unit_w) {
  doSpecular = 0.0f;
  skip_pdf = true;
  if (bin == BIN_LAMBERTIAN) {
    LT(LT_MS_LAMBERT);
    Onb b = onb_build_from_w(g.n);
    unit_w = b.w;
    f3 diffuse = cosine_sampling_wrt_Z(rng);
    diffuse = norm3(onb_get_local(b, diffuse));
    doSpecular = (rand2D(rng) < m.specularStrength) ? 1.0f : 0.0f;
    f3 specular = reflect3(din, g.n);
    specular = norm3(mix3(specular, diffuse, m.roughness));
    skip_pdf = (doSpecular == 1.0f);
    return norm3(mix3(diffuse, specular, doSpecular));
  } else if (bin == BIN_MIRROR) {
    LT(LT_MS_MIRROR);
    f3 reflected = reflect3(din, g.n);
    return norm3(reflected + m.roughness * uniform_random_in_unit_sphere(rng));
  } else if (bin == BIN_GLASS) {
    LT(LT_MS_GLASS);
    float ir = m.eta;
    if (g.front) ir = rcp_exact(ir);
    f3 ud = norm3(din);
    float cos_theta = ptm_min(dot3(-ud, g.n), 1.0f);
    float sin_theta = sqrt_exact(1.0f - cos_theta * cos_theta);
    f3 dir;
    if (ir * sin_theta > 1.0f || reflectance(cos_theta, ir) > rand2D(rng)) {
      dir = reflect3(ud, g.n);
    } else {
      dir = refract3(ud, g.n, ir);
    }
    return norm3(dir);
  } else if (bin == BIN_ISOTROPIC) {
    LT(LT_MS_ISO);
    float gg = m.specularStrength;
    float cos_hg = (1.0f + gg * gg - ptm_pow(((1.0f - gg * gg) / (1.0f - gg + 2.0f * gg * rand2D(rng))), 2.0f)) / (2.0f * gg);
    float sin_hg = sqrt_exact(1.0f - cos_hg * cos_hg);
    float phi = kTwoPi * rand2D(rng);
    f3 hg = mk3(sin_hg * ptm_cos(phi), sin_hg * ptm_sin(phi), cos_hg);
    Onb b = onb_build_from_w(din);
    unit_w = b.w;
    return norm3(onb_get_local(b, hg));
  }
  // unknown material_type: `scattered` stays Ray(0,0) (scatterRay.wgsl:4); the caller substitutes a
  // zero origin too.  skip_pdf would be stale private state in the shader: IS mode rejects such scenes.
  skip_pdf = false;
  return mk3(0, 0, 0);
}

DEV int lane_id() { return (int)(threadIdx.x & 63); }
DEV uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// One atomicAdd per wave for all lanes with pred set; returns each such lane's slot.
DEV uint32_t wave_append(uint32_t* counter, bool pred) {
  uint64_t m = __ballot(pred);
  if (m == 0) return 0;
  int leader = __ffsll((unsigned long long)m) - 1;
  uint32_t base = 0;
  if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + lanes_below(m);
}

}  // namespace ptmi
