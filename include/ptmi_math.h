/*
 * ptmi_math.h — deterministic f32 elementary functions, identical bits on x86-64 and gfx950.
 *
 * WGSL leaves the accuracy of sin/cos/acos/log/pow implementation-defined
 * (reference call sites: shaders/importanceSampling.wgsl:7-16,35-45,
 * shaders/scatterRay.wgsl:80-84, shaders/common.wgsl:134).  This header pins ONE
 * evaluation of each: range reduction + polynomial, written only with IEEE-754 correctly rounded
 * operations (+ - * / sqrt and explicit fmaf) and integer bit manipulation, so the host (oracle,
 * compiled with gcc) and the device (HIP kernels, compiled with hipcc) produce the same bits.
 *
 * Rules for every translation unit that includes this file:
 *   - compile with -ffp-contract=off and without fast-math (the only fused operations are the
 *     explicit ptm_fma calls below);
 *   - f32 denormals enabled (default on x86-64 SSE and on gfx950);
 *   - host builds want -mfma so that fmaf() is one instruction (libm's fmaf is also exact, only slow).
 *
 * Accuracy (checked by tests/test_math.py against float64 libm): sin/cos <= 2 ulp on [-64, 64],
 * acos <= 2 ulp, log/log2/exp2 <= 1 ulp; pow(x,y) = exp2(y*log2(x)) by definition (WGSL), so its error
 * grows with |y*log2(x)| (about 65 ulp for x^5 near 1e-6, where the value itself is ~1e-30).
 */
#ifndef PTMI_MATH_H
#define PTMI_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PTM_HD __host__ __device__ inline
#else
#define PTM_HD static inline
#endif

PTM_HD float ptm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PTM_HD float ptm_sqrt(float x) { return __builtin_sqrtf(x); }
PTM_HD float ptm_abs(float x) { return __builtin_fabsf(x); }

PTM_HD uint32_t ptm_f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
PTM_HD float ptm_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* min/max: WGSL "if one operand is a NaN, the other is returned"; additionally -0 < +0.  That is
 * exactly what gfx950's v_min_f32 / v_max_f32 (and the fused v_min3/v_max3) compute, so the device
 * uses the hardware instruction and the host spells the same rule out.  The rule is associative and
 * commutative, hence independent of how a compiler nests a 3-way min/max. */
#if defined(__HIP_DEVICE_COMPILE__)
PTM_HD float ptm_min(float a, float b) { return __builtin_fminf(a, b); }
PTM_HD float ptm_max(float a, float b) { return __builtin_fmaxf(a, b); }
#else
PTM_HD float ptm_min(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (ptm_f2u(a) >> 31) ? a : b; /* min(-0,+0) = -0 */
  return (a < b) ? a : b;
}
PTM_HD float ptm_max(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (ptm_f2u(a) >> 31) ? b : a; /* max(-0,+0) = +0 */
  return (a < b) ? b : a;
}
#endif

/* round to nearest integer (ties to even) for |x| < 2^22, by the add-magic trick: only + and - */
PTM_HD float ptm_rint_small(float x) {
  const float magic = 12582912.0f; /* 1.5 * 2^23 */
  float t = x + magic;             /* not foldable without -fassociative-math (never enabled here) */
  return t - magic;
}

/* ---- sin / cos ------------------------------------------------------------------------------ */
/* quadrant reduction r = x - k*pi/2 with a 3-part pi/2 and fma; |x| up to ~1e4 keeps <= 2 ulp,
 * beyond 2^22 the quadrant count is not exact any more and the result is merely bounded. */
PTM_HD float ptm__sin_poly(float r) {
  float z = r * r;
  float p = ptm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = ptm_fma(p, z, -1.6666654611e-1f);
  return ptm_fma(r * z, p, r);
}
PTM_HD float ptm__cos_poly(float r) {
  float z = r * r;
  float p = ptm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = ptm_fma(p, z, 4.166664568298827e-2f);
  float q = ptm_fma(-0.5f, z, 1.0f);
  return ptm_fma(z * z, p, q);
}
PTM_HD void ptm__reduce_pio2(float x, float* r, int* quad) {
  const float two_over_pi = 0.636619772367581343f;
  const float pio2_hi = 1.57079637050628662109375f;    /* f32(pi/2)            */
  const float pio2_mid = -4.37113900018624283e-8f;     /* f32(pi/2 - hi)       */
  const float pio2_lo = -1.71512449801389655e-15f;     /* f32(pi/2 - hi - mid) */
  float kf;
  if (ptm_abs(x) < 4194304.0f) {
    const float magic = 12582912.0f;
    float t = ptm_fma(x, two_over_pi, magic);
    kf = t - magic;
  } else {
    kf = 0.0f; /* out of the supported domain: no reduction (NaN/inf propagate below) */
  }
  float rr = ptm_fma(-kf, pio2_hi, x);
  rr = ptm_fma(-kf, pio2_mid, rr);
  rr = ptm_fma(-kf, pio2_lo, rr);
  *r = rr;
  *quad = (int)kf & 3;
}
PTM_HD float ptm_sin(float x) {
  float r;
  int q;
  ptm__reduce_pio2(x, &r, &q);
  float s = ptm__sin_poly(r);
  float c = ptm__cos_poly(r);
  float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
PTM_HD float ptm_cos(float x) {
  float r;
  int q;
  ptm__reduce_pio2(x, &r, &q);
  float s = ptm__sin_poly(r);
  float c = ptm__cos_poly(r);
  float v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}

/* ---- acos ----------------------------------------------------------------------------------- */
PTM_HD float ptm__asin_poly(float x) { /* asin(x) for |x| <= 0.5 */
  float z = x * x;
  float p = ptm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
  p = ptm_fma(p, z, 4.5470025998e-2f);
  p = ptm_fma(p, z, 7.4953002686e-2f);
  p = ptm_fma(p, z, 1.6666752422e-1f);
  return ptm_fma(x * z, p, x);
}
PTM_HD float ptm_acos(float x) {
  const float pio2_hi = 1.57079637050628662109375f;
  const float pio2_lo = -4.37113900018624283e-8f;
  const float pi_hi = 3.1415927410125732421875f;
  const float pi_lo = -8.74227800037248566e-8f;
  if (x > 0.5f) { /* acos(x) = 2 asin(sqrt((1-x)/2)) ; x > 1 gives sqrt(<0) = NaN */
    float s = ptm_sqrt(0.5f * (1.0f - x));
    return 2.0f * ptm__asin_poly(s);
  }
  if (x < -0.5f) { /* acos(x) = pi - 2 asin(sqrt((1+x)/2)) */
    float s = ptm_sqrt(0.5f * (1.0f + x));
    float a = 2.0f * ptm__asin_poly(s);
    return (pi_hi - a) + pi_lo;
  }
  /* |x| <= 0.5 (and NaN): pi/2 - asin(x) */
  float a = ptm__asin_poly(x);
  return (pio2_hi - a) + pio2_lo;
}

/* ---- log, log2 ------------------------------------------------------------------------------ */
/* decompose x = m * 2^e, m in [sqrt(1/2), sqrt(2)); returns f = m - 1 and e. x must be finite > 0. */
PTM_HD float ptm__log_reduce(float x, int* e_out) {
  uint32_t u = ptm_f2u(x);
  int e = 0;
  if (u < 0x00800000u) { /* denormal: scale by 2^23 */
    x = x * 8388608.0f;
    u = ptm_f2u(x);
    e = -23;
  }
  e += (int)(u >> 23) - 126;
  float m = ptm_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
  float f;
  if (m < 0.707106781186547524f) {
    e -= 1;
    f = (m + m) - 1.0f;
  } else {
    f = m - 1.0f;
  }
  *e_out = e;
  return f;
}
PTM_HD float ptm__log1p_poly(float f) { /* log(1+f) - f + f*f/2, f in [sqrt(.5)-1, sqrt(2)-1] */
  float z = f * f;
  float p = ptm_fma(7.0376836292e-2f, f, -1.1514610310e-1f);
  p = ptm_fma(p, f, 1.1676998740e-1f);
  p = ptm_fma(p, f, -1.2420140846e-1f);
  p = ptm_fma(p, f, 1.4249322787e-1f);
  p = ptm_fma(p, f, -1.6668057665e-1f);
  p = ptm_fma(p, f, 2.0000714765e-1f);
  p = ptm_fma(p, f, -2.4999993993e-1f);
  p = ptm_fma(p, f, 3.3333331174e-1f);
  return (f * z) * p;
}
PTM_HD float ptm_log(float x) {
  if (!(x > 0.0f)) { /* 0 -> -inf, negative / NaN -> NaN */
    if (x == 0.0f) return -INFINITY;
    return NAN;
  }
  if (x == INFINITY) return x;
  int e;
  float f = ptm__log_reduce(x, &e);
  float z = f * f;
  float y = ptm__log1p_poly(f);
  float fe = (float)e;
  y = ptm_fma(fe, -2.12194440e-4f, y);
  y = ptm_fma(-0.5f, z, y);
  float r = f + y;
  return ptm_fma(fe, 0.693359375f, r);
}
PTM_HD float ptm_log2(float x) {
  if (!(x > 0.0f)) {
    if (x == 0.0f) return -INFINITY;
    return NAN;
  }
  if (x == INFINITY) return x;
  int e;
  float f = ptm__log_reduce(x, &e);
  float z = f * f;
  float y = ptm__log1p_poly(f);
  y = ptm_fma(-0.5f, z, y); /* log(1+f) = f + y */
  /* log2(1+f) = (f + y) * log2(e), with log2(e) split hi + lo to keep the product accurate */
  const float l2e_hi = 1.44269502162933349609375f;
  const float l2e_lo = 1.92596303e-8f;
  float r = y * l2e_lo;
  r = ptm_fma(f, l2e_lo, r);
  r = ptm_fma(y, l2e_hi, r);
  r = ptm_fma(f, l2e_hi, r);
  return r + (float)e;
}

/* ---- exp2, pow ------------------------------------------------------------------------------ */
PTM_HD float ptm_exp2(float x) {
  if (!(x == x)) return x;                 /* NaN */
  if (x >= 128.0f) return INFINITY;
  if (x < -150.0f) return 0.0f;
  float k = ptm_rint_small(x);
  float r = x - k; /* [-0.5, 0.5], exact */
  float p = ptm_fma(1.535336188319500e-4f, r, 1.339887440266574e-3f);
  p = ptm_fma(p, r, 9.618437357674640e-3f);
  p = ptm_fma(p, r, 5.550332471162809e-2f);
  p = ptm_fma(p, r, 2.402264791363012e-1f);
  p = ptm_fma(p, r, 6.931472028550421e-1f);
  float v = ptm_fma(p, r, 1.0f); /* 2^r in [0.707, 1.415] */
  int ki = (int)k;
  /* scale by 2^ki in two steps so that denormal results round once at the end */
  if (ki > 127) {
    v = v * 2.0f;
    ki -= 1;
  }
  if (ki >= -126) return v * ptm_u2f((uint32_t)(ki + 127) << 23);
  v = v * ptm_u2f((uint32_t)(ki + 24 + 127) << 23);
  return v * 5.9604644775390625e-8f; /* 2^-24 */
}
/* WGSL: pow(e1,e2) inherits from exp2(e2 * log2(e1)); that is the definition used here. */
PTM_HD float ptm_pow(float x, float y) { return ptm_exp2(y * ptm_log2(x)); }

#endif /* PTMI_MATH_H */
