/* tad_detmath.h — deterministic FP64 log / exp / expm1 / log1p / frexp for the ARIMA detector.
 *
 * Why: calculate_arima (plugins/anomaly-detection/anomaly_detection.py:215-264) minimises a likelihood with
 * L-BFGS-B on forward-difference gradients and stops at factr = 1e7; a 1-ulp difference in ONE transcendental
 * (device ocml `log` vs host glibc `log`) moves the optimiser's end point by up to 1e-4 relative on flat
 * likelihoods.  So the engine does not call any libm transcendental on this path: these functions use only
 * IEEE-754 +, -, *, / and integer bit operations in a FIXED order (compile with -ffp-contract=off), which makes
 * their results identical bit for bit on gfx950 and on any host CPU.
 *
 * This ONE source is used by the product (tad_arima.hip, device code) and by the checker (oracle/arima_exact.c,
 * plain C on the host) — that is its purpose.  Its accuracy against the correctly rounded
 * functions is pinned separately by tests/test_detmath.py (log, exp < 1 ulp; expm1, log1p < 4 ulp).
 *
 * Algorithms (published, classical): log — argument reduction to [sqrt(1/2), sqrt(2)) and the degree-14
 * minimax polynomial in s = f / (2 + f) (Sun fdlibm constants); exp — Cody-Waite reduction by ln2 in two
 * pieces and the Remez rational form R(r^2); expm1 — Kahan's exp-and-log correction; log1p — the
 * (1 + u) correction term of HP-15C fame.  Plain C99 / C++ / HIP, no includes beyond <stdint.h>.
 */
#ifndef TAD_DETMATH_H_
#define TAD_DETMATH_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define TAD_DM_FN __host__ __device__ static inline
#else
#define TAD_DM_FN static inline
#endif

typedef union { double d; uint64_t u; } tad_dm_bits;

TAD_DM_FN uint64_t tad_dm_u64(double x) { tad_dm_bits b; b.d = x; return b.u; }
TAD_DM_FN double tad_dm_f64(uint64_t u) { tad_dm_bits b; b.u = u; return b.d; }

#define TAD_DM_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000: 32 significant bits */
#define TAD_DM_LN2_LO 1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */
#define TAD_DM_INV_LN2 1.44269504088896338700e+00
#define TAD_DM_LN2 6.931471805599453094e-01

/* frexp with the semantics of the gfx950 instructions v_frexp_mant_f64 / v_frexp_exp_i32_f64 (= C frexp, except that
 * infinities and NaNs come back unchanged with e = 0): x = m * 2^e, 0.5 <= |m| < 1, sign kept; +-0 -> (+-0, 0);
 * subnormals are normalised first.  The device code may therefore use the two instructions instead of this function. */
TAD_DM_FN double tad_det_frexp(double x, int *e) {
  uint64_t u = tad_dm_u64(x);
  int be = (int)((u >> 52) & 0x7ff), adj = 0;
  if (be == 0x7ff || (u << 1) == 0) { *e = 0; return x; }
  if (be == 0) {                                    /* subnormal: scale by 2^54 (exact) */
    tad_dm_bits b; b.u = u; b.d *= 18014398509481984.0; u = b.u;
    be = (int)((u >> 52) & 0x7ff);
    adj = -54;
  }
  *e = be - 1022 + adj;
  return tad_dm_f64((u & 0x800fffffffffffffULL) | 0x3fe0000000000000ULL);
}

/* 2^k for -1022 <= k <= 1023 */
TAD_DM_FN double tad_dm_pow2(int k) { return tad_dm_f64((uint64_t)(0x3ff + k) << 52); }

/* y * 2^k, y of moderate magnitude (|y| in [2^-2, 2^2]); at most two multiplications, each by a power of two */
TAD_DM_FN double tad_dm_scale(double y, int k) {
  if (k > 1023) {
    y *= tad_dm_pow2(1023);
    k -= 1023;
    if (k > 1023) k = 1023;
  } else if (k < -1022) {
    y *= tad_dm_pow2(-969); /* 2^-1022 * 2^53: keeps the intermediate normal, one rounding at the end */
    k += 969;
    if (k < -1022) k = -1022;
  }
  return y * tad_dm_pow2(k);
}

TAD_DM_FN double tad_det_log(double x) {
  uint64_t u = tad_dm_u64(x);
  int k = 0;
  if ((u >> 52) == 0 || (u >> 63)) {               /* +-0, subnormal, or negative */
    if ((u << 1) == 0) return tad_dm_f64(0xfff0000000000000ULL);   /* log(+-0) = -inf */
    if (u >> 63) return tad_dm_f64(0x7ff8000000000000ULL);         /* log(negative) = NaN */
    x *= 18014398509481984.0;                       /* 2^54: subnormal -> normal */
    u = tad_dm_u64(x);
    k = -54;
  } else if ((u >> 52) >= 0x7ff) {
    return x + x;                                   /* +inf -> +inf, NaN -> NaN */
  }
  if (u == 0x3ff0000000000000ULL) return 0.0;
  /* m in [sqrt(2)/2, sqrt(2)) */
  uint32_t hx = (uint32_t)(u >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  const double m = tad_dm_f64(((uint64_t)hx << 32) | (u & 0xffffffffULL));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double dk = (double)k;
  return s * (hfsq + R) + dk * TAD_DM_LN2_LO - hfsq + f + dk * TAD_DM_LN2_HI;
}

TAD_DM_FN double tad_det_exp(double x) {
  const uint64_t u = tad_dm_u64(x);
  const uint32_t ax = (uint32_t)(u >> 32) & 0x7fffffffu;
  const int sign = (int)(u >> 63);
  if (ax >= 0x4086232bu) {                          /* |x| >= 708.39 or NaN */
    if ((u << 1) > 0xffe0000000000000ULL) return x + x;         /* NaN */
    if (x > 709.782712893383973096) return tad_dm_f64(0x7ff0000000000000ULL);
    if (x < -745.13321910194110842) return 0.0;
  }
  double hi, lo;
  int k;
  if (ax > 0x3fd62e42u) {                           /* |x| > 0.5 ln2 */
    if (ax >= 0x3ff0a2b2u) k = (int)(TAD_DM_INV_LN2 * x + (sign ? -0.5 : 0.5));
    else k = 1 - sign - sign;
    hi = x - (double)k * TAD_DM_LN2_HI;             /* k * ln2_hi is exact */
    lo = (double)k * TAD_DM_LN2_LO;
    x = hi - lo;
  } else if (ax > 0x3e300000u) {                    /* |x| > 2^-28 */
    k = 0; hi = x; lo = 0.0;
  } else {
    return 1.0 + x;
  }
  const double xx = x * x;
  const double c = x - xx * (1.66666666666666019037e-01 + xx * (-2.77777777770155933842e-03 + xx * (6.61375632143793436117e-05 +
                   xx * (-1.65339022054652515390e-06 + xx * 4.13813679705723846039e-08))));
  const double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
  return k == 0 ? y : tad_dm_scale(y, k);
}

/* exp(x) - 1 (Kahan): with e = exp(x), (e - 1) * x / log(e) cancels the rounding of e */
TAD_DM_FN double tad_det_expm1(double x) {
  const double e = tad_det_exp(x);
  if (e == 1.0) return x;
  const double em1 = e - 1.0;
  if (em1 == -1.0) return -1.0;
  if (!(e < 1.7976931348623157e308)) return e;      /* overflow / NaN */
  return em1 * x / tad_det_log(e);
}

/* log(1 + x): log(w) corrected by the rounding error of w = 1 + x */
TAD_DM_FN double tad_det_log1p(double x) {
  const double w = 1.0 + x;
  if (w == 1.0) return x;
  if (!(w > 0.0) || !(w < 1.7976931348623157e308)) return tad_det_log(w);   /* -inf / NaN / +inf */
  return tad_det_log(w) - ((w - 1.0) - x) / w;
}

#endif /* TAD_DETMATH_H_ */
