// tad_arima.hip — ARIMA detector (anomaly_detection.py:215-309) on gfx950, FP64 throughout.
//
//   y, lam = scipy.stats.boxcox(x)                 (:239)   -> k_arima_prep  (one lane = one key)
//   preds[0:3] = y[0:3]                            (:241)
//   for t in 3..n-1: ARIMA(y[:t], (1,1,1)).fit().forecast()[0]   (:246-253) -> k_arima_fit (one lane = one fit)
//   algoCalc = inv_boxcox(preds, lam)              (:256)
//   anomaly  = |x - algoCalc| > stddev             (:306-307)
//   n <= 3, x <= 0 or constant x -> None -> the key yields no rows  (:232-234, :260-264, :284-287)
//
// The fit restates statsmodels 0.14 SARIMAX/ARIMA (not in /root/reference; pinned in
// plugins/anomaly-detection/requirements.txt:3):  state space Z=[1 1 0], T=[[1 1 0],[0 phi 1],[0 0 0]],
// R=[0 1 theta]', approximate-diffuse (1e6) + stationary initial covariance, loglikelihood_burn = 1,
// covariance frozen once ||P_t - P_{t+1}||_F^2 < 1e-19 (arithmetic contract: see kfc_*); start parameters by conditional sum of squares on
// diff(y) with numpy-pinv semantics; phi = -u/sqrt(1+u^2), theta = u/sqrt(1+u^2) (statsmodels' signs), sigma2 = u^2;
// objective -loglike/nobs minimised by L-BFGS-B (m = 10, factr = 1e7, pgtol = 1e-5, maxiter = 50, maxls = 20)
// with forward-difference gradients (h = 1e-5) and the More'-Thuente line search (ftol 1e-3, gtol 0.9,
// xtol 0.1) — the unconstrained path of L-BFGS-B 3.0, where the subspace step equals the two-loop
// L-BFGS direction with H0 = (s'y / y'y) I.
//
// GPU mapping: there is no dense contraction here (3 parameters, 3 states): the work is ~20 optimiser cycles of four likelihood
// evaluations per fit, each a sequential Kalman recursion over the history.  Parallelism is across the P - 3K independent
// fits.  A wavefront takes a chunk of keys at ONE series position (every lane's Kalman loop has the same length) and its
// lanes pull the next key as soon as their fit has converged; the four evaluations of an optimiser cycle run as four
// interleaved recursions per lane; the optimiser itself is a per-lane state machine stepped once per cycle (k_arima_fit).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tad_internal.h"
#include "tad_detmath.h"

// helper functions are host+device so that tools/arima_twin.cpp can run the same source on a machine without a GPU
// (tools/arima_twin_check.py: host instantiation == oracle/arima_exact.c bit for bit, before any GPU time is spent)
#define TAD_HD __host__ __device__

namespace tad {

static constexpr double kDiffuse = 1e6;
static constexpr double kConvTol = 1e-19;                   // statsmodels: KalmanFilter.tolerance on ||P_t - P_t+1||_F^2
static constexpr double kConvTolAbs = 3.1622776601683794e-10;  // its square root: the test on |p_t - p_t+1| (only p11 evolves)
static constexpr double kLog2Pi = 1.8378770664093453;  // log(2*pi)
static constexpr double kEpsMch = 2.220446049250313e-16;
static constexpr int kLbfgsM = 10;

TAD_HD inline bool tad_finite(double v) { return fabs(v) <= 1.7976931348623157e308; }  // false for NaN and +-inf

struct ArimaWs {
  double *xs;      // [T][K] compacted values (as double), position-major
  double *lx;      // [T][K] log(x)
  double *ys;      // [T][K] Box-Cox transformed
  double *ysk;     // [K][Tpad] the same, KEY-major (Tpad = T rounded up to 8): the fit kernel's lanes hold arbitrary keys
  double *u0[3];   // [T][K] start parameters of the fit on y[:p] (k_arima_start)
  uint32_t *tpos;  // [T][K] bucket of the p-th point
  double *lam;     // [K]
  uint8_t *state;  // [K] 0 = ok, 1 = no result
  double *hist;    // [wavefronts of k_arima_fit][kHistDoubles][64] every lane's L-BFGS (s, y) history (struct Lbfgs)
  unsigned long long *cursor;   // [T] next key of series position p not yet handed to a lane (k_arima_fit)
  unsigned int *yielded;        // [1] set by a wavefront of k_arima_fit that stopped because the engine asked it to (cooperative yield)
  unsigned int *saved;          // [wavefronts of k_arima_fit] 1: the wavefront suspended its fits in `save` and resumes them at the next launch
  double *save;                 // [wavefronts of k_arima_fit][kSaveDoubles][64] the suspended lanes' optimiser state (the history block stays where it is)
  uint32_t Tpad;
};

// ------------------------------------------------------------------------------------------------
// Box-Cox: llf of scipy 1.10.1 (_morestats.boxcox_llf), bracket() + Brent.optimize() of scipy.optimize
// ------------------------------------------------------------------------------------------------
// f(i, p[i * stride]) for i = 0 .. n-1 in order, the loads of the NEXT eight elements issued before the current eight are worked on.
// A lane of k_arima_prep walks its key's column with a stride of K doubles and there are ~1.5 wavefronts per SIMD at C3 (one lane
// per key, 1e5 keys), so a load placed inside the loop costs a full HBM round trip per element: that, not the arithmetic, was the time.
template <typename F>
TAD_HD void stream_col8(const double *__restrict__ p, size_t stride, uint32_t n, F f) {
  double a[8], b[8];
  uint32_t i = 0;
  if (n >= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = p[(size_t)j * stride];
  }
  for (; i + 16 <= n; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = p[(size_t)(i + 8 + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) f(i + j, a[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = b[j];
  }
  if (i + 8 <= n) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f(i + j, a[j]);
    i += 8;
  }
  for (; i < n; ++i) f(i, p[(size_t)i * stride]);
}

TAD_HD double bc_neg_llf(double lmb, const double *xs, const double *__restrict__ lx, double *__restrict__ tmp, size_t stride, uint32_t n,
                         double sumlog) {
  // variance of x**lmb / lmb (population), two passes like numpy.var.  The second pass reads the terms the first one computed back
  // from `tmp` (a free column of the workspace, same stride) instead of evaluating exp and the division again: the same doubles in
  // the same order.
  double mean = 0.0, s = 0.0;
  if (lmb == 0.0) {
    stream_col8(lx, stride, n, [&](uint32_t, double l) { mean += l; });
    mean /= (double)n;
    stream_col8(lx, stride, n, [&](uint32_t, double l) { const double d = l - mean; s += d * d; });
    return -((lmb - 1.0) * sumlog - (double)n / 2.0 * tad_det_log(s / (double)n));
  }
  stream_col8(lx, stride, n, [&](uint32_t i, double l) { const double v = tad_det_exp(lmb * l) / lmb; tmp[(size_t)i * stride] = v; mean += v; });
  mean /= (double)n;
  stream_col8(tmp, stride, n, [&](uint32_t, double v) { const double d = v - mean; s += d * d; });
  return -((lmb - 1.0) * sumlog - (double)n / 2.0 * tad_det_log(s / (double)n));
}

// returns false when no valid bracket / not finite (scipy raises -> calculate_arima returns None)
TAD_HD bool bc_mle_lambda(const double *xs, const double *lx, double *tmp, size_t stride, uint32_t n, double sumlog, double *lam_out) {
#define BCF(l) bc_neg_llf((l), xs, lx, tmp, stride, n, sumlog)
  const double gold = 1.618034, verysmall = 1e-21, grow_limit = 110.0;
  double xa = -2.0, xb = 2.0;
  double fa = BCF(xa), fb = BCF(xb);
  if (fa < fb) { double t = xa; xa = xb; xb = t; t = fa; fa = fb; fb = t; }
  double xc = xb + gold * (xb - xa);
  double fc = BCF(xc);
  int iter = 0;
  while (fc < fb) {
    const double tmp1 = (xb - xa) * (fb - fc);
    const double tmp2 = (xb - xc) * (fb - fa);
    const double val = tmp2 - tmp1;
    const double denom = fabs(val) < verysmall ? 2.0 * verysmall : 2.0 * val;
    double w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
    const double wlim = xb + grow_limit * (xc - xb);
    if (iter > 1000) return false;
    iter++;
    double fw;
    if ((w - xc) * (xb - w) > 0.0) {
      fw = BCF(w);
      if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
      else if (fw > fb) { xc = w; fc = fw; break; }
      w = xc + gold * (xc - xb);
      fw = BCF(w);
    } else if ((w - wlim) * (wlim - xc) >= 0.0) {
      w = wlim;
      fw = BCF(w);
    } else if ((w - wlim) * (xc - w) > 0.0) {
      fw = BCF(w);
      if (fw < fc) {
        xb = xc; xc = w; w = xc + gold * (xc - xb);
        fb = fc; fc = fw; fw = BCF(w);
      }
    } else {
      w = xc + gold * (xc - xb);
      fw = BCF(w);
    }
    xa = xb; xb = xc; xc = w;
    fa = fb; fb = fc; fc = fw;
  }
  const bool cond1 = (fb < fc && fb <= fa) || (fb < fa && fb <= fc);
  const bool cond2 = (xa < xb && xb < xc) || (xc < xb && xb < xa);
  const bool cond3 = tad_finite(xa) && tad_finite(xb) && tad_finite(xc);
  if (!(cond1 && cond2 && cond3)) return false;
  // Brent
  const double tol = 1.48e-8, mintol = 1.0e-11, cg = 0.3819660;
  double x = xb, w = xb, v = xb, fx = fb, fw = fb, fv = fb;
  double a = xa < xc ? xa : xc, b = xa < xc ? xc : xa;
  double deltax = 0.0, rat = 0.0;
  for (int it = 0; it < 500; ++it) {
    const double tol1 = tol * fabs(x) + mintol, tol2 = 2.0 * tol1, xmid = 0.5 * (a + b);
    if (fabs(x - xmid) < (tol2 - 0.5 * (b - a))) break;
    if (fabs(deltax) <= tol1) {
      deltax = x >= xmid ? a - x : b - x;
      rat = cg * deltax;
    } else {
      double tmp1 = (x - w) * (fx - fv);
      double tmp2 = (x - v) * (fx - fw);
      double p = (x - v) * tmp2 - (x - w) * tmp1;
      tmp2 = 2.0 * (tmp2 - tmp1);
      if (tmp2 > 0.0) p = -p;
      tmp2 = fabs(tmp2);
      const double dx_temp = deltax;
      deltax = rat;
      if (p > tmp2 * (a - x) && p < tmp2 * (b - x) && fabs(p) < fabs(0.5 * tmp2 * dx_temp)) {
        rat = p * 1.0 / tmp2;
        const double u = x + rat;
        if ((u - a) < tol2 || (b - u) < tol2) rat = xmid - x >= 0 ? tol1 : -tol1;
      } else {
        deltax = x >= xmid ? a - x : b - x;
        rat = cg * deltax;
      }
    }
    const double u = fabs(rat) < tol1 ? (rat >= 0 ? x + tol1 : x - tol1) : x + rat;
    const double fu = BCF(u);
    if (fu > fx) {
      if (u < x) a = u; else b = u;
      if (fu <= fw || w == x) { v = w; w = u; fv = fw; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    } else {
      if (u >= x) a = x; else b = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
    }
  }
#undef BCF
  *lam_out = x;
  return tad_finite(x);
}

TAD_HD inline double inv_boxcox(double y, double lam) {
  return lam == 0.0 ? tad_det_exp(y) : tad_det_exp(tad_det_log1p(lam * y) / lam);  // scipy.special.inv_boxcox
}

// one lane = one key: compact the series, Box-Cox it, first three predictions
__global__ __launch_bounds__(256) void k_arima_prep(Grid g, ArimaWs ws, const double *__restrict__ sigma,
                                                    double *__restrict__ calc, DevCounters *ctr) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned noresult = 0;
  if (k < g.K) {
    const size_t st = g.K;
    double *xs = ws.xs + k, *lx = ws.lx + k, *ys = ws.ys + k;
    uint32_t *tp = ws.tpos + k;
    uint32_t n = 0;
    bool nonpos = false, allsame = true;
    double x0 = 0.0, sumlog = 0.0;
    auto visit = [&](uint64_t t, uint8_t fl, unsigned long long v) {
      if (fl & FLAG_PRESENT) {
        const double x = (double)v;
        if (n == 0) x0 = x;
        if (x != x0) allsame = false;
        if (!(x > 0.0)) nonpos = true;
        const double l = tad_det_log(x);
        xs[(size_t)n * st] = x;
        lx[(size_t)n * st] = l;
        tp[(size_t)n * st] = (uint32_t)t;
        sumlog += l;
        n++;
      }
    };
    uint64_t t = 0;
    for (; t + 8 <= g.T; t += 8) {   // the cells of eight buckets in flight at once (see stream_col8)
      uint8_t fl[8];
      unsigned long long v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const uint64_t c = (t + j) * g.K + k; fl[j] = g.flag[c]; v[j] = g.val[c]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) visit(t + j, fl[j], v[j]);
    }
    for (; t < g.T; ++t) { const uint64_t c = t * g.K + k; visit(t, g.flag[c], g.val[c]); }
    bool ok = n > 3 && !nonpos && !allsame;
    double lam = 0.0;
    if (ok) ok = bc_mle_lambda(xs, lx, ys, st, n, sumlog, &lam);   // (ys: scratch for the llf's terms until the transform below fills it)
    ws.lam[k] = lam;
    ws.state[k] = ok ? 0 : 1;
    if (ok) {
      const double sg = sigma[k];
      stream_col8(lx, st, n, [&](uint32_t i, double l) {
        const double y = lam == 0.0 ? l : tad_det_expm1(lam * l) / lam;  // scipy.special.boxcox
        ys[(size_t)i * st] = y;
        ws.ysk[(size_t)k * ws.Tpad + i] = y;
        if (i < 3) {
          const uint64_t c = (uint64_t)tp[(size_t)i * st] * g.K + k;
          const double pred = inv_boxcox(y, lam);
          calc[c] = pred;
          if (fabs(xs[(size_t)i * st] - pred) > sg) g.flag[c] = FLAG_PRESENT | FLAG_ANOMALY;
        }
      });
    } else if (n > 0) {
      // calculate_arima returned None: arrays_zip/explode of a null array -> the key has no rows at all
      for (uint32_t i = 0; i < n; ++i) g.flag[(uint64_t)tp[(size_t)i * st] * g.K + k] = 0;
      noresult = 1;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) noresult += __shfl_down(noresult, d);
  if ((threadIdx.x & 63) == 0 && noresult) atomicAdd(&ctr->keys_no_result, (unsigned long long)noresult);
}

// ------------------------------------------------------------------------------------------------
// ARIMA(1,1,1) likelihood.  statsmodels' state space is Z = [1 1 0], T = [[1 1 0],[0 phi 1],[0 0 0]], R = [0 1 theta]',
// Q = sigma2, H = 0, approximate-diffuse (1e6) level + stationary ARMA block, loglikelihood_burn = 1, covariance frozen once
// ||P_t - P_t+1||_F^2 < 1e-19.  ARITHMETIC CONTRACT (the one the oracle, oracle/arima_exact.c:arima_nll4_collapsed, states
// expression for expression — IEEE double + - * / sqrt, fixed order, no FMA contraction, tad_detmath.h for log):
// H = 0 (no observation noise) puts Z' in the null space of the filtered covariance: C Z' = P Z' - P Z' (Z P Z') / F = 0.
// Row 0 of T is Z, so after EVERY update (T C T')_00 = (T C T')_01 = 0: from t = 1 on only p11 is non-zero (q12, q22
// are constants), the level is known exactly (a0_t = y_t-1) and the filter is the innovations recursion of the ARMA(1,1)
// on the differences:
//   v = (y_t - y_t-1) - a1;  F = p;  r = 1 / F;  g = q12 r;  w = r v;  a1' = phi (a1 + v) + g v;  p' = (q11 + q22) - q12 g
// (the textbook three-state form computes those zeros as differences of numbers of size 1e6 — the diffuse prior — and
// feeds ~1e-10 of rounding residue into F; round 2 ran it, round 3 measured this form 1.44x faster at C3 and made it THE
// contract).  The t = 0 step is the general update written out for p00 = 1e6, p01 = 0, a = 0; it does not depend on y
// except through a1.  sum_t log F_t is one log of the running product of the F_t (mantissa renormalised every step, the
// exponents summed as integers) plus, once p has converged, (number of converged steps) x log F.
// ------------------------------------------------------------------------------------------------
struct KfOut {
  double nll;       // -loglike / nobs
  double forecast;  // Z a_{n+1|n}
};

// prod * F renormalised to a mantissa in [0.5, 1): tad_det_frexp, on the device by the two instructions it restates
TAD_HD inline double kf_frexp(double x, int *e) {
#if defined(__HIP_DEVICE_COMPILE__)
  *e = __builtin_amdgcn_frexp_exp(x);
  return __builtin_amdgcn_frexp_mant(x);
#else
  return tad_det_frexp(x, e);
#endif
}

struct KfStateC {
  double phi, q12, qs;       // model: qs = q11 + q22
  double p, a1;              // predicted p11 (= F of the coming step; frozen once converged) and AR state
  double prod, q;            // running product of the F_t (mantissa; exponents in esum), sum of v^2 / F
  int esum;
};

TAD_HD inline void kfc_init(KfStateC &s, double u0, double u1, double u2, double *p11_out, double *r0_out) {
  const double phi = -(u0 / sqrt(1.0 + u0 * u0));   // statsmodels' convention: constrain_stationary_univariate returns -r,
  const double theta = u1 / sqrt(1.0 + u1 * u1);     // SARIMAX.transform_params negates it once more for the MA block
  const double s2 = u2 * u2;
  const double q11 = s2, q12 = s2 * theta, q22 = s2 * (theta * theta);
  const double p11 = s2 * (1.0 + theta * theta + 2.0 * phi * theta) / (1.0 - phi * phi);
  const double F0 = kDiffuse + p11, r0 = 1.0 / F0;
  const double m = kDiffuse * (p11 * r0), c12 = kDiffuse * (q12 * r0), c22 = q22 - (q12 * r0) * q12;
  s.phi = phi; s.q12 = q12; s.qs = q11 + q22;
  s.p = phi * (phi * m + c12) + (phi * c12 + c22) + q11;
  s.a1 = 0.0;
  s.prod = 1.0; s.q = 0.0;
  s.esum = 0;
  *p11_out = p11; *r0_out = r0;
}

// t = 0 (burned: no likelihood term); p11 and r0 = 1 / (1e6 + p11) from kfc_init
TAD_HD inline void kfc_first(KfStateC &s, double y0, double p11, double r0) {
  const double w0 = r0 * y0;
  s.a1 = s.phi * (p11 * w0) + s.q12 * w0;
}

// The optimiser always needs the objective at x and at the three forward-difference points together, so the contract runs
// the FOUR recursions jointly and takes their reciprocals 1 / F from ONE division (batched inversion): IEEE operations in a
// fixed order like everything else, a quarter of the divisions (a division is ~11 of the ~28 instructions of a step).
TAD_HD inline void kfc_recip4(const double (&F)[4], double (&r)[4]) {
  const double t12 = F[0] * F[1], t34 = F[2] * F[3];
  const double inv = 1.0 / (t12 * t34);
  const double i12 = inv * t34, i34 = inv * t12;
  r[0] = i12 * F[1]; r[1] = i12 * F[0]; r[2] = i34 * F[3]; r[3] = i34 * F[2];
}

// One time step t >= 1 of the four recursions, d = y_t - y_t-1.  Contract details (round 3; oracle/arima_exact.c:arima_nll4_collapsed
// states the same expressions): the three multiply-adds of a chain are FUSED (IEEE fma: q, a1, p'), the running product of
// the F_t is renormalised after every fourth step only (t & 3 == 0: an exact scaling by a power of two, so the product's bits
// are those of renormalising every step as long as four factors stay in range) and the convergence test
// ||P_t - P_t+1||_F^2 < 1e-19 is taken as |p_t - p_t+1| < sqrt(1e-19) (only p11 evolves).
// Convergence is ONE select, not a flag and a second code path: p moves only by steps of at least the tolerance.  From the
// step that detects convergence on, p stays (statsmodels reuses that step's F); its log keeps entering the running product and
// 1 / F, the gain and p' are recomputed from the frozen p, so the test keeps holding.  The first version froze F, r, g in extra
// state, stopped the product and counted the converged steps: per lane 10 % of the steps are converged ones, but a wavefront
// took the predicated path as soon as ANY chain of ANY lane had converged — 76 % of all steps at C3 ran ~120 instructions with
// five scratch reloads instead of 78 (measured on the oracle's convergence times grouped 64 by 64).  Now every step is 78 + 12
// instructions and a chain's state is 7 doubles instead of 10.
// RENORM = (t & 3) == 0: a constant in every unrolled step of the device loop (the 8-step stages start at multiples of 8).
TAD_HD inline void kfc_step4(KfStateC (&s)[4], double d, const bool RENORM) {
  double F[4], r[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) F[c] = s[c].p;
  kfc_recip4(F, r);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double v = d - s[c].a1;
    const double g = s[c].q12 * r[c];
    const double w = r[c] * v;
    s[c].q = fma(v, w, s[c].q);
    s[c].prod = s[c].prod * F[c];
    if (RENORM) { int e; s[c].prod = kf_frexp(s[c].prod, &e); s[c].esum += e; }
    s[c].a1 = fma(g, v, s[c].phi * (s[c].a1 + v));
    const double pn = fma(-s[c].q12, g, s[c].qs);
    s[c].p = fabs(s[c].p - pn) < kConvTolAbs ? s[c].p : pn;
  }
}

TAD_HD inline KfOut kfc_finish(const KfStateC &s, uint32_t n, double ylast) {
  const double sumlog = tad_det_log(s.prod) + (double)s.esum * TAD_DM_LN2;
  const double llf = -0.5 * ((double)(n - 1) * kLog2Pi + sumlog) - 0.5 * s.q;
  KfOut o;
  o.nll = -llf / (double)n;
  o.forecast = ylast + s.a1;
  return o;
}

// the four recursions over a strided series (tools/arima_twin.cpp): nll[0..3], forecast of the first
TAD_HD void arima_nll4_collapsed(const double (&xe)[4][3], const double *__restrict__ y, size_t stride, uint32_t n, double (&nll)[4],
                                 double &forecast) {
  KfStateC s4[4];
  double p11[4], r0[4];
  for (int c = 0; c < 4; ++c) kfc_init(s4[c], xe[c][0], xe[c][1], xe[c][2], &p11[c], &r0[c]);
  double yprev = 0.0;
  if (n >= 1) {
    yprev = y[0];
    for (int c = 0; c < 4; ++c) kfc_first(s4[c], yprev, p11[c], r0[c]);
  }
  for (uint32_t t = 1; t < n; ++t) {
    const double yt = y[(size_t)t * stride];
    kfc_step4(s4, yt - yprev, (t & 3u) == 0);
    yprev = yt;
  }
  for (int c = 0; c < 4; ++c) {
    const KfOut r = kfc_finish(s4[c], n, yprev);
    nll[c] = r.nll;
    if (c == 0) forecast = r.forecast;
  }
}

// ------------------------------------------------------------------------------------------------
// start parameters (SARIMAX.start_params -> _conditional_sum_squares, k_ar = k_ma = 1), streaming
// ------------------------------------------------------------------------------------------------
struct Ls2 { double a, b; };

// minimum-norm least squares of Y on two columns = numpy.linalg.pinv(X).dot(Y) with rcond 1e-15, via a
// one-sided Jacobi rotation.  g11 g12 g22 are the Gram entries; the rotated column norms and projections are recomputed from the
// data (second pass) so that small singular values keep accuracy.  `each(body)` visits the rows in order and calls body(c1, c2, yy).
template <typename Each>
TAD_HD Ls2 pinv2_solve(uint32_t rows, Each each) {
  Ls2 r{0.0, 0.0};
  if (rows == 0) return r;
  double g11 = 0.0, g12 = 0.0, g22 = 0.0;
  each([&](double c1, double c2, double) { g11 += c1 * c1; g12 += c1 * c2; g22 += c2 * c2; });
  double cs = 1.0, sn = 0.0;
  if (g12 != 0.0) {
    const double zeta = (g22 - g11) / (2.0 * g12);
    const double tn = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    cs = 1.0 / sqrt(1.0 + tn * tn);
    sn = cs * tn;
  }
  double s1 = 0.0, s2 = 0.0, b1 = 0.0, b2 = 0.0;
  each([&](double c1, double c2, double yy) {
    const double r1 = cs * c1 - sn * c2, r2 = sn * c1 + cs * c2;
    s1 += r1 * r1; s2 += r2 * r2; b1 += r1 * yy; b2 += r2 * yy;
  });
  const double smax = sqrt(fmax(s1, s2));
  const double cut = 1e-15 * smax;
  const double w1 = sqrt(s1) > cut ? b1 / s1 : 0.0;
  const double w2 = sqrt(s2) > cut ? b2 / s2 : 0.0;
  r.a = cs * w1 + sn * w2;
  r.b = -sn * w1 + cs * w2;
  return r;
}

// Rows i0 .. i1-1 of a strided series in order, f(w0 .. w4) with w_j = y[i + j]: a sliding window of five values in registers, ONE
// load per row instead of the 6 - 10 the expressions of the passes below name, and the loads of the NEXT four rows issued before the
// current four are worked on (a pass is a serial chain of adds per lane: with the load inside the chain every row paid an L2 round
// trip, ~130 cycles per row and SIMD against ~50 of arithmetic).  Indices past n - 1 read y[n - 1]: never used by a row < i1.
template <typename F>
TAD_HD void stream_window5(const double *__restrict__ y, size_t stride, uint32_t n, uint32_t i0, uint32_t i1, F f) {
  if (i0 >= i1) return;
  const uint32_t last = n - 1;
  auto Y = [&](uint32_t i) { return y[(size_t)(i < last ? i : last) * stride]; };
  double w0 = Y(i0), w1 = Y(i0 + 1), w2 = Y(i0 + 2), w3 = Y(i0 + 3), w4 = Y(i0 + 4);
  double n0 = Y(i0 + 5), n1 = Y(i0 + 6), n2 = Y(i0 + 7), n3 = Y(i0 + 8);
  uint32_t i = i0;
  for (; i + 4 <= i1; i += 4) {
    const double m0 = Y(i + 9), m1 = Y(i + 10), m2 = Y(i + 11), m3 = Y(i + 12);
    f(w0, w1, w2, w3, w4);
    f(w1, w2, w3, w4, n0);
    f(w2, w3, w4, n0, n1);
    f(w3, w4, n0, n1, n2);
    w0 = w4; w1 = n0; w2 = n1; w3 = n2; w4 = n3;
    n0 = m0; n1 = m1; n2 = m2; n3 = m3;
  }
  for (; i < i1; ++i) {
    f(w0, w1, w2, w3, w4);
    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = n0; n0 = n1; n1 = n2; n2 = n3;
  }
}

TAD_HD void arima_start_params(const double *__restrict__ y, size_t stride, uint32_t n, double *u) {
  const uint32_t m = n - 1;  // number of first differences e_i = y[i+1] - y[i]
  auto e = [&](uint32_t i) { return y[(size_t)(i + 1) * stride] - y[(size_t)i * stride]; };
  double phi0 = 0.0, theta0 = 0.0, var0 = 0.0;
  bool fallback = m <= 2 || m - 2 <= 1;  // lagmat(endog, 2) / lagmat(residuals, 1) raise ValueError
  if (!fallback) {
    // Row i of every pass needs y[i .. i + 4] (stream_window5).  The same differences of the same values as the expressions of
    // the reference name, in the same order: the bits do not change.
    // AR(2) by pinv-OLS: e_t on (e_{t-1}, e_{t-2}), t = 2..m-1
    const Ls2 ar = pinv2_solve(m - 2, [&](auto body) {
      stream_window5(y, stride, n, 0, m - 2, [&](double w0, double w1, double w2, double w3, double) {
        body(w2 - w1, w1 - w0, w3 - w2);   // e(i + 1), e(i), e(i + 2)
      });
    });
    // ARMA(1,1) by pinv-OLS: e_t on (e_{t-1}, res_{t-1}), t = 3..m-1;  res(j) = e(j + 2) - (e(j + 1) a + e(j) b): residual of t = j + 2
    const uint32_t rows = m - 3;
    const Ls2 am = pinv2_solve(rows, [&](auto body) {
      stream_window5(y, stride, n, 0, rows, [&](double w0, double w1, double w2, double w3, double w4) {
        body(w3 - w2, (w3 - w2) - ((w2 - w1) * ar.a + (w1 - w0) * ar.b), w4 - w3);   // e(i + 2), res(i), e(i + 3)
      });
    });
    phi0 = am.a;
    theta0 = am.b;
    if (rows > 1) {
      double s = 0.0;
      stream_window5(y, stride, n, 1, rows, [&](double w0, double w1, double w2, double w3, double w4) {
        const double resi = (w3 - w2) - ((w2 - w1) * ar.a + (w1 - w0) * ar.b);
        const double r2 = (w4 - w3) - ((w3 - w2) * am.a + resi * am.b);
        s += r2 * r2;
      });
      var0 = s / (double)(rows - 1);
    } else {
      double mean = 0.0;
      for (uint32_t i = 0; i < m; ++i) mean += e(i);
      mean /= (double)m;
      double s = 0.0;
      for (uint32_t i = 0; i < m; ++i) { const double d = e(i) - mean; s += d * d; }
      var0 = s / (double)m;  // numpy.var(endog)
    }
  } else {
    double mean = 0.0;
    for (uint32_t i = 0; i < m; ++i) mean += e(i);
    mean /= (double)m;
    double s = 0.0;
    for (uint32_t i = 0; i < m; ++i) { const double d = e(i) - mean; s += d * d; }
    var0 = s / (double)(m + 1);  // mean of residuals[1:] = [0, e - mean(e)]
  }
  if (!(fabs(phi0) < 1.0)) phi0 = 0.0;      // non-stationary start -> zeros
  if (!(fabs(theta0) < 1.0)) theta0 = 0.0;  // non-invertible start -> zeros
  var0 = fmax(var0, 1e-10);
  u[0] = -phi0 / sqrt(1.0 - phi0 * phi0);   // SARIMAX.untransform_params (the inverse of the signs in kf_init)
  u[1] = theta0 / sqrt(1.0 - theta0 * theta0);
  u[2] = sqrt(var0);
}

// ------------------------------------------------------------------------------------------------
// More'-Thuente line search (MINPACK-2 dcsrch / dcstep), restated
// ------------------------------------------------------------------------------------------------
struct LineSearch {
  double stx, fx, gx, sty, fy, gy, stmin, stmax, width, width1, finit, ginit, gtest;
  bool brackt;
  int stage;
};
enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2, LS_ERROR = 3 };

TAD_HD void mt_step(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp,
                        double dp, bool &brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / fabs(dx));
  double stpf;
  if (fp > fx) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    stpf = fabs(stpc - stx) <= fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
    brackt = true;
  } else if (fabs(dp) < fabs(dx)) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
    double stpc;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      stpf = fabs(stpc - stp) < fabs(stpq - stp) ? stpc : stpq;
      if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = fmax(fabs(theta), fmax(fabs(dy), fabs(dp)));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) stpf = stpmax;
    else stpf = stpmin;
  }
  if (fp > fx) { sty = stp; fy = fp; dy = dp; }
  else {
    if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
    stx = stp; fx = fp; dx = dp;
  }
  stp = stpf;
}

TAD_HD int mt_start(LineSearch &L, double stp, double f, double g, double stpmin, double stpmax) {
  if (stp < stpmin || stp > stpmax || g >= 0.0) return LS_ERROR;
  L.brackt = false; L.stage = 1; L.finit = f; L.ginit = g; L.gtest = 1e-3 * g;
  L.width = stpmax - stpmin; L.width1 = L.width / 0.5;
  L.stx = 0.0; L.fx = f; L.gx = g; L.sty = 0.0; L.fy = f; L.gy = g;
  L.stmin = 0.0; L.stmax = stp + 4.0 * stp;
  return LS_FG;
}

TAD_HD int mt_iterate(LineSearch &L, double &stp, double f, double g, double stpmin, double stpmax) {
  const double ftol = 1e-3, gtol = 0.9, xtol = 0.1;
  (void)ftol;
  const double ftest = L.finit + stp * L.gtest;
  if (L.stage == 1 && f <= ftest && g >= 0.0) L.stage = 2;
  int task = LS_FG;
  if (L.brackt && (stp <= L.stmin || stp >= L.stmax)) task = LS_WARN;
  if (L.brackt && L.stmax - L.stmin <= xtol * L.stmax) task = LS_WARN;
  if (stp == stpmax && f <= ftest && g <= L.gtest) task = LS_WARN;
  if (stp == stpmin && (f > ftest || g >= L.gtest)) task = LS_WARN;
  if (f <= ftest && fabs(g) <= gtol * (-L.ginit)) task = LS_CONV;
  if (task != LS_FG) return task;
  if (L.stage == 1 && f <= L.fx && f > ftest) {
    const double fm = f - stp * L.gtest;
    double fxm = L.fx - L.stx * L.gtest, fym = L.fy - L.sty * L.gtest;
    const double gm = g - L.gtest;
    double gxm = L.gx - L.gtest, gym = L.gy - L.gtest;
    mt_step(L.stx, fxm, gxm, L.sty, fym, gym, stp, fm, gm, L.brackt, L.stmin, L.stmax);
    L.fx = fxm + L.stx * L.gtest; L.fy = fym + L.sty * L.gtest;
    L.gx = gxm + L.gtest; L.gy = gym + L.gtest;
  } else {
    mt_step(L.stx, L.fx, L.gx, L.sty, L.fy, L.gy, stp, f, g, L.brackt, L.stmin, L.stmax);
  }
  if (L.brackt) {
    if (fabs(L.sty - L.stx) >= 0.66 * L.width1) stp = L.stx + 0.5 * (L.sty - L.stx);
    L.width1 = L.width;
    L.width = fabs(L.sty - L.stx);
  }
  if (L.brackt) { L.stmin = fmin(L.stx, L.sty); L.stmax = fmax(L.stx, L.sty); }
  else { L.stmin = stp + 1.1 * (stp - L.stx); L.stmax = stp + 4.0 * (stp - L.stx); }
  stp = fmax(stp, stpmin);
  stp = fmin(stp, stpmax);
  if ((L.brackt && (stp <= L.stmin || stp >= L.stmax)) || (L.brackt && L.stmax - L.stmin <= xtol * L.stmax)) stp = L.stx;
  return LS_FG;
}

// ------------------------------------------------------------------------------------------------
// L-BFGS-B 3.0, unconstrained path, as a per-lane state machine driven by (f, g) deliveries
// ------------------------------------------------------------------------------------------------
struct Lbfgs {
  double x[3], g[3], f;
  double d[3], t[3], r[3];  // search direction, iterate and gradient at the start of the line search
  double fold, gd, gdold, stp, theta;
  double fc, fcold;  // one-step forecast of the model at x / at the start of the line search (no extra filter run at the end)
  // The (s, y) history (kLbfgsM pairs of 3-vectors = 60 doubles) lives in MEMORY the caller provides, a circular buffer:
  // element c of s / y of physical slot i at hist[((i * 2 + {0, 1}) * 3 + c) * hstride].  On the device that is a per-wavefront
  // block of global memory with the lanes interleaved (hstride = 64: every access a coalesced 512-byte line) — as registers the
  // history pushed ~100 VGPRs of spills into the optimiser step (measured: a third of k_arima_fit's time went into that step);
  // lbfgs_direction fetches it in two halves right where the two-loop recursion needs it.
  double *hist;
  uint32_t hstride;
  double *park;       // where lbfgs_park / lbfgs_unpark keep the rest of this struct between steps (kParkDoubles values, element stride pstride)
  uint32_t pstride;
  int col, head, iter, ifun, iback, nit;
  bool in_ls, done;
  LineSearch ls;
};

TAD_HD inline double &lbfgs_hist(const Lbfgs &o, int slot, int which, int c) { return o.hist[(size_t)((slot * 2 + which) * 3 + c) * o.hstride]; }

// Five pairs of the history in LOGICAL order (0 = oldest), registers.  The two-loop recursion walks the ten pairs newest to oldest
// and back: it fetches them in halves — [5, 10), [0, 5), (forward: [0, 5) is still there), [5, 10) — each a batch of 30 independent
// loads, predicated on j < col.  All ten at once were 120 VGPRs next to the line search's state: at 256 VGPRs (two wavefronts per
// SIMD) the optimiser step spilled 131 of them.
static constexpr int kLbfgsHalf = kLbfgsM / 2;
struct LbfgsHalf { double S[kLbfgsHalf][3], Y[kLbfgsHalf][3]; };

TAD_HD inline void lbfgs_load_half(const Lbfgs &o, int j0, LbfgsHalf &h) {
#pragma unroll
  for (int jj = 0; jj < kLbfgsHalf; ++jj) {
    const int j = j0 + jj;
    int slot = o.head + j;
    if (slot >= kLbfgsM) slot -= kLbfgsM;
    const bool live = j < o.col;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      h.S[jj][c] = live ? lbfgs_hist(o, slot, 0, c) : 0.0;
      h.Y[jj][c] = live ? lbfgs_hist(o, slot, 1, c) : 0.0;
    }
  }
}

// -H g by the two-loop recursion, H0 = I / theta (the history is read from memory: the pair the update has just written included)
TAD_HD void lbfgs_direction(Lbfgs &o) {
  if (o.col == 0) {
    for (int i = 0; i < 3; ++i) o.d[i] = -o.g[i];  // Cauchy point with B = theta I, theta = 1
    return;
  }
  double q[3] = {o.g[0], o.g[1], o.g[2]}, alpha[kLbfgsM];
  LbfgsHalf h;
#pragma unroll
  for (int half = 1; half >= 0; --half) {
    const int j0 = half * kLbfgsHalf;
    if (o.col > j0) {
      lbfgs_load_half(o, j0, h);
#pragma unroll
      for (int jj = kLbfgsHalf - 1; jj >= 0; --jj) {
        const int j = j0 + jj;
        alpha[j] = 0.0;
        if (j < o.col) {
          const double sy = h.S[jj][0] * h.Y[jj][0] + h.S[jj][1] * h.Y[jj][1] + h.S[jj][2] * h.Y[jj][2];
          alpha[j] = (h.S[jj][0] * q[0] + h.S[jj][1] * q[1] + h.S[jj][2] * q[2]) / sy;
#pragma unroll
          for (int c = 0; c < 3; ++c) q[c] -= alpha[j] * h.Y[jj][c];
        }
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < kLbfgsHalf; ++jj) alpha[j0 + jj] = 0.0;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) q[c] /= o.theta;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int j0 = half * kLbfgsHalf;
    if (o.col > j0) {
      if (half == 1) lbfgs_load_half(o, j0, h);   // (half 0 is what the backward loop loaded last)
#pragma unroll
      for (int jj = 0; jj < kLbfgsHalf; ++jj) {
        const int j = j0 + jj;
        if (j < o.col) {
          const double sy = h.S[jj][0] * h.Y[jj][0] + h.S[jj][1] * h.Y[jj][1] + h.S[jj][2] * h.Y[jj][2];
          const double beta = (h.Y[jj][0] * q[0] + h.Y[jj][1] * q[1] + h.Y[jj][2] * q[2]) / sy;
#pragma unroll
          for (int c = 0; c < 3; ++c) q[c] += h.S[jj][c] * (alpha[j] - beta);
        }
      }
    }
  }
  for (int c = 0; c < 3; ++c) o.d[c] = -q[c];
}

// Everything of the optimiser's state that the likelihood pass does not need (all but x, col, head and the flags) is PARKED
// while the pass runs — on the device in LDS (kParkDoubles x 64 lanes x 8 B = 14.5 KB per wavefront: with the 4.5 KB staging
// buffer eight wavefronts per CU still fit), on the host twin in an array — and fetched back at the start of lbfgs_deliver: the
// pass has the registers to itself and nothing of the step's state is spilled to scratch memory (a reload from there costs a
// memory round trip; as members of one long-lived struct the fields also stayed live around the loop on the idle-lane path).
static constexpr int kParkDoubles = 29;
static constexpr int kSaveDoubles = 7 + kParkDoubles;   // a suspended lane (k_arima_fit's cooperative yield): busy, key, x[3], fc, col | head + the parked state
TAD_HD inline void lbfgs_park(const Lbfgs &o) {
  double *m = o.park;
  const size_t st = o.pstride;
  const uint64_t ints = (uint64_t)(uint16_t)o.iter | (uint64_t)(uint16_t)o.ifun << 16 | (uint64_t)(uint16_t)o.iback << 32 | (uint64_t)(uint16_t)o.nit << 48;
  const uint64_t flags = (o.in_ls ? 1u : 0u) | (o.ls.brackt ? 2u : 0u) | (uint64_t)(uint32_t)o.ls.stage << 8;
  const double v[kParkDoubles] = {o.d[0], o.d[1], o.d[2], o.t[0], o.t[1], o.t[2], o.r[0], o.r[1], o.r[2],
                                  o.fold, o.gdold, o.stp, o.theta, o.fcold,
                                  o.ls.stx, o.ls.fx, o.ls.gx, o.ls.sty, o.ls.fy, o.ls.gy, o.ls.stmin, o.ls.stmax, o.ls.width, o.ls.width1,
                                  o.ls.finit, o.ls.ginit, o.ls.gtest, tad_dm_f64(ints), tad_dm_f64(flags)};
#pragma unroll
  for (int i = 0; i < kParkDoubles; ++i) m[(size_t)i * st] = v[i];
}
TAD_HD inline void lbfgs_unpark(Lbfgs &o) {
  const double *m = o.park;
  const size_t st = o.pstride;
  double v[kParkDoubles];
#pragma unroll
  for (int i = 0; i < kParkDoubles; ++i) v[i] = m[(size_t)i * st];
  o.d[0] = v[0]; o.d[1] = v[1]; o.d[2] = v[2]; o.t[0] = v[3]; o.t[1] = v[4]; o.t[2] = v[5]; o.r[0] = v[6]; o.r[1] = v[7]; o.r[2] = v[8];
  o.fold = v[9]; o.gdold = v[10]; o.stp = v[11]; o.theta = v[12]; o.fcold = v[13];
  o.ls.stx = v[14]; o.ls.fx = v[15]; o.ls.gx = v[16]; o.ls.sty = v[17]; o.ls.fy = v[18]; o.ls.gy = v[19]; o.ls.stmin = v[20]; o.ls.stmax = v[21];
  o.ls.width = v[22]; o.ls.width1 = v[23]; o.ls.finit = v[24]; o.ls.ginit = v[25]; o.ls.gtest = v[26];
  const uint64_t ints = tad_dm_u64(v[27]), flags = tad_dm_u64(v[28]);
  o.iter = (int)(ints & 0xffff); o.ifun = (int)(ints >> 16 & 0xffff); o.iback = (int)(ints >> 32 & 0xffff); o.nit = (int)(ints >> 48);
  o.in_ls = (flags & 1) != 0; o.ls.brackt = (flags & 2) != 0; o.ls.stage = (int)(flags >> 8);
}
static constexpr int kHistDoubles = kLbfgsM * 6;   // doubles of (s, y) history per lane in global memory


// forward-difference point of scipy's approx_derivative(method='2-point', abs_step=1e-5): x + 1e-5, unless that does not
// change x (|x| > ~1e11: the huge sigma parameters of void Box-Cox regimes) — then the relative step sqrt(eps) * sign(x) *
// max(1, |x|) (scipy/optimize/_numdiff.py: "cannot have a zero step ... fall back to relative step").  Returns x + h.
TAD_HD inline double fd_point(double x0, double *dx) {
  double xe = x0 + 1e-5;
  *dx = xe - x0;
  if (*dx == 0.0) {
    const double h = 1.4901161193847656e-08 * (x0 >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(x0));
    xe = x0 + h;
    *dx = xe - x0;
  }
  return xe;
}

// begin a line search from the current (x, f, g); sets the first trial point in x
TAD_HD void lbfgs_begin_ls(Lbfgs &o) {
  for (;;) {
    lbfgs_direction(o);
    const double dtd = o.d[0] * o.d[0] + o.d[1] * o.d[1] + o.d[2] * o.d[2];
    const double dnorm = sqrt(dtd);
    o.stp = o.iter == 0 ? fmin(1.0 / dnorm, 1e10) : 1.0;
    for (int i = 0; i < 3; ++i) { o.t[i] = o.x[i]; o.r[i] = o.g[i]; }
    o.fold = o.f;
    o.fcold = o.fc;
    o.ifun = 0;
    o.iback = 0;
    o.gd = o.g[0] * o.d[0] + o.g[1] * o.d[1] + o.g[2] * o.d[2];
    o.gdold = o.gd;
    int task = LS_ERROR;
    if (o.gd < 0.0) task = mt_start(o.ls, o.stp, o.f, o.gd, 0.0, 1e10);
    if (task == LS_FG) break;
    // ascent direction / bad step: info != 0
    if (o.col == 0) { o.done = true; return; }  // ABNORMAL_TERMINATION_IN_LNSRCH (x, f already the old iterate)
    o.col = 0; o.head = 0; o.theta = 1.0;          // refresh the memory and restart with steepest descent
  }
  o.ifun = 1;
  o.iback = 0;
  for (int i = 0; i < 3; ++i) o.x[i] = o.stp == 1.0 ? o.t[i] + o.d[i] : o.stp * o.d[i] + o.t[i];
  o.in_ls = true;
}

// (the inner function; lbfgs_deliver below wraps it in the unpark / park of the state)
TAD_HD void lbfgs_deliver_core(Lbfgs &o, int maxiter) {
  const double pgtol = 1e-5, factr = 1e7;
  if (!o.in_ls) {  // first evaluation
    const double sbg = fmax(fabs(o.g[0]), fmax(fabs(o.g[1]), fabs(o.g[2])));
    if (sbg <= pgtol) { o.done = true; return; }
    lbfgs_begin_ls(o);
    return;
  }
  o.gd = o.g[0] * o.d[0] + o.g[1] * o.d[1] + o.g[2] * o.d[2];
  const int task = mt_iterate(o.ls, o.stp, o.f, o.gd, 0.0, 1e10);
  if (task == LS_FG) {
    o.ifun++;
    o.iback = o.ifun - 1;
    if (o.iback >= 20) {  // maxls: give up on this direction
      for (int i = 0; i < 3; ++i) { o.x[i] = o.t[i]; o.g[i] = o.r[i]; }
      o.f = o.fold;
      o.fc = o.fcold;
      if (o.col == 0) { o.done = true; return; }
      o.col = 0; o.head = 0; o.theta = 1.0;
      o.in_ls = false;  // restart from the restored iterate
      lbfgs_begin_ls(o);
      return;
    }
    for (int i = 0; i < 3; ++i) o.x[i] = o.stp == 1.0 ? o.t[i] + o.d[i] : o.stp * o.d[i] + o.t[i];
    return;  // evaluate the new trial point
  }
  // CONVERGENCE or WARNING: the trial point is the new iterate
  o.iter++;
  o.nit++;
  if (o.nit >= maxiter) { o.done = true; return; }  // the driver stops (scipy: n_iterations >= maxiter)
  const double sbg = fmax(fabs(o.g[0]), fmax(fabs(o.g[1]), fabs(o.g[2])));
  if (sbg <= pgtol) { o.done = true; return; }
  const double ddum0 = fmax(fabs(o.fold), fmax(fabs(o.f), 1.0));
  if ((o.fold - o.f) <= kEpsMch * factr * ddum0) { o.done = true; return; }
  // BFGS update
  double rr = 0.0;
  for (int i = 0; i < 3; ++i) { o.r[i] = o.g[i] - o.r[i]; rr += o.r[i] * o.r[i]; }
  double dr, ddum;
  if (o.stp == 1.0) { dr = o.gd - o.gdold; ddum = -o.gdold; }
  else { dr = (o.gd - o.gdold) * o.stp; for (int i = 0; i < 3; ++i) o.d[i] *= o.stp; ddum = -o.gdold * o.stp; }
  if (!(dr <= kEpsMch * ddum)) {
    int slot;
    if (o.col < kLbfgsM) { slot = o.head + o.col; if (slot >= kLbfgsM) slot -= kLbfgsM; }
    else {   // full: the oldest pair is overwritten; the logical order moves up by one
      slot = o.head;
      o.head = o.head + 1 == kLbfgsM ? 0 : o.head + 1;
      o.col = kLbfgsM - 1;
    }
    for (int i = 0; i < 3; ++i) { lbfgs_hist(o, slot, 0, i) = o.d[i]; lbfgs_hist(o, slot, 1, i) = o.r[i]; }
    o.col++;
    o.theta = rr / dr;
  }
  o.in_ls = false;
  lbfgs_begin_ls(o);
}

// What a lane keeps in REGISTERS between optimiser steps (across the likelihood pass): the point to evaluate, the history's
// shape and where its state block is.  Everything else is local to lbfgs_deliver — declared there, so that it is provably dead
// during the pass (as members of one long-lived struct the fields stayed live on the idle-lane path around the loop and the
// compiler spilled them, one memory round trip per reload).
struct LbfgsLive {
  double x[3];
  double fc;       // forecast of the accepted iterate (valid once done)
  double *hist;    // this lane's (s, y) history: kHistDoubles doubles, element stride hstride
  uint32_t hstride;
  double *park;    // this lane's parked state: kParkDoubles doubles, element stride pstride
  uint32_t pstride;
  int col, head;
  bool done;
};

// a fresh fit: start parameters in x; the parked state says "first evaluation"
TAD_HD inline void lbfgs_reset(LbfgsLive &L, double u0, double u1, double u2) {
  L.x[0] = u0; L.x[1] = u1; L.x[2] = u2;
  L.fc = 0.0; L.col = 0; L.head = 0; L.done = false;
  Lbfgs o{};
  o.park = L.park; o.pstride = L.pstride;
  o.theta = 1.0; o.in_ls = false;
  lbfgs_park(o);
}

// (f, g, forecast) at L.x have just been delivered: one optimiser step.  The parked state and the whole history come in as one
// batch of independent loads; the step's state goes back (the new (s, y) pair is written by the update itself).
TAD_HD void lbfgs_deliver(LbfgsLive &L, double f, const double (&g)[3], double fc, int maxiter) {
  Lbfgs o;
  o.hist = L.hist; o.hstride = L.hstride; o.park = L.park; o.pstride = L.pstride; o.col = L.col; o.head = L.head; o.done = false;
  lbfgs_unpark(o);
  for (int i = 0; i < 3; ++i) { o.x[i] = L.x[i]; o.g[i] = g[i]; }
  o.f = f; o.fc = fc;
  lbfgs_deliver_core(o, maxiter);
  lbfgs_park(o);
  for (int i = 0; i < 3; ++i) L.x[i] = o.x[i];
  L.fc = o.fc; L.col = o.col; L.head = o.head; L.done = o.done;
}

// ------------------------------------------------------------------------------------------------
// k_arima_start — start parameters of every fit (lane = (key, position), 64 consecutive keys per wavefront: coalesced)
// ------------------------------------------------------------------------------------------------
#ifdef TAD_START_WAVES   // measurement knob (tools/build_variants.py): pin the occupancy the register allocator aims for
#define TAD_START_ATTR __attribute__((amdgpu_waves_per_eu(TAD_START_WAVES, TAD_START_WAVES)))
#else
#define TAD_START_ATTR
#endif
__global__ __launch_bounds__(64) TAD_START_ATTR void k_arima_start(Grid g, ArimaWs ws, const uint32_t *__restrict__ n_pts, uint32_t pmax) {
#ifdef TAD_START_PMAJOR
  const uint32_t kblocks = (uint32_t)((g.K + 63) / 64);
  const uint32_t p = pmax - 1 - blockIdx.x / kblocks;
  const uint64_t k = (uint64_t)(blockIdx.x % kblocks) * 64 + threadIdx.x;
#else
  // Key-block major: the pmax - 3 workgroups that walk the SAME 64 keys' series (64 x T x 8 B = 128 KB at T = 250) are consecutive
  // block ids, so they run together and all but the first of them on an XCD find the rows in its L2.  Position major (every key
  // block at one position, then the next position) re-streamed the whole [T][K] plane - 200 MB at C3, past every L2 - once per position.
  const uint32_t npos = pmax - 3;
  const uint32_t p = pmax - 1 - blockIdx.x % npos;
  const uint64_t k = (uint64_t)(blockIdx.x / npos) * 64 + threadIdx.x;
#endif
  if (k >= g.K || ws.state[k] != 0 || n_pts[k] <= p) return;
  double u[3];
  arima_start_params(ws.ys + k, g.K, p, u);
  const size_t c = (size_t)p * g.K + k;
  ws.u0[0][c] = u[0]; ws.u0[1][c] = u[1]; ws.u0[2][c] = u[2];
}

// ------------------------------------------------------------------------------------------------
// k_arima_fit — one lane = one fit at a time; a wavefront = a chunk of keys at ONE series position p (so every lane's
// Kalman loop has the same length) whose lanes pull the next key of the chunk as soon as their fit has converged.
// Why: the optimiser needs 20 .. 370 likelihood evaluations per fit (mean 81, CV 0.5).  With a fixed lane <-> key map a
// wavefront runs until its slowest lane is done — measured 2.4x the mean (oracle/arima_exact.c:arima_exact_fit_profile);
// refilling lanes keeps all 64 busy until the chunk runs dry.
// Lanes then hold arbitrary keys, so the series are read from the KEY-major copy: every 8 time steps the wavefront loads
// the next 64 bytes of each lane's row cooperatively (4 lanes x 16 B per row: whole sectors, no over-fetch), stages them
// in LDS (row stride 9 doubles: conflict-free) and every lane picks up its 8 values.
// The optimiser runs in cycles: f at x and at the three forward-difference points as FOUR interleaved recursions per lane in
// one pass over the series (the joint recursion of the contract: one division per step for the four reciprocals), then one
// state-machine step, the same for all lanes; finished lanes are refilled between cycles.
// ------------------------------------------------------------------------------------------------
static constexpr int kStage = 8;  // time steps staged per round: 64 B per row

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#if defined(TAD_ARIMA_PROF)
// profiling build (tools/gpu_arima_prof.sh; never the shipped library): shader-clock cycles per wavefront spent in the
// likelihood pass / the optimiser step / refill, summed over all wavefronts
__device__ unsigned long long g_arima_prof[8];
#define TAD_PROF_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TAD_PROF_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#else
#define TAD_PROF_T(v)
#define TAD_PROF_ADD(slot, t0, t1)
#endif

__device__ __forceinline__ void arima_fit_body(Grid g, ArimaWs ws, const double *__restrict__ sigma,
                                               const uint32_t *__restrict__ n_pts, int maxiter, uint32_t pmax, uint32_t chunk,
                                               double *__restrict__ calc, DevCounters *ctr, double *buf, double *park, const int *pause, uint32_t grace) {
  const uint32_t nchunks = (uint32_t)((g.K + chunk - 1) / chunk);   // wavefronts per position
  const uint32_t p = pmax - 1 - blockIdx.x / nchunks;           // heaviest (longest history) positions first
  const unsigned lane = threadIdx.x;
  bool dry = false;                                             // wave-uniform: position p has no key left to hand out
  bool yielded = false;                                         // wave-uniform: ... or the engine asked the fit to make room (cooperative yield)
  unsigned polls = 0, iter = 0;
  uint64_t k = 0;
  bool busy = false;
  unsigned long long steps = 0, fits = 0, nanfits = 0;
#if defined(TAD_ARIMA_PROF)
  unsigned long long prof[4] = {0, 0, 0, 0};
#endif
  LbfgsLive o;
  o.x[0] = 0.0; o.x[1] = 0.0; o.x[2] = 1.0;
  o.fc = 0.0; o.col = 0; o.head = 0; o.done = true;
  o.hist = ws.hist + (size_t)blockIdx.x * (kHistDoubles * 64) + lane;   // this wavefront's block of global memory, lanes interleaved
  o.hstride = 64;
  o.park = park + lane;                                                  // LDS, lanes interleaved: conflict-free
  o.pstride = 64;
  size_t row[4] = {0, 0, 0, 0};                                 // element offset of the rows this lane loads for the wavefront

  auto refill = [&]() {
    // The wavefronts of one position share ONE cursor over its keys (a global atomic per refill): with a private chunk of 4096
    // keys per wavefront the lanes of a chunk run dry one after the other while its slowest fits (up to 370 evaluations against a
    // mean of 81) finish — now a position's wavefronts all end together.  Which lane fits which key does not enter the results.
    while (!dry) {
      const unsigned long long m = __ballot(!busy);
      if (m == 0) break;
      // Cooperative yield (see the main loop): no key is handed out while the engine's pause word is raised.  Asked on a wavefront's first
      // refill (a workgroup that starts while the word is raised retires at once) and every 16th after it; the main loop asks every cycle.
      if (pause != nullptr && iter >= grace && (polls++ & 15u) == 0u && __hip_atomic_load(pause, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        yielded = true;
        dry = true;
        break;
      }
      unsigned long long first = 0;
      if (lane == 0) first = atomicAdd(&ws.cursor[p], (unsigned long long)__popcll(m));
      first = __shfl(first, 0);
      if (first >= g.K) { dry = true; break; }
      const uint64_t cand = first + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
      if (!busy && cand < g.K && ws.state[cand] == 0 && n_pts[cand] > p) {
        k = cand;
        busy = true;
        const size_t c = (size_t)p * g.K + k;
        lbfgs_reset(o, ws.u0[0][c], ws.u0[1][c], ws.u0[2][c]);
      }
    }
    const unsigned long long mine = busy ? (unsigned long long)k : 0ull;   // idle lanes: row 0 (valid memory, values unused)
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = (size_t)__shfl(mine, j * 16 + (int)(lane >> 2)) * ws.Tpad + (lane & 3u) * 2u;
  };

  // one pass over the series: the four recursions of the contract (the difference y_t - y_t-1 is shared by them)
  auto evaluate4 = [&](const double (&xe)[4][3], double (&nll)[4], double &fc) {
    KfStateC s4[4];
    double yprev;
    {   // t = 0 (burned) up front, from the lane's own first value: p11 and r0 are not carried through the loop
      const double y0 = ws.ysk[(size_t)(busy ? k : 0ull) * ws.Tpad];
      yprev = y0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double p11, r0;
        kfc_init(s4[c], xe[c][0], xe[c][1], xe[c][2], &p11, &r0);
        kfc_first(s4[c], y0, p11, r0);
      }
    }
    // The rows of the NEXT stage are requested before this stage's steps run (two register sets of 4 x 16 B): the recursion
    // used to wait out a full memory round trip every eight steps.  The barriers order LDS only (one wavefront per workgroup;
    // __syncthreads() would also wait for the loads just issued).
    double2 vn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vn[j] = *reinterpret_cast<const double2 *>(ws.ysk + row[j]);
    for (uint32_t t0 = 0; t0 < p; t0 += kStage) {
      double2 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = vn[j];
      if (t0 + kStage < p) {   // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) vn[j] = *reinterpret_cast<const double2 *>(ws.ysk + row[j] + t0 + kStage);
      }
      lds_only_barrier();   // the previous round's reads are done
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double *d = buf + (size_t)(j * 16 + (int)(lane >> 2)) * (kStage + 1) + (lane & 3u) * 2u;
        d[0] = v[j].x; d[1] = v[j].y;
      }
      lds_only_barrier();
      const uint32_t nb = p - t0 < (uint32_t)kStage ? p - t0 : (uint32_t)kStage;   // wave-uniform
      double yv[kStage];
#pragma unroll
      for (int i = 0; i < kStage; ++i) yv[i] = buf[(size_t)lane * (kStage + 1) + i];
      // compile-time indices into yv (a runtime-indexed array would live in scratch memory); nb is wave-uniform
#pragma unroll
      for (int i = 0; i < kStage; ++i)
        if ((uint32_t)i < nb) {
          const double d = yv[i] - yprev;
          yprev = yv[i];
          if (i != 0 || t0 != 0) kfc_step4(s4, d, (i & 3) == 0);   // (t = 0 is done; t = t0 + i with t0 a multiple of 8: t & 3 == i & 3)
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const KfOut r = kfc_finish(s4[c], p, yprev);
      nll[c] = r.nll;
      if (c == 0) fc = r.forecast;
    }
  };

  // Resume: this wavefront suspended its fits at the last launch (below) — every lane takes its key and optimiser state back; the (s, y)
  // history never left this wavefront's block of global memory.
  unsigned int *const saved_flag = ws.saved + blockIdx.x;
  double *const save = ws.save + (size_t)blockIdx.x * (kSaveDoubles * 64) + lane;      // lanes interleaved: coalesced
  // (every lane reads the flag before lane 0 clears it: the shuffle is where the lanes meet)
  unsigned int was_saved = *saved_flag;
  was_saved = (unsigned int)__shfl((int)was_saved, 0);
  if (was_saved != 0u) {      // wave-uniform
    busy = save[0] != 0.0;
    k = (uint64_t)__double_as_longlong(save[64]);
    o.x[0] = save[2 * 64]; o.x[1] = save[3 * 64]; o.x[2] = save[4 * 64];
    o.fc = save[5 * 64];
    const long long ch = __double_as_longlong(save[6 * 64]);
    o.col = (int)(ch >> 32); o.head = (int)(ch & 0xffffffffll);
    o.done = !busy;
#pragma unroll 1      // (cold code: one double at a time keeps the kernel's register peak where the optimiser step put it)
    for (int i = 0; i < kParkDoubles; ++i) o.park[(size_t)i * o.pstride] = save[(size_t)(7 + i) * 64];
    if (lane == 0) *saved_flag = 0u;
  }
  refill();
  while (__any(busy)) {
    // Cooperative yield.  The fit is ~0.25 s of two-wavefront-per-SIMD work whose wavefronts live for tens of ms; a workgroup of another
    // job's pass B needs a WHOLE CU (1024 threads, 156 KB of LDS) and would wait until this grid is exhausted whatever the stream priorities
    // (measured: 212 ms, profiles/r6_a2_*).  `pause` is a word in DEVICE memory the engine raises while such a job is in flight (a 4-byte
    // fill on its signal stream).  It is read at the top of every fourth cycle (an agent-scope load: the XCDs' L2s are not coherent with each other,
    // the load goes to the memory side; a page-locked HOST word polled at this rate doubled the kernel's time) and acted on at the end of the cycle, where a fit's whole state is (key, x, fc, the history's shape) in registers, 29 parked doubles in
    // LDS and the history in this wavefront's own global block: the lanes write the first two to `save`, the wavefront retires — within
    // one likelihood pass (tens of us) of the word being raised, not after its longest fit (up to ~10 ms) — and the host relaunches the
    // kernel when the word clears (tad_capi.cpp): the same wavefront index takes the same lanes back, so which lane fits which key, and
    // every bit of every result, is what an undisturbed run gives.  `grace`: cycles during which the word is ignored (a relaunch that was
    // forced after the host's 2 ms wait must make progress although short jobs keep arriving).
    int paused_now = 0;
#if !defined(TAD_ARIMA_POLL_MASK)       // poll every (mask + 1)-th cycle (measurement builds, tools/build_variants.py: 0 = every cycle, 0xFFFFFFFF = never).
#define TAD_ARIMA_POLL_MASK 3u          // C3, same process: every cycle 276.4 ms, every 4th 275.0, every 16th 275.3, never 275.2 (profiles/r6_p1_ab_c3_poll.log)
#endif
    if (pause != nullptr && (iter & TAD_ARIMA_POLL_MASK) == 0u) paused_now = __hip_atomic_load(pause, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TAD_PROF_T(t_a);
    double xe[4][3], dx[3], nll[4], fc0 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xe[c][0] = o.x[0]; xe[c][1] = o.x[1]; xe[c][2] = o.x[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) xe[i + 1][i] = fd_point(xe[i + 1][i], &dx[i]);
    evaluate4(xe, nll, fc0);
    TAD_PROF_T(t_b);
    if (busy) {
      steps += 4ull * p;
      double gr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) gr[i] = (nll[i + 1] - nll[0]) / dx[i];
      lbfgs_deliver(o, nll[0], gr, fc0, maxiter);
      if (o.done) {
        const size_t st = g.K;
        const uint64_t c = (uint64_t)ws.tpos[(size_t)p * st + k] * g.K + k;
        const double pred = inv_boxcox(o.fc, ws.lam[k]);
        calc[c] = pred;
        if (fabs(ws.xs[(size_t)p * st + k] - pred) > sigma[k]) g.flag[c] = FLAG_PRESENT | FLAG_ANOMALY;
        if (!tad_finite(pred)) nanfits++;   // the optimiser walked into a non-finite likelihood: the point can never be an anomaly
        fits++;
        busy = false;
      }
    }
    TAD_PROF_T(t_c);
    if (paused_now != 0 && iter >= grace) {      // wave-uniform: suspend
      save[0] = busy ? 1.0 : 0.0;
      save[64] = __longlong_as_double((long long)k);
      save[2 * 64] = o.x[0]; save[3 * 64] = o.x[1]; save[4 * 64] = o.x[2];
      save[5 * 64] = o.fc;
      save[6 * 64] = __longlong_as_double(((long long)o.col << 32) | (long long)(unsigned int)o.head);
#pragma unroll 1
      for (int i = 0; i < kParkDoubles; ++i) save[(size_t)(7 + i) * 64] = o.park[(size_t)i * o.pstride];
      if (lane == 0) *saved_flag = 1u;
      yielded = true;
      break;
    }
    ++iter;
    refill();
    TAD_PROF_T(t_d);
    TAD_PROF_ADD(0, t_a, t_b); TAD_PROF_ADD(1, t_b, t_c); TAD_PROF_ADD(2, t_c, t_d); TAD_PROF_ADD(3, 0ull, 1ull);
  }
  for (int d = 32; d >= 1; d >>= 1) { steps += __shfl_down(steps, d); fits += __shfl_down(fits, d); nanfits += __shfl_down(nanfits, d); }
  if (threadIdx.x == 0 && yielded) atomicOr(ws.yielded, 1u);
  if (threadIdx.x == 0 && fits) {
    atomicAdd(&ctr->kalman_steps, steps);
    atomicAdd(&ctr->arima_fits, fits);
    if (nanfits) atomicAdd(&ctr->arima_nan_fits, nanfits);
  }
#if defined(TAD_ARIMA_PROF)
  if (threadIdx.x == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_arima_prof[i], prof[i]);
#endif
}

// Two wavefronts per SIMD (223 VGPRs, nothing spilled since the two-loop recursion fetches the history in halves): C3 0.293 s
// against 0.325 s at one — profiles/r3_v7_arima_history_halves_ab.log.  One wavefront issues an FP64 instruction every 5.2-6.5
// clocks whatever its instruction-level parallelism (55-70 % of the SIMD's rate), two reach 84 %
// (tools/probes/fp64_issue_probe.hip).  With all ten pairs in registers the step spilled 131 VGPRs at 256 and one wavefront with
// the whole register file was the faster configuration (0.342 against 0.361 s, profiles/r3_v7_arima_single_path_ab.log).
#if !defined(TAD_ARIMA_WAVES)
#define TAD_ARIMA_WAVES 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TAD_ARIMA_WAVES, TAD_ARIMA_WAVES))) void k_arima_fit(
    Grid g, ArimaWs ws, const double *__restrict__ sigma, const uint32_t *__restrict__ n_pts, int maxiter, uint32_t pmax,
    uint32_t chunk, double *__restrict__ calc, DevCounters *ctr, const int *pause, uint32_t grace) {
  __shared__ double buf[64 * (kStage + 1)];
  __shared__ double park[64 * kParkDoubles];
  arima_fit_body(g, ws, sigma, n_pts, maxiter, pmax, chunk, calc, ctr, buf, park, pause, grace);
}

static uint32_t arima_tpad(uint64_t T) { return (uint32_t)((T + kStage - 1) / kStage * kStage); }

static constexpr uint32_t kArimaChunk = 4096;   // keys per wavefront and position: ~64 fits per lane (measured: 256 -> 1.48 s, 1024 -> 1.28 s, 4096 -> 1.22 s at C3)
static uint64_t arima_fit_blocks(Grid g) { return g.T > 3 ? ((g.K + kArimaChunk - 1) / kArimaChunk) * (g.T - 3) : 0; }

size_t arima_workspace_bytes(Grid g) {
  const size_t cells = (size_t)g.K * g.T;
  return cells * (8 * 3 + 8 * 3 + 4) + (size_t)g.K * arima_tpad(g.T) * 8 + (size_t)g.K * 9 + 1024 +
         (size_t)arima_fit_blocks(g) * (kHistDoubles * 64 * 8) + 512 +   // + the L-BFGS history block of every wavefront of k_arima_fit (30 KB each)
         (size_t)g.T * 8 + 512 + 512 + 64 +                              // + the per-position key cursors + the yield word (each region 512-byte aligned)
         (size_t)arima_fit_blocks(g) * (4 + kSaveDoubles * 64 * 8) + 1024;   // + the suspend flag and save block of every wavefront of k_arima_fit (18 KB each)
}

static ArimaWs arima_carve(Grid g, void *workspace) {
  const size_t cells = (size_t)g.K * g.T;
  unsigned char *w = static_cast<unsigned char *>(workspace);
  ArimaWs ws;
  ws.Tpad = arima_tpad(g.T);
  ws.xs = reinterpret_cast<double *>(w); w += cells * 8;
  ws.lx = reinterpret_cast<double *>(w); w += cells * 8;
  ws.ys = reinterpret_cast<double *>(w); w += cells * 8;
  for (int i = 0; i < 3; ++i) { ws.u0[i] = reinterpret_cast<double *>(w); w += cells * 8; }
  ws.ysk = reinterpret_cast<double *>(w); w += (size_t)g.K * ws.Tpad * 8;
  ws.lam = reinterpret_cast<double *>(w); w += (size_t)g.K * 8;
  ws.tpos = reinterpret_cast<uint32_t *>(w); w += cells * 4;
  ws.state = w; w += (size_t)g.K;
  auto align512 = [](unsigned char *q) { return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(q) + 511) & ~(uintptr_t)511); };
  w = align512(w);   // (tpos is 4 bytes per cell: with an odd cell count everything after it was 4-byte aligned only — fatal for the 64-bit atomics on the cursors)
  ws.hist = reinterpret_cast<double *>(w); w = align512(w + (size_t)arima_fit_blocks(g) * (kHistDoubles * 64 * 8));
  ws.cursor = reinterpret_cast<unsigned long long *>(w); w = align512(w + (size_t)g.T * 8);
  ws.yielded = reinterpret_cast<unsigned int *>(w); w = align512(w + 64);
  ws.saved = reinterpret_cast<unsigned int *>(w); w = align512(w + (size_t)arima_fit_blocks(g) * 4);
  ws.save = reinterpret_cast<double *>(w);
  return ws;
}

// (Re)launch the fit over whatever keys the per-position cursors have not handed out yet; *yielded (device, in the workspace) says afterwards
// whether a wavefront stopped early because `pause` was raised.
int launch_arima_fit(hipStream_t s, Grid g, const double *sigma, const uint32_t *n_pts, int maxiter, double *calc, DevCounters *ctr, void *workspace,
                     const int *pause, const unsigned int **yielded, uint32_t grace) {
  if (yielded) *yielded = nullptr;
  if (g.K == 0 || g.T <= 3) return 0;
  const ArimaWs ws = arima_carve(g, workspace);
  const uint64_t blocks = arima_fit_blocks(g);
  if (blocks > 0x7FFFFFFFull) return -1;
  hipMemsetAsync(ws.yielded, 0, 4, s);
  hipLaunchKernelGGL(k_arima_fit, dim3((unsigned)blocks), dim3(64), 0, s, g, ws, sigma, n_pts, maxiter, (uint32_t)g.T, kArimaChunk, calc, ctr, pause, grace);
  if (yielded) *yielded = ws.yielded;
  return 0;
}

int launch_arima(hipStream_t s, Grid g, const double *sigma, const uint32_t *n_pts, int maxiter, double *calc,
                 DevCounters *ctr, void *workspace, size_t workspace_bytes, const int *pause, const unsigned int **yielded) {
  if (yielded) *yielded = nullptr;
  if (g.K == 0 || g.T == 0) return 0;
  if (workspace_bytes < arima_workspace_bytes(g)) return -1;
  const ArimaWs ws = arima_carve(g, workspace);
  hipLaunchKernelGGL(k_arima_prep, dim3((unsigned)((g.K + 255) / 256)), dim3(256), 0, s, g, ws, sigma, calc, ctr);
  if (g.T > 3) {
    const uint64_t kblocks = (g.K + 63) / 64;
    if (kblocks * (g.T - 3) > 0x7FFFFFFFull) return -1;
    hipLaunchKernelGGL(k_arima_start, dim3((unsigned)(kblocks * (g.T - 3))), dim3(64), 0, s, g, ws, n_pts, (uint32_t)g.T);
    hipMemsetAsync(ws.cursor, 0, (size_t)g.T * 8, s);
    hipMemsetAsync(ws.saved, 0, (size_t)arima_fit_blocks(g) * 4, s);
    if (launch_arima_fit(s, g, sigma, n_pts, maxiter, calc, ctr, workspace, pause, yielded, 0) != 0) return -1;
#if defined(TAD_ARIMA_PROF)
    {
      unsigned long long h[8] = {0};
      hipStreamSynchronize(s);
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_arima_prof), sizeof h);
      fprintf(stderr, "arima prof: wavefront cycles in likelihood pass %llu | optimiser step %llu | refill %llu | optimiser cycles %llu | wavefronts %llu\n",
              h[0], h[1], h[2], h[3], (unsigned long long)arima_fit_blocks(g));
      unsigned long long z[8] = {0};
      hipMemcpyToSymbol(HIP_SYMBOL(g_arima_prof), z, sizeof z);
    }
#endif
  }
  return 0;
}

// one kernel of this translation unit: tad_engine_create resolves it so that the unit's code object is loaded before the first job
const void *code_anchor_arima() { return reinterpret_cast<const void *>(&k_arima_prep); }

}  // namespace tad
