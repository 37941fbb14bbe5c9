/* arima_exact.c — TEST INFRASTRUCTURE (oracle), not product code.  Nothing under theia_amd/ may use it.
 *
 * Plain-C, one-series-at-a-time restatement of calculate_arima
 * (/root/reference/plugins/anomaly-detection/anomaly_detection.py:215-264):
 *     y, lam = scipy.stats.boxcox(x)                                        (:239)
 *     predictions[0:3] = y[0:3]                                             (:241, 255)
 *     for t in 3..n-1: ARIMA(y[:t], order=(1,1,1)).fit().forecast()[0]      (:246-253)
 *     result = inv_boxcox(predictions, lam)                                 (:256-259); any exception -> None (:260-264)
 * The arithmetic of those calls lives in statsmodels 0.14.0 / scipy 1.10.1 (plugins/anomaly-detection/requirements.txt:1-4),
 * which are not in /root/reference and not installed: this file restates their published algorithms (see
 * oracle/arima_oracle.py for the same restatement driven by the real scipy optimisers; tests/test_oracle_arima.py holds
 * the two against each other and against the reference's golden vectors).
 *
 * What this file adds over arima_oracle.py: a FIXED arithmetic.  The optimiser (L-BFGS-B on forward-difference gradients,
 * stopped at factr = 1e7) amplifies a 1-ulp difference in one likelihood value into up to 1e-4 relative in a prediction,
 * so "GPU within 1e-6 of the oracle on every point" is only attainable if both evaluate the same floating-point
 * expressions in the same order.  This file and theia_amd/csrc/tad_arima.hip are written to one arithmetic contract:
 * IEEE-754 double +, -, *, /, sqrt only, no FMA contraction (-ffp-contract=off), sums strictly left to right, and the
 * transcendental functions from the one shared source theia_amd/csrc/tad_detmath.h.  Control flow here is an ordinary
 * sequential program (the GPU runs the optimiser as a per-lane state machine); the expressions are the contract.
 * The likelihood recursion of the contract is the collapsed form (arima_nll4_collapsed = tad_arima.hip:kfc_*: the model's
 * structure used up, four recursions jointly with one division); the textbook three-state filter (arima_nll_general, round
 * 2's contract) stays here as a cross-check of the likelihood only — arima_exact_set_filter(0) selects it.
 *
 * PARITY STATUS vs the reference: "unpinned at 1e-6" (the reference's tests pin the verdict list and five leading
 * characters only, anomaly_detection_test.py:261-283, 320-345) — tests/test_oracle_arima.py records the distances.
 * Built by oracle/Makefile into oracle/_build/libarima_exact.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "tad_detmath.h"

#define DIFFUSE 1e6          /* statsmodels: initial_variance of the approximate diffuse prior */
#define CONV_TOL 1e-19       /* statsmodels: KalmanFilter.tolerance on ||P_t - P_t+1||_F^2 */
#define CONV_TOL_ABS 3.1622776601683794e-10   /* its square root: the contract's test on |p_t - p_t+1| (only p11 evolves) */
#define LOG_2PI 1.8378770664093453
#define EPSMCH 2.220446049250313e-16
#define LBFGS_M 10

static int finite_d(double v) { return fabs(v) <= 1.7976931348623157e308; }

/* ---------------------------------------------------------------------------------------------------------------
 * Box-Cox: -boxcox_llf of scipy 1.10.1 (_morestats.py), lambda by optimize.brent(brack=(-2, 2)) = bracket() + Brent
 * --------------------------------------------------------------------------------------------------------------- */
static double neg_llf(double lmb, const double *lx, long n, double sumlog) {
  double mean = 0.0, s = 0.0;
  long i;
  if (lmb == 0.0) {
    for (i = 0; i < n; ++i) mean += lx[i];
    mean /= (double)n;
    for (i = 0; i < n; ++i) { const double d = lx[i] - mean; s += d * d; }
  } else {
    for (i = 0; i < n; ++i) mean += tad_det_exp(lmb * lx[i]) / lmb;     /* x**lmb / lmb */
    mean /= (double)n;
    for (i = 0; i < n; ++i) { const double d = tad_det_exp(lmb * lx[i]) / lmb - mean; s += d * d; }
  }
  return -((lmb - 1.0) * sumlog - (double)n / 2.0 * tad_det_log(s / (double)n));
}

static int boxcox_lambda(const double *lx, long n, double sumlog, double *lam) {
#define FN(l) neg_llf((l), lx, n, sumlog)
  /* scipy.optimize.bracket(xa=-2, xb=2) */
  const double gold = 1.618034, verysmall = 1e-21, grow = 110.0;
  double xa = -2.0, xb = 2.0, fa = FN(xa), fb = FN(xb), xc, fc, tmp;
  int iter = 0;
  if (fa < fb) { tmp = xa; xa = xb; xb = tmp; tmp = fa; fa = fb; fb = tmp; }
  xc = xb + gold * (xb - xa);
  fc = FN(xc);
  while (fc < fb) {
    const double tmp1 = (xb - xa) * (fb - fc);
    const double tmp2 = (xb - xc) * (fb - fa);
    const double val = tmp2 - tmp1;
    const double denom = fabs(val) < verysmall ? 2.0 * verysmall : 2.0 * val;
    double w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
    const double wlim = xb + grow * (xc - xb);
    double fw;
    if (iter > 1000) return 0;
    iter++;
    if ((w - xc) * (xb - w) > 0.0) {
      fw = FN(w);
      if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
      else if (fw > fb) { xc = w; fc = fw; break; }
      w = xc + gold * (xc - xb);
      fw = FN(w);
    } else if ((w - wlim) * (wlim - xc) >= 0.0) {
      w = wlim;
      fw = FN(w);
    } else if ((w - wlim) * (xc - w) > 0.0) {
      fw = FN(w);
      if (fw < fc) {
        xb = xc; xc = w; w = xc + gold * (xc - xb);
        fb = fc; fc = fw; fw = FN(w);
      }
    } else {
      w = xc + gold * (xc - xb);
      fw = FN(w);
    }
    xa = xb; xb = xc; xc = w;
    fa = fb; fb = fc; fc = fw;
  }
  if (!(((fb < fc && fb <= fa) || (fb < fa && fb <= fc)) && ((xa < xb && xb < xc) || (xc < xb && xb < xa)) &&
        finite_d(xa) && finite_d(xb) && finite_d(xc)))
    return 0;                                         /* scipy raises BracketError -> calculate_arima returns None */
  {
    /* scipy.optimize Brent.optimize(), tol 1.48e-8, maxiter 500 */
    const double tol = 1.48e-8, mintol = 1.0e-11, cg = 0.3819660;
    double x = xb, w = xb, v = xb, fx = fb, fw = fb, fv = fb;
    double a = xa < xc ? xa : xc, b = xa < xc ? xc : xa;
    double deltax = 0.0, rat = 0.0;
    int it;
    for (it = 0; it < 500; ++it) {
      const double tol1 = tol * fabs(x) + mintol, tol2 = 2.0 * tol1, xmid = 0.5 * (a + b);
      double u, fu;
      if (fabs(x - xmid) < (tol2 - 0.5 * (b - a))) break;
      if (fabs(deltax) <= tol1) {
        deltax = x >= xmid ? a - x : b - x;
        rat = cg * deltax;
      } else {
        double tmp1 = (x - w) * (fx - fv);
        double tmp2 = (x - v) * (fx - fw);
        double p = (x - v) * tmp2 - (x - w) * tmp1;
        double dx_temp;
        tmp2 = 2.0 * (tmp2 - tmp1);
        if (tmp2 > 0.0) p = -p;
        tmp2 = fabs(tmp2);
        dx_temp = deltax;
        deltax = rat;
        if (p > tmp2 * (a - x) && p < tmp2 * (b - x) && fabs(p) < fabs(0.5 * tmp2 * dx_temp)) {
          rat = p * 1.0 / tmp2;
          u = x + rat;
          if ((u - a) < tol2 || (b - u) < tol2) rat = xmid - x >= 0 ? tol1 : -tol1;
        } else {
          deltax = x >= xmid ? a - x : b - x;
          rat = cg * deltax;
        }
      }
      u = fabs(rat) < tol1 ? (rat >= 0 ? x + tol1 : x - tol1) : x + rat;
      fu = FN(u);
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
    *lam = x;
    return finite_d(x);
  }
#undef FN
}

static double inv_boxcox(double y, double lam) {       /* scipy.special.inv_boxcox */
  return lam == 0.0 ? tad_det_exp(y) : tad_det_exp(tad_det_log1p(lam * y) / lam);
}

/* ---------------------------------------------------------------------------------------------------------------
 * -loglike / nobs of ARIMA(1,1,1) in statsmodels' state-space form and the one-step forecast.  The arithmetic
 * textbook three-state recursion (round 2's contract; cross-check only) written as ONE loop over the series with the
 * covariance update skipped once converged.
 * --------------------------------------------------------------------------------------------------------------- */
static long long g_steps;   /* filter time-steps executed (reported next to the GPU's kalman_steps counter) */
static double *g_trace; static long g_trace_cap, g_trace_n;   /* debugging: every evaluation (x[3], f) of a traced fit */
static int g_filter = 1;    /* 1: the contract the engine runs (arima_nll4_collapsed); 0: the textbook three-state form, kept to check that
                              * both are the same likelihood (tests/test_oracle_arima.py) */

static double nll_finish(const double u[3], double prod, int esum, long nconv, double F, double q, long n) {
  double sumlog = tad_det_log(prod) + (double)esum * TAD_DM_LN2, llf;
  if (nconv) sumlog += (double)nconv * tad_det_log(F);
  llf = -0.5 * ((double)(n - 1) * LOG_2PI + sumlog) - 0.5 * q;
  if (g_trace && g_trace_n < g_trace_cap) { double *r = g_trace + 4 * g_trace_n++; r[0] = u[0]; r[1] = u[1]; r[2] = u[2]; r[3] = -llf / (double)n; }
  return -llf / (double)n;
}

static double arima_nll_general(const double u[3], const double *y, long n, double *forecast) {
  const double phi = -(u[0] / sqrt(1.0 + u[0] * u[0]));     /* statsmodels: constrain_stationary_univariate returns -r, */
  const double theta = u[1] / sqrt(1.0 + u[1] * u[1]);       /* SARIMAX.transform_params negates it again for the MA block */
  const double s2 = u[2] * u[2];
  const double q11 = s2, q12 = s2 * theta, q22 = s2 * (theta * theta);
  double p00 = DIFFUSE, p01 = 0.0;
  double p11 = s2 * (1.0 + theta * theta + 2.0 * phi * theta) / (1.0 - phi * phi);
  double a0 = 0.0, a1 = 0.0, F = 1.0, rF = 1.0, pz0 = 0.0, pz1 = 0.0, prod = 1.0, q = 0.0;
  int esum = 0, conv = 0;
  long nconv = 0, t;
  for (t = 0; t < n; ++t) {
    const double v = y[t] - (a0 + a1);
    double w, f0, f1, f2;
    if (!conv) {
      pz0 = p00 + p01; pz1 = p01 + p11;
      F = pz0 + pz1;
      rF = 1.0 / F;
    }
    w = rF * v;
    if (t >= 1) {
      q += v * w;
      if (!conv) { int e; prod = tad_det_frexp(prod * F, &e); esum += e; }
      else nconv++;
    }
    f0 = a0 + pz0 * w; f1 = a1 + pz1 * w; f2 = q12 * w;
    a0 = f0 + f1;
    a1 = phi * f1 + f2;
    if (!conv) {
      const double g0 = pz0 * rF, g1 = pz1 * rF, g2 = q12 * rF;
      const double c00 = p00 - g0 * pz0, c01 = p01 - g0 * pz1, c02 = -(g0 * q12);
      const double c11 = p11 - g1 * pz1, c12 = q12 - g1 * q12, c22 = q22 - g2 * q12;
      const double n00 = c00 + 2.0 * c01 + c11;
      const double n01 = phi * (c01 + c11) + (c02 + c12);
      const double n11 = phi * (phi * c11 + c12) + (phi * c12 + c22) + q11;
      const double d00 = p00 - n00, d01 = p01 - n01, d11 = p11 - n11;
      const double dsq = d00 * d00 + 2.0 * (d01 * d01) + d11 * d11;
      conv = dsq < CONV_TOL;
      p00 = n00; p01 = n01; p11 = n11;
    }
  }
  g_steps += n;
  if (forecast) *forecast = a0 + a1;
  return nll_finish(u, prod, esum, nconv, F, q, n);
}

/* The same likelihood with the structure of the model used up (contract "collapsed", tad_arima.hip:kfc_*).  The
 * observation equation has no noise (H = 0), so the filtered covariance C_t = P_t - P_t Z' F^-1 Z P_t has Z' in its null
 * space: C Z' = P Z' - P Z' (Z P Z') / F = 0.  Row 0 of T is Z, hence after EVERY update (T C T')_00 = (T C T')_01 = 0: from
 * t = 1 on the predicted covariance is zero except p11 (and the constant q12, q22), the level state is known exactly
 * (a0_t = y_t-1) and the three-state filter is the innovations recursion of the ARMA(1,1) on the differences:
 *     v = (y_t - y_t-1) - a1;  F = p;  g = q12 / F;  a1' = phi (a1 + v) + g v;  p' = (q11 + q22) - q12 g
 * In the general form the same zeros are computed as differences of numbers of size 1e6 (the diffuse prior) and carry
 * rounding residue of ~1e-10 into F; the likelihood is the same function of the parameters in exact arithmetic.  The
 * t = 0 step (approximate diffuse prior, a = 0) is the general update written out with p00 = 1e6, p01 = 0.
 * The optimiser always needs the objective at x and at the three forward-difference points together, so the contract
 * evaluates the FOUR recursions jointly and takes the four reciprocals 1 / F from ONE division (batched inversion:
 * inv = 1 / (F0 F1 F2 F3), r0 = inv (F2 F3) F1, ...): IEEE operations in a fixed order like everything else, a quarter
 * of the divisions.  Round 3: the three multiply-adds of a chain are IEEE fma (q, a1, p'), the running product of the F_t is
 * renormalised after every fourth step (t & 3 == 0), the convergence test is |p_t - p_t+1| < sqrt(1e-19) — expression for
 * expression what tad_arima.hip:kfc_step4 executes. */
static void arima_nll4_collapsed(const double u4[4][3], const double *y, long n, double nll[4], double *forecast) {
  double phi[4], q12[4], qs[4], p[4], a1[4], prod[4], q[4], yprev = 0.0;
  int esum[4], c;
  long t;
  for (c = 0; c < 4; ++c) {
    const double ph = -(u4[c][0] / sqrt(1.0 + u4[c][0] * u4[c][0]));
    const double theta = u4[c][1] / sqrt(1.0 + u4[c][1] * u4[c][1]);
    const double s2 = u4[c][2] * u4[c][2];
    const double q11 = s2, q12c = s2 * theta, q22 = s2 * (theta * theta);
    const double p11 = s2 * (1.0 + theta * theta + 2.0 * ph * theta) / (1.0 - ph * ph);
    const double F0 = DIFFUSE + p11, r0 = 1.0 / F0;
    const double m = DIFFUSE * (p11 * r0), c12 = DIFFUSE * (q12c * r0), c22 = q22 - (q12c * r0) * q12c;
    phi[c] = ph; q12[c] = q12c; qs[c] = q11 + q22;
    p[c] = ph * (ph * m + c12) + (ph * c12 + c22) + q11;          /* predicted p11 for t = 1 */
    a1[c] = 0.0; prod[c] = 1.0; q[c] = 0.0;
    esum[c] = 0;
    if (n >= 1) {                                                 /* t = 0: burned (loglikelihood_burn = 1) */
      const double w0 = r0 * y[0];
      a1[c] = ph * (p11 * w0) + q12c * w0;
    }
  }
  if (n >= 1) yprev = y[0];
  for (t = 1; t < n; ++t) {
    const double d = y[t] - yprev;
    const int renorm = (t & 3) == 0;      /* the product is renormalised after every fourth step (exact scaling by 2^-e) */
    double r[4];
    {                                     /* the four reciprocals 1 / F_t (F_t = p_t) from ONE division */
      const double t12 = p[0] * p[1], t34 = p[2] * p[3];
      const double inv = 1.0 / (t12 * t34);
      const double i12 = inv * t34, i34 = inv * t12;
      r[0] = i12 * p[1]; r[1] = i12 * p[0]; r[2] = i34 * p[3]; r[3] = i34 * p[2];
    }
    for (c = 0; c < 4; ++c) {
      const double v = d - a1[c], g = q12[c] * r[c], w = r[c] * v;
      double pn;
      q[c] = fma(v, w, q[c]);
      prod[c] = prod[c] * p[c];
      if (renorm) { int e; prod[c] = tad_det_frexp(prod[c], &e); esum[c] += e; }
      a1[c] = fma(g, v, phi[c] * (a1[c] + v));
      pn = fma(-q12[c], g, qs[c]);
      /* covariance frozen from the step that detects convergence on (statsmodels reuses that step's F): p moves only by
       * steps of at least the tolerance; its log keeps entering the product, 1 / F and the gain are recomputed from the
       * frozen p, so the test keeps holding (no flag, no second code path) */
      p[c] = fabs(p[c] - pn) < CONV_TOL_ABS ? p[c] : pn;
    }
    yprev = y[t];
  }
  g_steps += 4 * n;
  for (c = 0; c < 4; ++c) nll[c] = nll_finish(u4[c], prod[c], esum[c], 0, 1.0, q[c], n);
  if (forecast) *forecast = yprev + a1[0];
}

/* one evaluation (unit tests of the pieces): the joint recursion with four times the same parameters */
static double arima_nll(const double u[3], const double *y, long n, double *forecast) {
  if (g_filter) {
    double u4[4][3], nll[4];
    int c;
    for (c = 0; c < 4; ++c) { u4[c][0] = u[0]; u4[c][1] = u[1]; u4[c][2] = u[2]; }
    arima_nll4_collapsed(u4, y, n, nll, forecast);
    g_steps -= 3 * n;
    return nll[0];
  }
  return arima_nll_general(u, y, n, forecast);
}

/* ---------------------------------------------------------------------------------------------------------------
 * start parameters: SARIMAX.start_params -> _conditional_sum_squares(k_ar = 1, k_ma = 1) on diff(y); the two
 * numpy.linalg.pinv(X).dot(Y) (rcond 1e-15) are 2-column minimum-norm least squares, solved by one Jacobi rotation
 * of the Gram matrix and a second pass over the rows for the rotated column norms / projections.
 * --------------------------------------------------------------------------------------------------------------- */
typedef void (*row_fn)(const void *ctx, long i, double *c1, double *c2, double *yy);

static void pinv2(long rows, row_fn row, const void *ctx, double *ra, double *rb) {
  double g11 = 0.0, g12 = 0.0, g22 = 0.0, cs = 1.0, sn = 0.0, s1 = 0.0, s2 = 0.0, b1 = 0.0, b2 = 0.0;
  double c1, c2, yy, smax, cut, w1, w2;
  long i;
  *ra = 0.0; *rb = 0.0;
  if (rows == 0) return;
  for (i = 0; i < rows; ++i) { row(ctx, i, &c1, &c2, &yy); g11 += c1 * c1; g12 += c1 * c2; g22 += c2 * c2; }
  if (g12 != 0.0) {
    const double zeta = (g22 - g11) / (2.0 * g12);
    const double tn = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    cs = 1.0 / sqrt(1.0 + tn * tn);
    sn = cs * tn;
  }
  for (i = 0; i < rows; ++i) {
    double r1, r2;
    row(ctx, i, &c1, &c2, &yy);
    r1 = cs * c1 - sn * c2; r2 = sn * c1 + cs * c2;
    s1 += r1 * r1; s2 += r2 * r2; b1 += r1 * yy; b2 += r2 * yy;
  }
  smax = sqrt(fmax(s1, s2));
  cut = 1e-15 * smax;
  w1 = sqrt(s1) > cut ? b1 / s1 : 0.0;
  w2 = sqrt(s2) > cut ? b2 / s2 : 0.0;
  *ra = cs * w1 + sn * w2;
  *rb = -sn * w1 + cs * w2;
}

struct sp_ctx { const double *y; double ar_a, ar_b; };
static double dif(const struct sp_ctx *c, long i) { return c->y[i + 1] - c->y[i]; }                 /* diff(y)[i] */
static double ar_res(const struct sp_ctx *c, long j) {                                           /* residual of row t = j + 2 */
  return dif(c, j + 2) - (dif(c, j + 1) * c->ar_a + dif(c, j) * c->ar_b);
}
static void row_ar2(const void *ctx, long i, double *c1, double *c2, double *yy) {               /* e_t on (e_t-1, e_t-2) */
  const struct sp_ctx *c = (const struct sp_ctx *)ctx;
  *c1 = dif(c, i + 1); *c2 = dif(c, i); *yy = dif(c, i + 2);
}
static void row_arma(const void *ctx, long i, double *c1, double *c2, double *yy) {              /* e_t on (e_t-1, res_t-1) */
  const struct sp_ctx *c = (const struct sp_ctx *)ctx;
  *c1 = dif(c, i + 2); *c2 = ar_res(c, i); *yy = dif(c, i + 3);
}

static void start_params(const double *y, long n, double u[3]) {
  const long m = n - 1;                               /* number of first differences */
  struct sp_ctx c;
  double phi0 = 0.0, theta0 = 0.0, var0, mean = 0.0, s = 0.0;
  long i;
  c.y = y; c.ar_a = 0.0; c.ar_b = 0.0;
  if (!(m <= 2 || m - 2 <= 1)) {                      /* else: lagmat raises ValueError -> zeros */
    const long rows = m - 3;
    double am_a, am_b;
    pinv2(m - 2, row_ar2, &c, &c.ar_a, &c.ar_b);
    pinv2(rows, row_arma, &c, &am_a, &am_b);
    phi0 = am_a; theta0 = am_b;
    if (rows > 1) {                                   /* mean(residuals[1:] ** 2) */
      for (i = 1; i < rows; ++i) { const double r2 = dif(&c, i + 3) - (dif(&c, i + 2) * am_a + ar_res(&c, i) * am_b); s += r2 * r2; }
      var0 = s / (double)(rows - 1);
    } else {                                          /* numpy.var(endog) */
      for (i = 0; i < m; ++i) mean += dif(&c, i);
      mean /= (double)m;
      for (i = 0; i < m; ++i) { const double d = dif(&c, i) - mean; s += d * d; }
      var0 = s / (double)m;
    }
  } else {                                            /* residuals = [0, 0, e - mean(e)] -> mean(residuals[1:] ** 2) */
    for (i = 0; i < m; ++i) mean += dif(&c, i);
    mean /= (double)m;
    for (i = 0; i < m; ++i) { const double d = dif(&c, i) - mean; s += d * d; }
    var0 = s / (double)(m + 1);
  }
  if (!(fabs(phi0) < 1.0)) phi0 = 0.0;
  if (!(fabs(theta0) < 1.0)) theta0 = 0.0;
  var0 = fmax(var0, 1e-10);
  u[0] = -phi0 / sqrt(1.0 - phi0 * phi0);             /* SARIMAX.untransform_params: unconstrain_stationary_univariate(phi), */
  u[1] = theta0 / sqrt(1.0 - theta0 * theta0);        /* unconstrain_stationary_univariate(-theta) */
  u[2] = sqrt(var0);
}

/* ---------------------------------------------------------------------------------------------------------------
 * More'-Thuente line search (MINPACK-2 dcsrch / dcstep as used by L-BFGS-B 3.0): ftol 1e-3, gtol 0.9, xtol 0.1
 * --------------------------------------------------------------------------------------------------------------- */
struct mt {
  double stx, fx, gx, sty, fy, gy, stmin, stmax, width, width1, finit, ginit, gtest;
  int brackt, stage;
};
enum { T_FG = 0, T_CONV = 1, T_WARN = 2, T_ERROR = 3 };

static void dcstep(double *stx, double *fx, double *dx, double *sty, double *fy, double *dy, double *stp, double fp, double dp,
                   int *brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (*dx / fabs(*dx));
  double stpf;
  if (fp > *fx) {
    const double theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    double p, q, r, stpc, stpq;
    if (*stp < *stx) gamma = -gamma;
    p = (gamma - *dx) + theta; q = ((gamma - *dx) + gamma) + dp; r = p / q;
    stpc = *stx + r * (*stp - *stx);
    stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
    stpf = fabs(stpc - *stx) <= fabs(stpq - *stx) ? stpc : stpc + (stpq - stpc) / 2.0;
    *brackt = 1;
  } else if (sgnd < 0.0) {
    const double theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    double p, q, r, stpc, stpq;
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + *dx; r = p / q;
    stpc = *stp + r * (*stx - *stp);
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    stpf = fabs(stpc - *stp) > fabs(stpq - *stp) ? stpc : stpq;
    *brackt = 1;
  } else if (fabs(dp) < fabs(*dx)) {
    const double theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    double gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
    double p, q, r, stpc, stpq;
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta; q = (gamma + (*dx - dp)) + gamma; r = p / q;
    if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
    else if (*stp > *stx) stpc = stpmax;
    else stpc = stpmin;
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (*brackt) {
      stpf = fabs(stpc - *stp) < fabs(stpq - *stp) ? stpc : stpq;
      if (*stp > *stx) stpf = fmin(*stp + 0.66 * (*sty - *stp), stpf);
      else stpf = fmax(*stp + 0.66 * (*sty - *stp), stpf);
    } else {
      stpf = fabs(stpc - *stp) > fabs(stpq - *stp) ? stpc : stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (*brackt) {
      const double theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
      const double s = fmax(fabs(theta), fmax(fabs(*dy), fabs(dp)));
      double gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
      double p, q, r;
      if (*stp > *sty) gamma = -gamma;
      p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + *dy; r = p / q;
      stpf = *stp + r * (*sty - *stp);
    } else if (*stp > *stx) stpf = stpmax;
    else stpf = stpmin;
  }
  if (fp > *fx) { *sty = *stp; *fy = fp; *dy = dp; }
  else {
    if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
    *stx = *stp; *fx = fp; *dx = dp;
  }
  *stp = stpf;
}

static int dcsrch_start(struct mt *L, double stp, double f, double g, double stpmin, double stpmax) {
  if (stp < stpmin || stp > stpmax || g >= 0.0) return T_ERROR;
  L->brackt = 0; L->stage = 1; L->finit = f; L->ginit = g; L->gtest = 1e-3 * g;
  L->width = stpmax - stpmin; L->width1 = L->width / 0.5;
  L->stx = 0.0; L->fx = f; L->gx = g; L->sty = 0.0; L->fy = f; L->gy = g;
  L->stmin = 0.0; L->stmax = stp + 4.0 * stp;
  return T_FG;
}

static int dcsrch_next(struct mt *L, double *stp, double f, double g, double stpmin, double stpmax) {
  const double gtol = 0.9, xtol = 0.1;
  const double ftest = L->finit + *stp * L->gtest;
  int task = T_FG;
  if (L->stage == 1 && f <= ftest && g >= 0.0) L->stage = 2;
  if (L->brackt && (*stp <= L->stmin || *stp >= L->stmax)) task = T_WARN;
  if (L->brackt && L->stmax - L->stmin <= xtol * L->stmax) task = T_WARN;
  if (*stp == stpmax && f <= ftest && g <= L->gtest) task = T_WARN;
  if (*stp == stpmin && (f > ftest || g >= L->gtest)) task = T_WARN;
  if (f <= ftest && fabs(g) <= gtol * (-L->ginit)) task = T_CONV;
  if (task != T_FG) return task;
  if (L->stage == 1 && f <= L->fx && f > ftest) {
    const double fm = f - *stp * L->gtest, gm = g - L->gtest;
    double fxm = L->fx - L->stx * L->gtest, fym = L->fy - L->sty * L->gtest;
    double gxm = L->gx - L->gtest, gym = L->gy - L->gtest;
    dcstep(&L->stx, &fxm, &gxm, &L->sty, &fym, &gym, stp, fm, gm, &L->brackt, L->stmin, L->stmax);
    L->fx = fxm + L->stx * L->gtest; L->fy = fym + L->sty * L->gtest;
    L->gx = gxm + L->gtest; L->gy = gym + L->gtest;
  } else {
    dcstep(&L->stx, &L->fx, &L->gx, &L->sty, &L->fy, &L->gy, stp, f, g, &L->brackt, L->stmin, L->stmax);
  }
  if (L->brackt) {
    if (fabs(L->sty - L->stx) >= 0.66 * L->width1) *stp = L->stx + 0.5 * (L->sty - L->stx);
    L->width1 = L->width;
    L->width = fabs(L->sty - L->stx);
  }
  if (L->brackt) { L->stmin = fmin(L->stx, L->sty); L->stmax = fmax(L->stx, L->sty); }
  else { L->stmin = *stp + 1.1 * (*stp - L->stx); L->stmax = *stp + 4.0 * (*stp - L->stx); }
  *stp = fmax(*stp, stpmin);
  *stp = fmin(*stp, stpmax);
  if ((L->brackt && (*stp <= L->stmin || *stp >= L->stmax)) || (L->brackt && L->stmax - L->stmin <= xtol * L->stmax)) *stp = L->stx;
  return T_FG;
}

/* ---------------------------------------------------------------------------------------------------------------
 * scipy.optimize.fmin_l_bfgs_b(approx_grad=True, epsilon=1e-5, m=10, factr=1e7, pgtol=1e-5, maxiter, maxls=20), no
 * bounds: L-BFGS-B 3.0's unconstrained path — every variable free, so the generalised Cauchy point followed by the
 * subspace minimisation is the quasi-Newton step -H g with the compact L-BFGS matrix, evaluated here by the two-loop
 * recursion with H0 = I / theta, theta = y'y / s'y.
 * --------------------------------------------------------------------------------------------------------------- */
struct fitctx { const double *y; long n; };

/* f, the forward-difference gradient (_approx_fprime) and the one-step forecast of the model at x */
static void eval_fg(const struct fitctx *c, const double x[3], double *f, double g[3], double *fc) {
  double u4[4][3], dx[3], nll[4];
  int i, j;
  for (i = 0; i < 4; ++i) for (j = 0; j < 3; ++j) u4[i][j] = x[j];
  for (i = 0; i < 3; ++i) {
    const double x0 = x[i];
    u4[i + 1][i] = x0 + 1e-5;
    dx[i] = u4[i + 1][i] - x0;
    if (dx[i] == 0.0) {   /* scipy _numdiff.approx_derivative: an absolute step that does not change x falls back to the relative step */
      const double h = 1.4901161193847656e-08 * (x0 >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(x0));
      u4[i + 1][i] = x0 + h;
      dx[i] = u4[i + 1][i] - x0;
    }
  }
  if (g_filter) arima_nll4_collapsed(u4, c->y, c->n, nll, fc);   /* the four recursions jointly (batched inversion) */
  else {
    nll[0] = arima_nll_general(u4[0], c->y, c->n, fc);
    for (i = 1; i < 4; ++i) nll[i] = arima_nll_general(u4[i], c->y, c->n, 0);
  }
  for (i = 0; i < 3; ++i) g[i] = (nll[i + 1] - nll[0]) / dx[i];
  *f = nll[0];
}

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

struct hist { double S[LBFGS_M][3], Y[LBFGS_M][3]; int col, head; double theta; };

static void two_loop(const struct hist *h, const double g[3], double d[3]) {
  double q[3] = {g[0], g[1], g[2]}, alpha[LBFGS_M];
  int j, c;
  if (h->col == 0) { for (c = 0; c < 3; ++c) d[c] = -g[c]; return; }
  for (j = h->col - 1; j >= 0; --j) {
    const int i = (h->head + j) % LBFGS_M;
    const double sy = dot3(h->S[i], h->Y[i]);
    alpha[j] = dot3(h->S[i], q) / sy;
    for (c = 0; c < 3; ++c) q[c] -= alpha[j] * h->Y[i][c];
  }
  for (c = 0; c < 3; ++c) q[c] /= h->theta;
  for (j = 0; j < h->col; ++j) {
    const int i = (h->head + j) % LBFGS_M;
    const double sy = dot3(h->S[i], h->Y[i]);
    const double beta = dot3(h->Y[i], q) / sy;
    for (c = 0; c < 3; ++c) q[c] += h->S[i][c] * (alpha[j] - beta);
  }
  for (c = 0; c < 3; ++c) d[c] = -q[c];
}

/* returns the one-step forecast of the fitted model; *nit_out = iterations */
static double fit_forecast(const double *y, long n, int maxiter, int *nit_out) {
  const double pgtol = 1e-5, factr = 1e7;
  struct fitctx c;
  struct hist h;
  struct mt L;
  double x[3], g[3], f, d[3], xs[3], gs[3], fold, gd, gdold, stp;
  double fc, fcold;   /* forecast at x / at the start of the line search: the fitted model's forecast needs no extra filter run */
  int iter = 0, nit = 0, i;
  c.y = y; c.n = n;
  h.col = 0; h.head = 0; h.theta = 1.0;
  start_params(y, n, x);
  eval_fg(&c, x, &f, g, &fc);
  fcold = fc;
  if (fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))) <= pgtol) goto done;
  for (;;) {
    int task, ifun, restart = 0;
    /* search direction; an ascent direction (or a bad first step) refreshes the memory once, then gives up */
    for (;;) {
      double dnorm;
      two_loop(&h, g, d);
      dnorm = sqrt(dot3(d, d));
      stp = iter == 0 ? fmin(1.0 / dnorm, 1e10) : 1.0;
      for (i = 0; i < 3; ++i) { xs[i] = x[i]; gs[i] = g[i]; }
      fold = f;
      fcold = fc;
      gd = dot3(g, d);
      gdold = gd;
      task = T_ERROR;
      if (gd < 0.0) task = dcsrch_start(&L, stp, f, gd, 0.0, 1e10);
      if (task == T_FG) break;
      if (h.col == 0) goto done;                      /* ABNORMAL_TERMINATION_IN_LNSRCH */
      h.col = 0; h.head = 0; h.theta = 1.0;
    }
    ifun = 1;
    for (i = 0; i < 3; ++i) x[i] = stp == 1.0 ? xs[i] + d[i] : stp * d[i] + xs[i];
    for (;;) {                                        /* line search */
      eval_fg(&c, x, &f, g, &fc);
      gd = dot3(g, d);
      task = dcsrch_next(&L, &stp, f, gd, 0.0, 1e10);
      if (task != T_FG) break;
      ifun++;
      if (ifun - 1 >= 20) {                           /* maxls: back to the start of the search, memory refreshed */
        for (i = 0; i < 3; ++i) { x[i] = xs[i]; g[i] = gs[i]; }
        f = fold;
        fc = fcold;
        if (h.col == 0) goto done;
        h.col = 0; h.head = 0; h.theta = 1.0;
        restart = 1;
        break;
      }
      for (i = 0; i < 3; ++i) x[i] = stp == 1.0 ? xs[i] + d[i] : stp * d[i] + xs[i];
    }
    if (restart) continue;
    iter++; nit++;
    if (nit >= maxiter) break;                        /* scipy's driver: n_iterations >= maxiter -> STOP */
    if (fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))) <= pgtol) break;
    if ((fold - f) <= EPSMCH * factr * fmax(fabs(fold), fmax(fabs(f), 1.0))) break;
    {                                                 /* matupd: s = stp d, y = g - g_old, theta = y'y / s'y */
      double r[3], rr = 0.0, dr, ddum;
      for (i = 0; i < 3; ++i) { r[i] = g[i] - gs[i]; rr += r[i] * r[i]; }
      if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
      else { dr = (gd - gdold) * stp; for (i = 0; i < 3; ++i) d[i] *= stp; ddum = -gdold * stp; }
      if (!(dr <= EPSMCH * ddum)) {
        int slot;
        if (h.col < LBFGS_M) { slot = (h.head + h.col) % LBFGS_M; h.col++; }
        else { slot = h.head; h.head = (h.head + 1) % LBFGS_M; }
        for (i = 0; i < 3; ++i) { h.S[slot][i] = d[i]; h.Y[slot][i] = r[i]; }
        h.theta = rr / dr;
      }
    }
  }
done:
  if (nit_out) *nit_out = nit;
  return fc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * exported entry points (ctypes)
 * --------------------------------------------------------------------------------------------------------------- */
/* calculate_arima on one series of n throughput values (already float(x), i.e. correctly rounded from UInt64).
 * Returns 1 and fills pred[n], or 0 = the reference returns None.  info[0] = lambda, info[1] = filter time-steps,
 * info[2] = fits, info[3] = optimiser iterations. */
int arima_exact_series(const double *x, long n, int maxiter, double *pred, double *info) {
  double *lx, *y, lam = 0.0, sumlog = 0.0, steps0 = (double)g_steps;
  long i, iters = 0;
  int ok = n > 3;
  if (info) { info[0] = 0.0; info[1] = 0.0; info[2] = 0.0; info[3] = 0.0; }
  if (!ok) return 0;                                  /* :232-234 */
  for (i = 0; i < n; ++i) if (!(x[i] > 0.0)) ok = 0;  /* stats.boxcox: "Data must be positive." */
  if (ok) { ok = 0; for (i = 1; i < n; ++i) if (x[i] != x[0]) ok = 1; }   /* "Data must not be constant." */
  if (!ok) return 0;
  lx = (double *)malloc(sizeof(double) * (size_t)n * 2);
  if (!lx) return -1;
  y = lx + n;
  for (i = 0; i < n; ++i) { lx[i] = tad_det_log(x[i]); sumlog += lx[i]; }
  if (!boxcox_lambda(lx, n, sumlog, &lam)) { free(lx); return 0; }
  for (i = 0; i < n; ++i) y[i] = lam == 0.0 ? lx[i] : tad_det_expm1(lam * lx[i]) / lam;   /* scipy.special.boxcox */
  for (i = 0; i < 3; ++i) pred[i] = inv_boxcox(y[i], lam);
  for (i = 3; i < n; ++i) {
    int nit = 0;
    pred[i] = inv_boxcox(fit_forecast(y, i, maxiter, &nit), lam);
    iters += nit;
  }
  if (info) { info[0] = lam; info[1] = (double)g_steps - steps0; info[2] = (double)(n - 3); info[3] = (double)iters; }
  free(lx);
  return 1;
}

/* which likelihood recursion the fits use: 1 = the contract (collapsed form, default), 0 = textbook three-state form */
void arima_exact_set_filter(int collapsed) { g_filter = collapsed != 0; }
int arima_exact_get_filter(void) { return g_filter; }

/* pieces, for the unit tests */
double arima_exact_nll(const double *y, long n, double u0, double u1, double u2, double *forecast) {
  const double u[3] = {u0, u1, u2};
  return arima_nll(u, y, n, forecast);
}
void arima_exact_start_params(const double *y, long n, double *u) { start_params(y, n, u); }
double arima_exact_fit_forecast(const double *y, long n, int maxiter) { return fit_forecast(y, n, maxiter, 0); }
int arima_exact_boxcox_lambda(const double *x, long n, double *lam) {
  double *lx = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)), sumlog = 0.0;
  long i;
  int ok;
  if (!lx) return -1;
  for (i = 0; i < n; ++i) { lx[i] = tad_det_log(x[i]); sumlog += lx[i]; }
  ok = boxcox_lambda(lx, n, sumlog, lam);
  free(lx);
  return ok;
}
double arima_exact_log(double x) { return tad_det_log(x); }
double arima_exact_exp(double x) { return tad_det_exp(x); }
double arima_exact_expm1(double x) { return tad_det_expm1(x); }
double arima_exact_log1p(double x) { return tad_det_log1p(x); }

/* per-fit work profile of one series (load-balance studies): evals[i] = likelihood evaluations of the fit on y[:i], i >= 3 */
int arima_exact_fit_profile(const double *x, long n, int maxiter, double *evals) {
  double *lx, *y, lam = 0.0, sumlog = 0.0;
  long i;
  if (n <= 3) return 0;
  for (i = 0; i < n; ++i) if (!(x[i] > 0.0)) return 0;
  lx = (double *)malloc(sizeof(double) * (size_t)n * 2);
  if (!lx) return -1;
  y = lx + n;
  for (i = 0; i < n; ++i) { lx[i] = tad_det_log(x[i]); sumlog += lx[i]; }
  if (!boxcox_lambda(lx, n, sumlog, &lam)) { free(lx); return 0; }
  for (i = 0; i < n; ++i) y[i] = lam == 0.0 ? lx[i] : tad_det_expm1(lam * lx[i]) / lam;
  for (i = 0; i < 3; ++i) evals[i] = 0.0;
  for (i = 3; i < n; ++i) {
    const long long s0 = g_steps;
    fit_forecast(y, i, maxiter, 0);
    evals[i] = (double)(g_steps - s0) / (double)i;
  }
  free(lx);
  return 1;
}
double arima_exact_frexp(double x, int *e) { return tad_det_frexp(x, e); }

/* debugging: the evaluations of one fit, rows of (x0, x1, x2, f) */
long arima_exact_trace_fit(const double *y, long n, int maxiter, double *buf, long cap, double *forecast) {
  g_trace = buf; g_trace_cap = cap; g_trace_n = 0;
  *forecast = fit_forecast(y, n, maxiter, 0);
  g_trace = 0;
  return g_trace_n;
}
