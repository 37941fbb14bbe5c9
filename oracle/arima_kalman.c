/* arima_kalman.c — TEST INFRASTRUCTURE (oracle), not product code.
 * Plain-C restatement of oracle/arima_oracle.py:kalman_arima111 (the conventional Kalman filter of
 * statsmodels' SARIMAX for ARIMA(1,1,1): Z=[1 1 0], T=[[1 1 0],[0 phi 1],[0 0 0]], R=[0 1 theta]',
 * approximate-diffuse (1e6) + stationary initialisation, loglikelihood_burn = 1, covariance frozen once
 * ||P_t - P_{t+1}||_F^2 < 1e-19).  Generic 3x3 loops on purpose: it must not share code or formulas
 * with theia_amd/csrc/tad_arima.hip.  Built by oracle/Makefile into oracle/_build/libarima_kalman.so. */
#include <math.h>

double arima111_filter(const double *y, long n, double phi, double theta, double sigma2, double *forecast) {
  const double Z[3] = {1.0, 1.0, 0.0};
  const double T[3][3] = {{1.0, 1.0, 0.0}, {0.0, phi, 1.0}, {0.0, 0.0, 0.0}};
  const double R[3] = {0.0, 1.0, theta};
  double RQR[3][3], P[3][3] = {{0}}, a[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) RQR[i][j] = sigma2 * (R[i] * R[j]);
  P[0][0] = 1e6;
  P[1][1] = sigma2 * (1.0 + theta * theta + 2.0 * phi * theta) / (1.0 - phi * phi);
  P[1][2] = P[2][1] = theta * sigma2;
  P[2][2] = theta * theta * sigma2;
  const double log2pi = log(2.0 * M_PI);
  double llf = 0.0, F = 0.0, K[3] = {0, 0, 0}, PZ[3] = {0, 0, 0};
  int converged = 0;
  for (long t = 0; t < n; ++t) {
    double za = 0.0;
    for (int i = 0; i < 3; ++i) za += Z[i] * a[i];
    const double v = y[t] - za;
    if (!converged) {
      for (int i = 0; i < 3; ++i) {
        PZ[i] = 0.0;
        for (int j = 0; j < 3; ++j) PZ[i] += P[i][j] * Z[j];
      }
      F = 0.0;
      for (int i = 0; i < 3; ++i) F += Z[i] * PZ[i];
      for (int i = 0; i < 3; ++i) K[i] = PZ[i] / F;
    }
    if (t >= 1) llf += -0.5 * (log2pi + log(F)) - 0.5 * v * v / F;
    double af[3], an[3];
    for (int i = 0; i < 3; ++i) af[i] = a[i] + K[i] * v;
    for (int i = 0; i < 3; ++i) {
      an[i] = 0.0;
      for (int j = 0; j < 3; ++j) an[i] += T[i][j] * af[j];
    }
    for (int i = 0; i < 3; ++i) a[i] = an[i];
    if (!converged) {
      double Pf[3][3], TP[3][3], Pn[3][3], d2 = 0.0;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Pf[i][j] = P[i][j] - K[i] * PZ[j];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          TP[i][j] = 0.0;
          for (int k = 0; k < 3; ++k) TP[i][j] += T[i][k] * Pf[k][j];
        }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          Pn[i][j] = RQR[i][j];
          for (int k = 0; k < 3; ++k) Pn[i][j] += TP[i][k] * T[j][k];
          const double d = P[i][j] - Pn[i][j];
          d2 += d * d;
        }
      if (d2 < 1e-19) converged = 1;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P[i][j] = Pn[i][j];
    }
  }
  double f = 0.0;
  for (int i = 0; i < 3; ++i) f += Z[i] * a[i];
  *forecast = f;
  return llf;
}
