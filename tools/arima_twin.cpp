// tools/arima_twin.cpp — debugging aid (not product, not oracle): the HOST instantiation of the __host__ __device__
// functions of theia_amd/csrc/tad_arima.hip, driven like k_arima_prep / k_arima_fit drive them on the GPU, exported
// for ctypes.  Used on a machine without a GPU to check that the device source and oracle/arima_exact.c follow the
// same arithmetic contract bit for bit before spending GPU time (tools/arima_twin_check.py).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fPIC -shared -Iinclude -Itheia_amd/csrc tools/arima_twin.cpp -o /tmp/libarima_twin.so
#include <vector>
#include "../theia_amd/csrc/tad_arima.hip"
using namespace tad;

static double twin_fit(const double *y, uint32_t p, int maxiter, unsigned long long *steps) {
  LbfgsLive o;
  double hist[kHistDoubles], park[kParkDoubles];
  double u[3];
  o.hist = hist; o.hstride = 1; o.park = park; o.pstride = 1;
  arima_start_params(y, 1, p, u);
  lbfgs_reset(o, u[0], u[1], u[2]);
  while (!o.done) {
    double xe[4][3], dx[3], nll[4], fc0 = 0.0, g[3];
    for (int c = 0; c < 4; ++c) { xe[c][0] = o.x[0]; xe[c][1] = o.x[1]; xe[c][2] = o.x[2]; }
    for (int i = 0; i < 3; ++i) xe[i + 1][i] = fd_point(xe[i + 1][i], &dx[i]);
    arima_nll4_collapsed(xe, y, 1, p, nll, fc0);            // the four recursions jointly (batched inversion)
    *steps += 4ull * p;
    for (int i = 0; i < 3; ++i) g[i] = (nll[i + 1] - nll[0]) / dx[i];
    lbfgs_deliver(o, nll[0], g, fc0, maxiter);
  }
  return o.fc;
}

extern "C" int twin_series(const double *x, long n, int maxiter, double *pred, double *info) {
  std::vector<double> lx(n), y(n);
  bool nonpos = false, allsame = true;
  double sumlog = 0.0;
  for (long i = 0; i < n; ++i) {
    if (x[i] != x[0]) allsame = false;
    if (!(x[i] > 0.0)) nonpos = true;
    lx[i] = tad_det_log(x[i]);
    sumlog += lx[i];
  }
  bool ok = n > 3 && !nonpos && !allsame;
  double lam = 0.0;
  if (ok) ok = bc_mle_lambda(x, lx.data(), y.data(), 1, (uint32_t)n, sumlog, &lam);   // (y: scratch for the terms of the llf until it is filled below)
  if (!ok) return 0;
  unsigned long long steps = 0;
  for (long i = 0; i < n; ++i) y[i] = lam == 0.0 ? lx[i] : tad_det_expm1(lam * lx[i]) / lam;
  for (long i = 0; i < 3; ++i) pred[i] = inv_boxcox(y[i], lam);
  for (long i = 3; i < n; ++i) pred[i] = inv_boxcox(twin_fit(y.data(), (uint32_t)i, maxiter, &steps), lam);
  if (info) { info[0] = lam; info[1] = (double)steps; }
  return 1;
}
