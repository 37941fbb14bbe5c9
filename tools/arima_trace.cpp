// tools/arima_trace.cpp — debugging aid (not product, not oracle): runs ONE ARIMA(1,1,1) fit with the
// host instantiation of the device functions of theia_amd/csrc/tad_arima.hip and prints the optimiser
// trace, to be diffed against scipy's fmin_l_bfgs_b trace (tools/arima_trace_scipy.py).
//   hipcc --offload-arch=gfx950 -O1 -ffp-contract=off -Iinclude -Itheia_amd/csrc tools/arima_trace.cpp -o /tmp/arima_trace
//   /tmp/arima_trace < series.txt      (first line n, then n transformed values)
#include <cstdio>
#include <vector>
#include "../theia_amd/csrc/tad_arima.hip"
using namespace tad;
int main(int argc, char **argv) {
  int n; if (scanf("%d", &n) != 1) return 1;
  if (argc > 1) {  // --boxcox: n raw values -> lambda by the restated bracket + Brent
    std::vector<double> x(n), lx(n); double sl = 0;
    for (int i = 0; i < n; ++i) { if (scanf("%lf", &x[i]) != 1) return 1; lx[i] = log(x[i]); sl += lx[i]; }
    double lam = 0; bool ok = bc_mle_lambda(x.data(), lx.data(), 1, n, sl, &lam);
    printf("ok %d lambda %.17g\n", (int)ok, lam);
    return 0;
  }
  std::vector<double> y(n);
  for (auto &v : y) if (scanf("%lf", &v) != 1) return 1;
  Lbfgs o; o.col = 0; o.head = 0; o.iter = 0; o.nit = 0; o.theta = 1.0; o.in_ls = false; o.done = false; o.f = 0;
  arima_start_params(y.data(), 1, n, o.x);
  printf("start %.17g %.17g %.17g\n", o.x[0], o.x[1], o.x[2]);
  int evals = 0;
  while (!o.done) {
    KfOut r = arima_nll(o.x[0], o.x[1], o.x[2], y.data(), 1, n);
    double f0 = r.nll;
    for (int i = 0; i < 3; ++i) {
      double xe[3] = {o.x[0], o.x[1], o.x[2]};
      xe[i] = xe[i] + 1e-5;
      double dx = xe[i] - o.x[i];
      o.g[i] = (arima_nll(xe[0], xe[1], xe[2], y.data(), 1, n).nll - f0) / dx;
    }
    o.f = f0; evals++;
    printf("eval %d x %.17g %.17g %.17g f %.17g g %.17g %.17g %.17g\n", evals, o.x[0], o.x[1], o.x[2], o.f, o.g[0], o.g[1], o.g[2]);
    int nit = o.nit;
    lbfgs_deliver(o, 50);
    if (o.nit != nit) printf("  iter %d done stp %.17g col %d theta %.17g\n", o.nit, o.stp, o.col, o.theta);
  }
  KfOut r = arima_nll(o.x[0], o.x[1], o.x[2], y.data(), 1, n);
  printf("final x %.17g %.17g %.17g f %.17g forecast %.17g nit %d\n", o.x[0], o.x[1], o.x[2], r.nll, r.forecast, o.nit);
  return 0;
}
