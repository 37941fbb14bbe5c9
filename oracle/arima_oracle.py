"""ARIMA(1,1,1) walk-forward oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates calculate_arima (/root/reference/plugins/anomaly-detection/anomaly_detection.py:215-264):

    y, lam = scipy.stats.boxcox(x)                       (:239)
    first 3 predictions = y[0:3]                         (:241, 255)
    for t in 3..n-1: ARIMA(y[:t], order=(1,1,1)).fit().forecast()[0]   (:246-253)
    result = inv_boxcox(train + predictions, lam)        (:256-259);  any exception -> None (:260-264)

The arithmetic lives in third-party libraries that are NOT in /root/reference and are not
installed here: statsmodels==0.14.0 (plugins/anomaly-detection/requirements.txt:3) on
scipy==1.10.1 (:2).  This file restates the published algorithm of statsmodels' SARIMAX/ARIMA
(state-space form, approximate-diffuse + stationary initialisation, loglikelihood_burn = 1,
Hannan-Rissanen-style conditional-sum-of-squares start parameters, stationarity/invertibility
transforms, L-BFGS-B with forward-difference gradients eps=1e-5, m=10, factr=1e7, pgtol=1e-5,
maxiter=50) and of scipy.stats.boxcox (MLE lambda by Brent on boxcox_llf), and drives the REAL
scipy optimisers installed here (scipy 1.15.3: optimize.fmin_l_bfgs_b, optimize.brent).

PARITY STATUS — "unpinned at 1e-6": the reference's own tests pin only (i) the first five characters
of each prediction (anomaly_detection_test.py:261-283) and (ii) the verdict list (:320-345); the
full-precision list in the same file (:288-318) is never asserted and disagrees with (i) at 12 of
90 indices.  tests/test_oracle_arima.py checks this restatement against all three and records the
distances; the GPU path is then held to this oracle.
"""
import ctypes
import math
import os

import numpy as np
from scipy import optimize, special

DIFFUSE_VAR = 1e6          # statsmodels ssm.initial_variance (approximate diffuse prior of the level state)
CONV_TOL = 1e-19           # statsmodels KalmanFilter.tolerance: ||P_t - P_{t+1}||_F^2 below this freezes P, F, K
LOG_2PI = math.log(2.0 * math.pi)


# ------------------------------------------------------------------------------------------------
# Box-Cox (scipy.stats.boxcox with lmbda=None -> MLE; scipy 1.10.1 _morestats.py)
# ------------------------------------------------------------------------------------------------
def boxcox_llf(lmb, data, logdata=None):
    """boxcox_llf of scipy 1.10.1: (lmb-1)*sum(log x) - N/2*log(var(x**lmb/lmb)) (population variance)."""
    n = data.shape[0]
    if logdata is None:
        logdata = np.log(data)
    if lmb == 0:
        variance = np.var(logdata)
    else:
        variance = np.var(data ** lmb / lmb)
    return (lmb - 1) * np.sum(logdata) - n / 2 * np.log(variance)


def boxcox_mle_lambda(x):
    logx = np.log(x)
    with np.errstate(all="ignore"):
        return float(optimize.brent(lambda lmb: -boxcox_llf(lmb, x, logx), brack=(-2.0, 2.0)))


def boxcox_transform(x, lam):
    return special.boxcox(x, lam)


# ------------------------------------------------------------------------------------------------
# ARIMA(1,1,1) state space (statsmodels SARIMAX with simple_differencing=False):
#   Z = [1 1 0], T = [[1 1 0],[0 phi 1],[0 0 0]], R = [0 1 theta]', Q = sigma2, H = 0
# ------------------------------------------------------------------------------------------------
def transform_params(u):
    """unconstrained -> (phi, theta, sigma2): SARIMAX.transform_params with enforce_stationarity /
    enforce_invertibility (constrain_stationary_univariate for one lag) and variance = u**2."""
    phi = -(u[0] / math.sqrt(1.0 + u[0] * u[0]))
    theta = u[1] / math.sqrt(1.0 + u[1] * u[1])
    return phi, theta, u[2] * u[2]


def untransform_params(phi, theta, sigma2):
    return np.array([-phi / math.sqrt(1.0 - phi * phi), theta / math.sqrt(1.0 - theta * theta), math.sqrt(sigma2)])


def kalman_arima111(y, phi, theta, sigma2, counters=None):
    """Conventional Kalman filter, generic matrix form.  Returns (sum of loglike_obs[1:], forecast).

    Initialisation: a_0 = 0; P_0 = blockdiag(1e6, stationary covariance of the ARMA block).
    """
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    Z = np.array([1.0, 1.0, 0.0])
    T = np.array([[1.0, 1.0, 0.0], [0.0, phi, 1.0], [0.0, 0.0, 0.0]])
    R = np.array([0.0, 1.0, theta])
    RQR = sigma2 * np.outer(R, R)
    # stationary covariance of s1_t = phi s1 + s2 + eps, s2_t = theta eps
    p22 = theta * theta * sigma2
    p12 = theta * sigma2
    p11 = sigma2 * (1.0 + theta * theta + 2.0 * phi * theta) / (1.0 - phi * phi)
    P = np.zeros((3, 3))
    P[0, 0] = DIFFUSE_VAR
    P[1, 1], P[1, 2], P[2, 1], P[2, 2] = p11, p12, p12, p22
    a = np.zeros(3)
    llf = 0.0
    converged = False
    F = K_gain = None
    for t in range(n):
        v = y[t] - Z @ a
        if not converged:
            PZ = P @ Z
            F = Z @ PZ
            K_gain = PZ / F
        if t >= 1:
            llf += -0.5 * (LOG_2PI + math.log(F)) - 0.5 * v * v / F
        a_f = a + K_gain * v
        a = T @ a_f
        if not converged:
            P_f = P - np.outer(K_gain, PZ)
            P_next = T @ P_f @ T.T + RQR
            d = P - P_next
            if float(np.sum(d * d)) < CONV_TOL:
                converged = True
            P = P_next
    if counters is not None:
        counters["kalman_steps"] = counters.get("kalman_steps", 0) + n
    return llf, float(Z @ a)


_clib = None


def _load_c_kalman():
    """Optional C restatement of the same filter (oracle/arima_kalman.c), ~100x faster than the numpy
    loop; used when built (oracle/Makefile).  tests/test_oracle_arima.py checks it against the numpy one."""
    global _clib
    if _clib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libarima_kalman.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.arima111_filter.restype = ctypes.c_double
            lib.arima111_filter.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
            _clib = lib
        else:
            _clib = False
    return _clib


def kalman_fast(y, phi, theta, sigma2, counters=None):
    lib = _load_c_kalman()
    if not lib:
        return kalman_arima111(y, phi, theta, sigma2, counters)
    fc = ctypes.c_double()
    llf = lib.arima111_filter(y.ctypes.data, y.size, phi, theta, sigma2, ctypes.byref(fc))
    if counters is not None:
        counters["kalman_steps"] = counters.get("kalman_steps", 0) + y.size
    return llf, fc.value


# ------------------------------------------------------------------------------------------------
# start parameters: SARIMAX.start_params / _conditional_sum_squares for k_ar = k_ma = 1 on diff(y)
# ------------------------------------------------------------------------------------------------
def _lagmat_first_lag_forward(x):
    """statsmodels lagmat(x, 1) (trim='forward')[:, 0]: row t holds x[t-1], row 0 holds 0; raises when
    maxlag >= nobs like the original."""
    if 1 >= x.size:
        raise ValueError("maxlag should be < nobs")
    out = np.zeros(x.size)
    out[1:] = x[:-1]
    return out


def start_params(y):
    e = np.diff(np.asarray(y, dtype=np.float64))        # simple_differencing=False: start params use diff(endog)
    k = 2                                                # 2 * k_ma
    r = 3                                                # max(k + k_ma, k_ar)
    try:
        if 2 >= e.size:
            raise ValueError("maxlag should be < nobs")  # lagmat(endog, 2, trim='both')
        Y = e[k:]
        X = np.column_stack([e[1:-1], e[:-2]])           # [e_{t-1}, e_{t-2}] for t = 2..
        params_ar = np.linalg.pinv(X).dot(Y)
        residuals = Y - X.dot(params_ar)
        Y = e[r:]
        x_ar = _lagmat_first_lag_forward(e)[r:]
        x_ma = _lagmat_first_lag_forward(residuals)[r - k:]
        X = np.column_stack([x_ar, x_ma])
        params = np.linalg.pinv(X).dot(Y)
        residuals = Y - X.dot(params)
    except ValueError:
        params = np.zeros(2)
        residuals = np.r_[np.zeros(2), e - np.mean(e)] if e.size else np.ones(3)
    phi0, theta0 = float(params[0]), float(params[1])
    if residuals.size > 1:
        var0 = float((residuals[1:] ** 2).mean())
    else:
        var0 = float(np.var(e))
    if not abs(phi0) < 1.0:      # non-stationary start -> zeros (is_invertible of a first-order polynomial)
        phi0 = 0.0
    if not abs(theta0) < 1.0:    # non-invertible start -> zeros
        theta0 = 0.0
    var0 = max(var0, 1e-10)
    return phi0, theta0, var0


# ------------------------------------------------------------------------------------------------
# one fit + forecast, and the walk-forward of calculate_arima
# ------------------------------------------------------------------------------------------------
def fit_forecast(history, maxiter=50, counters=None, kalman=kalman_fast):
    """ARIMA(history, order=(1,1,1)).fit().forecast()[0] on Box-Cox data."""
    h = np.ascontiguousarray(history, dtype=np.float64)
    nobs = h.size
    x0 = untransform_params(*start_params(h))

    def objective(u):
        phi, theta, s2 = transform_params(u)
        llf, _ = kalman(h, phi, theta, s2, counters)
        return -llf / nobs

    with np.errstate(all="ignore"):
        xopt, fval, info = optimize.fmin_l_bfgs_b(objective, x0, approx_grad=True, epsilon=1e-5, m=10, factr=1e7,
                                                  pgtol=1e-5, maxiter=maxiter, bounds=[(None, None)] * 3)
    phi, theta, s2 = transform_params(xopt)
    _, fc = kalman(h, phi, theta, s2, counters)
    if counters is not None:
        counters["fits"] = counters.get("fits", 0) + 1
        counters["iterations"] = counters.get("iterations", 0) + info["nit"]
        counters["funcalls"] = counters.get("funcalls", 0) + info["funcalls"]
    return fc


def calculate_arima(throughputs, maxiter=50, counters=None):
    """Returns list[float] (length n) or None, like anomaly_detection.py:215-264."""
    x = np.array([float(v) for v in throughputs], dtype=np.float64)
    n = x.size
    if n <= 3:
        return None                                  # :232-234
    if np.any(x <= 0) or np.all(x == x[0]):
        return None                                  # stats.boxcox raises ValueError -> caught at :260-264
    try:
        lam = boxcox_mle_lambda(x)
        y = boxcox_transform(x, lam)
        preds = list(y[:3])
        for t in range(3, n):
            preds.append(fit_forecast(y[:t], maxiter, counters))
        with np.errstate(all="ignore"):
            out = special.inv_boxcox(np.array(preds), lam)
        return [float(v) for v in out]
    except Exception:                                # :260-264 — any error (e.g. no valid Brent bracket) -> None
        return None


def calculate_arima_anomaly(throughput_row, stddev, maxiter=50):
    """anomaly_detection.py:267-309."""
    pred = calculate_arima(throughput_row, maxiter)
    if pred is None:
        return [False]
    if stddev is None:
        return [False] * len(pred)
    s = float(stddev)
    return [abs(float(x) - p) > s for x, p in zip(throughput_row, pred)]


# ------------------------------------------------------------------------------------------------
# the fixed-arithmetic restatement (oracle/arima_exact.c): what the GPU path is held to bit for bit
# ------------------------------------------------------------------------------------------------
_xlib = None


def _load_exact():
    global _xlib
    if _xlib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libarima_exact.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_build/libarima_exact.so is missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.arima_exact_series.restype = ctypes.c_int
        lib.arima_exact_series.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.arima_exact_nll.restype = ctypes.c_double
        lib.arima_exact_nll.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_double, ctypes.c_double, dp]
        lib.arima_exact_start_params.restype = None
        lib.arima_exact_start_params.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        lib.arima_exact_fit_forecast.restype = ctypes.c_double
        lib.arima_exact_fit_forecast.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
        lib.arima_exact_boxcox_lambda.restype = ctypes.c_int
        lib.arima_exact_boxcox_lambda.argtypes = [ctypes.c_void_p, ctypes.c_long, dp]
        for fn in ("log", "exp", "expm1", "log1p"):
            f = getattr(lib, "arima_exact_" + fn)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_double]
        lib.arima_exact_set_filter.restype = None
        lib.arima_exact_set_filter.argtypes = [ctypes.c_int]
        _xlib = lib
    return _xlib


def calculate_arima_exact(throughputs, maxiter=50, counters=None):
    """calculate_arima (anomaly_detection.py:215-264) in the fixed arithmetic of oracle/arima_exact.c.
    Returns list[float] (length n) or None."""
    lib = _load_exact()
    x = np.ascontiguousarray([float(v) for v in throughputs], dtype=np.float64)
    pred = np.empty(max(x.size, 1), dtype=np.float64)
    info = np.zeros(4, dtype=np.float64)
    rc = lib.arima_exact_series(x.ctypes.data, x.size, int(maxiter), pred.ctypes.data, info.ctypes.data)
    if rc < 0:
        raise MemoryError("arima_exact_series")
    if counters is not None and rc == 1:
        counters["lambda"] = float(info[0])
        counters["kalman_steps"] = counters.get("kalman_steps", 0) + int(info[1])
        counters["fits"] = counters.get("fits", 0) + int(info[2])
        counters["iterations"] = counters.get("iterations", 0) + int(info[3])
    return [float(v) for v in pred[:x.size]] if rc == 1 else None


def calculate_arima_anomaly_exact(throughput_row, stddev, maxiter=50):
    """anomaly_detection.py:267-309 on top of calculate_arima_exact."""
    pred = calculate_arima_exact(throughput_row, maxiter)
    if pred is None:
        return [False]
    if stddev is None:
        return [False] * len(pred)
    s = float(stddev)
    return [abs(float(x) - p) > s for x, p in zip(throughput_row, pred)]


def kalman_exact(y, phi, theta, sigma2, counters=None):
    """(loglike with burn 1, forecast) from the fixed-arithmetic filter, signature of kalman_arima111."""
    lib = _load_exact()
    y = np.ascontiguousarray(y, dtype=np.float64)
    u = untransform_params(phi, theta, sigma2)
    fc = ctypes.c_double()
    nll = lib.arima_exact_nll(y.ctypes.data, y.size, float(u[0]), float(u[1]), float(u[2]), ctypes.byref(fc))
    if counters is not None:
        counters["kalman_steps"] = counters.get("kalman_steps", 0) + y.size
    return -nll * y.size, fc.value
