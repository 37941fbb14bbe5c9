"""CPU: accuracy of the deterministic transcendental functions (theia_amd/csrc/tad_detmath.h, the ONE source shared by
the device code and oracle/arima_exact.c) against numpy's long-double functions.  Determinism across host and device is
what the bit-identical GPU parity tests (tests/test_gpu_arima.py) demonstrate; this file pins that the functions are
also CORRECT to within a few ulp, so that sharing them cannot hide an error."""
import numpy as np
import pytest

from oracle import arima_oracle as ao

import decimal

# This file is the ONLY check of tad_detmath.h that does not share code with it (oracle/arima_exact.c includes the header), so it never
# skips: on a platform whose long double is no wider than double the reference values come from `decimal` at 50 digits instead.
EXTENDED_LONG_DOUBLE = np.finfo(np.longdouble).eps < np.finfo(np.float64).eps
_DEC = decimal.Context(prec=50)
_DEC_FN = {"log": lambda d: _DEC.ln(d), "exp": lambda d: _DEC.exp(d), "expm1": lambda d: _DEC.subtract(_DEC.exp(d), 1),
           "log1p": lambda d: _DEC.ln(_DEC.add(d, 1))}


def ulp_err_decimal(fn_name, x):
    """|got - ref| in ulps of the (double-rounded) reference, reference by decimal arithmetic at 50 digits"""
    lib = ao._load_exact()
    f = getattr(lib, "arima_exact_" + fn_name)
    out = np.empty(len(x))
    for i, v in enumerate(x):
        ref = _DEC_FN[fn_name](decimal.Decimal(float(v)))
        got = decimal.Decimal(f(float(v)))
        ulp = decimal.Decimal(float(np.spacing(abs(float(ref))))) if ref != 0 else decimal.Decimal(5e-324)
        out[i] = float(abs(got - ref) / ulp)
    return out


def ulp_err(fn_name, ref_fn, x):
    if not EXTENDED_LONG_DOUBLE:
        return ulp_err_decimal(fn_name, x)
    lib = ao._load_exact()
    f = getattr(lib, "arima_exact_" + fn_name)
    got = np.array([f(float(v)) for v in x])
    ref = ref_fn(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    return np.abs((got.astype(np.longdouble) - ref) / ulp).astype(np.float64)


@pytest.mark.parametrize("name,ref,bound,gens", [
    ("log", np.log, 1.0, [lambda r: np.exp(r.uniform(-700, 700, 20000)), lambda r: r.uniform(0.5, 2, 20000), lambda r: r.uniform(1e-10, 1e12, 20000)]),
    ("exp", np.exp, 1.0, [lambda r: r.uniform(-700, 700, 20000), lambda r: r.uniform(-2, 2, 20000), lambda r: r.uniform(-1e-3, 1e-3, 20000)]),
    ("expm1", np.expm1, 4.0, [lambda r: r.uniform(-50, 50, 20000), lambda r: r.uniform(-1, 1, 20000), lambda r: r.uniform(-1e-5, 1e-5, 20000)]),
    ("log1p", np.log1p, 4.0, [lambda r: r.uniform(-0.999, 50, 20000), lambda r: r.uniform(-1e-5, 1e-5, 20000), lambda r: np.exp(r.uniform(-40, 40, 20000))]),
])
def test_accuracy_in_ulp(name, ref, bound, gens):
    rng = np.random.default_rng(5)
    for g in gens:
        assert ulp_err(name, ref, g(rng)).max() < bound


@pytest.mark.parametrize("name,gen", [("log", lambda r: np.exp(r.uniform(-700, 700, 1500))), ("exp", lambda r: r.uniform(-700, 700, 1500)),
                                      ("expm1", lambda r: r.uniform(-1, 1, 1500)), ("log1p", lambda r: r.uniform(-0.999, 50, 1500))])
def test_accuracy_in_ulp_decimal_reference(name, gen):
    """the long-double-free leg, always run: the same bounds against 50-digit decimal arithmetic"""
    assert ulp_err_decimal(name, gen(np.random.default_rng(6))).max() < (1.0 if name in ("log", "exp") else 4.0)


def test_special_values():
    lib = ao._load_exact()
    log, exp = lib.arima_exact_log, lib.arima_exact_exp
    assert log(0.0) == -np.inf and log(-0.0) == -np.inf and np.isnan(log(-1.0)) and log(np.inf) == np.inf and np.isnan(log(np.nan))
    assert log(1.0) == 0.0 and abs(log(5e-324) - np.log(5e-324)) < 1e-12 and abs(log(2.0 ** -1022) - np.log(2.0 ** -1022)) < 1e-12
    assert exp(0.0) == 1.0 and exp(-746.0) == 0.0 and exp(710.0) == np.inf and exp(-np.inf) == 0.0 and np.isnan(exp(np.nan))
    assert exp(-745.0) == 5e-324 and abs(exp(709.7) / np.exp(709.7) - 1.0) < 1e-15
    assert lib.arima_exact_expm1(0.0) == 0.0 and lib.arima_exact_expm1(-1e3) == -1.0 and lib.arima_exact_log1p(0.0) == 0.0
    assert lib.arima_exact_log1p(-1.0) == -np.inf and np.isnan(lib.arima_exact_log1p(-2.0))


def test_frexp_semantics():
    """tad_det_frexp = C frexp (sign kept, subnormals normalised) except that inf / NaN come back with e = 0 — the
    semantics of the two gfx950 instructions the device code uses in its place."""
    import ctypes
    import math
    lib = ao._load_exact()
    lib.arima_exact_frexp.restype = ctypes.c_double
    lib.arima_exact_frexp.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(9)
    xs = list(np.exp(rng.uniform(-700, 700, 2000)) * rng.choice([-1.0, 1.0], 2000)) + [0.0, -0.0, 1.0, -1.0, 0.5, 5e-324, -3e-310, 2.0 ** -1022, 1.7976931348623157e308]
    for x in xs:
        e = ctypes.c_int()
        m = lib.arima_exact_frexp(float(x), ctypes.byref(e))
        assert (m, e.value) == math.frexp(float(x)), x
        assert math.copysign(1.0, m) == math.copysign(1.0, x)
    for x in (np.inf, -np.inf):
        e = ctypes.c_int(7)
        assert lib.arima_exact_frexp(x, ctypes.byref(e)) == x and e.value == 0
    e = ctypes.c_int(7)
    assert np.isnan(lib.arima_exact_frexp(np.nan, ctypes.byref(e))) and e.value == 0
