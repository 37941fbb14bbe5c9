"""CPU: the HOST instantiation of the device ARIMA source against the oracle, bit for bit.

theia_amd/csrc/tad_arima.hip marks its arithmetic `__host__ __device__`; tools/arima_twin.cpp instantiates exactly those
functions for the host (hipcc compiles both halves) and drives them like k_arima_prep / k_arima_fit do.  If the twin and
oracle/arima_exact.c agree on every bit, the device SOURCE and the checker follow one arithmetic contract — what the
`-m gpu` tests then establish on the hardware is only that gfx950 executes that source the way the host does.  The reference's golden series and seeded random series.  Not a fallback: nothing in the product loads this library."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import arima_oracle as ao

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def twin(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("twin") / "libarima_twin.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-unused-result",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "theia_amd", "csrc"),
                           os.path.join(ROOT, "tools", "arima_twin.cpp"), "-o", str(out)], stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(str(out))
    lib.twin_series.restype = ctypes.c_int
    lib.twin_series.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def series_set(golden):
    out = {"golden": np.array(golden["throughput_list"], dtype=np.float64)}
    rng = np.random.default_rng(7)
    for i in range(12):
        n = int(rng.integers(4, 100))
        x = 10 ** rng.uniform(3, 10) * np.exp(rng.normal(0, rng.uniform(0.001, 0.6), n))
        if rng.random() < 0.5:
            x[rng.integers(0, n)] *= rng.uniform(2, 12)
        out["rand%d" % i] = np.floor(x) + 1.0
    out["constant"] = np.full(8, 5.0)             # -> None on both sides
    out["short"] = np.array([3.0, 4.0, 5.0])
    return out


def test_device_source_on_the_host_equals_the_oracle(twin, golden):
    for name, x in series_set(golden).items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        c = {}
        want = ao.calculate_arima_exact(x, counters=c)
        pred = np.empty(max(x.size, 1))
        info = np.zeros(4)
        rc = twin.twin_series(x.ctypes.data, x.size, 50, pred.ctypes.data, info.ctypes.data)
        if want is None:
            assert rc == 0, name
            continue
        assert rc == 1, name
        got = pred[:x.size]
        want = np.array(want)
        assert (got.view(np.uint64) == want.view(np.uint64)).all(), (name, float(np.nanmax(np.abs(got - want) / np.abs(want))))
        assert int(info[1]) == c["kalman_steps"], name      # the flop figure's counter too
