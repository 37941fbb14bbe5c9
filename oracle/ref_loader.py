"""Load the reference TAD job as a module WITHOUT pyspark / statsmodels (test infrastructure).

Only usable where /root/reference exists (the build container).  Nothing under tests/ -m gpu,
bench.py or __graft_entry__.smoke() may call this at run time: the GPU box has no /root/reference.
It is used by oracle/make_golden.py to (a) lift the golden vectors out of the reference's own
test file and (b) run the reference's pure functions (calculate_ewma, calculate_ewma_anomaly,
calculate_dbscan_anomaly, generate_tad_sql_query, remove_meaningless_labels) unchanged, from where
they lie, to produce fixtures.  No reference source is copied.

The job imports pyspark and statsmodels at module top
(/root/reference/plugins/anomaly-detection/anomaly_detection.py:24,29-36); stand-in modules with
exactly the attribute names it needs are registered first.  calculate_arima cannot run this way
(it needs statsmodels.ARIMA) — see oracle/arima_oracle.py for the restatement.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("THEIA_REFERENCE", "/root/reference")
JOB_DIR = os.path.join(REF_ROOT, "plugins", "anomaly-detection")
JOB_FILE = os.path.join(JOB_DIR, "anomaly_detection.py")
TEST_FILE = os.path.join(JOB_DIR, "anomaly_detection_test.py")


def reference_available():
    return os.path.isfile(JOB_FILE)


class _Anything:
    """Callable stand-in for any pyspark / statsmodels class."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Anything()


def _install_stubs():
    names = [
        "pyspark", "pyspark.sql", "pyspark.sql.functions", "pyspark.sql.types",
        "statsmodels", "statsmodels.tsa", "statsmodels.tsa.arima", "statsmodels.tsa.arima.model",
    ]
    for n in names:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []  # behave like a package
            sys.modules[n] = m
    sys.modules["pyspark.sql"].SparkSession = _Anything()
    sys.modules["pyspark.sql"].functions = sys.modules["pyspark.sql.functions"]
    for t in ("BooleanType", "ArrayType", "StructField", "DecimalType", "DoubleType", "StringType",
              "LongType", "TimestampType", "StructType"):
        setattr(sys.modules["pyspark.sql.types"], t, _Anything)
    sys.modules["statsmodels.tsa.arima.model"].ARIMA = _Anything


def load_reference_job():
    """Returns the reference module object (functions run from /root/reference, unmodified)."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % JOB_FILE)
    _install_stubs()
    spec = importlib.util.spec_from_file_location("anomaly_detection", JOB_FILE)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["anomaly_detection"] = mod
    spec.loader.exec_module(mod)
    import logging
    logging.getLogger("anomaly_detection").setLevel(logging.CRITICAL + 1)
    return mod


def load_reference_tests():
    """Returns the reference's test module (for its golden lists and parametrize tables)."""
    load_reference_job()
    spec = importlib.util.spec_from_file_location("anomaly_detection_test", TEST_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
