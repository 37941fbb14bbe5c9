"""bench.py's cpu_baseline leg (oracle/cpu_bench.py): the key-sharded multi-process run must produce the same job as the
single process (same rows, same anomaly count), and bench.cpu_baseline must return the contract's fields."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_cpu_job_equals_single_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--rows", "200000", "--keys", "200",
                          "--procs", "3"], check=True, capture_output=True, text=True, timeout=300).stdout
    r = json.loads(out.strip().splitlines()[-1])
    assert r["procs"] == 3 and r["multi_rows"] == r["rows"] == 200000
    assert r["multi_anomalies"] == r["single_anomalies"] > 0


def test_bench_cpu_baseline_fields():
    sys.path.insert(0, ROOT)
    import bench
    b = bench.cpu_baseline("EWMA", 100000, 100, 250, "svc")
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(b)
    assert b["kind"] == "port" and b["unit"] == "flow-records/s" and b["value"] > 0 and b["cores"] >= 1
    assert b["value"] == max(v for v in (b.get("single_core_value"), b.get("all_cores_value")) if v is not None)
