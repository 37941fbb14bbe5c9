"""theia_amd — MI355X-native Throughput Anomaly Detection engine for Theia.

Product layout:
  csrc/                 hand-written HIP kernels (gfx950) + the C ABI of include/tad.h
  lib/libtad_mi355x.so  built in-tree by theia_amd.build (hipcc --offload-arch=gfx950)
  _capi.py              ctypes view of include/tad.h (what a cgo binding would declare)
  engine.py             TadEngine: columnar batches in, anomalous points out
  anomaly_detection.py  host-side mirror of the reference job's interface
                        (plugins/anomaly-detection/anomaly_detection.py)
There is no CPU fallback: every compute entry point goes through the HIP library and fails
loudly if it (or a GPU) is missing.
"""
from .engine import TadEngine, TadError, TadPoints, TadResult, TadState  # noqa: F401

__all__ = ["TadEngine", "TadError", "TadPoints", "TadResult", "TadState"]
