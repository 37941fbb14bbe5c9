import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Golden vectors of the reference's own unit tests (tests/golden/reference_golden.json)."""
    with open(os.path.join(GOLDEN_DIR, "reference_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ref_outputs():
    """Outputs of the reference's pure functions on seeded series (oracle/make_golden.py)."""
    with open(os.path.join(GOLDEN_DIR, "reference_outputs.json")) as f:
        d = json.load(f)
    for e in d["series"].values():
        e["x"] = [int(v) for v in e["x"]]
    return d


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """On a box WITHOUT a GPU every `gpu` test is skipped, whether it takes the `engine` fixture or builds its own engine / C driver
    (round-4 advisor finding: three of them failed with 'no HIP device available' when the suite ran unfiltered on a CPU box).  On a GPU
    box nothing is skipped here: a missing library or device stays a failure."""
    if _gpu_present() or os.environ.get("TAD_LIBRARY_PATH"):      # (a library build named explicitly through TAD_LIBRARY_PATH is the caller's business)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through the C ABI.  On a GPU box a missing library/device is a FAILURE, never a skip."""
    from theia_amd import TadEngine, TadError
    try:
        eng = TadEngine(device=0)
    except (TadError, OSError) as exc:
        if not _gpu_present():
            pytest.skip("no GPU in this container: %s" % exc)
        raise
    yield eng
    eng.close()


def sort_rows(d):
    """Canonical (key, t) order for comparing row sets."""
    order = np.lexsort((d["flow_end_s"], d["key_id"]))
    return {k: np.asarray(v)[order] for k, v in d.items() if isinstance(v, np.ndarray) and np.asarray(v).shape[:1] == order.shape}
