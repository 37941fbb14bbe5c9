"""bench.py's HOST logic against the emulated library (development aid): torch.cuda is stubbed out, "device" tensors are host
tensors, so that an edit of bench.py (argument handling, the JSON line, the multi-config blocks) can be exercised without a GPU.
The numbers it prints mean nothing.  Usage, from the repo root after `python tools/hipemu/build.py`:
    python tools/hipemu/bench_on_host.py --config c3 --rows 20000 --keys 40 --buckets 60 --steps 1 --warmup 0 --no-cpu-baseline"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
os.environ["TAD_LIBRARY_PATH"] = os.path.join(HERE, "_build", "libtad_hipemu.so")

import torch  # noqa: E402

torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.device_count = lambda: 1


class HostDevice:
    """stands in for torch.device("cuda", i): index 0, tensors created "on" it live on the host"""
    index = 0
    type = "cpu"

    def __init__(self, *a, **k):
        pass


def _host(kw):
    if isinstance(kw.get("device"), HostDevice):
        kw["device"] = "cpu"
    return kw


torch.device = HostDevice
for name in ("empty", "as_tensor", "tensor", "zeros"):
    orig = getattr(torch, name)
    setattr(torch, name, (lambda f: lambda *a, **k: f(*a, **_host(k)))(orig))

sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
os.chdir(ROOT)
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
