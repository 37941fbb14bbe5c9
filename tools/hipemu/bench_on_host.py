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

# engine-owned "device" arrays are host memory here: expose them through the array interface numpy / torch read on the host
sys.path.insert(0, ROOT)
from theia_amd import distributed as _td  # noqa: E402


class HostColumn:
    def __init__(self, ptr, n, typestr="<i8"):
        self.__array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


_td.DeviceColumn = HostColumn
_as_tensor = torch.as_tensor
torch.as_tensor = lambda x, *a, **k: _as_tensor(__import__("numpy").asarray(x) if isinstance(x, HostColumn) else x, *a, **k)

# ... and a host tensor IS a device column here (the engine would otherwise stage it through its host-input path)
from theia_amd import engine as _eng  # noqa: E402

_as_column = _eng._as_column


def _as_column_host_is_device(x, dtype, n_expected=None):
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and not x.is_cuda and x.is_contiguous() and x.element_size() == 8:
        return x.data_ptr(), x.numel(), True, x
    return _as_column(x, dtype, n_expected)


_eng._as_column = _as_column_host_is_device

sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
os.chdir(ROOT)
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
