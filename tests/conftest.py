import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU cases are tiny: intra-op threads only add OpenMP barrier spinning (6x slower on a busy
    # 8-vCPU container), and one thread fixes torch's CPU reduction order.  BHG_TEST_THREADS overrides.
    torch.set_num_threads(int(os.environ.get("BHG_TEST_THREADS", "1")))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _poison_the_gpu_allocator():
    """Fill ~6 GB of PyTorch's caching allocator with NaN and free it before the first GPU test, so every
    `torch.empty` the suite (or the package) does afterwards starts as NaN instead of the zeros a fresh process
    gets from the driver: a kernel that reads a buffer it was supposed to fill first shows up as a failure here
    instead of passing by luck."""
    if torch.cuda.is_available():
        junk = [torch.full((256 << 20,), float("nan"), device="cuda:0") for _ in range(6)]
        junk += [torch.full((n,), float("nan"), device="cuda:0") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 22) for _ in range(8)]
        del junk
    yield


class _DebugArms:
    """Selects measurement / test arms of libbhg through bhg_debug_set (the library reads no environment variable).  The
    method names mirror pytest's monkeypatch so the arm tables of rounds 1-3 ("BHG_MLP_PROJ": "0", ...) read as before."""

    def setenv(self, key, value):
        from betty_amd import _native

        _native.debug_set(key, int(value))

    def delenv(self, key, raising=False):
        from betty_amd import _native

        _native.debug_set(key, None)

    def reset(self):
        from betty_amd import _native

        _native.debug_reset()


@pytest.fixture
def bhg_debug():
    """Measurement / test arms live in libbhg_ab.so (the same sources, -DBHG_AB): a test that takes this fixture runs on THAT library
    — its default arm is the product's only form — and the process goes back to the product libbhg.so afterwards."""
    from betty_amd import _native

    _native.use_ab(True)
    _native.debug_reset()
    try:
        yield _DebugArms()
    finally:
        _native.debug_reset()
        _native.use_ab(False)


_GOLDEN_CACHE = {}


def load_golden(family):
    """tests/golden/<family>.npz -> (inputs dict, outputs dict keyed '<case>/<kind>/<i>')."""
    if family not in _GOLDEN_CACHE:
        blob = np.load(os.path.join(HERE, "golden", f"{family}.npz"))
        inputs = {k[3:]: blob[k] for k in blob.files if k.startswith("in/")}
        outputs = {k[4:]: blob[k] for k in blob.files if k.startswith("out/")}
        _GOLDEN_CACHE[family] = (inputs, outputs)
    return _GOLDEN_CACHE[family]


def golden_list(outputs, case_name, kind):
    out = []
    i = 0
    while f"{case_name}/{kind}/{i}" in outputs:
        out.append(outputs[f"{case_name}/{kind}/{i}"])
        i += 1
    return out


def rel_err(got, want):
    """Relative L2 error over a list of arrays, and max-abs error relative to max |want|."""
    g = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in got])
    w = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in want])
    if not np.all(np.isfinite(g)):
        return float("inf"), float("inf")
    nw = np.linalg.norm(w)
    if nw == 0.0:
        return float(np.linalg.norm(g)), float(np.abs(g).max(initial=0.0))
    return float(np.linalg.norm(g - w) / nw), float(np.abs(g - w).max() / np.abs(w).max())
