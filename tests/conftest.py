import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(kind=None):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        with np.load(f, allow_pickle=False) as z:
            k = str(z["kind"])
        if kind is None or k == kind:
            out.append(f)
    return out


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    for k in ("fs", "window", "step", "mid_window", "mid_step"):
        if k in d:
            v = float(d[k])
            d[k] = int(v) if v == int(v) else v
    if "deltas" in d:
        d["deltas"] = bool(d["deltas"])
    return d


def golden_id(path):
    return os.path.splitext(os.path.basename(path))[0]


@pytest.fixture(scope="session")
def gpu_lib():
    """The loaded HIP library with a device selected; GPU tests fail loudly if it is missing."""
    from pyaudioanalysis_amd import _ffi
    lib = _ffi.lib()
    if _ffi.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a real MI355X")
    return lib
