import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  A hipcc-built test library loaded
    # BEFORE torch pulls in /opt/rocm's copy instead, the process ends up with two runtimes and the second one sees no device
    # (hipErrorNoDevice from every call) -- which made single-module runs of the drop-in caller tests fail.  Load torch first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # NVBIO_FUZZ_SEED=<k>: shift every integer seed handed to numpy's default_rng by k, i.e. rerun the whole parity suite on
    # fresh random data (golden-vector tests regenerate nothing and are unaffected)
    off = int(os.environ.get("NVBIO_FUZZ_SEED", "0"))
    if off:
        import numpy as np
        orig = np.random.default_rng
        np.random.default_rng = lambda seed=None: orig(seed + off if isinstance(seed, int) else seed)


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (no CPU fallback exists)")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_test_switches():
    """every test starts and ends on the library's default executions (the switches are process-wide integers, include/nvbio_hip.h)"""
    yield
    mod = sys.modules.get("nvbio_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        for name in mod.test_switch_names():          # the library's own list (nvbio_hip_test_switch_name)
            mod._lib.nvbio_hip_set_test_switch(name.encode(), 0)
