import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "comfyui-3d-pack_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import gs_oracle
    gs_oracle.build()
    return True


@pytest.fixture(scope="module", autouse=True)
def _release_cached_device_memory():
    """After every test module the caching allocator's free blocks go back to the device.  The full-size tests (tests/test_zz_baseline_1m.py) leave ~170 GB cached in the pytest
    process; the modules after them start other processes on the same GPU (eight bench.py ranks at BASELINE size, tests/test_zz_replay_gpu.py), and a box whose 288 GB were
    oversubscribed that way lost one rank to a GPU fault (round 6)."""
    yield
    import gc
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        gc.collect()
        torch.cuda.empty_cache()
