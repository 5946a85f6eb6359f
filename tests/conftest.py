import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mlx_audio_amd import _lib

    # GPU tests must exercise the HIP library, never a fallback.
    assert _lib.HAS_HIP_EXT, "libmia_hip.so is not built/loaded"
    return torch.device("cuda", 0)
