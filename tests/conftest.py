import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def whisper_golden_cases(n_mels):
    """Yield (name, pcm, expected (n_mels, 3000) float32) rebuilt from head + constant tail."""
    z = load_golden(f"whisper_{n_mels}.npz")
    names = sorted({k.split("__")[0] for k in z.files})
    for name in names:
        head = z[f"{name}__head"]
        full = np.full((n_mels, 3000), z[f"{name}__tail"], dtype=np.float32)
        full[:, :head.shape[1]] = head
        yield name, z[f"{name}__pcm"], full


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
