import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(path):
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


def rel_err(y, ref):
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def exact_layer(K, N, G, seed):
    """The seeded layer of tests/golden/gen_golden.py::_exact_layer (weights that quantise exactly), regenerated: logical
    (iw[K, N] uint8, s[K/G, N] fp16, z[K/G, N] uint8).  torch's CPU generator is deterministic for a given build, and the GPU
    box runs the same image, so the fixture only has to hold the seed."""
    import torch
    g = torch.Generator().manual_seed(seed)
    iw = torch.randint(0, 16, (N, K), generator=g)
    z = torch.randint(0, 16, (N, K // G), generator=g)
    s = (torch.rand(N, K // G, generator=g) * 0.02 + 0.005).half()
    return (np.ascontiguousarray(iw.numpy().T).astype(np.uint8), np.ascontiguousarray(s.numpy().T),
            np.ascontiguousarray(z.numpy().T).astype(np.uint8))
