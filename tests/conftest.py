import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

LOSS_CASES = ["synth_24x80", "synth_oov_24x80", "synth_ties_24x80", "synth_zcross_24x80", "synth_odd_19x37",
              "real_48x160"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def crc32(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def rel_to_max(a, b) -> float:
    """max-abs difference normalised by max|b| (the gradient/metric tolerance of BASELINE.json)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def real_frames(name: str):
    """uint8 frames (F,H,W,3) -> float32 (F,3,H,W)/255, the reference's to_tensor (dataset.py:51)."""
    u8 = np.load(os.path.join(GOLDEN, name + "_frames_u8.npz"))["frames"]
    return (u8.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255)).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import ds_oracle
    ds_oracle.build()
    return ds_oracle
