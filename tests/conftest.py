import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    # the oracle is the checker: compile it (gcc) if the prebuilt .so files are missing/stale
    from oracle import build as obuild

    obuild.build()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def assert_close(got, ref, tol=1e-4, what=""):
    """|got-ref| <= tol * max(1, |ref|)  (abs+rel: correlation values cross zero)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    worst = float(err.max()) if err.size else 0.0
    assert np.isfinite(got).all() or not np.isfinite(ref).all(), f"{what}: non-finite output"
    assert worst <= tol, f"{what}: max err {worst:.3e} > {tol:.1e} at {np.unravel_index(err.argmax(), err.shape)}"
    return worst
