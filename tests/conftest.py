# coding=utf-8
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def assert_parity(got, ref, tol=1e-5, what=""):
    """The bar of BASELINE.json / SURVEY.md §8d: |got - ref| <= 1e-5 + 1e-5 * |ref| (fp32 vs float64-accumulated oracle)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "{} shape {} vs {}".format(what, got.shape, ref.shape)
    err = np.abs(got - ref) - tol * np.abs(ref)
    worst = float(err.max()) if err.size else 0.0
    assert worst <= tol, "{}: parity violated, max(|d| - tol*|ref|) = {:.3e} > {:.1e}".format(what, worst, tol)


@pytest.fixture(scope="session")
def oracle():
    from oracle import tfg_oracle
    return tfg_oracle


@pytest.fixture(scope="session")
def tfg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tf_geometric_amd
    tf_geometric_amd._lib.require_gpu()   # fail loudly if the HIP library is not built
    return tf_geometric_amd
