# coding=utf-8
"""The HIP path vs the REFERENCE'S OWN outputs (tests/golden/reference_cases.npz, produced by the unmodified
/root/reference/tf_geometric through tests/golden/make_golden_from_reference.py).  Every §8(a)/(f) row has a case in
tests/reference_cases.py; index work and max/min reductions are held bit-exact, floating point to 1e-5 (north_star)
unless the case documents a wider band."""
import os

import numpy as np
import pytest

from conftest import ROOT, assert_parity
import reference_cases as rc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_cases.npz")


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLDEN))


@pytest.mark.parametrize("case", [c for c in rc.CASES if c.hip is not None], ids=lambda c: c.name)
def test_hip_matches_reference(case, tfg, golden):
    g = case.inputs()
    got = case.hip(tfg, g, golden) if case.name == "layers" else case.hip(tfg, g)
    keys = [k for k in golden if k.startswith(case.name + "::")]
    assert keys
    checked = 0
    for full in keys:
        k = full[len(case.name) + 2:]
        assert k in got, "product did not produce {}".format(full)
        a, b = np.asarray(got[k]), golden[full]
        assert a.shape == b.shape, "{} shape {} vs reference {}".format(full, a.shape, b.shape)
        if k in case.exact or a.dtype.kind in "iub":
            assert np.array_equal(a, b), "{} must be bit-identical to the reference".format(full)
        else:
            assert_parity(a, b, tol=case.tol_of(k), what="hip vs reference " + full)
        checked += 1
    assert checked == len(keys)
